"""Applies that read SEVERAL inputs at non-zero offsets (the shape of the reference's shallow-water
residuals: h and q both at +-1) on the march kernel: one register ring per halo input, shared radii,
one LDS exchange and barrier per plane step.  Every case is lowered from NeptuneIR text and compared
bit for bit with the oracle; NEPTUNE_HIP_VARIANT / NEPTUNE_HIP_CHUNK force each default tile and put
chunk seams inside the field."""
import os

import numpy as np
import pytest

import helpers
from helpers import bits_equal, mismatch_report, oracle

pytestmark = pytest.mark.gpu

# exactly representable, all different, so a swapped operand or ring slot changes bits
COEF = [0.5, -0.25, 1.5, 0.125, -0.75, 2.0, -1.25, 0.375, 0.0625, -3.0, 1.75, 0.3125, -0.4375, 2.5, -0.1875, 0.875,
        1.125, -2.25, 0.6875, -0.5625, 3.5, -1.375, 0.21875, 0.9375, -1.625, 2.75, -0.3125, 1.0625]
COEF = COEF + [c * 0.5 + 0.03125 for c in COEF]   # 49 taps for a radius-8 star: dyadic and distinct
COEF = COEF + [c * 0.25 - 0.015625 for c in COEF] + [0.75 - c * 0.125 for c in COEF]   # 168: a 5x5x5 box has 125 taps


def module_text(shape, elem, nin, accesses, lb, ub):
    """one nonlinear_opdef @resid holding one apply; the body is sum_i COEF[i] * access_i evaluated left to
    right, plus an index-dependent term so a shifted tile origin is visible"""
    rank = len(shape)
    dims = "x".join("?" * rank)
    zeros = ", ".join("0" * rank)
    ubs = ", ".join(str(n) for n in shape)
    tys = ", ".join(["!t"] * nin)
    idx = ", ".join(f"%i{d}: index" for d in range(rank))
    ins = ", ".join(f"%a{k}: !t" for k in range(nin))
    L = [f'#l = #neptune_ir.location<"cell">',
         f"#b = #neptune_ir.bounds<lb = [{zeros}], ub = [{ubs}]>",
         f"!t = !neptune_ir.temp<element = {elem}, bounds = #b, location = #l>",
         f"!f = !neptune_ir.field<element = {elem}, bounds = #b, location = #l>",
         "module {",
         f"  neptune_ir.nonlinear_opdef @resid : ({tys}) -> !t {{",
         "  ^bb0(" + ", ".join(f"%u{k}: !t" for k in range(nin)) + "):",
         "    %r = neptune_ir.apply(" + ", ".join(f"%u{k}" for k in range(nin)) + ") attributes {bounds = "
         f"#neptune_ir.bounds<lb = [{', '.join(map(str, lb))}], ub = [{', '.join(map(str, ub))}]>}} : ({tys}) -> !t {{",
         f"      ^bb0({idx}, {ins}):"]
    for i, (k, off) in enumerate(accesses):
        L.append(f"        %v{i} = neptune_ir.access %a{k}[{', '.join(map(str, off))}] : !t -> {elem}")
    for i in range(len(accesses)):
        L.append(f"        %c{i} = arith.constant {COEF[i]!r} : {elem}")
        L.append(f"        %m{i} = arith.mulf %c{i}, %v{i} : {elem}")
        if i == 0:
            L.append(f"        %s0 = arith.addf %m0, %m0 : {elem}")
        else:
            L.append(f"        %s{i} = arith.addf %s{i - 1}, %m{i} : {elem}")
    n = len(accesses) - 1
    last = f"%i{rank - 1}"
    L += [f"        %w = arith.index_cast {last} : index to i64",
          f"        %wf = arith.sitofp %w : i64 to {elem}",
          f"        %o = arith.addf %s{n}, %wf : {elem}",
          f"        neptune_ir.yield %o : {elem}",
          "    }",
          "    neptune_ir.return %r : !t",
          "  }",
          f"  func.func @entry(%out: memref<{dims}x{elem}>, " + ", ".join(f"%in{k}: memref<{dims}x{elem}>" for k in range(nin))
          + f") -> memref<{dims}x{elem}> {{",
          f"    %fo = neptune_ir.wrap %out : memref<{dims}x{elem}> -> !f"]
    for k in range(nin):
        L.append(f"    %f{k} = neptune_ir.wrap %in{k} : memref<{dims}x{elem}> -> !f")
        L.append(f"    %t{k} = neptune_ir.load %f{k} : !f -> !t")
    L += ["    %y = neptune_ir.apply_nonlinear @resid(" + ", ".join(f"%t{k}" for k in range(nin)) + f") : ({tys}) -> !t",
          "    neptune_ir.store %y to %fo : !t to !f",
          f"    %res = neptune_ir.unwrap %fo : !f -> memref<{dims}x{elem}>",
          f"    func.return %res : memref<{dims}x{elem}>",
          "  }", "}"]
    return "\n".join(L) + "\n"


def star(rank, r=1):
    out = [tuple([0] * rank)]
    for d in range(rank):
        for s in range(1, r + 1):
            for sign in (-1, 1):
                o = [0] * rank
                o[d] = sign * s
                out.append(tuple(o))
    return out


CASES = {
    # name: (shape, elem, nin, accesses, bounds margin, expected kernel)
    "swe3d_two_stars": ((13, 19, 256), "f64", 2, [(0, o) for o in star(3)] + [(1, o) for o in star(3)], 1, "march"),
    "point0_then_two_stars": ((11, 10, 128), "f64", 3,
                              [(0, (0, 0, 0))] + [(1, o) for o in star(3)] + [(2, o) for o in star(3)[1:]], 1, "march"),
    "unequal_radii": ((9, 12, 128), "f64", 2,
                      [(0, (0, 0, 0)), (0, (-1, 0, 0)), (0, (1, 0, 0)), (1, (0, -1, 0)), (1, (0, 1, 0)), (1, (0, 0, 1)), (1, (0, 0, -1))],
                      1, "march"),
    "two_boxes_f32": ((10, 13, 256), "f32", 2,
                      [(0, (0, 0, 0)), (0, (-1, -1, -1)), (0, (1, 1, 1)), (0, (1, -1, 0)), (1, (-1, 1, 1)), (1, (1, -1, -1)),
                       (1, (0, 1, -1)), (1, (-1, 0, 1)), (1, (0, 0, 0))], 1, "march"),
    "swe2d": ((40, 512), "f64", 2, [(0, o) for o in star(2)] + [(1, o) for o in star(2)], 1, "march"),
    "four_halo_inputs_2d": ((24, 256), "f64", 4, [(k, o) for k in range(4) for o in star(2)], 1, "march"),
    "radius2_pair_2d": ((30, 256), "f64", 2, [(0, o) for o in star(2, 2)] + [(1, o) for o in star(2, 2)[1:]], 2, "march"),
    # three and four halo inputs of radius 2: the LDS exchange of the 8-wave tile would need 192-256 KiB (> 160 KiB of
    # a CU); the kernel then takes its halo rows from global memory instead (found by tools/soak_fuzz.py, seeds 1038/1044)
    "three_radius2_2d": ((26, 256), "f64", 3, [(k, o) for k in range(3) for o in star(2, 2)], 2, "march"),
    "four_radius2_2d_f32": ((22, 512), "f32", 4, [(k, o) for k in range(4) for o in star(2, 2)[:7]], 2, "march"),
    # high-order stars in 1-D / 2-D: radius 3-4 (6th / 8th-order operators).  K neighbours beyond one lane vector (fp64: 2 cells)
    # come through a second wave shift; rows of radius 4 through the LDS exchange of the 4-wave tile
    "radius4_2d": ((30, 384), "f64", 1, [(0, o) for o in star(2, 4)], 4, "march"),
    "radius3_2d_ragged": ((25, 261), "f64", 1, [(0, o) for o in star(2, 3)], 3, "march"),
    "radius4_2d_f32": ((28, 512), "f32", 1, [(0, o) for o in star(2, 4)], 4, "march"),
    "radius4_1d": ((5000,), "f64", 1, [(0, o) for o in star(1, 4)], 4, "march"),
    "radius8_1d_f32": ((3000,), "f32", 1, [(0, o) for o in star(1, 8)], 8, "march"),
    "radius3_pair_2d": ((26, 256), "f64", 2, [(0, o) for o in star(2, 3)] + [(1, o) for o in star(2, 3)[1:]], 3, "march"),
    "radius5_2d": ((30, 256), "f64", 1, [(0, o) for o in star(2, 5)], 5, "march"),
    # 2-D footprints beyond the march kernel's registers: the window of a tile in LDS (neptune_apply_tile2)
    "radius8_2d_f32_ragged": ((47, 771), "f32", 1, [(0, o) for o in star(2, 8)], 8, "march"),
    "box49_2d": ((40, 258), "f64", 1, [(0, (a, b)) for a in range(-3, 4) for b in range(-3, 4)], 3, "march"),
    "box_radius4_sparse_2d_f32": ((35, 512), "f32", 2, [(1, (a, b)) for a in (-4, -1, 0, 3) for b in (-4, 0, 2, 4)] + [(0, (0, 0))], 4, "march"),
    "four_radius3_2d": ((33, 384), "f64", 4, [(k, o) for k in range(4) for o in star(2, 3)[k:9 + k]], 3, "march"),
    "pair_1d": ((4096,), "f64", 2, [(0, (0,)), (0, (-1,)), (0, (1,)), (1, (1,)), (1, (-1,)), (1, (0,))], 1, "march"),
    "pair_1d_f32_r2": ((2048,), "f32", 2, [(0, (0,)), (0, (-2,)), (1, (2,)), (1, (-1,))], 2, "march"),
    # radius 2 with two halo inputs in 3-D exceeds the register budget: the lowering picks the direct kernel
    # 2-D 25-point box (5x5 window): K halos for every ring row, two-row LDS exchange with corners
    "box25_2d": ((30, 256), "f64", 1, [(0, (a, b)) for a in range(-2, 3) for b in range(-2, 3)], 2, "march"),
    "box25_2d_f32_ragged": ((29, 515), "f32", 1, [(0, (a, b)) for a in range(-2, 3) for b in range(-2, 3) if (a + b) % 3], 2, "march"),
    "box25_pair_2d": ((20, 256), "f64", 2, [(k, (a, b)) for k in range(2) for a in (-2, 0, 2) for b in (-2, 1)], 2, "march"),
    "radius3_3d": ((14, 18, 256), "f64", 1, [(0, o) for o in star(3, 3)], 3, "march"),
    "radius3_3d_f32_ragged": ((13, 17, 261), "f32", 1, [(0, o) for o in star(3, 3)], 3, "march"),
    "radius4_3d": ((12, 14, 128), "f64", 1, [(0, o) for o in star(3, 4)], 4, "march"),
    "radius4_3d_f32": ((13, 12, 260), "f32", 1, [(0, o) for o in star(3, 4)], 4, "march"),
    "radius2_pair_3d": ((9, 10, 128), "f64", 2, [(0, o) for o in star(3, 2)] + [(1, o) for o in star(3, 2)[1:]], 2, "march"),
    # several inputs read at offsets beyond what the march kernel's registers hold: a ring and an LDS window each in the plane kernel
    "radius4_pair_3d": ((12, 37, 256), "f64", 2, [(0, o) for o in star(3, 4)[:13]] + [(1, o) for o in star(3, 4)[1:] if o[0] == 0 or abs(o[0]) == 4], 4, "march"),
    "radius2_triple_3d_f32": ((9, 20, 515), "f32", 3, [(k, o) for k in range(3) for o in star(3, 2)[k:7 + 2 * k]], 2, "march"),
    "four_stars_3d": ((7, 36, 128), "f64", 4, [(k, o) for k in range(4) for o in star(3, 1)[k % 2:]], 1, "march"),
    "point0_two_wide_stars_3d": ((11, 12, 130), "f64", 3, [(0, (0, 0, 0))] + [(1, o) for o in star(3, 3)] + [(2, o) for o in star(3, 2)[1:]], 3, "march"),
    "radius5_pair_3d": ((14, 15, 128), "f64", 2, [(0, o) for o in star(3, 5)] + [(1, o) for o in star(3, 1)], 5, "direct"),
    # the plane-in-LDS kernel (apply_plane.hpp): 3-D stars of one halo input, radius 3-4 by default and up to radius 8 on
    # every tile (beyond radius 4 nothing else holds the ring); a second input read at the centre (leapfrog schemes);
    # unequal radii per axis; a window two waves wide is reached through the full variant list only
    "radius2_3d": ((10, 21, 256), "f64", 1, [(0, o) for o in star(3, 2)], 2, "march"),
    "radius5_3d": ((15, 41, 256), "f64", 1, [(0, o) for o in star(3, 5)], 5, "march"),
    "radius6_3d_f32_ragged": ((17, 37, 263), "f32", 1, [(0, o) for o in star(3, 6)], 6, "march"),
    "radius8_3d": ((21, 23, 130), "f64", 1, [(0, o) for o in star(3, 8)], 8, "march"),
    "radius8_3d_f32": ((20, 22, 512), "f32", 1, [(0, o) for o in star(3, 8) if o[2] % 3 != 1], 8, "march"),
    "radius4_3d_leapfrog": ((12, 35, 256), "f64", 2, [(0, o) for o in star(3, 4)] + [(1, (0, 0, 0))], 4, "march"),
    "radius4_3d_point0": ((13, 14, 128), "f64", 2, [(0, (0, 0, 0))] + [(1, o) for o in star(3, 4)], 4, "march"),
    "unequal_radii_3d": ((11, 30, 256), "f64", 1,
                         [(0, (0, 0, 0)), (0, (-2, 0, 0)), (0, (1, 0, 0)), (0, (0, -5, 0)), (0, (0, 3, 0)), (0, (0, 0, 1)), (0, (0, 0, -3))],
                         5, "march"),
    # 3-D boxes of radius 2 (every live plane in LDS: neptune_apply_planes); the march kernel's registers stop at 27 points
    "box125_3d_f32": ((9, 20, 256), "f32", 1, [(0, (a, b, c)) for a in range(-2, 3) for b in range(-2, 3) for c in range(-2, 3)], 2, "march"),
    "box_radius2_sparse_3d": ((8, 13, 130), "f64", 1,
                              [(0, (a, b, c)) for a in (-2, 0, 1) for b in (-2, -1, 2) for c in (-1, 0, 2)] + [(0, (0, 0, 0))], 2, "march"),
    "radius3_jk_only_3d": ((7, 21, 384), "f32", 1, [(0, o) for o in star(3, 3) if o[0] == 0], 3, "march"),
}


@pytest.fixture(scope="module")
def env(built_libs, tmp_path_factory):
    import torch
    assert torch.cuda.is_available()
    os.environ["NEPTUNE_CACHE_DIR"] = str(tmp_path_factory.mktemp("neptune_cache_mh"))
    from neptune_hip import lowering
    helpers.prefetch_modules([case_text(name)[0] for name in CASES])   # every case's module, compiled side by side
    return lowering, torch


def case_text(name):
    shape, elem, nin, accesses, margin, kernel = CASES[name]
    rank = len(shape)
    lb = [margin] * rank
    ub = [n - margin for n in shape]
    if rank == 3:
        lb[0], ub[1] = margin + 1, shape[1] - margin - 2      # bounds tighter than the halo margin: more copy-through
    return module_text(shape, elem, nin, accesses, lb, ub), lb, ub


@pytest.fixture
def launch_env():
    saved = {k: os.environ.get(k) for k in ("NEPTUNE_HIP_KERNEL", "NEPTUNE_HIP_VARIANT", "NEPTUNE_HIP_CHUNK")}
    yield
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize("name", list(CASES))
def test_several_halo_inputs_match_the_oracle(env, launch_env, name):
    lowering, torch = env
    shape, elem, nin, accesses, margin, kernel = CASES[name]
    rank = len(shape)
    dt = np.float64 if elem == "f64" else np.float32
    text, lb, ub = case_text(name)
    ins = [helpers.hash_field(shape, dt, seed=40 + k) for k in range(nin)]
    want = np.full(shape, -7.0, dtype=dt)
    oracle.Module.parse(text).call("entry", want, *ins)
    mod = lowering.compile_module(text)
    assert {a["function"]: a["kernel"] for a in mod.report["applies"]}["resid"] == kernel
    d_ins = [torch.from_numpy(a).cuda() for a in ins]
    tdt = torch.float64 if elem == "f64" else torch.float32
    # automatic tile, then every default tile with chunk seams inside the field, then the direct kernel
    settings = [{}]
    if kernel == "march":
        nvar = {3: 8, 2: 3, 1: 1}[rank]
        settings += [{"NEPTUNE_HIP_VARIANT": str(v), "NEPTUNE_HIP_CHUNK": c} for v in range(nvar) for c in ("1", "4")]
        settings += [{"NEPTUNE_HIP_KERNEL": "direct"}]
    for s in settings:
        for k in ("NEPTUNE_HIP_KERNEL", "NEPTUNE_HIP_VARIANT", "NEPTUNE_HIP_CHUNK"):
            os.environ.pop(k, None)
        os.environ.update(s)
        d_out = torch.full(shape, -7.0, dtype=tdt, device="cuda")
        mod.call("entry", d_out, *d_ins)
        got = d_out.cpu().numpy()
        assert bits_equal(got, want), f"{name} {s}: " + mismatch_report(got, want)
    # outside the apply bounds the result is input 0, whichever ring or register it came from
    assert bits_equal(want[0], ins[0][0])
