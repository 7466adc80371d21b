"""Inputs that live in boxes of their own on the fast kernels.  The reference's apply only asks that input 0 has the
result's shape (DataflowLowering.cpp:283-287); inputs 1.. index through their OWN lower bounds (:382-410) -- a
face-located field of extent N+1 beside a cell-located result (staggered grids), a field that carries its ghost layers.
Such an input used to send the whole apply to the direct kernel; the march / plane-in-LDS / rank-2 tile kernels read
it through a per-input view (MarchParams::view: own pitch, shift, clamp range) whenever its box contains the result's.
Every case: lowered from NeptuneIR text, bit for bit against the oracle on the automatic tile, on every default tile with
chunk seams inside the field and on the direct kernel; and the launch really is the march kernel."""
import ctypes as C
import os

import numpy as np
import pytest

import helpers
from helpers import bits_equal, mismatch_report, oracle

pytestmark = pytest.mark.gpu

COEF = [0.5, -0.25, 1.5, 0.125, -0.75, 2.0, -1.25, 0.375, 0.0625, -3.0, 1.75, 0.3125, -0.4375, 2.5, -0.1875, 0.875,
        1.125, -2.25, 0.6875, -0.5625, 3.5, -1.375, 0.21875, 0.9375, -1.625, 2.75, -0.3125, 1.0625, 0.4375, -0.8125,
        1.3125, -2.0625, 0.15625, 2.125, -0.65625, 1.4375, -1.1875, 0.78125, 0.28125, -2.625]


def module_text(elem, out_box, bounds, in_boxes, accesses):
    """@resid(u0, u1, ...) = one apply over `bounds`; input k has box in_boxes[k] (input 0: the result's); the body is
    sum_i COEF[i] * access_i left to right plus the last index, so a wrong shift or pitch changes bits"""
    rank = len(out_box[0])
    nin = len(in_boxes)
    dims = "x".join("?" * rank)

    def battr(b):
        return f"#neptune_ir.bounds<lb = [{', '.join(map(str, b[0]))}], ub = [{', '.join(map(str, b[1]))}]>"
    L = ['#l = #neptune_ir.location<"cell">', f"#bo = {battr(out_box)}",
         f"!to = !neptune_ir.temp<element = {elem}, bounds = #bo, location = #l>",
         f"!fo = !neptune_ir.field<element = {elem}, bounds = #bo, location = #l>"]
    for k, b in enumerate(in_boxes):
        L += [f"#b{k} = {battr(b)}", f"!t{k} = !neptune_ir.temp<element = {elem}, bounds = #b{k}, location = #l>",
              f"!f{k} = !neptune_ir.field<element = {elem}, bounds = #b{k}, location = #l>"]
    tys = ", ".join(f"!t{k}" for k in range(nin))
    idx = ", ".join(f"%i{d}: index" for d in range(rank))
    L += ["module {", f"  neptune_ir.nonlinear_opdef @resid : ({tys}) -> !to {{",
          "  ^bb0(" + ", ".join(f"%u{k}: !t{k}" for k in range(nin)) + "):",
          "    %r = neptune_ir.apply(" + ", ".join(f"%u{k}" for k in range(nin)) + f") attributes {{bounds = {battr(bounds)}}} : ({tys}) -> !to {{",
          f"      ^bb0({idx}, " + ", ".join(f"%a{k}: !t{k}" for k in range(nin)) + "):"]
    for i, (k, off) in enumerate(accesses):
        L.append(f"        %v{i} = neptune_ir.access %a{k}[{', '.join(map(str, off))}] : !t{k} -> {elem}")
    for i in range(len(accesses)):
        L += [f"        %c{i} = arith.constant {COEF[i]!r} : {elem}", f"        %m{i} = arith.mulf %c{i}, %v{i} : {elem}",
              f"        %s{i} = arith.addf " + ("%m0, %m0" if i == 0 else f"%s{i - 1}, %m{i}") + f" : {elem}"]
    n = len(accesses) - 1
    L += [f"        %w = arith.index_cast %i{rank - 1} : index to i64", f"        %wf = arith.sitofp %w : i64 to {elem}",
          f"        %o = arith.addf %s{n}, %wf : {elem}", f"        neptune_ir.yield %o : {elem}", "    }",
          "    neptune_ir.return %r : !to", "  }",
          f"  func.func @entry(%out: memref<{dims}x{elem}>, " + ", ".join(f"%in{k}: memref<{dims}x{elem}>" for k in range(nin))
          + f") -> memref<{dims}x{elem}> {{", f"    %fout = neptune_ir.wrap %out : memref<{dims}x{elem}> -> !fo"]
    for k in range(nin):
        L += [f"    %g{k} = neptune_ir.wrap %in{k} : memref<{dims}x{elem}> -> !f{k}", f"    %x{k} = neptune_ir.load %g{k} : !f{k} -> !t{k}"]
    L += ["    %y = neptune_ir.apply_nonlinear @resid(" + ", ".join(f"%x{k}" for k in range(nin)) + f") : ({tys}) -> !to",
          "    neptune_ir.store %y to %fout : !to to !fo", f"    %res = neptune_ir.unwrap %fout : !fo -> memref<{dims}x{elem}>",
          f"    func.return %res : memref<{dims}x{elem}>", "  }", "}"]
    return "\n".join(L) + "\n"


def star(rank, r, dims=None):
    out = []
    for d in (range(rank) if dims is None else dims):
        for s in range(1, r + 1):
            for sign in (-1, 1):
                o = [0] * rank
                o[d] = sign * s
                out.append(tuple(o))
    return out


def box(lb, shape):
    return (list(lb), [a + n for a, n in zip(lb, shape)])


def grow(b, lo, hi):
    return ([a - g for a, g in zip(b[0], lo)], [a + g for a, g in zip(b[1], hi)])


def _cases():
    c = {}
    # staggered 3-D: cell result; input 0 cell field at the centre; input 1 on the k-faces (extent N2+1, read at k and k+1);
    # input 2 carries two ghost layers and is read with a radius-2 star -- over the WHOLE result box, rim cells included
    ob = box((0, 0, 0), (14, 20, 256))
    c["staggered_3d_faces_and_ghosts"] = ("f64", ob, ob, [ob, grow(ob, (0, 0, 0), (0, 0, 1)), grow(ob, (2, 2, 2), (2, 2, 2))],
                                          [(0, (0, 0, 0)), (1, (0, 0, 0)), (1, (0, 0, 1))] + [(2, (0, 0, 0))] + [(2, o) for o in star(3, 2)])
    # one input read at offsets, in a box with one ghost layer: the 7-point tiles of the march kernel
    c["ghosted_7pt_3d"] = ("f64", ob, ob, [ob, grow(ob, (1, 1, 1), (1, 1, 1))], [(0, (0, 0, 0)), (1, (0, 0, 0))] + [(1, o) for o in star(3, 1)])
    # faces along all three dimensions, shifted logical origin, bounds tighter than the box
    ob2 = box((3, -2, 5), (11, 18, 512))      # (rows end on a span boundary: the k-face field's one extra cell is a halo cell)
    c["faces_ijk_shifted_origin_f32"] = ("f32", ob2, ([4, -1, 6], [13, 15, 516]),
                                         [ob2, grow(ob2, (0, 0, 0), (1, 0, 0)), grow(ob2, (0, 0, 0), (0, 1, 0)), grow(ob2, (0, 0, 0), (0, 0, 1))],
                                         [(0, (0, 0, 0)), (1, (0, 0, 0)), (1, (1, 0, 0)), (2, (0, 0, 0)), (2, (0, 1, 0)), (3, (0, 0, 0)), (3, (0, 0, 1))])
    # 27-point box on a ghosted input (march box tile and the all-planes-in-LDS kernel)
    c["ghosted_27pt_f32"] = ("f32", ob, ob, [ob, grow(ob, (1, 1, 1), (1, 1, 1))],
                             [(0, (0, 0, 0))] + [(1, (a, b, cc)) for a in (-1, 0, 1) for b in (-1, 0, 1) for cc in (-1, 0, 1)])
    # rows that end in the middle of a wave's span (130 and 131 cells: ragged), the second input two cells wider on the
    # right and one row taller on each side
    obr = box((0, 0, 0), (9, 13, 130))
    c["odd_rows_wider_input"] = ("f64", obr, ([0, 1, 0], [9, 12, 130]), [obr, grow(obr, (0, 1, 0), (0, 1, 2))],
                                 [(0, (0, 0, 0)), (1, (0, 0, 0)), (1, (0, 0, 1)), (1, (0, -1, 0)), (1, (0, 1, 0))])
    obr2 = box((0, 0, 0), (7, 9, 131))
    c["ragged_rows_ghosted"] = ("f64", obr2, obr2, [obr2, grow(obr2, (1, 1, 1), (1, 1, 3))], [(0, (0, 0, 0)), (1, (0, 0, 0))] + [(1, o) for o in star(3, 1)])
    # radius-4 star on a ghosted input: the plane-in-LDS kernel
    c["ghosted_radius4_star"] = ("f64", ob, ob, [ob, grow(ob, (4, 4, 4), (4, 4, 4))], [(0, (0, 0, 0)), (1, (0, 0, 0))] + [(1, o) for o in star(3, 4)])
    # rank 2: faces along d0, ghosts on the third input (tile form), and a wide radius-5 ghosted star (LDS tile kernel)
    o2 = box((0, 0), (40, 512))
    c["staggered_2d"] = ("f64", o2, o2, [o2, grow(o2, (0, 0), (1, 0)), grow(o2, (1, 1), (1, 1))],
                         [(0, (0, 0)), (1, (0, 0)), (1, (1, 0)), (2, (0, 0))] + [(2, o) for o in star(2, 1)])
    c["ghosted_radius5_2d"] = ("f64", o2, o2, [o2, grow(o2, (5, 5), (5, 5))], [(0, (0, 0)), (1, (0, 0))] + [(1, o) for o in star(2, 5)])
    # fewer than a lane vector of cells right of rows that end INSIDE a wave's span: the lane past the result's last vector
    # loads the row's last whole vector and rotates it into place (InView::fix_k / fix_d) -- f64 by one element, f32 by
    # three, two and one; on the march tiles, the plane-in-LDS kernels and the rank-2 tile kernels
    obm = box((0, 0, 0), (6, 8, 130))
    c["kfaces_mid_span_f64"] = ("f64", obm, obm, [obm, grow(obm, (0, 0, 0), (0, 0, 1))], [(0, (0, 0, 0)), (1, (0, 0, 0)), (1, (0, 0, 1))])
    obn = box((0, 0, 0), (6, 9, 520))
    for extra in (1, 2, 3):
        c[f"kfaces_mid_span_f32_plus{extra}"] = ("f32", obn, obn, [obn, grow(obn, (0, 0, 0), (0, 0, extra))],
                                                 [(0, (0, 0, 0))] + [(1, (0, 0, e)) for e in range(extra + 1)])
    c["ghosted_7pt_mid_span_f64"] = ("f64", obm, obm, [obm, grow(obm, (1, 1, 1), (1, 1, 1))], [(0, (0, 0, 0)), (1, (0, 0, 0))] + [(1, o) for o in star(3, 1)])
    c["ghosted_27pt_mid_span_f32"] = ("f32", obn, obn, [obn, grow(obn, (1, 1, 1), (1, 1, 1))],
                                      [(0, (0, 0, 0))] + [(1, (a, b, cc)) for a in (-1, 0, 1) for b in (-1, 0, 1) for cc in (-1, 0, 1)])
    c["ghosted_radius3_star_mid_span_f32"] = ("f32", obn, obn, [obn, grow(obn, (3, 3, 3), (3, 3, 3))], [(0, (0, 0, 0)), (1, (0, 0, 0))] + [(1, o) for o in star(3, 3)])
    c["ghosted_radius5_star_mid_span_f64"] = ("f64", obm, obm, [obm, grow(obm, (5, 5, 5), (5, 5, 5))], [(0, (0, 0, 0)), (1, (0, 0, 0))] + [(1, o) for o in star(3, 5)])
    o2m = box((0, 0), (40, 130))
    c["staggered_2d_mid_span"] = ("f64", o2m, o2m, [o2m, grow(o2m, (0, 0), (0, 1)), grow(o2m, (1, 1), (1, 1))],
                                  [(0, (0, 0)), (1, (0, 0)), (1, (0, 1)), (2, (0, 0))] + [(2, o) for o in star(2, 1)])
    o2n = box((0, 0), (40, 520))
    c["ghosted_radius5_2d_mid_span_f32"] = ("f32", o2n, o2n, [o2n, grow(o2n, (5, 5), (5, 5))], [(0, (0, 0)), (1, (0, 0))] + [(1, o) for o in star(2, 5)])
    # logical origins near the front end's limit of 2^40: the index arithmetic is 64-bit end to end (the body adds the last index)
    oh = box((1 << 39, -(1 << 39) + 7), (24, 512))
    c["huge_origin_2d"] = ("f64", oh, oh, [oh, grow(oh, (1, 1), (1, 1))], [(0, (0, 0)), (1, (0, 0))] + [(1, o) for o in star(2, 1)])
    oh3 = box((-(1 << 38), 3, (1 << 39) - 100), (6, 9, 260))
    c["huge_origin_3d_f32"] = ("f32", oh3, oh3, [oh3, grow(oh3, (0, 0, 0), (0, 0, 1))], [(0, (0, 0, 0)), (1, (0, 0, 0)), (1, (0, 0, 1))])
    # rank 1
    o1 = box((0,), (4096,))
    c["staggered_1d"] = ("f64", o1, o1, [o1, grow(o1, (0,), (1,)), grow(o1, (2,), (2,))],
                         [(0, (0,)), (1, (0,)), (1, (1,)), (2, (-2,)), (2, (2,)), (2, (0,))])
    o1m = box((0,), (1000,))
    c["staggered_1d_mid_span_f32"] = ("f32", o1m, o1m, [o1m, grow(o1m, (0,), (1,)), grow(o1m, (2,), (2,))],
                                      [(0, (0,)), (1, (0,)), (1, (1,)), (2, (-2,)), (2, (2,)), (2, (0,))])
    return c


CASES = _cases()


def case_text(name):
    elem, ob, bounds, in_boxes, accesses = CASES[name]
    return module_text(elem, ob, bounds, in_boxes, accesses)


@pytest.fixture(scope="module")
def env(built_libs, tmp_path_factory):
    import torch
    assert torch.cuda.is_available()
    os.environ["NEPTUNE_CACHE_DIR"] = str(tmp_path_factory.mktemp("neptune_cache_ownbox"))
    from neptune_hip import _capi, lowering
    helpers.prefetch_modules([case_text(name) for name in CASES])
    return lowering, torch, _capi


@pytest.mark.parametrize("name", list(CASES))
def test_inputs_in_their_own_boxes_run_the_fast_kernels(env, name):
    lowering, torch, capi = env
    elem, ob, bounds, in_boxes, accesses = CASES[name]
    rank = len(ob[0])
    dt = np.float64 if elem == "f64" else np.float32
    tdt = torch.float64 if elem == "f64" else torch.float32
    text = case_text(name)
    shape = tuple(u - l for l, u in zip(*ob))
    ins = [helpers.hash_field(tuple(u - l for l, u in zip(*b)), dt, seed=70 + k) for k, b in enumerate(in_boxes)]
    want = np.full(shape, -7.0, dtype=dt)
    oracle.Module.parse(text).call("entry", want, *ins)
    mod = lowering.compile_module(text)
    assert {a["function"]: a["kernel"] for a in mod.report["applies"]}["resid"] == "march"
    d_ins = [torch.from_numpy(a).cuda() for a in ins]
    lib = capi.load()
    saved = {k: os.environ.get(k) for k in ("NEPTUNE_HIP_KERNEL", "NEPTUNE_HIP_VARIANT", "NEPTUNE_HIP_CHUNK")}
    nvar = {3: 8, 2: 3, 1: 1}[rank]
    settings = [{}] + [{"NEPTUNE_HIP_VARIANT": str(v), "NEPTUNE_HIP_CHUNK": c} for v in range(nvar) for c in ("1", "4")] + [{"NEPTUNE_HIP_KERNEL": "direct"}]
    try:
        for s in settings:
            for k in saved:
                os.environ.pop(k, None)
            os.environ.update(s)
            d_out = torch.full(shape, -7.0, dtype=tdt, device="cuda")
            mod.call("entry", d_out, *d_ins)
            got = d_out.cpu().numpy()
            assert bits_equal(got, want), f"{name} {s}: " + mismatch_report(got, want)
            last = capi.LaunchCfg()
            assert lib.neptune_hip_last_launch(C.byref(last)) == 1
            if "NEPTUNE_HIP_KERNEL" not in s and shape[-1] % (16 // np.dtype(dt).itemsize) == 0:
                # (ragged rows end with a direct launch for the row tails; everything else must END on the march kernel)
                assert last.kernel == capi.KERNEL_MARCH, f"{name} {s}: ran kernel {last.kernel}, not the march kernel"
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    # cells outside apply.bounds carry input 0
    if bounds != ob:
        lo = [b - o for b, o in zip(bounds[0], ob[0])]
        assert bits_equal(want[tuple(slice(0, max(l, 1)) for l in lo)], ins[0][tuple(slice(0, max(l, 1)) for l in lo)]) or any(l == 0 for l in lo)
