"""Seeded random applies through the whole route (NeptuneIR text -> lowering -> hipcc -> module ABI -> kernels),
bit for bit against the oracle: random rank / element type / number of inputs, star and box footprints of radius
1-2 (1-5 in 1-D / 2-D, 1-4 in 3-D) spread over several inputs, bodies mixing arith/math ops, selects, scf.if on region index arguments with a
conditional access, ragged and aligned rows, apply bounds tighter than the halo margin.  Every case runs on the
automatic kernel choice, on both forms of the direct kernel and -- when the body can march -- on each default
tile with chunk seams."""
import os

import numpy as np
import pytest

import helpers
from helpers import bits_equal, mismatch_report, oracle

pytestmark = pytest.mark.gpu

CONSTS = [0.5, -0.25, 1.5, 0.125, -0.75, 2.0, -1.25, 0.375, 3.0, -0.0625, 1.75, 0.3125]


def gen_apply(rng, name, rank, elem, shape, origin):
    nin = int(rng.integers(1, 4))
    radius = int(rng.choice([1, 1, 2]))
    if rank < 3 and min(shape) >= 11 and rng.random() < 0.3:       # high-order stars (march kernel up to radius 4)
        radius = int(rng.choice([3, 4, 5]))
    elif rank == 3 and min(shape) >= 9 and rng.random() < 0.25:    # 3-D: up to radius 3
        radius = int(rng.choice([3, 3, 4]))
    box = (radius == 1 and rank > 1 and rng.random() < 0.35) or (radius == 2 and rank == 2 and rng.random() < 0.4)
    accesses = []                       # (input, offsets)
    for k in range(nin):
        style = rng.choice(["centre", "halo"]) if k > 0 else "halo"
        accesses.append((k, (0,) * rank))
        if style == "centre":
            continue
        n_off = int(rng.integers(2, 7))
        for _ in range(n_off):
            if box:
                off = tuple(int(rng.integers(-radius, radius + 1)) for _ in range(rank))
            else:
                d = int(rng.integers(0, rank))
                off = tuple(int(rng.choice([-radius, -1, 1, radius])) if a == d else 0 for a in range(rank))
            if any(off) and (k, off) not in accesses:
                accesses.append((k, off))
    L = []
    vals = []
    for n, (k, off) in enumerate(accesses):
        L.append(f"%a{n} = neptune_ir.access %in{k}[{', '.join(map(str, off))}] : !t -> {elem}")
        vals.append(f"%a{n}")
    cnt = 0

    def const():
        nonlocal cnt
        cnt += 1
        L.append(f"%c{cnt} = arith.constant {CONSTS[int(rng.integers(0, len(CONSTS)))]!r} : {elem}")
        return f"%c{cnt}"

    def binop(a, b):
        nonlocal cnt
        cnt += 1
        op = rng.choice(["arith.addf", "arith.subf", "arith.mulf", "arith.addf", "arith.subf", "arith.maximumf", "arith.minimumf"])
        L.append(f"%v{cnt} = {op} {a}, {b} : {elem}")
        return f"%v{cnt}"

    acc = vals[0]
    for v in vals[1:]:
        term = v
        if rng.random() < 0.6:
            c = const()
            cnt += 1
            term = f"%v{cnt}"
            L.append(f"{term} = arith.mulf {c}, {v} : {elem}")
        acc = binop(acc, term)
    if rng.random() < 0.5:
        cnt += 1
        a = f"%v{cnt}"
        L.append(f"{a} = math.absf {acc} : {elem}")
        acc = binop(a, vals[0])
    if rng.random() < 0.5:
        c = const()
        cnt += 1
        L.append(f"%v{cnt} = arith.divf {acc}, {c} : {elem}")
        acc = f"%v{cnt}"
    if rng.random() < 0.25:                     # floor / ceil / copysign: exact
        cnt += 1
        f = rng.choice(["math.floor", "math.ceil"])
        L.append(f"%h{cnt} = {f} {acc} : {elem}")
        L.append(f"%v{cnt} = math.copysign %h{cnt}, {vals[-1]} : {elem}")
        acc = binop(f"%v{cnt}", vals[0])
    if rng.random() < 0.3:                      # sqrt(|x|): IEEE-exact on both sides, unlike exp/log/...
        cnt += 1
        L.append(f"%g{cnt} = math.absf {acc} : {elem}")
        L.append(f"%v{cnt} = math.sqrt %g{cnt} : {elem}")
        acc = f"%v{cnt}"
    if rng.random() < 0.5:                      # select on a float compare
        z = const()
        cnt += 1
        L.append(f"%p{cnt} = arith.cmpf {rng.choice(['olt', 'oge', 'une', 'ogt'])}, {vals[0]}, {z} : {elem}")
        L.append(f"%v{cnt} = arith.select %p{cnt}, {acc}, {vals[-1]} : {elem}")
        acc = f"%v{cnt}"
    margin = radius
    if rng.random() < 0.6:                      # scf.if on an index argument, with a conditional access
        d = int(rng.integers(0, rank))
        thr = origin[d] + int(shape[d] // 2)
        cnt += 1
        off = tuple(margin if a == d else 0 for a in range(rank))      # in range: taken only for i_d < thr
        L.append(f"%thr{cnt} = arith.constant {thr} : index")
        L.append(f"%q{cnt} = arith.cmpi slt, %i{d}, %thr{cnt} : index")
        L.append(f"%v{cnt} = scf.if %q{cnt} -> ({elem}) {{")
        L.append(f"  %ca{cnt} = neptune_ir.access %in0[{', '.join(map(str, off))}] : !t -> {elem}")
        L.append(f"  %cb{cnt} = arith.addf {acc}, %ca{cnt} : {elem}")
        L.append(f"  scf.yield %cb{cnt} : {elem}")
        L.append("} else {")
        L.append(f"  %w{cnt} = arith.index_cast %i{d} : index to i64")
        L.append(f"  %wf{cnt} = arith.sitofp %w{cnt} : i64 to {elem}")
        L.append(f"  %cc{cnt} = arith.subf {acc}, %wf{cnt} : {elem}")
        L.append(f"  scf.yield %cc{cnt} : {elem}")
        L.append("}")
        acc = f"%v{cnt}"
    L.append(f"neptune_ir.yield {acc} : {elem}")
    lb = [o + margin + int(rng.integers(0, 2)) for o in origin]
    ub = [o + n - margin - int(rng.integers(0, 2)) for o, n in zip(origin, shape)]
    tys = ", ".join(["!t"] * nin)
    idx = ", ".join(f"%i{d}: index" for d in range(rank))
    ins = ", ".join(f"%in{k}: !t" for k in range(nin))
    text = [f"  neptune_ir.nonlinear_opdef @{name} : ({tys}) -> !t {{",
            "  ^bb0(" + ", ".join(f"%u{k}: !t" for k in range(nin)) + "):",
            "    %r = neptune_ir.apply(" + ", ".join(f"%u{k}" for k in range(nin)) + ") attributes {bounds = "
            f"#neptune_ir.bounds<lb = [{', '.join(map(str, lb))}], ub = [{', '.join(map(str, ub))}]>}} : ({tys}) -> !t {{",
            f"      ^bb0({idx}, {ins}):"] + ["        " + l for l in L] + ["    }", "    neptune_ir.return %r : !t", "  }"]
    return "\n".join(text), nin


def gen_module(seed):
    """one module = one field box (random logical origin) and element type, three random opdefs and an @entry that
    composes them: op0 on the loaded inputs, optionally a second operator on its result, a full or sub-box store"""
    rng = np.random.default_rng(seed)
    rank = int(rng.choice([1, 2, 3, 3]))
    elem = str(rng.choice(["f64", "f64", "f32"]))
    vk = 2 if elem == "f64" else 4
    last = int(rng.choice([128, 192, 256, 320])) * (vk // 2) + (int(rng.integers(1, vk)) if rng.random() < 0.5 else 0)
    shape = [int(rng.integers(6, 14)) for _ in range(rank - 1)] + [last]
    origin = [int(rng.integers(-3, 5)) for _ in range(rank)]
    lbs = ", ".join(map(str, origin))
    ubs = ", ".join(str(o + n) for o, n in zip(origin, shape))
    head = ['#l = #neptune_ir.location<"cell">', f"#b = #neptune_ir.bounds<lb = [{lbs}], ub = [{ubs}]>",
            f"!t = !neptune_ir.temp<element = {elem}, bounds = #b, location = #l>",
            f"!f = !neptune_ir.field<element = {elem}, bounds = #b, location = #l>", "module {"]
    ops = []
    for n in range(3):
        text, nin = gen_apply(rng, f"op{n}", rank, elem, shape, origin)
        head.append(text)
        ops.append((f"op{n}", nin))
    # @entry(out, in0, in1, in2): y = op0(ins...); optionally y = op_k(y, ins[1:]...); store y to out {bounds?}
    mr = "x".join("?" * rank) + "x" + elem
    nmax = max(n for _, n in ops)
    E = [f"  func.func @entry(%out: memref<{mr}>, " + ", ".join(f"%m{k}: memref<{mr}>" for k in range(nmax)) + f") -> memref<{mr}> {{",
         f"    %fo = neptune_ir.wrap %out : memref<{mr}> -> !f"]
    for k in range(nmax):
        E.append(f"    %f{k} = neptune_ir.wrap %m{k} : memref<{mr}> -> !f")
        E.append(f"    %t{k} = neptune_ir.load %f{k} : !f -> !t")
    n0 = ops[0][1]
    E.append("    %y0 = neptune_ir.apply_nonlinear @op0(" + ", ".join(f"%t{k}" for k in range(n0)) + ") : (" + ", ".join(["!t"] * n0) + ") -> !t")
    cur = "%y0"
    if rng.random() < 0.6:
        which = int(rng.integers(1, 3))
        nk = ops[which][1]
        args = [cur] + [f"%t{k}" for k in range(1, nk)]
        E.append(f"    %y1 = neptune_ir.apply_nonlinear @op{which}(" + ", ".join(args) + ") : (" + ", ".join(["!t"] * nk) + ") -> !t")
        cur = "%y1"
    if rng.random() < 0.4:
        slb = [o + int(rng.integers(0, 3)) for o in origin]
        sub = [o + n - int(rng.integers(0, 3)) for o, n in zip(origin, shape)]
        E.append(f"    neptune_ir.store {cur} to %fo {{bounds = #neptune_ir.bounds<lb = [{', '.join(map(str, slb))}], ub = [{', '.join(map(str, sub))}]>}} : !t to !f")
    else:
        E.append(f"    neptune_ir.store {cur} to %fo : !t to !f")
    E += [f"    %res = neptune_ir.unwrap %fo : !f -> memref<{mr}>", f"    func.return %res : memref<{mr}>", "  }"]
    head.append("\n".join(E))
    head.append("}")
    return "\n".join(head) + "\n", tuple(shape), elem, ops


@pytest.fixture(scope="module")
def env(built_libs, tmp_path_factory):
    import torch
    assert torch.cuda.is_available()
    os.environ["NEPTUNE_CACHE_DIR"] = str(tmp_path_factory.mktemp("neptune_cache_fuzz"))
    from neptune_hip import lowering
    helpers.prefetch_modules([gen_module(seed)[0] for seed in SEEDS])   # every seed's module, compiled side by side
    return lowering, torch


SEEDS = [101, 202, 303, 404, 505, 606, 703, 709, 722, 725, 730, 739, 757]


@pytest.mark.parametrize("seed", SEEDS)
def test_random_applies_match_the_oracle(env, monkeypatch, seed):
    lowering, torch = env
    text, shape, elem, ops = gen_module(seed)
    dt = np.float64 if elem == "f64" else np.float32
    m = oracle.Module.parse(text)
    mod = lowering.compile_module(text)
    kern = {a["function"]: a["kernel"] for a in mod.report["applies"]}
    rank = len(shape)
    for name, nin in ops:
        ins = [helpers.hash_field(shape, dt, seed=seed + 7 * k) for k in range(nin)]
        want = m.call(name, *ins)
        d_ins = [torch.from_numpy(a).cuda() for a in ins]
        settings = [{}, {"NEPTUNE_HIP_KERNEL": "direct"}, {"NEPTUNE_HIP_KERNEL": "direct-flat"}]
        if kern[name] == "march":
            nvar = {3: 8, 2: 3, 1: 1}[rank]
            settings += [{"NEPTUNE_HIP_KERNEL": "march", "NEPTUNE_HIP_VARIANT": str(v), "NEPTUNE_HIP_CHUNK": "3"} for v in range(nvar)]
        for s in settings:
            for k in ("NEPTUNE_HIP_KERNEL", "NEPTUNE_HIP_VARIANT", "NEPTUNE_HIP_CHUNK"):
                monkeypatch.delenv(k, raising=False)
            for k, v in s.items():
                monkeypatch.setenv(k, v)
            got = mod.call(name, *d_ins).cpu().numpy()
            assert bits_equal(got, want), f"seed={seed} {name} shape={shape} {elem} kernel={kern[name]} {s}\n" + \
                mismatch_report(got, want) + "\n" + text
    # the composed @entry: fresh destination, then in place (destination = input 0), device and host buffers
    nmax = max(n for _, n in ops)
    ins = [helpers.hash_field(shape, dt, seed=seed + 11 * k) for k in range(nmax)]
    for inplace in (False, True):
        h_ins = [a.copy() for a in ins]
        h_out = h_ins[0] if inplace else np.full(shape, 9.0, dtype=dt)
        m.call("entry", h_out, *h_ins)
        for device in (True, False):
            for k in ("NEPTUNE_HIP_KERNEL", "NEPTUNE_HIP_VARIANT", "NEPTUNE_HIP_CHUNK"):
                monkeypatch.delenv(k, raising=False)
            g_ins = [torch.from_numpy(a.copy()).cuda() if device else a.copy() for a in ins]
            g_out = g_ins[0] if inplace else (torch.full(shape, 9.0, dtype=g_ins[0].dtype, device="cuda") if device
                                              else np.full(shape, 9.0, dtype=dt))
            mod.call("entry", g_out, *g_ins)
            got = g_out.cpu().numpy() if device else g_out
            assert bits_equal(got, h_out), f"seed={seed} entry inplace={inplace} device={device} shape={shape} {elem}\n" + \
                mismatch_report(got, h_out) + "\n" + text
