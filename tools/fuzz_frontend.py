#!/usr/bin/env python3
"""Mutation fuzzing of the NeptuneIR front end (parser, verifier, emitter) on the CPU: every fixture, the reference's
own .mlir inputs when mounted, and random generator modules are truncated / spliced / bit-flipped / number-swapped and
fed to neptune-opt.  A run may accept or reject an input (exit 0 / 1 / 2) but must terminate quickly and cleanly;
with an AddressSanitizer/UBSan build of neptune-opt (see the recipe printed by --help) it also must not trip a sanitizer.
usage: tools/fuzz_frontend.py [--opt PATH] [--mutants-per-input N] [--seed S] [--asan-build DIR]"""
import argparse
import glob
import random
import re
import subprocess
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "tests"))
sys.path.insert(0, str(REPO / "neptune-pde-solver_amd"))
SOURCES = ["capi.cpp", "emit_hip.cpp", "neptune_opt_main.cpp", "parser.cpp", "verify.cpp"]


def asan_build(directory: Path) -> Path:
    exe = directory / "neptune-opt-asan"
    src = [str(REPO / "neptune-pde-solver_amd/csrc/lowering" / s) for s in SOURCES]
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", *src, "-o", str(exe)],
                   check=True)
    return exe


def corpus(generator_seeds=range(7000, 7030)):
    import test_fuzz_gpu as fz
    files = glob.glob(str(REPO / "tests/**/*.mlir"), recursive=True) + glob.glob("/root/reference/test/**/*.mlir", recursive=True)
    import test_batched_gpu as tb
    import test_ownbox_gpu as ob
    extra = [ob.case_text(n) for n in ob.CASES] + [tb.rank5_text(), tb.stencil4d_text(), tb.stencil5d_text()] + list(tb.step4_texts().values())
    return [Path(f).read_text() for f in sorted(files)] + [fz.gen_module(s)[0] for s in generator_seeds] + extra


def mutate(rng, s):
    k = rng.randrange(5)
    if k == 0:
        return s[:rng.randrange(1, len(s))]
    if k == 1:
        a = rng.randrange(len(s))
        return s[:a] + s[min(len(s), a + rng.randrange(1, 60)):]
    if k == 2:
        a = rng.randrange(len(s))
        return s[:a] + rng.choice('{}[]()<>%#@!,:=-0 9x"') + s[a + 1:]
    if k == 3:
        a = rng.randrange(len(s))
        b = min(len(s), a + rng.randrange(1, 40))
        return s[:a] + s[a:b] * 3 + s[b:]
    nums = list(re.finditer(r"-?\d+", s))
    if not nums:
        return s
    m = rng.choice(nums)
    return s[:m.start()] + str(rng.choice([0, -1, 99999999999, -7, 3, 2 ** 63])) + s[m.end():]


def run(opt: Path, mutants_per_input: int, seed: int, timeout: float = 20.0):
    rng = random.Random(seed)
    texts = corpus()
    inputs = texts + [mutate(rng, t) for t in texts for _ in range(mutants_per_input)]
    codes, findings = {}, []
    with tempfile.TemporaryDirectory() as d:
        src, out = Path(d) / "in.mlir", Path(d) / "out.hip"
        for n, text in enumerate(inputs):
            src.write_text(text)
            for args in (["--neptuneir-to-hip", "-o", str(out), "--report"], ["--verify-only"]):
                try:
                    r = subprocess.run([str(opt), str(src)] + args, capture_output=True, text=True, timeout=timeout)
                    code, err = r.returncode, r.stderr
                except subprocess.TimeoutExpired:
                    code, err = "timeout", ""
                codes[code] = codes.get(code, 0) + 1
                if code not in (0, 1, 2) or "Sanitizer" in err or "runtime error" in err:
                    findings.append((n, args[0], code, err[-800:], text))
    return len(texts), len(inputs) - len(texts), codes, findings


def main():
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--opt", default=str(REPO / "neptune-pde-solver_amd/bin/neptune-opt"))
    ap.add_argument("--asan-build", default=None, help="directory to build a sanitizer-instrumented neptune-opt in, then use it")
    ap.add_argument("--mutants-per-input", type=int, default=8)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    opt = asan_build(Path(args.asan_build)) if args.asan_build else Path(args.opt)
    originals, mutants, codes, findings = run(opt, args.mutants_per_input, args.seed)
    for n, mode, code, err, text in findings[:5]:
        print(f"FINDING input #{n} {mode} -> {code}\n{err}\n--- input ---\n{text[:1500]}\n")
    print(f"FUZZ_FRONTEND originals={originals} mutants={mutants} exit_codes={codes} findings={len(findings)}")
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
