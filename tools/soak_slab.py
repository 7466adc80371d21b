#!/usr/bin/env python3
"""One-off soak of the slab route on ONE GPU: random (kind, shape, ranks, steps) through tests/slab_gpu_worker.py -- the real
ShardedApply (regions, ghost planes, stream/event ordering, HIP kernels) with gloo as transport -- each run checked by the
worker against the single-process result and the oracle.  With `peer` as third argument: tests/slab_peer_worker.py instead, the C-ABI
sharded step on the peer-copy transport BETWEEN PROCESSES (IPC mappings, handshake kernels, pushes; 2-5 ranks on the one
GPU, up to 40 steps, with and without overlap).      usage: tools/soak_slab.py [RUNS] [SEED] [gloo|peer]"""
import os
import random
import socket
import subprocess
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    peer = len(sys.argv) > 3 and sys.argv[3] == "peer"
    t0 = time.time()
    for r in range(runs):
        kind = rng.choice(["3d7", "3d27", "2d5"])
        world = rng.choice([2, 3, 4, 5] if peer else [2, 3, 4])
        n0 = rng.randrange(world * 2, 40)
        last = rng.choice([128, 129, 256, 257, 384, 130, 512])
        shape = (n0, rng.randrange(4, 14), last) if kind != "2d5" else (n0, last)
        steps = rng.randrange(1, 41) if peer else rng.randrange(1, 6)
        overlap = rng.random() < 0.7
        env = dict(os.environ, SLAB_KIND=kind, SLAB_SHAPE=",".join(map(str, shape)), SLAB_STEPS=str(steps), OMP_NUM_THREADS="1",
                   HSA_ENABLE_IPC_MODE_LEGACY="0", SLAB_OVERLAP="1" if overlap else "0", NEPTUNE_HIP_PEER_TIMEOUT_S="15")
        worker = "slab_peer_worker.py" if peer else "slab_gpu_worker.py"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), str(REPO / "tests" / worker)]
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        ok = p.returncode == 0 and f"{'SLAB_PEER_OK' if peer else 'SLAB_GPU_OK'} world={world}" in p.stdout
        print(f"run {r}: kind={kind} shape={shape} world={world} steps={steps} overlap={overlap} {'ok' if ok else 'FAILED'} ({time.time() - t0:.0f} s)", flush=True)
        if not ok:
            print(p.stdout[-1500:], p.stderr[-3000:])
            return 1
    print(f"SOAK_SLAB_OK runs={runs} seconds={time.time() - t0:.0f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
