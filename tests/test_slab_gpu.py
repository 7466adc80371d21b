"""Multi-rank rehearsal on one GPU: the real ShardedApply (regions, ghost planes, stream/event
ordering, HIP kernels) with 2-3 ranks sharing cuda:0 and gloo as transport.  The 8-GPU RCCL run is
the driver's; this pins everything except the transport call."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
HERE = Path(__file__).resolve().parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,kind,shape,steps", [(2, "3d7", (24, 12, 256), 4), (3, "3d27", (21, 9, 256), 3),
                                                    (2, "2d5", (40, 512), 5)])
def test_sharded_apply_on_one_gpu(built_libs, world, kind, shape, steps):
    env = dict(os.environ, SLAB_KIND=kind, SLAB_SHAPE=",".join(map(str, shape)), SLAB_STEPS=str(steps),
               OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(HERE / "slab_gpu_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert f"SLAB_GPU_OK world={world}" in p.stdout
