"""N > 1 path on CPU: world_size-2 (and 3) gloo runs of the slab decomposition + halo exchange,
launched the way the driver launches bench.py (python -m torch.distributed.run, 127.0.0.1)."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

from neptune_hip import slab

HERE = Path(__file__).resolve().parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world, kind, shape, steps=3):
    env = dict(os.environ, SLAB_KIND=kind, SLAB_SHAPE=",".join(map(str, shape)), SLAB_STEPS=str(steps),
               OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(HERE / "slab_gloo_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert f"SLAB_OK world={world}" in p.stdout


def test_two_ranks_3d_7pt():
    _launch(2, "3d7", (13, 6, 8))


def test_three_ranks_3d_27pt_f32_uneven_split():
    _launch(3, "3d27", (11, 5, 8), steps=2)


def test_two_ranks_2d_5pt():
    _launch(2, "2d5", (10, 12))


def test_decomposition_geometry():
    box = ([0, 0, 0], [1024, 1024, 1024])
    slabs = [slab.decompose(box, 1, g, 8) for g in range(8)]
    assert [s.n_own for s in slabs] == [128] * 8
    assert slabs[0].local_shape == (129, 1024, 1024) and slabs[3].local_shape == (130, 1024, 1024)
    assert slabs[0].local_lb[0] == 0 and slabs[7].local_ub[0] == 1024
    interior, edges = slabs[3].regions()
    assert interior == ([2, 0, 0], [128, 1024, 1024])
    assert edges == [([1, 0, 0], [2, 1024, 1024]), ([128, 0, 0], [129, 1024, 1024])]
    interior0, edges0 = slabs[0].regions()          # global boundary side needs no ghosts
    assert interior0 == ([0, 0, 0], [127, 1024, 1024]) and edges0 == [([127, 0, 0], [128, 1024, 1024])]
    assert slabs[0].clip_bounds(([1, 1, 1], [1023, 1023, 1023])) == ([1, 1, 1], [128, 1023, 1023])
    assert slabs[7].clip_bounds(([1, 1, 1], [1023, 1023, 1023])) == ([896, 1, 1], [1023, 1023, 1023])
    # uneven split, shifted global origin
    s = [slab.decompose(([5, 0], [15, 4]), 1, g, 3) for g in range(3)]
    assert [(x.start, x.stop) for x in s] == [(5, 9), (9, 12), (12, 15)]
    one = slab.decompose(box, 1, 0, 1)
    assert one.local_shape == (1024, 1024, 1024) and one.regions() == (([0, 0, 0], [1024, 1024, 1024]), [])
    with pytest.raises(ValueError):
        slab.decompose(([0], [3]), 1, 0, 4)
