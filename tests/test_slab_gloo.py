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


def test_slab_geometry_properties():
    """for random (planes, ranks, radius, origin): the slabs tile the global planes exactly once, ghost planes exist only
    towards existing neighbours, interior + edges tile each rank's owned planes exactly once, edge planes are precisely
    those within `radius` of a neighbour, and clipped bounds tile the apply bounds"""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=300, deadline=None)
    @given(st.integers(1, 8), st.integers(1, 4), st.integers(-7, 7), st.data())
    def check(world, radius, origin, data):
        n0 = data.draw(st.integers(max(world * radius, world), 200))
        box = ([origin, 0], [origin + n0, 6])
        slabs = [slab.decompose(box, radius, g, world) for g in range(world)]
        # owned planes: a partition of [origin, origin + n0), balanced to within one plane
        assert slabs[0].start == origin and slabs[-1].stop == origin + n0
        assert all(a.stop == b.start for a, b in zip(slabs, slabs[1:]))
        sizes = [s.n_own for s in slabs]
        assert max(sizes) - min(sizes) <= 1 and min(sizes) >= radius
        lb0 = data.draw(st.integers(origin, origin + n0))
        ub0 = data.draw(st.integers(lb0, origin + n0))
        covered = []
        for s in slabs:
            assert s.r_lo == (radius if s.rank > 0 else 0) and s.r_hi == (radius if s.rank < world - 1 else 0)
            assert s.local_lb[0] == s.start - s.r_lo and s.local_ub[0] == s.stop + s.r_hi
            lo, hi = s.owned_planes()
            interior, edges = s.regions()
            planes = []
            for reg in ([interior] if interior is not None else []) + edges:
                assert reg[0][1:] == [0] and reg[1][1:] == [6]
                planes += list(range(reg[0][0], reg[1][0]))
            assert sorted(planes) == list(range(lo, hi))            # exactly once, nothing outside the owned planes
            if interior is not None:                                # interior planes never touch a ghost plane
                assert interior[0][0] >= lo + (radius if s.r_lo else 0)
                assert interior[1][0] <= hi - (radius if s.r_hi else 0)
            clb, cub = s.clip_bounds(([lb0, 1], [ub0, 5]))
            assert clb[1:] == [1] and cub[1:] == [5] and s.start <= clb[0] <= cub[0] <= s.stop or clb[0] == cub[0]
            covered += list(range(clb[0], cub[0]))
        assert sorted(covered) == list(range(lb0, ub0))             # the clipped bounds tile apply.bounds

    check()
