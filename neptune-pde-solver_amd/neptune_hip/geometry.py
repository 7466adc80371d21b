"""Geometry of one `neptune_ir.apply` (neptune_hip_apply_geom_t) built from logical boxes.

Mirrors the attributes the reference attaches to the op and its types:
  bounds            #neptune_ir.bounds<lb, ub>  on the apply (NeptuneIROps.td:164-197)
  result / input box  bounds of the !neptune_ir.temp types (NeptuneIRTypes.td:47-58)
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

from . import _capi

Box = Tuple[Sequence[int], Sequence[int]]


def make_geom(out_box: Box, bounds: Box, in_boxes: Optional[Sequence[Box]] = None,
              region: Optional[Box] = None) -> _capi.ApplyGeom:
    """out_box/bounds/in_boxes are logical half-open boxes (lb, ub); region is result-physical.
    in_boxes defaults to one input sharing the result's box."""
    olb, oub = [list(map(int, x)) for x in out_box]
    rank = len(olb)
    if not (1 <= rank <= _capi.MAX_RANK) or len(oub) != rank:
        raise ValueError("rank must be 1..3")
    if in_boxes is None:
        in_boxes = [out_box]
    if not (1 <= len(in_boxes) <= _capi.MAX_INPUTS):
        raise ValueError("1..4 inputs")
    g = _capi.ApplyGeom()
    g.rank = rank
    g.num_inputs = len(in_boxes)
    lb, ub = bounds
    if len(lb) != rank or len(ub) != rank:
        raise ValueError("bounds rank mismatch")
    for d in range(rank):
        g.out_lb[d], g.out_ub[d] = olb[d], oub[d]
        g.lb[d], g.ub[d] = int(lb[d]), int(ub[d])
        g.region_lb[d] = 0 if region is None else int(region[0][d])
        g.region_ub[d] = (oub[d] - olb[d]) if region is None else int(region[1][d])
    for k, (ilb, iub) in enumerate(in_boxes):
        if len(ilb) != rank or len(iub) != rank:
            raise ValueError("input box rank mismatch")
        for d in range(rank):
            g.in_lb[k][d], g.in_ub[k][d] = int(ilb[d]), int(iub[d])
    return g


def interior_geom(shape: Sequence[int], halo: int = 1) -> _capi.ApplyGeom:
    """field box [0,shape), apply.bounds = the interior `halo` cells in from every face -- the
    shape of every committed fixture (e.g. apply-3d-7pt.mlir: [0,512)^3, bounds [1,511)^3)."""
    n = [int(x) for x in shape]
    return make_geom(([0] * len(n), n), ([halo] * len(n), [x - halo for x in n]))
