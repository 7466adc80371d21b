// rt_bodies.hpp -- the built-in bodies of the runtime library, one translation unit each (rt_body_*.hip): every
// march tile x every body is minutes of device code generation, which `make -j` now spreads over the host's cores.
// neptune_hip_rt.hip dispatches on the body id through these tables.
#pragma once
#include "../../../include/neptune_hip.h"

namespace neptune_hip {
namespace rtbody {
struct Entry {
  // launch (geometry already validated against the body's radii by the callee), plan, automatic tile
  int (*apply)(const neptune_hip_apply_geom_t*, const void* const*, void*, hipStream_t, const neptune_hip_launch_cfg_t*);
  int (*plan)(const neptune_hip_apply_geom_t*, const void* const*, const void*, const neptune_hip_launch_cfg_t*);
  int (*variant)(const neptune_hip_apply_geom_t*, const neptune_hip_launch_cfg_t*);
  // two / three chained applies in one launch (apply_march2.hpp); nullptr for bodies without that form
  int (*chain)(int applies, const neptune_hip_apply_geom_t*, const void* const*, void*, hipStream_t, const neptune_hip_launch_cfg_t*);
};
const Entry& lap2d5();
const Entry& lap3d7();
const Entry& lap3d27();
const Entry& lap1d3();
}  // namespace rtbody
}  // namespace neptune_hip
