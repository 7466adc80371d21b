// rt_body_lap3d27.hip -- built-in body Lap3D27 (builtin_bodies.hpp) on every march tile of the library: see rt_bodies.hpp.
#define NEPTUNE_HIP_FULL_VARIANTS 1
#include <hip/hip_runtime.h>

#include "../kernels/apply_launch.hpp"
#include "builtin_bodies.hpp"
#include "rt_bodies.hpp"

namespace neptune_hip {
namespace rtbody {
namespace {
using B = builtin::Lap3D27;
int apply(const neptune_hip_apply_geom_t* g, const void* const* in, void* out, hipStream_t stream, const neptune_hip_launch_cfg_t* cfg) {
  const int rc = geom_check_radius(g, B::radius);
  if (rc != NEPTUNE_HIP_OK) return rc;
  return launch_apply<B, B::T, B::RANK, B::NIN, B::FP>(B{}, g, in, out, stream, cfg);
}
int plan(const neptune_hip_apply_geom_t* g, const void* const* in, const void* out, const neptune_hip_launch_cfg_t* cfg) {
  const int rc = geom_check_radius(g, B::radius);
  if (rc != NEPTUNE_HIP_OK) return rc;
  return plan_apply<B::T, B::RANK, B::NIN, B::FP>(g, in, out, cfg);
}
int variant(const neptune_hip_apply_geom_t* g, const neptune_hip_launch_cfg_t* cfg) { return pick_march_variant<B::T, B::RANK, B::FP>(g, cfg); }
}  // namespace
const Entry& lap3d27() {
  static const Entry e = {apply, plan, variant, nullptr};
  return e;
}
}  // namespace rtbody
}  // namespace neptune_hip
