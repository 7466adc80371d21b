#!/usr/bin/env bash
# Round profile set: the default bench command (plan-time tuning on) and, per workload, the library's default
# tile (--no-autotune) with separate PMC passes.  Output under gpurun_out/prof_<tag>/ (copy summaries to profiles/).
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
# (--no-extra-configs --no-verify: the profiled process runs the headline's launches only; the other configurations get a
# profile of their own below, plan-time tuned like the driver's config.configs entries)
bash tools/profile_bench.sh 3d7_1024_default --steps 50 --warmup 5 --no-cpu-baseline --no-extra-configs --no-verify > gpurun_out/prof_3d7_1024_default.log 2>&1 || exit 1
for W in 2d5_8192 3d27_512 3d7_512; do
  bash tools/profile_bench.sh $W --workload $W --steps 200 --warmup 5 --no-cpu-baseline --no-verify > gpurun_out/prof_$W.log 2>&1 || exit 1
done
echo PROFILE_ALL_OK
