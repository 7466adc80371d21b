#!/usr/bin/env bash
# Profile bench.py on the GPU box: kernel-trace stats pass + separate PMC passes (FETCH_SIZE and
# WRITE_SIZE cannot share a pass on gfx950: TCC has 4 slots, they need 3 + 2).
# usage: tools/profile_bench.sh <tag> [bench args...]      -> gpurun_out/prof_<tag>/
set -euo pipefail
TAG=${1:?tag}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS=("$@")
if [ ${#ARGS[@]} -eq 0 ]; then ARGS=(--steps 10 --warmup 2 --no-cpu-baseline); fi
# fill the module cache OUTSIDE the profiler: the profiled runs must not start hipcc
python3 "$ROOT/bench.py" "${ARGS[@]}" --compile-only
echo "== stats pass" | tee "$OUT/README.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" "${ARGS[@]}" > "$OUT/bench_stats.json" 2> "$OUT/stats.err" || { tail -20 "$OUT/stats.err"; exit 1; }
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  N=$(echo "$C" | tr ' ' '_')
  echo "== pmc pass $C" | tee -a "$OUT/README.txt"
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$N" -- python3 "$ROOT/bench.py" "${ARGS[@]}" > "$OUT/bench_pmc_$N.json" 2> "$OUT/pmc_$N.err" || { tail -20 "$OUT/pmc_$N.err"; echo "pmc pass $C failed (continuing)"; }
done
python3 "$ROOT/tools/profile_summary.py" "$OUT" | tee "$OUT/summary.txt"
