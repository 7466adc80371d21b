#!/usr/bin/env bash
# usage: tools/variant_traffic.sh WORKLOAD...   -> gpurun_out/vt_<workload>/{fetch,write}, gpurun_out/vt_<workload>.txt
set -euo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
  OUT=$ROOT/gpurun_out/vt_$W
  mkdir -p "$OUT"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/tools/variant_traffic.py" run "$W" > "$OUT/fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 "$ROOT/tools/variant_traffic.py" run "$W" > "$OUT/write.log" 2>&1
  python3 "$ROOT/tools/variant_traffic.py" parse "$W" "$OUT/fetch" "$OUT/write" ${UPDATE:+--update "$UPDATE"} | tee "$ROOT/gpurun_out/vt_$W.txt"
  # keep only the counter tables (the raw dirs also hold large agent-info files)
  find "$OUT" -name "*agent_info*" -delete
done
