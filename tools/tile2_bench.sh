# Rank-2 LDS tile kernel (neptune_apply_tile2) at bench size against the direct kernel: 2-D stars beyond radius 4, boxes beyond
# radius 2, several inputs read at wide offsets.     usage: bash tools/tile2_bench.sh   -> gpurun_out/tile2_time.log
mkdir -p gpurun_out
set -e
export TMPDIR=/tmp
N=${N:-8192}
: > gpurun_out/tile2_time.log
for c in ${CASES:-radius5_2d radius8_2d_f32_ragged box49_2d box_radius4_sparse_2d_f32 four_radius3_2d radius3_pair_2d box25_pair_2d}; do
  M=$N; case $c in *ragged) M=$((N+1));; esac
  python tools/make_multihalo_mlir.py $c $M $M > /tmp/$c.mlir
  for v in auto direct; do
    unset NEPTUNE_HIP_VARIANT NEPTUNE_HIP_KERNEL
    if [ $v = direct ]; then export NEPTUNE_HIP_KERNEL=direct; fi
    echo "== $c $v $(timeout -k 10 300 python tools/time_module.py /tmp/$c.mlir --reps 20 2>&1 | grep -o '"ms_per_call": [0-9.]*, "GBps": [0-9.]*')" >> gpurun_out/tile2_time.log
  done
done
cat gpurun_out/tile2_time.log
