# First-use tuning of lowered modules (NEPTUNE_HIP_TUNE=1), default build (4 tiles) and full build (all tiles):
# per-call time of @entry at sizes where the fixed default tile is not the best one.  Output: gpurun_out/tune_bench.txt
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/tune_bench.txt
for n in 320 384 640; do
  python tools/make_stencil_mlir.py 3d7 $n > /tmp/t$n.mlir
  for mode in default tune tune_full; do
    unset NEPTUNE_HIP_TUNE NEPTUNE_HIP_FULL_VARIANTS
    [ $mode = tune ] && export NEPTUNE_HIP_TUNE=1
    [ $mode = tune_full ] && export NEPTUNE_HIP_TUNE=1 NEPTUNE_HIP_FULL_VARIANTS=1
    echo "== 3d7 ${n}^3 $mode" >> gpurun_out/tune_bench.txt
    timeout -k 10 400 python tools/time_module.py /tmp/t$n.mlir --reps 50 2>&1 | grep -v amdgpu.ids | cut -c1-120 >> gpurun_out/tune_bench.txt || exit 1
  done
done
