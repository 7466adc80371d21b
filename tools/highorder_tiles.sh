set -e
export TMPDIR=/tmp NEPTUNE_HIP_FULL_VARIANTS=1 NEPTUNE_HIP_TUNE=0
N=512
python tools/make_multihalo_mlir.py radius3_3d $N $N $N > /tmp/r3_3d.mlir
python tools/make_multihalo_mlir.py radius4_3d $N $N $N > /tmp/r4_3d.mlir
python tools/make_multihalo_mlir.py radius4_3d_f32 $N $N $N > /tmp/r4_3d_f32.mlir
: > gpurun_out/r03_hi_time.log
for m in r4_3d r3_3d r4_3d_f32; do
  for v in 7 37 39 42 44 45 46; do
    for c in 0 256; do
      export NEPTUNE_HIP_VARIANT=$v NEPTUNE_HIP_CHUNK=$c
      echo "== $m variant=$v chunk=$c" >> gpurun_out/r03_hi_time.log
      timeout -k 10 300 python tools/time_module.py /tmp/$m.mlir --reps 30 2>&1 | grep -o '"ms_per_call": [0-9.]*, "GBps": [0-9.]*\|Error.*\|error.*' >> gpurun_out/r03_hi_time.log || true
    done
  done
done
cat gpurun_out/r03_hi_time.log
