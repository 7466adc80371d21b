# Time the high-order stars (radius 3-4; 3-D radius 3) and the 2-D 25-point box at bench size: automatic tile, each default tile, direct kernel.
mkdir -p gpurun_out
set -e
export TMPDIR=/tmp
python tools/make_multihalo_mlir.py radius4_2d 8192 8192 > /tmp/r4_2d.mlir
python tools/make_multihalo_mlir.py radius4_2d_f32 8192 8192 > /tmp/r4_2d_f32.mlir
python tools/make_multihalo_mlir.py radius3_2d_ragged 8193 8193 > /tmp/r3_2d_ragged.mlir
python tools/make_multihalo_mlir.py radius4_1d 134217728 > /tmp/r4_1d.mlir
python tools/make_multihalo_mlir.py radius8_1d_f32 134217728 > /tmp/r8_1d_f32.mlir
python tools/make_multihalo_mlir.py radius3_3d 512 512 512 > /tmp/r3_3d.mlir
python tools/make_multihalo_mlir.py radius3_3d_f32_ragged 513 513 513 > /tmp/r3_3d_f32_ragged.mlir
python tools/make_multihalo_mlir.py box25_2d 8192 8192 > /tmp/box25_2d.mlir
python tools/make_multihalo_mlir.py box25_2d_f32_ragged 8193 8193 > /tmp/box25_2d_f32_ragged.mlir
python tools/make_multihalo_mlir.py radius4_3d 512 512 512 > /tmp/r4_3d.mlir
python tools/make_multihalo_mlir.py radius4_3d_f32 512 512 512 > /tmp/r4_3d_f32.mlir
: > gpurun_out/ho_time.log
for m in ${CASES:-r4_2d r4_2d_f32 r3_2d_ragged r4_1d r8_1d_f32 r3_3d r3_3d_f32_ragged box25_2d box25_2d_f32_ragged r4_3d r4_3d_f32}; do
  for v in ${VARIANTS:-auto 0 1 2 3 direct}; do
    unset NEPTUNE_HIP_VARIANT NEPTUNE_HIP_KERNEL
    if [ $v = direct ]; then export NEPTUNE_HIP_KERNEL=direct; elif [ $v != auto ]; then export NEPTUNE_HIP_VARIANT=$v; fi
    echo "== $m variant=$v" >> gpurun_out/ho_time.log
    timeout -k 10 300 python tools/time_module.py /tmp/$m.mlir --reps 20 2>&1 | cut -c1-200 >> gpurun_out/ho_time.log
  done
done
