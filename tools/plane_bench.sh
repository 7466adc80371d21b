# Plane-in-LDS kernel (apply_plane.hpp) at bench size: 3-D stars of radius 2..8 on its tiles, next to the march tiles that
# could hold them before (2: radius 2-3, 5: radius 4) and the direct kernel.   VARIANTS / CASES select; full variant list
# with NEPTUNE_HIP_FULL_VARIANTS=1 (7 37 38 39 40 41 are the plane tiles).
mkdir -p gpurun_out
set -e
export TMPDIR=/tmp
N=${N:-512}
python tools/make_multihalo_mlir.py radius2_3d $N $N $N > /tmp/r2_3d.mlir
python tools/make_multihalo_mlir.py radius3_3d $N $N $N > /tmp/r3_3d.mlir
python tools/make_multihalo_mlir.py radius4_3d $N $N $N > /tmp/r4_3d.mlir
python tools/make_multihalo_mlir.py radius4_3d_f32 $N $N $N > /tmp/r4_3d_f32.mlir
python tools/make_multihalo_mlir.py radius4_3d_leapfrog $N $N $N > /tmp/r4_3d_leapfrog.mlir
python tools/make_multihalo_mlir.py radius5_3d $N $N $N > /tmp/r5_3d.mlir
python tools/make_multihalo_mlir.py radius6_3d_f32_ragged $((N+1)) $((N+1)) $((N+1)) > /tmp/r6_3d_f32_ragged.mlir
python tools/make_multihalo_mlir.py radius8_3d $N $N $N > /tmp/r8_3d.mlir
python tools/make_multihalo_mlir.py radius8_3d_f32 $N $N $N > /tmp/r8_3d_f32.mlir
: > gpurun_out/plane_time.log
for m in ${CASES:-r2_3d r3_3d r4_3d r4_3d_f32 r4_3d_leapfrog r5_3d r6_3d_f32_ragged r8_3d r8_3d_f32}; do
  for v in ${VARIANTS:-auto 2 5 7 direct}; do
    unset NEPTUNE_HIP_VARIANT NEPTUNE_HIP_KERNEL
    if [ $v = direct ]; then export NEPTUNE_HIP_KERNEL=direct; elif [ $v != auto ]; then export NEPTUNE_HIP_VARIANT=$v; fi
    echo "== $m variant=$v" >> gpurun_out/plane_time.log
    timeout -k 10 300 python tools/time_module.py /tmp/$m.mlir --reps 20 2>&1 | grep -o '"ms_per_call": [0-9.]*, "GBps": [0-9.]*\|"variant": "[a-z0-9_]*"\|Error.*' >> gpurun_out/plane_time.log || true
  done
done
