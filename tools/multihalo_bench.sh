# Run the several-halo-input parity tests, then time each case at bench size for every default tile.
mkdir -p gpurun_out
set -e
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_multihalo_gpu.py -x -q > gpurun_out/mh_tests.log 2>&1
python tools/make_multihalo_mlir.py swe3d_two_stars 512 512 512 > /tmp/swe3d.mlir
python tools/make_multihalo_mlir.py swe2d 8192 8192 > /tmp/swe2d.mlir
python tools/make_multihalo_mlir.py four_halo_inputs_2d 8192 8192 > /tmp/four2d.mlir
python tools/make_multihalo_mlir.py two_boxes_f32 512 512 512 > /tmp/box2.mlir
python tools/make_multihalo_mlir.py pair_1d 134217728 > /tmp/pair1d.mlir
: > gpurun_out/mh_time.log
for m in swe3d swe2d four2d box2 pair1d; do
  for v in auto 0 1 2 direct; do
    unset NEPTUNE_HIP_VARIANT NEPTUNE_HIP_KERNEL
    if [ $v = direct ]; then export NEPTUNE_HIP_KERNEL=direct; elif [ $v != auto ]; then export NEPTUNE_HIP_VARIANT=$v; fi
    echo "== $m variant=$v" >> gpurun_out/mh_time.log
    timeout -k 10 300 python tools/time_module.py /tmp/$m.mlir --reps 20 2>&1 | cut -c1-200 >> gpurun_out/mh_time.log
  done
done
