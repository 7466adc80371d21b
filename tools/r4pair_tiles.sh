# Several inputs read at offsets on the plane-in-LDS kernel: the radius-4 pair and the two-7-point-stars residual at 512^3, automatic tile and a few
# forced ones (NEPTUNE_HIP_FULL_VARIANTS=1: 39 = pln_rj2_wj8_wk1_pf2, 43 = pln_rj1_wj16_wk1_pf2), next to the direct kernel.
set -e
export TMPDIR=/tmp NEPTUNE_HIP_FULL_VARIANTS=1 NEPTUNE_HIP_TUNE=0
python tools/make_multihalo_mlir.py radius4_pair_3d 512 512 512 > /tmp/r4p.mlir
python tools/make_multihalo_mlir.py swe3d_two_stars 512 512 512 > /tmp/swe.mlir
python tools/make_multihalo_mlir.py radius2_pair_3d 512 512 512 > /tmp/r2p.mlir
: > gpurun_out/r03_r4pair.log
for m in r4p swe r2p; do
for v in auto 7 39 direct; do
  unset NEPTUNE_HIP_VARIANT NEPTUNE_HIP_KERNEL
  if [ $v = direct ]; then export NEPTUNE_HIP_KERNEL=direct; elif [ $v != auto ]; then export NEPTUNE_HIP_VARIANT=$v; fi
  echo "== $m variant=$v" >> gpurun_out/r03_r4pair.log
  timeout -k 10 300 python tools/time_module.py /tmp/$m.mlir --reps 20 2>&1 | grep -o '"ms_per_call": [0-9.]*, "GBps": [0-9.]*\|"variant": "[a-z0-9_]*"\|Error.*\|error.*' >> gpurun_out/r03_r4pair.log || true
done
done
cat gpurun_out/r03_r4pair.log
