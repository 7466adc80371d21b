# Time the fused explicit Euler step (one kernel: rhs apply + axpy) against the plain operator apply
# on the bench sizes, through lowered modules.  Output: gpurun_out/timestep.log
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/timestep.log
python tools/make_stencil_mlir.py 3d7 1024 --time-step 0.125 > /tmp/ts3d.mlir
python tools/make_stencil_mlir.py 2d5 8192 --time-step 0.125 > /tmp/ts2d.mlir
python tools/make_stencil_mlir.py 3d7 512 --time-step 0.125 > /tmp/ts3d512.mlir
for m in ts3d ts3d512 ts2d; do
  for sym in entry step; do
    echo "== $m @$sym" >> gpurun_out/timestep.log
    timeout -k 10 300 python tools/time_module.py /tmp/$m.mlir --symbol $sym --reps 20 2>&1 | grep -v amdgpu.ids | cut -c1-160 >> gpurun_out/timestep.log || exit 1
  done
done
