#!/usr/bin/env bash
# End-to-end smoke test in the shape of the reference's test/smoke_tests/smoke_apply.sh:
#   lower the module -> build the kernel library -> generate a tiny C++ driver -> link -> run.
# The reference pipeline was  neptune-opt --neptuneir-to-llvm | mlir-translate | llvm-as | llc | clang++ ;
# here steps 1-3 are  neptune-opt --neptuneir-to-hip --emit=so .  The driver is the reference's driver
# (same prototype, same in[i] = i+1 / out[i] = 0 convention); unlike the reference script this one ASSERTS.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NEPTUNE_OPT=${NEPTUNE_OPT:-$ROOT/neptune-pde-solver_amd/bin/neptune-opt}
CXX=${CXX:-g++}
INPUT_MLIR=${1:-$ROOT/tests/smoke_tests/smoke_apply_1d.mlir}
WORKDIR=${WORKDIR:-$(mktemp -d /tmp/neptune_smoke_apply_hip.XXXXXX)}
mkdir -p "$WORKDIR" && cd "$WORKDIR"

echo "[1/4] Lower to HIP and build the kernel library"
"$NEPTUNE_OPT" "$INPUT_MLIR" --neptuneir-to-hip --emit=so -o "$WORKDIR/kernel.so" --report

echo "[2/4] Generate tiny driver"
cat > driver.cpp <<'CPP'
#include <cstdint>
#include <cstdio>
#include <cstdlib>
struct MemRef1D { void* allocated; void* aligned; int64_t offset; int64_t sizes[1]; int64_t strides[1]; };
// prototype = the LLVM-dialect expansion of (memref<?xf64>, memref<?xf64>) -> memref<?xf64>: 10 args
extern "C" MemRef1D entry(void* a_alloc, void* a_aligned, int64_t a_off, int64_t a_s0, int64_t a_st0,
                          void* b_alloc, void* b_aligned, int64_t b_off, int64_t b_s0, int64_t b_st0);
int main() {
  const int64_t n = 16;
  double* out = (double*)aligned_alloc(64, sizeof(double) * n);
  double* rhs = (double*)aligned_alloc(64, sizeof(double) * n);
  for (int i = 0; i < n; ++i) { out[i] = 0.0; rhs[i] = (double)(i + 1); }
  MemRef1D r = entry(out, out, 0, n, 1, rhs, rhs, 0, n, 1);
  std::printf("[driver] ret aligned=%p size=%ld stride=%ld off=%ld\n", r.aligned, (long)r.sizes[0], (long)r.strides[0], (long)r.offset);
  double* x = (double*)r.aligned;
  int bad = (r.aligned != (void*)out) || r.sizes[0] != n;          // entry returns its destination field
  for (int i = 0; i < (int)r.sizes[0]; ++i) {
    std::printf("x[%d]=%.6f\n", i, x[i]);
    // Lap(u)[i] = 100 * ((u[i-1] - 2 u[i]) + u[i+1]) = 0 on the ramp u = i+1; ends are copy-through
    const double want = (i == 0 || i == n - 1) ? (double)(i + 1) : 0.0;
    if (x[i] != want) bad = 1;
  }
  std::puts(bad ? "SMOKE_FAIL" : "SMOKE_OK");
  return bad;
}
CPP

echo "[3/4] Link test exe (-rdynamic, as the reference does for dlsym)"
"$CXX" -O2 driver.cpp "$WORKDIR/kernel.so" -L"$ROOT/neptune-pde-solver_amd/lib" -lneptune_hip \
  -Wl,-rpath,"$ROOT/neptune-pde-solver_amd/lib" -Wl,-rpath,"$WORKDIR" -rdynamic -ldl -o smoke_apply_test

echo "[4/4] RUN ./smoke_apply_test"
./smoke_apply_test
