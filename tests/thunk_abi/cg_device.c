/*
 * cg_device.c -- a Krylov loop in C around a lowered operator, with every vector resident in device memory.
 *
 * The reference's matrix-free solve reaches the operator from PETSc's KSP through a MatShell thunk
 * (lib/Runtime/PETSc/NeptunePETScRuntime.cpp:182-230: dlsym of the symbol, expanded memref arguments over the Vec
 * arrays, result copied out); its Vecs live on the host, and a backend that stages them would move 2 N values over
 * PCIe per iteration.  Here the same ABI is called with DEVICE pointers in the memref arguments:
 *   @matmult(y, x)  writes A x straight into y (store elision), @dot(a, b) is one fused reduce(apply) kernel that
 *   returns 8 bytes, and the vector updates are neptune_hip_axpy / neptune_hip_xpay.
 * Nothing of size N crosses PCIe inside the loop; the block pool staying empty is the evidence (host arguments would
 * get pooled device shadows).
 *
 * usage: cg_device <module.so> <n0> <n1> <n2> <iterations> <b.bin> <x_out.bin>
 *   b.bin: n0*n1*n2 doubles (zero on the rim); x_out.bin receives the iterate after `iterations` steps
 *   stdout: one line per iteration "it <k> rs <hex float>", then "pool_cached_bytes <n>"
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "neptune_hip.h"

typedef NeptuneMemRef3D (*MatMultFn)(void *, void *, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, /* y */
                                     void *, void *, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t  /* x */);
typedef double (*DotFn)(void *, void *, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                        void *, void *, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t);

static int64_t N0, N1, N2;
static MatMultFn matmult;
static DotFn dot;

static void A(void *y, void *x) {
  NeptuneMemRef3D r = matmult(y, y, 0, N0, N1, N2, N1 * N2, N2, 1, x, x, 0, N0, N1, N2, N1 * N2, N2, 1);
  if (r.aligned != y) { fprintf(stderr, "matmult did not return its destination\n"); exit(4); }
}
static double DOT(void *a, void *b) {
  return dot(a, a, 0, N0, N1, N2, N1 * N2, N2, 1, b, b, 0, N0, N1, N2, N1 * N2, N2, 1);
}

int main(int argc, char **argv) {
  if (argc != 8) return 2;
  if (!dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL)) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
  matmult = (MatMultFn)dlsym(RTLD_DEFAULT, "matmult");
  dot = (DotFn)dlsym(RTLD_DEFAULT, "dot");
  if (!matmult || !dot) { fprintf(stderr, "[NeptuneRT] dlsym failed\n"); abort(); }
  N0 = atoll(argv[2]); N1 = atoll(argv[3]); N2 = atoll(argv[4]);
  const int iters = atoi(argv[5]);
  const int64_t n = N0 * N1 * N2;
  const size_t bytes = (size_t)n * sizeof(double);
  neptune_hip_init(0);
  double *host = (double *)malloc(bytes);
  FILE *f = fopen(argv[6], "rb");
  if (!f || fread(host, 1, bytes, f) != bytes) { fprintf(stderr, "cannot read %s\n", argv[6]); return 3; }
  fclose(f);
  void *b = neptune_hip_malloc(bytes), *x = neptune_hip_malloc(bytes), *r = neptune_hip_malloc(bytes),
       *p = neptune_hip_malloc(bytes), *ap = neptune_hip_malloc(bytes);
  neptune_hip_memcpy_h2d(b, host, bytes, NULL);          /* the right-hand side: once, before the loop */
  memset(host, 0, bytes);
  neptune_hip_memcpy_h2d(x, host, bytes, NULL);          /* x0 = 0 */
  neptune_hip_memcpy_d2d(r, b, bytes, NULL);             /* r = b - A x0 = b */
  neptune_hip_memcpy_d2d(p, r, bytes, NULL);
  neptune_hip_device_sync();
  double rs = DOT(r, r);
  for (int k = 0; k < iters; ++k) {
    A(ap, p);
    const double alpha = rs / DOT(p, ap);
    neptune_hip_axpy(NEPTUNE_HIP_F64, n, alpha, p, x, NULL);      /* x += alpha p  */
    neptune_hip_axpy(NEPTUNE_HIP_F64, n, -alpha, ap, r, NULL);    /* r -= alpha Ap */
    const double rs_new = DOT(r, r);
    neptune_hip_xpay(NEPTUNE_HIP_F64, n, r, rs_new / rs, p, NULL); /* p = r + beta p */
    rs = rs_new;
    printf("it %d rs %a\n", k + 1, rs);
  }
  neptune_hip_device_sync();
  printf("pool_cached_bytes %zu\n", neptune_hip_pool_cached_bytes());
  neptune_hip_memcpy_d2h(host, x, bytes, NULL);
  neptune_hip_device_sync();
  f = fopen(argv[7], "wb");
  if (!f || fwrite(host, 1, bytes, f) != bytes) return 5;
  fclose(f);
  neptune_hip_free(b); neptune_hip_free(x); neptune_hip_free(r); neptune_hip_free(p); neptune_hip_free(ap);
  free(host);
  return 0;
}
