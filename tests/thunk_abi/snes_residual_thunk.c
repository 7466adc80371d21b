/*
 * snes_residual_thunk.c -- a C caller in the shape of the reference's SNES residual callback
 * (lib/Runtime/PETSc/NeptunePETScRuntime.cpp:1303-1361, NL<2, Caps>::FormFunction; the two-capture form follows at :1363):
 *   - the residual is a lowered `nonlinear_opdef`, found with dlsym(RTLD_DEFAULT, symbol)                  (:1459)
 *   - arguments: the iterate x and every capture as EXPANDED rank-2 memrefs
 *       (allocated, aligned, offset, size0, size1, stride0, stride1)  -- view_x2D over the Vec array       (:881-892)
 *   - the result comes back as a NeptuneMemRef2D by value, is packed into the residual Vec through its own
 *     sizes / strides (pack2D_to_contig_rm) and then freed by the CALLER                                   (:1351-1352)
 * Two modes:
 *   host    host arrays in the memref arguments, result released with plain free() -- exactly what the reference's
 *           thunk does, so that thunk runs unchanged against a module lowered by this backend (arguments are staged)
 *   device  DEVICE pointers in the memref arguments (an iterate that already lives on the GPU), result is device
 *           memory, copied out with neptune_hip_memcpy_d2h and released with neptune_rt_free
 *
 * usage: snes_residual_thunk <module.so> <symbol> <s0> <s1> <x.bin> <cap0.bin> <cap1.bin> <F_out.bin> host|device
 * The callback runs three times (a Newton iteration calls it once per residual evaluation); F_out.bin receives the
 * last result, s0*s1 doubles row-major.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "neptune_hip.h"

typedef NeptuneMemRef2D (*ResidualFn)(void *x_alloc, void *x_aligned, int64_t x_off, int64_t x_s0, int64_t x_s1, int64_t x_st0,
                                      int64_t x_st1, void *c0_alloc, void *c0_aligned, int64_t c0_off, int64_t c0_s0,
                                      int64_t c0_s1, int64_t c0_st0, int64_t c0_st1, void *c1_alloc, void *c1_aligned,
                                      int64_t c1_off, int64_t c1_s0, int64_t c1_s1, int64_t c1_st0, int64_t c1_st1);

static NeptuneMemRef2D view2d(void *p, int64_t s0, int64_t s1) {   /* view_x2D: dense row-major view of a Vec array */
  NeptuneMemRef2D m;
  m.allocated = p;
  m.aligned = p;
  m.offset = 0;
  m.sizes[0] = s0;
  m.sizes[1] = s1;
  m.strides[0] = s1;
  m.strides[1] = 1;
  return m;
}

static double *read_doubles(const char *path, size_t n) {
  double *a = (double *)malloc(n * sizeof(double));
  FILE *f = fopen(path, "rb");
  if (!a || !f || fread(a, sizeof(double), n, f) != n) { fprintf(stderr, "cannot read %s\n", path); exit(3); }
  fclose(f);
  return a;
}

int main(int argc, char **argv) {
  if (argc != 10) return 2;
  if (!dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL)) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
  ResidualFn fn = (ResidualFn)dlsym(RTLD_DEFAULT, argv[2]);
  if (!fn) { fprintf(stderr, "[NeptuneRT] dlsym failed for %s\n", argv[2]); abort(); }
  const int64_t s0 = atoll(argv[3]), s1 = atoll(argv[4]);
  const size_t n = (size_t)(s0 * s1), bytes = n * sizeof(double);
  const int device = strcmp(argv[9], "device") == 0;
  double *hx = read_doubles(argv[5], n), *hc0 = read_doubles(argv[6], n), *hc1 = read_doubles(argv[7], n);
  double *F = (double *)malloc(bytes);
  void *x = hx, *c0 = hc0, *c1 = hc1;
  if (device) {
    neptune_hip_init(0);
    x = neptune_hip_malloc(bytes); c0 = neptune_hip_malloc(bytes); c1 = neptune_hip_malloc(bytes);
    neptune_hip_memcpy_h2d(x, hx, bytes, NULL);
    neptune_hip_memcpy_h2d(c0, hc0, bytes, NULL);
    neptune_hip_memcpy_h2d(c1, hc1, bytes, NULL);
    neptune_hip_device_sync();
  }
  const NeptuneMemRef2D cap0 = view2d(c0, s0, s1), cap1 = view2d(c1, s0, s1);   /* the captures: bound once, like Ctx::cap0 */
  for (int eval = 0; eval < 3; ++eval) {                                        /* FormFunction, once per residual evaluation */
    const NeptuneMemRef2D xin = view2d(x, s0, s1);
    NeptuneMemRef2D fout = fn(xin.allocated, xin.aligned, xin.offset, xin.sizes[0], xin.sizes[1], xin.strides[0], xin.strides[1],
                              cap0.allocated, cap0.aligned, cap0.offset, cap0.sizes[0], cap0.sizes[1], cap0.strides[0], cap0.strides[1],
                              cap1.allocated, cap1.aligned, cap1.offset, cap1.sizes[0], cap1.sizes[1], cap1.strides[0], cap1.strides[1]);
    if (fout.sizes[0] != s0 || fout.sizes[1] != s1) { fprintf(stderr, "residual has the wrong shape\n"); return 4; }
    if (device) {
      if (!neptune_hip_is_device_ptr(fout.aligned)) { fprintf(stderr, "device arguments but a host result\n"); return 5; }
      if (fout.strides[1] != 1 || fout.strides[0] != s1 || fout.offset != 0) { fprintf(stderr, "result is not dense row-major\n"); return 6; }
      neptune_hip_memcpy_d2h(F, fout.aligned, bytes, NULL);
      neptune_hip_device_sync();
      neptune_rt_free(fout.allocated);
    } else {
      /* pack2D_to_contig_rm: through the result's own offset and strides */
      const double *src = (const double *)fout.aligned + fout.offset;
      for (int64_t i = 0; i < s0; ++i)
        for (int64_t j = 0; j < s1; ++j) F[i * s1 + j] = src[i * fout.strides[0] + j * fout.strides[1]];
      free(fout.allocated);   /* the reference releases the callback's result with plain free() */
    }
  }
  FILE *f = fopen(argv[8], "wb");
  if (!f || fwrite(F, 1, bytes, f) != bytes) return 7;
  fclose(f);
  if (device) {
    printf("pool_cached_bytes %zu\n", neptune_hip_pool_cached_bytes());
    neptune_hip_free(x); neptune_hip_free(c0); neptune_hip_free(c1);
  }
  free(hx); free(hc0); free(hc1); free(F);
  printf("SNES_THUNK_OK %s\n", argv[9]);
  return 0;
}
