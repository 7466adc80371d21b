/*
 * matmult_thunk.c -- a C caller that reaches a lowered operator the way the reference's PETSc
 * runtime does, to show the lowered-function ABI is a drop-in for that caller:
 *   - symbol lookup through dlsym(RTLD_DEFAULT, name)      NeptunePETScRuntime.cpp:22-30, 752-755
 *   - argument = expanded rank-1 memref over a HOST array  :198-212 (view of the PETSc Vec)
 *   - result copied out, then free(result.allocated)       :215-221
 * usage: matmult_thunk <module.so> <symbol> <n>   (input x[i] = i+1, the smoke drivers' convention)
 * prints y[i] as C99 hex floats, one per line.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  void *allocated;
  void *aligned;
  int64_t offset;
  int64_t sizes[1];
  int64_t strides[1];
} MemRef1D;

typedef MemRef1D (*LinearFn)(void *alloc, void *aligned, int64_t offset, int64_t size0, int64_t stride0);

int main(int argc, char **argv) {
  if (argc != 4) return 2;
  /* RTLD_GLOBAL: the operator must be visible to dlsym(RTLD_DEFAULT, ...) like a -rdynamic link */
  if (!dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL)) {
    fprintf(stderr, "dlopen: %s\n", dlerror());
    return 1;
  }
  LinearFn fn = (LinearFn)dlsym(RTLD_DEFAULT, argv[2]);
  if (!fn) {
    fprintf(stderr, "[NeptuneRT] dlsym failed for %s\n", argv[2]);
    abort();
  }
  const int64_t n = atoll(argv[3]);
  double *x = (double *)malloc(sizeof(double) * (size_t)n);
  double *y = (double *)malloc(sizeof(double) * (size_t)n);
  for (int64_t i = 0; i < n; ++i) x[i] = (double)(i + 1);
  for (int rep = 0; rep < 3; ++rep) { /* a Krylov loop calls it once per iteration */
    MemRef1D yout = fn(x, x, 0, n, 1);
    const double *src = (const double *)yout.aligned + yout.offset;
    memcpy(y, src, sizeof(double) * (size_t)n);
    free(yout.allocated); /* the reference frees the operator's result with plain free() */
  }
  for (int64_t i = 0; i < n; ++i) printf("%a\n", y[i]);
  free(x);
  free(y);
  return 0;
}
