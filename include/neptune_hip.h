/*
 * neptune_hip.h -- C ABI of the MI355X (gfx950) backend for NeptuneIR's stencil hot path.
 *
 * This is the drop-in boundary.  Everything a host program (emitted host C++, the Python
 * frontend through ctypes, a PETSc MatMult thunk, the bench) needs is reachable through
 * the plain-C entry points below: plain pointers and sizes, no C++ or torch types.
 *
 * The reference project (levia-than/neptune-pde-solver) lowers `neptune_ir.apply` & co to
 * scalar CPU loop nests (lib/Passes/DataflowLowering.cpp:258-448).  The entry points here
 * are what a `backend=hip` variant of that lowering calls instead.  Each declaration cites
 * the reference construct it replaces.
 *
 * Error convention: like the reference runtime (NeptunePETScRuntime.cpp:15-30), ABI
 * functions that cannot report an error print "[NeptuneRT][HIP] ..." to stderr and abort().
 * Functions returning `int` return 0 on success and a negative NEPTUNE_HIP_E* code when the
 * request is rejected before anything is launched (bad geometry, unsupported shape).
 */
#ifndef NEPTUNE_HIP_H
#define NEPTUNE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------
 * 1. memref descriptors of lowered functions
 *    Same field order as the reference's MLIR->LLVM memref ABI
 *    (include/Runtime/PETSc/NeptunePETScRuntime.h:22-42; rank 3 added the same way).
 *    A rank-r memref *argument* is passed expanded:
 *      (void* allocated, void* aligned, int64 offset, int64 size[0..r), int64 stride[0..r))
 *    and a memref *result* is returned as the struct by value
 *    (driver prototype: test/smoke_tests/smoke_apply.sh:39-50).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  void *allocated;
  void *aligned;
  int64_t offset;
  int64_t sizes[1];
  int64_t strides[1];
} NeptuneMemRef1D;

typedef struct {
  void *allocated;
  void *aligned;
  int64_t offset;
  int64_t sizes[2];
  int64_t strides[2];
} NeptuneMemRef2D;

typedef struct {
  void *allocated;
  void *aligned;
  int64_t offset;
  int64_t sizes[3];
  int64_t strides[3];
} NeptuneMemRef3D;
/* rank 4..6 (fields with leading batch / component dimensions: the lowering peels them off and launches one rank-3
 * apply per leading index, see neptune-pde-solver_amd/csrc/runtime/lowered_runtime.hpp run_apply_batched) */
typedef struct {
  void *allocated;
  void *aligned;
  int64_t offset;
  int64_t sizes[4];
  int64_t strides[4];
} NeptuneMemRef4D;
typedef struct {
  void *allocated;
  void *aligned;
  int64_t offset;
  int64_t sizes[5];
  int64_t strides[5];
} NeptuneMemRef5D;
typedef struct {
  void *allocated;
  void *aligned;
  int64_t offset;
  int64_t sizes[6];
  int64_t strides[6];
} NeptuneMemRef6D;

/* ------------------------------------------------------------------------------------
 * 2. constants
 * ---------------------------------------------------------------------------------- */
#define NEPTUNE_HIP_MAX_RANK 3
#define NEPTUNE_HIP_MAX_INPUTS 4

#define NEPTUNE_HIP_OK 0
#define NEPTUNE_HIP_EINVAL (-1)      /* malformed geometry / null pointer            */
#define NEPTUNE_HIP_EUNSUPPORTED (-2) /* valid request this build cannot serve         */
#define NEPTUNE_HIP_EOOB (-3)        /* an access would leave its input's box (UB in the
                                        reference, DataflowLowering.cpp:382-410: no
                                        bounds check; rejected here at plan time)      */
#define NEPTUNE_HIP_ECOMM (-4)       /* RCCL / stream failure in the slab halo exchange; text in
                                        neptune_hip_slab_last_error()                   */

/* element types (the `element =` of !neptune_ir.field / !neptune_ir.temp,
 * include/Dialect/NeptuneIR/NeptuneIRTypes.td:22-33) */
#define NEPTUNE_HIP_F64 0
#define NEPTUNE_HIP_F32 1

/* kernel selection for neptune_hip_launch_cfg_t.kernel */
#define NEPTUNE_HIP_KERNEL_AUTO 0
#define NEPTUNE_HIP_KERNEL_DIRECT 1 /* one thread per cell, neighbours through L1/L2   */
#define NEPTUNE_HIP_KERNEL_MARCH 2  /* wave tiles marching along dim 0, planes in VGPRs */
/* neptune_hip_launch_cfg_t.flags */
#define NEPTUNE_HIP_FLAG_DIRECT_FLAT 1 /* direct kernel: flat one-lane-per-cell form instead of the rows form */

/* built-in stencil bodies; each one is the body of a committed fixture
 * (tests/mlir_tests/conversion_tests/) evaluated in that file's textual op order */
#define NEPTUNE_HIP_BODY_LAP2D5_F64 0  /* apply-2d-5pt.mlir   */
#define NEPTUNE_HIP_BODY_LAP3D7_F64 1  /* apply-3d-7pt.mlir   */
#define NEPTUNE_HIP_BODY_LAP3D27_F32 2 /* apply-3d-27pt.mlir  */
#define NEPTUNE_HIP_BODY_LAP1D3_F64 3  /* @ac_lap of the reference's smoke_time_advance.mlir:13-29 */
#define NEPTUNE_HIP_BODY_COUNT 4

/* ------------------------------------------------------------------------------------
 * 3. geometry of one `neptune_ir.apply`
 *    (include/Dialect/NeptuneIR/NeptuneIROps.td:164-197; semantics
 *     lib/Passes/DataflowLowering.cpp:258-448)
 *
 *    All boxes are half-open logical boxes [lb, ub) as in #neptune_ir.bounds
 *    (NeptuneIRAttrs.td:9-26).  A temp with box [lb,ub) is a dense row-major buffer of
 *    shape ub-lb (DataflowLowering.cpp:41-49); logical point p lives at physical index
 *    p - lb.  The result has box [out_lb,out_ub); input k has box [in_lb[k],in_ub[k]).
 *    shape(input 0) must equal shape(result) (cast at DataflowLowering.cpp:285-286).
 *
 *    result[q]      = input0[q]                       for every physical q   (copy-through, :283-287)
 *    result<p>      = body(p; access(k,off) = input_k<p+off>)  for p in [lb,ub)  (:289-444)
 *
 *    region: physical sub-box (result coordinates) this launch is responsible for; cells
 *    outside are not touched.  Whole field: region_lb = 0, region_ub = shape.  Used by the
 *    slab decomposition to split one apply into edge planes + interior.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  int32_t rank;       /* 1..NEPTUNE_HIP_MAX_RANK */
  int32_t num_inputs; /* 1..NEPTUNE_HIP_MAX_INPUTS */
  int64_t out_lb[NEPTUNE_HIP_MAX_RANK], out_ub[NEPTUNE_HIP_MAX_RANK];
  int64_t lb[NEPTUNE_HIP_MAX_RANK], ub[NEPTUNE_HIP_MAX_RANK]; /* apply.bounds */
  int64_t in_lb[NEPTUNE_HIP_MAX_INPUTS][NEPTUNE_HIP_MAX_RANK];
  int64_t in_ub[NEPTUNE_HIP_MAX_INPUTS][NEPTUNE_HIP_MAX_RANK];
  int64_t region_lb[NEPTUNE_HIP_MAX_RANK], region_ub[NEPTUNE_HIP_MAX_RANK];
} neptune_hip_apply_geom_t;

/* launch tuning; pass NULL (or variant = -1, kernel = chunk = 0) for the defaults */
typedef struct {
  int32_t kernel;   /* NEPTUNE_HIP_KERNEL_* */
  int32_t variant;  /* march tile variant; negative = automatic (default tile for the stencil shape);
                       see neptune_hip_march_variant_name */
  int32_t chunk;    /* march: planes per workgroup along dim 0, 0 = auto */
  int32_t flags;    /* NEPTUNE_HIP_FLAG_* bits, 0 = defaults */
} neptune_hip_launch_cfg_t;

/* ------------------------------------------------------------------------------------
 * 4. runtime: device, memory, streams
 *    The reference allocates results with malloc (memref.alloc, DataflowLowering.cpp:281)
 *    and lets the caller free() them (NeptunePETScRuntime.cpp:219-221).  Device-resident
 *    buffers follow the same callee-allocates / caller-frees rule through
 *    neptune_hip_malloc / neptune_rt_free.
 * ---------------------------------------------------------------------------------- */
/* Select the HIP device for this process (one process per GPU).  Idempotent. */
void neptune_hip_init(int device);
/* Release cached workspaces.  Safe to call more than once. */
void neptune_hip_finalize(void);
/* 1 when a HIP device is usable, 0 otherwise (never aborts). */
int neptune_hip_available(void);
/* e.g. "gfx950:sramecc+:xnack-"; pointer valid for the process lifetime */
const char *neptune_hip_arch(void);
int neptune_hip_cu_count(void);
const char *neptune_hip_version(void);

void *neptune_hip_malloc(size_t bytes);
void neptune_hip_free(void *dptr);
void neptune_hip_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream);
void neptune_hip_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream);
void neptune_hip_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream);
void neptune_hip_stream_sync(void *stream);
void neptune_hip_device_sync(void);
/* 1 if p is device (hipMalloc) memory, 0 if host/unknown */
int neptune_hip_is_device_ptr(const void *p);

/* ---- block pool for the temporaries of lowered functions ---------------------------------------------
 * The reference mallocs every apply result (DataflowLowering.cpp:281 -> malloc).  On the device a
 * field-sized hipMalloc/hipFree pair costs orders of magnitude more than the kernel using the block, so
 * idle blocks are cached (at most NEPTUNE_HIP_POOL_BYTES, default a quarter of the device memory) and
 * reused.  _release hands back a block that no stream still uses; a block that left a lowered function
 * as its result is simply hipFree'd by neptune_rt_free, the pool does not track live blocks. */
void *neptune_hip_pool_alloc(size_t bytes);
void neptune_hip_pool_release(void *p, size_t bytes);
void neptune_hip_pool_trim(void);
size_t neptune_hip_pool_cached_bytes(void);

/* ---- slab view for lowered modules (one process per GPU) --------------------------
 * A lowered module is compiled once, for the GLOBAL field boxes its types declare.  While a slab is
 * set, every lowered function of this process reads its memref arguments as the caller's LOCAL
 * buffers: dim 0 of every declared box [lb0,ub0) becomes [max(lb0,start-ghost_lo), min(ub0,stop+ghost_hi)),
 * apply / store / reduce bounds are clipped to the owned planes [start,stop) (logical coordinates),
 * and a reduce returns this rank's partial sum.  The caller refreshes the ghost planes of the inputs
 * (neptune_hip.slab.exchange_halos) before the call; result ghost planes are unspecified.  An apply
 * that would read ghost planes of a value produced inside the same call aborts (it needs an exchange
 * the module cannot do).  The reference has no counterpart (every PETSc object lives on
 * PETSC_COMM_SELF, NeptunePETScRuntime.cpp:136,244,257). */
int neptune_hip_set_slab(int64_t start, int64_t stop, int64_t ghost_lo, int64_t ghost_hi);
int neptune_hip_clear_slab(void);
/* out = {start, stop, ghost_lo, ghost_hi}; returns 1 if a slab is set, else 0 */
int neptune_hip_get_slab(int64_t out[4]);
/* Exchange beside interior for whole lowered functions: `event` (a hipEvent_t) marks a halo exchange of the call's
 * inputs that is still in flight on another stream.  The next lowered function then launches the interior planes of
 * its first stencil apply, makes its stream wait for the event, and launches the planes next to the ghosts; anything
 * else that comes first (a store, a reduce, a pointwise apply over the ghost planes) waits for the event before it
 * runs.  The event is consumed by that call (or by neptune_hip_clear_slab).  Needs a slab to be set. */
int neptune_hip_set_slab_pending(void *event);
void *neptune_hip_get_slab_pending(void);

/* replaces the reference's neptune_rt_free (NeptunePETScRuntime.cpp:1825 -> free()):
 * frees either a malloc'ed host result or a device result of a lowered function */
void neptune_rt_free(void *p);

/* ------------------------------------------------------------------------------------
 * 5. the hot path: apply
 * ---------------------------------------------------------------------------------- */
/* Validate a geometry against a footprint (max |offset| per input and dim): every access
 * of every in-bounds point must stay inside its input's box.  Returns NEPTUNE_HIP_OK,
 * NEPTUNE_HIP_EINVAL or NEPTUNE_HIP_EOOB.  radius[k][d] = max |offset| of input k, dim d. */
int neptune_hip_check_geom(const neptune_hip_apply_geom_t *g,
                           const int32_t radius[NEPTUNE_HIP_MAX_INPUTS][NEPTUNE_HIP_MAX_RANK]);

/* Launch one built-in body over `g` on `stream` (a hipStream_t, NULL = default stream).
 * in[k], out are DEVICE pointers to dense row-major buffers of the shapes `g` implies.
 * out must not overlap any input.  Asynchronous.
 * Replaces: the scf.for nest of ApplyToSCFForLowering (DataflowLowering.cpp:289-444)
 * plus its copy-through memref.copy (:283-287), for the body of the named fixture. */
int neptune_hip_apply_builtin(int body, const neptune_hip_apply_geom_t *g,
                              const void *const *in, void *out, void *stream,
                              const neptune_hip_launch_cfg_t *cfg);

/* A geometry-level apply entry with its body bound: what every lowered apply exports as
 * <function>_<k>__geom (csrc/lowering/emit_hip.cpp), same arguments as neptune_hip_apply_builtin
 * minus the body id. */
typedef int (*neptune_hip_apply_fn)(const neptune_hip_apply_geom_t *g, const void *const *in, void *out,
                                    void *stream, const neptune_hip_launch_cfg_t *cfg);

/* `steps` applies in a row on two ping-pong fields: step s reads fields[s % 2] as input 0 and writes
 * fields[(s + 1) % 2]; inputs 1.. (in[1..], in[0] is ignored) stay the same every step.  After the call
 * the newest state is in fields[steps % 2].  Asynchronous on `stream`.
 * The pair of launches is captured ONCE into a hipGraph and replayed (8 pairs = 16 launches per graph) (graphs are cached by
 * geometry, pointers and configuration), so the per-step host cost is a fraction of a kernel launch: for
 * fields of a few MiB -- the reference's own 1-D/2-D smoke sizes, 1024^2 -- a step is otherwise bound
 * by launch overhead, not by the kernel.  body_or_fn: pass fn = NULL to use built-in body `body`.
 * The reference's counterpart is the host loop around `entry` in a driver (smoke_apply.sh:70-80) and the
 * forward-Euler loop of its runtime (NeptunePETScRuntime.cpp:677-712). */
int neptune_hip_step_loop(neptune_hip_apply_fn fn, int body, const neptune_hip_apply_geom_t *g,
                          void *const fields[2], const void *const *in, int64_t steps, void *stream,
                          const neptune_hip_launch_cfg_t *cfg);

/* Several steps per pass over HBM.  neptune_hip_apply_chain_builtin computes out = A(A(in)) (applies = 2) or A(A(A(in)))
 * (applies = 3) for built-in body A in ONE launch -- the intermediate fields exist in registers only; the same operations on
 * the same operands as separate launches, hence the same bits.  Scope: rank 3: star footprints of input 0 up to radius 2
 * per axis (the 7-point family: two or three applies per pass; 13-point 4th-order operators: two), the fused explicit
 * Euler step of such an operator included, and -- for lowered applies -- further inputs read at the centre only
 * (coefficient fields: in[1..], the same field at every stage); rank 2: the single-input 5-point family
 * (neptune_apply_march2_rank2); when the geometry qualifies (all boxes equal, rows a whole number of 64-byte granules,
 * launch region restricted along dim 0 only); otherwise NEPTUNE_HIP_EUNSUPPORTED and nothing is launched.  neptune_hip_apply2_builtin is the
 * applies = 2 form.  Lowered applies export the same as <function>_<k>__geom2 / __geom3.
 * neptune_hip_step_loop uses them for the built-in bodies ON ITS OWN for fields of NEPTUNE_HIP_CHAIN_MIN_CELLS cells
 * (default 4e6) and more; neptune_hip_step_loop_chain is the same loop with a lowered apply's pair / triple entries `fn2`,
 * `fn3` (NULL = none) next to its single-step entry `fn` (neptune_hip_step_loop_pairs: fn3 = NULL).  The newest state ends
 * in fields[steps % 2] whatever the grouping -- and that is the ONLY field defined after the loop: with chained launches the
 * intermediate states live in registers, so fields[(steps + 1) % 2] does NOT hold state steps - 1 (with one launch per step
 * it would).  A caller that needs the previous state as well sets NEPTUNE_HIP_NO_PAIRS=1 (one launch per step) or raises
 * NEPTUNE_HIP_CHAIN_MIN_CELLS.  The reference steps one apply per pass on the host (runtime forward Euler,
 * NeptunePETScRuntime.cpp:677-712). */
int neptune_hip_apply_chain_builtin(int body, int applies, const neptune_hip_apply_geom_t *g, const void *const *in,
                                    void *out, void *stream, const neptune_hip_launch_cfg_t *cfg);
int neptune_hip_apply2_builtin(int body, const neptune_hip_apply_geom_t *g, const void *const *in, void *out,
                               void *stream, const neptune_hip_launch_cfg_t *cfg);
int neptune_hip_step_loop_pairs(neptune_hip_apply_fn fn, neptune_hip_apply_fn fn2, int body,
                                const neptune_hip_apply_geom_t *g, void *const fields[2], const void *const *in,
                                int64_t steps, void *stream, const neptune_hip_launch_cfg_t *cfg);
int neptune_hip_step_loop_chain(neptune_hip_apply_fn fn, neptune_hip_apply_fn fn2, neptune_hip_apply_fn fn3, int body,
                                const neptune_hip_apply_geom_t *g, void *const fields[2], const void *const *in,
                                int64_t steps, void *stream, const neptune_hip_launch_cfg_t *cfg);

/* Which kernel neptune_hip_apply_builtin would run for (body, g, cfg):
 * NEPTUNE_HIP_KERNEL_DIRECT / _MARCH, or a negative error. */
int neptune_hip_apply_builtin_plan(int body, const neptune_hip_apply_geom_t *g,
                                   const void *const *in, const void *out,
                                   const neptune_hip_launch_cfg_t *cfg);

/* Name of the device kernel (as rocprofv3 lists it, without template arguments) that the
 * plan above launches; pointer valid for the process lifetime. */
const char *neptune_hip_kernel_name(int kernel);
/* Which march tile neptune_hip_apply_builtin would use for (body, g, cfg) if it plans the march kernel:
 * cfg->variant when valid, else the automatic choice (stencil shape, field size, launch region). */
int neptune_hip_apply_builtin_variant(int body, const neptune_hip_apply_geom_t *g,
                                      const neptune_hip_launch_cfg_t *cfg);
/* march tile variants compiled into the library, per field rank (2 or 3) */
int neptune_hip_march_variant_count(int rank);
const char *neptune_hip_march_variant_name(int rank, int variant);

/* ------------------------------------------------------------------------------------
 * 6. store (lib/Passes/DataflowLowering.cpp:165-220)
 *    no bounds : whole-buffer copy (memref.copy, :176-179)
 *    bounds    : logical box [lb,ub) copied between two buffers that each use their own
 *                logical origin (:184-217)
 * ---------------------------------------------------------------------------------- */
int neptune_hip_store_full(int dtype, const void *src, void *dst, int64_t count, void *stream);
int neptune_hip_store_box(int dtype, int rank, const void *src, const int64_t *src_lb,
                          const int64_t *src_ub, void *dst, const int64_t *dst_lb,
                          const int64_t *dst_ub, const int64_t *lb, const int64_t *ub,
                          void *stream);

/* ------------------------------------------------------------------------------------
 * 6b. reduce {kind = "sum"}  (lib/Passes/DataflowLowering.cpp:589-698; NeptuneIROps.td:272-299)
 *    Sum of the temp `src` (box [src_lb,src_ub)) over the logical domain [lb,ub) (NULL = the
 *    whole temp), accumulated in the element type and returned widened to double.  Blocking.
 *    The reference sums serially in row-major order; this is a fixed-tree parallel sum: bit-wise
 *    reproducible run to run, within 2(n-1) eps sum|x| of the serial result.
 * ---------------------------------------------------------------------------------- */
int neptune_hip_reduce_sum(int dtype, int rank, const void *src, const int64_t *src_lb,
                           const int64_t *src_ub, const int64_t *lb, const int64_t *ub,
                           double *result, void *stream);
/* device scratch of the reductions: (2048 + 1) elements of 8 bytes, owned by the library.  Used by the
 * fused apply+reduce kernels a lowered module carries (csrc/kernels/reduce_apply.hpp); calls that use it
 * are serialised by the stream they run on. */
void *neptune_hip_reduce_workspace(void);

/* ------------------------------------------------------------------------------------
 * 6c. vector updates for device-resident Krylov loops
 *    The reference's matrix-free solvers call the lowered operator from a host KSP loop over host Vecs
 *    (NeptunePETScRuntime.cpp:182-230, 719-786).  With device pointers in the memref arguments the operator and the
 *    dot products (reduce of an apply) already run without any host traffic; these two updates complete a solver
 *    loop that never leaves the GPU (a run-time scalar cannot enter a NeptuneIR apply region: IsolatedFromAbove).
 *    Element type of `a` follows dtype (rounded to float for F32).  Two roundings, no FMA.  Asynchronous.
 *      axpy: y[i] = y[i] + a * x[i]          xpay: y[i] = x[i] + a * y[i]
 * ---------------------------------------------------------------------------------- */
int neptune_hip_axpy(int dtype, int64_t n, double a, const void *x, void *y, void *stream);
int neptune_hip_xpay(int dtype, int64_t n, const void *x, double a, void *y, void *stream);

/* ------------------------------------------------------------------------------------
 * 7. helpers for tests and the bench (device-side, so 8 GiB fields never cross PCIe)
 * ---------------------------------------------------------------------------------- */
/* Deterministic field: value depends only on (global linear index + index_offset, seed);
 * integer hash mapped exactly to [-1,1) -- the same bits on host and device
 * (host twin: neptune_hip_hash_value). */
int neptune_hip_fill_hash(int dtype, void *dst, int64_t count, int64_t index_offset,
                          uint64_t seed, void *stream);
double neptune_hip_hash_value(int dtype, int64_t index, uint64_t seed);
/* Number of elements whose bit patterns differ between two device buffers (blocking). */
int64_t neptune_hip_count_mismatch(int dtype, const void *a, const void *b, int64_t count,
                                   void *stream);
/* Time `reps` launches of one built-in apply with HIP events on `stream`; returns the
 * average milliseconds per launch (blocking).  in/out as in neptune_hip_apply_builtin. */
double neptune_hip_time_apply_builtin(int body, const neptune_hip_apply_geom_t *g,
                                      const void *const *in, void *out, void *stream,
                                      const neptune_hip_launch_cfg_t *cfg, int warmup, int reps);
/* Plan-time tuning (in the spirit of FFTW_MEASURE): time every march tile of the library (and a
 * few chunk lengths) for exactly this geometry and these buffers, and return the fastest
 * configuration in *best (average ms per launch in *best_ms, may be NULL).  All candidates
 * produce identical bits; `out` ends up holding the result of a normal launch.  Blocking. */
int neptune_hip_autotune_builtin(int body, const neptune_hip_apply_geom_t *g, const void *const *in,
                                 void *out, void *stream, int reps, neptune_hip_launch_cfg_t *best,
                                 double *best_ms);
/* The same two calls for a lowered apply's geometry-level entry (`fn`, a module's <function>_<k>__geom): the
 * plan-time tuning a lowered module gets.  num_variants = how many march tiles that module holds
 * (its <function>_<k>__geom_variants(rank) export; the library's default tiles unless the module was built with
 * NEPTUNE_HIP_FULL_VARIANTS=1).  A tile that cannot take the geometry is rejected by the entry and skipped. */
double neptune_hip_time_apply_fn(neptune_hip_apply_fn fn, const neptune_hip_apply_geom_t *g,
                                 const void *const *in, void *out, void *stream,
                                 const neptune_hip_launch_cfg_t *cfg, int warmup, int reps);
int neptune_hip_autotune_fn(neptune_hip_apply_fn fn, int num_variants, const neptune_hip_apply_geom_t *g,
                            const void *const *in, void *out, void *stream, int reps,
                            neptune_hip_launch_cfg_t *best, double *best_ms);
/* ---- launch wisdom: measured launch choices, remembered across processes --------------------------------
 * An apply launched without an explicit configuration (cfg NULL or all-automatic) on a field of
 * NEPTUNE_HIP_TUNE_MIN_CELLS cells (default 2^24) or more measures ONCE which of its march tiles and chunk lengths is
 * fastest for exactly that (body, geometry): the first launch times the candidates (it synchronises the stream; every
 * candidate writes the same bits), later launches of the process reuse the choice, and the choice is appended to a
 * wisdom file so that every later PROCESS on the same device reuses it without timing anything (FFTW's wisdom, for
 * stencil launches).  File: $NEPTUNE_HIP_WISDOM, else <$NEPTUNE_CACHE_DIR or ~/.neptune/cache>/wisdom_v1.txt -- next
 * to the module cache; NEPTUNE_HIP_WISDOM= (empty) keeps choices in the process only.  NEPTUNE_HIP_TUNE=0 switches the
 * measuring off (fixed automatic tiles), NEPTUNE_HIP_TUNE=1 measures fields of any size.  A launch inside a stream
 * capture never measures.  Keys name the kernel build, the module and body, the element type and everything of the
 * geometry the launcher looks at; the library adds the device (name, architecture, CU count).
 * lookup: 1 = found (*cfg filled), 0 = unknown.  store: appends one line; 0 on success.
 * The reference has nothing to tune (one scalar loop nest, DataflowLowering.cpp:289-310). */
int neptune_hip_wisdom_lookup(const char *key, neptune_hip_launch_cfg_t *cfg);
int neptune_hip_wisdom_store(const char *key, const neptune_hip_launch_cfg_t *cfg, double ms);
/* the wisdom file of this process ("" when disabled); pointer valid for the process lifetime */
const char *neptune_hip_wisdom_path(void);
/* counters of this process: out[0] = first-use measurements made, out[1] = choices taken from the wisdom file,
 * out[2] = choices appended to it */
void neptune_hip_tune_stats(int64_t out[3]);
/* What the most recent apply launch of the calling thread really ran: kernel, march tile, planes per workgroup
 * (chunk as launched, never 0 for the march kernel).  Launchers call _note_launch; returns 0 if nothing was launched yet. */
void neptune_hip_note_launch(int kernel, int variant, int chunk);
int neptune_hip_last_launch(neptune_hip_launch_cfg_t *out);

/* Plain 16-byte-per-lane device copy, timed the same way: the measured HBM ceiling.
 * mode selects the copy kernel shape (0 .. neptune_hip_copy_mode_count()-1: grid-stride, or
 * 1/2/4/8 loads in flight per lane with optional non-temporal loads/stores). */
double neptune_hip_time_copy(void *dst, const void *src, size_t bytes, void *stream, int mode,
                             int warmup, int reps);
int neptune_hip_copy_mode_count(void);

/* HIP events for callers that time a stream themselves (bench.py). */
void *neptune_hip_event_create(void);
void neptune_hip_event_destroy(void *ev);
void neptune_hip_event_record(void *ev, void *stream);
void neptune_hip_event_sync(void *ev);
double neptune_hip_event_elapsed_ms(void *start, void *stop);
/* make `stream` wait for `ev` (stream-to-stream ordering without blocking the host) */
void neptune_hip_stream_wait_event(void *stream, void *ev);

/* ------------------------------------------------------------------------------------
 * 8. slab decomposition across the GPUs of a node: halo exchange overlapped with the interior
 *    (the reference has no counterpart -- every PETSc object lives on PETSC_COMM_SELF,
 *    NeptunePETScRuntime.cpp:136,244,257).  One process per GPU.  Rank g owns planes [start_g, stop_g) of dim 0
 *    and keeps `radius` ghost planes per existing neighbour in the SAME dense buffer:
 *        local buffer = [ r_lo ghost planes | n_own owned planes | r_hi ghost planes ] x plane
 *    so a halo is one contiguous run of memory, sent and received in place.  No periodic wrap: at the global boundary
 *    r_lo / r_hi are 0 and the apply's copy-through semantics hold (DataflowLowering.cpp:283-287, 382-410).
 *    Two transports behind one communicator type:
 *      NEPTUNE_HIP_TRANSPORT_RCCL  ncclSend/ncclRecv, grouped, on a communication stream.  RCCL is loaded at the first
 *          call (dlopen librccl.so.1, reusing a copy already in the process); programs that never call these entry
 *          points never load it.
 *      NEPTUNE_HIP_TRANSPORT_PEER  the ranks of ONE node: every rank pushes its edge planes into its neighbour's ghost
 *          planes with hipMemcpyAsync through a mapping of the neighbour's buffer (hipIpcOpenMemHandle) -- SDMA engines
 *          over xGMI between two devices, no CU moves data -- and two one-wave kernels per exchange carry the
 *          "ghost planes free" / "planes landed" handshake through counters in a shared-memory segment
 *          (csrc/runtime/slab_peer.hpp).  Two processes may share one device on this transport.  Buffers exchanged
 *          through it must stay allocated while the communicator lives (their mappings are cached).
 *    Failures return NEPTUNE_HIP_ECOMM / NULL with the text in neptune_hip_slab_last_error() -- they never fall back to
 *    another transport.  Either transport issues work on a stream of its own: synchronise the compute stream (or wait
 *    for the plan's completion) before issuing collectives of ANOTHER communicator library on the same device
 *    (torch.distributed's, say) -- two communicators progressing concurrently can deadlock.
 *    Multi-rank runs of the RCCL transport between two devices are unverified on the authors' hardware (one-GPU boxes:
 *    loop-back and two-process tests only); a start-up exchange check like bench.py's is recommended.
 * ---------------------------------------------------------------------------------- */
#define NEPTUNE_HIP_SLAB_ID_BYTES 128
#define NEPTUNE_HIP_TRANSPORT_RCCL 0
#define NEPTUNE_HIP_TRANSPORT_PEER 1
typedef struct neptune_hip_slab_comm neptune_hip_slab_comm_t;
typedef struct neptune_hip_slab_plan neptune_hip_slab_plan_t;

/* rank 0 makes the id (RCCL: ncclGetUniqueId; peer: 128 random bytes naming the node's shared segment) and hands the
 * 128 bytes to every rank by any means.  The plain names are the RCCL transport. */
int neptune_hip_slab_unique_id(void *id_out);
int neptune_hip_slab_unique_id_ex(int transport, void *id_out);
/* collective over the `world` ranks (RCCL: ncclCommInitRank on the current device; peer: every rank maps the segment
 * and waits for the others, bounded by NEPTUNE_HIP_PEER_TIMEOUT_S, default 20 s); id may be NULL when world == 1 */
neptune_hip_slab_comm_t *neptune_hip_slab_comm_create(const void *id, int rank, int world);
neptune_hip_slab_comm_t *neptune_hip_slab_comm_create_ex(int transport, const void *id, int rank, int world);
void neptune_hip_slab_comm_destroy(neptune_hip_slab_comm_t *comm);
const char *neptune_hip_slab_last_error(void);
/* "rccl" / "peer" */
const char *neptune_hip_slab_comm_transport(const neptune_hip_slab_comm_t *comm);
/* NEPTUNE_HIP_OK, or NEPTUNE_HIP_ECOMM once a device-side wait of the peer transport has timed out (a neighbour died
 * or fell more than the timeout behind): every exchange since then left stale ghost planes.  Never blocks. */
int neptune_hip_slab_comm_status(neptune_hip_slab_comm_t *comm);

/* Refresh the ghost planes of one local buffer on `stream`: the first r_lo owned planes go to peer_lo and its last
 * planes arrive in my lower ghosts; likewise r_hi / peer_hi above (a side with 0 ghost planes is skipped, its peer is
 * ignored).  Asynchronous.  peer == own rank is allowed (a loop-back used by the single-GPU tests: RCCL matches the
 * two send/receive pairs in order, so the lower ghosts receive the rank's FIRST owned planes; the peer transport
 * pushes to the neighbour's opposite side, so they receive its LAST owned planes -- a periodic wrap).
 * _many: the ghost planes of `nfields` buffers (<= NEPTUNE_HIP_MAX_INPUTS) in one grouped exchange / one handshake. */
int neptune_hip_halo_exchange(neptune_hip_slab_comm_t *comm, void *field, size_t plane_bytes, int64_t n_own,
                              int r_lo, int r_hi, int peer_lo, int peer_hi, void *stream);
int neptune_hip_halo_exchange_many(neptune_hip_slab_comm_t *comm, void *const *fields, const size_t *plane_bytes,
                                   int nfields, int64_t n_own, int r_lo, int r_hi, int peer_lo, int peer_hi,
                                   void *stream);

/* One apply over this rank's slab, planned once.  `fn`: a lowered apply's geometry-level entry, or NULL to use
 * built-in body `body`.  `local`: the geometry of the LOCAL buffers (boxes include the ghost planes; apply.bounds
 * already clipped to the owned planes; the region is ignored).  radius = the apply's reach along dim 0.
 * neptune_hip_slab_apply then runs, per call:
 *     comm stream   : wait for the input on `compute_stream`, exchange the ghost planes of EVERY input
 *                     (a stream of the greatest priority the device offers: the exchange is dispatched ahead of the
 *                     interior grid, which fills every CU)
 *     compute stream: interior planes (those that need no ghost data)                 -- overlaps the exchange
 *     compute stream: wait for the exchange, then the `radius` edge planes per side that has a neighbour
 * overlap = 0 makes the interior wait for the exchange as well (debugging).  Asynchronous on compute_stream. */
neptune_hip_slab_plan_t *neptune_hip_slab_plan_create(neptune_hip_slab_comm_t *comm, neptune_hip_apply_fn fn, int body,
                                                      int dtype, const neptune_hip_apply_geom_t *local, int radius,
                                                      int r_lo, int r_hi, int peer_lo, int peer_hi,
                                                      const neptune_hip_launch_cfg_t *cfg);
int neptune_hip_slab_apply(neptune_hip_slab_plan_t *plan, const void *const *in, void *out, void *compute_stream,
                           int overlap);
void neptune_hip_slab_plan_destroy(neptune_hip_slab_plan_t *plan);
/* Where a sharded step spends its time.  _timing(plan, 1) makes every following neptune_hip_slab_apply record five
 * timed events (a ring of 64 steps; no synchronisation is added); _timing_read waits for the recorded steps and
 * returns their averages in milliseconds:
 *   out[0] exchange   (communication stream: input ready -> ghost planes landed)
 *   out[1] interior   (compute stream: the interior launch)
 *   out[2] edge wait  (how long after the interior's end the exchange ended; 0 when it was hidden behind it)
 *   out[3] edges      (the edge launches)
 *   out[4] step       (interior start -> edges done)
 *   out[5] steps averaged */
int neptune_hip_slab_plan_timing(neptune_hip_slab_plan_t *plan, int on);
int neptune_hip_slab_plan_timing_read(neptune_hip_slab_plan_t *plan, double out[6]);

#ifdef __cplusplus
}
#endif
#endif /* NEPTUNE_HIP_H */
