/*
 * neptune_lowering.h -- C ABI of libneptune_lowering.so: the NeptuneIR -> HIP lowering as a
 * library (the `neptune-opt --neptuneir-to-hip` tool is a thin main() over the same calls).
 *
 * Takes the place, for the stencil hot path, of the reference's pass pipeline entry points
 * buildNeptuneToLLVMPipeline / NeptuneCompiler::runPipeline + compileToObjectFile
 * (lib/Pipeline/NeptuneIRPassesPipeline.cpp:5-47, lib/Compiler/NeptuneCompiler.cpp:291-358).
 * All strings returned through char** are malloc'ed; release them with neptune_lowering_free.
 * Return value: 0 on success, negative on failure (then *diag_out holds "line N: message").
 */
#ifndef NEPTUNE_LOWERING_H
#define NEPTUNE_LOWERING_H
#ifdef __cplusplus
extern "C" {
#endif

/* Parse + verify only (ApplyOp::verify, checkApplyLike, linear_opdef body rules;
 * reference lib/Dialect/NeptuneIR/NeptuneIRVerifier.cpp:34-171, lib/Passes/VerifyAndAnnotate.cpp:87-214). */
int neptune_lowering_verify(const char *mlir_text, char **diag_out);

/* Lower a module to one HIP translation unit.  *report_out is a JSON object:
 *   lowered     exported symbols
 *   signatures  per symbol: argument/result kinds, element types, ranks, shapes, origins
 *   skipped     functions left alone (solver ops and implicit time stepping: they stay on the host path)
 *   applies     per neptune_ir.apply (and per explicit time_advance): rank, inputs, element type, kernel family
 *               (march | direct | reduce = evaluated inside the consuming reduce's kernel), stencil shape,
 *               halo0 (reach along dim 0: the ghost planes a slab decomposition must hold) and geom_symbol,
 *               the exported geometry-level entry of that apply (neptune_hip_apply_fn, include/neptune_hip.h)
 * Set NEPTUNE_HIP_FULL_VARIANTS=1 in the environment of neptune_lowering_compile to build every march tile
 * into the module instead of the defaults (for NEPTUNE_HIP_TUNE=1). */
int neptune_lowering_to_hip(const char *mlir_text, char **source_out, char **report_out, char **diag_out);

/* Lower and compile to a shared object with hipcc (--offload-arch=gfx950 -ffp-contract=off),
 * linked against <repo_root>/neptune-pde-solver_amd/lib/libneptune_hip.so.  The emitted source is
 * kept next to the library as <so_path>.hip.  hipcc may be NULL (/opt/rocm/bin/hipcc). */
int neptune_lowering_compile(const char *mlir_text, const char *so_path, const char *repo_root,
                             const char *hipcc, char **report_out, char **diag_out);

void neptune_lowering_free(char *p);
const char *neptune_lowering_version(void);

#ifdef __cplusplus
}
#endif
#endif
