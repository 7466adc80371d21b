"""AOT compile + load (mirror of python_frontend/neptune/backend.py:15-96): same cache contract --
sha256 of the IR text, first 16 hex digits, `neptune_kernel_<hash>.so` under $NEPTUNE_CACHE_DIR or
~/.neptune/cache, week-old entries swept -- but the shared object comes from the HIP lowering
(neptune_hip.lowering) and links against libneptune_hip.so instead of the non-existent
`-lneptune_runtime` the reference asks clang++ for (backend.py:55-70)."""
import time

from neptune_hip import lowering


class AOTCompiler:
    def __init__(self):
        self.cache_dir = lowering.cache_dir()
        self._cleanup_old_cache()

    def compile_and_load(self, compiler_instance):
        ir_str = compiler_instance.dump()
        mod = lowering.compile_module(ir_str)      # cache hit -> plain dlopen
        return mod

    def _cleanup_old_cache(self):
        try:
            now, cutoff = time.time(), 7 * 24 * 3600
            for p in self.cache_dir.glob("neptune_kernel_*"):
                if now - p.stat().st_atime > cutoff:
                    p.unlink()
        except Exception:
            pass


_compiler = None


def jit_compile(compiler_instance):
    """-> neptune_hip.lowering.LoweredModule (attribute access falls through to the ctypes library)"""
    global _compiler
    if _compiler is None:
        _compiler = AOTCompiler()
    return _compiler.compile_and_load(compiler_instance)
