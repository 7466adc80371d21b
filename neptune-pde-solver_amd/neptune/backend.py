"""Ahead-of-time compilation of the module the DSL has built, with an on-disk cache.

Keeps the reference's cache contract (python_frontend/neptune/backend.py:15-96): the key is the first 16 hex
digits of sha256(IR text), the artefact is `neptune_kernel_<key>.so` under $NEPTUNE_CACHE_DIR (default
~/.neptune/cache), and entries not touched for a week are swept when the compiler object is created.  What is
cached differs: the shared object comes from the HIP lowering (neptune_hip.lowering: emitter + hipcc for gfx950,
linked against libneptune_hip.so) instead of `clang++ ... -lneptune_runtime` (backend.py:55-70, a library the
reference tree does not contain)."""
import time
from pathlib import Path
from typing import Optional

from neptune_hip import lowering

_WEEK_SECONDS = 7 * 24 * 3600


def sweep_cache(directory: Path, max_idle_seconds: float = _WEEK_SECONDS) -> int:
    """delete cached kernels (and their sidecar files) not accessed for `max_idle_seconds`; returns how many"""
    removed = 0
    deadline = time.time() - max_idle_seconds
    for entry in directory.glob("neptune_kernel_*"):
        try:
            if entry.stat().st_atime < deadline:
                entry.unlink()
                removed += 1
        except OSError:
            continue            # raced with another process, or not ours to delete: the cache is best effort
    return removed


class AOTCompiler:
    """`compile_and_load(builder)` -> loaded module; same public surface as the reference class of this name"""

    def __init__(self, cache_directory: Optional[Path] = None):
        self.cache_dir = Path(cache_directory) if cache_directory else lowering.cache_dir()
        sweep_cache(self.cache_dir)

    def compile_and_load(self, builder):
        # a cache hit inside compile_module is a plain dlopen of the cached object
        return lowering.compile_module(builder.dump(), cache_directory=self.cache_dir)


_shared_compiler: Optional[AOTCompiler] = None


def jit_compile(builder):
    """compile the builder's current module; returns a neptune_hip.lowering.LoweredModule"""
    global _shared_compiler
    if _shared_compiler is None:
        _shared_compiler = AOTCompiler()
    return _shared_compiler.compile_and_load(builder)
