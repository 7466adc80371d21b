"""The C-ABI library loads on a box without a GPU and exports exactly what include/neptune_hip.h
declares (no compute calls here)."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import pytest

from neptune_hip import _capi


def _declared_functions(header_text: str):
    text = re.sub(r"/\*.*?\*/", "", header_text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    names = re.findall(r"\b(neptune_(?:hip|rt)_\w+)\s*\(", text)
    return sorted(set(n for n in names if not n.endswith("_t")))


def test_library_loads_and_exports_every_declared_symbol(built_libs):
    lib = _capi.load()
    declared = _declared_functions(_capi.HEADER_PATH.read_text())
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/neptune_hip.h but not exported"
        assert name in _capi.SIGNATURES, f"{name} has no ctypes signature in neptune_hip/_capi.py"
    extra = set(_capi.SIGNATURES) - set(declared)
    assert not extra, f"bound but not declared in the header: {extra}"


def test_struct_layouts_match_the_header(built_libs, tmp_path):
    src = tmp_path / "layout.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "neptune_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(neptune_hip_apply_geom_t), offsetof(neptune_hip_apply_geom_t, lb),
         offsetof(neptune_hip_apply_geom_t, in_lb), offsetof(neptune_hip_apply_geom_t, region_lb),
         sizeof(neptune_hip_launch_cfg_t), sizeof(NeptuneMemRef3D));
  printf("%zu %zu %zu\n", sizeof(NeptuneMemRef1D), sizeof(NeptuneMemRef2D), offsetof(NeptuneMemRef2D, strides));
  return 0;
}''')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", str(_capi.REPO_ROOT / "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    got = [int(x) for x in out]
    G = _capi.ApplyGeom
    want = [C.sizeof(G), G.lb.offset, G.in_lb.offset, G.region_lb.offset, C.sizeof(_capi.LaunchCfg),
            C.sizeof(_capi.NeptuneMemRef3D), C.sizeof(_capi.NeptuneMemRef1D), C.sizeof(_capi.NeptuneMemRef2D),
            _capi.NeptuneMemRef2D.strides.offset]
    assert got == want


def test_memref_descriptor_field_order_matches_the_reference_abi():
    # {allocated, aligned, offset, sizes[r], strides[r]} -- reference include/Runtime/PETSc/NeptunePETScRuntime.h:22-42
    for r, cls in _capi.MEMREF.items():
        assert [f[0] for f in cls._fields_] == ["allocated", "aligned", "offset", "sizes", "strides"]
        assert C.sizeof(cls) == 8 * (3 + 2 * r)


def test_host_only_entry_points_work_without_a_gpu(built_libs):
    lib = _capi.load()
    assert lib.neptune_hip_version().decode().startswith("neptune-hip")
    assert lib.neptune_hip_kernel_name(_capi.KERNEL_MARCH) == b"neptune_apply_march"
    assert lib.neptune_hip_kernel_name(_capi.KERNEL_DIRECT) == b"neptune_apply_direct"
    assert lib.neptune_hip_march_variant_count(3) >= 1
    assert lib.neptune_hip_march_variant_count(2) >= 1
    assert lib.neptune_hip_march_variant_count(1) == 1 and lib.neptune_hip_march_variant_count(4) == 0
    names = {lib.neptune_hip_march_variant_name(3, v) for v in range(lib.neptune_hip_march_variant_count(3))}
    assert len(names) == lib.neptune_hip_march_variant_count(3)
    assert lib.neptune_hip_is_device_ptr(None) == 0
    # neptune_rt_free on a malloc'ed host block takes the free() path (reference: neptune_rt_free -> free)
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    p = libc.malloc(64)
    lib.neptune_rt_free(p)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setenv("NEPTUNE_HIP_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_capi, "_lib", None)
    with pytest.raises(ImportError, match="no CPU fallback"):
        _capi.load()
