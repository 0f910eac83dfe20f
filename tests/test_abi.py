"""CPU tier: the C-ABI libraries load and export exactly what include/*.h declares; no compute calls."""
import ctypes as C
import os
import re
import subprocess

import pytest

from pathtracer_amd import _lib, host
from util import ROOT


def _declared(header: str):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt[xh]_[a-z0-9_]+)\s*\(", text)))


def test_ptx355_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared("ptx355.h")
    assert len(declared) >= 24
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(n for n, _, _ in _lib.SYMBOLS) == declared


def test_pth_host_exports_every_declared_symbol():
    lib = host.load()
    declared = _declared("pth_host.h")
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(n for n, _, _ in host.SYMBOLS) == declared


def test_headers_compile_as_c_with_reference_record_sizes(tmp_path):
    src = tmp_path / "abi.c"
    src.write_text('#include "pth_host.h"\n'
                   '_Static_assert(sizeof(ptx_camera) == 112, "camera");\n_Static_assert(sizeof(ptx_shape) == 112, "shape");\n'
                   '_Static_assert(sizeof(ptx_material) == 48, "material");\n_Static_assert(sizeof(ptx_bvh_node) == 32, "node");\n'
                   '_Static_assert(sizeof(ptx_texture_info) == 16, "texinfo");\nint main(void) { return 0; }\n')
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                           "-o", str(tmp_path / "abi.o")])


def test_kernel_code_object_is_gfx950_and_embedded():
    data = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in data and b"ptx_trace_kernel" in data


def test_null_and_no_device_errors_are_reported_not_raised():
    lib = _lib.load()
    assert lib.ptx_create(0, None) == -1
    h = C.c_void_p()
    rc = lib.ptx_create(99, C.byref(h))  # no such device anywhere
    assert rc != 0 and not h
    assert lib.ptx_trace(None, None) == -1 and lib.ptx_resize(None, 4, 4) == -1
    assert lib.ptx_last_error(None) == b"null context"
    lib.ptx_destroy(None)


def test_device_module_has_no_cpu_fallback():
    from pathtracer_amd import device
    src = open(device.__file__).read() + open(_lib.__file__).read()
    assert "oracle" not in src
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(device.PtxError):
            device.PathTracer(0)
