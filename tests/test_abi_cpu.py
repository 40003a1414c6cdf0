"""CPU checks of the C-ABI boundary: the library builds/loads, exports every symbol
include/hydra_mi.h declares, and fails loudly (never silently falls back) without a GPU."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(hm):
    from hydra_mi import _lib
    assert os.path.exists(_lib.SO_PATH), "run `python __graft_entry__.py build` first"
    assert _lib.missing_symbols() == []
    header = open(os.path.join(ROOT, "include", "hydra_mi.h")).read()
    declared = set(re.findall(r"\b(hm_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.lib().hm_version().startswith(b"hydra_mi")


def test_header_cites_the_reference_interfaces():
    header = open(os.path.join(ROOT, "include", "hydra_mi.h")).read()
    for cite in ("src/optical_flow_ext.cpp:310", "renderer.py:310-325", "cuda.py:972-980", "cuda.py:982-1010",
                 "kalman.py:437-449", "renderer.py:485-501"):
        assert cite in header, cite


def test_argument_errors_do_not_need_a_gpu(hm):
    import ctypes
    from hydra_mi import _lib
    L = _lib.lib()
    h = _lib.c_vp()
    assert L.hm_brox_create(0, 0, 64, 1, 0.197, 50.0, 0.8, 10, 77, 10, ctypes.byref(h)) == -1
    assert b"bad size" in L.hm_last_error()
    assert L.hm_brox_create(0, 64, 64, 1, 0.197, 50.0, 1.5, 10, 77, 10, ctypes.byref(h)) == -1
    assert b"scale_factor" in L.hm_last_error()
    tri = np.array([[0, 1, 7]], np.int32)
    uv = np.zeros((3, 2), np.float32)
    assert L.hm_ctx_create(0, 64, 64, 3, 1, _lib.ptr(tri), _lib.ptr(uv), 1.0, 1.0, 1.0, ctypes.byref(h)) == -1
    assert b"vertex 7" in L.hm_last_error()
    assert L.hm_brox_sync(None) == -1


def test_no_cpu_fallback(hm):
    """On a machine without a GPU the compute entry points must raise, not compute."""
    from hydra_mi import _lib, brox, renderer, mesh
    if _lib.lib().hm_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        brox.BroxOpticalFlow(64, 64)
    dm = mesh.square4_mesh(10, 30)
    with pytest.raises(RuntimeError):
        renderer.Renderer(dm, np.zeros((4, 2)), np.zeros((64, 64, 2), np.float32), 64, np.zeros((64, 64), np.uint8),
                          True, 1, 1, 1)
    with pytest.raises(RuntimeError):
        brox.op_blur(np.zeros((8, 8), np.float32))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "kalman-hydra_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+\.*oracle", text, re.M), f
                assert not re.search(r"#\s*include.*oracle", text), f
                assert "libbrox_oracle" not in text and "import_module(\"oracle" not in text, f
