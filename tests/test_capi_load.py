"""CPU checks of the C-ABI boundary: the gfx950 library cross-compiles, loads, exports exactly the
symbols include/eds_hip.h declares, and refuses to work without a GPU (no CPU fallback).
No compute entry point is called here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "eds_hip.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(eds_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree(capi):
    assert _declared_functions() == sorted(capi.EXPORTS)


def test_library_exports_every_declared_symbol(capi):
    L = capi.lib()
    for name in _declared_functions():
        assert hasattr(L, name), f"libeds_hip.so does not export {name}"
    assert L.eds_abi_version() == 6


def test_library_contains_gfx950_code_object(capi):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          f"--input={capi.LIB_PATH}"], capture_output=True, text=True)
    blob = open(capi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob, "no gfx950 code object embedded"
    for k in (b"eds_resjac_kernel", b"eds_reduce_kernel", b"eds_fused6_kernel"):
        assert k in blob


def test_struct_layouts_match_header(capi):
    L = capi.lib()
    assert C.sizeof(capi.Cfg) == L.eds_trk_cfg_size() == 144
    assert C.sizeof(capi.Info) == L.eds_trk_info_size() == 64
    cfg = capi.default_config()
    assert cfg.sampling == capi.SAMPLE_BICUBIC and cfg.num_blocks == 1 and cfg.gradient_tolerance == 1e-8
    assert cfg.parameter_tolerance == 1e-6 and list(cfg.max_num_iterations) == [10] * 8


def test_no_cpu_fallback(capi):
    """Without a GPU the product path must fail loudly instead of computing on the host."""
    if capi.device_count() > 0:
        pytest.skip("a GPU is visible here; the refusal path is covered on CPU-only hosts")
    cfg = capi.default_config()
    with pytest.raises(capi.EdsError) as ei:
        capi.Handle(cfg, 1, 64, 48, 64)
    assert ei.value.code == capi.ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "slam-eds_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in txt and "np_oracle" not in txt and "eds_oracle" not in txt, f
