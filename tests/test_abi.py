"""CPU: the C-ABI library loads without a GPU and exports exactly the symbols include/pv_yield_hip.h declares."""
import ctypes
import os
import re
import subprocess

import pytest

from predict_pv_yield_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pv_yield_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(pv_[a-z0-9_]+)\s*\(", text))


def test_library_builds_and_loads():
    _lib.build_library()
    assert os.path.exists(_lib.LIB_PATH)
    lib = _lib.get_lib()
    assert lib.pv_abi_version() == 1


def test_every_declared_symbol_is_exported_and_bound():
    decl = declared_symbols()
    assert len(decl) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert decl == set(_lib.SIGNATURES), (decl ^ set(_lib.SIGNATURES))
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (pv_[a-z0-9_]+)", out))
    assert exported == decl, exported ^ decl


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "pv_yield_hip.h"\nint main(void){pv_farneback_params p; pv_conv3d_dims d; (void)p; (void)d; return PV_OK;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o",
                           str(tmp_path / "t.o")])


def test_argument_errors_without_gpu():
    lib = _lib.get_lib()
    # argument validation happens before any launch, so it can be exercised on a GPU-less box
    assert lib.pv_bf16_cpad(11) == 16 and lib.pv_bf16_cpad(32) == 32 and lib.pv_bf16_cpad(33) < 0
    assert lib.pv_conv3d_packed_weight_elems(32) == 2 * 27 * 2 * 64 * 8   # v1 + v2 fragment orders
    rc = lib.pv_remap_bilinear_f32(None, 0, None, 0, None, 0, 0, 1, 1, 1.0, 4, 4, 0, 0.0, None)
    assert rc == -1 and b"null pointer" in lib.pv_last_error()
    p = _lib.FarnebackParams(0.5, 2, 40, 3, 5, 0.7, 0)
    need = ctypes.c_size_t(0)
    assert lib.pv_farneback_workspace_bytes(4, 64, 64, ctypes.byref(p), ctypes.byref(need)) == -1
    p.flags = 256
    assert lib.pv_farneback_workspace_bytes(4, 64, 64, ctypes.byref(p), ctypes.byref(need)) == 0 and need.value > 0


def test_shape_predicates_of_the_f32_path_without_gpu():
    """Host-side shape rules of the round-5 f32 kernels (pure C, no launch): which Conv3d geometries the half-float form takes
    (pv_conv3d_fwd_f16_f32out_covers: 64-byte voxels, padding 0..2, at least two output slices per time chunk, samples within the
    buffer descriptors) and which Linears the streamed fc1 kernels take (pv_linear_f32_skinny_covers); argument errors are loud."""
    lib = _lib.get_lib()

    def covers(b, ci, co, t, h, w, pad):
        d = _lib.Conv3dDims(b, ci, co, t, h, w, pad[0], pad[1], pad[2])
        return bool(lib.pv_conv3d_fwd_f16_f32out_covers(ctypes.byref(d)))

    assert covers(32, 32, 32, 16, 62, 62, (0, 0, 0))          # the model's second layer, forward
    assert covers(32, 32, 32, 14, 60, 60, (2, 2, 2))          # ... and its data gradient (pad = 2 - pad)
    assert covers(32, 32, 32, 18, 64, 64, (0, 0, 0))          # the first layer (11 channels in a 32-channel operand image)
    assert covers(6, 32, 32, 6, 40, 52, (1, 1, 1))
    assert not covers(32, 16, 32, 16, 62, 62, (0, 0, 0))      # 32-byte voxels: the caller pads the image to 32 channels
    assert not covers(32, 32, 32, 3, 62, 62, (0, 0, 0))       # one output slice: no time march
    assert not covers(32, 32, 32, 16, 62, 62, (3, 0, 0))      # padding beyond 2
    assert not covers(32, 32, 32, 2, 62, 62, (0, 0, 0))       # input shorter than the kernel
    assert not covers(70000, 32, 32, 16, 62, 62, (0, 0, 0))   # batch beyond grid.z
    assert not covers(1, 32, 32, 600, 512, 512, (0, 0, 0))    # one sample beyond 1 GiB of operand image
    assert lib.pv_conv3d_split2_weight_elems() == 4 * 27 * 2 * 64 * 8
    assert lib.pv_linear_f32_skinny_covers(32, 128, 1003520) == 1 and lib.pv_linear_f32_skinny_covers(1, 64, 65536) == 1
    assert lib.pv_linear_f32_skinny_covers(33, 128, 1003520) == 0 and lib.pv_linear_f32_skinny_covers(32, 129, 1003520) == 0
    assert lib.pv_linear_f32_skinny_covers(32, 128, 1003520 + 64) == 0 and lib.pv_linear_f32_skinny_covers(32, 128, 32768) == 0
    assert lib.pv_linear_fwd_f32_skinny_workspace_bytes(128) >= 512 * 32 * 128 * 4
    assert lib.pv_linear_fwd_f32_skinny(None, None, None, None, 32, 128, 1003520, 0, None, 0, None) == -1
    assert b"null pointer" in lib.pv_last_error()
    assert lib.pv_sum3_ndhwc_to_ncdhw_f32(None, None, None, None, 0, None, None, None, None, None, None, 0, 1, 64, None) == -1
    d = _lib.Conv3dDims(32, 16, 32, 16, 62, 62, 0, 0, 0)
    one = ctypes.c_float(0)
    assert lib.pv_conv3d_fwd_f16_f32out(ctypes.byref(one), ctypes.byref(one), ctypes.byref(one), 0, ctypes.byref(d), None) != 0
    assert b"channels" in lib.pv_last_error()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "predict_pv_yield_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{f} imports the oracle"
