"""CPU suite: the run-time shader compiler's stage mode (csrc/user_shader.cpp) up to the point where a GPU is needed --
which programs are eligible, and that both the fixture program and the eligibility rule's hash behave.  No compute."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF_SHADER = "/root/reference/samples/shader.cl"
CLANG = "/opt/rocm/lib/llvm/bin/clang"


@pytest.fixture(scope="module")
def lib():
    L = ctypes.CDLL(os.path.join(ROOT, "radiance-ray-tracing_amd", "librdx.so"))
    L.rdx_last_error.restype = ctypes.c_char_p
    L.rdx_debug_stage_reduced_hash.restype = ctypes.c_ulonglong
    L.rdx_debug_stage_reduced_hash.argtypes = [ctypes.c_char_p, ctypes.c_uint32]
    L.rdx_debug_jit_compiles.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_int]
    return L


def fixture_program():
    text = open(os.path.join(GOLD, "user_stages.cl")).read()
    return text.replace('#include "user_material.inc"', open(os.path.join(GOLD, "user_material.inc")).read()).replace('#include "user_environment.inc"', open(os.path.join(GOLD, "user_environment.inc")).read()).encode()


@pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang is not installed")
def test_fixture_program_compiles_in_both_modes(lib):
    """tests/golden/user_stages.cl against the product's own device library: as the stage kernel (record / replay traceRay,
    the stage-mode get_global_id) and as a plain megakernel program"""
    t = fixture_program()
    for stages in (1, 0):
        assert lib.rdx_debug_jit_compiles(t, len(t), b"gfx950", stages) == 0, lib.rdx_last_error().decode()
    bad = t.replace(b"payload->hit = true;", b"payload->hit = this is not OpenCL C;")
    assert lib.rdx_debug_jit_compiles(bad, len(bad), b"gfx950", 1) != 0
    assert b"compilation failed" in lib.rdx_last_error()


@pytest.mark.skipif(not os.path.exists(REF_SHADER), reason="/root/reference is absent (GPU box)")
def test_stage_mode_eligibility_rule(lib):
    """A program is moved to the wavefront pipeline when it equals the stock program outside the bodies of its closest-hit /
    miss stage functions: editing `material` or `environment` keeps the reduced hash, editing raygen, a helper or the any-hit
    row changes it.  (Reads the reference's file as data, here only.)"""
    stock = open(REF_SHADER).read()
    want = lib.rdx_debug_stage_reduced_hash(None, 0)
    h = lambda s: lib.rdx_debug_stage_reduced_hash(s.encode(), len(s.encode()))
    assert h(stock) == want
    edited = stock.replace("color += albedo * 0.1f;", "color += albedo * 0.25f; /* { a brace in a comment */")
    assert edited != stock and h(edited) == want
    sky = re.sub(r"payload->color\.z = 0\.5f;", "payload->color.z = 0.9f;", stock)
    assert sky != stock and h(sky) == want
    assert h(stock.replace("0.001f, 1000, &payload", "0.002f, 1000, &payload")) != want          # the raygen loop
    assert h(stock.replace("*cont = false;", "*cont = true;")) != want                            # the any-hit row
    assert h(stock.replace("struct Payload\n{", "struct Payload\n{ int extra;")) != want


@pytest.mark.skipif(not (os.path.exists(REF_SHADER) and os.path.exists(CLANG)), reason="needs /root/reference and ROCm clang")
def test_stock_program_with_an_edited_material_compiles_as_stage_kernel(lib):
    stock = open(REF_SHADER).read().replace("color += albedo * 0.1f;", "color += albedo * 0.25f;").encode()
    assert lib.rdx_debug_jit_compiles(stock, len(stock), b"gfx950", 1) == 0, lib.rdx_last_error().decode()
