"""The C-ABI library loads and exports exactly what include/monoport_hip.h declares (no compute,
no GPU), and the product refuses to run without a GPU instead of falling back."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "monoport_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mp_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from monoport_amd import _lib, build
    build.build()
    return _lib.load()


def test_header_and_binding_table_agree():
    from monoport_amd import _lib
    assert header_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol(lib):
    for name in header_symbols():
        assert hasattr(lib, name), name


def test_version(lib):
    assert lib.mp_version() >= 100


def test_no_cpu_fallback(lib):
    """Without a visible GPU mp_create fails with a message; with one it succeeds."""
    import torch
    handle = ctypes.c_void_p()
    rc = lib.mp_create(0, ctypes.byref(handle))
    if torch.cuda.is_available():
        assert rc == 0
        lib.mp_destroy(handle)
    else:
        assert rc != 0
        assert b"no HIP device" in lib.mp_last_error(None)
        from monoport_amd import ops
        from monoport_amd._lib import MonoportError
        with pytest.raises(MonoportError):
            ops.get_context("cpu")


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under monoport_amd/ may reference it."""
    pkg = os.path.join(ROOT, "monoport_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f
                assert "pifu_oracle" not in text or f.endswith((".hip", ".h")) or \
                    not re.search(r"import.*pifu_oracle", text), f


def test_header_is_plain_c_and_links_from_c(tmp_path, lib):
    """The boundary is a C ABI: include/monoport_hip.h compiles as C11 (no C++-isms, no torch
    types) and a C program links against the shared library and reads the version / an error
    string through it -- the binding any other host language would make."""
    import shutil
    import subprocess
    from monoport_amd import _lib
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "c_client.c"
    src.write_text(
        '#include <stdio.h>\n#include "monoport_hip.h"\n'
        "int main(void) {\n"
        "  mp_ctx *ctx = NULL;\n"
        "  int v = mp_version();\n"
        "  int rc = mp_query(NULL, 0, NULL, 0, 0, 0, NULL, 0, 0, 0, NULL, 0.0f, NULL, NULL);\n"
        '  printf("%d %d %s\\n", v, rc, rc == MP_ERR_ARG ? "arg" : "other");\n'
        "  mp_destroy(ctx);\n"
        "  return 0;\n}\n")
    exe = tmp_path / "c_client"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run([gcc, "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    str(src), "-o", str(exe), "-L", libdir, "-lmonoport_hip",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
                    "-Wl,--allow-shlib-undefined"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) == lib.mp_version() and out[2] == "arg"
