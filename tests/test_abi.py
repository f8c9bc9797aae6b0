"""CPU-side checks of the C ABI: the library loads and exports every symbol include/gslim.h declares, struct
layouts match, and argument validation works without a GPU."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from graphslim_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "graphslim_amd", "csrc"), "-j4"], check=True)
    return _lib.lib()


def test_header_symbols_are_exported(L):
    from graphslim_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "gslim.h")).read()
    declared = set(re.findall(r"^(?:int|void|size_t|const char\*)\s+(gs_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s
    assert L.gs_version() >= 100
    # no process-global knob setter or debug hook is left in the product header: they live in gslim_internal.h
    assert not [s for s in declared if re.search(r"^gs_(gemm|sample)_set_|_debug$|_trace$", s)], declared


def test_internal_header_symbols_are_exported(L):
    """include/gslim_internal.h — the measurement knobs and debug hooks, NOT product ABI — declares exactly _lib.INTERNAL_SYMBOLS,
    compiles as plain C on top of gslim.h, and every one of them is exported."""
    from graphslim_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "gslim_internal.h")).read()
    declared = set(re.findall(r"^(?:int|void|size_t|const char\*)\s+(gs_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(_lib.INTERNAL_SYMBOLS), declared ^ set(_lib.INTERNAL_SYMBOLS)
    assert not (declared & set(_lib.SYMBOLS))
    for s in declared:
        assert hasattr(L, s), s
    # the enum's size is EVALUATED at compile time (`-fsyntax-only` never ran the old comparison) against the Python mirror
    subprocess.run(["gcc", "-x", "c", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-"],
                   input=('#include "gslim_internal.h"\n_Static_assert(GS_KNOB_COUNT == %d, "GS_KNOB_COUNT drifted from '
                          'graphslim_amd._lib.KNOBS");\n' % len(_lib.KNOBS)).encode() +
                   "".join('_Static_assert(GS_KNOB_%s == %d, "knob id");\n' % (k, v) for k, v in _lib.KNOBS.items()).encode() +
                   b"int main(void){return 0;}\n", check=True)


def test_struct_layouts_match_the_header(L):
    """sizeof as seen by gcc on include/gslim.h must equal the ctypes mirrors."""
    from graphslim_amd import _lib
    src = '#include "gslim.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu\\n",sizeof(gs_gemm_desc),' \
          'sizeof(gs_blocks),sizeof(gs_gcond_config),sizeof(gs_gcond_io));return 0;}'
    exe = "/tmp/gs_sizes"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    sizes = [int(v) for v in subprocess.run([exe], capture_output=True, check=True).stdout.split()]
    assert sizes == [C.sizeof(_lib.GemmDesc), C.sizeof(_lib.Blocks), C.sizeof(_lib.GCondConfig), C.sizeof(_lib.GCondIO)]


def test_integration_stub_structs_match_the_header(L):
    """The ctypes stub INTEGRATION.md shows a maintainer (section 2) is executed as written — against the built library — and its
    two structures must have the sizes gcc computes from include/gslim.h (the stub had lost `with_bn`)."""
    from graphslim_amd import _lib
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(# graphslim/condensation/_gslim\.py.*?)```", md, flags=re.S).group(1)
    code = code.replace('"libgslim_hip.so"', repr(_lib.LIB_PATH))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    src = '#include "gslim.h"\n#include <stdio.h>\nint main(){printf("%zu %zu\\n",sizeof(gs_gcond_config),sizeof(gs_gcond_io));return 0;}'
    exe = "/tmp/gs_sizes_stub"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    sizes = [int(v) for v in subprocess.run([exe], capture_output=True, check=True).stdout.split()]
    assert [C.sizeof(ns["gs_gcond_config"]), C.sizeof(ns["gs_gcond_io"])] == sizes
    assert [f for f, _ in ns["gs_gcond_config"]._fields_] == [f for f, _ in _lib.GCondConfig._fields_]
    assert [f for f, _ in ns["gs_gcond_io"]._fields_] == [f for f, _ in _lib.GCondIO._fields_]


def test_argument_validation_without_gpu(L):
    from graphslim_amd import _lib
    cfg = _lib.GCondConfig(100, 8, 8, 3, 9, 3, 16, 16, 256, 10, 5, 0, 1, 0.01, 0.01, 0.01, 1, 0)
    n = C.c_size_t()
    assert L.gs_gcond_workspace_bytes(C.byref(cfg), C.byref(n)) == -4          # ntrans = 3 unsupported
    assert b"ntrans" in L.gs_last_error()
    cfg.ntrans = 2
    assert L.gs_gcond_workspace_bytes(C.byref(cfg), C.byref(n)) == 0 and n.value > 0
    cfg.dis_metric = 7
    assert L.gs_gcond_workspace_bytes(C.byref(cfg), C.byref(n)) == -1
    assert b"unknown distance function" in L.gs_last_error()                   # condensation/utils.py:63
    assert L.gs_gemm_f32(None, None) == -1
    # arxiv-sized workspace fits one MI355X many times over
    big = _lib.GCondConfig(169343, 128, 128, 40, 454, 2, 256, 256, 256, 10, 5, 0, 3, 0.01, 0.01, 0.01, 1, 0)
    assert L.gs_gcond_workspace_bytes(C.byref(big), C.byref(n)) == 0
    assert 1 << 30 < n.value < 8 << 30


def test_missing_library_fails_loudly(monkeypatch):
    from graphslim_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgslim_hip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.lib()
