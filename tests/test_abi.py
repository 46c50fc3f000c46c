"""The C-ABI library loads and exports exactly what include/chirpgp_hip.h declares (no compute: no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'chirpgp_hip.h')


def _declared():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(cgp_[a-z_]+)\s*\(', src)))


def _lib():
    import __graft_entry__ as ge
    ge.build()
    from chirpgp_amd import _engine
    return _engine.load_library(), _engine


def test_header_symbols_are_exported():
    lib, eng = _lib()
    names = _declared()
    assert set(names) == set(eng.EXPORTS), (names, eng.EXPORTS)
    for n in names:
        assert hasattr(lib, n), f'{n} declared in chirpgp_hip.h but not exported'
    assert lib.cgp_version() == 160


def test_struct_layouts_match_header():
    _, eng = _lib()
    assert ctypes.sizeof(eng.CgpModel) == 48
    assert ctypes.sizeof(eng.CgpSigma) == 40
    assert ctypes.sizeof(eng.CgpInit) == 64
    assert ctypes.sizeof(eng.CgpSmoothOut) == 72


def test_null_context_is_rejected_without_gpu():
    lib, _ = _lib()
    assert lib.cgp_filter(None, 0, None, None, None, 0.0, None, 1, 1, None, 1, 1, None, None, None, 0, None) == -1
    assert lib.cgp_smoother(None, 0, None, None, 0.0, None, None, 1, 1, None, None, 0, None) == -1


def test_python_callables_are_refused():
    """No CPU fallback: a plain closure is a TypeError, never a silent host evaluation."""
    import numpy as np
    from chirpgp_amd import filters_smoothers as fs
    with pytest.raises(TypeError):
        fs.ekf(lambda u, dt: (u, np.eye(2)), np.ones(2), 0.1, np.zeros(2), np.eye(2), 0.1, np.zeros(5))
    with pytest.raises(TypeError):
        fs.cd_ekf(lambda u: u, lambda u: np.eye(2), np.ones(2), 0.1, np.zeros(2), np.eye(2), 0.1, np.zeros(5))


def test_product_does_not_import_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, 'chirpgp_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.h')):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in txt, f'{f} mentions oracle'


def test_a_stale_library_is_refused(monkeypatch):
    """the GPU box runs the shipped .so, nothing there calls build().  The library carries the sha256 of the
    sources it was built from (cgp_source_hash, csrc/Makefile) and the Python layer refuses one that does not match the tree."""
    lib, eng = _lib()
    assert lib.cgp_source_hash().decode() == eng.source_hash() and len(eng.source_hash()) == 64
    monkeypatch.setattr(eng, '_lib', None)
    monkeypatch.setattr(eng, 'source_hash', lambda: '0' * 64)
    with pytest.raises(RuntimeError, match='stale'):
        eng.load_library()
    monkeypatch.setattr(eng, 'source_hash', lambda: None)            # binary-only install: nothing to compare with
    assert eng.load_library() is not None


def test_the_header_is_plain_c_and_a_c_program_links_against_the_library(tmp_path):
    """The drop-in boundary is a C-ABI: include/chirpgp_hip.h compiles as pedantic C99 (and as C++11), its struct sizes are what the ctypes mirror
    assumes, and a C program that links against libchirpgp_hip.so sees the version and source hash -- no Python, no torch type in any signature."""
    import shutil
    import subprocess
    if not shutil.which('gcc'):
        pytest.skip('no gcc')
    _lib()
    src = tmp_path / 'abi.c'
    src.write_text('#include <stdio.h>\n#include <string.h>\n#include "chirpgp_hip.h"\n'
                   'int main(void) {\n'
                   '    printf("%d %zu %zu %zu %zu %zu\\n", cgp_version(), sizeof(cgp_model), sizeof(cgp_sigma), sizeof(cgp_init), sizeof(cgp_smooth_out), strlen(cgp_source_hash()));\n'
                   '    return cgp_version() == CGP_VERSION && cgp_filter(NULL, 0, NULL, NULL, NULL, 0.0, NULL, 1, 1, NULL, 1, 1, NULL, NULL, NULL, 0u, NULL) == CGP_E_ARG ? 0 : 1;\n}\n')
    inc, libdir = os.path.join(ROOT, 'include'), os.path.join(ROOT, 'chirpgp_amd')
    exe = tmp_path / 'abi'
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-Werror', '-pedantic', f'-I{inc}', str(src), '-o', str(exe), f'-L{libdir}', '-lchirpgp_hip',
                    f'-Wl,-rpath,{libdir}', '-Wl,-rpath,/opt/rocm/lib'], check=True, capture_output=True)
    subprocess.run(['g++', '-std=c++11', '-Wall', '-Wextra', '-Werror', f'-I{inc}', '-x', 'c++', '-fsyntax-only', str(src)], check=True, capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert [int(v) for v in out] == [160, 48, 40, 64, 72, 64], out
