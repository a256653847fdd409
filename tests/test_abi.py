"""
The C-ABI library loads and exports every symbol include/tracer_amd.h declares; the Python constants agree with
the header's enums; without a GPU the context creation fails loudly (no CPU path).  CPU only, no compute calls.
"""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'tracer_amd.h')


@pytest.fixture(scope='module')
def lib_path():
    p = os.path.join(ROOT, 'tracer_amd', 'lib', 'libtracer_amd.so')
    if not os.path.exists(p):
        subprocess.check_call(['make', '-C', ROOT, 'all'])
    return p


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(trc_[a-z0-9_]+)\s*\(', src)))


def test_every_declared_symbol_is_exported(lib_path):
    lib = ctypes.CDLL(lib_path)
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), n


def test_binding_covers_the_header(lib_path):
    from tracer_amd import _cabi
    assert sorted(_cabi.SIGNATURES) == declared_functions()
    lib = _cabi.load_library()
    assert lib.trc_abi_version() == 2


def test_enum_values_match_header():
    from tracer_amd import _cabi
    from oracle import kinds
    src = open(HEADER).read()
    for m in re.finditer(r'\b(TRC_(GM|OPT|SRC)_[A-Z_]+)\s*=\s*(\d+)', src):
        name, val = m.group(1), int(m.group(3))
        if name.endswith('_COUNT'):
            continue
        py = name[4:]
        assert getattr(_cabi, py) == val, name
        assert getattr(kinds, py) == val, name
    assert ctypes.sizeof(_cabi.SurfaceDesc) == 24 + 8 * 36
    assert ctypes.sizeof(_cabi.SourceDesc) == 8 + 8 * (3 + 9 + 9 + 8 + 1 + 639)
    assert ctypes.sizeof(_cabi.TraceStats) == 56


def test_struct_sizes_match_the_compiler():
    """the ctypes mirrors of the C-ABI structs have the sizes the header gives them under a C++ compiler (trc_rays grew this round:
    complex indices, material rows, spectra)"""
    from tracer_amd import _cabi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(['make', '-s', '-C', root, 'hostcheck'])
    hc = ctypes.CDLL(os.path.join(root, 'tests', 'hostcheck', 'libtrc_hostcheck.so'))
    hc.hc_sizeof.restype = ctypes.c_long
    for which, cls in enumerate((_cabi.SurfaceDesc, _cabi.Rays, _cabi.SourceDesc, _cabi.KdTreeDesc, _cabi.TraceStats)):
        assert hc.hc_sizeof(which) == ctypes.sizeof(cls), cls.__name__
    assert ctypes.sizeof(_cabi.Rays) == 16 + 11 * 8 + 3 * 8 + 2 * 8


def test_no_gpu_means_loud_failure(lib_path):
    """the product has no CPU path: on a machine without a GPU creating a context raises"""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from tracer_amd import _cabi
    with pytest.raises(_cabi.TracerAmdError) as e:
        _cabi.Context(0)
    assert 'no CPU path' in str(e.value) or 'HIP' in str(e.value) or 'hip' in str(e.value)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'tracer_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
