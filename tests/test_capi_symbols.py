"""The C-ABI library loads and exports every symbol include/tartangan_amd.h declares
(no compute calls: there is no GPU in the build container)."""
import ctypes
import os

import pytest

from tartangan_amd import backend


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(backend.LIBRARY):
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(backend.LIBRARY)


def test_header_parses():
    protos = backend.parse_header()
    assert len(protos) >= 45
    for name, (ret, params) in protos.items():
        assert ret in backend._CTYPES, (name, ret)
        for typ, _ in params:
            assert typ in backend._CTYPES, (name, typ)


def test_every_declared_symbol_is_exported(lib):
    missing = [n for n in backend.parse_header() if not hasattr(lib, n)]
    assert not missing, missing


def test_host_side_queries(lib):
    lib.tg_version.restype = ctypes.c_int
    lib.tg_arch.restype = ctypes.c_char_p
    assert lib.tg_version() >= 100
    assert lib.tg_arch() == b'gfx950'
    lib.tg_conv2d_wgrad_workspace.restype = ctypes.c_size_t
    lib.tg_conv2d_wgrad_workspace.argtypes = [ctypes.c_int] * 6
    assert lib.tg_conv2d_wgrad_workspace(64, 16, 16, 128, 128, 3) >= 16 * 16 * 9 * 4
    assert lib.tg_conv2d_wgrad_workspace(64, 16, 16, 128, 128, 5) == 0      # unsupported kernel size
    lib.tg_bn_workspace.restype = ctypes.c_size_t
    lib.tg_bn_workspace.argtypes = [ctypes.c_int] * 3
    assert lib.tg_bn_workspace(64, 16, 128 * 128) > 0


def test_bad_arguments_return_codes(lib):
    lib.tg_add.restype = ctypes.c_int
    lib.tg_add.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_void_p]
    assert lib.tg_add(None, None, None, 16, None) == -1          # TG_EINVAL, nothing launched
    lib.tg_conv2d_fwd.restype = ctypes.c_int
    lib.tg_conv2d_fwd.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
    assert lib.tg_conv2d_fwd(None, None, None, None, None, 1, 1, 1, 4, 4, 3, None) == -1
