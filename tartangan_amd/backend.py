"""ctypes binding of libtartangan_amd.so (the C ABI in include/tartangan_amd.h).

The header is the single source of truth: it is parsed at import time to build
the ctypes prototypes, so Python, C and the symbol-export test cannot drift.

There is NO CPU fallback.  ``get()`` raises if the shared library is missing
(build it with ``python __graft_entry__.py build`` / ``make -C
tartangan_amd/csrc``), and every call checks that its tensors live on a ROCm
device.  Unit tests of the host logic may install a stand-in with
``_set_backend_for_testing`` (tests/ only; see tests/emulator.py).
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'tartangan_amd.h')
LIBRARY = os.environ.get('TG_LIBRARY') or os.path.join(_HERE, 'csrc', 'libtartangan_amd.so')   # TG_LIBRARY: a diagnostic build

_CTYPES = {
    'const float*': ctypes.c_void_p, 'float*': ctypes.c_void_p,
    'const uint8_t*': ctypes.c_void_p, 'uint8_t*': ctypes.c_void_p, 'int64_t*': ctypes.c_void_p,
    'const double*': ctypes.c_void_p, 'double*': ctypes.c_void_p, 'const int64_t*': ctypes.c_void_p,
    'const int*': ctypes.c_void_p,
    'void*': ctypes.c_void_p, 'int': ctypes.c_int, 'int64_t': ctypes.c_int64,
    'size_t': ctypes.c_size_t, 'float': ctypes.c_float, 'const char*': ctypes.c_char_p,
    'const tg_host_i64*': ctypes.c_void_p,          # HOST array (a CPU int64 tensor), not a device pointer
}


def parse_header(path=HEADER):
    """-> {name: (return_type, [(type, argname), ...])} for every tg_* prototype."""
    with open(path) as f:
        src = f.read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    src = re.sub(r'//[^\n]*', ' ', src)
    protos = {}
    for m in re.finditer(r'(int|size_t|const char\*)\s+(tg_\w+)\s*\(([^)]*)\)\s*;', src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        params = []
        if args and args != 'void':
            for a in args.split(','):
                a = ' '.join(a.split())
                mm = re.match(r'(.+?)\s*(\w+)$', a)
                typ = mm.group(1).replace(' *', '*').strip()
                params.append((typ, mm.group(2)))
        protos[name] = (ret, params)
    return protos


class KernelError(RuntimeError):
    """A tg_* entry point returned a non-zero code (bad argument, unsupported shape or a HIP error)."""


class HipBackend:
    """Thin callable view of the shared library: ``K.conv2d_fwd(x, w, ...)``.

    Tensor arguments are passed as raw device pointers, ``None`` as NULL; the
    trailing ``stream`` argument is filled with the current PyTorch HIP stream.
    """
    name = 'hip'

    def __init__(self, path=LIBRARY):
        if not os.path.exists(path):
            raise RuntimeError(
                f'tartangan_amd: HIP library not found at {path}. Build it first '
                f'(python __graft_entry__.py build). There is no CPU fallback.')
        self._lib = ctypes.CDLL(path)
        self._protos = parse_header()
        for name, (ret, params) in self._protos.items():
            fn = getattr(self._lib, name)
            fn.restype = _CTYPES[ret]
            fn.argtypes = [_CTYPES[t] for t, _ in params]
            setattr(self, name[3:], self._wrap(name, fn, ret, params))

    @staticmethod
    def _wrap(name, fn, ret, params):
        has_stream = bool(params) and params[-1] == ('void*', 'stream')
        n_user = len(params) - (1 if has_stream else 0)
        ptr_pos = [i for i, (t, _) in enumerate(params[:n_user]) if t.endswith('*')]
        checked = ret == 'int' and has_stream

        def call(*args):
            if len(args) != n_user:
                raise TypeError(f'{name} takes {n_user} arguments, got {len(args)}')
            args = list(args)
            for i in ptr_pos:
                t = args[i]
                if t is None:
                    continue
                if 'tg_host' in params[i][0]:
                    if t.is_cuda or t.dtype != torch.int64 or not t.is_contiguous():
                        raise RuntimeError(f'{name}: argument {params[i][1]} must be a contiguous CPU int64 tensor')
                    args[i] = t.data_ptr()
                    continue
                if not t.is_cuda:
                    raise RuntimeError(f'{name}: argument {params[i][1]} is not on a ROCm device '
                                       f'(tartangan_amd has no CPU path)')
                if not t.is_contiguous():
                    raise RuntimeError(f'{name}: argument {params[i][1]} is not contiguous')
                args[i] = t.data_ptr()
            if has_stream:
                args.append(torch.cuda.current_stream().cuda_stream)
            rc = fn(*args)
            if checked and rc != 0:
                raise KernelError(f'{name} failed with code {rc}')
            return rc
        call.__name__ = name
        return call


_backend = None


def get():
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend


def is_available():
    return os.path.exists(LIBRARY)


def _set_backend_for_testing(b):
    """tests/ only: swap the kernel provider (None restores lazy loading of the HIP library)."""
    global _backend
    prev = _backend
    _backend = b
    return prev
