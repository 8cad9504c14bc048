"""Loader of the in-tree HIP library.  There is no fallback: a missing library is an ImportError-like
failure at first use, and a missing GPU is reported by tbx_create (TBX_E_NO_DEVICE)."""
import ctypes
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libtoybox_amd.so")
_lib = None


class ToyboxAmdError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("toybox_amd error %d: %s" % (code, message))
        self.code = code


def load():
    """Return the bound ctypes library (libtoybox_amd.so built in-tree by toybox_amd/csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ToyboxAmdError(_abi.E_NO_DEVICE,
                                 "HIP extension not built: %s is missing (run `make -C toybox_amd/csrc` "
                                 "or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        _abi.bind(lib)
        v = lib.tbx_abi_version()
        if v != _abi.ABI_VERSION:
            raise ToyboxAmdError(_abi.E_INVALID, "ABI version mismatch: library %d, host %d" % (v, _abi.ABI_VERSION))
        _lib = lib
    return _lib
