"""`ctoybox`-named shim used ONLY by the test-suite in the build container: it lets the UNMODIFIED reference
code (/root/reference/toybox/interventions, /root/reference/test/interventions) import `ctoybox` and run
against this repo's host layer.  Engines are backed by the CPU oracle here because the container has no GPU
(tests/test_reference_suite.py); the same host code is exercised over the HIP library by the GPU tests."""
import ctypes
import os

from toybox_amd import Engine, _abi
from toybox_amd import toybox as _tb
from toybox_amd.toybox import Input, Simulator, State, Toybox  # noqa: F401

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
_backend = os.environ.get("TOYBOX_AMD_TEST_BACKEND", "oracle")
if _backend == "oracle":
    _lib = ctypes.CDLL(os.path.join(_ROOT, "oracle", "liboracle.so"))
    _abi.bind(_lib)
    _tb.set_engine_factory(lambda game, n: Engine(game, n, lib=_lib))
