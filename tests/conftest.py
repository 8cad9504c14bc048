import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU side of a parity test is what takes the time (GPUTEST_r05: 822 s of the driver's 1 200 s; the SpaceInvaders agent
    # pipeline of the checker alone 20-30 s per test on ONE thread): the checker's batch loops are OpenMP-parallel over envs when
    # TBX_ORACLE_THREADS says so at engine creation, and envs never interact -- results do not depend on the thread count
    # (tests/test_oracle_golden.py::test_checker_results_do_not_depend_on_its_thread_count).  Every test engine gets the usable cores.
    os.environ.setdefault("TBX_ORACLE_THREADS", str(max(1, min(16, len(os.sched_getaffinity(0))))))


def _build(directory, target):
    path = os.path.join(ROOT, directory, target)
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, directory), target])
    return path


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle (test infrastructure): raw orc_* functions plus its restatement of the tbx_* ABI."""
    from toybox_amd import _abi
    path = _build("oracle", "liboracle.so")
    lib = ctypes.CDLL(path)
    _abi.bind(lib)
    lib.orc_rng_next.restype = ctypes.c_uint64
    lib.orc_rng_next.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    lib.orc_rng_range.restype = ctypes.c_uint64
    lib.orc_rng_range.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint64]
    lib.orc_rng_seed.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint32]
    lib.orc_rng_child.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    lib.orc_splitmix64.restype = ctypes.c_uint64
    lib.orc_splitmix64.argtypes = [ctypes.c_uint64]
    lib.orc_synthetic_action.restype = ctypes.c_int32
    lib.orc_synthetic_action.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64]
    return lib


@pytest.fixture(scope="session")
def hip_lib():
    """The product library (HIP, gfx950)."""
    from toybox_amd import _lib
    _build(os.path.join("toybox_amd", "csrc"), "libtoybox_amd.so")
    return _lib.load()


def has_gpu():
    return os.path.exists("/dev/kfd")
