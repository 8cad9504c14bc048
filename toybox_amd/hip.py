"""Minimal ctypes binding of the HIP runtime (streams, events, device memory) -- plumbing for
callers that keep everything resident on the GPU and do not want a PyTorch dependency."""
import ctypes as C
import ctypes.util

_hip = None


class HipError(RuntimeError):
    pass


def runtime():
    global _hip
    if _hip is None:
        for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so", ctypes.util.find_library("amdhip64")):
            if not name:
                continue
            try:
                _hip = C.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise HipError("libamdhip64.so not found")
        _hip.hipGetErrorString.restype = C.c_char_p
        _hip.hipGetErrorString.argtypes = [C.c_int]
    return _hip


def check(rc, what=""):
    if rc != 0:
        raise HipError("%s failed: %s" % (what or "HIP call", runtime().hipGetErrorString(rc).decode()))


def device_count():
    n = C.c_int(0)
    rc = runtime().hipGetDeviceCount(C.byref(n))
    return n.value if rc == 0 else 0


def set_device(i):
    check(runtime().hipSetDevice(C.c_int(i)), "hipSetDevice")


def synchronize():
    check(runtime().hipDeviceSynchronize(), "hipDeviceSynchronize")


class Stream:
    def __init__(self):
        self.handle = C.c_void_p()
        check(runtime().hipStreamCreate(C.byref(self.handle)), "hipStreamCreate")

    @property
    def ptr(self):
        return self.handle.value or 0

    def synchronize(self):
        check(runtime().hipStreamSynchronize(self.handle), "hipStreamSynchronize")

    def close(self):
        if self.handle:
            runtime().hipStreamDestroy(self.handle)
            self.handle = C.c_void_p()


class Event:
    def __init__(self):
        self.handle = C.c_void_p()
        check(runtime().hipEventCreate(C.byref(self.handle)), "hipEventCreate")

    def record(self, stream=None):
        s = stream.handle if isinstance(stream, Stream) else C.c_void_p(int(stream or 0))
        check(runtime().hipEventRecord(self.handle, s), "hipEventRecord")

    def synchronize(self):
        check(runtime().hipEventSynchronize(self.handle), "hipEventSynchronize")

    def elapsed_ms(self, later):
        ms = C.c_float(0)
        check(runtime().hipEventElapsedTime(C.byref(ms), self.handle, later.handle), "hipEventElapsedTime")
        return ms.value

    def close(self):
        if self.handle:
            runtime().hipEventDestroy(self.handle)
            self.handle = C.c_void_p()


def malloc(nbytes):
    p = C.c_void_p()
    check(runtime().hipMalloc(C.byref(p), C.c_size_t(nbytes)), "hipMalloc")
    return p.value


def free(ptr):
    if ptr:
        runtime().hipFree(C.c_void_p(ptr))


def memcpy_dtoh(dst_array, src_ptr, nbytes):
    check(runtime().hipMemcpy(dst_array.ctypes.data_as(C.c_void_p), C.c_void_p(src_ptr), C.c_size_t(nbytes), C.c_int(2)),
          "hipMemcpy D2H")


def memcpy_htod(dst_ptr, src_array, nbytes):
    check(runtime().hipMemcpy(C.c_void_p(dst_ptr), src_array.ctypes.data_as(C.c_void_p), C.c_size_t(nbytes), C.c_int(1)),
          "hipMemcpy H2D")


def _free_pinned(address):
    try:
        runtime().hipHostFree(C.c_void_p(address))
    except Exception:
        pass


class PinnedArray:
    """A numpy array over page-locked host memory (hipHostMalloc): the destination of frame copies that are to run at the PCIe
    link's rate (a copy into pageable memory is staged through a bounce buffer and faults fresh pages in).  The memory belongs to
    the array and its views -- a finaliser on the ctypes buffer under them frees it when the last one has gone, by reference
    counting alone (no owner <-> buffer cycle); close() only drops this object's reference (an observation kept past
    close() used to point at freed memory: ADVICE r04)."""

    def __init__(self, shape, dtype="uint8"):
        import weakref
        import numpy as np
        count = int(np.prod(shape))
        self.nbytes = count * np.dtype(dtype).itemsize
        ptr = C.c_void_p()
        check(runtime().hipHostMalloc(C.byref(ptr), C.c_size_t(max(self.nbytes, 1)), C.c_uint(0)), "hipHostMalloc")
        buf = (C.c_uint8 * max(self.nbytes, 1)).from_address(ptr.value)
        weakref.finalize(buf, _free_pinned, ptr.value)
        self.array = np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)

    def close(self):
        self.array = None


def mem_info():
    free_b, total_b = C.c_size_t(), C.c_size_t()
    check(runtime().hipMemGetInfo(C.byref(free_b), C.byref(total_b)), "hipMemGetInfo")
    return free_b.value, total_b.value


def memcpy_dtod_async(dst_ptr, src_ptr, nbytes, stream=0):
    s = stream.handle if isinstance(stream, Stream) else C.c_void_p(int(stream or 0))
    check(runtime().hipMemcpyAsync(C.c_void_p(dst_ptr), C.c_void_p(src_ptr), C.c_size_t(nbytes), C.c_int(3), s),
          "hipMemcpyAsync D2D")
