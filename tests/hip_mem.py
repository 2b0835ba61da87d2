"""Device buffers for tests that call the *_device entry points, through the HIP runtime that libiris_hip.so itself
is linked against (found in this process's map after the library is loaded) -- no second runtime, no torch."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        from iris_amd import ffi
        ffi.load()
        path = None
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        assert path, "libiris_hip.so did not bring a HIP runtime with it"
        _hip = C.CDLL(path)
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        _hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    return _hip


class DeviceFilm:
    """(h, w, 3) f32 in device memory, filled with a byte pattern so that unwritten pixels show."""

    def __init__(self, h, w, fill_byte=0x7f):
        self.shape = (h, w, 3)
        self.nbytes = h * w * 3 * 4
        self.ptr = C.c_void_p()
        assert hip().hipMalloc(C.byref(self.ptr), max(self.nbytes, 16)) == 0
        assert hip().hipMemset(self.ptr, fill_byte, self.nbytes) == 0

    def data_ptr(self):
        return self.ptr.value

    def numpy(self):
        out = np.empty(self.shape, np.float32)
        assert hip().hipDeviceSynchronize() == 0
        assert hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes, 2) == 0  # hipMemcpyDeviceToHost
        return out

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = C.c_void_p()

    __del__ = free


# ---- plain buffers ----
def alloc(nbytes):
    p = C.c_void_p()
    assert hip().hipMalloc(C.byref(p), max(int(nbytes), 16)) == 0
    return p.value


def free(ptr):
    if ptr:
        hip().hipFree(C.c_void_p(ptr))


def to_device(ptr, array):
    a = np.ascontiguousarray(array)
    assert hip().hipMemcpy(C.c_void_p(ptr), a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0  # hipMemcpyHostToDevice


def to_host(ptr, shape, dtype):
    out = np.empty(shape, dtype)
    assert hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes, 2) == 0  # hipMemcpyDeviceToHost
    return out


def synchronize():
    assert hip().hipDeviceSynchronize() == 0


def free_bytes():
    """Free device memory on the current GPU as the runtime reports it (hipMemGetInfo)."""
    free_, total = C.c_size_t(), C.c_size_t()
    assert hip().hipDeviceSynchronize() == 0
    assert hip().hipMemGetInfo(C.byref(free_), C.byref(total)) == 0
    return free_.value
