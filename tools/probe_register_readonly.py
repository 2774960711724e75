#!/usr/bin/env python3
"""Probe (dev tool): what hipHostRegister / hipHostUnregister / hipPointerGetAttributes answer for a READ-ONLY file mapping and for an
anonymous one, and whether anything is left behind after the unregister."""
import ctypes as C
import mmap
import os
import tempfile

hip = C.CDLL("libamdhip64.so")
hip.hipGetErrorName.restype = C.c_char_p


class Attr(C.Structure):
    _fields_ = [("type", C.c_int), ("device", C.c_int), ("devicePointer", C.c_void_p), ("hostPointer", C.c_void_p), ("isManaged", C.c_int),
                ("allocationFlags", C.c_uint)]


def name(rc):
    return hip.hipGetErrorName(rc).decode()


def attrs(p):
    a = Attr()
    rc = hip.hipPointerGetAttributes(C.byref(a), C.c_void_p(p))
    hip.hipGetLastError()
    return f"{name(rc)} type={a.type}" if rc == 0 else name(rc)


def main():
    hip.hipInit(0)
    hip.hipSetDevice(0)
    n = 3 << 20
    path = os.path.join(tempfile.mkdtemp(), "x.bin")
    open(path, "wb").write(os.urandom(n))
    libc = C.CDLL(None)
    libc.mmap.restype = C.c_void_p
    libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
    libc.munmap.argtypes = [C.c_void_p, C.c_size_t]
    fd = os.open(path, os.O_RDONLY)
    for label, prot, flags, f in (("read-only file mapping (PROT_READ, MAP_SHARED)", 1, 1, fd), ("read-only file mapping (PROT_READ, MAP_PRIVATE)", 1, 2, fd),
                                  ("anonymous read-write", 3, 0x22, -1)):
        p = libc.mmap(None, n, prot, flags, f, 0)
        print(f"{label}: at {p:#x}")
        print("  before:", attrs(p))
        for flag, fname in ((1, "hipHostRegisterPortable"), (1 | 8, "Portable | ReadOnly(0x8)")):
            rc = hip.hipHostRegister(C.c_void_p(p), C.c_size_t(n), C.c_uint(flag))
            hip.hipGetLastError()
            print(f"  hipHostRegister({fname}): {name(rc)}; attributes now: {attrs(p)}; last byte: {attrs(p + n - 1)}")
            if rc == 0:
                d = C.c_void_p()
                hip.hipMalloc(C.byref(d), C.c_size_t(n))
                rc2 = hip.hipMemcpy(d, C.c_void_p(p), C.c_size_t(n), 1)
                rc3 = hip.hipHostUnregister(C.c_void_p(p))
                hip.hipGetLastError()
                print(f"    hipMemcpy H2D from it: {name(rc2)}; hipHostUnregister: {name(rc3)}; attributes after: {attrs(p)}")
                hip.hipFree(d)
        libc.munmap(C.c_void_p(p), n)
        # the same addresses again, as ordinary memory
        q = libc.mmap(C.c_void_p(p), n, 3, 0x22 | 0x10, -1, 0)  # MAP_FIXED
        d = C.c_void_p()
        hip.hipMalloc(C.byref(d), C.c_size_t(n))
        print(f"  remapped anonymous at the same address {q:#x}: attributes {attrs(q)}; pageable hipMemcpy H2D {name(hip.hipMemcpy(d, C.c_void_p(q), C.c_size_t(n), 1))}, "
              f"D2H {name(hip.hipMemcpy(C.c_void_p(q), d, C.c_size_t(n), 2))}; hipHostRegister {name(hip.hipHostRegister(C.c_void_p(q), C.c_size_t(n), C.c_uint(1)))}, "
              f"unregister {name(hip.hipHostUnregister(C.c_void_p(q)))}")
        hip.hipFree(d)
        libc.munmap(C.c_void_p(q), n)


if __name__ == "__main__":
    main()
