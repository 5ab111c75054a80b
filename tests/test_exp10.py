"""10^x of the LPC path.  The reference evaluates `pow(10.f, e)` with the C library in double and rounds the product with
the band compensation to float (src/freq.c:317-318); one differing bit there changes an LPC coefficient and a free-running
stream never recovers.  The engine uses its own correctly rounded routine (lpcnet_amd/csrc/lpcnet_exp10.h) instead of the
device math library: the CPU test pins that routine against glibc over a strided sweep of the reachable argument range
(tolerance 0 on the float products), the GPU test checks that the device evaluates it to the same doubles as the host
build, and that the resulting floats equal glibc's over a dense sweep."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAND_COMP = np.array([0.8, 1., 1., 1., 1., 1., 1., 1., 0.666667, 0.5, 0.5, 0.5, 0.333333, 0.25, 0.25, 0.2, 0.166667, 0.173913], np.float32)
# e = log10 of a band energy.  |e| < 2^4 covers 10^-16..10^16; arguments below 2^-12 in magnitude give 10^e = 1 +- 6e-4
E_LO, E_HI = 127 - 12, 127 + 4


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("exp10") / "libexp10_host.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(ROOT, "lpcnet_amd", "csrc"),
                           os.path.join(ROOT, "tests", "tools", "exp10_host.cpp"), "-o", out])
    L = C.CDLL(out)
    fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
    L.exp10_engine.argtypes = [fp, dp, C.c_long]
    L.exp10_glibc.argtypes = [fp, dp, C.c_long]
    L.exp10_sweep.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_long)]
    return L


def test_engine_exp10_products_equal_glibc_over_the_reachable_range(host_lib):
    counts = (C.c_long * 3)()
    host_lib.exp10_sweep(E_LO, E_HI, 24, counts)               # 11 M arguments x 18 bands in about a second
    n, bad_d, bad_f = list(counts)
    assert n > 10_000_000
    assert bad_f == 0                                          # tolerance 0 on what the reference stores
    # (glibc itself is not correctly rounded for ~2^-10 of the arguments; those doubles differ by one ULP -- 80-digit
    # arithmetic says the engine's value is the correctly rounded one -- and never decide a float rounding here)
    assert bad_d < n * 2e-3
    x = np.array([0, 1, -1, 2, -2, 0.5, 10, -10, 38, -38, -45.5, 39, np.inf, -np.inf], np.float32)
    a, b = np.zeros(x.size), np.zeros(x.size)
    host_lib.exp10_engine(x, a, x.size); host_lib.exp10_glibc(x, b, x.size)
    assert np.array_equal(a, b)
    nan = np.array([np.nan], np.float32)
    host_lib.exp10_engine(nan, a[:1], 1)
    assert np.isnan(a[0])


@pytest.mark.gpu
def test_device_exp10_dense_sweep(host_lib, hip_lib):
    """every 3rd float of the reachable range through the device routine: identical doubles to the host build of the
    same routine, identical float products to glibc"""
    fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
    hip_lib.lpcnet_hip_exp10_device.argtypes = [fp, dp, C.c_int]
    total = bad = 0
    for sign in (0, 1):
        for e in range(E_LO, E_HI):
            bits = (np.uint32(sign) << np.uint32(31)) | (np.uint32(e) << np.uint32(23)) | np.arange(e % 3, 1 << 23, 3, dtype=np.uint32)
            x = bits.view(np.float32)
            dev, host, ref = np.zeros(x.size), np.zeros(x.size), np.zeros(x.size)
            assert hip_lib.lpcnet_hip_exp10_device(x, dev, x.size) == 0
            host_lib.exp10_engine(x, host, x.size)
            assert np.array_equal(dev, host), (sign, e)
            host_lib.exp10_glibc(x, ref, x.size)
            d = np.nonzero(dev != ref)[0]
            for c in BAND_COMP:
                bad += int(((dev[d] * np.float64(c)).astype(np.float32) != (ref[d] * np.float64(c)).astype(np.float32)).sum())
            total += x.size
    assert total > 80_000_000 and bad == 0
