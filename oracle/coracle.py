"""ctypes access to oracle/liboracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY -- see the
header of oracle/typlonk_oracle.c.  Arrays are numpy uint64 in the C-ABI (arkworks) limb form."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            import subprocess
            subprocess.run(["gcc", "-O2", "-std=c11", "-shared", "-fPIC", os.path.join(_HERE, "typlonk_oracle.c"),
                            "-o", _PATH], check=True)
        _lib = C.CDLL(_PATH)
    return _lib


def _p64(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _p8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def msm_reference(scalars, xy, inf=None):
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    xy = np.ascontiguousarray(xy, dtype=np.uint64).reshape(-1, 12)
    m, n = scalars.shape[0], xy.shape[0]
    out = np.zeros(12, dtype=np.uint64)
    oinf = np.zeros(1, dtype=np.uint8)
    infp = None
    if inf is not None:
        inf = np.ascontiguousarray(inf, dtype=np.uint8)
        infp = _p8(inf)
    rc = lib().oracle_msm_reference(_p64(scalars), _p64(xy), infp, C.c_size_t(m), C.c_size_t(n), _p64(out), _p8(oinf))
    if rc:
        raise AssertionError("srs.len() > polynomial.degree() failed (kzg/src/lib.rs:43)")
    return out, int(oinf[0])


def msm_pippenger(scalars, xy, inf=None, c: int = 13):
    """bucket-method MSM on all host cores (the "fair CPU" baseline); returns (xy, inf, group_ops, threads)"""
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    xy = np.ascontiguousarray(xy, dtype=np.uint64).reshape(-1, 12)
    m, n = scalars.shape[0], xy.shape[0]
    out = np.zeros(12, dtype=np.uint64)
    oinf = np.zeros(1, dtype=np.uint8)
    infp = None
    if inf is not None:
        inf = np.ascontiguousarray(inf, dtype=np.uint8)
        infp = _p8(inf)
    ops, thr = C.c_uint64(), C.c_int()
    rc = lib().oracle_msm_pippenger(_p64(scalars), _p64(xy), infp, C.c_size_t(m), C.c_size_t(n), C.c_uint32(c), _p64(out),
                                    _p8(oinf), C.byref(ops), C.byref(thr))
    if rc:
        raise AssertionError(f"oracle_msm_pippenger failed: {rc}")
    return out, int(oinf[0]), int(ops.value), int(thr.value)


def srs_from_secret(s_mont, length):
    s = np.ascontiguousarray(s_mont, dtype=np.uint64).reshape(4)
    xy = np.zeros((length, 12), dtype=np.uint64)
    inf = np.zeros(length, dtype=np.uint8)
    lib().oracle_srs_from_secret(_p64(s), C.c_size_t(length), _p64(xy), _p8(inf))
    return xy, inf


def srs_pow2_secret(log2_s, length):
    xy = np.zeros((length, 12), dtype=np.uint64)
    inf = np.zeros(length, dtype=np.uint8)
    rc = lib().oracle_srs_pow2_secret(C.c_uint32(log2_s), C.c_size_t(length), _p64(xy), _p8(inf))
    assert rc == 0
    return xy, inf


def g1_mul_generator(k_mont):
    k = np.ascontiguousarray(k_mont, dtype=np.uint64).reshape(4)
    out = np.zeros(12, dtype=np.uint64)
    oinf = np.zeros(1, dtype=np.uint8)
    lib().oracle_g1_mul_generator(_p64(k), _p64(out), _p8(oinf))
    return out, int(oinf[0])


def poly_eval(coeffs, x_mont):
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    x = np.ascontiguousarray(x_mont, dtype=np.uint64).reshape(4)
    out = np.zeros(4, dtype=np.uint64)
    lib().oracle_poly_eval(_p64(coeffs), C.c_size_t(coeffs.shape[0]), _p64(x), _p64(out))
    return out


def poly_div_linear(coeffs, z_mont):
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    n = coeffs.shape[0]
    z = np.ascontiguousarray(z_mont, dtype=np.uint64).reshape(4)
    q = np.zeros((max(n - 1, 0), 4), dtype=np.uint64)
    y = np.zeros(4, dtype=np.uint64)
    lib().oracle_poly_div_linear(_p64(coeffs), C.c_size_t(n), _p64(z), _p64(q), _p64(y))
    return q, y


def ntt(data, log_n, inverse=False, coset=None, threads=1):
    """threads = 1: oracle_ntt (one thread, like the reference); 0 = all cores, k = k threads: oracle_ntt_mt (OpenMP)"""
    data = np.ascontiguousarray(data, dtype=np.uint64).reshape(-1, 4).copy()
    assert data.shape[0] == (1 << log_n)
    cp = None
    if coset is not None:
        coset = np.ascontiguousarray(coset, dtype=np.uint64).reshape(4)
        cp = _p64(coset)
    if threads == 1:
        rc = lib().oracle_ntt(_p64(data), C.c_uint32(log_n), C.c_int(int(inverse)), cp)
    else:
        rc = lib().oracle_ntt_mt(_p64(data), C.c_uint32(log_n), C.c_int(int(inverse)), cp, C.c_int(threads))
    if rc:
        raise AssertionError("GeneralEvaluationDomain::new(..).unwrap() failed (plonk/src/builder.rs:70)")
    return data


def quotient_schoolbook_products(poly, n):
    """the twelve naive_mul products of quotient_polynomial with their operand shapes, one thread (timing only)"""
    poly = np.ascontiguousarray(poly, dtype=np.uint64).reshape(-1, 4)
    assert poly.shape[0] >= n
    out = np.zeros(4, dtype=np.uint64)
    rc = lib().oracle_quotient_schoolbook_products(_p64(poly), C.c_size_t(n), _p64(out))
    assert rc == 0
    return out


def fr_vec_add(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    out = np.empty_like(a)
    lib().oracle_fr_vec_add(_p64(a), _p64(b), C.c_size_t(a.shape[0]), _p64(out))
    return out


def group_ops_reset():
    lib().oracle_group_ops_reset()


def group_ops() -> int:
    f = lib().oracle_group_ops
    f.restype = C.c_uint64
    return int(f())
