"""Shared helpers for the tests: limb packing between Python ints and the C-ABI's arkworks form."""
from __future__ import annotations
import ctypes
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import bls12_381 as O  # noqa: E402


def fr_pack(vals) -> np.ndarray:
    """list of canonical ints -> (n,4) uint64 Montgomery limbs"""
    out = np.empty((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i] = O.fr_to_mont_limbs(v)
    return out


def fr_unpack(arr) -> list[int]:
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [O.fr_from_mont_limbs([int(x) for x in row]) for row in arr]


def g1_pack(points):
    """list of affine points (or None) -> ((n,12) uint64, (n,) uint8)"""
    xy = np.empty((len(points), 12), dtype=np.uint64)
    inf = np.zeros(len(points), dtype=np.uint8)
    for i, p in enumerate(points):
        limbs, f = O.g1_to_limbs(p)
        xy[i] = limbs
        inf[i] = f
    return xy, inf


def g1_unpack_one(xy, inf):
    return O.g1_from_limbs([int(x) for x in np.asarray(xy, dtype=np.uint64).reshape(12)], int(inf))


def u32p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))


def load_golden(name):
    import json
    with open(os.path.join(ROOT, "tests", "golden", name)) as f:
        return json.load(f)


def hex_pt(p):
    return None if p is None else (int(p[0], 16), int(p[1], 16))
