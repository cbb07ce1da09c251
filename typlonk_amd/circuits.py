"""Synthetic benchmark circuit of SURVEY.md section 8(d): the squaring chain x_{j+1} = x_j * x_j with
n - 3 multiplication gates (README idiom `a.clone() * a`), built directly in HBM.

This is plumbing for bench.py and the tests -- the reference's circuit front-end (plonk::builder) is
out of scope -- but it produces exactly the tables that front-end would: selector columns
(Mul = [0, 0, 1, 1, 0], builder.rs:318-324), the copy-constraint permutation (a_j ~ b_j,
c_j ~ a_{j+1} ~ b_{j+1}) as sigma columns k_i' * w^j' (permutation/src/lib.rs:108-119, cosets 2, 3, 4)
and a satisfying witness with three blinding rows per column (proof.rs:43-49).
Big-integer work is limited to the n sequential squarings; everything else is numpy on limb arrays or
device kernels (typlonk_ntt / typlonk_lincomb)."""
from __future__ import annotations

import numpy as np

from .capi import Context, DeviceBuffer

FR_MODULUS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
COSETS = (2, 3, 4)


def fr_mont_limbs(x: int) -> np.ndarray:
    v = (x % FR_MODULUS) * (1 << 256) % FR_MODULUS
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def _canon_limbs(values) -> np.ndarray:
    """list of canonical ints -> (n, 4) u64 little-endian limbs (no Montgomery factor)"""
    raw = b"".join(int(v).to_bytes(32, "little") for v in values)
    return np.frombuffer(raw, dtype=np.uint64).reshape(-1, 4).copy()


def _to_montgomery(ctx: Context, canon: np.ndarray) -> DeviceBuffer:
    """upload canonical limbs and multiply by R on the device: lincomb with the scalar whose value is
    R = 2^256 (Montgomery form R^2) turns x (read as the residue of x/R) into the residue of x"""
    n = canon.shape[0]
    tmp, out = ctx.alloc(n), ctx.alloc(n)
    tmp.upload(canon)
    ctx.lincomb_dev([tmp], [fr_mont_limbs(1 << 256)], n, out)
    ctx.sync()
    tmp.free()
    return out


class SquaringChain:
    """Device-resident circuit tables + witness for n = 2^log_n rows."""

    def __init__(self, ctx: Context, log_n: int, x0: int = 3, blinder_seed: int = 0x5EED0000, keep_host: bool = False):
        """keep_host: keep the host copies of the selector / sigma evaluation tables (host_inputs(): what a CPU prover or
        a verifier-side check needs; 8 x 32 n bytes)"""
        self.ctx, self.log_n = ctx, log_n
        n = self.n = 1 << log_n
        g = self.gates = n - 3
        # ---- witness: a_j = b_j = x_j, c_j = x_j^2, then 3 blinding rows per column ----------------------
        xs = [x0 % FR_MODULUS]
        for _ in range(g):
            xs.append(xs[-1] * xs[-1] % FR_MODULUS)
        rng = np.random.default_rng(blinder_seed + log_n)
        bl = [[int(v) for v in rng.integers(1, 1 << 62, size=3)] for _ in range(3)]
        cols = [xs[:g] + bl[0], xs[:g] + bl[1], xs[1:g + 1] + bl[2]]
        self.wire_evals = [_to_montgomery(ctx, _canon_limbs(c)) for c in cols]
        self.pi_evals = None   # public inputs [0] (README.md:31): the zero polynomial
        # ---- selectors (evaluations): q_o = q_m = 1 on the gate rows, everything else 0 -------------------
        one = fr_mont_limbs(1)
        sel_ones = np.zeros((n, 4), dtype=np.uint64)
        sel_ones[:g] = one
        zeros = np.zeros((n, 4), dtype=np.uint64)
        sel_evals = [zeros, zeros, sel_ones, sel_ones, zeros]
        # ---- sigma columns from the domain elements w^j (= NTT of the polynomial X) ------------------------
        xpoly = np.zeros((n, 4), dtype=np.uint64)
        xpoly[1 % n] = one
        roots = ctx.alloc(n)
        roots.upload(xpoly)
        ctx.ntt_dev(roots, log_n)
        kr = []
        for k in COSETS:                           # k_i * w^j as limb arrays
            b = ctx.alloc(n)
            ctx.lincomb_dev([roots], [fr_mont_limbs(k)], n, b)
            kr.append(b.download())
            b.free()
        roots.free()
        ka, kb, kc = kr
        sa, sb, sc = ka.copy(), kb.copy(), kc.copy()   # identity permutation first
        sa[0] = kb[0]                                  # (a_0 b_0)
        sb[0] = ka[0]
        if g >= 2:                                     # (c_j a_{j+1} b_{j+1}) for j <= g - 2
            sc[0:g - 1] = ka[1:g]
            sa[1:g] = kb[1:g]
            sb[1:g] = kc[0:g - 1]
        # ---- coefficient forms and the cached circuit ------------------------------------------------------
        bufs = []
        for ev in sel_evals + [sa, sb, sc]:
            b = ctx.alloc(n)
            b.upload(ev)
            ctx.ntt_dev(b, log_n, inverse=True)
            bufs.append(b)
        self.circuit = ctx.circuit_load(log_n, bufs[:5], bufs[5:])
        for b in bufs:
            b.free()
        self.cosets = [fr_mont_limbs(k) for k in COSETS]
        self._host = {"selectors": sel_evals, "sigma": [sa, sb, sc]} if keep_host else None

    def host_inputs(self):
        """the prover's inputs as host arrays of Montgomery limbs: wire / selector / sigma evaluations over the domain"""
        if self._host is None:
            raise RuntimeError("SquaringChain(keep_host=True) keeps the host tables")
        return {"log_n": self.log_n, "wires": [b.download() for b in self.wire_evals], "selectors": self._host["selectors"],
                "sigma": self._host["sigma"], "cosets": list(COSETS)}

    def free(self):
        self.ctx.circuit_free(self.circuit)
        for b in self.wire_evals:
            b.free()
