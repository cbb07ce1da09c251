"""Fiat-Shamir transcript of the reference prover, host side.

Mirror of plonk::proof::challenges::ChallengeGenerator (/root/reference/plonk/src/proof/challenges.rs:9-46):

    digest(c)               append ark-serialize `serialize_unchecked` bytes of the G1Affine commitment (:17-22)
    generate_challenges<N>  Blake2b-512 over the bytes, first 8 bytes LE -> u64 -> StdRng::seed_from_u64 -> N x Fr::rand
                            (:30-45)

and of the two squeezes in prove() (/root/reference/plonk/src/proof.rs:111, :133-136): (beta, gamma) from
[a], [b], [c]; (alpha, zeta) from [a], [b], [c], [Z].

Everything below the Blake2b call lives in crates that are not in this container and there is no Rust toolchain,
so this module is written from the published crate behaviour and is NOT verified against the reference:

  * ark-serialize 0.3.0 / ark-ec 0.3.0 (Cargo.lock:95-96, 28-29): uncompressed G1Affine = x (48 B little-endian,
    canonical) || y (48 B little-endian, canonical) with SWFlags in the two top bits of the LAST byte: infinity =
    0x40, otherwise 0 (the y-sign flag is only set by the compressed form); the identity is stored as (0, 1).
  * rand_core 0.6.3 `SeedableRng::seed_from_u64` (Cargo.lock:457-458): PCG32 (MUL 6364136223846793005, INC
    11634580027462260723, xorshift 18/27, rotate by the top 5 bits), eight 32-bit outputs -> 32-byte seed.
  * rand 0.8.4 `StdRng` = rand_chacha 0.3.1 `ChaCha12Rng` (Cargo.lock:435-436, 447-448): key = seed, 64-bit block
    counter from 0 in words 12-13, stream id 0 in words 14-15, 12 rounds; `next_u64` = two consecutive 32-bit
    output words, low word first.
  * ark-ff 0.3.0 `Fp256::rand` (Cargo.lock:42-43): four `next_u64` limbs, clear the top REPR_SHAVE_BITS = 1 bit of
    the last limb, retry while the value is >= r; the limbs are used AS the Montgomery representation.

Pinned here only as far as public vectors go: the ChaCha block function against RFC 7539 section 2.3.2 (20 rounds) and
Blake2b by hashlib (tests/test_host.py).  Challenges are returned in the C-ABI form (4 little-endian u64 Montgomery
limbs), ready for typlonk_prover_round2 / round3.
"""
from __future__ import annotations

import hashlib
import struct

import numpy as np

FQ_MODULUS = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
FR_MODULUS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
_FQ_RINV = pow(1 << 384, -1, FQ_MODULUS)
_M32, _M64 = 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF


def serialize_unchecked_g1(xy, inf) -> bytes:
    """(xy[12] u64 Montgomery limbs, infinity flag) -> 96 bytes"""
    limbs = [int(v) for v in np.asarray(xy, dtype=np.uint64).reshape(12)]
    if inf:
        x, y, flag = 0, 1, 0x40
    else:
        x = sum(l << (64 * i) for i, l in enumerate(limbs[:6])) * _FQ_RINV % FQ_MODULUS
        y = sum(l << (64 * i) for i, l in enumerate(limbs[6:])) * _FQ_RINV % FQ_MODULUS
        flag = 0
    out = bytearray(x.to_bytes(48, "little") + y.to_bytes(48, "little"))
    out[-1] |= flag
    return bytes(out)


def seed_from_u64(state: int) -> bytes:
    seed = b""
    for _ in range(8):
        state = (state * 6364136223846793005 + 11634580027462260723) & _M64
        xorshifted = (((state >> 18) ^ state) >> 27) & _M32
        rot = state >> 59
        seed += struct.pack("<I", ((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & _M32)
    return seed


def _rotl(v, n):
    return ((v << n) | (v >> (32 - n))) & _M32


def chacha_block(key_words, counter: int, stream: int = 0, rounds: int = 12, state_tail=None):
    """16 output words of one ChaCha block; state words 12..15 = 64-bit counter, 64-bit stream id (or state_tail)"""
    tail = state_tail if state_tail is not None else [counter & _M32, (counter >> 32) & _M32, stream & _M32, (stream >> 32) & _M32]
    init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + list(tail)
    x = list(init)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & _M32; x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & _M32; x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & _M32; x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & _M32; x[b] = _rotl(x[b] ^ x[c], 7)

    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & _M32 for a, b in zip(x, init)]


class StdRng:
    """ChaCha12Rng as rand 0.8.4's StdRng: a stream of 32-bit words, next_u64 = (low, high)"""

    def __init__(self, seed: bytes):
        self.key = list(struct.unpack("<8I", seed))
        self.counter = 0
        self.buf: list[int] = []

    @classmethod
    def seed_from_u64(cls, state: int) -> "StdRng":
        return cls(seed_from_u64(state))

    def next_u32(self) -> int:
        if not self.buf:
            self.buf = chacha_block(self.key, self.counter)
            self.counter += 1
        return self.buf.pop(0)

    def next_u64(self) -> int:
        lo = self.next_u32()
        return lo | (self.next_u32() << 32)


def fr_rand(rng: StdRng) -> np.ndarray:
    """ark-ff Fp256::rand -> the 4 limbs (which ARE the Montgomery representation)"""
    while True:
        limbs = [rng.next_u64() for _ in range(4)]
        limbs[3] &= _M64 >> 1
        if sum(l << (64 * i) for i, l in enumerate(limbs)) < FR_MODULUS:
            return np.array(limbs, dtype=np.uint64)


class ChallengeGenerator:
    def __init__(self):
        self.data = bytearray()

    def digest(self, commitment) -> "ChallengeGenerator":
        """commitment = (xy[12], inf) as the C ABI returns it"""
        self.data += serialize_unchecked_g1(commitment[0], commitment[1])
        return self

    @classmethod
    def with_digest(cls, commitments) -> "ChallengeGenerator":
        g = cls()
        for c in commitments:
            g.digest(c)
        return g

    def generate_challenges(self, n: int):
        h = hashlib.blake2b(bytes(self.data), digest_size=64).digest()
        rng = StdRng.seed_from_u64(int.from_bytes(h[:8], "little"))
        return [fr_rand(rng) for _ in range(n)]


def challenge12(commitments):
    """(beta, gamma) from [a], [b], [c]  (proof.rs:111)"""
    return tuple(ChallengeGenerator.with_digest(commitments[:3]).generate_challenges(2))


def challenge34(commitments_and_z):
    """(alpha, zeta) from [a], [b], [c], [Z]  (proof.rs:133-136)"""
    return tuple(ChallengeGenerator.with_digest(commitments_and_z[:4]).generate_challenges(2))
