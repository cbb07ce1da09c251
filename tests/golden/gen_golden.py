"""Generates tests/golden/*.json with the Python big-int oracle (oracle/bls12_381.py).

The reference (Rust, un-vendored arkworks) cannot be run in this environment, so these vectors
are NOT outputs of the reference; they are outputs of the independent affine / O(n^2) big-int
restatement, plus the literal known answers the reference's own tests imply (kzg `commit`:
commit(1+2X+3X^2) under s = 2 is 17*G, kzg/src/lib.rs:95-109).

    python tests/golden/gen_golden.py      # rewrites the fixtures deterministically
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import bls12_381 as O  # noqa: E402


def hx(v):
    return hex(v)


def pt(p):
    return None if p is None else [hx(p[0]), hx(p[1])]


def main():
    kat = {
        "p": hx(O.P), "r": hx(O.R), "g1": pt(O.G1),
        "two_g": pt(O.g1_mul(O.G1, 2)),
        "commit_1_2_3_s2": pt(O.kzg_commit(O.srs_from_secret(2, 10), [1, 2, 3])),
        "open_1_2_3_s2_z1": {"w": pt(O.kzg_open(O.srs_from_secret(2, 10), [1, 2, 3], 1)[0]), "y": hx(6)},
        "fr_root_of_unity_2_32": hx(O.FR_ROOT_OF_UNITY),
        "domain_roots": {str(k): hx(O.domain_root(k)) for k in (2, 3, 16, 20, 22, 24)},
        "ntt4_1_2_3_4": [hx(x) for x in O.ntt([1, 2, 3, 4], 2)],
        "g1_x_mont_limbs": [hx(x) for x in O.fq_to_mont_limbs(O.GX)],
        "g1_y_mont_limbs": [hx(x) for x in O.fq_to_mont_limbs(O.GY)],
        "fr_one_mont_limbs": [hx(x) for x in O.fr_to_mont_limbs(1)],
    }
    json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=1)

    # MSM vectors: secret, scalars -> expected affine point (naive per-term double-and-add + sum)
    msm = []
    for case, (s, m, seed) in enumerate([(2, 1, 11), (2, 3, 12), (0x0123456789ABCDEF0123456789ABCDEF, 8, 13),
                                         (0x0123456789ABCDEF0123456789ABCDEF, 33, 14), (5, 64, 15)]):
        srs = O.srs_from_secret_fast(s, m + 3)
        sc = O.random_frs(seed, m)
        msm.append({"secret": hx(s), "srs_len": m + 3, "scalars": [hx(x) for x in sc],
                    "expected": pt(O.msm_naive(sc, srs))})
    # edge cases
    srs = O.srs_from_secret_fast(2, 8)
    msm.append({"secret": hx(2), "srs_len": 8, "scalars": [], "expected": None})
    msm.append({"secret": hx(2), "srs_len": 8, "scalars": [hx(0)] * 5, "expected": None})
    msm.append({"secret": hx(2), "srs_len": 8, "scalars": [hx(O.R - 1)] * 5,
                "expected": pt(O.msm_naive([O.R - 1] * 5, srs))})
    json.dump(msm, open(os.path.join(HERE, "msm.json"), "w"), indent=1)

    # NTT vectors (the O(n^2) DFT is the ground truth for the forward direction)
    ntt = []
    for log_n in (1, 2, 3, 4, 6):
        n = 1 << log_n
        v = O.random_frs(100 + log_n, n)
        fwd = O.dft_naive(v, O.domain_root(log_n))
        assert fwd == O.ntt(v, log_n)
        ntt.append({"log_n": log_n, "input": [hx(x) for x in v], "forward": [hx(x) for x in fwd],
                    "inverse": [hx(x) for x in O.ntt(v, log_n, inverse=True)],
                    "coset7_forward": [hx(x) for x in O.ntt(v, log_n, coset=7)],
                    "coset7_inverse": [hx(x) for x in O.ntt(v, log_n, inverse=True, coset=7)]})
    json.dump(ntt, open(os.path.join(HERE, "ntt.json"), "w"), indent=1)
    print("golden fixtures written")


if __name__ == "__main__":
    main()
