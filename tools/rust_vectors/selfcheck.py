"""Writes a file in the format of tests/emit.rs from THIS repository's oracle and transcript statement and runs
tests/test_reference_vectors.py against it: checks that the loader's code paths work.  It pins nothing (the file is the
oracle's own output) and is never committed."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import bls12_381 as O
import transcript_ref as T

hx = lambda l: ["%016x" % int(v) for v in l]
fr = lambda x: hx(O.fr_to_mont_limbs(x))
def pt(p):
    l, f = O.g1_to_limbs(p)
    return {"xy": hx(l), "inf": int(f)}
def abi(p):
    l, f = O.g1_to_limbs(p)
    return (l, f)
G = O.G1
mult = lambda k: O.g1_mul(G, k)
out = {"serialize_unchecked": [{"point": pt(p), "bytes": T.serialize_unchecked_g1(*abi(p)).hex()} for p in (G, mult(2), None)]}
trs = []
for pts, n in [([mult(i) for i in range(1, k + 1)], 2) for k in range(5)] + [([G, None, mult(7)], 3)]:
    ch = T.ChallengeGenerator.with_digest([abi(p) for p in pts]).generate_challenges(n)
    trs.append({"points": [pt(p) for p in pts], "challenges": [hx(c) for c in ch]})
out["transcripts"] = trs
rng = T.StdRng.seed_from_u64(1)
out["stdrng_seed_1_next_u64"] = hx([rng.next_u64() for _ in range(8)])
rng = T.StdRng.seed_from_u64(1)
out["fr_rand_seed_1"] = [hx(T.fr_rand(rng)) for _ in range(4)]
srs = O.srs_from_secret(2, 10)
w, y = O.kzg_open(srs, [1, 2, 3], 1)
out["kzg_commit_1_2_3_s2"] = {"commitment": pt(O.kzg_commit(srs, [1, 2, 3])), "open_at_1": {"witness": pt(w), "eval": fr(y)}}
out["srs_s2"] = [pt(p) for p in srs[:6]]
secret = 0x0123456789abcdef0123456789abcdef
rng = T.StdRng.seed_from_u64(7)
coeffs = [O.fr_from_mont_limbs([int(v) for v in T.fr_rand(rng)]) for _ in range(8)]
out["msm8"] = {"secret": fr(secret), "coeffs": [fr(c) for c in coeffs], "commitment": pt(O.kzg_commit(O.srs_from_secret(secret, 8), coeffs))}
srs3 = O.srs_from_secret(secret, 40)
out["srs_slice"] = {"secret": fr(secret), "start": 30, "points": [pt(p) for p in srs3[30:36]]}
aff = O.g1_add(O.g1_mul(G, 2 * 0x1234567), O.g1_mul(G, coeffs[0]))
z = 0x1F2E3D4C5B6A79880112233445566778899AABBCCDDEEFF % O.P
out["into_affine"] = {"x": hx(O.fq_to_mont_limbs(aff[0] * z * z % O.P)), "y": hx(O.fq_to_mont_limbs(aff[1] * z * z * z % O.P)),
                      "z": hx(O.fq_to_mont_limbs(z)), "affine": pt(aff)}
v = [1, 2, 3, 4]
out["fft"] = {"input": [fr(x) for x in v], "fft4": [fr(x) for x in O.ntt(v, 2)], "ifft4": [fr(x) for x in O.ntt(v, 2, inverse=True)],
              "fft8_of_msm8_coeffs": [fr(x) for x in O.ntt(coeffs, 3)], "coset_fft8_of_msm8_coeffs": [fr(x) for x in O.ntt(coeffs, 3, coset=7)],
              "group_gen_8": fr(O.domain_root(3))}
cols = [[O.fr_from_mont_limbs([int(v) for v in T.fr_rand(rng)]) for _ in range(8)] for _ in range(3)]   # drawn after the msm8 coefficients
out["interpolate3"] = {"columns": [[fr(x) for x in c] for c in cols], "polys": [[fr(x) for x in O.ntt(c, 3, inverse=True)] for c in cols]}
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "reference_vectors.json")
    json.dump(out, open(path, "w"))
    env = dict(os.environ, TYPLONK_REFERENCE_VECTORS=path)
    sys.exit(subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_reference_vectors.py"), "-q", "-m", "not gpu"], env=env, cwd=ROOT).returncode)
