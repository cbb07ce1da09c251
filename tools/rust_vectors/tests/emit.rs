//! Emits `reference_vectors.json`: outputs of the reference crate `kzg` and of the third-party crates its prover
//! relies on (ark-ff / ark-ec / ark-poly / ark-serialize 0.3, rand 0.8 StdRng, blake2 0.9), in the limb formats this
//! repository's C ABI uses (include/typlonk.h): Fr = 4 x u64 Montgomery limbs (`fr.0.0`), Fq = 6 x u64 (`pt.x.0.0`).
//!
//! What each vector pins (file:line in fabrizio-m/TyPLONK):
//!   serialize_unchecked      plonk/src/proof/challenges.rs:17-22   (G, 2G, the point at infinity)
//!   challenges               plonk/src/proof/challenges.rs:30-45   (Blake2b-512 -> first 8 bytes LE -> StdRng::seed_from_u64
//!                                                                   -> Fr::rand), for transcripts of 0..4 commitments
//!   stdrng / fr_rand         rand 0.8 StdRng::seed_from_u64(1): next_u64 words and Fr::rand outputs
//!   commit / open            kzg/src/lib.rs:37-64, the crate's own test polynomial 1 + 2X + 3X^2 under s = 2 (:95-109)
//!   srs                      kzg/src/srs.rs:15-34
//!   fft / ifft               ark-poly Radix2EvaluationDomain as reached from plonk/src/proof.rs:50, 115
//!   interpolate3             a group of three `Evaluations::interpolate` calls (plonk/src/proof.rs:50, builder.rs:84-88):
//!                            what one typlonk_ntt_fr_batch_devptr(inverse) call must return
//!   msm8                     evaluate_in_s on an 8-term polynomial with a large secret
//!   srs_slice                kzg/src/srs.rs:15-24 at a NON-ZERO start: powers 30..36 of the large secret (what
//!                            typlonk_srs_generate(start = 30) must reproduce -- fixed-base comb + divsteps inversion)
//!   into_affine              kzg/src/lib.rs:50 / srs.rs:20: the per-term `into()` of a Jacobian point with Z != 1
//!                            (ark-ec 0.3 `From<GroupProjective> for GroupAffine`: x = X / Z^2, y = Y / Z^3)
//! `plonk::proof::challenges` is a private module, so the generator's call sequence is issued here directly against
//! the same crates.
use ark_bls12_381::{Fr, G1Affine};
use ark_ec::{AffineCurve, ProjectiveCurve};
use ark_ff::{PrimeField, UniformRand, Zero};
use ark_poly::{univariate::DensePolynomial, EvaluationDomain, GeneralEvaluationDomain, UVPolynomial};
use ark_serialize::CanonicalSerialize;
use blake2::{Blake2b, Digest};
use kzg::{srs::Srs, KzgCommitment, KzgScheme};
use rand::{rngs::StdRng, RngCore, SeedableRng};
use std::convert::TryInto;
use std::fmt::Write as _;

fn hex_limbs(l: &[u64]) -> String {
    let mut s = String::from("[");
    for (i, v) in l.iter().enumerate() {
        if i > 0 {
            s.push(',');
        }
        write!(s, "\"{:016x}\"", v).unwrap();
    }
    s.push(']');
    s
}
fn fr_json(x: &Fr) -> String {
    hex_limbs(&x.0 .0) // Montgomery limbs, little-endian u64
}
fn g1_json(p: &G1Affine) -> String {
    let mut l: Vec<u64> = p.x.0 .0.to_vec();
    l.extend_from_slice(&p.y.0 .0);
    format!("{{\"xy\":{},\"inf\":{}}}", hex_limbs(&l), if p.infinity { 1 } else { 0 })
}
fn bytes_hex(b: &[u8]) -> String {
    b.iter().map(|x| format!("{:02x}", x)).collect()
}
fn list(items: Vec<String>) -> String {
    format!("[{}]", items.join(","))
}

// What the reference's private ChallengeGenerator computes (plonk/src/proof/challenges.rs:17-45), issued against the same
// crates: transcript bytes = the commitments' serialize_unchecked forms back to back; seed = first eight bytes of their
// Blake2b-512 digest, little-endian; challenges = successive Fr::rand draws from StdRng::seed_from_u64(seed).
fn challenges(commitments: &[KzgCommitment], n: usize) -> Vec<Fr> {
    let mut transcript = Vec::with_capacity(96 * commitments.len());
    for c in commitments {
        c.inner().serialize_unchecked(&mut transcript).expect("serialising into a Vec cannot fail");
    }
    let digest = Blake2b::digest(&transcript);
    let seed = u64::from_le_bytes(digest[..8].try_into().expect("eight bytes"));
    let mut rng = StdRng::seed_from_u64(seed);
    std::iter::repeat_with(|| Fr::rand(&mut rng)).take(n).collect()
}

#[test]
fn emit_reference_vectors() {
    let g = G1Affine::prime_subgroup_generator();
    let mult = |k: u64| -> G1Affine { g.mul(Fr::from(k)).into_affine() };
    let mut out = String::from("{\n");

    // serialize_unchecked
    let mut ser = vec![];
    for p in [g, mult(2), G1Affine::zero()] {
        let mut b = vec![];
        p.serialize_unchecked(&mut b).unwrap();
        ser.push(format!("{{\"point\":{},\"bytes\":\"{}\"}}", g1_json(&p), bytes_hex(&b)));
    }
    writeln!(out, "\"serialize_unchecked\": {},", list(ser)).unwrap();

    // transcripts of 0..4 commitments (k G), two challenges each; one with the identity in it
    let mut tr = vec![];
    for k in 0..=4u64 {
        let pts: Vec<G1Affine> = (1..=k).map(mult).collect();
        let cs: Vec<KzgCommitment> = pts.iter().map(|p| KzgCommitment(*p)).collect();
        let ch = challenges(&cs, 2);
        tr.push(format!(
            "{{\"points\":{},\"challenges\":{}}}",
            list(pts.iter().map(g1_json).collect()),
            list(ch.iter().map(fr_json).collect())
        ));
    }
    {
        let pts = vec![g, G1Affine::zero(), mult(7)];
        let cs: Vec<KzgCommitment> = pts.iter().map(|p| KzgCommitment(*p)).collect();
        let ch = challenges(&cs, 3);
        tr.push(format!(
            "{{\"points\":{},\"challenges\":{}}}",
            list(pts.iter().map(g1_json).collect()),
            list(ch.iter().map(fr_json).collect())
        ));
    }
    writeln!(out, "\"transcripts\": {},", list(tr)).unwrap();

    // StdRng::seed_from_u64(1): raw words, then Fr::rand from a fresh generator
    let mut rng = StdRng::seed_from_u64(1);
    let words: Vec<u64> = (0..8).map(|_| rng.next_u64()).collect();
    writeln!(out, "\"stdrng_seed_1_next_u64\": {},", hex_limbs(&words)).unwrap();
    let mut rng = StdRng::seed_from_u64(1);
    let rands: Vec<String> = (0..4).map(|_| fr_json(&Fr::rand(&mut rng))).collect();
    writeln!(out, "\"fr_rand_seed_1\": {},", list(rands)).unwrap();

    // kzg: the crate's own test polynomial
    let srs = Srs::from_secret(Fr::from(2u64), 10);
    let scheme = KzgScheme::new(&srs);
    let poly = DensePolynomial::from_coefficients_slice(&[Fr::from(1u64), Fr::from(2u64), Fr::from(3u64)]);
    let c = scheme.commit(&poly);
    let o = scheme.open(poly.clone(), Fr::from(1u64));
    writeln!(
        out,
        "\"kzg_commit_1_2_3_s2\": {{\"commitment\":{},\"open_at_1\":{{\"witness\":{},\"eval\":{}}}}},",
        g1_json(c.inner()),
        g1_json(&o.0),
        fr_json(&o.1)
    )
    .unwrap();
    writeln!(out, "\"srs_s2\": {},", list(srs.g1_ref().iter().take(6).map(g1_json).collect())).unwrap();

    // evaluate_in_s with a large secret and full-width coefficients (drawn from seed 7)
    let secret = Fr::from(0x0123456789abcdef0123456789abcdefu128);
    let srs2 = Srs::from_secret(secret, 8);
    let mut rng = StdRng::seed_from_u64(7);
    let coeffs: Vec<Fr> = (0..8).map(|_| Fr::rand(&mut rng)).collect();
    let p8 = DensePolynomial::from_coefficients_vec(coeffs.clone());
    let c8 = KzgScheme::new(&srs2).commit(&p8);
    writeln!(
        out,
        "\"msm8\": {{\"secret\":{},\"coeffs\":{},\"commitment\":{}}},",
        fr_json(&secret),
        list(coeffs.iter().map(fr_json).collect()),
        g1_json(c8.inner())
    )
    .unwrap();

    // Srs::from_secret at a non-zero start, and one into_affine of a projective point whose Z is not 1
    let srs3 = Srs::from_secret(secret, 40);
    writeln!(out, "\"srs_slice\": {{\"secret\":{},\"start\":30,\"points\":{}}},", fr_json(&secret),
             list(srs3.g1_ref()[30..36].iter().map(g1_json).collect())).unwrap();
    let j = g.mul(Fr::from(0x1234567u64)).double() + g.mul(coeffs[0]);   // GroupProjective (Jacobian), Z != 1
    let ja = j.into_affine();
    writeln!(out, "\"into_affine\": {{\"x\":{},\"y\":{},\"z\":{},\"affine\":{}}},", hex_limbs(&j.x.0 .0), hex_limbs(&j.y.0 .0),
             hex_limbs(&j.z.0 .0), g1_json(&ja)).unwrap();

    // ark-poly radix-2 transforms, natural order in and out
    let dom = GeneralEvaluationDomain::<Fr>::new(4).unwrap();
    let v: Vec<Fr> = (1..=4u64).map(Fr::from).collect();
    let f = dom.fft(&v);
    let i = dom.ifft(&v);
    let dom8 = GeneralEvaluationDomain::<Fr>::new(8).unwrap();
    let f8 = dom8.fft(&coeffs);
    let c8f = dom8.coset_fft(&coeffs); // coset generator Fr::multiplicative_generator() = 7
    // a GROUP of interpolations, as the reference issues them (the three wire columns, plonk/src/proof.rs:50; the five
    // selector columns, plonk/src/builder.rs:84-88): three columns of eight values drawn after the msm8 coefficients from
    // the same generator, each through Evaluations::interpolate -- what ONE typlonk_ntt_fr_batch_devptr(inverse) call returns
    let cols: Vec<Vec<Fr>> = (0..3).map(|_| (0..8).map(|_| Fr::rand(&mut rng)).collect()).collect();
    let polys: Vec<Vec<Fr>> = cols
        .iter()
        .map(|c| {
            let mut co = ark_poly::Evaluations::from_vec_and_domain(c.clone(), dom8).interpolate().coeffs;
            co.resize(8, Fr::zero()); // ark-poly trims trailing zeros
            co
        })
        .collect();
    writeln!(
        out,
        "\"interpolate3\": {{\"columns\":{},\"polys\":{}}},",
        list(cols.iter().map(|c| list(c.iter().map(fr_json).collect())).collect()),
        list(polys.iter().map(|c| list(c.iter().map(fr_json).collect())).collect())
    )
    .unwrap();
    writeln!(
        out,
        "\"fft\": {{\"input\":{},\"fft4\":{},\"ifft4\":{},\"fft8_of_msm8_coeffs\":{},\"coset_fft8_of_msm8_coeffs\":{},\"group_gen_8\":{}}}",
        list(v.iter().map(fr_json).collect()),
        list(f.iter().map(fr_json).collect()),
        list(i.iter().map(fr_json).collect()),
        list(f8.iter().map(fr_json).collect()),
        list(c8f.iter().map(fr_json).collect()),
        fr_json(&dom8.element(1))
    )
    .unwrap();
    out.push_str("}\n");
    std::fs::write("reference_vectors.json", &out).unwrap();
    println!("{}", out);
    // canonical values for the curious
    println!("fft4 canonical: {:?}", f.iter().map(|x| x.into_repr().to_string()).collect::<Vec<_>>());
}
