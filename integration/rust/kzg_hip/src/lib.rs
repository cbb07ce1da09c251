//! Rust side of `libtyplonk_hip.so` -- the safe layer a TyPLONK maintainer calls from the `kzg` and `plonk` crates.
//!
//! SOURCE ONLY: this repository's image has no Rust toolchain, so nothing in this crate has been through `rustc`.
//! What IS tested is the C side every function here forwards to (`include/typlonk.h`, exercised through ctypes and
//! through the C++ mirror `typlonk_amd/host/typlonk_host.hpp`, which has the same shape as this file).
//!
//! Seams (file:line under the reference):
//!   * `Backend::msm`            replaces the body of `KzgScheme::evaluate_in_s`      kzg/src/lib.rs:41-54
//!   * `Backend::upload_srs`     once per `Srs` (`Srs::from_secret`)                  kzg/src/srs.rs:30-34
//!   * `Backend::interpolate` / `interpolate_batch` / `evaluate_over_domain`
//!                               replace `Evaluations::interpolate()` / `evaluate_over_domain()`
//!                               plonk/src/proof.rs:50,106,115,125,128  plonk/src/builder.rs:85
//!   * `Backend::load_circuit` + `Backend::prove`
//!                               replace the body of `plonk::proof::prove`           plonk/src/proof.rs:96-194
//!
//! Data crosses the boundary in arkworks' own in-memory form (`Fp256.0 .0`: 4 LE u64 limbs of the Montgomery residue;
//! `Fp384.0 .0`: 6), so nothing is converted -- coordinates are copied limb-wise because `GroupAffine` is `repr(Rust)`.
pub mod ffi;

use ark_bls12_381::{Fq, Fr, G1Affine};
use ark_ff::{BigInteger256, BigInteger384, Zero};
use ark_poly::{univariate::DensePolynomial, UVPolynomial};
use std::ffi::CStr;
use std::os::raw::c_int;
use std::ptr;

pub type G1Point = G1Affine;
pub type Poly = DensePolynomial<Fr>;

/// One HIP device + stream + workspaces (`typlonk_ctx`).  It holds a raw pointer, so it is neither `Send` nor `Sync`
/// -- and therefore neither is anything that embeds it (`kzg::srs::Srs`, `plonk::CompiledCircuit` after the patches):
/// one host thread per context, as typlonk.h requires.  A program that proves on several threads gives each thread its
/// own `Backend::shared` (the registry is thread-local) or wraps the calls in its own lock.
pub struct Backend {
    ctx: *mut ffi::TyplonkCtx,
}

thread_local! {
    // Backend::shared: one context per (thread, device ordinal), created on first use
    static SHARED: std::cell::RefCell<Vec<(i32, std::rc::Rc<Backend>)>> = std::cell::RefCell::new(Vec::new());
}

/// the device this process works on: `TYPLONK_DEVICE` (default 0).  One process per GPU: the launcher sets it per rank.
pub fn device_from_env() -> i32 {
    std::env::var("TYPLONK_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0)
}
/// fixed-base window tables for an uploaded SRS are opt-in (`TYPLONK_TABLES=1`): they multiply the SRS's footprint in
/// HBM by 13-17 and only pay when many commitments follow (a prover; not a one-off `commit`)
pub fn tables_from_env() -> bool {
    std::env::var("TYPLONK_TABLES").map(|v| v != "0" && !v.is_empty()).unwrap_or(false)
}

/// An SRS resident in HBM (`typlonk_srs_load`), with the fixed-base tables the library chooses for its length.
#[derive(Debug, Clone, Copy)]
pub struct SrsHandle {
    pub id: u32,
    pub len: usize,
}

/// Per-circuit constants of the quotient on the device (`typlonk_circuit_load`): selectors and sigmas on the 4n coset.
#[derive(Debug, Clone, Copy)]
pub struct CircuitHandle {
    pub id: u32,
    pub log_n: u32,
}

/// A device-resident vector of Fr (`typlonk_buf`), freed on drop.
pub struct DeviceVec<'a> {
    backend: &'a Backend,
    buf: *mut ffi::TyplonkBuf,
}

impl std::fmt::Debug for Backend {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "typlonk::Backend({:p})", self.ctx)
    }
}

fn fr_limbs(x: &Fr) -> [u64; 4] {
    (x.0).0
}
fn fr_from_limbs(l: [u64; 4]) -> Fr {
    // the limbs ARE the Montgomery residue: construct without a conversion (ark-ff 0.3: `Fp256::new(BigInteger256)`)
    Fr::new(BigInteger256(l))
}
fn g1_from_abi(xy: &[u64; 12], inf: u8) -> G1Point {
    let mut x = [0u64; 6];
    let mut y = [0u64; 6];
    x.copy_from_slice(&xy[0..6]);
    y.copy_from_slice(&xy[6..12]);
    // identity comes back as (0, 1, inf = 1) = GroupAffine::zero()
    G1Point::new(Fq::new(BigInteger384(x)), Fq::new(BigInteger384(y)), inf != 0)
}

impl Backend {
    /// `typlonk_init`: fails (panics, like the reference's `unwrap()`s) when there is no HIP device -- no CPU fallback.
    pub fn new(device_ordinal: i32) -> Self {
        let mut ctx = ptr::null_mut();
        let rc = unsafe { ffi::typlonk_init(&mut ctx, device_ordinal as c_int) };
        if rc != ffi::TYPLONK_OK {
            panic!("typlonk_init: {}", strerror(rc));
        }
        Backend { ctx }
    }

    /// negative code -> panic with the library's message: the reference panics at the same places
    /// (`TYPLONK_ERR_LENGTH` <=> `assert!(srs.len() > polynomial.degree())`, kzg/src/lib.rs:43)
    fn check(&self, rc: c_int) {
        if rc != ffi::TYPLONK_OK {
            let detail = unsafe { CStr::from_ptr(ffi::typlonk_last_error(self.ctx)) }.to_string_lossy().into_owned();
            panic!("{}: {}", strerror(rc), detail);
        }
    }

    /// The context of this thread for `device_ordinal`, created on first use and shared by every `Srs` /
    /// `CompiledCircuit` of the thread: `Srs::from_secret` must not open a context (streams, workspaces, an SRS upload)
    /// per call.
    pub fn shared(device_ordinal: i32) -> std::rc::Rc<Backend> {
        SHARED.with(|reg| {
            let mut reg = reg.borrow_mut();
            if let Some((_, b)) = reg.iter().find(|(d, _)| *d == device_ordinal) {
                return b.clone();
            }
            let b = std::rc::Rc::new(Backend::new(device_ordinal));
            reg.push((device_ordinal, b.clone()));
            b
        })
    }

    // ---- SRS ------------------------------------------------------------------------------------------------------
    /// once per `Srs` (kzg/src/srs.rs:30-34): copy the G1 powers to HBM; `tables`: also build the fixed-base window
    /// tables (`precompute_tables`)
    pub fn upload_srs(&self, g1: &[G1Point], tables: bool) -> SrsHandle {
        let mut xy = Vec::with_capacity(g1.len() * 12);
        let mut inf = Vec::with_capacity(g1.len());
        for p in g1 {
            xy.extend_from_slice(&(p.x.0).0);
            xy.extend_from_slice(&(p.y.0).0);
            inf.push(p.infinity as u8);
        }
        let mut id = 0u32;
        self.check(unsafe { ffi::typlonk_srs_load(self.ctx, xy.as_ptr(), inf.as_ptr(), g1.len(), &mut id) });
        let h = SrsHandle { id, len: g1.len() };
        if tables {
            self.precompute_tables(h);
        }
        h
    }

    /// `typlonk_srs_free`: the device copy of an SRS and its fixed-base tables (13-17 x the SRS when built).  Called by
    /// `OwnedSrs::drop`; a handle must not be used afterwards.
    pub fn free_srs(&self, srs: SrsHandle) {
        self.check(unsafe { ffi::typlonk_srs_free(self.ctx, srs.id) });
    }
    /// `typlonk_circuit_free`: the per-circuit coset evaluations (9 x 4n + 11 x n Fr).  Called by `OwnedCircuit::drop`.
    pub fn free_circuit(&self, circuit: CircuitHandle) {
        self.check(unsafe { ffi::typlonk_circuit_free(self.ctx, circuit.id) });
    }

    /// `typlonk_srs_precompute`: speed only, results unchanged; the window is chosen by length (nothing at all below
    /// TYPLONK_TABLES_AUTO_MIN_LEN points).  13-17 copies of the SRS in HBM.
    pub fn precompute_tables(&self, srs: SrsHandle) {
        self.check(unsafe { ffi::typlonk_srs_precompute(self.ctx, srs.id, 0) });
    }

    /// `Srs::from_secret` on the device: `[s^(start + i)] G`, i < len (each GPU of a node builds only its shard)
    pub fn generate_srs(&self, secret: &Fr, start: u64, len: usize, tables: bool) -> SrsHandle {
        let s = fr_limbs(secret);
        let mut id = 0u32;
        self.check(unsafe { ffi::typlonk_srs_generate(self.ctx, s.as_ptr(), start, len, &mut id) });
        let h = SrsHandle { id, len };
        if tables {
            self.precompute_tables(h);
        }
        h
    }

    // ---- MSM seam: the body of KzgScheme::evaluate_in_s (kzg/src/lib.rs:41-54) -------------------------------------
    /// sum_i coeffs[i] * srs[i].  `coeffs` are the polynomial's coefficients with ark-poly's trailing-zero trim, so the
    /// MSM length is what the reference's `zip` would have consumed.
    pub fn msm(&self, srs: SrsHandle, coeffs: &[Fr]) -> G1Point {
        let mut scalars = Vec::with_capacity(coeffs.len() * 4);
        for c in coeffs {
            scalars.extend_from_slice(&fr_limbs(c));
        }
        let (mut xy, mut inf) = ([0u64; 12], 0u8);
        self.check(unsafe {
            ffi::typlonk_msm_g1(self.ctx, srs.id, scalars.as_ptr(), coeffs.len(), xy.as_mut_ptr(), &mut inf)
        });
        g1_from_abi(&xy, inf)
    }

    /// the same with the coefficients already in HBM (an iNTT result feeding a commitment: proof.rs:50 -> :109)
    pub fn msm_dev(&self, srs: SrsHandle, v: &DeviceVec, offset: usize, m: usize) -> G1Point {
        let (mut xy, mut inf) = ([0u64; 12], 0u8);
        self.check(unsafe { ffi::typlonk_msm_g1_dev(self.ctx, srs.id, v.buf, offset, m, xy.as_mut_ptr(), &mut inf) });
        g1_from_abi(&xy, inf)
    }

    // ---- NTT seam ---------------------------------------------------------------------------------------------------
    /// `Evaluations::from_vec_and_domain(evals, domain).interpolate()`: ifft, then ark-poly's trailing-zero trim
    /// (`from_coefficients_vec`), which is what sets MSM lengths downstream
    pub fn interpolate(&self, mut evals: Vec<Fr>, log_n: u32) -> Poly {
        assert_eq!(evals.len(), 1usize << log_n);
        let mut limbs: Vec<u64> = evals.iter().flat_map(|e| fr_limbs(e)).collect();
        self.check(unsafe { ffi::typlonk_ntt_fr(self.ctx, limbs.as_mut_ptr(), log_n, 1, ptr::null()) });
        for (e, l) in evals.iter_mut().zip(limbs.chunks_exact(4)) {
            *e = fr_from_limbs([l[0], l[1], l[2], l[3]]);
        }
        Poly::from_coefficients_vec(evals)
    }

    /// A GROUP of interpolations in one call -- the three wire columns (plonk/src/proof.rs:50), the three sigma columns
    /// (:334-338), the five selector columns (plonk/src/builder.rs:84-88): one upload, ONE `typlonk_ntt_fr_batch_devptr`
    /// (every pass of the transform is a single launch carrying all the columns), one download; each result gets ark-poly's
    /// trailing-zero trim like `interpolate`.
    pub fn interpolate_batch(&self, columns: Vec<Vec<Fr>>, log_n: u32) -> Vec<Poly> {
        let n = 1usize << log_n;
        let count = columns.len();
        if count == 0 {
            return Vec::new();
        }
        let mut limbs: Vec<u64> = Vec::with_capacity(4 * n * count);
        for col in &columns {
            assert_eq!(col.len(), n);
            limbs.extend(col.iter().flat_map(|e| fr_limbs(e)));
        }
        let mut buf = ptr::null_mut();
        self.check(unsafe { ffi::typlonk_buf_alloc(self.ctx, n * count, &mut buf) });
        let dev = DeviceVec { backend: self, buf }; // freed on drop, also when a check below panics
        self.check(unsafe { ffi::typlonk_buf_upload(self.ctx, dev.buf, 0, limbs.as_ptr(), n * count) });
        let base = unsafe { ffi::typlonk_buf_devptr(dev.buf) } as *mut u8;
        let ptrs: Vec<*mut std::os::raw::c_void> =
            (0..count).map(|v| unsafe { base.add(32 * n * v) } as *mut std::os::raw::c_void).collect();
        self.check(unsafe { ffi::typlonk_ntt_fr_batch_devptr(self.ctx, ptrs.as_ptr(), count, log_n, 1, ptr::null()) });
        self.check(unsafe { ffi::typlonk_buf_download(self.ctx, dev.buf, 0, limbs.as_mut_ptr(), n * count) });
        limbs
            .chunks_exact(4 * n)
            .map(|col| Poly::from_coefficients_vec(col.chunks_exact(4).map(|l| fr_from_limbs([l[0], l[1], l[2], l[3]])).collect()))
            .collect()
    }

    /// `poly.evaluate_over_domain(domain).evals`: zero-pad to the domain size, fft, natural order
    pub fn evaluate_over_domain(&self, poly: &Poly, log_n: u32) -> Vec<Fr> {
        let n = 1usize << log_n;
        assert!(poly.coeffs.len() <= n);
        let mut limbs = vec![0u64; 4 * n];
        for (c, l) in poly.coeffs.iter().zip(limbs.chunks_exact_mut(4)) {
            l.copy_from_slice(&fr_limbs(c));
        }
        self.check(unsafe { ffi::typlonk_ntt_fr(self.ctx, limbs.as_mut_ptr(), log_n, 0, ptr::null()) });
        limbs.chunks_exact(4).map(|l| fr_from_limbs([l[0], l[1], l[2], l[3]])).collect()
    }

    // ---- device vectors ---------------------------------------------------------------------------------------------
    pub fn upload(&self, v: &[Fr], capacity: usize) -> DeviceVec<'_> {
        let mut buf = ptr::null_mut();
        self.check(unsafe { ffi::typlonk_buf_alloc(self.ctx, capacity.max(v.len()), &mut buf) });
        let d = DeviceVec { backend: self, buf };
        self.check(unsafe { ffi::typlonk_buf_zero(self.ctx, buf, 0, capacity.max(v.len())) });
        let limbs: Vec<u64> = v.iter().flat_map(|e| fr_limbs(e)).collect();
        self.check(unsafe { ffi::typlonk_buf_upload(self.ctx, buf, 0, limbs.as_ptr(), v.len()) });
        d
    }

    // ---- whole prover ---------------------------------------------------------------------------------------------
    /// once per `CompiledCircuit` (plonk/src/lib.rs:19-35): the five selector polynomials (coefficients, zero-padded
    /// to n) and the three sigma polynomials (`interpolate` of `copy_constrains.cols[i]`'s second components)
    pub fn load_circuit(&self, selectors: [&Poly; 5], sigma: [&Poly; 3], log_n: u32) -> CircuitHandle {
        let n = 1usize << log_n;
        let sel: Vec<DeviceVec> = selectors.iter().map(|p| self.upload(&p.coeffs, n)).collect();
        let sig: Vec<DeviceVec> = sigma.iter().map(|p| self.upload(&p.coeffs, n)).collect();
        let selp: Vec<*const ffi::TyplonkBuf> = sel.iter().map(|d| d.buf as *const _).collect();
        let sigp: Vec<*const ffi::TyplonkBuf> = sig.iter().map(|d| d.buf as *const _).collect();
        let mut id = 0u32;
        self.check(unsafe { ffi::typlonk_circuit_load(self.ctx, selp.as_ptr(), sigp.as_ptr(), log_n, &mut id) });
        CircuitHandle { id, log_n }
    }

    /// `plonk::proof::prove` (plonk/src/proof.rs:96-194) in one native call: `wire_evals` are the three padded and
    /// blinded witness COLUMNS (what `CompiledCircuit::prove` builds at :43-49, before `.interpolate()`),
    /// `public_inputs` the padded public-input column (:52-53), `cosets` = `copy_constrains.cosets`.
    /// Panics with "r(zeta) != 0" where the reference panics in `vanishes()` (:321, :361).
    pub fn prove(&self, srs: SrsHandle, circuit: CircuitHandle, wire_evals: [&[Fr]; 3], public_inputs: &[Fr],
                 cosets: [Fr; 3]) -> ffi::TyplonkProof {
        // The columns go over as they are, in host memory (`typlonk_prove_host`, round 6): the library uploads each one right
        // before its interpolation and commitment are queued, so column i + 1 crosses PCIe while column i is transformed,
        // sorted and accumulated -- instead of four uploads (128 MiB at 2^20) before the first kernel starts.
        let n = 1usize << circuit.log_n;
        let flat = |c: &[Fr]| -> Vec<u64> {
            assert_eq!(c.len(), n);
            c.iter().flat_map(|e| fr_limbs(e)).collect()
        };
        let w: Vec<Vec<u64>> = wire_evals.iter().map(|c| flat(c)).collect();
        let wp: [*const u64; 3] = [w[0].as_ptr(), w[1].as_ptr(), w[2].as_ptr()];
        let pi = if public_inputs.iter().all(|x| x.is_zero()) { None } else { Some(flat(public_inputs)) };
        let k = [fr_limbs(&cosets[0]), fr_limbs(&cosets[1]), fr_limbs(&cosets[2])];
        let mut out = std::mem::MaybeUninit::<ffi::TyplonkProof>::zeroed();
        self.check(unsafe {
            ffi::typlonk_prove_host(self.ctx, srs.id, circuit.id, wp.as_ptr(), pi.as_ref().map_or(ptr::null(), |v| v.as_ptr()),
                                    k.as_ptr(), out.as_mut_ptr())
        });
        unsafe { out.assume_init() }
    }

    // ---- multi-GPU: one process per GPU, RCCL inside the library --------------------------------------------------
    /// this process holds bases [first, first + len) of a `total`-point SRS
    pub fn set_shard(&self, srs: SrsHandle, first: usize, total: usize) {
        self.check(unsafe { ffi::typlonk_srs_set_shard(self.ctx, srs.id, first, total) });
    }
    /// rank 0: the 128-byte rendezvous id, to be handed to the other ranks by any channel the host program has
    pub fn comm_unique_id() -> [u8; ffi::TYPLONK_COMM_ID_BYTES] {
        let mut id = [0u8; ffi::TYPLONK_COMM_ID_BYTES];
        let rc = unsafe { ffi::typlonk_comm_unique_id(id.as_mut_ptr()) };
        if rc != ffi::TYPLONK_OK {
            panic!("typlonk_comm_unique_id: {}", strerror(rc));
        }
        id
    }
    /// every rank (collective).  Call `typlonk_comm_available()` on every rank first and agree on the answer.
    pub fn comm_init(&self, id: &[u8; ffi::TYPLONK_COMM_ID_BYTES], rank: i32, world: i32) {
        self.check(unsafe { ffi::typlonk_comm_init(self.ctx, id.as_ptr(), rank, world) });
    }
    /// `evaluate_in_s` over the whole node: every rank passes the device address of coefficient 0 of the full vector
    /// and gets the FULL sum (local partial MSM + one all-gather + fold in rank order)
    pub fn msm_sharded(&self, srs: SrsHandle, v: &DeviceVec, m: usize) -> G1Point {
        let (mut xy, mut inf) = ([0u64; 12], 0u8);
        let p = unsafe { ffi::typlonk_buf_devptr(v.buf) };
        self.check(unsafe { ffi::typlonk_msm_g1_sharded_devptr(self.ctx, srs.id, p, m, xy.as_mut_ptr(), &mut inf) });
        g1_from_abi(&xy, inf)
    }
}

/// `typlonk_proof` -> the pieces of the reference's `Proof` (plonk/src/proof.rs:65-95)
pub struct ProofParts {
    pub commitments: [G1Point; 3],          // [a], [b], [c]
    pub z_commitment: G1Point,               // permutation.commitment
    pub t: [G1Point; 3],                     // quotient slices
    pub witnesses: [G1Point; 6],             // a, b, c at zeta; Z at zeta; Z at zeta*w; r at zeta
    pub evals: [Fr; 6],                      // a(zeta) b(zeta) c(zeta) Z(zeta) Z(zeta w) r(zeta)
    pub evaluation_point: Fr,
}
impl From<&ffi::TyplonkProof> for ProofParts {
    fn from(p: &ffi::TyplonkProof) -> Self {
        let g = |xy: &[u64; 12], inf: u8| g1_from_abi(xy, inf);
        ProofParts {
            commitments: [0, 1, 2].map(|i| g(&p.commit_xy[i], p.commit_inf[i])),
            z_commitment: g(&p.z_xy, p.z_inf),
            t: [0, 1, 2].map(|i| g(&p.tail.t_xy[i], p.tail.t_inf[i])),
            witnesses: [0, 1, 2, 3, 4, 5].map(|i| g(&p.tail.w_xy[i], p.tail.w_inf[i])),
            evals: [0, 1, 2, 3, 4, 5].map(|i| fr_from_limbs(p.tail.evals[i])),
            evaluation_point: fr_from_limbs(p.zeta),
        }
    }
}

/// An SRS resident on the shared context of its thread, freed when dropped: what the patched `kzg::srs::Srs` holds.
/// (`Backend::shared` keeps ONE context per thread and device alive for the thread's lifetime, so the device memory of
/// an `Srs` must be returned when the `Srs` goes -- not when the context does.)
#[derive(Debug)]
pub struct OwnedSrs {
    backend: std::rc::Rc<Backend>,
    handle: SrsHandle,
}
impl OwnedSrs {
    pub fn new(backend: std::rc::Rc<Backend>, handle: SrsHandle) -> Self {
        OwnedSrs { backend, handle }
    }
    pub fn backend(&self) -> &Backend {
        &self.backend
    }
    pub fn backend_rc(&self) -> std::rc::Rc<Backend> {
        self.backend.clone()
    }
    pub fn handle(&self) -> SrsHandle {
        self.handle
    }
}
impl Drop for OwnedSrs {
    fn drop(&mut self) {
        unsafe { ffi::typlonk_srs_free(self.backend.ctx, self.handle.id) }; // (no panic in drop: the status is ignored)
    }
}
/// The per-circuit device constants (`typlonk_circuit_load`), freed when dropped: what the patched `CompiledCircuit` holds.
#[derive(Debug)]
pub struct OwnedCircuit {
    backend: std::rc::Rc<Backend>,
    handle: CircuitHandle,
}
impl OwnedCircuit {
    pub fn new(backend: std::rc::Rc<Backend>, handle: CircuitHandle) -> Self {
        OwnedCircuit { backend, handle }
    }
    pub fn handle(&self) -> CircuitHandle {
        self.handle
    }
}
impl Drop for OwnedCircuit {
    fn drop(&mut self) {
        unsafe { ffi::typlonk_circuit_free(self.backend.ctx, self.handle.id) };
    }
}

impl Drop for Backend {
    fn drop(&mut self) {
        unsafe { ffi::typlonk_destroy(self.ctx) }
    }
}
impl Drop for DeviceVec<'_> {
    fn drop(&mut self) {
        unsafe { ffi::typlonk_buf_free(self.backend.ctx, self.buf) };
    }
}

fn strerror(rc: c_int) -> String {
    unsafe { CStr::from_ptr(ffi::typlonk_strerror(rc)) }.to_string_lossy().into_owned()
}
