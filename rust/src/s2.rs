//! Seam S2 (Marlin / GM17 / anything that calls ark-ec and ark-poly directly: /root/reference tests/mnt4_marlin.rs:72-75,
//! tests/mnt4_gm17.rs): the CONCRETE side of the two type-erased hooks the `ark-ec` / `ark-poly` forks expose
//! (`rust/s2_patch/ark-ec/src/msm/hook.rs`, `rust/s2_patch/ark-poly/src/domain/hook.rs`).
//!
//! Dependency direction (checked by tools/check_rust_boundary.py): the forks know no curve crate; THIS crate depends on the forks
//! and on the four curve crates, recognises the eight affine types / four scalar fields by `TypeId`, marshals through their public
//! fields and calls libpcdhip.so.  `install()` registers both hooks; a program that wants the S2 seam calls it once before proving
//! (first line of the test's `main`, or of `ark-pcd`'s `universal_setup` caller).  Feature `s2` (needs the forks via
//! `[patch."https://github.com/arkworks-rs/algebra"]`); without it this module is not compiled and the crate builds against
//! unmodified upstream.
//!
//! Source only (no Rust toolchain in the build container); the C-ABI call sequence made here -- ONE upload of a base vector, MSMs
//! over prefixes of it, in-place transforms of host vectors -- is what tests/c_driver/driver.c runs from plain C against the oracle.
use crate::ffi;
use crate::prover::{with_device, Device};
use crate::HipCurve;
use ark_ec::AffineCurve;
use ark_ff::PrimeField;
use core::any::TypeId;
use std::os::raw::c_int;

/// below this many pairs the PCIe round trip costs more than the CPU
pub const MIN_PAIRS: usize = 1 << 12;
/// below this size a transform stays on the CPU (the vector crosses PCIe both ways)
pub const MIN_LOG_N: u32 = 14;
/// resident base vectors kept at most (least recently used goes first; its device memory is released with `pcdhip_bases_free`)
pub const MAX_RESIDENT: usize = 16;

/// device memory one resident base vector may take with its window-shifted copies (`pcdhip_set_precompute_budget`): the S2 vectors share
/// the one device context with the Groth16 keys, and the library's default (as many copies as hipMalloc grants: 15 x the vector at 2^20)
/// would let the first large KZG vector take most of HBM.  Fewer copies = a Horner combine at the end, same result.
pub const RESIDENT_BUDGET_BYTES: usize = 8 << 30;
/// positions of a resident vector whose packed limbs are kept for the confirmation of a digest hit
const SAMPLE_HEAD: usize = 64;
const SAMPLE_STRIDED: usize = 192;

/// A base vector resident on the device.  `running[i]` = digest of the points 0 ..= i, so a call on a PREFIX of a resident vector
/// (KZG: `powers_of_g[..deg + 1]`) is recognised by content and served as `pcdhip_msm(handle, 0, .., n)`.  The digest (two SipHash lanes
/// over every limb and flag, keyed with per-process random keys: `Device::digest_keys` -- ADVICE r05: the unkeyed FNV-1a of round 5 is not
/// collision resistant, and a false hit runs the MSM over the wrong bases) NOMINATES a candidate; a hit is confirmed on a SAMPLE of the points -- the first 64 and 192 spread over
/// the vector, their packed limbs kept here -- instead of a second full copy on the host (round 4 kept up to sixteen of them, 0.6 GB
/// each for a 2^20-point G2 vector, and compared all n points on every call: ADVICE r04).  Two different vectors that agree on both
/// digest lanes at the call's length AND on every sampled point do not occur by accident; an adversary who controls the bases controls
/// the proving key anyway.
pub(crate) struct Resident {
    curve: u32, group: c_int, n: usize, words: usize,
    running: Vec<[u64; 2]>, sample_at: Vec<usize>, sample_xy: Vec<u64>, sample_inf: Vec<u8>,
    handle: *mut ffi::pcdhip_bases, last_use: u64,
}

fn sample_positions(n: usize) -> Vec<usize> {
    let mut at: Vec<usize> = (0..n.min(SAMPLE_HEAD)).collect();
    if n > SAMPLE_HEAD {
        let step = ((n - SAMPLE_HEAD) / SAMPLE_STRIDED).max(1);
        let mut i = SAMPLE_HEAD;
        while i < n && at.len() < SAMPLE_HEAD + SAMPLE_STRIDED { at.push(i); i += step; }
    }
    at
}

/// every fallback to the upstream CPU code is said once per cause (a silent fallback looks like a slow GPU)
fn log_fallback(cause: &'static str) {
    use std::sync::Mutex;
    static SEEN: Mutex<Vec<&'static str>> = Mutex::new(Vec::new());
    if let Ok(mut seen) = SEEN.lock() {
        if !seen.contains(&cause) { seen.push(cause); eprintln!("ark-pcd-hip: MSM falls back to the CPU: {}", cause); }
    }
}

/// two keyed SipHash lanes (std's `RandomState`: random 128-bit keys drawn once per process and unknown to whoever supplies the bases) over
/// EVERY limb and flag, point after point; `finish()` does not consume the state, so entry i is the digest of the points 0 ..= i.
/// Both lanes see the whole stream (independent keys), so a false hit needs a simultaneous collision of two keyed 64-bit PRFs.
fn running_digest(keys: &[std::collections::hash_map::RandomState; 2], xy: &[u64], inf: &[u8], words: usize) -> Vec<[u64; 2]> {
    use std::hash::{BuildHasher, Hasher};
    let mut h = [keys[0].build_hasher(), keys[1].build_hasher()];
    let mut out = Vec::with_capacity(inf.len());
    for (i, flag) in inf.iter().enumerate() {
        for w in &xy[i * words..(i + 1) * words] {
            h[0].write_u64(*w);
            h[1].write_u64(*w);
        }
        h[0].write_u8(*flag);
        h[1].write_u8(*flag);
        out.push([h[0].finish(), h[1].finish()]);
    }
    out
}

/// `pcdhip_msm` over (a prefix of) a resident vector; uploads the vector on first sight.  `None` = let upstream run.
fn msm_packed(dev: &mut Device, curve: u32, group: c_int, words: usize, xy: Vec<u64>, inf: Vec<u8>, scalars: &[u64], out_words: usize) -> Option<Vec<u64>> {
    let n = inf.len();
    let running = running_digest(&dev.digest_keys, &xy, &inf, words);
    let d = running[n - 1];
    dev.clock += 1;
    let now = dev.clock;
    let hit = dev.bases.iter().position(|r| {
        r.curve == curve && r.group == group && r.n >= n && r.running[n - 1] == d
            && r.sample_at.iter().enumerate().all(|(k, &i)| {
                i >= n || (r.sample_xy[k * words..(k + 1) * words] == xy[i * words..(i + 1) * words] && r.sample_inf[k] == inf[i])
            })
    });
    let idx = match hit {
        Some(i) => i,
        None => {
            let evict = |dev: &mut Device| -> bool {
                match dev.bases.iter().enumerate().min_by_key(|(_, r)| r.last_use).map(|(i, _)| i) {
                    Some(victim) => { let old = dev.bases.swap_remove(victim); unsafe { ffi::pcdhip_bases_free(dev.ctx, old.handle) }; true }
                    None => false,
                }
            };
            if dev.bases.len() >= MAX_RESIDENT { evict(dev); }
            // (the budget is a property of the context and read at upload time: set for this upload and put back afterwards -- ADVICE r05:
            //  writing 0 here overwrote whatever budget the host had configured on the shared context)
            let mut saved_budget: usize = 0;
            unsafe { ffi::pcdhip_get_precompute_budget(dev.ctx, &mut saved_budget) };
            unsafe { ffi::pcdhip_set_precompute_budget(dev.ctx, if saved_budget != 0 { saved_budget.min(RESIDENT_BUDGET_BYTES) } else { RESIDENT_BUDGET_BYTES }) };
            let mut h = core::ptr::null_mut();
            let mut rc = unsafe { ffi::pcdhip_bases_upload(dev.ctx, curve as c_int, group, xy.as_ptr(), inf.as_ptr(), n, &mut h) };
            if rc != 0 && evict(dev) {   // out of memory, most likely: let the least recently used vector go and try once more
                rc = unsafe { ffi::pcdhip_bases_upload(dev.ctx, curve as c_int, group, xy.as_ptr(), inf.as_ptr(), n, &mut h) };
            }
            unsafe { ffi::pcdhip_set_precompute_budget(dev.ctx, saved_budget) };
            if rc != 0 { log_fallback("pcdhip_bases_upload failed (device memory?)"); return None; }
            let sample_at = sample_positions(n);
            let mut sample_xy = Vec::with_capacity(sample_at.len() * words);
            let mut sample_inf = Vec::with_capacity(sample_at.len());
            for &i in &sample_at { sample_xy.extend_from_slice(&xy[i * words..(i + 1) * words]); sample_inf.push(inf[i]); }
            dev.bases.push(Resident { curve, group, n, words, running, sample_at, sample_xy, sample_inf, handle: h, last_use: now });
            dev.bases.len() - 1
        }
    };
    dev.bases[idx].last_use = now;
    debug_assert_eq!(dev.bases[idx].words, words);
    let mut out = vec![0u64; out_words];
    if unsafe { ffi::pcdhip_msm(dev.ctx, dev.bases[idx].handle, 0, scalars.as_ptr(), n, out.as_mut_ptr()) } != 0 {
        log_fallback("pcdhip_msm failed");
        return None;
    }
    Some(out)
}

fn flat_scalars<B: AsRef<[u64]>>(s: &[B], n: usize, limbs: usize) -> Vec<u64> {
    let mut sc = Vec::with_capacity(n * limbs);
    for b in &s[..n] { sc.extend_from_slice(b.as_ref()); }   // `BigInteger`s: canonical limbs, what upstream passes and pcdhip_msm takes
    sc
}

fn run_g1<E: HipCurve>(bases: &[E::G1Affine], scalars: &[<E::Fr as PrimeField>::BigInt]) -> Option<E::G1Projective> {
    let n = bases.len().min(scalars.len());
    if n < MIN_PAIRS { return None; }
    let (mut xy, mut inf) = (Vec::with_capacity(n * 2 * E::FQ_LIMBS), Vec::with_capacity(n));
    for p in &bases[..n] { E::push_g1(p, &mut xy, &mut inf); }
    let sc = flat_scalars(scalars, n, E::FQ_LIMBS);   // Fr and Fq of a cycle curve have the same limb count
    let out = with_device(|dev| Ok(msm_packed(dev, E::CURVE_ID, 1, 2 * E::FQ_LIMBS, xy, inf, &sc, 3 * E::FQ_LIMBS))).ok()??;
    Some(E::g1_projective(&out))
}
fn run_g2<E: HipCurve>(bases: &[E::G2Affine], scalars: &[<E::Fr as PrimeField>::BigInt]) -> Option<E::G2Projective> {
    let n = bases.len().min(scalars.len());
    if n < MIN_PAIRS { return None; }
    let w = 2 * E::G2_DEG * E::FQ_LIMBS;
    let (mut xy, mut inf) = (Vec::with_capacity(n * w), Vec::with_capacity(n));
    for p in &bases[..n] { E::push_g2(p, &mut xy, &mut inf); }
    let sc = flat_scalars(scalars, n, E::FQ_LIMBS);
    let out = with_device(|dev| Ok(msm_packed(dev, E::CURVE_ID, 2, w, xy, inf, &sc, 3 * E::G2_DEG * E::FQ_LIMBS))).ok()??;
    Some(E::g2_projective(&out))
}

/// The registered `ark_ec::msm::hook::MsmHook`.  `affine` says which of the eight types the erased pointers are.
unsafe fn msm_hook(affine: TypeId, bases: *const u8, n_bases: usize, scalars: *const u8, n_scalars: usize, out: *mut u8) -> bool {
    macro_rules! route {
        ($engine:ty, $aff:ty, $run:ident) => {
            if affine == TypeId::of::<$aff>() {
                // sound: `affine` is the TypeId of the slice element at the (only) call site, ark_ec::msm::hook::try_hook::<G>
                let b = core::slice::from_raw_parts(bases as *const $aff, n_bases);
                let s = core::slice::from_raw_parts(scalars as *const <<$aff as AffineCurve>::ScalarField as PrimeField>::BigInt, n_scalars);
                return match $run::<$engine>(b, s) {
                    Some(r) => { core::ptr::write(out as *mut <$aff as AffineCurve>::Projective, r); true }
                    None => false,
                };
            }
        };
    }
    route!(ark_mnt4_298::MNT4_298, ark_mnt4_298::G1Affine, run_g1); route!(ark_mnt4_298::MNT4_298, ark_mnt4_298::G2Affine, run_g2);
    route!(ark_mnt6_298::MNT6_298, ark_mnt6_298::G1Affine, run_g1); route!(ark_mnt6_298::MNT6_298, ark_mnt6_298::G2Affine, run_g2);
    route!(ark_mnt4_753::MNT4_753, ark_mnt4_753::G1Affine, run_g1); route!(ark_mnt4_753::MNT4_753, ark_mnt4_753::G2Affine, run_g2);
    route!(ark_mnt6_753::MNT6_753, ark_mnt6_753::G1Affine, run_g1); route!(ark_mnt6_753::MNT6_753, ark_mnt6_753::G2Affine, run_g2);
    false   // any other curve (the CRH's ed_on_mnt4_298, ...): upstream
}

/// `field_id` of include/pcdhip.h for the four supported scalar / base fields
fn field_of(t: TypeId) -> Option<(c_int, usize)> {
    if t == TypeId::of::<ark_mnt4_298::Fq>() { return Some((0, 5)); }    // = MNT6-298 Fr
    if t == TypeId::of::<ark_mnt4_298::Fr>() { return Some((1, 5)); }    // = MNT6-298 Fq
    if t == TypeId::of::<ark_mnt4_753::Fq>() { return Some((2, 12)); }   // = MNT6-753 Fr
    if t == TypeId::of::<ark_mnt4_753::Fr>() { return Some((3, 12)); }   // = MNT6-753 Fq
    None
}

/// The registered `ark_poly::domain::hook::FftHook`: in place on the vector's own storage (an `Fp320` / `Fp768` is its Montgomery
/// `BigInteger` limbs in memory, the C-ABI's encoding -- `marshal::limbs_of` asserts the size).
unsafe fn fft_hook(field: TypeId, data: *mut u8, len: usize, log_n: u32, inverse: bool) -> bool {
    let (field_id, _limbs) = match field_of(field) { Some(f) => f, None => return false };
    if log_n < MIN_LOG_N || len != 1usize << log_n || (data as usize) % core::mem::align_of::<u64>() != 0 { return false; }
    with_device(|dev| ffi::check(ffi::pcdhip_fft(dev.ctx, field_id, data as *mut u64, log_n, inverse as c_int, 0))).is_ok()
}

/// A chain of transforms on ONE vector with a single trip over PCIe (`pcdhip_fft_seq`): what `R1CSToQAP::witness_map` (`ifft` then
/// `coset_fft`) and Marlin's AHP rounds do with a polynomial.  The per-call hook above moves 2 x n x 40 / 96 bytes per transform --
/// 5 .. 7 ms at n = 2^20 for 0.3 ms of passes (bench.py `fft.*.call_wall_ms`) -- so a caller that knows its chain hands it over whole.
/// `ops`: (inverse, coset) per transform, all over the `GeneralEvaluationDomain` of `data.len()` elements.  false: run upstream.
pub fn fft_chain<F: PrimeField + 'static>(data: &mut [F], ops: &[(bool, bool)]) -> bool {
    let (field_id, limbs) = match field_of(TypeId::of::<F>()) { Some(f) => f, None => return false };
    if data.len() < (1usize << MIN_LOG_N) || ops.is_empty() || ops.len() > 64 || core::mem::size_of::<F>() != limbs * 8 { return false; }
    let codes: Vec<c_int> = ops.iter().map(|&(inv, coset)| (inv as c_int) | ((coset as c_int) << 1)).collect();
    let (ptr, n) = (data.as_mut_ptr() as *mut u64, data.len());
    with_device(|dev| unsafe { ffi::check(ffi::pcdhip_fft_seq(dev.ctx, field_id, ptr, n, codes.as_ptr(), codes.len() as c_int)) }).is_ok()
}

/// Register both hooks with the forks.  Idempotent; returns whether THIS call installed them (false: someone else already had).
pub fn install() -> bool {
    let a = ark_ec::msm::hook::set_msm_hook(msm_hook);
    let b = ark_poly::domain::hook::set_fft_hook(fft_hook);
    a && b
}
