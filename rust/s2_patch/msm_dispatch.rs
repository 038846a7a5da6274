//! ark-ec fork, `src/msm_dispatch.rs`: route `VariableBaseMSM::multi_scalar_mul` for the eight groups libpcdhip.so supports to
//! `pcdhip_msm`, keyed by the concrete affine type; `None` = run the upstream CPU code.  Bases are uploaded once per distinct
//! slice (address + length + a digest of the first and last point) and kept resident: a KZG committer key is one vector that
//! every commitment indexes by prefix (`powers_of_g[..deg + 1]`), which is exactly `pcdhip_msm(bases, offset, .., n)`.
use crate::AffineCurve;
use ark_ff::PrimeField;
use core::any::TypeId;
use std::os::raw::c_int;

extern "C" {
    fn pcdhip_init(device_id: c_int, out: *mut *mut u8) -> c_int;
    fn pcdhip_bases_upload(ctx: *mut u8, curve: c_int, group: c_int, xy: *const u64, inf: *const u8, n: usize, out: *mut *mut u8) -> c_int;
    fn pcdhip_msm(ctx: *mut u8, bases: *const u8, offset: usize, scalars: *const u64, n: usize, out_xyz: *mut u64) -> c_int;
}

/// below this many pairs the PCIe round trip costs more than the CPU
const MIN_PAIRS: usize = 1 << 12;

/// (curve_id, group_id, u64 limbs of one base-field element, base-field coefficients per coordinate) of a supported group
fn group_of<G: AffineCurve>() -> Option<(c_int, c_int, usize, usize)> {
    let t = TypeId::of::<G>();
    macro_rules! row { ($ty:ty, $c:expr, $g:expr, $l:expr, $d:expr) => { if t == TypeId::of::<$ty>() { return Some(($c, $g, $l, $d)); } }; }
    row!(ark_mnt4_298::G1Affine, 0, 1, 5, 1); row!(ark_mnt4_298::G2Affine, 0, 2, 5, 2);
    row!(ark_mnt6_298::G1Affine, 1, 1, 5, 1); row!(ark_mnt6_298::G2Affine, 1, 2, 5, 3);
    row!(ark_mnt4_753::G1Affine, 2, 1, 12, 1); row!(ark_mnt4_753::G2Affine, 2, 2, 12, 2);
    row!(ark_mnt6_753::G1Affine, 3, 1, 12, 1); row!(ark_mnt6_753::G2Affine, 3, 2, 12, 3);
    None
}

struct Resident { key: (usize, usize, u64), handle: *mut u8, n: usize }
struct State { ctx: *mut u8, bases: Vec<Resident> }
unsafe impl Send for State {}
static STATE: std::sync::Mutex<Option<State>> = std::sync::Mutex::new(None);

pub fn try_msm<G: AffineCurve>(bases: &[G], scalars: &[<G::ScalarField as PrimeField>::BigInt]) -> Option<G::Projective> {
    let (curve, group, limbs, deg) = group_of::<G>()?;
    let n = bases.len().min(scalars.len());
    if n < MIN_PAIRS { return None; }
    let mut guard = STATE.lock().ok()?;
    if guard.is_none() {
        let mut ctx = core::ptr::null_mut();
        if unsafe { pcdhip_init(0, &mut ctx) } != 0 { return None; }
        *guard = Some(State { ctx, bases: Vec::new() });
    }
    let st = guard.as_mut()?;
    // A prefix of an already resident vector (same start address) reuses it; anything else is uploaded.
    let words = 2 * deg * limbs;
    let start = bases.as_ptr() as usize;
    let digest = crate::msm_dispatch::slice_digest(bases, 0);
    let hit = st.bases.iter().position(|r| r.key.0 == start && r.n >= n && r.key.2 == digest);
    let idx = match hit {
        Some(i) => i,
        None => {
            // repack: `GroupAffine { x, y, infinity }` -> x || y limbs + flag bytes (coordinates are Montgomery `BigInteger` limbs)
            let mut xy = Vec::with_capacity(n * words);
            let mut inf = Vec::with_capacity(n);
            for p in &bases[..n] { crate::msm_dispatch::push_point(p, &mut xy, &mut inf, words); }
            let mut h = core::ptr::null_mut();
            if unsafe { pcdhip_bases_upload(st.ctx, curve, group, xy.as_ptr(), inf.as_ptr(), n, &mut h) } != 0 { return None; }
            st.bases.push(Resident { key: (start, n, digest), handle: h, n });
            st.bases.len() - 1
        }
    };
    let mut sc = Vec::with_capacity(n * limbs);
    for s in &scalars[..n] { sc.extend_from_slice(s.as_ref()); }
    let mut out = vec![0u64; 3 * deg * limbs];
    if unsafe { pcdhip_msm(st.ctx, st.bases[idx].handle, 0, sc.as_ptr(), n, out.as_mut_ptr()) } != 0 { return None; }
    Some(crate::msm_dispatch::projective_from_limbs::<G>(&out))
}

/// x || y Montgomery limbs of one point, as they sit in memory (the affine structs of the supported curves are
/// `{ x, y, infinity: bool, PhantomData }`: two field elements followed by the flag)
pub(crate) fn push_point<G: AffineCurve>(p: &G, xy: &mut Vec<u64>, inf: &mut Vec<u8>, words: usize) {
    inf.push(p.is_zero() as u8);
    xy.extend_from_slice(unsafe { core::slice::from_raw_parts(p as *const G as *const u64, words) });
}
/// X || Y || Z Montgomery limbs -> `G::Projective` (same layout argument; Z = 0 is the identity)
pub(crate) fn projective_from_limbs<G: AffineCurve>(xyz: &[u64]) -> G::Projective {
    let mut r = <G::Projective as ark_ff::Zero>::zero();
    unsafe { core::ptr::copy_nonoverlapping(xyz.as_ptr(), &mut r as *mut G::Projective as *mut u64, xyz.len()) };
    r
}
pub(crate) fn slice_digest<G: AffineCurve>(bases: &[G], _seed: u64) -> u64 {
    let bytes = |p: &G| unsafe { core::slice::from_raw_parts(p as *const G as *const u8, core::mem::size_of::<G>()) };
    let mut h: u64 = 0xcbf29ce484222325;
    for b in bytes(&bases[0]).iter().chain(bytes(&bases[bases.len() - 1]).iter()) { h ^= *b as u64; h = h.wrapping_mul(0x100000001b3); }
    h
}
