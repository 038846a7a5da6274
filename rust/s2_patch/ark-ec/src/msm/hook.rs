//! ark-ec fork, `src/msm/hook.rs` (declared by `pub mod hook;` in `src/msm/mod.rs`): a TYPE-ERASED acceleration hook for
//! `VariableBaseMSM::multi_scalar_mul`.
//!
//! This file names no curve crate -- it cannot: `ark-mnt4-298` & co. depend on `ark-ec`, so a fork of `ark-ec` that mentioned their
//! types would close a dependency cycle that Cargo rejects.  The fork only offers a slot for ONE function pointer; the crate that
//! knows the concrete curve types (`ark-pcd-hip`, which may depend on the curve crates) registers its marshalling routine there
//! (`ark_pcd_hip::s2::install()`), keyed by `TypeId`.  With nothing registered the cost is one relaxed atomic load per MSM.
//! `core` only: the fork keeps building under `no_std`.
use core::any::TypeId;
use core::sync::atomic::{AtomicUsize, Ordering};

/// `affine` = `TypeId::of::<G>()` of the caller's `G: AffineCurve`;
/// `bases` / `n_bases` = the caller's `&[G]`; `scalars` / `n_scalars` = its `&[<G::ScalarField as PrimeField>::BigInt]`;
/// `out` = uninitialised storage for ONE `G::Projective`.
/// Returns `true` after writing the result to `out`; `false` ("not mine": unknown type, too small, no device, any error) leaves
/// `out` untouched and the upstream CPU code runs.
///
/// Safety contract (upheld by the one call site below): the pointers really are slices of the types `affine` identifies, valid for
/// the duration of the call; the hook must not retain them.
pub type MsmHook = unsafe fn(affine: TypeId, bases: *const u8, n_bases: usize, scalars: *const u8, n_scalars: usize, out: *mut u8) -> bool;

static MSM_HOOK: AtomicUsize = AtomicUsize::new(0);

/// Register the hook (process-wide, first registration wins; returns whether this call installed it).
pub fn set_msm_hook(hook: MsmHook) -> bool {
    MSM_HOOK.compare_exchange(0, hook as usize, Ordering::AcqRel, Ordering::Acquire).is_ok()
}

#[inline]
pub(crate) fn msm_hook() -> Option<MsmHook> {
    match MSM_HOOK.load(Ordering::Acquire) {
        0 => None,
        // a value stored by `set_msm_hook` is a valid `MsmHook` by construction
        p => Some(unsafe { core::mem::transmute::<usize, MsmHook>(p) }),
    }
}

/// What `multi_scalar_mul` calls first (the ONE edited call site, `src/msm/variable_base.rs`):
/// ```ignore
/// pub fn multi_scalar_mul<G: AffineCurve>(bases: &[G], scalars: &[<G::ScalarField as PrimeField>::BigInt]) -> G::Projective {
///     if let Some(r) = super::hook::try_hook::<G>(bases, scalars) { return r; }      // <- added
///     /* upstream body unchanged */
/// }
/// ```
/// (`AffineCurve: 'static` upstream, so `TypeId::of::<G>()` needs no new bound.)
#[inline]
pub fn try_hook<G: crate::AffineCurve>(
    bases: &[G], scalars: &[<G::ScalarField as ark_ff::PrimeField>::BigInt],
) -> Option<G::Projective> {
    let hook = msm_hook()?;
    let mut out = core::mem::MaybeUninit::<G::Projective>::uninit();
    let done = unsafe {
        hook(TypeId::of::<G>(), bases.as_ptr() as *const u8, bases.len(), scalars.as_ptr() as *const u8, scalars.len(),
             out.as_mut_ptr() as *mut u8)
    };
    if done { Some(unsafe { out.assume_init() }) } else { None }
}
