//! ark-poly fork, `src/domain/hook.rs` (declared by `pub mod hook;` in `src/domain/mod.rs`): a TYPE-ERASED acceleration hook for
//! `Radix2EvaluationDomain::{fft, ifft}_in_place` (the coset variants upstream builds from these by a distribute-powers pass keep
//! working unchanged, as do `GeneralEvaluationDomain`'s radix-2 arm and everything in ark-marlin / ark-poly-commit above it).
//!
//! Like the ark-ec hook this file names no curve crate (the curve crates depend on `ark-ff`, `ark-ec`; `ark-poly` must stay
//! below them): one slot for a function pointer, registered by `ark_pcd_hip::s2::install()` and keyed by the field's `TypeId`.
use core::any::TypeId;
use core::sync::atomic::{AtomicUsize, Ordering};

/// `field` = `TypeId::of::<F>()`; `data` / `len` = the caller's coefficient vector, already resized to the domain (`len` =
/// `1 << log_n` elements of `F`, transformed IN PLACE, natural order in and out); `inverse` = `ifft` (includes the `1 / n` scaling).
/// `true` = done; `false` = untouched, the upstream CPU code runs.
pub type FftHook = unsafe fn(field: TypeId, data: *mut u8, len: usize, log_n: u32, inverse: bool) -> bool;

static FFT_HOOK: AtomicUsize = AtomicUsize::new(0);

pub fn set_fft_hook(hook: FftHook) -> bool {
    FFT_HOOK.compare_exchange(0, hook as usize, Ordering::AcqRel, Ordering::Acquire).is_ok()
}

/// The edited call sites (`src/domain/radix2/mod.rs`, the `EvaluationDomain` impl):
/// ```ignore
/// fn fft_in_place<T: DomainCoeff<F>>(&self, coeffs: &mut Vec<T>) {
///     coeffs.resize(self.size(), T::zero());
///     if crate::domain::hook::try_hook::<F, T>(coeffs, self.log_size_of_group, false) { return; }   // <- added (ifft_in_place: `true`)
///     /* upstream body unchanged */
/// }
/// ```
/// Only vectors of field elements are offered (`T == F`; Marlin also transforms nothing else).  `DomainCoeff<F>` does not imply
/// `'static`, so `TypeId::of::<T>()` does not compile; `type_id_of::<T>()` below gets the same `TypeId` through a trait object whose
/// `'static` bound is asserted for the call only (the lifetime-erased identity of `T`; dtolnay's `typeid` crate does the same).  Round 4
/// compared `type_name`s, which Rust does not promise to be unique.
fn type_id_of<T: ?Sized>() -> TypeId {
    trait NonStaticAny { fn get_type_id(&self) -> TypeId where Self: 'static; }
    impl<T: ?Sized> NonStaticAny for core::marker::PhantomData<T> {
        fn get_type_id(&self) -> TypeId where Self: 'static { TypeId::of::<T>() }
    }
    let phantom = core::marker::PhantomData::<T>;
    // sound: only the vtable's method is called, which reads no data of lifetime-limited type
    let erased: &(dyn NonStaticAny + 'static) = unsafe { core::mem::transmute::<&dyn NonStaticAny, &(dyn NonStaticAny + 'static)>(&phantom) };
    erased.get_type_id()
}

#[inline]
pub fn try_hook<F: ark_ff::FftField, T>(coeffs: &mut [T], log_n: u32, inverse: bool) -> bool {
    let hook = match FFT_HOOK.load(Ordering::Acquire) {
        0 => return false,
        p => unsafe { core::mem::transmute::<usize, FftHook>(p) },
    };
    if core::mem::size_of::<T>() != core::mem::size_of::<F>()
        || core::mem::align_of::<T>() != core::mem::align_of::<F>()
        || type_id_of::<T>() != TypeId::of::<F>()
        || coeffs.len() != 1usize << log_n
    {
        return false;
    }
    unsafe { hook(TypeId::of::<F>(), coeffs.as_mut_ptr() as *mut u8, coeffs.len(), log_n, inverse) }
}
