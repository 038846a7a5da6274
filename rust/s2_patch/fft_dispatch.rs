//! ark-poly fork, `src/fft_dispatch.rs`: route `Radix2EvaluationDomain::{fft, ifft}_in_place` over the four scalar fields
//! libpcdhip.so supports to `pcdhip_fft` (the coset variants upstream builds from these by a distribute-powers pass keep working
//! unchanged); `false` = run the upstream CPU code.  Only field-element coefficients (`T == F`) are routed.
use ark_ff::FftField;
use core::any::TypeId;
use std::os::raw::c_int;

extern "C" {
    fn pcdhip_init(device_id: c_int, out: *mut *mut u8) -> c_int;
    fn pcdhip_fft(ctx: *mut u8, field: c_int, data: *mut u64, log_n: u32, inverse: c_int, coset: c_int) -> c_int;
}

const MIN_LOG_N: u32 = 14;

fn field_of<F: 'static>() -> Option<c_int> {
    let t = TypeId::of::<F>();
    if t == TypeId::of::<ark_mnt4_298::Fq>() { return Some(0); }   // = MNT6-298 Fr
    if t == TypeId::of::<ark_mnt4_298::Fr>() { return Some(1); }   // = MNT6-298 Fq
    if t == TypeId::of::<ark_mnt4_753::Fq>() { return Some(2); }
    if t == TypeId::of::<ark_mnt4_753::Fr>() { return Some(3); }
    None
}

struct Ctx(*mut u8);
unsafe impl Send for Ctx {}
static CTX: std::sync::Mutex<Option<Ctx>> = std::sync::Mutex::new(None);

pub fn try_fft<F: FftField, T: 'static>(coeffs: &mut Vec<T>, log_n: u32, inverse: bool) -> bool {
    if TypeId::of::<T>() != TypeId::of::<F>() || log_n < MIN_LOG_N { return false; }
    let field = match field_of::<F>() { Some(f) => f, None => return false };
    let mut guard = match CTX.lock() { Ok(g) => g, Err(_) => return false };
    if guard.is_none() {
        let mut ctx = core::ptr::null_mut();
        if unsafe { pcdhip_init(0, &mut ctx) } != 0 { return false; }
        *guard = Some(Ctx(ctx));
    }
    // in place on the vector's own storage: elements are Montgomery `BigInteger` limbs, the C-ABI's encoding
    unsafe { pcdhip_fft(guard.as_ref().unwrap().0, field, coeffs.as_mut_ptr() as *mut u64, log_n, inverse as c_int, 0) == 0 }
}
