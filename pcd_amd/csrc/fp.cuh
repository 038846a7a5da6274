// Device prime-field arithmetic for the four MNT scalar/base fields (gfx950).
//
// Replaces, on device, ark-ff `Fp320` / `Fp768` as used by the prover arithmetic reached from
// /root/reference src/ec_cycle_pcd/mod.rs:171,179 (SNARK::prove).  Elements are Montgomery
// residues with R = 2^(32 N) (N = 10 / 24 u32 limbs == 5 / 12 u64 limbs, little-endian), i.e. the
// in-memory image of the upstream field types, so the C-ABI needs no conversion.
//
// Instruction-rate facts this file is written against (profiles/r01_k0_int_rates.txt, MI355X):
// v_mad_u64_u32 issues at half the v_add_u32 rate and so does every carry op (v_addc_co_u32), so
// the product loops keep a 64-bit running value per mad and avoid separate carry instructions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "params_gen.h"

namespace pcd {

#define PCD_DEV __device__ __forceinline__
#define PCD_HD __host__ __device__ __forceinline__

// ------------------------------------------------------------------------------------------------
#define PCD_DEF_FIELD(NAME, PFX)                                                             \
  struct NAME {                                                                              \
    static constexpr int ID = PFX##_ID;                                                      \
    static constexpr int N = PFX##_N32;                                                      \
    static constexpr int BITS = PFX##_BITS;                                                  \
    static constexpr int TWO_ADICITY = PFX##_TWO_ADICITY;                                    \
    static constexpr uint32_t INV = PFX##_INV32;                                             \
    PCD_HD static uint32_t mod(int i) { constexpr uint32_t m[N] = PFX##_MOD; return m[i]; }  \
    PCD_HD static uint32_t r1(int i) { constexpr uint32_t m[N] = PFX##_R; return m[i]; }     \
    PCD_HD static uint32_t r2(int i) { constexpr uint32_t m[N] = PFX##_R2; return m[i]; }    \
    PCD_HD static uint32_t gen(int i) { constexpr uint32_t m[N] = PFX##_GEN_MONT; return m[i]; }   \
    PCD_HD static uint32_t root(int i) { constexpr uint32_t m[N] = PFX##_ROOT_MONT; return m[i]; } \
    PCD_HD static uint32_t modm2(int i) { constexpr uint32_t m[N] = PFX##_MOD_MINUS_2; return m[i]; } \
  };
PCD_DEF_FIELD(F298A, PCD_F298A)
PCD_DEF_FIELD(F298B, PCD_F298B)
PCD_DEF_FIELD(F753A, PCD_F753A)
PCD_DEF_FIELD(F753B, PCD_F753B)

template <class P>
struct Fp {
  typedef P Params;
  typedef Fp<P> Base;
  static constexpr int N = P::N;
  static constexpr int DEG = 1;
  static constexpr int WORDS = N;  // u32 words per element
  uint32_t v[N];

  PCD_HD static Fp zero() { Fp r; for (int i = 0; i < N; i++) r.v[i] = 0; return r; }
  PCD_HD static Fp one() { Fp r; for (int i = 0; i < N; i++) r.v[i] = P::r1(i); return r; }
  PCD_HD static Fp r2() { Fp r; for (int i = 0; i < N; i++) r.v[i] = P::r2(i); return r; }
  PCD_HD static Fp generator() { Fp r; for (int i = 0; i < N; i++) r.v[i] = P::gen(i); return r; }
  PCD_HD static Fp two_adic_root() { Fp r; for (int i = 0; i < N; i++) r.v[i] = P::root(i); return r; }

  PCD_HD bool is_zero() const { uint32_t o = 0; for (int i = 0; i < N; i++) o |= v[i]; return o == 0; }
  PCD_HD bool operator==(const Fp& b) const { uint32_t o = 0; for (int i = 0; i < N; i++) o |= v[i] ^ b.v[i]; return o == 0; }
  PCD_HD bool operator!=(const Fp& b) const { return !(*this == b); }

  // r = (a >= p) ? a - p : a      (a < 2p)
  PCD_HD void reduce_once() {
    uint32_t d[N];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint64_t x = (uint64_t)v[i] - P::mod(i) - borrow;
      d[i] = (uint32_t)x;
      borrow = (x >> 32) & 1;
    }
    if (!borrow) {
#pragma unroll
      for (int i = 0; i < N; i++) v[i] = d[i];
    }
  }
  PCD_HD Fp operator+(const Fp& b) const {
    Fp r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { uint64_t x = (uint64_t)v[i] + b.v[i] + c; r.v[i] = (uint32_t)x; c = x >> 32; }
    r.reduce_once();  // the top limb has >= 15 spare bits: no carry out of limb N-1
    return r;
  }
  PCD_HD Fp operator-(const Fp& b) const {
    Fp r;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { uint64_t x = (uint64_t)v[i] - b.v[i] - borrow; r.v[i] = (uint32_t)x; borrow = (x >> 32) & 1; }
    uint32_t mask = (uint32_t)0 - (uint32_t)borrow;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { uint64_t x = (uint64_t)r.v[i] + (P::mod(i) & mask) + c; r.v[i] = (uint32_t)x; c = x >> 32; }
    return r;
  }
  PCD_HD Fp neg() const { return is_zero() ? *this : (zero() - *this); }
  PCD_HD Fp dbl() const { return *this + *this; }

  // CIOS Montgomery product, canonical output in [0, p).  Deliberately NOT inlined: one fully
  // unrolled copy of the multiplier per field per code object keeps the point kernels (11-16
  // products per group operation, x3 / x6 for Fq2 / Fq3) at a compilable size; operands travel by
  // value in VGPRs.  (A rolled row loop reads a.v[i] through scratch: ~500 cycles of exposed latency
  // per row for a lone wave -- measured 2.5 us per 298-bit product in the serial MSM tail.)
  __host__ __device__ __noinline__ static Fp mul(Fp a, Fp b) {
    uint32_t t[N + 2];
#pragma unroll
    for (int i = 0; i < N + 2; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      const uint32_t ai = a.v[i];
      uint32_t c = 0;
#pragma unroll
      for (int j = 0; j < N; j++) { uint64_t x = (uint64_t)ai * b.v[j] + t[j] + c; t[j] = (uint32_t)x; c = (uint32_t)(x >> 32); }
      uint64_t x = (uint64_t)t[N] + c;
      t[N] = (uint32_t)x;
      t[N + 1] = (uint32_t)(x >> 32);
      uint32_t m = t[0] * P::INV;
      x = (uint64_t)m * P::mod(0) + t[0];
      c = (uint32_t)(x >> 32);
#pragma unroll
      for (int j = 1; j < N; j++) { x = (uint64_t)m * P::mod(j) + t[j] + c; t[j - 1] = (uint32_t)x; c = (uint32_t)(x >> 32); }
      x = (uint64_t)t[N] + c;
      t[N - 1] = (uint32_t)x;
      t[N] = t[N + 1] + (uint32_t)(x >> 32);
    }
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = t[i];
    r.reduce_once();  // t < 2p since p < 2^(32N - 1)
    return r;
  }
  PCD_HD Fp operator*(const Fp& b) const { return mul(*this, b); }
  PCD_HD Fp sqr() const { return *this * *this; }

  PCD_HD Fp mul_small(unsigned k) const {
    Fp acc = zero(), base = *this;
    while (k) { if (k & 1) acc = acc + base; k >>= 1; if (k) base = base.dbl(); }
    return acc;
  }
  // a^(p-2) (inverse; zero maps to zero)
  PCD_HD Fp inv() const {
    Fp r = one();
    bool started = false;
    for (int i = N * 32 - 1; i >= 0; i--) {
      if (started) r = r.sqr();
      if ((P::modm2(i >> 5) >> (i & 31)) & 1) { r = started ? r * *this : *this; started = true; }
    }
    return r;
  }
  PCD_HD Fp pow_u64(uint64_t e) const {
    Fp r = one(), b = *this;
    while (e) { if (e & 1) r = r * b; e >>= 1; if (e) b = b.sqr(); }
    return r;
  }
  PCD_HD static Fp from_u64(uint64_t x) {
    Fp r = zero();
    r.v[0] = (uint32_t)x;
    r.v[1] = (uint32_t)(x >> 32);
    return r * r2();
  }
  PCD_HD Fp from_mont() const { Fp o = zero(); o.v[0] = 1; return *this * o; }  // -> canonical limbs
  PCD_HD Fp to_mont() const { return *this * r2(); }                          // canonical -> Montgomery

  PCD_HD static Fp load(const uint32_t* p) {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = p[i];
    return r;
  }
  PCD_HD void store(uint32_t* p) const {
#pragma unroll
    for (int i = 0; i < N; i++) p[i] = v[i];
  }
};

// ------------------------------------------------------------------------------------------------
// F[u]/(u^2 - NR)   (ark-ff Fp2; G2 coordinates of MNT4)
template <class F, unsigned NR>
struct Fp2 {
  typedef typename F::Params Params;
  typedef F Base;
  static constexpr int DEG = 2;
  static constexpr int WORDS = 2 * F::WORDS;
  F c0, c1;
  PCD_HD static Fp2 zero() { return {F::zero(), F::zero()}; }
  PCD_HD static Fp2 one() { return {F::one(), F::zero()}; }
  PCD_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  PCD_HD bool operator==(const Fp2& b) const { return c0 == b.c0 && c1 == b.c1; }
  PCD_HD bool operator!=(const Fp2& b) const { return !(*this == b); }
  PCD_HD Fp2 operator+(const Fp2& b) const { return {c0 + b.c0, c1 + b.c1}; }
  PCD_HD Fp2 operator-(const Fp2& b) const { return {c0 - b.c0, c1 - b.c1}; }
  PCD_HD Fp2 neg() const { return {c0.neg(), c1.neg()}; }
  PCD_HD Fp2 dbl() const { return {c0.dbl(), c1.dbl()}; }
  // (not inlined, like Fp::mul: bounds the code size of the G2 point kernels; operands by reference --
  // large by-value aggregates passed on the stack miscompiled for Fp3 on gfx950 / ROCm 7.2)
  __host__ __device__ __noinline__ static void mul(Fp2& o, const Fp2& a, const Fp2& b) {
    F v0 = a.c0 * b.c0, v1 = a.c1 * b.c1;
    F s = (a.c0 + a.c1) * (b.c0 + b.c1);
    o.c0 = v0 + v1.mul_small(NR);
    o.c1 = s - v0 - v1;
  }
  __host__ __device__ __noinline__ static void sqr_(Fp2& o, const Fp2& a) {  // complex squaring: 2 base multiplications
    F ab = a.c0 * a.c1;
    F t = (a.c0 + a.c1) * (a.c0 + a.c1.mul_small(NR));
    o.c0 = t - ab - ab.mul_small(NR);
    o.c1 = ab.dbl();
  }
  PCD_HD Fp2 operator*(const Fp2& b) const { Fp2 o; mul(o, *this, b); return o; }
  PCD_HD Fp2 sqr() const { Fp2 o; sqr_(o, *this); return o; }
  PCD_HD Fp2 mul_small(unsigned k) const { return {c0.mul_small(k), c1.mul_small(k)}; }
  PCD_HD Fp2 mul_base(const F& k) const { return {c0 * k, c1 * k}; }
  PCD_HD Fp2 inv() const {
    F n = (c0.sqr() - c1.sqr().mul_small(NR)).inv();
    return {c0 * n, (c1 * n).neg()};
  }
  PCD_HD static Fp2 load(const uint32_t* p) { return {F::load(p), F::load(p + F::WORDS)}; }
  PCD_HD void store(uint32_t* p) const { c0.store(p); c1.store(p + F::WORDS); }
};

// F[u]/(u^3 - NR)   (ark-ff Fp3; G2 coordinates of MNT6)
template <class F, unsigned NR>
struct Fp3 {
  typedef typename F::Params Params;
  typedef F Base;
  static constexpr int DEG = 3;
  static constexpr int WORDS = 3 * F::WORDS;
  F c0, c1, c2;
  PCD_HD static Fp3 zero() { return {F::zero(), F::zero(), F::zero()}; }
  PCD_HD static Fp3 one() { return {F::one(), F::zero(), F::zero()}; }
  PCD_HD bool is_zero() const { return c0.is_zero() && c1.is_zero() && c2.is_zero(); }
  PCD_HD bool operator==(const Fp3& b) const { return c0 == b.c0 && c1 == b.c1 && c2 == b.c2; }
  PCD_HD bool operator!=(const Fp3& b) const { return !(*this == b); }
  PCD_HD Fp3 operator+(const Fp3& b) const { return {c0 + b.c0, c1 + b.c1, c2 + b.c2}; }
  PCD_HD Fp3 operator-(const Fp3& b) const { return {c0 - b.c0, c1 - b.c1, c2 - b.c2}; }
  PCD_HD Fp3 neg() const { return {c0.neg(), c1.neg(), c2.neg()}; }
  PCD_HD Fp3 dbl() const { return {c0.dbl(), c1.dbl(), c2.dbl()}; }
  __host__ __device__ __noinline__ static void mul(Fp3& o, const Fp3& a, const Fp3& b) {
    F ad = a.c0 * b.c0, be = a.c1 * b.c1, cf = a.c2 * b.c2;
    F x = (a.c1 + a.c2) * (b.c1 + b.c2) - be - cf;
    F y = (a.c0 + a.c1) * (b.c0 + b.c1) - ad - be;
    F z = (a.c0 + a.c2) * (b.c0 + b.c2) - ad + be - cf;
    o.c0 = ad + x.mul_small(NR);
    o.c1 = y + cf.mul_small(NR);
    o.c2 = z;
  }
  __host__ __device__ __noinline__ static void sqr_(Fp3& o, const Fp3& a) {  // CH-SQR2: 2 mul + 3 sqr in the base field
    F s0 = a.c0.sqr();
    F ab = a.c0 * a.c1;
    F s1 = ab.dbl();
    F s2 = (a.c0 - a.c1 + a.c2).sqr();
    F bc = a.c1 * a.c2;
    F s3 = bc.dbl();
    F s4 = a.c2.sqr();
    o.c0 = s0 + s3.mul_small(NR);
    o.c1 = s1 + s4.mul_small(NR);
    o.c2 = s1 + s2 + s3 - s0 - s4;
  }
  PCD_HD Fp3 operator*(const Fp3& b) const { Fp3 o; mul(o, *this, b); return o; }
  PCD_HD Fp3 sqr() const { Fp3 o; sqr_(o, *this); return o; }
  PCD_HD Fp3 mul_small(unsigned k) const { return {c0.mul_small(k), c1.mul_small(k), c2.mul_small(k)}; }
  PCD_HD Fp3 mul_base(const F& k) const { return {c0 * k, c1 * k, c2 * k}; }
  PCD_HD Fp3 inv() const {
    F t0 = c0.sqr() - (c1 * c2).mul_small(NR);
    F t1 = c2.sqr().mul_small(NR) - c0 * c1;
    F t2 = c1.sqr() - c0 * c2;
    F n = (c0 * t0 + (c2 * t1 + c1 * t2).mul_small(NR)).inv();
    return {t0 * n, t1 * n, t2 * n};
  }
  PCD_HD static Fp3 load(const uint32_t* p) { return {F::load(p), F::load(p + F::WORDS), F::load(p + 2 * F::WORDS)}; }
  PCD_HD void store(uint32_t* p) const { c0.store(p); c1.store(p + F::WORDS); c2.store(p + 2 * F::WORDS); }
};

}  // namespace pcd
