// TEST INFRASTRUCTURE ONLY (CPU oracle + reported CPU baseline) -- never linked into libpcdhip.so.
//
// PARITY UNPINNED: the reference holds no golden vectors for this path and its arithmetic lives in
// un-vendored upstream crates (Cargo.toml:16-42); see oracle/pyoracle.py and DESIGN.md.  This file
// restates, in C++17 with 64-bit limbs, the field layer those crates provide to the call sites
// src/ec_cycle_pcd/mod.rs:171,179 (SNARK::prove) and :239 (SNARK::verify):
//   ark-ff Fp320 / Fp768  : Montgomery residues, R = 2^(64 N), little-endian u64 limbs
//   ark-ff Fp2 / Fp3      : Fq[u]/(u^2 - nr), Fq[u]/(u^3 - nr)
//   ark-ff Fp4 / Fp6_2over3 : quadratic towers over Fp2 / Fp3 with v^2 = u
// It is validated against the independent big-integer oracle (tests/test_oracle_cpu.py).
#pragma once
#include <stdint.h>
#include <string.h>

#include "params_gen.hpp"

namespace orc {

typedef uint64_t u64;
typedef unsigned __int128 u128;

// ------------------------------------------------------------------------------------------------
// Prime-field parameter packs
#define ORC_DEF_FIELD(NAME, PFX)                                   \
  struct NAME {                                                    \
    static constexpr int ID = PFX##_ID;                            \
    static constexpr int N = PFX##_N64;                            \
    static constexpr int BITS = PFX##_BITS;                        \
    static constexpr int TWO_ADICITY = PFX##_TWO_ADICITY;          \
    static constexpr u64 INV = PFX##_INV64;                        \
    static constexpr u64 MOD[N] = PFX##_MOD;                       \
    static constexpr u64 R[N] = PFX##_R;                           \
    static constexpr u64 R2[N] = PFX##_R2;                         \
    static constexpr u64 GEN[N] = PFX##_GEN_MONT;                  \
    static constexpr u64 ROOT[N] = PFX##_ROOT_MONT;                \
    static constexpr u64 MODM2[N] = PFX##_MOD_MINUS_2;             \
  };
ORC_DEF_FIELD(F298A, PCD_F298A)
ORC_DEF_FIELD(F298B, PCD_F298B)
ORC_DEF_FIELD(F753A, PCD_F753A)
ORC_DEF_FIELD(F753B, PCD_F753B)

// ------------------------------------------------------------------------------------------------
template <class P>
struct Fp {
  static constexpr int N = P::N;
  typedef P Params;
  typedef Fp<P> Base;           // prime subfield
  static constexpr int DEG = 1;  // degree over the prime field
  u64 v[N];

  static Fp zero() { Fp r; memset(r.v, 0, sizeof r.v); return r; }
  static Fp one() { Fp r; memcpy(r.v, P::R, sizeof r.v); return r; }
  static Fp from_raw(const u64* p) { Fp r; memcpy(r.v, p, sizeof r.v); return r; }
  static Fp from_u64(u64 x) {  // canonical small integer -> Montgomery
    Fp r = zero(); r.v[0] = x;
    Fp r2 = from_raw(P::R2);
    return r * r2;
  }
  static Fp generator() { return from_raw(P::GEN); }
  static Fp two_adic_root() { return from_raw(P::ROOT); }

  bool is_zero() const { u64 o = 0; for (int i = 0; i < N; i++) o |= v[i]; return o == 0; }
  bool operator==(const Fp& b) const { u64 o = 0; for (int i = 0; i < N; i++) o |= v[i] ^ b.v[i]; return o == 0; }
  bool operator!=(const Fp& b) const { return !(*this == b); }

  static inline bool geq_mod(const u64* a) {
    for (int i = N - 1; i >= 0; i--) { if (a[i] > P::MOD[i]) return true; if (a[i] < P::MOD[i]) return false; }
    return true;
  }
  static inline void sub_mod(u64* a) {
    u64 borrow = 0;
    for (int i = 0; i < N; i++) { u128 d = (u128)a[i] - P::MOD[i] - borrow; a[i] = (u64)d; borrow = (u64)(d >> 64) & 1; }
  }
  Fp operator+(const Fp& b) const {
    Fp r; u64 c = 0;
    for (int i = 0; i < N; i++) { u128 s = (u128)v[i] + b.v[i] + c; r.v[i] = (u64)s; c = (u64)(s >> 64); }
    if (c || geq_mod(r.v)) sub_mod(r.v);  // top limb has spare bits, c is always 0
    return r;
  }
  Fp operator-(const Fp& b) const {
    Fp r; u64 borrow = 0;
    for (int i = 0; i < N; i++) { u128 d = (u128)v[i] - b.v[i] - borrow; r.v[i] = (u64)d; borrow = (u64)(d >> 64) & 1; }
    if (borrow) { u64 c = 0; for (int i = 0; i < N; i++) { u128 s = (u128)r.v[i] + P::MOD[i] + c; r.v[i] = (u64)s; c = (u64)(s >> 64); } }
    return r;
  }
  Fp neg() const { return is_zero() ? *this : (zero() - *this); }
  Fp dbl() const { return *this + *this; }

  // CIOS Montgomery multiplication (ark-ff `mul_assign`; canonical result in [0, p))
  Fp operator*(const Fp& b) const {
    u64 t[N + 2];
    for (int i = 0; i < N + 2; i++) t[i] = 0;
    for (int i = 0; i < N; i++) {
      u64 c = 0;
      for (int j = 0; j < N; j++) { u128 x = (u128)v[i] * b.v[j] + t[j] + c; t[j] = (u64)x; c = (u64)(x >> 64); }
      u128 x = (u128)t[N] + c; t[N] = (u64)x; t[N + 1] = (u64)(x >> 64);
      u64 m = t[0] * P::INV;
      x = (u128)m * P::MOD[0] + t[0]; c = (u64)(x >> 64);
      for (int j = 1; j < N; j++) { x = (u128)m * P::MOD[j] + t[j] + c; t[j - 1] = (u64)x; c = (u64)(x >> 64); }
      x = (u128)t[N] + c; t[N - 1] = (u64)x; t[N] = t[N + 1] + (u64)(x >> 64);
    }
    Fp r; memcpy(r.v, t, sizeof r.v);
    if (t[N] || geq_mod(r.v)) sub_mod(r.v);
    return r;
  }
  Fp sqr() const { return *this * *this; }

  template <int K>
  Fp pow_limbs(const u64 (&e)[K]) const { return pow(e, K); }
  Fp pow(const u64* e, int k) const {
    Fp r = one();
    bool started = false;
    for (int i = k * 64 - 1; i >= 0; i--) {
      if (started) r = r.sqr();
      if ((e[i / 64] >> (i % 64)) & 1) { r = started ? r * *this : *this; started = true; }
    }
    return r;
  }
  Fp inv() const { return pow(P::MODM2, N); }  // value is unique; upstream uses binary Euclid

  // canonical (non-Montgomery) limbs, as `into_repr()`
  void to_canonical(u64* out) const {
    Fp o = zero(); o.v[0] = 1;  // raw 1 => multiply by R^-1
    Fp r = *this * o;
    memcpy(out, r.v, sizeof r.v);
  }
  static Fp from_canonical(const u64* in) { return from_raw(in) * from_raw(P::R2); }

  // multiplication by a small non-negative integer (curve / tower coefficients)
  Fp mul_small(unsigned k) const {
    Fp acc = zero(), base = *this;
    while (k) { if (k & 1) acc = acc + base; base = base.dbl(); k >>= 1; }
    return acc;
  }
  Fp frobenius(int) const { return *this; }
};

// ------------------------------------------------------------------------------------------------
// Quadratic extension F[u]/(u^2 - NR), NR a small integer in the prime field.
template <class F, unsigned NR>
struct Fp2 {
  typedef typename F::Params Params;
  typedef F Base;
  static constexpr int DEG = 2;
  F c0, c1;
  static Fp2 zero() { return {F::zero(), F::zero()}; }
  static Fp2 one() { return {F::one(), F::zero()}; }
  bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  bool operator==(const Fp2& b) const { return c0 == b.c0 && c1 == b.c1; }
  bool operator!=(const Fp2& b) const { return !(*this == b); }
  Fp2 operator+(const Fp2& b) const { return {c0 + b.c0, c1 + b.c1}; }
  Fp2 operator-(const Fp2& b) const { return {c0 - b.c0, c1 - b.c1}; }
  Fp2 neg() const { return {c0.neg(), c1.neg()}; }
  Fp2 dbl() const { return {c0.dbl(), c1.dbl()}; }
  Fp2 operator*(const Fp2& b) const {  // Karatsuba
    F v0 = c0 * b.c0, v1 = c1 * b.c1;
    F s = (c0 + c1) * (b.c0 + b.c1);
    return {v0 + v1.mul_small(NR), s - v0 - v1};
  }
  Fp2 sqr() const { return *this * *this; }
  Fp2 mul_base(const F& k) const { return {c0 * k, c1 * k}; }
  Fp2 mul_small(unsigned k) const { return {c0.mul_small(k), c1.mul_small(k)}; }
  Fp2 inv() const {
    F n = (c0.sqr() - c1.sqr().mul_small(NR)).inv();
    return {c0 * n, (c1 * n).neg()};
  }
  Fp2 frobenius(int power) const { return (power & 1) ? Fp2{c0, c1.neg()} : *this; }  // u^q = -u
  // u * (c0 + c1 u) = nr c1 + c0 u
  Fp2 mul_by_u() const { return {c1.mul_small(NR), c0}; }
  Fp2 pow(const u64* e, int k) const {
    Fp2 r = one();
    for (int i = k * 64 - 1; i >= 0; i--) { r = r.sqr(); if ((e[i / 64] >> (i % 64)) & 1) r = r * *this; }
    return r;
  }
};

// Cubic extension F[u]/(u^3 - NR)
template <class F, unsigned NR>
struct Fp3 {
  typedef typename F::Params Params;
  typedef F Base;
  static constexpr int DEG = 3;
  F c0, c1, c2;
  static Fp3 zero() { return {F::zero(), F::zero(), F::zero()}; }
  static Fp3 one() { return {F::one(), F::zero(), F::zero()}; }
  bool is_zero() const { return c0.is_zero() && c1.is_zero() && c2.is_zero(); }
  bool operator==(const Fp3& b) const { return c0 == b.c0 && c1 == b.c1 && c2 == b.c2; }
  bool operator!=(const Fp3& b) const { return !(*this == b); }
  Fp3 operator+(const Fp3& b) const { return {c0 + b.c0, c1 + b.c1, c2 + b.c2}; }
  Fp3 operator-(const Fp3& b) const { return {c0 - b.c0, c1 - b.c1, c2 - b.c2}; }
  Fp3 neg() const { return {c0.neg(), c1.neg(), c2.neg()}; }
  Fp3 dbl() const { return {c0.dbl(), c1.dbl(), c2.dbl()}; }
  Fp3 operator*(const Fp3& b) const {  // Karatsuba-style (6 base multiplications)
    F ad = c0 * b.c0, be = c1 * b.c1, cf = c2 * b.c2;
    F x = (c1 + c2) * (b.c1 + b.c2) - be - cf;
    F y = (c0 + c1) * (b.c0 + b.c1) - ad - be;
    F z = (c0 + c2) * (b.c0 + b.c2) - ad + be - cf;
    return {ad + x.mul_small(NR), y + cf.mul_small(NR), z};
  }
  Fp3 sqr() const { return *this * *this; }
  Fp3 mul_base(const F& k) const { return {c0 * k, c1 * k, c2 * k}; }
  Fp3 mul_small(unsigned k) const { return {c0.mul_small(k), c1.mul_small(k), c2.mul_small(k)}; }
  Fp3 inv() const {
    F t0 = c0.sqr() - (c1 * c2).mul_small(NR);
    F t1 = c2.sqr().mul_small(NR) - c0 * c1;
    F t2 = c1.sqr() - c0 * c2;
    F n = (c0 * t0 + (c2 * t1 + c1 * t2).mul_small(NR)).inv();
    return {t0 * n, t1 * n, t2 * n};
  }
  // u^(q^i) = u * NR^((q^i - 1)/3); coefficients computed once (see frob_coeffs)
  static const F* frob_coeffs() {
    static F tab[3][2];
    static bool init = false;
    if (!init) {
      // e = (q - 1)/3 as limbs
      constexpr int N = F::N;
      u64 e[N]; u64 rem = 0;
      u64 qm1[N]; memcpy(qm1, Params::MOD, sizeof qm1); qm1[0] -= 1;
      for (int i = N - 1; i >= 0; i--) { u128 cur = ((u128)rem << 64) | qm1[i]; e[i] = (u64)(cur / 3); rem = (u64)(cur % 3); }
      F w = F::from_u64(NR).pow(e, N);  // NR^((q-1)/3): primitive cube root of unity
      tab[0][0] = F::one(); tab[0][1] = F::one();
      tab[1][0] = w; tab[1][1] = w * w;
      tab[2][0] = w * w; tab[2][1] = w * w * w * w;
      init = true;
    }
    return &tab[0][0];
  }
  Fp3 frobenius(int power) const {
    const F* t = frob_coeffs();
    int i = ((power % 3) + 3) % 3;
    return {c0, c1 * t[i * 2 + 0], c2 * t[i * 2 + 1]};
  }
  // u * (c0 + c1 u + c2 u^2) = nr c2 + c0 u + c1 u^2
  Fp3 mul_by_u() const { return {c2.mul_small(NR), c0, c1}; }
  Fp3 pow(const u64* e, int k) const {
    Fp3 r = one();
    for (int i = k * 64 - 1; i >= 0; i--) { r = r.sqr(); if ((e[i / 64] >> (i % 64)) & 1) r = r * *this; }
    return r;
  }
};

// Quadratic tower E[v]/(v^2 - u) over E = Fp2 or Fp3 (=> Fp4 / Fp6_2over3: the pairing target fields)
template <class E>
struct FpT2 {
  typedef typename E::Params Params;
  typedef typename E::Base F;
  static constexpr int DEG = 2 * E::DEG;
  E c0, c1;
  static FpT2 zero() { return {E::zero(), E::zero()}; }
  static FpT2 one() { return {E::one(), E::zero()}; }
  bool operator==(const FpT2& b) const { return c0 == b.c0 && c1 == b.c1; }
  bool operator!=(const FpT2& b) const { return !(*this == b); }
  FpT2 operator*(const FpT2& b) const {
    E v0 = c0 * b.c0, v1 = c1 * b.c1;
    E s = (c0 + c1) * (b.c0 + b.c1);
    return {v0 + v1.mul_by_u(), s - v0 - v1};
  }
  FpT2 sqr() const { return *this * *this; }
  FpT2 inv() const {
    E n = (c0.sqr() - c1.sqr().mul_by_u()).inv();
    return {c0 * n, (c1 * n).neg()};
  }
  FpT2 unitary_inverse() const { return {c0, c1.neg()}; }
  // v^(q^i) = v * NR^((q^i - 1)/(2 DEG_E)) ; the coefficient lies in the prime field
  static F frob_coeff(int power) {
    static F tab[2 * E::DEG];
    static bool init = false;
    constexpr int K = 2 * E::DEG;
    if (!init) {
      constexpr int N = F::N;
      u64 e[N]; u64 rem = 0;
      u64 qm1[N]; memcpy(qm1, Params::MOD, sizeof qm1); qm1[0] -= 1;
      for (int i = N - 1; i >= 0; i--) { u128 cur = ((u128)rem << 64) | qm1[i]; e[i] = (u64)(cur / K); rem = (u64)(cur % K); }
      F nr = E::one().mul_by_u().c0;                      // = NR for Fp3 (u*1 = (0,1,0))... see below
      // mul_by_u() of one() gives u itself; recover NR as u^DEG_E:
      E up = E::one();
      for (int i = 0; i < E::DEG; i++) up = up.mul_by_u();
      nr = up.c0;
      F w = nr.pow(e, N);  // NR^((q-1)/K)
      tab[0] = F::one();
      for (int i = 1; i < K; i++) tab[i] = tab[i - 1] * w;  // NR^((q^i-1)/K) = w^i since q = 1 mod K... (q^i-1)/K = (q-1)/K * (1+q+..+q^(i-1)) and w^q = w
      init = true;
    }
    return tab[((power % K) + K) % K];
  }
  FpT2 frobenius(int power) const {
    return {c0.frobenius(power), c1.frobenius(power).mul_base(frob_coeff(power))};
  }
  FpT2 pow(const u64* e, int k) const {
    FpT2 r = one();
    bool started = false;
    for (int i = k * 64 - 1; i >= 0; i--) {
      if (started) r = r.sqr();
      if ((e[i / 64] >> (i % 64)) & 1) { r = started ? r * *this : *this; started = true; }
    }
    return r;
  }
};

}  // namespace orc
