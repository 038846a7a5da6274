// MNT4 / MNT6 ate pairing on gfx950 (K6 of SURVEY.md section 8).
//
// Replaces ark-ec `PairingEngine::{miller_loop, final_exponentiation, product_of_pairings}` for
// `models::mnt4` / `models::mnt6`, reached from /root/reference src/ec_cycle_pcd/mod.rs:239
// (`IC::HelpSNARK::verify` -> Groth16 verify: three Miller loops + one final exponentiation) and
// mod.rs:71 (`process_vk`).  Same algorithm as upstream: G2 in extended Jacobian coordinates
// (X, Y, Z, T = Z^2) stepping through |q - r| with doubling / mixed-addition line coefficients, lines
// evaluated at the twisted G1 point ("flipped" Miller loop), final exponentiation
// (q^(k/2) - 1)[(q + 1)] then q + w0.  One lane per (P, Q) pair (the THROUGHPUT form, used for batches above 4096 pairs; smaller batches
// run one wave per pairing: pairing_vm.hip.h): the G2 steps are fused with the line
// evaluations instead of being precomputed; independent pairs fill a wave.
#pragma once
#include "ec.hip.h"

namespace pcd {

// E[v]/(v^2 - u) over E = Fp2 / Fp3  (ark-ff Fp4 / Fp6_2over3)
template <class E>
struct FpT2 {
  typedef typename E::Base F;
  static constexpr int WORDS = 2 * E::WORDS;
  static constexpr int K = 2 * E::DEG;
  E c0, c1;
  PCD_HD static FpT2 one() { return {E::one(), E::zero()}; }
  PCD_HD static E mul_by_u(const E& x);
  PCD_HD FpT2 operator*(const FpT2& b) const {
    E v0 = c0 * b.c0, v1 = c1 * b.c1;
    E s = (c0 + c1) * (b.c0 + b.c1);
    return {v0 + mul_by_u(v1), s - v0 - v1};
  }
  PCD_HD FpT2 sqr() const {
    E ab = c0 * c1;
    E t = (c0 + c1) * (c0 + mul_by_u(c1));
    return {t - ab - mul_by_u(ab), ab.dbl()};
  }
  PCD_HD FpT2 inv() const {
    E n = (c0.sqr() - mul_by_u(c1.sqr())).inv();
    return {c0 * n, (c1 * n).neg()};
  }
  PCD_HD bool operator==(const FpT2& b) const { return c0 == b.c0 && c1 == b.c1; }
  static constexpr int ABI_WORDS = 2 * E::ABI_WORDS;
  PCD_HD static FpT2 load(const uint32_t* p) { return {E::load(p), E::load(p + E::WORDS)}; }
  PCD_HD void store(uint32_t* p) const { c0.store(p); c1.store(p + E::WORDS); }
  PCD_HD void to_abi(uint32_t* w) const { c0.to_abi(w); c1.to_abi(w + E::ABI_WORDS); }
};
template <class F, unsigned NR>
PCD_HD Fp2<F, NR> ext_mul_by_u(const Fp2<F, NR>& x) { return {x.c1.mul_small(NR), x.c0}; }
template <class F, unsigned NR>
PCD_HD Fp3<F, NR> ext_mul_by_u(const Fp3<F, NR>& x) { return {x.c2.mul_small(NR), x.c0, x.c1}; }
template <class E>
PCD_HD E FpT2<E>::mul_by_u(const E& x) { return ext_mul_by_u(x); }

// Frobenius data: powers w^i of w = NR^((q-1)/K) (K = 4 or 6), computed once per kernel on the device.
template <class F, int K>
struct FrobTable {
  F w[K];
};
template <class F, int K, unsigned NR>
PCD_HD void frob_init(FrobTable<F, K>& t) {
  typedef typename F::Params P;
  // e = (q - 1) / K, long division on 32-bit limbs
  // q - 1 = (q - 2) + 1 from the 32-bit words of the exponent table (the +1 may ripple: q = 1 mod 2^32 for
  // the fields with 2-adicity >= 32), then long division by K
  constexpr int NW = P::N32;
  uint32_t qm1[NW], e[NW];
  uint64_t carry = 1;
  for (int i = 0; i < NW; i++) { uint64_t x = (uint64_t)P::modm2(i) + carry; qm1[i] = (uint32_t)x; carry = x >> 32; }
  uint64_t rem = 0;
  for (int i = NW - 1; i >= 0; i--) {
    uint64_t cur = (rem << 32) | (uint64_t)qm1[i];
    e[i] = (uint32_t)(cur / K);
    rem = cur % K;
  }
  F base = F::from_u64(NR), r = F::one();
  bool started = false;
  for (int i = NW * 32 - 1; i >= 0; i--) {
    if (started) r = r.sqr();
    if ((e[i >> 5] >> (i & 31)) & 1) { r = started ? r * base : base; started = true; }
  }
  t.w[0] = F::one();
  for (int i = 1; i < K; i++) t.w[i] = t.w[i - 1] * r;
}
// x^(q^i) for x in Fp2 / Fp3 / FpT2 given the table (u^(q^i) = u * w^(i * K / deg), v^(q^i) = v * w^i)
template <class F, unsigned NR, int K>
PCD_HD Fp2<F, NR> frob(const Fp2<F, NR>& x, int i, const FrobTable<F, K>& t) { return {x.c0, x.c1 * t.w[(i * (K / 2)) % K]}; }
template <class F, unsigned NR, int K>
PCD_HD Fp3<F, NR> frob(const Fp3<F, NR>& x, int i, const FrobTable<F, K>& t) {
  return {x.c0, x.c1 * t.w[(i * (K / 3)) % K], x.c2 * t.w[(2 * i * (K / 3)) % K]};
}
template <class E, class F, int K>
PCD_HD FpT2<E> frob(const FpT2<E>& x, int i, const FrobTable<F, K>& t) {
  return {frob(x.c0, i, t), frob(x.c1, i, t).mul_base(t.w[i % K])};
}

// PC: pairing config -- G1 / G2 group configs, NR, loop count and final-exponent constants
template <class PC>
struct Pairing {
  typedef typename PC::G1 GA;
  typedef typename PC::G2 GB;
  typedef typename GA::F Fq;
  typedef typename GB::F E;
  typedef FpT2<E> Fqk;
  static constexpr int K = Fqk::K;
  typedef FrobTable<Fq, K> Frob;
  struct Ext { E x, y, z, t; };

  PCD_HD static E twist_mul(const Fq& v) {  // v * twist, twist = u
    E r = E::zero();
    r.c1 = v;
    return r;
  }
  PCD_HD static E lift(const Fq& v) { E r = E::zero(); r.c0 = v; return r; }
  PCD_HD static bool loop_bit(int i) { return (PC::loop(i >> 5) >> (i & 31)) & 1; }

  // f_{|T|,Q}(P) (inverted when T < 0), as ark-ec `ate_miller_loop(prepare(P), prepare(Q))`
  PCD_HD static Fqk miller_loop(const Aff<Fq>& p, const Aff<E>& q) {
    if (p.is_inf() || q.is_inf()) return Fqk::one();
    const E px_twist = twist_mul(p.x), py_twist = twist_mul(p.y);
    E tinv = E::zero();
    tinv.c1 = Fq::one();
    tinv = tinv.inv();  // twist^-1
    const E qx_over = q.x * tinv, qy_over = q.y * tinv;
    const E l1_coeff = lift(p.x) - qx_over;
    Ext r = {q.x, q.y, E::one(), E::one()};
    Fqk f = Fqk::one();
    for (int i = PC::LOOP_BITS - 2; i >= 0; i--) {
      {  // doubling step + line
        E a = r.t.sqr(), b = r.x.sqr(), c = r.y.sqr(), d = c.sqr();
        E e = (r.x + c).sqr() - b - d;
        E fq = b.dbl() + b + GB::mul_by_a(a);
        E g = fq.sqr();
        Ext o;
        o.x = g - e.dbl().dbl();
        o.y = fq * (e.dbl() - o.x) - d.dbl().dbl().dbl();
        o.z = (r.y + r.z).sqr() - c - r.z.sqr();
        o.t = o.z.sqr();
        E c_h = (o.z + r.t).sqr() - o.t - a;
        E c_4c = c.dbl().dbl();
        E c_j = (fq + r.t).sqr() - g - a;
        E c_l = (fq + r.x).sqr() - g - b;
        Fqk g_rr = {c_l - c_4c - c_j * px_twist, c_h * py_twist};
        f = f.sqr() * g_rr;
        r = o;
      }
      if (loop_bit(i)) f = f * add_step(r, q.x, q.y, py_twist, qy_over, l1_coeff);
    }
    if (PC::LOOP_NEG) {
      E zi = r.z.inv(), zi2 = zi.sqr();
      E mx = r.x * zi2, my = (r.y * zi2 * zi).neg();
      f = (f * add_step(r, mx, my, py_twist, qy_over, l1_coeff)).inv();
    }
    return f;
  }
  // mixed addition r += (x, y) in extended coordinates, returns the line value at P
  PCD_HD static Fqk add_step(Ext& r, const E& x, const E& y, const E& py_twist, const E& qy_over, const E& l1_coeff) {
    E a = y.sqr();
    E b = r.t * x;
    E d = ((r.z + y).sqr() - a - r.t) * r.t;
    E h = b - r.x;
    E i = h.sqr();
    E e = i.dbl().dbl();
    E j = h * e;
    E v = r.x * e;
    E l1 = d - r.y.dbl();
    Ext o;
    o.x = l1.sqr() - j - v.dbl();
    o.y = l1 * (v - o.x) - j * r.y.dbl();
    o.z = (r.z + h).sqr() - r.t - i;
    o.t = o.z.sqr();
    r = o;
    return {o.z * py_twist, (qy_over * o.z + l1_coeff * l1).neg()};
  }
  PCD_HD static Fqk pow_w0(const Fqk& x) {
    Fqk r = Fqk::one();
    bool started = false;
    for (int i = PC::W0_BITS - 1; i >= 0; i--) {
      if (started) r = r.sqr();
      if ((PC::w0(i >> 5) >> (i & 31)) & 1) { r = started ? r * x : x; started = true; }
    }
    return r;
  }
  PCD_HD static Fqk final_exponentiation(const Fqk& v, const Frob& t) {
    Fqk vi = v.inv();
    Fqk first, first_inv;
    if (K == 4) {  // v^(q^2 - 1)
      first = frob(v, 2, t) * vi;
      first_inv = frob(vi, 2, t) * v;
    } else {      // v^((q^3 - 1)(q + 1))
      Fqk a = frob(v, 3, t) * vi, ai = frob(vi, 3, t) * v;
      first = frob(a, 1, t) * a;
      first_inv = frob(ai, 1, t) * ai;
    }
    return frob(first, 1, t) * pow_w0(PC::W0_NEG ? first_inv : first);
  }
};

#define PCD_DEF_PAIRING(NAME, G1T, G2T, PFX)                                                                   \
  struct NAME {                                                                                                 \
    typedef G1T G1;                                                                                             \
    typedef G2T G2;                                                                                             \
    static constexpr int LOOP_BITS = PFX##_ATE_LOOP_BITS;                                                       \
    static constexpr bool LOOP_NEG = PFX##_ATE_NEG;                                                             \
    static constexpr int W0_BITS = PFX##_W0_BITS;                                                               \
    static constexpr bool W0_NEG = PFX##_W0_NEG;                                                                \
    static constexpr unsigned NR = PFX##_NR_SMALL;                                                              \
    PCD_HD static uint32_t loop(int i) { constexpr uint32_t m[PFX##_ATE_LOOP_NLIMBS] = PFX##_ATE_LOOP; return m[i]; } \
    PCD_HD static uint32_t w0(int i) { constexpr uint32_t m[PFX##_W0_NLIMBS] = PFX##_W0; return m[i]; }          \
  };
PCD_DEF_PAIRING(PC_MNT4_298, G1_MNT4_298_C, G2_MNT4_298_C, PCD_MNT4_298)  // compact field variant: single-lane kernels
PCD_DEF_PAIRING(PC_MNT6_298, G1_MNT6_298_C, G2_MNT6_298_C, PCD_MNT6_298)
PCD_DEF_PAIRING(PC_MNT4_753, G1_MNT4_753, G2_MNT4_753, PCD_MNT4_753)
PCD_DEF_PAIRING(PC_MNT6_753, G1_MNT6_753, G2_MNT6_753, PCD_MNT6_753)

}  // namespace pcd
