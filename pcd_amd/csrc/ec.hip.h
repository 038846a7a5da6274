// Device short-Weierstrass Jacobian arithmetic (a != 0) for MNT4/MNT6 G1 and G2.
//
// Device counterpart of ark-ec `short_weierstrass_jacobian` as used by
// `VariableBaseMSM::multi_scalar_mul` under SNARK::prove (/root/reference
// src/ec_cycle_pcd/mod.rs:171,179).  Formulas are the EFD ones upstream uses (madd-2007-bl,
// add-2007-bl, dbl-2007-bl); results are compared with the oracle on affine coordinates only,
// since projective representatives are not unique.
//
// Affine infinity is encoded as literal zeros (0, 0) on device: (0,0) is not on any of the eight curves
// (b != 0), and pcdhip_bases_upload rewrites flagged points to it.  Coordinates live in the device-internal
// field image (fp.hip.h); from_abi / to_abi convert at the C-ABI.
#pragma once
#include <type_traits>
#include "fp.hip.h"

namespace pcd {

template <class F>
struct Jac {
  F X, Y, Z;
  PCD_HD static Jac infinity() { return {F::zero(), F::one(), F::zero()}; }
  PCD_HD bool is_inf() const { return Z.is_zero(); }
  static constexpr int WORDS = 3 * F::WORDS;          // device-internal image
  static constexpr int ABI_WORDS = 3 * F::ABI_WORDS;  // C-ABI image (X || Y || Z, upstream Montgomery limbs)
  PCD_HD static Jac load(const uint32_t* p) { return {F::load(p), F::load(p + F::WORDS), F::load(p + 2 * F::WORDS)}; }
  PCD_HD void store(uint32_t* p) const { X.store(p); Y.store(p + F::WORDS); Z.store(p + 2 * F::WORDS); }
  PCD_HD static Jac from_abi(const uint32_t* w) { return {F::from_abi(w), F::from_abi(w + F::ABI_WORDS), F::from_abi(w + 2 * F::ABI_WORDS)}; }
  PCD_HD void to_abi(uint32_t* w) const { X.to_abi(w); Y.to_abi(w + F::ABI_WORDS); Z.to_abi(w + 2 * F::ABI_WORDS); }
};

template <class F>
struct Aff {
  F x, y;
  static constexpr int WORDS = 2 * F::WORDS;
  static constexpr int ABI_WORDS = 2 * F::ABI_WORDS;
  PCD_HD bool is_inf() const { return x.is_raw_zero() && y.is_raw_zero(); }  // (0,0) is stored as literal zeros
  PCD_HD static Aff load(const uint32_t* p) { return {F::load(p), F::load(p + F::WORDS)}; }
  PCD_HD void store(uint32_t* p) const { x.store(p); y.store(p + F::WORDS); }
  PCD_HD static Aff from_abi(const uint32_t* w) { return {F::from_abi(w), F::from_abi(w + F::ABI_WORDS)}; }
  PCD_HD void to_abi(uint32_t* w) const { x.to_abi(w); y.to_abi(w + F::ABI_WORDS); }
};

// ------------------------------------------------------------------------------------------------ group configs
// G: coordinate field F, scalar-field parameters FR, and multiplication by the curve coefficient a.
// INL selects the inlined / compact variant of the field arithmetic (fp.hip.h); the memory image is the same.
template <class FQ, class FRP, unsigned A, int CURVE, bool INL = (FQ::N <= 11)>
struct G1Cfg {
  typedef Fp<FQ, INL> F;
  typedef FRP FR;
  static constexpr int CURVE_ID = CURVE;
  static constexpr int GROUP = 1;
  PCD_HD static F mul_by_a(const F& x) { return x.mul_small(A); }
};
template <class FQ, class FRP, unsigned A, unsigned NR, int CURVE, bool INL = (FQ::N <= 11)>
struct G2Cfg2 {  // twist over Fq2: a' = (a*nr, 0)
  typedef Fp2<Fp<FQ, INL>, NR> F;
  typedef FRP FR;
  static constexpr int CURVE_ID = CURVE;
  static constexpr int GROUP = 2;
  PCD_HD static F mul_by_a(const F& x) { return x.mul_small(A * NR); }
};
template <class FQ, class FRP, unsigned A, unsigned NR, int CURVE, bool INL = (FQ::N <= 11)>
struct G2Cfg3 {  // twist over Fq3: a' = (0, 0, a) = a u^2;  x u^2 = (nr c1, nr c2, c0)
  typedef Fp3<Fp<FQ, INL>, NR> F;
  typedef FRP FR;
  static constexpr int CURVE_ID = CURVE;
  static constexpr int GROUP = 2;
  PCD_HD static F mul_by_a(const F& x) { return {x.c1.mul_small(A * NR), x.c2.mul_small(A * NR), x.c0.mul_small(A)}; }
};

// lane-split form of the Fq2 twist (fp.hip.h Fp2S): used by the bucket accumulation of the 753-bit G2
template <class FQ, class FRP, unsigned A, unsigned NR, int CURVE, bool INL = false>
struct G2Cfg2S {
  typedef Fp2S<Fp<FQ, INL>, NR> F;
  typedef FRP FR;
  static constexpr int CURVE_ID = CURVE;
  static constexpr int GROUP = 2;
  PCD_DEV static F mul_by_a(const F& x) { return x.mul_small(A * NR); }
};
// lane-split form of the Fq3 twist (fp.hip.h Fp3S): bucket accumulation of the MNT6 G2 groups, three lanes per point
template <class FQ, class FRP, unsigned A, unsigned NR, int CURVE, bool INL = (FQ::N <= 11)>
struct G2Cfg3S {
  typedef Fp3S<Fp<FQ, INL>, NR> F;
  typedef FRP FR;
  static constexpr int CURVE_ID = CURVE;
  static constexpr int GROUP = 2;
  PCD_DEV static F mul_by_a(const F& x) { return x.mul_by_au2(A); }
};
// SplitOf<G>: the group configuration a throughput kernel should compute in, and how many adjacent lanes share one point
#ifndef PCD_SPLIT_FQ2_298
#define PCD_SPLIT_FQ2_298 0  // the 298-bit Fq2 twist unsplit (inlined arithmetic, one wave per SIMD) or split over lane pairs
#endif
template <class G> struct SplitOf { typedef G type; static constexpr int LANES = 1; };
template <class FQ, class FRP, unsigned A, unsigned NR, int CURVE>
struct SplitOf<G2Cfg2<FQ, FRP, A, NR, CURVE, false>> { typedef G2Cfg2S<FQ, FRP, A, NR, CURVE, false> type; static constexpr int LANES = 2; };
#if PCD_SPLIT_FQ2_298
template <class FQ, class FRP, unsigned A, unsigned NR, int CURVE>
struct SplitOf<G2Cfg2<FQ, FRP, A, NR, CURVE, true>> { typedef G2Cfg2S<FQ, FRP, A, NR, CURVE, true> type; static constexpr int LANES = 2; };
#endif
template <class FQ, class FRP, unsigned A, unsigned NR, int CURVE, bool INL>
struct SplitOf<G2Cfg3<FQ, FRP, A, NR, CURVE, INL>> { typedef G2Cfg3S<FQ, FRP, A, NR, CURVE, INL> type; static constexpr int LANES = 3; };
// AccOf<G>: the configuration msm_accumulate computes in -- SplitOf<G>, with the LDS-mailbox field variant (fp.hip.h, MB) where the
// products are calls on 27-word operands (64-lane workgroups: the mailbox is per lane of ONE wave)
#ifndef PCD_MAILBOX
#define PCD_MAILBOX 1
#endif
template <class FQ, class FRP, unsigned A, int CURVE>
struct G1CfgMB {
  typedef Fp<FQ, false, true> F;
  typedef FRP FR;
  static constexpr int CURVE_ID = CURVE;
  static constexpr int GROUP = 1;
  PCD_HD static F mul_by_a(const F& x) { return x.mul_small(A); }
};
template <class FQ, class FRP, unsigned A, unsigned NR, int CURVE>
struct G2Cfg2SMB {
  typedef Fp2S<Fp<FQ, false, true>, NR> F;
  typedef FRP FR;
  static constexpr int CURVE_ID = CURVE;
  static constexpr int GROUP = 2;
  PCD_DEV static F mul_by_a(const F& x) { return x.mul_small(A * NR); }
};
template <class FQ, class FRP, unsigned A, unsigned NR, int CURVE>
struct G2Cfg3SMB {
  typedef Fp3S<Fp<FQ, false, true>, NR> F;
  typedef FRP FR;
  static constexpr int CURVE_ID = CURVE;
  static constexpr int GROUP = 2;
  PCD_DEV static F mul_by_a(const F& x) { return x.mul_by_au2(A); }
};
template <class G> struct AccOf { typedef typename SplitOf<G>::type type; static constexpr int LANES = SplitOf<G>::LANES; };
// (tried and left off: the 298-bit Fq2 twist accumulated in the split mailbox form at 2 / 3 waves per SIMD instead of inlined at one --
//  same-box A/B at 2^20: accumulation 6.33 -> 7.21 / 7.75 ms; an 11-limb product is too short to carry a call and six LDS posts)
#ifndef PCD_FQ2_298_MB
#define PCD_FQ2_298_MB 0
#endif
#if PCD_FQ2_298_MB
template <class FQ, class FRP, unsigned A, unsigned NR, int CURVE>
struct AccOf<G2Cfg2<FQ, FRP, A, NR, CURVE, true>> { typedef G2Cfg2SMB<FQ, FRP, A, NR, CURVE> type; static constexpr int LANES = 2; };
#endif
template <class FQ, class FRP, unsigned A, unsigned NR, int CURVE>
struct AccOf<G2Cfg2<FQ, FRP, A, NR, CURVE, false>> {
  typedef typename std::conditional<(PCD_MAILBOX && FQ::N > 11), G2Cfg2SMB<FQ, FRP, A, NR, CURVE>, G2Cfg2S<FQ, FRP, A, NR, CURVE, false>>::type type;
  static constexpr int LANES = 2;
};
template <class FQ, class FRP, unsigned A, unsigned NR, int CURVE>
struct AccOf<G2Cfg3<FQ, FRP, A, NR, CURVE, false>> {
  typedef typename std::conditional<(PCD_MAILBOX && FQ::N > 11), G2Cfg3SMB<FQ, FRP, A, NR, CURVE>, G2Cfg3S<FQ, FRP, A, NR, CURVE, false>>::type type;
  static constexpr int LANES = 3;
};
template <class FQ, class FRP, unsigned A, int CURVE>
struct AccOf<G1Cfg<FQ, FRP, A, CURVE, false>> {
  typedef typename std::conditional<(PCD_MAILBOX && FQ::N > 11), G1CfgMB<FQ, FRP, A, CURVE>, G1Cfg<FQ, FRP, A, CURVE, false>>::type type;
  static constexpr int LANES = 1;
};
// SplitOfTail<G>: the same choice for the latency-bound kernels behind the accumulation (pieces, bucket reduction, combine): there
// two lanes per point halve the latency of every level, so the inlined 298-bit Fq2 group is split as well
// (and the 753-bit groups compute in their mailbox variants there too: all of these kernels run 64-lane workgroups)
#ifndef PCD_MAILBOX_TAIL
#define PCD_MAILBOX_TAIL 1
#endif
template <class G> struct SplitOfTail {
  typedef typename std::conditional<(PCD_MAILBOX_TAIL != 0 && AccOf<G>::LANES == SplitOf<G>::LANES), typename AccOf<G>::type, typename SplitOf<G>::type>::type type;
  static constexpr int LANES = SplitOf<G>::LANES;
};
// (PCD_MAILBOX_TAIL_FQ3: the Fq3-753 reduction kernels in the mailbox form too.  Round 2 kept them on the plain form because ONE kernel
//  returned a wrong limb there: msm_merge_ones_kernel -- one item, three active lanes, both operands finite; localised with
//  tools/probe_fq3_tail.py, not cured by keeping every item slot of the wave active, gone when the result is canonicalised before the
//  store: code generation, not arithmetic.  That kernel and msm_horner_kernel (the other single-item kernel) now always compute in the
//  plain lane-split form -- one addition per MSM, the memory image is the same -- and with that every probe case, the edge-case tests
//  on all eight groups and the at-size tests are exact in the mailbox form: Fq3-753 MSM at 2^15 20.8 -> 17.5 ms.)
#ifndef PCD_MAILBOX_TAIL_FQ3
#define PCD_MAILBOX_TAIL_FQ3 1
#endif
#if !PCD_MAILBOX_TAIL_FQ3
template <class FQ, class FRP, unsigned A, unsigned NR, int CURVE, bool INL>
struct SplitOfTail<G2Cfg3<FQ, FRP, A, NR, CURVE, INL>> : SplitOf<G2Cfg3<FQ, FRP, A, NR, CURVE, INL>> {};
#endif
template <class FQ, class FRP, unsigned A, unsigned NR, int CURVE>
struct SplitOfTail<G2Cfg2<FQ, FRP, A, NR, CURVE, true>> { typedef G2Cfg2S<FQ, FRP, A, NR, CURVE, true> type; static constexpr int LANES = 2; };

typedef G1Cfg<F298A, F298B, PCD_MNT4_298_A_SMALL, 0> G1_MNT4_298;
typedef G1Cfg<F298B, F298A, PCD_MNT6_298_A_SMALL, 1> G1_MNT6_298;
typedef G1Cfg<F753A, F753B, PCD_MNT4_753_A_SMALL, 2> G1_MNT4_753;
typedef G1Cfg<F753B, F753A, PCD_MNT6_753_A_SMALL, 3> G1_MNT6_753;
typedef G2Cfg2<F298A, F298B, PCD_MNT4_298_A_SMALL, PCD_MNT4_298_NR_SMALL, 0> G2_MNT4_298;
typedef G2Cfg3<F298B, F298A, PCD_MNT6_298_A_SMALL, PCD_MNT6_298_NR_SMALL, 1> G2_MNT6_298;
typedef G2Cfg2<F753A, F753B, PCD_MNT4_753_A_SMALL, PCD_MNT4_753_NR_SMALL, 2> G2_MNT4_753;
typedef G2Cfg3<F753B, F753A, PCD_MNT6_753_A_SMALL, PCD_MNT6_753_NR_SMALL, 3> G2_MNT6_753;
// compact variants (non-inlined field arithmetic) for the latency-bound single-lane kernels
typedef G1Cfg<F298A, F298B, PCD_MNT4_298_A_SMALL, 0, false> G1_MNT4_298_C;
typedef G1Cfg<F298B, F298A, PCD_MNT6_298_A_SMALL, 1, false> G1_MNT6_298_C;
typedef G2Cfg2<F298A, F298B, PCD_MNT4_298_A_SMALL, PCD_MNT4_298_NR_SMALL, 0, false> G2_MNT4_298_C;
typedef G2Cfg3<F298B, F298A, PCD_MNT6_298_A_SMALL, PCD_MNT6_298_NR_SMALL, 1, false> G2_MNT6_298_C;

// fields whose mixed addition has the lazily reduced form (EC::madd_lz): the inlined 298-bit prime fields
template <class F> struct LazyCapable { static constexpr bool value = false; };
template <class P> struct LazyCapable<Fp<P, true>> { static constexpr bool value = (P::N <= 11); };

// extension fields whose XYZZ mixed addition has lazily reduced internals (EC::madd_x_lz2): Fq2 over the inlined 298-bit prime fields
#ifndef PCD_LAZY_FQ2
#define PCD_LAZY_FQ2 1
#endif
template <class F> struct LazyFq2 { static constexpr bool value = false; };
template <class P, unsigned NR> struct LazyFq2<Fp2<Fp<P, true>, NR>> { static constexpr bool value = PCD_LAZY_FQ2 && (P::N <= 11); };

// ------------------------------------------------------------------------------------------------ group law
template <class G>
struct EC {
  typedef typename G::F F;
  typedef Jac<F> J;
  typedef Aff<F> A;

  // dbl-2007-bl
  PCD_HD static J dbl(const J& p) {
    if (p.is_inf()) return p;
    F XX = p.X.sqr(), YY = p.Y.sqr(), YYYY = YY.sqr(), ZZ = p.Z.sqr();
    F S = ((p.X + YY).sqr() - XX - YYYY).dbl();
    F M = XX.dbl() + XX + G::mul_by_a(ZZ.sqr());
    F T = M.sqr() - S.dbl();
    J r;
    r.X = T;
    r.Y = M * (S - T) - YYYY.dbl().dbl().dbl();
    r.Z = (p.Y + p.Z).sqr() - YY - ZZ;
    return r;
  }
  // madd-2007-bl; q = (0,0) is the identity
  PCD_HD static J madd(const J& p, const A& q) {
    if (q.is_inf()) return p;
    if (p.is_inf()) return {q.x, q.y, F::one()};
    F Z1Z1 = p.Z.sqr();
    F U2 = q.x * Z1Z1;
    F S2 = q.y * p.Z * Z1Z1;
    F H = U2 - p.X;
    F r = S2 - p.Y;
    if (H.is_zero()) return r.is_zero() ? dbl(p) : J::infinity();  // same point / opposite points
    F HH = H.sqr();
    F I = HH.dbl().dbl();
    F Jv = H * I;
    r = r.dbl();
    F V = p.X * I;
    J o;
    o.X = r.sqr() - Jv - V.dbl();
    o.Y = r * (V - o.X) - (p.Y * Jv).dbl();
    o.Z = (p.Z + H).sqr() - Z1Z1 - HH;
    return o;
  }
  // add-2007-bl
  PCD_HD static J add(const J& p, const J& q) {
    if (p.is_inf()) return q;
    if (q.is_inf()) return p;
    F Z1Z1 = p.Z.sqr(), Z2Z2 = q.Z.sqr();
    F U1 = p.X * Z2Z2, U2 = q.X * Z1Z1;
    F S1 = p.Y * q.Z * Z2Z2, S2 = q.Y * p.Z * Z1Z1;
    F H = U2 - U1;
    F r = S2 - S1;
    if (H.is_zero()) return r.is_zero() ? dbl(p) : J::infinity();
    F I = H.dbl().sqr();
    F Jv = H * I;
    r = r.dbl();
    F V = U1 * I;
    J o;
    o.X = r.sqr() - Jv - V.dbl();
    o.Y = r * (V - o.X) - (S1 * Jv).dbl();
    o.Z = ((p.Z + q.Z).sqr() - Z1Z1 - Z2Z2) * H;
    return o;
  }
  PCD_HD static J neg(const J& p) { return {p.X, p.Y.neg(), p.Z}; }
  // affine (x, y) or (0, 0) for the identity
  PCD_HD static A to_affine(const J& p) {
    if (p.is_inf()) return {F::zero(), F::zero()};
    F zi = p.Z.inv(), zi2 = zi.sqr();
    return {p.X * zi2, p.Y * zi2 * zi};
  }
  // k given as `nwords` canonical little-endian u32 words
  PCD_HD static J mul(const J& p, const uint32_t* k, int nwords) {
    J r = J::infinity();
    for (int i = nwords * 32 - 1; i >= 0; i--) {
      r = dbl(r);
      if ((k[i >> 5] >> (i & 31)) & 1) r = add(r, p);
    }
    return r;
  }
  // ---- mixed addition in XYZZ coordinates (x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2; EFD madd-2008-s) ------------------------------
  // What the bucket accumulation of every group runs on: 8M + 2S and 7 field additions / subtractions against 7M + 4S and 14 for the
  // Jacobian madd-2007-bl (in Fq2: 58 N^2 multiply-adds instead of 62 N^2 and half the carry-chain work).  The identity is ZZ = 0.
  struct AccX {
    F X, Y, ZZ, ZZZ;
    PCD_HD bool is_inf() const { return ZZ.is_zero(); }
  };
  static constexpr int ACCX_WORDS = 4 * F::WORDS;  // flushed record: X || Y || ZZ || ZZZ
  PCD_HD static AccX x_infinity() { return {F::zero(), F::one(), F::zero(), F::zero()}; }
  PCD_HD static AccX x_from(const J& p) {
    if (p.is_inf()) return x_infinity();
    const F zz = p.Z.sqr();
    return {p.X, p.Y, zz, zz * p.Z};
  }
  // the Jacobian point (X ZZ : Y ZZZ : ZZ): x = X ZZ / ZZ^2, y = Y ZZZ / ZZ^3 since ZZ^3 = ZZZ^2
  PCD_HD static J x_to_jac(const AccX& a) {
    if (a.is_inf()) return J::infinity();
    return {a.X * a.ZZ, a.Y * a.ZZZ, a.ZZ};
  }
  PCD_HD static AccX madd_x(const AccX& p, const A& q) {
    if constexpr (LazyFq2<F>::value) return madd_x_lz2(p, q);
    else return madd_x_plain(p, q);
  }
  PCD_HD static AccX madd_x_plain(const AccX& p, const A& q) {
    if (q.is_inf()) return p;
    if (p.is_inf()) return {q.x, q.y, F::one(), F::one()};
    const F U2 = q.x * p.ZZ;
    const F S2 = q.y * p.ZZZ;
    const F P = U2 - p.X;
    const F R = S2 - p.Y;
    if (P.is_zero()) return R.is_zero() ? x_from(dbl(x_to_jac(p))) : x_infinity();  // same point / opposite points
    const F PP = P.sqr();
    const F PPP = P * PP;
    const F Q = p.X * PP;
    AccX o;
    o.X = R.sqr() - PPP - Q.dbl();
    o.Y = R * (Q - o.X) - p.Y * PPP;
    o.ZZ = p.ZZ * PP;
    o.ZZZ = p.ZZZ * PPP;
    return o;
  }

  // ---- the same addition over Fq2 = Fq[u]/(u^2 - nr) of the 298-bit curves with lazily reduced INTERNALS -----------------------------
  // All four coordinates stay reduced between steps (unlike G1's madd_lz: a coefficient of a product is a0 b0 + nr a1 b1 with nr = 17
  // or 13, and R'/p = 2^10 of headroom then allows operand bounds with ca cb <= 1024 / (nr + 1) only, i.e. differences may carry 4p of
  // bias, not 16p), but every addition / subtraction inside a step is limb-wise, the multiplications by the non-residue are a carrying
  // scale of the operand that is shared by the products using it (6 per step instead of one per product: 10), X3 = R^2 - PPP - 2Q is ONE
  // reduction of a signed limb sum, and Y3 = R (Q - X3) - Y1 PPP is one four-term dot product per coefficient.  Bounds (multiples of p):
  //   U2, S2, PP, RR, PPP, Q < 2      P = U2 - X1 + 4 in (2, 6)   R = S2 - Y1 + 4 in (2, 6)   t = Q - X3 + 4 in (2, 6)   Y1' = 4 - Y1 in (2, 4]
  //   PP.c0 = P0^2 + nr P1^2: 36 (nr + 1) <= 648   Y3.c0 = R0 t0 + nr R1 t1 + Y0' PPP0 + nr Y1' PPP1: (36 + 8) (nr + 1) <= 792
  // 56 N^2 multiply-adds against 58 N^2, and about 2 500 other instructions against 4 800 (N = 11).
  typedef typename F::Base B_;
  template <class FF = F>
  PCD_HD static AccX madd_x_lz2(const AccX& p, const A& q) {
    typedef typename B_::Lz L;
    constexpr int32_t NRV = (int32_t)FF::NONRESIDUE;
    if (q.is_inf()) return p;
    if (p.is_inf()) return {q.x, q.y, F::one(), F::one()};
    auto mulr = [](const L& a0, const L& a1, const L& b0, const L& b1, const L& nb1) {  // (a0 + a1 u)(b0 + b1 u), nb1 = nr b1
      FF o; o.c0 = B_::lz_dot2(a0, b0, a1, nb1); o.c1 = B_::lz_dot2(a0, b1, a1, b0); return o; };
    const L zz0 = p.ZZ.c0.lz(), zz1 = p.ZZ.c1.lz(), zzz0 = p.ZZZ.c0.lz(), zzz1 = p.ZZZ.c1.lz();
    const L nzz1 = B_::lz_scale_carry(zz1, NRV), nzzz1 = B_::lz_scale_carry(zzz1, NRV);
    const FF U2 = mulr(q.x.c0.lz(), q.x.c1.lz(), zz0, zz1, nzz1);
    const FF S2 = mulr(q.y.c0.lz(), q.y.c1.lz(), zzz0, zzz1, nzzz1);
    const L P0 = B_::lz_carry(B_::template lz_sub<0>(U2.c0.lz(), p.X.c0.lz())), P1 = B_::lz_carry(B_::template lz_sub<0>(U2.c1.lz(), p.X.c1.lz()));
    const L R0 = B_::lz_carry(B_::template lz_sub<0>(S2.c0.lz(), p.Y.c0.lz())), R1 = B_::lz_carry(B_::template lz_sub<0>(S2.c1.lz(), p.Y.c1.lz()));
    const L nP1 = B_::lz_scale_carry(P1, NRV), nR1 = B_::lz_scale_carry(R1, NRV);
    FF PP, RR;
    PP.c0 = B_::lz_dot2(P0, P0, P1, nP1); PP.c1 = B_::lz_mul(B_::lz_shl(P0, 1), P1);
    RR.c0 = B_::lz_dot2(R0, R0, R1, nR1); RR.c1 = B_::lz_mul(B_::lz_shl(R0, 1), R1);
    if (PP.is_zero()) return RR.is_zero() ? x_from(dbl(x_to_jac(p))) : x_infinity();  // same x: the same point, or opposite points
    const L pp0 = PP.c0.lz(), pp1 = PP.c1.lz(), npp1 = B_::lz_scale_carry(pp1, NRV);
    const FF PPP = mulr(P0, P1, pp0, pp1, npp1);
    const FF Q = mulr(p.X.c0.lz(), p.X.c1.lz(), pp0, pp1, npp1);
    AccX o;
    {  // X3 = R^2 - PPP - 2Q, reduced: one signed limb sum per coefficient
      int64_t s0[B_::N], s1[B_::N];
#pragma unroll
      for (int i = 0; i < B_::N; i++) {
        s0[i] = (int64_t)RR.c0.v[i] - (int64_t)PPP.c0.v[i] - 2 * (int64_t)Q.c0.v[i];
        s1[i] = (int64_t)RR.c1.v[i] - (int64_t)PPP.c1.v[i] - 2 * (int64_t)Q.c1.v[i];
      }
      o.X.c0 = B_::from_signed_sum(s0);
      o.X.c1 = B_::from_signed_sum(s1);
    }
    const L t0 = B_::lz_carry(B_::template lz_sub<0>(Q.c0.lz(), o.X.c0.lz())), t1 = B_::lz_carry(B_::template lz_sub<0>(Q.c1.lz(), o.X.c1.lz()));
    const L y0n = B_::template lz_sub<0>(B_::zero().lz(), p.Y.c0.lz()), y1n = B_::template lz_sub<0>(B_::zero().lz(), p.Y.c1.lz());
    const L ppp0 = PPP.c0.lz(), ppp1 = PPP.c1.lz(), nppp1 = B_::lz_scale_carry(ppp1, NRV);
    o.Y.c0 = B_::lz_dot4(R0, t0, nR1, t1, y0n, ppp0, y1n, nppp1);
    o.Y.c1 = B_::lz_dot4(R0, t1, R1, t0, y0n, ppp1, y1n, ppp0);
    o.ZZ = mulr(zz0, zz1, pp0, pp1, npp1);
    o.ZZZ = mulr(zzz0, zzz1, ppp0, ppp1, nppp1);
    return o;
  }

  // ---- mixed addition with lazily reduced coordinates (G1 of the 298-bit curves: F = Fp with the Lz helpers of fp.hip.h) ----
  // The bucket accumulation is a long chain acc <- acc + P_i.  The accumulator lives in XYZZ coordinates (x = X / ZZ, y = Y / ZZZ,
  // ZZ^3 = ZZZ^2; EFD madd-2008-s: 8M + 2S against 7M + 4S for the Jacobian madd-2007-bl, and no (Z1 + H)^2 trick to unfold) with
  // X kept UNREDUCED between steps:
  //   X  carry-propagated limbs, value < 16p      Y, ZZ, ZZZ  in [0, 2p) as usual
  // and every addition / subtraction of the formula is limb-wise (no carry chain, no reduction); the products absorb it.
  //   U2 = x2 ZZ1   S2 = y2 ZZZ1   P = U2 - X1   R = S2 - Y1   PP = P^2   PPP = P PP   Q = X1 PP
  //   X3 = R^2 - PPP - 2Q   Y3 = R (Q - X3) - Y1 PPP (one fused two-term product)   ZZ3 = ZZ1 PP   ZZZ3 = ZZZ1 PPP
  // 6 products of 2 N^2 mads, 2 squares, one fused two-term product of 3 N^2: 2 189 mads (N = 11) against 2 376 for the lazily
  // reduced Jacobian form this replaces.  Bounds (value as a multiple of p | needed for a product: ca cb <= 1024; limb products
  // per column < 2^63):
  //   P = U2 - X1 + 16p < 18p (carried)   R = S2 - Y1 + 4p < 6p (carried)   PP, R^2, PPP, Q < 2p
  //   X3 = R^2 - PPP - 2Q + 8p in (2p, 10p) (carried)   t = Q - X3 + 16p in (6p, 18p), limbs in (-2^28, 1.25 * 2^30)
  //   Y1' = 4p - Y1 in (2p, 4p]   Y3 = R t + Y1' PPP: 6 * 18 + 4 * 2 = 116;  columns < 11 (2^28 * 1.25 * 2^30 + 2^29 * 2^28) + 11 * 2^56 < 2^62
  struct AccLz {
    typename F::Lz X;
    F Y, ZZ, ZZZ;
    bool inf;
  };
  static constexpr int ACC_WORDS = 4 * F::WORDS;  // record the accumulation flushes: X (unreduced limbs) || Y || ZZ || ZZZ, identity = ZZ 0
  PCD_HD static AccLz lz_infinity() { AccLz a; a.X = F::zero().lz(); a.Y = F::one(); a.ZZ = F::zero(); a.ZZZ = F::zero(); a.inf = true; return a; }
  PCD_HD static AccLz lz_from(const J& p) {
    AccLz a;
    a.inf = p.is_inf();
    a.X = p.X.lz(); a.Y = p.Y; a.ZZ = p.Z.sqr(); a.ZZZ = a.ZZ * p.Z;
    return a;
  }
  // fully reduced Jacobian point (X ZZ : Y ZZZ : ZZ) -- x = X ZZ / ZZ^2, y = Y ZZZ / ZZ^3 since ZZ^3 = ZZZ^2
  PCD_HD static J lz_to_jac(const AccLz& a) {
    if (a.inf) return J::infinity();
    return {F::lz_mul(a.X, a.ZZ.lz()), a.Y * a.ZZZ, a.ZZ};
  }
  PCD_HD static AccLz madd_lz(const AccLz& p, const A& q) {
    typedef typename F::Lz L;
    if (q.is_inf()) return p;
    if (p.inf) { AccLz o; o.X = q.x.lz(); o.Y = q.y; o.ZZ = F::one(); o.ZZZ = F::one(); o.inf = false; return o; }
    const L zz = p.ZZ.lz(), zzz = p.ZZZ.lz();
    const F U2 = F::lz_mul(q.x.lz(), zz);
    const F S2 = F::lz_mul(q.y.lz(), zzz);
    const L P = F::lz_carry(F::template lz_sub<2>(U2.lz(), p.X));
    const L R = F::lz_carry(F::template lz_sub<0>(S2.lz(), p.Y.lz()));
    const F PP = F::lz_sqr(P);
    const F RR = F::lz_sqr(R);
    if (PP.is_zero()) {  // same x: the same point (double it the ordinary way) or opposite points
      if (!RR.is_zero()) return lz_infinity();
      return lz_from(dbl(lz_to_jac(p)));
    }
    const L pp = PP.lz();
    const F PPP = F::lz_mul(P, pp);
    const F Q = F::lz_mul(p.X, pp);
    AccLz o;
    o.inf = false;
    {  // X3 = R^2 - PPP - 2Q + 8p
      L t;
#pragma unroll
      for (int i = 0; i < F::N; i++)
        t.v[i] = (int32_t)RR.v[i] - (int32_t)PPP.v[i] - (int32_t)(Q.v[i] << 1) + (int32_t)(F::Params::mod4(i) << 1);
      o.X = F::lz_carry(t);
    }
    const L t2 = F::template lz_sub<2>(Q.lz(), o.X);
    const L y1n = F::template lz_sub<0>(F::zero().lz(), p.Y.lz());
    const L ppp = PPP.lz();
    o.Y = F::lz_dot2(R, t2, y1n, ppp);
    o.ZZ = F::lz_mul(zz, pp);
    o.ZZZ = F::lz_mul(zzz, ppp);
    return o;
  }

  // The same addition with the accumulator held OUTSIDE the register file (ST: four slots of N words per lane -- msm.hip.h keeps them in LDS):
  // every coordinate is fetched when a product needs it and written back as soon as its new value exists, so at most one of the four is in
  // registers at a time.  The point of it: the 298-bit G1 accumulation needs 206 registers with the accumulator resident -- two waves per
  // SIMD; without it the working set is P, R, PP, RR, PPP, Q, one operand and the product's temporaries (experiment PCD_ACC_LDS).
  // ST: ld(slot) -> F (as stored), st(slot, F), bool& inf().  Slot 0 holds X's unreduced limbs (as uint32 images of the signed limbs).
  template <class ST>
  PCD_HD static void madd_lz_st(ST& st, const A& q) {
    typedef typename F::Lz L;
    if (q.is_inf()) return;
    if (st.inf()) { st.st(0, q.x); st.st(1, q.y); st.st(2, F::one()); st.st(3, F::one()); st.inf() = false; return; }
    auto as_lz = [](const F& f) { L l;
#pragma unroll
      for (int i = 0; i < F::N; i++) l.v[i] = (int32_t)f.v[i];
      return l; };
    auto as_f = [](const L& l) { F f;
#pragma unroll
      for (int i = 0; i < F::N; i++) f.v[i] = (uint32_t)l.v[i];
      return f; };
    const F U2 = F::lz_mul(q.x.lz(), st.ld(2).lz());
    const F S2 = F::lz_mul(q.y.lz(), st.ld(3).lz());
    const L P = F::lz_carry(F::template lz_sub<2>(U2.lz(), as_lz(st.ld(0))));
    const L R = F::lz_carry(F::template lz_sub<0>(S2.lz(), st.ld(1).lz()));
    const F PP = F::lz_sqr(P);
    const F RR = F::lz_sqr(R);
    if (PP.is_zero()) {  // same x: the same point (double it the ordinary way) or opposite points -- through the ordinary record
      AccLz p;
      p.X = as_lz(st.ld(0)); p.Y = st.ld(1); p.ZZ = st.ld(2); p.ZZZ = st.ld(3); p.inf = false;
      AccLz o = RR.is_zero() ? lz_from(dbl(lz_to_jac(p))) : lz_infinity();
      st.st(0, as_f(o.X)); st.st(1, o.Y); st.st(2, o.ZZ); st.st(3, o.ZZZ); st.inf() = o.inf;
      return;
    }
    const L pp = PP.lz();
    const F PPP = F::lz_mul(P, pp);
    const F Q = F::lz_mul(as_lz(st.ld(0)), pp);
    L x3;
    {  // X3 = R^2 - PPP - 2Q + 8p
      L t;
#pragma unroll
      for (int i = 0; i < F::N; i++)
        t.v[i] = (int32_t)RR.v[i] - (int32_t)PPP.v[i] - (int32_t)(Q.v[i] << 1) + (int32_t)(F::Params::mod4(i) << 1);
      x3 = F::lz_carry(t);
    }
    const L t2 = F::template lz_sub<2>(Q.lz(), x3);
    st.st(0, as_f(x3));
    const L ppp = PPP.lz();
    {
      const L y1n = F::template lz_sub<0>(F::zero().lz(), st.ld(1).lz());
      st.st(1, F::lz_dot2(R, t2, y1n, ppp));
    }
    st.st(2, F::lz_mul(st.ld(2).lz(), pp));
    st.st(3, F::lz_mul(st.ld(3).lz(), ppp));
  }

  // the same product with a fixed 4-bit window (one lane: 14 group operations for the table, then 4 doublings and at
  // most one addition per nibble, leading zero nibbles skipped); `table` = 15 caller-provided Jacobian slots
  PCD_HD static J mul_w4(const J& p, const uint32_t* k, int nwords, J* table) {
    table[0] = p;
    for (int i = 1; i < 15; i++) table[i] = (i & 1) ? dbl(table[i >> 1]) : add(table[i - 1], p);
    int top = nwords * 8 - 1;
    while (top >= 0 && ((k[top >> 3] >> ((top & 7) * 4)) & 15) == 0) top--;
    J r = J::infinity();
    for (int i = top; i >= 0; i--) {
      if (i != top) { r = dbl(r); r = dbl(r); r = dbl(r); r = dbl(r); }
      const uint32_t nib = (k[i >> 3] >> ((i & 7) * 4)) & 15;
      if (nib) r = (i == top) ? table[nib - 1] : add(r, table[nib - 1]);
    }
    return r;
  }
};

// ------------------------------------------------------------------------------------------------ two lanes per group operation
// The latency-bound levels of the bucket reduction run ONE addition (or doubling) per handful of lanes; in one lane that is 16 (9)
// dependent field products.  add-2007-bl has two independent operand chains (Z1Z1, U2, S2 | Z2Z2, U1, S1) and pairs of independent
// products after them, dbl-2007-bl likewise: here an even / odd lane pair shares one operation, each lane computing one product of
// every slot with role-selected operands and handing results across with `ds_swizzle`-class shuffles (`__shfl_xor 1`).  8 product
// slots instead of 16 products for an addition, 5 instead of 9 for a doubling.  Both lanes pass the same points and both return the
// whole result.  Prime-field groups only (the extension-field groups are already spread over lanes at the field level).
// (Round 6: the lane-split extension-field groups take the same form with an item of 2 L lanes, L = 2 / 3 lanes per point: the first L
//  lanes are the "even" half, the next L the "odd" one, each half a whole lane-split point, products collective inside a half as before,
//  results handed across at distance L.  Their pair levels were the chain the help proofs wait for: DESIGN.md section 4.)
template <class F, class = void> struct HalfLanes { static constexpr int L = 1; };
template <class F> struct HalfLanes<F, typename std::enable_if<(F::LANES > 1)>::type> { static constexpr int L = F::LANES; };
template <class G>
struct EC2 {
  typedef typename G::F F;
  typedef Jac<F> J;
  typedef EC<G> E;
  static constexpr int L = HalfLanes<F>::L;   // lanes per point; an item is 2 L adjacent lanes starting at a multiple of 2 L
  PCD_DEV static bool odd() { return (((threadIdx.x & 63u) / (unsigned)L) & 1u) != 0; }
  template <class B> PCD_DEV static B xch_base(const B& a) { B r;
    if constexpr (L == 1) {
#pragma unroll
      for (int i = 0; i < B::N; i++) r.v[i] = (uint32_t)__shfl_xor((int)a.v[i], 1, 64);
    } else {
      const int partner = (int)(threadIdx.x & 63u) + (odd() ? -L : L);
#pragma unroll
      for (int i = 0; i < B::N; i++) r.v[i] = (uint32_t)__shfl((int)a.v[i], partner, 64);
    }
    return r; }
  template <class B> PCD_DEV static B sel_base(bool t, const B& a, const B& b) { B r;
#pragma unroll
    for (int i = 0; i < B::N; i++) r.v[i] = t ? a.v[i] : b.v[i];
    return r; }
  PCD_DEV static F xch(const F& a) { if constexpr (L == 1) return xch_base(a); else return F{xch_base(a.c)}; }
  PCD_DEV static F sel(bool t, const F& a, const F& b) { if constexpr (L == 1) return sel_base(t, a, b); else return F{sel_base(t, a.c, b.c)}; }
  // one product slot: the even lane computes ea * eb, the odd lane oa * ob
  PCD_DEV static F slot(const F& ea, const F& eb, const F& oa, const F& ob) { const bool o = odd(); return sel(o, oa, ea) * sel(o, ob, eb); }

  PCD_DEV static J dbl2(const J& p) {
    if (p.is_inf()) return p;
    const bool o = odd();
    const F s1 = slot(p.X, p.X, p.Y, p.Y);            // even: XX          odd: YY
    const F s1x = xch(s1);
    const F XX = sel(o, s1x, s1), YY = sel(o, s1, s1x);
    const F s2 = slot(p.Z, p.Z, YY, YY);               // even: ZZ          odd: YYYY
    const F s2x = xch(s2);
    const F ZZ = sel(o, s2x, s2), YYYY = sel(o, s2, s2x);
    const F xyy = p.X + YY;
    const F s3 = slot(ZZ, ZZ, xyy, xyy);               // even: ZZ^2        odd: (X + YY)^2
    const F s3x = xch(s3);
    const F ZZ2 = sel(o, s3x, s3), XYY2 = sel(o, s3, s3x);
    const F S = (XYY2 - XX - YYYY).dbl();
    const F M = XX.dbl() + XX + G::mul_by_a(ZZ2);
    const F yz = p.Y + p.Z;
    const F s4 = slot(M, M, yz, yz);                   // even: M^2         odd: (Y + Z)^2
    const F s4x = xch(s4);
    const F MM = sel(o, s4x, s4), YZ2 = sel(o, s4, s4x);
    J r;
    r.X = MM - S.dbl();
    r.Z = YZ2 - YY - ZZ;
    r.Y = M * (S - r.X) - YYYY.dbl().dbl().dbl();     // (the last product is computed by both lanes)
    return r;
  }
  PCD_DEV static J add2(const J& p, const J& q) {
    if (p.is_inf()) return q;
    if (q.is_inf()) return p;
    const bool o = odd();
    const F s1 = slot(p.Z, p.Z, q.Z, q.Z);             // even: Z1Z1        odd: Z2Z2
    const F s1x = xch(s1);
    const F Z1Z1 = sel(o, s1x, s1), Z2Z2 = sel(o, s1, s1x);
    const F s2 = slot(p.X, Z2Z2, q.X, Z1Z1);           // even: U1          odd: U2
    const F s3 = slot(p.Y, q.Z, q.Y, p.Z);             // even: Y1 Z2       odd: Y2 Z1
    const F s4 = slot(s3, Z2Z2, s3, Z1Z1);             // even: S1          odd: S2
    const F s2x = xch(s2), s4x = xch(s4);
    const F U1 = sel(o, s2x, s2), U2 = sel(o, s2, s2x), S1 = sel(o, s4x, s4), S2 = sel(o, s4, s4x);
    const F H = U2 - U1;
    F r = S2 - S1;
    if (H.is_zero()) return r.is_zero() ? dbl2(p) : J::infinity();
    r = r.dbl();
    const F h2 = H.dbl(), zs = p.Z + q.Z;
    const F s5 = slot(h2, h2, zs, zs);                 // even: I = (2H)^2  odd: (Z1 + Z2)^2
    const F s5x = xch(s5);
    const F I = sel(o, s5x, s5), ZS = sel(o, s5, s5x);
    const F zc = ZS - Z1Z1 - Z2Z2;
    const F s6 = slot(H, I, zc, H);                    // even: J = H I     odd: Z3
    const F s7 = slot(U1, I, r, r);                    // even: V = U1 I    odd: r^2
    const F s6x = xch(s6), s7x = xch(s7);
    const F Jv = sel(o, s6x, s6), Z3 = sel(o, s6, s6x), V = sel(o, s7x, s7), rr = sel(o, s7, s7x);
    J out;
    out.X = rr - Jv - V.dbl();
    out.Z = Z3;
    const F vx = V - out.X;
    const F s8 = slot(S1, Jv, r, vx);                  // even: S1 J        odd: r (V - X3)
    const F s8x = xch(s8);
    const F S1J = sel(o, s8x, s8), rv = sel(o, s8, s8x);
    out.Y = rv - S1J.dbl();
    return out;
  }
};
// FOUR lanes per group operation (round 6; prime-field groups): a quad of adjacent lanes shares one addition or doubling.  The formulas are
// five / four product levels deep (add-2007-bl: {Z1Z1, Z2Z2, Y1 Z2, Y2 Z1} -> {U1, U2, S1, S2} -> {I, (Z1+Z2)^2, r^2} -> {J, V, Z3} ->
// {S1 J, r (V - X3)};  dbl-2007-bl: {XX, YY, ZZ, (Y+Z)^2} -> {YYYY, ZZ^2, (X+YY)^2} -> M^2 -> M (S - X3)) against eight / five slots of the
// two-lane form: a pair level of the bucket reduction is 9 dependent products instead of 13.  Every lane of the quad computes one product
// of a level with role-selected operands; results travel by quad broadcasts (DPP quad_perm: a VALU move, no LDS round trip).  All four
// lanes pass the same points and all four return the whole result.
template <class G>
struct EC4 {
  typedef typename G::F F;
  typedef Jac<F> J;
  template <int K> PCD_DEV static F bc(const F& a) { F r;   // lane K of every quad to all four of its lanes
#pragma unroll
    for (int i = 0; i < F::N; i++) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], K * 0x55, 0xF, 0xF, false);
    return r; }
  PCD_DEV static F sel4(uint32_t l, const F& a0, const F& a1, const F& a2, const F& a3) { F r;
#pragma unroll
    for (int i = 0; i < F::N; i++) { const uint32_t lo = (l & 1u) ? a1.v[i] : a0.v[i], hi = (l & 1u) ? a3.v[i] : a2.v[i]; r.v[i] = (l & 2u) ? hi : lo; }
    return r; }
  PCD_DEV static J dbl4(const J& p) {
    if (p.is_inf()) return p;
    const uint32_t l = threadIdx.x & 3u;
    const F yz = p.Y + p.Z;
    const F a1 = sel4(l, p.X, p.Y, p.Z, yz);
    const F s1 = a1 * a1;                                   // 0: XX   1: YY   2: ZZ   3: (Y + Z)^2
    const F XX = bc<0>(s1), YY = bc<1>(s1), ZZ = bc<2>(s1), YZ2 = bc<3>(s1);
    const F xyy = p.X + YY;
    const F a2 = sel4(l, YY, ZZ, xyy, xyy);
    const F s2 = a2 * a2;                                   // 0: YYYY   1: ZZ^2   2: (X + YY)^2   (3: the same again)
    const F YYYY = bc<0>(s2), ZZ2 = bc<1>(s2), XYY2 = bc<2>(s2);
    const F S = (XYY2 - XX - YYYY).dbl();
    const F M = XX.dbl() + XX + G::mul_by_a(ZZ2);
    J r;
    r.X = M * M - S.dbl();                                  // (computed by all four lanes)
    r.Z = YZ2 - YY - ZZ;
    r.Y = M * (S - r.X) - YYYY.dbl().dbl().dbl();
    return r;
  }
  PCD_DEV static J add4(const J& p, const J& q) {
    if (p.is_inf()) return q;
    if (q.is_inf()) return p;
    const uint32_t l = threadIdx.x & 3u;
    const F a1 = sel4(l, p.Z, q.Z, p.Y, q.Y), b1 = sel4(l, p.Z, q.Z, q.Z, p.Z);
    const F s1 = a1 * b1;                                   // 0: Z1Z1   1: Z2Z2   2: Y1 Z2   3: Y2 Z1
    const F Z1Z1 = bc<0>(s1), Z2Z2 = bc<1>(s1);
    const F a2 = sel4(l, p.X, q.X, s1, s1), b2 = sel4(l, Z2Z2, Z1Z1, Z2Z2, Z1Z1);
    const F s2 = a2 * b2;                                   // 0: U1   1: U2   2: S1   3: S2
    const F U1 = bc<0>(s2), U2 = bc<1>(s2), S1 = bc<2>(s2), S2 = bc<3>(s2);
    const F H = U2 - U1;
    F r = S2 - S1;
    if (H.is_zero()) return r.is_zero() ? dbl4(p) : J::infinity();
    r = r.dbl();
    const F h2 = H.dbl(), zs = p.Z + q.Z;
    const F a3 = sel4(l, h2, zs, r, r);
    const F s3 = a3 * a3;                                   // 0: I = (2H)^2   1: (Z1 + Z2)^2   2: r^2   (3: the same again)
    const F I = bc<0>(s3), ZS = bc<1>(s3), rr = bc<2>(s3);
    const F zc = ZS - Z1Z1 - Z2Z2;
    const F a4 = sel4(l, H, U1, zc, zc), b4 = sel4(l, I, I, H, H);
    const F s4 = a4 * b4;                                   // 0: J = H I   1: V = U1 I   2: Z3   (3: the same again)
    const F Jv = bc<0>(s4), V = bc<1>(s4);
    J out;
    out.Z = bc<2>(s4);
    out.X = rr - Jv - V.dbl();
    const F vx = V - out.X;
    const F a5 = sel4(l, S1, r, S1, r), b5 = sel4(l, Jv, vx, Jv, vx);
    const F s5 = a5 * b5;                                   // 0, 2: S1 J   1, 3: r (V - X3)
    out.Y = bc<1>(s5) - bc<0>(s5).dbl();
    return out;
  }
};
#ifndef PCD_EC4
#define PCD_EC4 1   // 0: the prime-field groups' pair levels keep two lanes per operation (EC2)
#endif
// groups whose latency-bound pair levels run two lanes per operation
template <class G> struct TwoLaneOps { static constexpr bool value = false; };
template <class FQ, class FRP, unsigned A, int CURVE, bool INL> struct TwoLaneOps<G1Cfg<FQ, FRP, A, CURVE, INL>> { static constexpr bool value = true; };
// ... and the extension-field groups with two HALVES of L lanes per operation (round 6; PCD_EC2_SPLIT=0 restores one lane-split point per item)
#ifndef PCD_EC2_SPLIT
#define PCD_EC2_SPLIT 1
#endif
template <class FQ, class FRP, unsigned A, unsigned NR, int CURVE, bool INL> struct TwoLaneOps<G2Cfg2<FQ, FRP, A, NR, CURVE, INL>> { static constexpr bool value = PCD_EC2_SPLIT != 0; };
template <class FQ, class FRP, unsigned A, unsigned NR, int CURVE, bool INL> struct TwoLaneOps<G2Cfg3<FQ, FRP, A, NR, CURVE, INL>> { static constexpr bool value = PCD_EC2_SPLIT != 0; };

}  // namespace pcd
