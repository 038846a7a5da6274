// Wire format of proofs and verifying keys: the ark-serialize `CanonicalSerialize` /
// `CanonicalDeserialize` images of `GroupAffine`, ark-groth16 `Proof` and `VerifyingKey`, so that proofs can move between GPU
// hosts without the Rust host (the reference serialises through these traits wherever a proof or key leaves the process; its
// in-circuit byte order, /root/reference src/ec_cycle_pcd/mod.rs:101-139, is a different, gadget-side encoding and stays in Rust).
//
// Host-side code: byte shuffling plus a handful of field operations per point (Montgomery -> canonical, the sign of y, a square
// root when a compressed point is read), done with the library's own `__host__ __device__` field templates -- no GPU is needed
// and none is used.  Format [UPSTREAM ark-serialize / ark-ec 0.3, restated -- see DESIGN.md "parity unpinned"]:
//   Fp            canonical integer (`into_repr()`), little-endian, ceil(bits / 8) bytes (38 / 95); flag bits, when an element
//                 carries them, are OR-ed into the last byte
//   Fp2 / Fp3     c0, c1 (, c2) in order; flags travel with the LAST coefficient
//   SWFlags       bit 7 = y is the larger of (y, -y) (`y > -y`: integers for Fp; c1 then c0 for Fp2; c2, c1, c0 for Fp3),
//                 bit 6 = point at infinity
//   compressed    x with flags (infinity: x = 0 | 0x40)           uncompressed    x, then y with flags (infinity: (0, 1) | 0x40)
//   Proof         A (G1) || B (G2) || C (G1)
//   VerifyingKey  alpha_g1, beta_g2, gamma_g2, delta_g2, then gamma_abc_g1 as u64 LE length + points
// Reading accepts what writing produces; a point read back is checked to lie on the curve (uncompressed) or rebuilt from x
// (compressed: y = sqrt(x^3 + a x + b), the root selected by the sign flag) and, as upstream's checked `deserialize` does
// (`is_in_correct_subgroup_assuming_on_curve`), to lie in the prime-order subgroup: G1 has cofactor 1 on all four curves, so
// there is nothing to test; a G2 point Q is accepted only if [r]Q = O (the twists have cofactors of ~300 / ~1500 bits, and
// neither the Miller loop nor e(rho A, B) = e(A, B)^rho of the batched verification means anything for a point outside the
// r-torsion).  One scalar multiplication on the host per G2 point (10 ms at 298 bits, ~0.1 s at 753 bits with these templates): a
// proof has one, a key three.  pcdhip_deserialize_points_unchecked skips it (bulk reads of keys a trusted party wrote).
#include <string.h>

#include <vector>

#include "../../include/pcdhip.h"
#include "ec.hip.h"

using namespace pcd;

namespace {

// ---- tiny unsigned big integers (u32 words, little endian) for the square-root exponents
typedef std::vector<uint32_t> Big;
void big_trim(Big& a) { while (a.size() > 1 && a.back() == 0) a.pop_back(); }
Big big_mul(const Big& a, const Big& b) {
  Big r(a.size() + b.size(), 0);
  for (size_t i = 0; i < a.size(); i++) {
    uint64_t c = 0;
    for (size_t j = 0; j < b.size(); j++) { c += (uint64_t)a[i] * b[j] + r[i + j]; r[i + j] = (uint32_t)c; c >>= 32; }
    r[i + b.size()] = (uint32_t)c;
  }
  big_trim(r);
  return r;
}
Big big_add(const Big& a, const Big& b) {
  Big r(std::max(a.size(), b.size()) + 1, 0);
  uint64_t c = 0;
  for (size_t i = 0; i < r.size(); i++) { c += (i < a.size() ? a[i] : 0u) + (uint64_t)(i < b.size() ? b[i] : 0u); r[i] = (uint32_t)c; c >>= 32; }
  big_trim(r);
  return r;
}
void big_sub_small(Big& a, uint32_t k) {  // a >= k
  uint64_t borrow = k;
  for (size_t i = 0; i < a.size() && borrow; i++) { uint64_t v = (uint64_t)a[i] - borrow; a[i] = (uint32_t)v; borrow = (v >> 63) & 1; }
  big_trim(a);
}
void big_shr(Big& a, int bits) {
  for (; bits > 0; bits--) {
    uint32_t c = 0;
    for (size_t i = a.size(); i-- > 0;) { uint32_t n = a[i] & 1u; a[i] = (a[i] >> 1) | (c << 31); c = n; }
  }
  big_trim(a);
}
template <class E>
E f_pow(const E& a, const Big& e) {
  E r = E::one();
  bool started = false;
  for (size_t i = e.size() * 32; i-- > 0;) {
    if (started) r = r.sqr();
    if ((e[i >> 5] >> (i & 31)) & 1) { r = started ? r * a : a; started = true; }
  }
  return r;
}

// ---- coefficient access: an extension element as DEG base-field elements, in serialisation order
template <class F> int f_deg(const F&) { return 1; }
template <class F, unsigned NR> int f_deg(const Fp2<F, NR>&) { return 2; }
template <class F, unsigned NR> int f_deg(const Fp3<F, NR>&) { return 3; }
template <class F> const F& f_coeff(const F& e, int) { return e; }
template <class F, unsigned NR> const F& f_coeff(const Fp2<F, NR>& e, int i) { return i ? e.c1 : e.c0; }
template <class F, unsigned NR> const F& f_coeff(const Fp3<F, NR>& e, int i) { return i == 0 ? e.c0 : i == 1 ? e.c1 : e.c2; }
template <class F> F& f_coeff(F& e, int) { return e; }
template <class F, unsigned NR> F& f_coeff(Fp2<F, NR>& e, int i) { return i ? e.c1 : e.c0; }
template <class F, unsigned NR> F& f_coeff(Fp3<F, NR>& e, int i) { return i == 0 ? e.c0 : i == 1 ? e.c1 : e.c2; }

template <class B>  // B = base field Fp<P, false>
struct Ser {
  typedef typename B::Params P;
  static constexpr int NW = P::N32, BYTES = (P::BITS + 7) / 8;
  static void canon(const B& a, uint32_t* w) { a.to_canonical_words(w); }
  static int cmp(const B& a, const B& b) {  // as integers
    uint32_t x[NW], y[NW];
    canon(a, x); canon(b, y);
    for (int i = NW; i-- > 0;) if (x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
    return 0;
  }
  static void write(const B& a, uint8_t* out, uint8_t flags) {
    uint32_t w[NW];
    canon(a, w);
    for (int i = 0; i < BYTES; i++) out[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
    out[BYTES - 1] |= flags;
  }
  // false when the integer is not a reduced residue
  static bool read(const uint8_t* in, bool with_flags, B* a, uint8_t* flags) {
    uint32_t w[NW] = {0};
    for (int i = 0; i < BYTES; i++) {
      uint8_t b = in[i];
      if (i == BYTES - 1 && with_flags) { *flags = b & 0xC0; b &= 0x3F; }
      w[i >> 2] |= (uint32_t)b << (8 * (i & 3));
    }
    uint32_t m[NW];  // the modulus: (p - 2) + 2
    uint64_t c = 2;
    for (int i = 0; i < NW; i++) { c += P::modm2(i); m[i] = (uint32_t)c; c >>= 32; }
    bool less = false;
    for (int i = NW; i-- > 0;)
      if (w[i] != m[i]) { less = w[i] < m[i]; break; }
    if (!less) return false;  // not a reduced residue
    *a = B::from_canonical_words(w);
    return true;
  }
};

// y > -y in upstream's field order (most significant coefficient first)
template <class E>
bool y_is_larger(const E& y) {
  typedef typename E::Base B;
  const E ny = y.neg();
  for (int i = f_deg(y); i-- > 0;) {
    const int c = Ser<B>::cmp(f_coeff(y, i), f_coeff(ny, i));
    if (c) return c > 0;
  }
  return false;
}

// ---- square roots (compressed points)
template <class E>
bool ts_sqrt(const E& a, int S, const E& z0, const Big& e_half, E* out) {  // Tonelli-Shanks; z0 of order 2^S, e_half = (t - 1) / 2
  if (a.is_zero()) { *out = a; return true; }
  const E w = f_pow(a, e_half);
  E x = a * w, b = x * w, z = z0;
  int v = S;
  {
    E t = b;
    for (int i = 0; i + 1 < S; i++) t = t.sqr();
    if (!(t == E::one())) return false;  // not a square
  }
  while (!(b == E::one())) {
    int k = 0;
    E t = b;
    while (!(t == E::one())) { t = t.sqr(); k++; }
    E wz = z;
    for (int j = 0; j < v - k - 1; j++) wz = wz.sqr();
    z = wz.sqr();
    b = b * z;
    x = x * wz;
    v = k;
  }
  *out = x;
  return true;
}
template <class B>
Big field_modulus() {
  typedef typename B::Params P;
  Big q(P::N32);
  uint64_t c = 2;  // (p - 2) + 2
  for (int i = 0; i < P::N32; i++) { c += P::modm2(i); q[i] = (uint32_t)c; c >>= 32; }
  return q;
}
template <class B>
bool f_sqrt(const B& a, B* out) {
  typedef typename B::Params P;
  Big t = field_modulus<B>();
  big_sub_small(t, 1);
  big_shr(t, P::TWO_ADICITY);
  big_sub_small(t, 1);
  big_shr(t, 1);
  return ts_sqrt(a, P::TWO_ADICITY, B::two_adic_root(), t, out);
}
// Fq2: through the norm (q = 1 mod 4, so the "complex" method does not apply): with n = sqrt(a0^2 - nr a1^2) and a square
// d = (a0 + n) / 2 or (a0 - n) / 2:  sqrt(a) = sqrt(d) + a1 / (2 sqrt(d)) u
template <class B, unsigned NR>
bool f_sqrt(const Fp2<B, NR>& a, Fp2<B, NR>* out) {
  if (a.c1.is_zero()) {
    B r;
    if (f_sqrt(a.c0, &r)) { *out = {r, B::zero()}; return true; }
    // a0 is a non-residue: sqrt = sqrt(a0 / nr) u
    const B nr_inv = B::from_u64(NR).inv();
    if (!f_sqrt(a.c0 * nr_inv, &r)) return false;
    *out = {B::zero(), r};
    return true;
  }
  B n;
  if (!f_sqrt(a.c0.sqr() - a.c1.sqr().mul_small(NR), &n)) return false;
  const B half = B::from_u64(2).inv();
  B d = (a.c0 + n) * half, x0;
  if (!f_sqrt(d, &x0)) { d = (a.c0 - n) * half; if (!f_sqrt(d, &x0)) return false; }
  const B x1 = a.c1 * x0.dbl().inv();
  *out = {x0, x1};
  return true;
}
// Fq3: Tonelli-Shanks in the extension; |Fq3*| = (q - 1)(q^2 + q + 1) has the 2-adicity of q - 1, so with t = (q - 1) / 2^S the odd
// part is t (q^2 + q + 1), and an element of order 2^S is root^(q^2 + q + 1) = root^3 (root lies in Fq)
template <class B, unsigned NR>
bool f_sqrt(const Fp3<B, NR>& a, Fp3<B, NR>* out) {
  typedef typename B::Params P;
  const Big q = field_modulus<B>();
  Big t = q;
  big_sub_small(t, 1);
  big_shr(t, P::TWO_ADICITY);
  Big one(1, 1u);
  Big t3 = big_mul(t, big_add(big_add(big_mul(q, q), q), one));
  big_sub_small(t3, 1);
  big_shr(t3, 1);
  const B r = B::two_adic_root();
  const Fp3<B, NR> z = {r * r * r, B::zero(), B::zero()};
  return ts_sqrt(a, P::TWO_ADICITY, z, t3, out);
}

// ---- curves
template <class B> B elem_from_abi32(const uint32_t* w) { return B::from_abi(w); }
template <class E> E ext_from_abi32(const uint32_t* w) { return E::from_abi(w); }

template <class E>  // E: coordinate field
struct CurveIO {
  typedef typename E::Base B;
  static constexpr int DEG = E::DEG;
  static constexpr int EB = DEG * Ser<B>::BYTES;  // bytes of one coordinate
  static size_t size(int compressed) { return compressed ? EB : 2 * EB; }
  static void write_elem(const E& e, uint8_t* out, uint8_t flags) {
    for (int i = 0; i < DEG; i++) Ser<B>::write(f_coeff(e, i), out + i * Ser<B>::BYTES, i == DEG - 1 ? flags : 0);
  }
  static bool read_elem(const uint8_t* in, bool with_flags, E* e, uint8_t* flags) {
    for (int i = 0; i < DEG; i++)
      if (!Ser<B>::read(in + i * Ser<B>::BYTES, with_flags && i == DEG - 1, &f_coeff(*e, i), flags)) return false;
    return true;
  }
  static void serialize(const uint32_t* xy_abi, bool inf, int compressed, uint8_t* out) {
    if (inf) {
      write_elem(E::zero(), out, compressed ? 0x40 : 0);
      if (!compressed) write_elem(E::one(), out + EB, 0x40);
      return;
    }
    const E x = E::from_abi(xy_abi), y = E::from_abi(xy_abi + E::ABI_WORDS);
    if (compressed) { write_elem(x, out, y_is_larger(y) ? 0x80 : 0); return; }
    write_elem(x, out, 0);
    write_elem(y, out + EB, 0);
  }
  // a, b: curve coefficients (device-image elements)
  static int deserialize(const uint8_t* in, int compressed, const E& a, const E& b, uint32_t* xy_abi, uint8_t* inf) {
    E x, y;
    uint8_t flags = 0;
    *inf = 0;
    if (!read_elem(in, compressed != 0, &x, &flags)) return PCDHIP_E_ARG;
    if (!compressed && !read_elem(in + EB, true, &y, &flags)) return PCDHIP_E_ARG;
    if ((flags & 0xC0) == 0xC0) return PCDHIP_E_ARG;
    if (flags & 0x40) {  // infinity: the library's convention is zero coordinates + flag
      memset(xy_abi, 0, (size_t)2 * E::ABI_WORDS * 4);
      *inf = 1;
      return PCDHIP_OK;
    }
    const E rhs = (x.sqr() + a) * x + b;
    if (compressed) {
      if (!f_sqrt(rhs, &y)) return PCDHIP_E_ARG;  // x is not the abscissa of a point
      if (y_is_larger(y) != ((flags & 0x80) != 0)) y = y.neg();
    } else if (!(y.sqr() == rhs)) return PCDHIP_E_ARG;  // not on the curve
    x.to_abi(xy_abi);
    y.to_abi(xy_abi + E::ABI_WORDS);
    return PCDHIP_OK;
  }
};

template <class FQ, unsigned NR, int DEG> struct G2Field;
template <class FQ, unsigned NR> struct G2Field<FQ, NR, 2> { typedef Fp2<Fp<FQ, false>, NR> type; };
template <class FQ, unsigned NR> struct G2Field<FQ, NR, 3> { typedef Fp3<Fp<FQ, false>, NR> type; };

// [r]Q == O for a point of the twist (affine, C-ABI image), r = the modulus of the scalar field FRP
template <class G2C>
bool g2_in_subgroup(const uint32_t* xy_abi) {
  typedef typename G2C::F E;
  typedef typename G2C::FR FRP;
  uint32_t r[FRP::N32];
  uint64_t carry = 2;  // r = (r - 2) + 2
  for (int i = 0; i < FRP::N32; i++) { const uint64_t x = (uint64_t)FRP::modm2(i) + carry; r[i] = (uint32_t)x; carry = x >> 32; }
  const Jac<E> q = {E::from_abi(xy_abi), E::from_abi(xy_abi + E::ABI_WORDS), E::one()};
  return EC<G2C>::mul(q, r, FRP::N32).is_inf();
}

#define PCD_WIRE_CURVE(NAME, FQ, NRV, DEGV)                                                                          \
  struct Wire_##NAME {                                                                                               \
    typedef Fp<FQ, false> B;                                                                                         \
    typedef G2Field<FQ, NRV, DEGV>::type E2;                                                                         \
    typedef G2_##NAME##_CHK G2C;                                                                                     \
    static B a1() { static const uint32_t w[] = PCD_##NAME##_A_MONT; return B::from_abi(w); }                        \
    static B b1() { static const uint32_t w[] = PCD_##NAME##_B_MONT; return B::from_abi(w); }                        \
    static E2 a2() { static const uint32_t w[] = PCD_##NAME##_TWIST_A_MONT; return E2::from_abi(w); }                \
    static E2 b2() { static const uint32_t w[] = PCD_##NAME##_TWIST_B_MONT; return E2::from_abi(w); }                \
  };
typedef G2_MNT4_298_C G2_MNT4_298_CHK;   // (the compact, non-inlined field variants: the same memory image as Wire_*::E2)
typedef G2_MNT6_298_C G2_MNT6_298_CHK;
typedef G2_MNT4_753 G2_MNT4_753_CHK;
typedef G2_MNT6_753 G2_MNT6_753_CHK;
PCD_WIRE_CURVE(MNT4_298, F298A, PCD_MNT4_298_NR_SMALL, 2)
PCD_WIRE_CURVE(MNT6_298, F298B, PCD_MNT6_298_NR_SMALL, 3)
PCD_WIRE_CURVE(MNT4_753, F753A, PCD_MNT4_753_NR_SMALL, 2)
PCD_WIRE_CURVE(MNT6_753, F753B, PCD_MNT6_753_NR_SMALL, 3)

template <class W>
size_t w_size(int group, int compressed) {
  return group == 1 ? CurveIO<typename W::B>::size(compressed) : CurveIO<typename W::E2>::size(compressed);
}
template <class W>
int w_serialize(int group, const uint64_t* xy, const uint8_t* inf, size_t n, int compressed, uint8_t* out) {
  if (group == 1) {
    typedef CurveIO<typename W::B> IO;
    for (size_t i = 0; i < n; i++)
      IO::serialize((const uint32_t*)xy + i * 2 * W::B::ABI_WORDS, inf && inf[i], compressed, out + i * IO::size(compressed));
  } else {
    typedef CurveIO<typename W::E2> IO;
    for (size_t i = 0; i < n; i++)
      IO::serialize((const uint32_t*)xy + i * 2 * W::E2::ABI_WORDS, inf && inf[i], compressed, out + i * IO::size(compressed));
  }
  return PCDHIP_OK;
}
template <class W>
int w_deserialize(int group, const uint8_t* in, size_t n, int compressed, uint64_t* xy, uint8_t* inf, bool check_subgroup = true) {
  if (group == 1) {
    typedef CurveIO<typename W::B> IO;
    const typename W::B a = W::a1(), b = W::b1();
    for (size_t i = 0; i < n; i++) {
      int rc = IO::deserialize(in + i * IO::size(compressed), compressed, a, b, (uint32_t*)xy + i * 2 * W::B::ABI_WORDS, inf + i);
      if (rc) return rc;
    }
  } else {
    typedef CurveIO<typename W::E2> IO;
    const typename W::E2 a = W::a2(), b = W::b2();
    for (size_t i = 0; i < n; i++) {
      int rc = IO::deserialize(in + i * IO::size(compressed), compressed, a, b, (uint32_t*)xy + i * 2 * W::E2::ABI_WORDS, inf + i);
      if (rc) return rc;
      static_assert(std::is_same<typename W::G2C::F, typename W::E2>::value, "subgroup check runs on the field type the point was read in");
      if (check_subgroup && !inf[i] && !g2_in_subgroup<typename W::G2C>((const uint32_t*)xy + i * 2 * W::E2::ABI_WORDS)) return PCDHIP_E_ARG;
    }
  }
  return PCDHIP_OK;
}

#define PCD_WIRE_DISPATCH(curve, CALL)                   \
  switch (curve) {                                       \
    case 0: return CALL(Wire_MNT4_298);                  \
    case 1: return CALL(Wire_MNT6_298);                  \
    case 2: return CALL(Wire_MNT4_753);                  \
    case 3: return CALL(Wire_MNT6_753);                  \
    default: break;                                      \
  }

}  // namespace

extern "C" {

size_t pcdhip_serialized_size(int curve_id, int group_id, int compressed) {
  if (group_id != 1 && group_id != 2) return 0;
#define CALL(W) w_size<W>(group_id, compressed)
  PCD_WIRE_DISPATCH(curve_id, CALL)
#undef CALL
  return 0;
}
int pcdhip_serialize_points(int curve_id, int group_id, const uint64_t* xy_mont, const uint8_t* inf, size_t n, int compressed, uint8_t* out) {
  if ((group_id != 1 && group_id != 2) || (n && (!xy_mont || !out))) return PCDHIP_E_ARG;
  try {
#define CALL(W) w_serialize<W>(group_id, xy_mont, inf, n, compressed, out)
    PCD_WIRE_DISPATCH(curve_id, CALL)
#undef CALL
  } catch (...) { return PCDHIP_E_OOM; }
  return PCDHIP_E_ARG;
}
int pcdhip_deserialize_points_unchecked(int curve_id, int group_id, const uint8_t* in, size_t n, int compressed, uint64_t* xy_mont, uint8_t* inf) {
  if ((group_id != 1 && group_id != 2) || (n && (!in || !xy_mont || !inf))) return PCDHIP_E_ARG;
  try {
#define CALL(W) w_deserialize<W>(group_id, in, n, compressed, xy_mont, inf, false)
    PCD_WIRE_DISPATCH(curve_id, CALL)
#undef CALL
  } catch (...) { return PCDHIP_E_OOM; }
  return PCDHIP_E_ARG;
}
int pcdhip_deserialize_points(int curve_id, int group_id, const uint8_t* in, size_t n, int compressed, uint64_t* xy_mont, uint8_t* inf) {
  if ((group_id != 1 && group_id != 2) || (n && (!in || !xy_mont || !inf))) return PCDHIP_E_ARG;
  try {
#define CALL(W) w_deserialize<W>(group_id, in, n, compressed, xy_mont, inf)
    PCD_WIRE_DISPATCH(curve_id, CALL)
#undef CALL
  } catch (...) { return PCDHIP_E_OOM; }
  return PCDHIP_E_ARG;
}

size_t pcdhip_proof_serialized_size(int curve_id, int compressed) {
  return 2 * pcdhip_serialized_size(curve_id, 1, compressed) + pcdhip_serialized_size(curve_id, 2, compressed);
}
int pcdhip_proof_serialize(int curve_id, const uint64_t* proof, const uint8_t* proof_inf, int compressed, uint8_t* out) {
  const size_t s1 = pcdhip_serialized_size(curve_id, 1, compressed), s2 = pcdhip_serialized_size(curve_id, 2, compressed);
  if (!s1 || !proof || !out) return PCDHIP_E_ARG;
  const size_t l1 = (size_t)pcdhip_point_limbs(curve_id, 1), l2 = (size_t)pcdhip_point_limbs(curve_id, 2);
  int rc = pcdhip_serialize_points(curve_id, 1, proof, proof_inf, 1, compressed, out);
  rc = rc ? rc : pcdhip_serialize_points(curve_id, 2, proof + l1, proof_inf ? proof_inf + 1 : nullptr, 1, compressed, out + s1);
  rc = rc ? rc : pcdhip_serialize_points(curve_id, 1, proof + l1 + l2, proof_inf ? proof_inf + 2 : nullptr, 1, compressed, out + s1 + s2);
  return rc;
}
int pcdhip_proof_deserialize(int curve_id, const uint8_t* in, int compressed, uint64_t* proof, uint8_t* proof_inf) {
  const size_t s1 = pcdhip_serialized_size(curve_id, 1, compressed), s2 = pcdhip_serialized_size(curve_id, 2, compressed);
  if (!s1 || !in || !proof || !proof_inf) return PCDHIP_E_ARG;
  const size_t l1 = (size_t)pcdhip_point_limbs(curve_id, 1), l2 = (size_t)pcdhip_point_limbs(curve_id, 2);
  int rc = pcdhip_deserialize_points(curve_id, 1, in, 1, compressed, proof, proof_inf);
  rc = rc ? rc : pcdhip_deserialize_points(curve_id, 2, in + s1, 1, compressed, proof + l1, proof_inf + 1);
  rc = rc ? rc : pcdhip_deserialize_points(curve_id, 1, in + s1 + s2, 1, compressed, proof + l1 + l2, proof_inf + 2);
  return rc;
}

size_t pcdhip_vk_serialized_size(int curve_id, size_t num_inputs, int compressed) {
  const size_t s1 = pcdhip_serialized_size(curve_id, 1, compressed), s2 = pcdhip_serialized_size(curve_id, 2, compressed);
  return s1 ? s1 + 3 * s2 + 8 + num_inputs * s1 : 0;
}
int pcdhip_vk_serialize(int curve_id, const uint64_t* alpha_g1, const uint64_t* beta_g2, const uint64_t* gamma_g2, const uint64_t* delta_g2,
                        const uint64_t* gamma_abc_g1, const uint8_t* gamma_abc_inf, size_t num_inputs, int compressed, uint8_t* out) {
  const size_t s1 = pcdhip_serialized_size(curve_id, 1, compressed), s2 = pcdhip_serialized_size(curve_id, 2, compressed);
  if (!s1 || !alpha_g1 || !beta_g2 || !gamma_g2 || !delta_g2 || (num_inputs && !gamma_abc_g1) || !out) return PCDHIP_E_ARG;
  int rc = pcdhip_serialize_points(curve_id, 1, alpha_g1, nullptr, 1, compressed, out);
  rc = rc ? rc : pcdhip_serialize_points(curve_id, 2, beta_g2, nullptr, 1, compressed, out + s1);
  rc = rc ? rc : pcdhip_serialize_points(curve_id, 2, gamma_g2, nullptr, 1, compressed, out + s1 + s2);
  rc = rc ? rc : pcdhip_serialize_points(curve_id, 2, delta_g2, nullptr, 1, compressed, out + s1 + 2 * s2);
  uint8_t* p = out + s1 + 3 * s2;
  for (int i = 0; i < 8; i++) p[i] = (uint8_t)((uint64_t)num_inputs >> (8 * i));
  rc = rc ? rc : pcdhip_serialize_points(curve_id, 1, gamma_abc_g1, gamma_abc_inf, num_inputs, compressed, p + 8);
  return rc;
}
int pcdhip_vk_deserialize(int curve_id, const uint8_t* in, size_t in_len, int compressed, uint64_t* alpha_g1, uint64_t* beta_g2, uint64_t* gamma_g2,
                          uint64_t* delta_g2, uint64_t* gamma_abc_g1, uint8_t* gamma_abc_inf, size_t max_inputs, size_t* num_inputs) {
  const size_t s1 = pcdhip_serialized_size(curve_id, 1, compressed), s2 = pcdhip_serialized_size(curve_id, 2, compressed);
  if (!s1 || !in || !alpha_g1 || !beta_g2 || !gamma_g2 || !delta_g2 || !num_inputs || in_len < s1 + 3 * s2 + 8) return PCDHIP_E_ARG;
  uint8_t f[4] = {0, 0, 0, 0};
  int rc = pcdhip_deserialize_points(curve_id, 1, in, 1, compressed, alpha_g1, f);
  rc = rc ? rc : pcdhip_deserialize_points(curve_id, 2, in + s1, 1, compressed, beta_g2, f + 1);
  rc = rc ? rc : pcdhip_deserialize_points(curve_id, 2, in + s1 + s2, 1, compressed, gamma_g2, f + 2);
  rc = rc ? rc : pcdhip_deserialize_points(curve_id, 2, in + s1 + 2 * s2, 1, compressed, delta_g2, f + 3);
  if (rc) return rc;
  if (f[0] | f[1] | f[2] | f[3]) return PCDHIP_E_ARG;  // a key with a point at infinity in these slots is not a key
  const uint8_t* p = in + s1 + 3 * s2;
  uint64_t cnt = 0;
  for (int i = 0; i < 8; i++) cnt |= (uint64_t)p[i] << (8 * i);
  *num_inputs = (size_t)cnt;
  if (cnt > max_inputs || in_len < s1 + 3 * s2 + 8 + cnt * s1) return PCDHIP_E_ARG;
  if (cnt && (!gamma_abc_g1 || !gamma_abc_inf)) return PCDHIP_E_ARG;
  return pcdhip_deserialize_points(curve_id, 1, p + 8, (size_t)cnt, compressed, gamma_abc_g1, gamma_abc_inf);
}

}  // extern "C"
