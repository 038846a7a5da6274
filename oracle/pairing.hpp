// TEST INFRASTRUCTURE ONLY -- CPU restatement of ark-ec `models::mnt4` / `models::mnt6` ate pairing
// (SURVEY.md Appendix A.6): G2 precomputation in extended-Jacobian (X,Y,Z,T=Z^2) coordinates producing
// doubling / addition line coefficients, "flipped" Miller loop over |q - r|, and the two-chunk final
// exponentiation.  Reached from src/ec_cycle_pcd/mod.rs:239 (HelpSNARK::verify) and :71 (process_vk).
// PARITY UNPINNED (see field.hpp / DESIGN.md); the reduced value is validated against the textbook
// affine ate pairing in oracle/pyoracle.py (independent formulas, flat F_{q^k} representation).
#pragma once
#include <vector>

#include "curve.hpp"

namespace orc {

// C: curve config with  Fq, G2F (Fp2/Fp3 over Fq), Fqk = FpT2<G2F>, twist_a() (G2F), loop limbs, flags
template <class C>
struct Pairing {
  typedef typename C::Fq Fq;
  typedef typename C::G2F E;
  typedef FpT2<E> Fqk;

  struct DblCoeffs { E c_h, c_4c, c_j, c_l; };
  struct AddCoeffs { E c_l1, c_rz; };
  struct G1Prep { Fq x, y; E x_twist, y_twist; bool inf; };
  struct G2Prep { E x, y, x_over_twist, y_over_twist; std::vector<DblCoeffs> dbl; std::vector<AddCoeffs> add; bool inf; };
  struct Ext { E x, y, z, t; };

  static E lift(const Fq& v) { E r = E::zero(); r.c0 = v; return r; }
  static E twist() { return E::one().mul_by_u(); }

  static G1Prep prepare_g1(const Affine<Fq>& p) {
    G1Prep r;
    r.inf = p.inf; r.x = p.x; r.y = p.y;
    r.x_twist = twist().mul_base(p.x);
    r.y_twist = twist().mul_base(p.y);
    return r;
  }
  static void doubling_step(const Ext& r, Ext& o, DblCoeffs& co) {
    E a = r.t.sqr(), b = r.x.sqr(), c = r.y.sqr(), d = c.sqr();
    E e = (r.x + c).sqr() - b - d;
    E f = b.dbl() + b + C::twist_a() * a;
    E g = f.sqr();
    E d8 = d.dbl().dbl().dbl();
    o.x = g - e.dbl().dbl();
    o.y = f * (e.dbl() - o.x) - d8;
    o.z = (r.y + r.z).sqr() - c - r.z.sqr();
    o.t = o.z.sqr();
    co.c_h = (o.z + r.t).sqr() - o.t - a;
    co.c_4c = c.dbl().dbl();
    co.c_j = (f + r.t).sqr() - g - a;
    co.c_l = (f + r.x).sqr() - g - b;
  }
  static void mixed_addition_step(const E& x, const E& y, const Ext& r, Ext& o, AddCoeffs& co) {
    E a = y.sqr();
    E b = r.t * x;
    E d = ((r.z + y).sqr() - a - r.t) * r.t;
    E h = b - r.x;
    E i = h.sqr();
    E e = i.dbl().dbl();
    E j = h * e;
    E v = r.x * e;
    E l1 = d - r.y.dbl();
    o.x = l1.sqr() - j - v.dbl();
    o.y = l1 * (v - o.x) - j * r.y.dbl();
    o.z = (r.z + h).sqr() - r.t - i;
    o.t = o.z.sqr();
    co.c_l1 = l1;
    co.c_rz = o.z;
  }
  static int loop_bit(int i) { return (int)((C::ate_loop()[i / 64] >> (i % 64)) & 1); }

  static G2Prep prepare_g2(const Affine<E>& q) {
    G2Prep r;
    r.inf = q.inf;
    if (q.inf) return r;
    E tinv = twist().inv();
    r.x = q.x; r.y = q.y;
    r.x_over_twist = q.x * tinv;
    r.y_over_twist = q.y * tinv;
    Ext cur = {q.x, q.y, E::one(), E::one()}, nxt;
    for (int i = C::ATE_LOOP_BITS - 2; i >= 0; i--) {
      DblCoeffs dc; doubling_step(cur, nxt, dc); r.dbl.push_back(dc); cur = nxt;
      if (loop_bit(i)) { AddCoeffs ac; mixed_addition_step(q.x, q.y, cur, nxt, ac); r.add.push_back(ac); cur = nxt; }
    }
    if (C::ATE_NEG) {
      E zi = cur.z.inv(), zi2 = zi.sqr(), zi3 = zi2 * zi;
      E mx = cur.x * zi2, my = (cur.y * zi3).neg();
      AddCoeffs ac; mixed_addition_step(mx, my, cur, nxt, ac); r.add.push_back(ac);
    }
    return r;
  }
  static Fqk miller_loop(const G1Prep& p, const G2Prep& q) {
    if (p.inf || q.inf) return Fqk::one();
    E l1_coeff = lift(p.x) - q.x_over_twist;
    Fqk f = Fqk::one();
    size_t di = 0, ai = 0;
    for (int i = C::ATE_LOOP_BITS - 2; i >= 0; i--) {
      const DblCoeffs& dc = q.dbl[di++];
      Fqk g_rr = {dc.c_l - dc.c_4c - dc.c_j * p.x_twist, dc.c_h * p.y_twist};
      f = f.sqr() * g_rr;
      if (loop_bit(i)) {
        const AddCoeffs& ac = q.add[ai++];
        Fqk g_rq = {ac.c_rz * p.y_twist, (q.y_over_twist * ac.c_rz + l1_coeff * ac.c_l1).neg()};
        f = f * g_rq;
      }
    }
    if (C::ATE_NEG) {
      const AddCoeffs& ac = q.add[ai++];
      Fqk g = {ac.c_rz * p.y_twist, (q.y_over_twist * ac.c_rz + l1_coeff * ac.c_l1).neg()};
      f = (f * g).inv();
    }
    return f;
  }
  static Fqk first_chunk(const Fqk& elt, const Fqk& elt_inv) {
    if (E::DEG == 2) {  // k = 4: elt^(q^2 - 1)
      return elt.frobenius(2) * elt_inv;
    }
    // k = 6: elt^((q^3 - 1)(q + 1))
    Fqk a = elt.frobenius(3) * elt_inv;
    return a.frobenius(1) * a;
  }
  static Fqk last_chunk(const Fqk& elt, const Fqk& elt_inv) {
    Fqk w1_part = elt.frobenius(1);  // W1 = 1
    Fqk w0_part = (C::W0_NEG ? elt_inv : elt).pow(C::w0(), C::W0_NLIMBS);
    return w1_part * w0_part;
  }
  static Fqk final_exponentiation(const Fqk& v) {
    Fqk vi = v.inv();
    Fqk first = first_chunk(v, vi), first_inv = first_chunk(vi, v);
    return last_chunk(first, first_inv);
  }
  static Fqk pairing(const Affine<Fq>& p, const Affine<E>& q) {
    return final_exponentiation(miller_loop(prepare_g1(p), prepare_g2(q)));
  }
};

}  // namespace orc
