// TEST INFRASTRUCTURE ONLY -- CPU restatement of the Groth16 prover arithmetic that
// `IC::MainSNARK::prove` / `IC::HelpSNARK::prove` (src/ec_cycle_pcd/mod.rs:171,179) run for the
// tests' configuration `Groth16<MNT4_298>` / `Groth16<MNT6_298>` (tests/mnt4_groth16.rs:22-30):
// R1CSToQAP::witness_map (libsnark reduction) + create_proof, plus the
// generator (`circuit_specific_setup`, mod.rs:69,78) with caller-supplied toxic waste and the
// verifier (`verify`, mod.rs:239).  PARITY UNPINNED (see field.hpp / DESIGN.md).
#pragma once
#include <functional>
#include <vector>

#include "curve.hpp"
#include "fft.hpp"
#include "pairing.hpp"

namespace orc {

// ------------------------------------------------------------------------------------------------ curve configs
#define ORC_ARR(...) __VA_ARGS__
template <class F, int CNT>
inline void load_ext(const u64* raw, F* out) {  // CNT consecutive base-field elements
  memcpy((void*)out, raw, sizeof(u64) * F::Base::N * CNT);
}

#define ORC_DEF_CURVE(NAME, FQ, FR, G2TYPE, PFX)                                                   \
  struct NAME {                                                                                    \
    static constexpr int ID = PFX##_ID;                                                            \
    typedef Fp<FQ> Fq;                                                                             \
    typedef Fp<FR> Fr;                                                                             \
    typedef G2TYPE G2F;                                                                            \
    static constexpr int ATE_LOOP_BITS = PFX##_ATE_LOOP_BITS;                                      \
    static constexpr bool ATE_NEG = PFX##_ATE_NEG;                                                 \
    static constexpr bool W0_NEG = PFX##_W0_NEG;                                                   \
    static constexpr int W0_NLIMBS = PFX##_W0_NLIMBS;                                              \
    static const u64* ate_loop() { static const u64 v[] = PFX##_ATE_LOOP; return v; }              \
    static const u64* w0() { static const u64 v[] = PFX##_W0; return v; }                          \
    static Fq a() { static const u64 v[] = PFX##_A_MONT; return Fq::from_raw(v); }                 \
    static Fq b() { static const u64 v[] = PFX##_B_MONT; return Fq::from_raw(v); }                 \
    static G2F twist_a() { static const u64 v[] = PFX##_TWIST_A_MONT; G2F r; load_ext<G2F, G2F::DEG>(v, &r); return r; } \
    static G2F twist_b() { static const u64 v[] = PFX##_TWIST_B_MONT; G2F r; load_ext<G2F, G2F::DEG>(v, &r); return r; } \
    static Affine<Fq> g1() { static const u64 v[] = PFX##_G1_MONT; return {Fq::from_raw(v), Fq::from_raw(v + Fq::N), false}; } \
    static Affine<G2F> g2() { static const u64 v[] = PFX##_G2_MONT; Affine<G2F> r; load_ext<G2F, G2F::DEG>(v, &r.x); \
                              load_ext<G2F, G2F::DEG>(v + Fq::N * G2F::DEG, &r.y); r.inf = false; return r; } \
    static Group<Fq> G1() { return {a()}; }                                                        \
    static Group<G2F> G2() { return {twist_a()}; }                                                 \
  };
typedef Fp2<Fp<F298A>, PCD_MNT4_298_NR_SMALL> Fq2_298;
typedef Fp3<Fp<F298B>, PCD_MNT6_298_NR_SMALL> Fq3_298;
typedef Fp2<Fp<F753A>, PCD_MNT4_753_NR_SMALL> Fq2_753;
typedef Fp3<Fp<F753B>, PCD_MNT6_753_NR_SMALL> Fq3_753;
ORC_DEF_CURVE(MNT4_298, F298A, F298B, Fq2_298, PCD_MNT4_298)
ORC_DEF_CURVE(MNT6_298, F298B, F298A, Fq3_298, PCD_MNT6_298)
ORC_DEF_CURVE(MNT4_753, F753A, F753B, Fq2_753, PCD_MNT4_753)
ORC_DEF_CURVE(MNT6_753, F753B, F753A, Fq3_753, PCD_MNT6_753)

// ------------------------------------------------------------------------------------------------ R1CS (CSR)
template <class Fr>
struct Csr {
  size_t rows;
  const u64* row_ptr;     // rows + 1
  const uint32_t* col;    // nnz
  const Fr* coeff;        // nnz, Montgomery
  Fr dot(size_t r, const Fr* z) const {
    Fr acc = Fr::zero();
    for (u64 k = row_ptr[r]; k < row_ptr[r + 1]; k++) acc = acc + coeff[k] * z[col[k]];
    return acc;
  }
};

inline int domain_log_for(size_t need) { int l = 0; while (((size_t)1 << l) < need) l++; return l; }

// R1CSToQAP::witness_map -- libsnark reduction (Appendix A.2). Returns n coefficients of h.
template <class Fr, class Dom>
std::vector<Fr> witness_map_on(const Dom& dom, const Csr<Fr>& A, const Csr<Fr>& B, const Csr<Fr>& C, const Fr* z,
                               size_t num_inputs, int nthreads) {
  size_t nc = A.rows;
  size_t n = dom.n;
  std::vector<Fr> a(n, Fr::zero()), b(n, Fr::zero()), c(n, Fr::zero());
  Radix2Domain<Fr>::parallel_for(nc, nthreads, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; i++) { a[i] = A.dot(i, z); b[i] = B.dot(i, z); c[i] = C.dot(i, z); }
  });
  for (size_t j = 0; j < num_inputs; j++) a[nc + j] = z[j];
  dom.ifft(a.data(), nthreads); dom.ifft(b.data(), nthreads);
  dom.coset_fft(a.data(), nthreads); dom.coset_fft(b.data(), nthreads);
  dom.ifft(c.data(), nthreads); dom.coset_fft(c.data(), nthreads);
  Fr zinv = dom.vanishing_on_coset().inv();
  Radix2Domain<Fr>::parallel_for(n, nthreads, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; i++) a[i] = (a[i] * b[i] - c[i]) * zinv;
  });
  dom.coset_ifft(a.data(), nthreads);
  return a;
}
// domain = GeneralEvaluationDomain::new(num_constraints + num_inputs): radix-2 if it fits the 2-adicity, else mixed radix
template <class Fr>
std::vector<Fr> witness_map(const Csr<Fr>& A, const Csr<Fr>& B, const Csr<Fr>& C, const Fr* z, size_t num_inputs, int nthreads) {
  typedef typename Fr::Params P;
  size_t need = A.rows + num_inputs;
  int log_n = domain_log_for(need);
  if (log_n <= P::TWO_ADICITY) return witness_map_on(Radix2Domain<Fr>(log_n), A, B, C, z, num_inputs, nthreads);
  size_t q = (P::ID == 0) ? 7 : (P::ID == 2) ? 5 : 0, m = 0;
  int a = 0;
  if (!q || !best_mixed_domain_size(need, q, 2, P::TWO_ADICITY, &m, &a)) return {};
  return witness_map_on(MixedDomain<Fr>(m, a), A, B, C, z, num_inputs, nthreads);
}

// ------------------------------------------------------------------------------------------------ keys / proof
template <class C>
struct G16 {
  typedef typename C::Fq Fq;
  typedef typename C::Fr Fr;
  typedef typename C::G2F E;
  typedef Pairing<C> PE;
  static constexpr int NS = Fr::N;

  struct PK {
    Affine<Fq> alpha_g1, beta_g1, delta_g1;
    Affine<E> beta_g2, delta_g2;
    const Affine<Fq>*a_query, *b_g1_query, *h_query, *l_query;
    const Affine<E>* b_g2_query;
    size_t m, num_inputs, h_len, l_len;
  };
  struct Proof { Affine<Fq> a; Affine<E> b; Affine<Fq> c; };

  static std::vector<u64> to_repr(const Fr* v, size_t n) {
    std::vector<u64> out(n * NS);
    for (size_t i = 0; i < n; i++) v[i].to_canonical(&out[i * NS]);
    return out;
  }
  template <class F>
  static Jac<F> msm(const Group<F>& G, const Affine<F>* bases, const u64* repr, size_t n, int nthreads) {
    return msm_pippenger(G, bases, repr, NS, n, Fr::Params::BITS, nthreads);
  }

  // create_proof (Appendix A.1) with r, s supplied by the caller (upstream draws them from rng first)
  static Proof prove(const PK& pk, const Csr<Fr>& A, const Csr<Fr>& B, const Csr<Fr>& Cm, const Fr* z,
                     const Fr& r, const Fr& s, int nthreads, std::vector<Fr>* h_out = nullptr) {
    Group<Fq> G1 = C::G1();
    Group<E> G2 = C::G2();
    std::vector<Fr> h = witness_map(A, B, Cm, z, pk.num_inputs, nthreads);
    if (h_out) *h_out = h;
    size_t hl = std::min(pk.h_len, h.size());
    std::vector<u64> h_repr = to_repr(h.data(), hl);
    Jac<Fq> h_acc = msm(G1, pk.h_query, h_repr.data(), hl, nthreads);
    std::vector<u64> z_repr = to_repr(z, pk.m);
    const u64* aux = z_repr.data() + pk.num_inputs * NS;
    Jac<Fq> l_acc = msm(G1, pk.l_query, aux, std::min(pk.l_len, pk.m - pk.num_inputs), nthreads);
    const u64* asg = z_repr.data() + NS;  // input[1..] ++ aux
    u64 r_repr[NS], s_repr[NS], rs_repr[NS];
    r.to_canonical(r_repr); s.to_canonical(s_repr); (r * s).to_canonical(rs_repr);
    Jac<Fq> delta1 = Jac<Fq>::from_affine(pk.delta_g1);
    Jac<E> delta2 = Jac<E>::from_affine(pk.delta_g2);
    auto coeff1 = [&](const Jac<Fq>& init, const Affine<Fq>* q, const Affine<Fq>& vk) {
      Jac<Fq> acc = msm(G1, q + 1, asg, pk.m - 1, nthreads);
      Jac<Fq> res = G1.madd(init, q[0]);
      res = G1.add(res, acc);
      return G1.madd(res, vk);
    };
    Jac<Fq> g_a = coeff1(G1.mul(delta1, r_repr, NS), pk.a_query, pk.alpha_g1);
    Jac<Fq> g1_b = coeff1(G1.mul(delta1, s_repr, NS), pk.b_g1_query, pk.beta_g1);
    Jac<E> g2_b;
    {
      Jac<E> acc = msm(G2, pk.b_g2_query + 1, asg, pk.m - 1, nthreads);
      Jac<E> res = G2.madd(G2.mul(delta2, s_repr, NS), pk.b_g2_query[0]);
      res = G2.add(res, acc);
      g2_b = G2.madd(res, pk.beta_g2);
    }
    Jac<Fq> rs_delta = G1.mul(delta1, rs_repr, NS);
    Jac<Fq> g_c = G1.mul(g_a, s_repr, NS);
    g_c = G1.add(g_c, G1.mul(g1_b, r_repr, NS));
    g_c = G1.add(g_c, G1.neg(rs_delta));
    g_c = G1.add(g_c, l_acc);
    g_c = G1.add(g_c, h_acc);
    return {G1.to_affine(g_a), G2.to_affine(g2_b), G1.to_affine(g_c)};
  }

  // generate_parameters with fixed toxic waste (alpha, beta, gamma, delta, tau) and the standard
  // generators.  Plain double-and-add per query element: meant for n <= 2^12 test circuits.
  struct Keys {
    std::vector<Affine<Fq>> a_query, b_g1_query, h_query, l_query, gamma_abc_g1;
    std::vector<Affine<E>> b_g2_query;
    Affine<Fq> alpha_g1, beta_g1, delta_g1;
    Affine<E> beta_g2, delta_g2, gamma_g2;
  };
  static Keys setup(const Csr<Fr>& A, const Csr<Fr>& B, const Csr<Fr>& Cm, size_t m, size_t num_inputs,
                    const Fr toxic[5], int nthreads) {
    const Fr &alpha = toxic[0], &beta = toxic[1], &gamma = toxic[2], &delta = toxic[3], &tau = toxic[4];
    size_t nc = A.rows;
    int log_n = domain_log_for(nc + num_inputs);
    Radix2Domain<Fr> dom(log_n);
    size_t n = dom.n;
    // Lagrange basis at tau: L_i = Z(tau) w^i / (n (tau - w^i))
    Fr tn = tau; for (int i = 0; i < log_n; i++) tn = tn.sqr();
    Fr zt = tn - Fr::one();
    std::vector<Fr> L(n);
    Fr wi = Fr::one();
    for (size_t i = 0; i < n; i++) { L[i] = zt * wi * dom.size_inv * (tau - wi).inv(); wi = wi * dom.group_gen; }
    std::vector<Fr> At(m, Fr::zero()), Bt(m, Fr::zero()), Ct(m, Fr::zero());
    for (size_t j = 0; j < nc; j++) {
      for (u64 k = A.row_ptr[j]; k < A.row_ptr[j + 1]; k++) At[A.col[k]] = At[A.col[k]] + A.coeff[k] * L[j];
      for (u64 k = B.row_ptr[j]; k < B.row_ptr[j + 1]; k++) Bt[B.col[k]] = Bt[B.col[k]] + B.coeff[k] * L[j];
      for (u64 k = Cm.row_ptr[j]; k < Cm.row_ptr[j + 1]; k++) Ct[Cm.col[k]] = Ct[Cm.col[k]] + Cm.coeff[k] * L[j];
    }
    for (size_t i = 0; i < num_inputs; i++) At[i] = At[i] + L[nc + i];
    Group<Fq> G1 = C::G1();
    Group<E> G2 = C::G2();
    Jac<Fq> g1 = Jac<Fq>::from_affine(C::g1());
    Jac<E> g2 = Jac<E>::from_affine(C::g2());
    auto mul1 = [&](const Fr& k) { u64 rp[NS]; k.to_canonical(rp); return G1.to_affine(G1.mul(g1, rp, NS)); };
    auto mul2 = [&](const Fr& k) { u64 rp[NS]; k.to_canonical(rp); return G2.to_affine(G2.mul(g2, rp, NS)); };
    Keys K;
    K.alpha_g1 = mul1(alpha); K.beta_g1 = mul1(beta); K.delta_g1 = mul1(delta);
    K.beta_g2 = mul2(beta); K.delta_g2 = mul2(delta); K.gamma_g2 = mul2(gamma);
    Fr dinv = delta.inv(), ginv = gamma.inv();
    // every query is a fixed-base batch (ark-groth16 generate_parameters -> FixedBaseMSM::multi_scalar_mul)
    std::vector<Fr> tp(n - 1), lt(m);
    { Fr cur = zt * dinv; for (size_t i = 0; i + 1 < n; i++) { tp[i] = cur; cur = cur * tau; } }
    for (size_t i = 0; i < m; i++) lt[i] = (beta * At[i] + alpha * Bt[i] + Ct[i]) * (i < num_inputs ? ginv : dinv);
    auto batch1 = [&](const std::vector<Fr>& k) {
      std::vector<u64> can(k.size() * NS);
      for (size_t i = 0; i < k.size(); i++) k[i].to_canonical(&can[i * NS]);
      return fixed_base_msm(G1, C::g1(), can.data(), NS, k.size(), Fr::Params::BITS, nthreads);
    };
    K.a_query = batch1(At);
    K.b_g1_query = batch1(Bt);
    {
      std::vector<u64> can(m * NS);
      for (size_t i = 0; i < m; i++) Bt[i].to_canonical(&can[i * NS]);
      K.b_g2_query = fixed_base_msm(G2, C::g2(), can.data(), NS, m, Fr::Params::BITS, nthreads);
    }
    K.h_query = batch1(tp);
    auto lq = batch1(lt);
    K.gamma_abc_g1.assign(lq.begin(), lq.begin() + num_inputs);
    K.l_query.assign(lq.begin() + num_inputs, lq.end());
    return K;
  }

  // e(A,B) == e(alpha,beta) e(sum x_i gamma_abc_i, gamma) e(C,delta), x_0 = 1
  static bool verify(const Affine<Fq>& alpha_g1, const Affine<E>& beta_g2, const Affine<E>& gamma_g2,
                     const Affine<E>& delta_g2, const Affine<Fq>* gamma_abc, size_t num_inputs,
                     const Fr* public_inputs /* num_inputs - 1 */, const Proof& pr) {
    Group<Fq> G1 = C::G1();
    Jac<Fq> acc = Jac<Fq>::from_affine(gamma_abc[0]);
    for (size_t i = 1; i < num_inputs; i++) {
      u64 rp[NS]; public_inputs[i - 1].to_canonical(rp);
      acc = G1.add(acc, G1.mul(Jac<Fq>::from_affine(gamma_abc[i]), rp, NS));
    }
    auto lhs = PE::pairing(pr.a, pr.b);
    auto rhs = PE::pairing(alpha_g1, beta_g2) * PE::pairing(G1.to_affine(acc), gamma_g2) * PE::pairing(pr.c, delta_g2);
    return lhs == rhs;
  }
};

}  // namespace orc
