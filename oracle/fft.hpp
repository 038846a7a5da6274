// TEST INFRASTRUCTURE ONLY -- CPU restatement of ark-poly `Radix2EvaluationDomain`
// {fft, ifft, coset_fft, coset_ifft}_in_place, reached from
// src/ec_cycle_pcd/mod.rs:171,179 through Groth16's R1CSToQAP::witness_map.
// PARITY UNPINNED (see field.hpp / DESIGN.md); validated against oracle/pyoracle.py (naive DFT).
#pragma once
#include <thread>
#include <vector>

#include "field.hpp"

namespace orc {

template <class F>
struct Radix2Domain {
  int log_n;
  size_t n;
  F group_gen, group_gen_inv, size_inv, gen, gen_inv;

  explicit Radix2Domain(int log_n_) : log_n(log_n_), n((size_t)1 << log_n_) {
    typedef typename F::Params P;
    // group_gen = TWO_ADIC_ROOT_OF_UNITY ^ (2^(TWO_ADICITY - log_n))
    group_gen = F::two_adic_root();
    for (int i = log_n; i < P::TWO_ADICITY; i++) group_gen = group_gen.sqr();
    group_gen_inv = group_gen.inv();
    size_inv = F::from_u64((u64)n).inv();
    gen = F::generator();
    gen_inv = gen.inv();
  }

  static void parallel_for(size_t count, int nthreads, const std::function<void(size_t, size_t)>& fn) {
    if (nthreads <= 1 || count < 1024) { fn(0, count); return; }
    std::vector<std::thread> th;
    size_t chunk = (count + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; t++) {
      size_t lo = t * chunk, hi = std::min(count, lo + chunk);
      if (lo < hi) th.emplace_back([=, &fn]() { fn(lo, hi); });
    }
    for (auto& x : th) x.join();
  }

  // in-order in / in-order out: bit-reversal permutation, then log_n DIT butterfly layers
  void transform(F* a, const F& omega, int nthreads) const {
    for (size_t k = 0; k < n; k++) {
      size_t rk = 0;
      for (int b = 0; b < log_n; b++) rk |= ((k >> b) & 1) << (log_n - 1 - b);
      if (k < rk) std::swap(a[k], a[rk]);
    }
    // twiddle table w^i, i < n/2
    std::vector<F> tw(n / 2 ? n / 2 : 1);
    tw[0] = F::one();
    for (size_t i = 1; i < n / 2; i++) tw[i] = tw[i - 1] * omega;
    for (int s = 1; s <= log_n; s++) {
      size_t m = (size_t)1 << s, half = m >> 1, stride = n / m;
      parallel_for(n / 2, nthreads, [&](size_t lo, size_t hi) {
        for (size_t idx = lo; idx < hi; idx++) {
          size_t blk = idx / half, j = idx % half;
          size_t i0 = blk * m + j, i1 = i0 + half;
          F t = a[i1] * tw[j * stride];
          a[i1] = a[i0] - t;
          a[i0] = a[i0] + t;
        }
      });
    }
  }
  void distribute_powers(F* a, const F& g, int nthreads) const {
    parallel_for(n, nthreads, [&](size_t lo, size_t hi) {
      u64 e[1] = {lo};
      F cur = g.pow(e, 1);
      for (size_t i = lo; i < hi; i++) { a[i] = a[i] * cur; cur = cur * g; }
    });
  }
  void fft(F* a, int nthreads = 1) const { transform(a, group_gen, nthreads); }
  void ifft(F* a, int nthreads = 1) const {
    transform(a, group_gen_inv, nthreads);
    parallel_for(n, nthreads, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) a[i] = a[i] * size_inv; });
  }
  void coset_fft(F* a, int nthreads = 1) const { distribute_powers(a, gen, nthreads); fft(a, nthreads); }
  void coset_ifft(F* a, int nthreads = 1) const { ifft(a, nthreads); distribute_powers(a, gen_inv, nthreads); }
  // Z(g) = g^n - 1 on the coset (constant there)
  F vanishing_on_coset() const {
    F t = gen;
    for (int i = 0; i < log_n; i++) t = t.sqr();
    return t - F::one();
  }
};

// ark-poly `MixedRadixEvaluationDomain` (sizes 2^a q^b, q = 7 for MNT4-298 Fq / 5 for MNT4-753 Fq, b <= 2), chosen by
// `GeneralEvaluationDomain::new` when the radix-2 domain would exceed the field's 2-adicity -- the help proof of a PCD
// step (src/ec_cycle_pcd/mod.rs:179) as soon as the help circuit has more than 2^17 / 2^15 rows.  Same DFT
// convention as the radix-2 domain with group_gen = GENERATOR^((p-1)/n); computed here as q^b-point DFTs across
// 2^a-point radix-2 transforms.
template <class F>
struct MixedDomain {
  size_t n, m;  // n = m * 2^a, m = q^b
  int a;
  F group_gen, group_gen_inv, size_inv, gen, gen_inv;
  Radix2Domain<F> sub;

  static F pow_big_div(const F& base, size_t divisor) {  // base^((p-1)/divisor)
    typedef typename F::Params P;
    constexpr int N = F::N;
    u64 pm1[N]; memcpy(pm1, P::MOD, sizeof pm1); pm1[0] -= 1;
    u64 e[N]; u128 rem = 0;
    for (int i = N - 1; i >= 0; i--) { u128 cur = (rem << 64) | pm1[i]; e[i] = (u64)(cur / divisor); rem = cur % divisor; }
    return base.pow(e, N);
  }
  MixedDomain(size_t m_, int a_) : n(m_ << a_), m(m_), a(a_), sub(a_) {
    group_gen = pow_big_div(F::generator(), n);
    group_gen_inv = group_gen.inv();
    size_inv = F::from_u64((u64)n).inv();
    gen = F::generator();
    gen_inv = gen.inv();
  }
  void transform(F* x, const F& w, int nthreads) const {
    const size_t N2 = (size_t)1 << a;
    u64 em[1] = {(u64)N2};
    F wm = w.pow(em, 1);  // m-th root
    std::vector<F> y(n);
    // step 1: m-point DFTs over j1 (stride N2), twiddled by w^(j2 k1)
    Radix2Domain<F>::parallel_for(N2, nthreads, [&](size_t lo, size_t hi) {
      for (size_t j2 = lo; j2 < hi; j2++) {
        u64 ej[1] = {(u64)j2};
        F t = w.pow(ej, 1), tw = F::one(), wk = F::one();
        for (size_t k1 = 0; k1 < m; k1++) {
          F acc = F::zero(), pw = F::one();
          for (size_t j1 = 0; j1 < m; j1++) { acc = acc + x[N2 * j1 + j2] * pw; pw = pw * wk; }
          y[k1 * N2 + j2] = acc * tw;
          tw = tw * t;
          wk = wk * wm;
        }
      }
    });
    // step 2: radix-2 transforms of the rows (root w^m), step 3: X[k1 + m k2] = Z[k1][k2]
    F w2 = w; { u64 e2[1] = {(u64)m}; w2 = w.pow(e2, 1); }
    for (size_t k1 = 0; k1 < m; k1++) sub.transform(&y[k1 * N2], w2, nthreads);
    for (size_t k1 = 0; k1 < m; k1++) for (size_t k2 = 0; k2 < N2; k2++) x[k1 + m * k2] = y[k1 * N2 + k2];
  }
  void distribute_powers(F* x, const F& g, int nthreads) const {
    Radix2Domain<F>::parallel_for(n, nthreads, [&](size_t lo, size_t hi) {
      u64 e[1] = {lo}; F cur = g.pow(e, 1);
      for (size_t i = lo; i < hi; i++) { x[i] = x[i] * cur; cur = cur * g; }
    });
  }
  void fft(F* x, int nt = 1) const { transform(x, group_gen, nt); }
  void ifft(F* x, int nt = 1) const { transform(x, group_gen_inv, nt); for (size_t i = 0; i < n; i++) x[i] = x[i] * size_inv; }
  void coset_fft(F* x, int nt = 1) const { distribute_powers(x, gen, nt); fft(x, nt); }
  void coset_ifft(F* x, int nt = 1) const { ifft(x, nt); distribute_powers(x, gen_inv, nt); }
  F vanishing_on_coset() const { u64 e[1] = {(u64)n}; return gen.pow(e, 1) - F::one(); }
};

// ark-poly `best_mixed_domain_size`
inline size_t best_mixed_domain_size(size_t min_size, size_t q, int q_adicity, int two_adicity, size_t* m_out, int* a_out) {
  size_t best = 0;
  for (int b = 0; b <= q_adicity; b++) {
    size_t r = 1; for (int i = 0; i < b; i++) r *= q;
    size_t mm = r; int a = 0;
    while (r < min_size) { r *= 2; a++; }
    if (a <= two_adicity && (best == 0 || r < best)) { best = r; *m_out = mm; *a_out = a; }
  }
  return best;
}

}  // namespace orc
