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

}  // namespace orc
