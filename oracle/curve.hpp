// TEST INFRASTRUCTURE ONLY -- CPU restatement of ark-ec `short_weierstrass_jacobian` (a != 0 curves)
// and of `VariableBaseMSM::multi_scalar_mul` as reached from src/ec_cycle_pcd/mod.rs:171,179
// (SNARK::prove -> Groth16 create_proof -> 4 G1 MSMs + 1 G2 MSM).  SURVEY.md Appendix A.4/A.5.
// PARITY UNPINNED (see field.hpp / DESIGN.md); validated against oracle/pyoracle.py.
#pragma once
#include <functional>
#include <thread>
#include <vector>

#include "field.hpp"

namespace orc {

template <class F>
struct Affine {
  F x, y;
  bool inf;
  static Affine infinity() { return {F::zero(), F::zero(), true}; }
};

// Jacobian (X:Y:Z), x = X/Z^2, y = Y/Z^3, identity Z = 0
template <class F>
struct Jac {
  F X, Y, Z;
  static Jac infinity() { return {F::zero(), F::one(), F::zero()}; }
  bool is_inf() const { return Z.is_zero(); }
  static Jac from_affine(const Affine<F>& p) { return p.inf ? infinity() : Jac{p.x, p.y, F::one()}; }
};

// Group = (coordinate field F, curve coefficient a)
template <class F>
struct Group {
  F a;

  // dbl-2007-bl
  Jac<F> dbl(const Jac<F>& p) const {
    if (p.is_inf()) return p;
    F XX = p.X.sqr(), YY = p.Y.sqr(), YYYY = YY.sqr(), ZZ = p.Z.sqr();
    F S = ((p.X + YY).sqr() - XX - YYYY).dbl();
    F M = XX.dbl() + XX + a * ZZ.sqr();
    F T = M.sqr() - S.dbl();
    F Y3 = M * (S - T) - YYYY.dbl().dbl().dbl();
    F Z3 = (p.Y + p.Z).sqr() - YY - ZZ;
    return {T, Y3, Z3};
  }
  // madd-2007-bl (`add_assign_mixed`)
  Jac<F> madd(const Jac<F>& p, const Affine<F>& q) const {
    if (q.inf) return p;
    if (p.is_inf()) return {q.x, q.y, F::one()};
    F Z1Z1 = p.Z.sqr();
    F U2 = q.x * Z1Z1;
    F S2 = q.y * p.Z * Z1Z1;
    if (p.X == U2 && p.Y == S2) return dbl(p);
    F H = U2 - p.X;
    F HH = H.sqr();
    F I = HH.dbl().dbl();
    F J = H * I;
    F r = (S2 - p.Y).dbl();
    F V = p.X * I;
    F X3 = r.sqr() - J - V.dbl();
    F Y3 = r * (V - X3) - (p.Y * J).dbl();
    F Z3 = (p.Z + H).sqr() - Z1Z1 - HH;
    return {X3, Y3, Z3};
  }
  // add-2007-bl
  Jac<F> add(const Jac<F>& p, const Jac<F>& q) const {
    if (p.is_inf()) return q;
    if (q.is_inf()) return p;
    F Z1Z1 = p.Z.sqr(), Z2Z2 = q.Z.sqr();
    F U1 = p.X * Z2Z2, U2 = q.X * Z1Z1;
    F S1 = p.Y * q.Z * Z2Z2, S2 = q.Y * p.Z * Z1Z1;
    if (U1 == U2 && S1 == S2) return dbl(p);
    F H = U2 - U1;
    F I = H.dbl().sqr();
    F J = H * I;
    F r = (S2 - S1).dbl();
    F V = U1 * I;
    F X3 = r.sqr() - J - V.dbl();
    F Y3 = r * (V - X3) - (S1 * J).dbl();
    F Z3 = ((p.Z + q.Z).sqr() - Z1Z1 - Z2Z2) * H;
    return {X3, Y3, Z3};
  }
  Jac<F> neg(const Jac<F>& p) const { return {p.X, p.Y.neg(), p.Z}; }
  Affine<F> to_affine(const Jac<F>& p) const {
    if (p.is_inf()) return Affine<F>::infinity();
    F zi = p.Z.inv(), zi2 = zi.sqr();
    return {p.X * zi2, p.Y * zi2 * zi, false};
  }
  // scalar given as canonical little-endian limbs
  Jac<F> mul(const Jac<F>& p, const u64* k, int nlimbs) const {
    Jac<F> r = Jac<F>::infinity();
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
      r = dbl(r);
      if ((k[i / 64] >> (i % 64)) & 1) r = add(r, p);
    }
    return r;
  }
};

// ------------------------------------------------------------------------------------------------
// VariableBaseMSM::multi_scalar_mul  (SURVEY.md Appendix A.4), threads over windows like rayon.
//   scalars: canonical limbs (NS u64 each);   num_bits = modulus bits of the scalar field.
inline int ln_without_floats(size_t a) {
  int lg = 0;
  while (((size_t)1 << lg) < a) lg++;  // ceil(log2 a)
  return lg * 69 / 100;
}
inline int upstream_window(size_t size) { return size < 32 ? 3 : ln_without_floats(size) + 2; }

template <class F>
Jac<F> msm_pippenger(const Group<F>& G, const Affine<F>* bases, const u64* scalars, int NS, size_t size,
                     int num_bits, int nthreads, int c_override = 0) {
  const int c = c_override ? c_override : upstream_window(size);
  std::vector<int> starts;
  for (int w = 0; w < num_bits; w += c) starts.push_back(w);
  const int W = (int)starts.size();
  std::vector<Jac<F>> window_sums(W);

  auto is_one = [&](const u64* s) { if (s[0] != 1) return false; for (int i = 1; i < NS; i++) if (s[i]) return false; return true; };
  auto is_zero = [&](const u64* s) { u64 o = 0; for (int i = 0; i < NS; i++) o |= s[i]; return o == 0; };
  auto digit = [&](const u64* s, int w_start) -> u64 {
    int limb = w_start / 64, off = w_start % 64;
    u64 d = s[limb] >> off;
    if (off + c > 64 && limb + 1 < NS) d |= s[limb + 1] << (64 - off);
    return d & (((u64)1 << c) - 1);
  };
  auto do_window = [&](int wi) {
    int w_start = starts[wi];
    Jac<F> res = Jac<F>::infinity();
    std::vector<Jac<F>> buckets(((size_t)1 << c) - 1, Jac<F>::infinity());
    for (size_t i = 0; i < size; i++) {
      const u64* s = scalars + i * NS;
      if (is_zero(s)) continue;
      if (is_one(s)) { if (w_start == 0) res = G.madd(res, bases[i]); continue; }
      u64 d = digit(s, w_start);
      if (d != 0) buckets[d - 1] = G.madd(buckets[d - 1], bases[i]);
    }
    Jac<F> running = Jac<F>::infinity();
    for (size_t b = buckets.size(); b-- > 0;) { running = G.add(running, buckets[b]); res = G.add(res, running); }
    window_sums[wi] = res;
  };
  if (nthreads <= 1) {
    for (int wi = 0; wi < W; wi++) do_window(wi);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++)
      th.emplace_back([&, t]() { for (int wi = t; wi < W; wi += nthreads) do_window(wi); });
    for (auto& x : th) x.join();
  }
  // lowest + fold(windows[1..].rev()): total += w; total = 2^c * total
  Jac<F> total = Jac<F>::infinity();
  for (int wi = W - 1; wi >= 1; wi--) {
    total = G.add(total, window_sums[wi]);
    for (int k = 0; k < c; k++) total = G.dbl(total);
  }
  return G.add(window_sums[0], total);
}

// ------------------------------------------------------------------------------------------------
// ark-ec `FixedBaseMSM` (fixed_base.rs): get_mul_window_size, get_window_table, windowed_mul, multi_scalar_mul, followed
// by batch normalisation into affine -- what ark-groth16 `generate_parameters` (circuit_specific_setup,
// src/ec_cycle_pcd/mod.rs:69,78) calls for every query of the key.  out[i] = k_i * g.
inline int fixed_base_window(size_t n) { return n < 32 ? 3 : ln_without_floats(n); }

template <class F>
std::vector<Affine<F>> fixed_base_msm(const Group<F>& G, const Affine<F>& g, const u64* scalars, int NS, size_t n, int scalar_bits,
                                      int nthreads) {
  const int window = fixed_base_window(n);
  const int outerc = (scalar_bits + window - 1) / window;
  const size_t last_in_window = (size_t)1 << (scalar_bits - (outerc - 1) * window);
  // table[outer][inner] = inner * 2^(window * outer) * g, normalised to affine
  std::vector<std::vector<Affine<F>>> table(outerc);
  Jac<F> g_outer = Jac<F>::from_affine(g);
  std::vector<Jac<F>> g_outers(outerc);
  for (int outer = 0; outer < outerc; outer++) {
    g_outers[outer] = g_outer;
    for (int k = 0; k < window; k++) g_outer = G.dbl(g_outer);
  }
  auto run = [&](size_t count, const std::function<void(size_t)>& fn) {
    if (nthreads <= 1) { for (size_t i = 0; i < count; i++) fn(i); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back([&, t]() { for (size_t i = t; i < count; i += nthreads) fn(i); });
    for (auto& x : th) x.join();
  };
  run(outerc, [&](size_t outer) {
    const size_t cur = (outer == (size_t)outerc - 1) ? last_in_window : ((size_t)1 << window);
    std::vector<Jac<F>> row(cur);
    Jac<F> g_inner = Jac<F>::infinity();
    for (size_t inner = 0; inner < cur; inner++) { row[inner] = g_inner; g_inner = G.add(g_inner, g_outers[outer]); }
    // batch normalisation (Montgomery's trick over the non-zero Z)
    table[outer].resize(cur);
    std::vector<F> pre(cur);
    F acc = F::one();
    for (size_t i = 0; i < cur; i++) { pre[i] = acc; if (!row[i].is_inf()) acc = acc * row[i].Z; }
    F inv = acc.inv();
    for (size_t i = cur; i-- > 0;) {
      if (row[i].is_inf()) { table[outer][i] = Affine<F>::infinity(); continue; }
      F zi = inv * pre[i];
      inv = inv * row[i].Z;
      F zi2 = zi.sqr();
      table[outer][i] = {row[i].X * zi2, row[i].Y * zi2 * zi, false};
    }
  });
  std::vector<Affine<F>> out(n);
  const size_t blocks = (n + 255) / 256;
  run(blocks, [&](size_t blk) {
    for (size_t i = blk * 256; i < std::min(n, blk * 256 + 256); i++) {
      const u64* s = scalars + i * NS;
      Jac<F> res = Jac<F>::infinity();
      for (int outer = 0; outer < outerc; outer++) {
        size_t inner = 0;
        for (int b = 0; b < window; b++) {
          const int bit = outer * window + b;
          if (bit < scalar_bits && ((s[bit / 64] >> (bit % 64)) & 1)) inner |= (size_t)1 << b;
        }
        res = G.madd(res, table[outer][inner]);
      }
      out[i] = G.to_affine(res);
    }
  });
  return out;
}

}  // namespace orc
