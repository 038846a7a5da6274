// Second half of the host-side harness (see hostcheck.hip): the pairing templates, the wave-per-pairing VM interpreted on the host, and
// Fp::from_signed_sum -- a translation unit of its own so that the two compile side by side (each takes minutes: the unrolled 27-limb
// templates for an x86 host).
#include <stdint.h>
#include "../../pcd_amd/csrc/ec.hip.h"
using namespace pcd;

#include "../../pcd_amd/csrc/pairing.hip.h"
template <class PC>
static void pairing_host(const uint32_t* g1, const uint32_t* g2, uint32_t* out) {
  typedef Pairing<PC> PE;
  typename PE::Frob t;
  frob_init<typename PE::Fq, PE::K, PC::NR>(t);
  auto f = PE::miller_loop(Aff<typename PE::Fq>::from_abi(g1), Aff<typename PE::E>::from_abi(g2));
  PE::final_exponentiation(f, t).to_abi(out);
}
extern "C" int hc_pairing(int curve, const uint32_t* g1, const uint32_t* g2, uint32_t* out) {
  switch (curve) {
    case 0: pairing_host<PC_MNT4_298>(g1, g2, out); break;
    case 1: pairing_host<PC_MNT6_298>(g1, g2, out); break;
    case 2: pairing_host<PC_MNT4_753>(g1, g2, out); break;
    case 3: pairing_host<PC_MNT6_753>(g1, g2, out); break;
    default: return -1;
  }
  return 0;
}

// ---- the wave-per-pairing VM (pairing_vm.hip.h) on the host: the generated programs and scripts interpreted with the device's own MUL /
// SQR / LIN arithmetic, lanes one after the other (every lane of a step reads before any lane writes, as in lockstep), the same drivers as
// vm_miller_kernel / vm_final_exp_kernel of inst_pairing.hip.  n pairs (G1 points Jacobian: x, y, z) -> final_exp(prod miller).
#include <vector>
#include "../../pcd_amd/csrc/pairing_vm.hip.h"
template <class PC, class VG>
struct HostVm {
  typedef typename Pairing<PC>::Fq Fq;
  typedef VmArith<Fq> A;
  std::vector<uint32_t> regs;
  uint64_t bank = 0;
  VmTables tb;
  HostVm(const VmTables& t) : regs((size_t)VG::NREGS * A::STRIDE, 0u), tb(t) {
    for (int c = 0; c < VG::NCONST; c++) { Fq v; for (int i = 0; i < Fq::N; i++) v.v[i] = tb.consts[c * Fq::N + i]; A::st(regs.data(), VG::CONST_BASE + c, v); }
  }
  Fq get_state(int s) { return A::ld(regs.data(), A::template reg_of_g<VG>(A::state_op((uint32_t)s), bank)); }
  void set_state(int s, const Fq& v) { A::st(regs.data(), A::template reg_of_g<VG>(A::state_op((uint32_t)s), bank), v); }
  void set_reg(int r, const Fq& v) { A::st(regs.data(), (uint32_t)r, v); }
  void run(int pid) {
    const uint32_t first = tb.progs[3 * pid], cnt = tb.progs[3 * pid + 1];
    static const uint32_t zero12[12] = {0};
    for (uint32_t s = first; s < first + cnt; s++) {
      const uint32_t kind = tb.steps[3 * s] & 0xFFu /* bits 8 up: the step's largest LIN term count, for the device */, off = tb.steps[3 * s + 1], n = tb.steps[3 * s + 2];
      std::vector<Fq> res(n);
      for (uint32_t l = 0; l < n; l++) {
        const uint32_t* w = tb.code + (size_t)(off + l) * 12;
        if ((w[0] & 0xFFu) == 0xFFu) continue;
        const uint32_t* w2 = ((w[0] >> 8) & 0xFFu) > 8 ? w + 12 : zero12;
        res[l] = kind == 1 ? A::template mul<VG>(w, regs.data(), bank) : kind == 2 ? A::template sqr<VG>(w, regs.data(), bank) : A::template lin<VG>(w, w2, regs.data(), bank);
      }
      for (uint32_t l = 0; l < n; l++) {
        const uint32_t* w = tb.code + (size_t)(off + l) * 12;
        if ((w[0] & 0xFFu) != 0xFFu) A::st(regs.data(), A::template reg_of_g<VG>(w[0] >> 16, bank), res[l]);
      }
    }
    bank ^= (uint64_t)tb.progs[3 * pid + 2];
  }
  void run_script() {
    for (uint32_t i = 0; i < tb.script_len; i++) {
      const uint32_t e = (tb.script[i >> 2] >> (8 * (i & 3))) & 0xFFu;
      if (e >= 0xF0u) bank = (bank & 0xFFFFFFFFull) | ((uint64_t)(e - 0xF0u) << 32);
      else if (e == 0xEFu) set_reg(VG::R_FT1_0, A::ld(regs.data(), (uint32_t)VG::R_NRM0).inv());  // SCRIPT_INV (Fp::inv: divsteps)
      else run((int)e);
    }
  }
};
template <class PC, class VG>
static void vm_pairing_host(const VmCurveTables& tb, int p_fe_mul, const uint32_t* g1, const uint32_t* g1z, const uint32_t* g2, int n, uint32_t* out) {
  typedef Pairing<PC> PE;
  typedef typename PE::Fq Fq;
  constexpr int K = PE::K, D = K / 2;
  std::vector<std::vector<Fq>> fs;
  for (int p = 0; p < n; p++) {
    HostVm<PC, VG> vm(tb.miller);
    const uint32_t* a = g1 + (size_t)p * 2 * Fq::ABI_WORDS;
    const uint32_t* b = g2 + (size_t)p * 2 * D * Fq::ABI_WORDS;
    vm.set_reg(VG::R_PX0, Fq::from_abi(a)); vm.set_reg(VG::R_PY0, Fq::from_abi(a + Fq::ABI_WORDS));
    vm.set_reg(VG::R_PZ0, g1z ? Fq::from_abi(g1z + (size_t)p * Fq::ABI_WORDS) : Fq::one());
    for (int j = 0; j < D; j++) { vm.set_reg(VG::R_QX0 + j, Fq::from_abi(b + j * Fq::ABI_WORDS)); vm.set_reg(VG::R_QY0 + j, Fq::from_abi(b + (D + j) * Fq::ABI_WORDS)); }
    vm.run_script();
    std::vector<Fq> f(K);
    for (int j = 0; j < K; j++) f[j] = vm.get_state(VG::S_F0 + j);
    fs.push_back(f);
  }
  HostVm<PC, VG> vm(tb.final_exp);
  for (int j = 0; j < K; j++) vm.set_state(VG::S_ACC0 + j, n ? fs[0][j] : (j ? Fq::zero() : Fq::one()));
  for (int p = 1; p < n; p++) { for (int j = 0; j < K; j++) vm.set_reg(VG::R_G0 + j, fs[p][j]); vm.run(p_fe_mul); }
  vm.run_script();
  for (int j = 0; j < K; j++) vm.get_state(VG::S_ACC0 + j).to_abi(out + ((j & 1) * D + (j >> 1)) * Fq::ABI_WORDS);
}
#define HC_VM_SET(C, S) VmTables{&vmgen::C##_##S##_progs[0][0], &vmgen::C##_##S##_steps[0][0], vmgen::C##_##S##_code, &vmgen::C##_consts[0][0], \
                                 vmgen::C##_##S##_script, 0, 0, 0, vmgen::C##_##S##_script_len, nullptr, 0, 0, 0}
#define HC_VM_TABLES(C) VmCurveTables{HC_VM_SET(C, miller), HC_VM_SET(C, final_exp)}, vmgen::C##_final_exp_P_FE_MUL
// g1z: nullable (affine points) or one Fq element per pair (Jacobian points)
extern "C" int hc_vm_pairing(int curve, const uint32_t* g1, const uint32_t* g1z, const uint32_t* g2, int n, uint32_t* out) {
  switch (curve) {
    case 0: vm_pairing_host<PC_MNT4_298, vmgen::MNT4_298>(HC_VM_TABLES(MNT4_298), g1, g1z, g2, n, out); break;
    case 1: vm_pairing_host<PC_MNT6_298, vmgen::MNT6_298>(HC_VM_TABLES(MNT6_298), g1, g1z, g2, n, out); break;
    case 2: vm_pairing_host<PC_MNT4_753, vmgen::MNT4_753>(HC_VM_TABLES(MNT4_753), g1, g1z, g2, n, out); break;
    case 3: vm_pairing_host<PC_MNT6_753, vmgen::MNT6_753>(HC_VM_TABLES(MNT6_753), g1, g1z, g2, n, out); break;
    default: return -1;
  }
  return 0;
}
// Fp::from_signed_sum on the host: out = sum c_t a_t (a: T elements in the C-ABI image, below p; c: T coefficients, sum |c| <= 4000 -- the weight 2000 on operands below 2p)
template <class FQ>
static void signed_sum_host(const uint32_t* a_abi, const int32_t* c, int T, uint32_t* out) {
  typedef Fp<FQ, false> F;
  int64_t s[F::N] = {0};
  for (int t = 0; t < T; t++) {
    const F a = F::from_abi(a_abi + (size_t)t * F::ABI_WORDS);
    for (int i = 0; i < F::N; i++) s[i] += (int64_t)c[t] * (int64_t)a.v[i];
  }
  const F o = F::from_signed_sum(s);
  // the contract: limbs normalised and the value below 2p (everything downstream assumes it)
  bool ok = true, below = false;
  for (int i = 0; i < F::N - 1; i++) ok = ok && o.v[i] <= F::MASK;
  for (int i = F::N - 1; i >= 0 && !below; i--) {
    if (o.v[i] < FQ::mod2(i)) below = true;
    else if (o.v[i] > FQ::mod2(i)) break;
  }
  if (!ok || !below) { for (int i = 0; i < F::ABI_WORDS; i++) out[i] = 0xFFFFFFFFu; return; }
  o.to_abi(out);
}
extern "C" int hc_signed_sum(int field, const uint32_t* a_abi, const int32_t* c, int T, uint32_t* out) {
  switch (field) {
    case 0: signed_sum_host<F298A>(a_abi, c, T, out); break;
    case 1: signed_sum_host<F298B>(a_abi, c, T, out); break;
    case 2: signed_sum_host<F753A>(a_abi, c, T, out); break;
    case 3: signed_sum_host<F753B>(a_abi, c, T, out); break;
    default: return -1;
  }
  return 0;
}
