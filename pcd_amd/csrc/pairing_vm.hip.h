// The MNT4 / MNT6 ate pairing with ONE WAVE PER PAIRING (K6 of SURVEY.md section 8; the latency form).
//
// Replaces, for small batches, the one-lane-per-pairing kernels of pairing.hip.h behind the same entry points
// (`PairingEngine::{miller_loop, final_exponentiation}` under `Groth16::verify`, /root/reference src/ec_cycle_pcd/mod.rs:239, and
// `process_vk`, mod.rs:71).  A pairing is ~13 000 (298-bit) field products of which a lane can only run one at a time; a verification
// has three or four pairings, so the lane-per-pairing kernels used 4 of the chip's 65 536 lanes and took 28 ms where one host core
// takes 5.  Here a wave is a little vector machine over Fq:
//
//   * every value is a REGISTER IN LDS (N 28-bit limbs, 16-byte aligned records);
//   * a program is a list of steps; in a step up to 64 lanes execute ONE instruction each --
//       MUL  dst = (sum_{t < T} a_t b_t) / R' mod p     all products into one set of 64-bit column sums, one Montgomery reduction
//       LIN  dst = sum_{t < 8} c_t a_t mod p            small signed integer coefficients, one weak reduction
//     -- reading their operands from LDS and writing dst back; the independent products of a tower operation (an Fq4 product is
//     16 of them) and of a curve step sit on sibling lanes, so the dependent chain of a Miller-loop iteration is ~6 products long
//     instead of ~90;
//   * state that a program overwrites (the running point, f, a power) is double-banked per register, the bank bits travel in a
//     scalar, so no copy-back steps exist;
//   * the programs (tools/gen_pairing_vm.py -> pairing_vm_gen.h) are traced from the same formulas as pairing.hip.h, levelled and
//     register-allocated offline, and evaluated against the textbook pairing in tests/test_pairing_vm.py.
// The interpreter is one copy of the unrolled product and of the reductions (it stays in the instruction cache); everything
// that is data-dependent in a pairing -- the loop bits -- is compile-time constant, so control flow is uniform across the wave.
// The arithmetic (vm_mul / vm_lin) is __host__ __device__: tests/hostcheck runs whole programs on the host against the oracle.
#pragma once
#include "fp.hip.h"
#include "pairing_vm_gen.h"
#include "vm_tables.h"

namespace pcd {

template <class F>
struct VmArith {
  typedef typename F::Params P;
  static constexpr int N = F::N;
  static constexpr int STRIDE = (N + 3) & ~3;  // words per register record
  static constexpr uint32_t MASK = F::MASK;

  // operand -> register.  `bank`: low half the bank bit of every state slot, high half the table selector.
  template <class G>
  PCD_HD static uint32_t reg_of_g(uint32_t op, uint64_t bank) {
    // operand (16 bits) = A | B << 8 | f << 14 (tools/gen_pairing_vm.py enc_operand): register A + bit B of {bank, ~bank} + sel * stride of
    // table 0 (f & 1) / table 1 (f & 2).  Everything derived from `bank` is wave-uniform (scalar registers, once per program); what a
    // lane does per operand is four instructions -- the decode by operand space this replaces was twenty, in front of every term.
    const uint32_t lo = (uint32_t)bank, sel = (uint32_t)(bank >> 32);
    const uint64_t bankx = (uint64_t)lo | ((uint64_t)~lo << 32);
    const uint32_t s0 = sel * (uint32_t)G::TAB0_STRIDE, s1 = sel * (uint32_t)G::TAB1_STRIDE;
    const uint32_t m0 = (uint32_t)((int32_t)(op << 17) >> 31), m1 = (uint32_t)((int32_t)(op << 16) >> 31);   // (masks: a 32-bit multiply is quarter rate)
    return (op & 0xFFu) + ((uint32_t)(bankx >> ((op >> 8) & 63u)) & 1u) + (m0 & s0) + (m1 & s1);
  }
  PCD_HD static uint32_t state_op(uint32_t slot) { return 2u * slot | (slot << 8); }   // the operand of state slot `slot` in its current bank
  template <class PTR>
  PCD_HD static F ld(PTR regs, uint32_t r) {
    F v;
#pragma unroll
    for (int i = 0; i < N; i++) v.v[i] = regs[r * STRIDE + i];
    return v;
  }
  template <class PTR>
  PCD_HD static void st(PTR regs, uint32_t r, const F& v) {
#pragma unroll
    for (int i = 0; i < N; i++) regs[r * STRIDE + i] = v.v[i];
  }
#if defined(__HIP_DEVICE_COMPILE__)
  // the LDS register file: records are 16-byte aligned, so a register moves as STRIDE / 4 ds_read_b128 / ds_write_b128
  typedef __attribute__((address_space(3))) uint32_t* LdsPtr;
  typedef uint32_t V4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) V4* LdsV4;
  PCD_DEV static F ld(LdsPtr regs, uint32_t r) {
    F v;
    const LdsV4 p = (LdsV4)(regs + r * STRIDE);
#pragma unroll
    for (int k = 0; k < STRIDE / 4; k++) {
      const V4 w = p[k];
      v.v[4 * k] = w.x;
      if (4 * k + 1 < N) v.v[4 * k + 1] = w.y;
      if (4 * k + 2 < N) v.v[4 * k + 2] = w.z;
      if (4 * k + 3 < N) v.v[4 * k + 3] = w.w;
    }
    return v;
  }
  PCD_DEV static void st(LdsPtr regs, uint32_t r, const F& v) {
    const LdsV4 p = (LdsV4)(regs + r * STRIDE);
#pragma unroll
    for (int k = 0; k < STRIDE / 4; k++) {
      V4 w;
      w.x = v.v[4 * k];
      w.y = 4 * k + 1 < N ? v.v[4 * k + 1] : 0u;
      w.z = 4 * k + 2 < N ? v.v[4 * k + 2] : 0u;
      w.w = 4 * k + 3 < N ? v.v[4 * k + 3] : 0u;
      p[k] = w;
    }
  }
#endif

  // dst = a b / R', operands and result in [0, 2p): the product's 2N - 1 column sums in 64 bits (N 2^56 < 2^63 with room for the
  // reduction's terms), then one Montgomery reduction over the columns: 4 p^2 / R' + p < 2p (R' > 8p).  SQR: the same for a = b with
  // the off-diagonal products taken once, doubled (N (N + 1) / 2 + N^2 multiply-adds instead of 2 N^2).
  static_assert(vmgen::VM_TMAX == 1, "one product per MUL instruction");
  PCD_HD static F reduce_columns(uint64_t* col) {
#pragma unroll
    for (int k = 0; k < N; k++) {
      const uint32_t m = ((uint32_t)col[k] * P::INV) & MASK;
#pragma unroll
      for (int j = 0; j < N; j++) col[k + j] += (uint64_t)m * P::mod(j);
      col[k + 1] += col[k] >> 28;
    }
    F o;
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
      o.v[i] = (uint32_t)col[N + i] & MASK;
      col[N + i + 1] += col[N + i] >> 28;
    }
    o.v[N - 1] = (uint32_t)col[2 * N - 1];
    return o;
  }
  template <class G, class PTR>
  PCD_HD static F mul(const uint32_t* w, PTR regs, uint64_t bank) {
    uint64_t col[2 * N];
#pragma unroll
    for (int i = 0; i < 2 * N; i++) col[i] = 0;
    const F a = ld(regs, reg_of_g<G>(w[1] & 0xFFFFu, bank)), b = ld(regs, reg_of_g<G>(w[1] >> 16, bank));
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
      for (int j = 0; j < N; j++) col[i + j] += (uint64_t)a.v[i] * b.v[j];
    return reduce_columns(col);
  }
  template <class G, class PTR>
  PCD_HD static F sqr(const uint32_t* w, PTR regs, uint64_t bank) {
    uint64_t col[2 * N];
#pragma unroll
    for (int i = 0; i < 2 * N; i++) col[i] = 0;
    const F a = ld(regs, reg_of_g<G>(w[1] & 0xFFFFu, bank));
#pragma unroll
    for (int i = 0; i < N; i++) {
      col[2 * i] += (uint64_t)a.v[i] * a.v[i];
      const uint32_t a2 = a.v[i] << 1;
#pragma unroll
      for (int j = i + 1; j < N; j++) col[i + j] += (uint64_t)a2 * a.v[j];
    }
    return reduce_columns(col);
  }

  // dst = sum c_t a_t mod p (up to 16 terms, sum of |c_t| <= VM_LIN_WEIGHT), operands and result in [0, 2p): the signed limb-wise sum, then
  // Fp::from_signed_sum.  Terms 8 .. 15 sit in the continuation slot w2.
  template <class G, class PTR>
  PCD_HD static F lin(const uint32_t* w, const uint32_t* w2, PTR regs, uint64_t bank) {
    const int T = (int)((w[0] >> 8) & 0xFFu);
    int64_t s[N];
#pragma unroll
    for (int i = 0; i < N; i++) s[i] = 0;
#pragma unroll
    for (int t = 0; t < vmgen::VM_LIN_TERMS; t++) {
      if (t < T) {
        const uint32_t* ww = t < 8 ? w : w2;
        const int tt = t & 7;
        const uint32_t op = (ww[1 + tt / 2] >> (16 * (tt & 1))) & 0xFFFFu;
        const int32_t c = (int32_t)(int16_t)((ww[5 + tt / 2] >> (16 * (tt & 1))) & 0xFFFFu);
        const F a = ld(regs, reg_of_g<G>(op, bank));
        // (both factors as int32 -- limbs are below 2^29, |c| <= VM_LIN_WEIGHT -- so that a term costs ONE v_mad_i64_i32 per limb; as
        //  int64 x uint32 the same line was two multiply-adds and two moves per limb: 45 instructions per term instead of 11)
#pragma unroll
        for (int i = 0; i < N; i++) s[i] += (int64_t)c * (int64_t)(int32_t)a.v[i];
      }
    }
    return F::from_signed_sum(s);
  }
#if defined(__HIP_DEVICE_COMPILE__)
  // The same on the device, WAVE-UNIFORM: every lane runs the step's largest term count (the step table carries it; a lane with fewer
  // terms has zero coefficients on register 0 in the unused places, which add nothing), so the terms are straight-line code -- all
  // operand reads of a part in flight together, one LDS latency per part instead of one per term, and no exec-mask bookkeeping between
  // terms.  (Per-lane `if (t < T)` blocks cost ~375 cycles a term, two thirds of it the exposed ds_read and the divergence scaffolding.)
  template <class G, int TT>
  PCD_DEV static void lin_terms(const uint32_t* ww, LdsPtr regs, uint64_t bank, int64_t* s) {
    F a[TT];
    int32_t c[TT];
#pragma unroll
    for (int t = 0; t < TT; t++) {
      const uint32_t op = (ww[1 + t / 2] >> (16 * (t & 1))) & 0xFFFFu;
      c[t] = (int32_t)(int16_t)((ww[5 + t / 2] >> (16 * (t & 1))) & 0xFFFFu);
      a[t] = ld(regs, reg_of_g<G>(op, bank));
    }
#pragma unroll
    for (int t = 0; t < TT; t++) {
#pragma unroll
      for (int i = 0; i < N; i++) s[i] += (int64_t)c[t] * (int64_t)(int32_t)a[t].v[i];
    }
  }
  template <class G>
  PCD_DEV static void lin_part(const uint32_t* ww, LdsPtr regs, uint64_t bank, uint32_t cnt /* wave-uniform, 1 .. 8 */, int64_t* s) {
    switch (cnt) {
      case 1: lin_terms<G, 1>(ww, regs, bank, s); break;
      case 2: lin_terms<G, 2>(ww, regs, bank, s); break;
      case 3: lin_terms<G, 3>(ww, regs, bank, s); break;
      case 4: lin_terms<G, 4>(ww, regs, bank, s); break;
      case 5: lin_terms<G, 5>(ww, regs, bank, s); break;
      case 6: lin_terms<G, 6>(ww, regs, bank, s); break;
      case 7: lin_terms<G, 7>(ww, regs, bank, s); break;
      default: lin_terms<G, 8>(ww, regs, bank, s); break;
    }
  }
  template <class G>
  PCD_DEV static F lin_uniform(const uint32_t* w, const uint32_t* w2, LdsPtr regs, uint64_t bank, uint32_t tmax) {
    int64_t s[N];
#pragma unroll
    for (int i = 0; i < N; i++) s[i] = 0;
    lin_part<G>(w, regs, bank, tmax < 8u ? tmax : 8u, s);
    if (tmax > 8u) lin_part<G>(w2, regs, bank, tmax - 8u, s);
    return F::from_signed_sum(s);
  }
#endif
};

#if defined(__HIPCC__)
// ---- the interpreter: one wave, registers in LDS ---------------------------------------------------------------------------------
template <class F, class G /* vmgen::<curve> */>
struct VmWave {
  typedef VmArith<F> A;
  typedef __attribute__((address_space(3))) uint32_t* Lds;
  Lds regs;
  Lds code, steps, progs, script;  // this kernel's tables, copied into LDS once: an instruction fetch is a ~100-cycle LDS read, not a
                                   // dependent trip to L2 in front of every one of the thousands of steps of a pairing
  uint64_t bank;                   // low half: the bank of every state slot; high half: the table selector
  uint32_t script_len;

  // LDS words: the register file, then the tables
  static constexpr uint32_t REG_WORDS = (uint32_t)G::NREGS * A::STRIDE;
  __host__ __device__ static uint32_t lds_words(const VmTables& t) { return REG_WORDS + t.ncode + 3 * t.nsteps + 3 * t.nprogs + (t.script_len + 3) / 4; }

  PCD_DEV void init(Lds r, const VmTables& t) {
    regs = r; bank = 0; script_len = t.script_len;
    code = r + REG_WORDS; steps = code + t.ncode; progs = steps + 3 * t.nsteps; script = progs + 3 * t.nprogs;
    for (uint32_t i = threadIdx.x; i < t.ncode; i += 64) code[i] = t.code[i];
    for (uint32_t i = threadIdx.x; i < 3 * t.nsteps; i += 64) steps[i] = t.steps[i];
    for (uint32_t i = threadIdx.x; i < 3 * t.nprogs; i += 64) progs[i] = t.progs[i];
    for (uint32_t i = threadIdx.x; i < (t.script_len + 3) / 4; i += 64) script[i] = t.script[i];
    for (uint32_t c = threadIdx.x; c < (uint32_t)G::NCONST; c += 64) {
      F v;
#pragma unroll
      for (int i = 0; i < F::N; i++) v.v[i] = t.consts[c * F::N + i];
      A::st(regs, G::CONST_BASE + c, v);
    }
  }
  PCD_DEV F get_state(int slot) const { return A::ld(regs, A::template reg_of_g<G>(A::state_op((uint32_t)slot), bank)); }
  PCD_DEV void set_state(int slot, const F& v) { A::st(regs, A::template reg_of_g<G>(A::state_op((uint32_t)slot), bank), v); }
  PCD_DEV F get_reg(int r) const { return A::ld(regs, (uint32_t)r); }
  PCD_DEV void set_reg(int r, const F& v) { A::st(regs, (uint32_t)r, v); }

  // one program: steps in order, a barrier (one wave: a fence) after each; lanes beyond a step's slot count, and the lanes of
  // continuation slots, idle.  The fetch is software-pipelined: while step s computes, the instruction words of step s + 1 and the
  // header of step s + 2 are already on their way from LDS (the fetch chain header -> words -> operands was three exposed LDS round
  // trips in front of every step).  Everything the loop needs of *this is held in locals: the object lives in private memory, and
  // members read through `this` are reloaded from there after every barrier -- two flat loads per step on the critical path.
  __device__ __noinline__ void run(int pid) {
    const Lds regs_ = regs, code_ = code, steps_ = steps;
    const uint64_t bank_ = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(bank >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)bank);
    const uint32_t first = __builtin_amdgcn_readfirstlane(progs[3 * pid]), cnt = __builtin_amdgcn_readfirstlane(progs[3 * pid + 1]);
    const uint32_t flip = progs[3 * pid + 2];
    const uint32_t lane = threadIdx.x;
    const uint32_t last = first + cnt - 1;   // (headers past the end are read from the last step: harmless, never used)
    auto header = [&](uint32_t s, uint32_t& kw, uint32_t& off, uint32_t& n) {
      s = s < last ? s : last;
      kw = steps_[3 * s]; off = steps_[3 * s + 1]; n = steps_[3 * s + 2];
    };
    auto fetch = [&](uint32_t off, uint32_t n, uint32_t* w) {   // (lanes beyond the step's slots re-read its last slot: in bounds, unused)
      const uint32_t slot = off + (lane < n ? lane : n - 1);
#pragma unroll
      for (int i = 0; i < 12; i++) w[i] = code_[slot * 12 + i];
    };
    uint32_t kw0, off0, n0, kw1, off1, n1;
    header(first, kw0, off0, n0);
    header(first + 1, kw1, off1, n1);
    uint32_t w[12];
    fetch(__builtin_amdgcn_readfirstlane(off0), __builtin_amdgcn_readfirstlane(n0), w);
    for (uint32_t s = first; s <= last; s++) {
      // (the step's words are the same for every lane: taken through readfirstlane so that the dispatch below is scalar branches)
      const uint32_t kw = __builtin_amdgcn_readfirstlane(kw0), off = __builtin_amdgcn_readfirstlane(off0), n = __builtin_amdgcn_readfirstlane(n0);
      const uint32_t off_next = __builtin_amdgcn_readfirstlane(off1), n_next = __builtin_amdgcn_readfirstlane(n1);
      const uint32_t kind = kw & 0xFFu, tmax = kw >> 8;   // tmax: the largest term count among the step's LIN instructions
      uint32_t wn[12];
      fetch(off_next, n_next, wn);               // step s + 1's words
      kw0 = kw1; off0 = off1; n0 = n1;
      header(s + 2, kw1, off1, n1);              // step s + 2's header
      if (lane < n && (w[0] & 0xFFu) != 0xFFu) {
        F o;
        if (kind == 1) o = A::template mul<G>(w, regs_, bank_);
        else if (kind == 2) o = A::template sqr<G>(w, regs_, bank_);
        else {
          uint32_t w2[12];
          const bool wide = tmax > 8u && ((w[0] >> 8) & 0xFFu) > 8;   // (tmax is scalar: steps without a wide instruction skip the reads)
#pragma unroll
          for (int i = 0; i < 12; i++) w2[i] = wide ? code_[(off + lane + 1) * 12 + i] : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
          o = A::template lin_uniform<G>(w, w2, regs_, bank_, tmax);
#else
          o = A::template lin<G>(w, w2, regs_, bank_);   // (host pass of the compiler: never run)
          (void)tmax;
#endif
        }
        A::st(regs_, A::template reg_of_g<G>(w[0] >> 16, bank_), o);   // (every lane has read its operands before any lane stores: lockstep)
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 12; i++) w[i] = wn[i];
    }
    bank ^= (uint64_t)flip;
  }
  // the kernel's script: program ids in order; entries from 0xF0 up select a table entry for the programs that follow, 0xEF inverts
  PCD_DEV void run_script() {
    for (uint32_t i = 0; i < script_len; i++) {
      const uint32_t e = (script[i >> 2] >> (8 * (i & 3))) & 0xFFu;
      if (e >= 0xF0u) bank = (bank & 0xFFFFFFFFull) | ((uint64_t)(e - 0xF0u) << 32);
      else if (e == 0xEFu) {  // SCRIPT_INV: the one field inversion of a final exponentiation, by divsteps on one lane
        if (threadIdx.x == 0) set_reg(G::R_FT1_0, get_reg(G::R_NRM0).inv());
        __syncthreads();
      } else run((int)e);
    }
  }
};
#endif

}  // namespace pcd
