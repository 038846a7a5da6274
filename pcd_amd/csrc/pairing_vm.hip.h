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
  Lds code;                        // this kernel's instruction words, copied into LDS once: an instruction fetch is a ~100-cycle LDS read, not a
                                   // dependent trip to L2 in front of every one of the thousands of steps of a pairing
  uint64_t bank;                   // low half: the bank of every state slot; high half: the table selector

  // LDS words: the register file, then the instruction words
  static constexpr uint32_t REG_WORDS = (uint32_t)G::NREGS * A::STRIDE;
  __host__ __device__ static uint32_t lds_words(const VmTables& t) { return REG_WORDS + t.ncode; }

  PCD_DEV void init(Lds r, const VmTables& t) {
    regs = r; bank = 0;
    code = r + REG_WORDS;
    for (uint32_t i = threadIdx.x; i < t.ncode; i += 64) code[i] = t.code[i];
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

  // Steps first .. first + cnt - 1 of the kernel's flat list (vm_tables.h: the script with its programs expanded), a barrier (one wave:
  // a fence) after each; lanes beyond a step's slot count, and the lanes of continuation slots, idle.  The records are wave-uniform
  // (every lane loads the same 16 bytes from global memory); the fetch is software-pipelined ACROSS program
  // boundaries: while step i computes, the instruction words of step i + 1 are on their way from LDS and the record of step i + 2 from
  // memory (the fetch chain record -> words -> operands was three exposed round trips in front of every step, and every program began
  // with an empty pipeline: 400 of them in a verification).  Everything the loop needs of *this is held in locals: the object lives in
  // private memory, and members read through `this` are reloaded from there after every barrier.
  __device__ __noinline__ void run_flat(const uint32_t* __restrict__ flat, uint32_t first, uint32_t cnt) {
    if (cnt == 0) return;
    const Lds regs_ = regs, code_ = code;
    uint32_t bank_lo = __builtin_amdgcn_readfirstlane((uint32_t)bank), sel = __builtin_amdgcn_readfirstlane((uint32_t)(bank >> 32));
    first = __builtin_amdgcn_readfirstlane(first); cnt = __builtin_amdgcn_readfirstlane(cnt);
    typedef uint32_t U4 __attribute__((ext_vector_type(4)));
    // (a GLOBAL pointer, not a generic one: a flat load counts on the LDS counter as well, and every wait for an operand would wait for it)
    typedef const __attribute__((address_space(1))) U4* RecPtr;
    const RecPtr recs = (RecPtr)(uintptr_t)flat;
    const uint32_t lane = threadIdx.x;
    const uint32_t last = first + cnt - 1;   // (records past the end are read from the last step: harmless, never used)
    auto fetch = [&](uint32_t off, uint32_t n, uint32_t* w) {   // (lanes beyond the step's slots re-read its last slot: in bounds, unused)
      const uint32_t slot = off + (lane < n ? lane : n - 1);
#pragma unroll
      for (int i = 0; i < 12; i++) w[i] = code_[slot * 12 + i];
    };
    U4 r0 = recs[first], r1 = recs[first < last ? first + 1 : last];
    uint32_t w[12];
    fetch(__builtin_amdgcn_readfirstlane(r0.y), __builtin_amdgcn_readfirstlane(r0.z), w);
    for (uint32_t s = first; s <= last; s++) {
      // (every lane loaded the same record: through readfirstlane the dispatch below is scalar branches)
      const uint32_t kw = __builtin_amdgcn_readfirstlane(r0.x), off = __builtin_amdgcn_readfirstlane(r0.y), n = __builtin_amdgcn_readfirstlane(r0.z),
                     flip = __builtin_amdgcn_readfirstlane(r0.w);
      const uint32_t kind = kw & 0xFFu, tmax = (kw >> 8) & 0xFFu, flags = kw >> 16;   // tmax: the largest term count among the step's LIN instructions
      uint32_t wn[12];
      fetch(__builtin_amdgcn_readfirstlane(r1.y), __builtin_amdgcn_readfirstlane(r1.z), wn);   // step s + 1's words
      r0 = r1;
      r1 = recs[s + 2 < last ? s + 2 : last];    // step s + 2's record
      if (flags & 2u) sel = (flags >> 4) & 0xFu;
      if (flags & 4u) {   // SCRIPT_INV: the one field inversion of a final exponentiation, by divsteps on one lane
        if (lane == 0) A::st(regs_, (uint32_t)G::R_FT1_0, A::ld(regs_, (uint32_t)G::R_NRM0).inv());
        __syncthreads();
      }
      const uint64_t bank_ = ((uint64_t)sel << 32) | bank_lo;
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
#endif
        }
        A::st(regs_, A::template reg_of_g<G>(w[0] >> 16, bank_), o);   // (every lane has read its operands before any lane stores: lockstep)
      }
      __syncthreads();
      if (flags & 1u) bank_lo ^= flip;           // a program ends: the slots it wrote change banks
#pragma unroll
      for (int i = 0; i < 12; i++) w[i] = wn[i];
    }
    bank = ((uint64_t)sel << 32) | bank_lo;
  }
};
#endif

}  // namespace pcd
