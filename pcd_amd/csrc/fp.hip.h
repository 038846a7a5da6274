// Device prime-field arithmetic for the four MNT scalar/base fields (gfx950), second design.
//
// Replaces, on device, ark-ff `Fp320` / `Fp768` as used by the prover arithmetic reached from
// /root/reference src/ec_cycle_pcd/mod.rs:171,179 (SNARK::prove).
//
// Representation (device-internal; the C-ABI keeps the upstream image, see from_abi / to_abi):
//   * N unsaturated 28-bit limbs in 32-bit words (N = 11 for the 298-bit fields, 27 for the 753-bit ones),
//   * Montgomery residues with R' = 2^(28 N)  (2^308 / 2^756),
//   * every value normalised (limbs < 2^28) and kept in [0, 2p), not [0, p).
// Why (profiles/r01_k0_int_rates.txt, MI355X): v_mad_u64_u32 issues at half the v_add_u32 rate and so does
// every carry instruction (v_addc_co_u32, v_lshl_add_u64), so a saturated 32-bit CIOS product pays one carry
// op per multiply (2 half-rate ops per limb product, plus the v_mov traffic hipcc adds: 27 % of the mad peak
// in round 1's first kernel).  With 28-bit limbs a whole column of the product sums into ONE 64-bit register
// by plain `acc += a*b` (2 N 2^56 < 2^64): one v_mad_u64_u32 per limb product and nothing else; the carry is a
// shift per column.  R' > 4p makes the Montgomery product closed on [0, 2p) without a final subtraction;
// additions are limb-wise v_add_u32 plus one normalising pass.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Column accumulators as ONE multiply-add chain (round 6).  The product loops are written `acc += x * y; ... acc >>= 28`, but LLVM's Reassociate
// pass rebuilds every column's sum as (p1 + p2 + ... + pn) + carry -- the carried-in accumulator ranks last -- which costs a 64-bit
// v_lshl_add_u64 per column on top of the multiply-adds (180 of the 3 117 instructions of a 298-bit mixed addition; three work-arounds failed in
// round 4: profiles/DESIGN_history_r01-r05.md section 7).  A partial sum with a SECOND use is a leaf for that pass: an always-true
// `__builtin_assume` on every partial sum (no signed overflow / below 2^64 - 1: the column bounds derived above) gives it one, the chain
// stays a chain of v_mad_*64_*32 whose addend is the running sum, and the assumes vanish before instruction selection.  Device code only.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PCD_NO_KEEP_CHAIN)
__device__ __attribute__((weak)) int64_t pcd_never_s = INT64_MIN;   // (weak: the optimiser may not look through the initialiser, so the
__device__ __attribute__((weak)) uint64_t pcd_never_u = UINT64_MAX;  //  comparisons below cannot be proven and folded away early)
#define PCD_KEEP_S(acc) __builtin_assume((acc) != pcd_never_s)
// (the plain products: only where they are CALLS -- the 27-limb fields, one body per kernel: 1 689 -> 1 630 instructions a product.  The inlined
//  11-limb ones sit hundreds of times in the latency kernels, where the assumes cost minutes of compile time: PCD_KEEP_CHAIN_UNSIGNED forces them on)
#if defined(PCD_KEEP_CHAIN_UNSIGNED)
#define PCD_KEEP_U(acc) __builtin_assume((acc) != pcd_never_u)
#else
#define PCD_KEEP_U(acc) do { if constexpr (N > 11) __builtin_assume((acc) != pcd_never_u); } while (0)
#endif
// (single products and squarings: the chain form where the code is inlined -- the 753-bit transform passes, -4 % -- but NOT in the mailbox
//  call bodies of the 753-bit G1 accumulation, which run one wave per SIMD: there LLVM's two interleaved sub-chains per column are the better
//  schedule -- same-box A/B 27.45 -> 28.08 ms with the chain form; the fused two- / three-term products of the lane-split Fq2 / Fq3 take it:
//  22.06 -> 21.04 ms, 8.34 -> 7.09 ms; profiles/r06_ab_keep_chain_753.txt)
#if defined(PCD_KEEP_CHAIN_UNSIGNED)
#define PCD_KEEP_UM(acc) __builtin_assume((acc) != pcd_never_u)
#else
#define PCD_KEEP_UM(acc) do { if constexpr (N > 11 && INLINE_ARITH) __builtin_assume((acc) != pcd_never_u); } while (0)
#endif
#else
#define PCD_KEEP_S(acc) ((void)0)
#define PCD_KEEP_U(acc) ((void)0)
#define PCD_KEEP_UM(acc) ((void)0)
#endif


#include "params28_gen.h"
#include "params_gen.h"

namespace pcd {

#define PCD_DEV __device__ __forceinline__
#define PCD_HD __host__ __device__ __forceinline__

#define PCD_DEF_FIELD(NAME, PFX, PFX28)                                                                       \
  struct NAME {                                                                                               \
    static constexpr int ID = PFX##_ID;                                                                       \
    static constexpr int N = PFX28##_N;    /* 28-bit limbs */                                                 \
    static constexpr int N32 = PFX##_N32;  /* 32-bit words of the ABI image */                                \
    static constexpr int BITS = PFX##_BITS;                                                                   \
    static constexpr int TWO_ADICITY = PFX##_TWO_ADICITY;                                                     \
    static constexpr uint32_t INV = PFX28##_INV;                                                              \
    static constexpr int EST_SHIFT = PFX28##_EST_SHIFT;                                                       \
    static constexpr uint32_t EST_RECIP = PFX28##_EST_RECIP;                                                  \
    static constexpr double RECIP3_D = PFX28##_RECIP3_D;                                                      \
    PCD_HD static uint32_t mod(int i) { constexpr uint32_t m[N] = PFX28##_MOD; return m[i]; }                 \
    PCD_HD static uint32_t mod2(int i) { constexpr uint32_t m[N] = PFX28##_MOD2; return m[i]; }               \
    PCD_HD static uint32_t mod4(int i) { constexpr uint32_t m[N] = PFX28##_MOD4; return m[i]; }               \
    PCD_HD static uint32_t one(int i) { constexpr uint32_t m[N] = PFX28##_ONE; return m[i]; }                 \
    PCD_HD static uint32_t r2(int i) { constexpr uint32_t m[N] = PFX28##_R2; return m[i]; }                   \
    PCD_HD static uint32_t cin(int i) { constexpr uint32_t m[N] = PFX28##_CIN; return m[i]; }                 \
    PCD_HD static uint32_t cout(int i) { constexpr uint32_t m[N] = PFX28##_COUT; return m[i]; }               \
    PCD_HD static uint32_t gen(int i) { constexpr uint32_t m[N] = PFX28##_GEN; return m[i]; }                 \
    PCD_HD static uint32_t root(int i) { constexpr uint32_t m[N] = PFX28##_ROOT; return m[i]; }               \
    PCD_HD static uint32_t modm2(int i) { constexpr uint32_t m[N32] = PFX##_MOD_MINUS_2; return m[i]; }       \
  };
PCD_DEF_FIELD(F298A, PCD_F298A, PCD28_F298A)
PCD_DEF_FIELD(F298B, PCD_F298B, PCD28_F298B)
PCD_DEF_FIELD(F753A, PCD_F753A, PCD28_F753A)
PCD_DEF_FIELD(F753B, PCD_F753B, PCD28_F753B)

// INL: inline the product / square into every caller (throughput kernels of the 298-bit fields) or keep ONE
// non-inlined copy per code object (753-bit fields; and the latency-bound single-lane kernels -- proof assembly,
// pairing -- whose loop bodies must fit the instruction cache).
// MB ("mailbox", device code of 64-lane workgroups only; non-inlined variant): the operands and the result of the product / square
// calls travel through per-lane LDS slots instead of the stack.  hipcc passes a 27-word struct argument and returns one through
// scratch memory -- 54 words stored by the caller and loaded by the callee, 27 back -- two dependent round trips on the vector-memory
// path per product, at one wave per SIMD with nothing to hide them behind (tools/isa_mix.py: 231 of the 301 scratch instructions
// of the 753-bit accumulation loop were this traffic).  Through LDS the same words move with a ~100-cycle round trip and stay off
// the counter the global loads use.  Same memory image as the plain variant.
template <class P, bool INL = (P::N <= 11), bool MB = false>
struct Fp {
  typedef P Params;
  typedef Fp<P, INL, MB> Base;
  static constexpr int N = P::N;
  static constexpr int DEG = 1;
  static constexpr int WORDS = N;           // u32 words per element in device memory
  static constexpr int ABI_WORDS = P::N32;  // u32 words per element at the C-ABI
  static constexpr uint32_t MASK = 0x0FFFFFFFu;
  static constexpr bool INLINE_ARITH = INL;
  static constexpr bool MAILBOX = MB;
  uint32_t v[N];
  PCD_HD static Fp zero() { Fp r; for (int i = 0; i < N; i++) r.v[i] = 0; return r; }
  PCD_HD static Fp one() { Fp r; for (int i = 0; i < N; i++) r.v[i] = P::one(i); return r; }
  PCD_HD static Fp r2() { Fp r; for (int i = 0; i < N; i++) r.v[i] = P::r2(i); return r; }
  PCD_HD static Fp generator() { Fp r; for (int i = 0; i < N; i++) r.v[i] = P::gen(i); return r; }
  PCD_HD static Fp two_adic_root() { Fp r; for (int i = 0; i < N; i++) r.v[i] = P::root(i); return r; }

  PCD_HD bool is_raw_zero() const { uint32_t o = 0; for (int i = 0; i < N; i++) o |= v[i]; return o == 0; }
  // value = 0 mod p  <=>  value in {0, p}
  PCD_HD bool is_zero() const {
    uint32_t o = 0, q = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { o |= v[i]; q |= v[i] ^ P::mod(i); }
    return o == 0 || q == 0;
  }
  PCD_HD bool operator==(const Fp& b) const { return (*this - b).is_zero(); }
  PCD_HD bool operator!=(const Fp& b) const { return !(*this == b); }

  // t: signed limb values of an integer in [0, 2c), c = p (MODP) or 2p; returns it reduced to [0, c): the
  // normalised value and value - c come out of two independent carry chains, the sign of the second selects.
  template <bool MODP>
  PCD_HD static Fp norm_reduce(const int32_t* t) {
    Fp s, d;
    int32_t cs = 0, cd = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      int32_t x = t[i] + cs;
      int32_t y = t[i] - (int32_t)(MODP ? P::mod(i) : P::mod2(i)) + cd;
      if (i < N - 1) {
        s.v[i] = (uint32_t)x & MASK; cs = x >> 28;
        d.v[i] = (uint32_t)y & MASK; cd = y >> 28;
      } else {
        s.v[i] = (uint32_t)x;
        d.v[i] = (uint32_t)y;
      }
    }
    return ((int32_t)d.v[N - 1] < 0) ? s : d;
  }
  // a + b and a - b, back in [0, 2p).  Whether 2p has to come off (go on) is read from the two top limbs of the limb-wise sum
  // (difference): with E = t[N-1] 2^28 + t[N-2] the integer is E B^(N-2) + L, 0 <= L < 2 B^(N-2) for a sum and |L| < B^(N-2) for a
  // difference (B = 2^28), so only E in {P2 - 1, P2} (P2 = the same two limbs of 2p) resp. E = 0 leave the question open -- one pair in
  // 2^55 for random operands, and equal operands of a subtraction -- and take the exact carry chain in a branch that is normally
  // skipped.  One carry chain with the masked constant then normalises: 5 instructions per limb against 9 for the two parallel chains
  // of norm_reduce (tools/isa_mix.py: 27-limb additions were 243 instructions each, 16 % of the 753-bit accumulation loop).
  // Used for the 27-limb fields only.  Same-box A/B on MI355X (two chains -> estimate): G1-753 accumulation 12.4 -> 11.3 ms (2^18),
  // split Fq2-753 / Fq3-753 MSMs 15.1 -> 14.5 / 30.9 -> 29.2 ms (2^15); for the inlined 11-limb code the skipped branches cost more
  // than the shorter chains save (G1-298 unchanged, split Fq3-298 accumulation 3.90 -> 4.16 ms at 2^18), so it keeps norm_reduce.
  static constexpr bool ESTIMATE_ADDSUB = N > 11;
  PCD_HD static uint64_t mod2_top2() { return ((uint64_t)P::mod2(N - 1) << 28) + P::mod2(N - 2); }
  PCD_HD Fp operator+(const Fp& b) const {
    if constexpr (!ESTIMATE_ADDSUB) {
      int32_t s[N];
#pragma unroll
      for (int i = 0; i < N; i++) s[i] = (int32_t)(v[i] + b.v[i]);
      return norm_reduce<false>(s);
    }
    uint32_t t[N];
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = v[i] + b.v[i];
    const uint64_t E = ((uint64_t)t[N - 1] << 28) + t[N - 2];
    bool off = E > mod2_top2();
    if (__builtin_expect(E + 1 - mod2_top2() < 2, 0)) {  // E in {P2 - 1, P2}: the sign of (a + b) - 2p, exactly
      int32_t x = 0;
#pragma unroll
      for (int i = 0; i < N; i++) x = (int32_t)t[i] - (int32_t)P::mod2(i) + (x >> 28);
      off = x >= 0;
    }
    const uint32_t m = off ? 0xFFFFFFFFu : 0u;
    Fp r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
      const int32_t x = (int32_t)t[i] + (int32_t)((0u - P::mod2(i)) & m) + c;
      r.v[i] = (uint32_t)x & MASK; c = x >> 28;
    }
    r.v[N - 1] = t[N - 1] + ((0u - P::mod2(N - 1)) & m) + (uint32_t)c;
    return r;
  }
  PCD_HD Fp operator-(const Fp& b) const {
    if constexpr (!ESTIMATE_ADDSUB) {  // a - b + 2p in (0, 4p)
      int32_t s[N];
#pragma unroll
      for (int i = 0; i < N; i++) s[i] = (int32_t)v[i] - (int32_t)b.v[i] + (int32_t)P::mod2(i);
      return norm_reduce<false>(s);
    }
    int32_t t[N];
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = (int32_t)v[i] - (int32_t)b.v[i];
    const int64_t E = (int64_t)t[N - 1] * ((int64_t)1 << 28) + t[N - 2];
    bool on = E < 0;
    if (__builtin_expect(E == 0, 0)) {  // the sign of a - b is the carry out of the limbs below
      int32_t c = 0;
#pragma unroll
      for (int i = 0; i < N - 2; i++) c = (t[i] + c) >> 28;
      on = c < 0;
    }
    const uint32_t m = on ? 0xFFFFFFFFu : 0u;
    Fp r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
      const int32_t x = t[i] + (int32_t)(P::mod2(i) & m) + c;
      r.v[i] = (uint32_t)x & MASK; c = x >> 28;
    }
    r.v[N - 1] = (uint32_t)(t[N - 1] + (int32_t)(P::mod2(N - 1) & m) + c);
    return r;
  }
  PCD_HD Fp neg() const { return zero() - *this; }
  PCD_HD Fp dbl() const { return *this + *this; }
  // the representative in [0, p)
  PCD_HD Fp canonical() const {
    int32_t t[N];
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = (int32_t)v[i];
    return norm_reduce<true>(t);
  }

  // Montgomery product (product scanning, reduction interleaved): inputs and output in [0, 2p).
  // NOT inlined: one fully unrolled copy per field per code object (2 N^2 v_mad_u64_u32 + ~6 N others).
  // (operands and result by value: through references the 753-bit G1 accumulation is 20 % SLOWER -- 55.6 against 45.7 ms at 2^20,
  //  same-box A/B -- the caller then keeps every operand addressable in its frame (2 064 B against 1 408 B of scratch per lane))
  __host__ __device__ __noinline__ static Fp mul_call(Fp a, Fp b) { return mul_impl(a, b); }
  PCD_HD static Fp mul(const Fp& a, const Fp& b) {
    if constexpr (INLINE_ARITH) return mul_impl(a, b);
    else {
#if defined(__HIP_DEVICE_COMPILE__)
      if constexpr (MB) return mb_mul(a, b);
#endif
      return mul_call(a, b);
    }
  }
  // ---- LDS mailbox of the MB variant: 2 slots per lane, an element as MB_CH 16-byte pieces, lane-linear ([slot][piece][lane])
  typedef uint32_t MbVec __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) MbVec* MbPtr;
  static constexpr int MB_CH = (N + 3) / 4, MB_LANES = 64;
  PCD_DEV static MbPtr mb_base() { __shared__ MbVec box[2 * MB_CH * MB_LANES]; return (MbPtr)box; }
  PCD_DEV static void mb_put(MbPtr mb, int slot, const Fp& a) {
    const unsigned l = threadIdx.x;
#pragma unroll
    for (int k = 0; k < MB_CH; k++) {
      MbVec w;
      w.x = a.v[4 * k];
      w.y = 4 * k + 1 < N ? a.v[4 * k + 1] : 0u;
      w.z = 4 * k + 2 < N ? a.v[4 * k + 2] : 0u;
      w.w = 4 * k + 3 < N ? a.v[4 * k + 3] : 0u;
      mb[(slot * MB_CH + k) * MB_LANES + l] = w;
    }
  }
  PCD_DEV static Fp mb_get(MbPtr mb, int slot) { return mb_get_lane(mb, slot, threadIdx.x); }
  PCD_DEV static Fp mb_get_lane(MbPtr mb, int slot, unsigned l) {  // the slot of another lane of the wave (lane-split extension fields)
    Fp r;
#pragma unroll
    for (int k = 0; k < MB_CH; k++) {
      const MbVec w = mb[(slot * MB_CH + k) * MB_LANES + l];
      r.v[4 * k] = w.x;
      if (4 * k + 1 < N) r.v[4 * k + 1] = w.y;
      if (4 * k + 2 < N) r.v[4 * k + 2] = w.z;
      if (4 * k + 3 < N) r.v[4 * k + 3] = w.w;
    }
    return r;
  }
  // Ordering of the cross-lane exchange (lane-split extension fields: a lane reads slots its PARTNERS wrote, and later overwrites a slot
  // its partners have read).  In HIP's per-thread memory model those are accesses of different threads to the same LDS words, so
  // without synchronisation the compiler is free to move them past each other; lockstep execution of a wave makes the hardware order
  // equal to the program order of the ds_ instructions, so all that is needed is that the COMPILER keeps that order: a release /
  // acquire fence pair at wavefront scope around a wave barrier (a scheduling barrier; no instruction is emitted for either -- LDS
  // operations of one wave complete in issue order).  PCD_MB_FENCE=0 builds the round-3 code without it (tools/microbench A/B).
#ifndef PCD_MB_FENCE
#define PCD_MB_FENCE 1
#endif
  PCD_DEV static void mb_sync() {
#if PCD_MB_FENCE && defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
  }
  __device__ __noinline__ static void mb_mul_call(MbPtr mb) { mb_put(mb, 0, mul_impl(mb_get(mb, 0), mb_get(mb, 1))); }
  __device__ __noinline__ static void mb_sqr_call(MbPtr mb) { mb_put(mb, 0, sqr_impl(mb_get(mb, 0))); }
  PCD_DEV static Fp mb_mul(const Fp& a, const Fp& b) {
    const MbPtr mb = mb_base();
    mb_put(mb, 0, a); mb_put(mb, 1, b);
    mb_mul_call(mb);
    return mb_get(mb, 0);
  }
  PCD_DEV static Fp mb_sqr(const Fp& a) {
    const MbPtr mb = mb_base();
    mb_put(mb, 0, a);
    mb_sqr_call(mb);
    return mb_get(mb, 0);
  }
  PCD_HD static Fp mul_impl(const Fp& a, const Fp& b) {
    uint32_t m[N];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) { acc += (uint64_t)a.v[i] * b.v[k - i]; PCD_KEEP_UM(acc); }
#pragma unroll
      for (int i = 0; i < k; i++) { acc += (uint64_t)m[i] * P::mod(k - i); PCD_KEEP_UM(acc); }
      m[k] = ((uint32_t)acc * P::INV) & MASK;
      acc += (uint64_t)m[k] * P::mod(0); PCD_KEEP_UM(acc);
      acc >>= 28;
    }
    Fp r;
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
      for (int i = k - N + 1; i < N; i++) { acc += (uint64_t)a.v[i] * b.v[k - i]; PCD_KEEP_UM(acc); }
#pragma unroll
      for (int i = k - N + 1; i < N; i++) { acc += (uint64_t)m[i] * P::mod(k - i); PCD_KEEP_UM(acc); }
      r.v[k - N] = (uint32_t)acc & MASK;
      acc >>= 28;
    }
    r.v[N - 1] = (uint32_t)acc;
    return r;
  }
  // Montgomery square: the a_i a_j (i < j) products are taken once with a doubled operand (limbs < 2^29, column
  // sums still < 2^62): N(N+1)/2 + N^2 mads instead of 2 N^2.
  __host__ __device__ __noinline__ static Fp sqr_call(Fp a) { return sqr_impl(a); }
  PCD_HD static Fp sqr_(const Fp& a) {
    if constexpr (INLINE_ARITH) return sqr_impl(a);
    else {
#if defined(__HIP_DEVICE_COMPILE__)
      if constexpr (MB) return mb_sqr(a);
#endif
      return sqr_call(a);
    }
  }
  PCD_HD static Fp sqr_impl(const Fp& a) {
    uint32_t m[N], a2[N];
#pragma unroll
    for (int i = 0; i < N; i++) a2[i] = a.v[i] << 1;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
      for (int i = 0; 2 * i < k; i++) { acc += (uint64_t)a2[i] * a.v[k - i]; PCD_KEEP_UM(acc); }
      if ((k & 1) == 0) { acc += (uint64_t)a.v[k / 2] * a.v[k / 2]; PCD_KEEP_UM(acc); }
#pragma unroll
      for (int i = 0; i < k; i++) { acc += (uint64_t)m[i] * P::mod(k - i); PCD_KEEP_UM(acc); }
      m[k] = ((uint32_t)acc * P::INV) & MASK;
      acc += (uint64_t)m[k] * P::mod(0); PCD_KEEP_UM(acc);
      acc >>= 28;
    }
    Fp r;
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
      for (int i = k - N + 1; 2 * i < k; i++) { acc += (uint64_t)a2[i] * a.v[k - i]; PCD_KEEP_UM(acc); }
      if ((k & 1) == 0) { acc += (uint64_t)a.v[k / 2] * a.v[k / 2]; PCD_KEEP_UM(acc); }
#pragma unroll
      for (int i = k - N + 1; i < N; i++) { acc += (uint64_t)m[i] * P::mod(k - i); PCD_KEEP_UM(acc); }
      r.v[k - N] = (uint32_t)acc & MASK;
      acc >>= 28;
    }
    r.v[N - 1] = (uint32_t)acc;
    return r;
  }
  PCD_HD Fp operator*(const Fp& b) const { return mul(*this, b); }
  PCD_HD Fp sqr() const { return sqr_(*this); }

  // ---- lazily reduced arithmetic (298-bit fields only: R'/p > 2^10 leaves room) -----------------------------------------------
  // Lz = an integer >= 0 held in SIGNED 28-bit-radix limbs that are not carry-propagated: additions and subtractions are
  // limb-wise, without any carry chain or reduction mod p.  Only the products reduce: for values a < ca p, b < cb p with
  // ca cb <= 1024 the Montgomery product (a b + m p) / R' is < 2p with normalised limbs.  The callers (ec.hip.h madd_lz) keep
  // track of the value bounds (the multiple of p) and of the limb magnitudes (|a_i| |b_j| summed over a column < 2^63).
  struct Lz { int32_t v[N]; };
  // host-side harness only (tests/hostcheck, -DPCD_LZ_CHECK): every column sum is recomputed in 128 bits and compared
#if defined(PCD_LZ_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
#define PCD_LZ_WIDE_DECL __int128 wide_ = 0;
#define PCD_LZ_WIDE_MAC(x, y) wide_ += (__int128)(x) * (__int128)(y);
#define PCD_LZ_WIDE_CHECK() do { if (wide_ != (__int128)acc) abort(); wide_ >>= 28; } while (0)
#else
#define PCD_LZ_WIDE_DECL
#define PCD_LZ_WIDE_MAC(x, y)
#define PCD_LZ_WIDE_CHECK() do { } while (0)
#endif
  PCD_HD Lz lz() const { Lz r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = (int32_t)v[i];
    return r; }
  PCD_HD static Lz lz_add(const Lz& a, const Lz& b) { Lz r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = a.v[i] + b.v[i];
    return r; }
  // a - b + (4 << S) p: non-negative as long as b < (4 << S) p
  template <int S>
  PCD_HD static Lz lz_sub(const Lz& a, const Lz& b) { Lz r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = a.v[i] - b.v[i] + (int32_t)(P::mod4(i) << S);
    return r; }
  PCD_HD static Lz lz_shl(const Lz& a, int s) { Lz r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = a.v[i] * (1 << s);
    return r; }
  // one carry chain: limbs 0 .. N-2 into [0, 2^28), the top limb takes the rest (value unchanged, still unreduced)
  PCD_HD static Lz lz_carry(const Lz& a) { Lz r; int32_t c = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++) { int32_t x = a.v[i] + c; r.v[i] = x & (int32_t)MASK; c = x >> 28; }
    r.v[N - 1] = a.v[N - 1] + c;
    return r; }
  // (a0 b0 [+ a1 b1]) / R' in [0, 2p), limbs normalised
  template <int TERMS>
  PCD_HD static Fp lz_dot(const Lz& a0, const Lz& b0, const Lz& a1, const Lz& b1) {
    int32_t m[N];
    int64_t acc = 0;
    PCD_LZ_WIDE_DECL
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) {
        acc += (int64_t)a0.v[i] * b0.v[k - i]; PCD_KEEP_S(acc);
        PCD_LZ_WIDE_MAC(a0.v[i], b0.v[k - i])
        if (TERMS == 2) { acc += (int64_t)a1.v[i] * b1.v[k - i]; PCD_KEEP_S(acc); PCD_LZ_WIDE_MAC(a1.v[i], b1.v[k - i]) }
      }
#pragma unroll
      for (int i = 0; i < k; i++) { acc += (int64_t)m[i] * (int32_t)P::mod(k - i); PCD_KEEP_S(acc); PCD_LZ_WIDE_MAC(m[i], P::mod(k - i)) }
      m[k] = (int32_t)(((uint32_t)acc * P::INV) & MASK);
      acc += (int64_t)m[k] * (int32_t)P::mod(0); PCD_KEEP_S(acc);
      PCD_LZ_WIDE_MAC(m[k], P::mod(0))
      PCD_LZ_WIDE_CHECK();
      acc >>= 28;
    }
    Fp r;
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
      for (int i = k - N + 1; i < N; i++) {
        acc += (int64_t)a0.v[i] * b0.v[k - i]; PCD_KEEP_S(acc);
        PCD_LZ_WIDE_MAC(a0.v[i], b0.v[k - i])
        if (TERMS == 2) { acc += (int64_t)a1.v[i] * b1.v[k - i]; PCD_KEEP_S(acc); PCD_LZ_WIDE_MAC(a1.v[i], b1.v[k - i]) }
      }
#pragma unroll
      for (int i = k - N + 1; i < N; i++) { acc += (int64_t)m[i] * (int32_t)P::mod(k - i); PCD_KEEP_S(acc); PCD_LZ_WIDE_MAC(m[i], P::mod(k - i)) }
      r.v[k - N] = (uint32_t)acc & MASK;
      PCD_LZ_WIDE_CHECK();
      acc >>= 28;
    }
    if (acc < 0 || acc >= ((int64_t)1 << 28)) {
#if defined(PCD_LZ_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
      abort();
#endif
    }
    r.v[N - 1] = (uint32_t)acc;
    return r;
  }
  // k a with limbs carried back to 28 bits (the top limb takes the rest): what a product needs of "non-residue times coefficient" --
  // k = 17 on limbs of up to 29 bits does not fit 32 bits limb-wise
  PCD_HD static Lz lz_scale_carry(const Lz& a, int32_t k) { Lz r; int64_t c = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++) { const int64_t x = (int64_t)a.v[i] * k + c; r.v[i] = (int32_t)((uint32_t)x & MASK); c = x >> 28; }
    r.v[N - 1] = (int32_t)((int64_t)a.v[N - 1] * k + c);
    return r; }
  // (a0 b0 + a1 b1 + a2 b2 + a3 b3) / R' in [0, 2p): the four-term form of lz_dot (one reduction for a sum of two Fq2 products'
  // coefficient); the sum of the four value-bound products must stay <= 1024 p^2, limb products per column < 2^63 as there
  PCD_HD static Fp lz_dot4(const Lz& a0, const Lz& b0, const Lz& a1, const Lz& b1, const Lz& a2, const Lz& b2, const Lz& a3, const Lz& b3) {
    int32_t m[N];
    int64_t acc = 0;
    PCD_LZ_WIDE_DECL
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) {
        acc += (int64_t)a0.v[i] * b0.v[k - i]; PCD_KEEP_S(acc); PCD_LZ_WIDE_MAC(a0.v[i], b0.v[k - i])
        acc += (int64_t)a1.v[i] * b1.v[k - i]; PCD_KEEP_S(acc); PCD_LZ_WIDE_MAC(a1.v[i], b1.v[k - i])
        acc += (int64_t)a2.v[i] * b2.v[k - i]; PCD_KEEP_S(acc); PCD_LZ_WIDE_MAC(a2.v[i], b2.v[k - i])
        acc += (int64_t)a3.v[i] * b3.v[k - i]; PCD_KEEP_S(acc); PCD_LZ_WIDE_MAC(a3.v[i], b3.v[k - i])
      }
#pragma unroll
      for (int i = 0; i < k; i++) { acc += (int64_t)m[i] * (int32_t)P::mod(k - i); PCD_KEEP_S(acc); PCD_LZ_WIDE_MAC(m[i], P::mod(k - i)) }
      m[k] = (int32_t)(((uint32_t)acc * P::INV) & MASK);
      acc += (int64_t)m[k] * (int32_t)P::mod(0); PCD_KEEP_S(acc);
      PCD_LZ_WIDE_MAC(m[k], P::mod(0))
      PCD_LZ_WIDE_CHECK();
      acc >>= 28;
    }
    Fp r;
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
      for (int i = k - N + 1; i < N; i++) {
        acc += (int64_t)a0.v[i] * b0.v[k - i]; PCD_KEEP_S(acc); PCD_LZ_WIDE_MAC(a0.v[i], b0.v[k - i])
        acc += (int64_t)a1.v[i] * b1.v[k - i]; PCD_KEEP_S(acc); PCD_LZ_WIDE_MAC(a1.v[i], b1.v[k - i])
        acc += (int64_t)a2.v[i] * b2.v[k - i]; PCD_KEEP_S(acc); PCD_LZ_WIDE_MAC(a2.v[i], b2.v[k - i])
        acc += (int64_t)a3.v[i] * b3.v[k - i]; PCD_KEEP_S(acc); PCD_LZ_WIDE_MAC(a3.v[i], b3.v[k - i])
      }
#pragma unroll
      for (int i = k - N + 1; i < N; i++) { acc += (int64_t)m[i] * (int32_t)P::mod(k - i); PCD_KEEP_S(acc); PCD_LZ_WIDE_MAC(m[i], P::mod(k - i)) }
      r.v[k - N] = (uint32_t)acc & MASK;
      PCD_LZ_WIDE_CHECK();
      acc >>= 28;
    }
    if (acc < 0 || acc >= ((int64_t)1 << 28)) {
#if defined(PCD_LZ_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
      abort();
#endif
    }
    r.v[N - 1] = (uint32_t)acc;
    return r;
  }
  PCD_HD static Fp lz_mul(const Lz& a, const Lz& b) { return lz_dot<1>(a, b, a, b); }
  PCD_HD static Fp lz_dot2(const Lz& a0, const Lz& b0, const Lz& a1, const Lz& b1) { return lz_dot<2>(a0, b0, a1, b1); }
  PCD_HD static Fp lz_sqr(const Lz& a) {
    int32_t m[N], a2[N];
#pragma unroll
    for (int i = 0; i < N; i++) a2[i] = a.v[i] * 2;
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
      for (int i = 0; 2 * i < k; i++) { acc += (int64_t)a2[i] * a.v[k - i]; PCD_KEEP_S(acc); }
      if ((k & 1) == 0) { acc += (int64_t)a.v[k / 2] * a.v[k / 2]; PCD_KEEP_S(acc); }
#pragma unroll
      for (int i = 0; i < k; i++) { acc += (int64_t)m[i] * (int32_t)P::mod(k - i); PCD_KEEP_S(acc); }
      m[k] = (int32_t)(((uint32_t)acc * P::INV) & MASK);
      acc += (int64_t)m[k] * (int32_t)P::mod(0); PCD_KEEP_S(acc);
      acc >>= 28;
    }
    Fp r;
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
      for (int i = k - N + 1; 2 * i < k; i++) { acc += (int64_t)a2[i] * a.v[k - i]; PCD_KEEP_S(acc); }
      if ((k & 1) == 0) { acc += (int64_t)a.v[k / 2] * a.v[k / 2]; PCD_KEEP_S(acc); }
#pragma unroll
      for (int i = k - N + 1; i < N; i++) { acc += (int64_t)m[i] * (int32_t)P::mod(k - i); PCD_KEEP_S(acc); }
      r.v[k - N] = (uint32_t)acc & MASK;
      acc >>= 28;
    }
    r.v[N - 1] = (uint32_t)acc;
    return r;
  }

  // Fused sums of products (a0 b0 + a1 b1 [+ a2 b2]) / R' with ONE Montgomery reduction: the building block of the
  // Fq2 / Fq3 products (schoolbook with lazy reduction: no Karatsuba additions, one reduction per output coefficient).
  // Column sums stay below (TERMS + 1) N 2^56 < 2^63.  TERMS = 3 can reach 2.5p for the 753-bit fields (R'/p < 16),
  // so it ends with one normalising pass.
  // (operands by reference: the callers hold them in memory anyway, and large by-value argument lists
  // miscompiled for the 753-bit fields on gfx950 / ROCm 7.2)
  template <int TERMS>
  __host__ __device__ __noinline__ static void dot_call(Fp& out, const Fp& a0, const Fp& b0, const Fp& a1, const Fp& b1, const Fp& a2,
                                                        const Fp& b2) {
    dot_impl<TERMS>(out, a0, b0, a1, b1, a2, b2);
  }
  template <int TERMS>
  PCD_HD static void dot(Fp& out, const Fp& a0, const Fp& b0, const Fp& a1, const Fp& b1, const Fp& a2, const Fp& b2) {
    if constexpr (INLINE_ARITH) dot_impl<TERMS>(out, a0, b0, a1, b1, a2, b2); else dot_call<TERMS>(out, a0, b0, a1, b1, a2, b2);
  }
  template <int TERMS>
  PCD_HD static void dot_impl(Fp& out, const Fp& a0, const Fp& b0, const Fp& a1, const Fp& b1, const Fp& a2, const Fp& b2) {
    uint32_t m[N];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) {
        acc += (uint64_t)a0.v[i] * b0.v[k - i]; PCD_KEEP_U(acc);
        acc += (uint64_t)a1.v[i] * b1.v[k - i]; PCD_KEEP_U(acc);
        if (TERMS == 3) { acc += (uint64_t)a2.v[i] * b2.v[k - i]; PCD_KEEP_U(acc); }
      }
#pragma unroll
      for (int i = 0; i < k; i++) { acc += (uint64_t)m[i] * P::mod(k - i); PCD_KEEP_U(acc); }
      m[k] = ((uint32_t)acc * P::INV) & MASK;
      acc += (uint64_t)m[k] * P::mod(0); PCD_KEEP_U(acc);
      acc >>= 28;
    }
    int32_t r[N];
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
      for (int i = k - N + 1; i < N; i++) {
        acc += (uint64_t)a0.v[i] * b0.v[k - i]; PCD_KEEP_U(acc);
        acc += (uint64_t)a1.v[i] * b1.v[k - i]; PCD_KEEP_U(acc);
        if (TERMS == 3) { acc += (uint64_t)a2.v[i] * b2.v[k - i]; PCD_KEEP_U(acc); }
      }
#pragma unroll
      for (int i = k - N + 1; i < N; i++) { acc += (uint64_t)m[i] * P::mod(k - i); PCD_KEEP_U(acc); }
      r[k - N] = (int32_t)((uint32_t)acc & MASK);
      acc >>= 28;
    }
    r[N - 1] = (int32_t)(uint32_t)acc;
    if (TERMS == 3) { out = norm_reduce<false>(r); return; }  // value < 4p -> [0, 2p)
#pragma unroll
    for (int i = 0; i < N; i++) out.v[i] = (uint32_t)r[i];  // TERMS == 2: < 2p already (8 p^2 / R' + p)
  }
  PCD_HD static Fp dot2(const Fp& a0, const Fp& b0, const Fp& a1, const Fp& b1) { Fp o; dot<2>(o, a0, b0, a1, b1, a0, a0); return o; }
  PCD_HD static Fp dot3(const Fp& a0, const Fp& b0, const Fp& a1, const Fp& b1, const Fp& a2, const Fp& b2) {
    Fp o;
    dot<3>(o, a0, b0, a1, b1, a2, b2);
    return o;
  }

  // multiplication by a small non-negative integer (curve / tower constants, k < 2^8)
  PCD_HD Fp mul_small(unsigned k) const {
    if (k == 0) return zero();
    if (k == 1) return *this;
    if (k == 2) return dbl();
    if (k == 3) return dbl() + *this;
    if (k == 4) return dbl().dbl();
    return mul_small_var(k);
  }
  // the same without the shortcuts for tiny k: branch-free, so k may differ from lane to lane (lane-split extension fields)
  PCD_HD Fp mul_small_var(unsigned k) const {
    uint32_t t[N];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++) { c += (uint64_t)v[i] * k; t[i] = (uint32_t)c & MASK; c >>= 28; }
    c += (uint64_t)v[N - 1] * k;  // top limb, kept whole
    // q <= floor(value / p), and value - q p < 2p + tiny
    uint64_t top2 = (c << 28) | t[N - 2];
    uint32_t est = (uint32_t)(top2 >> P::EST_SHIFT);
    uint32_t q = (uint32_t)(((uint64_t)est * P::EST_RECIP) >> 32);
    int32_t r[N];
    int64_t cc = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
      cc += (int64_t)t[i] - (int64_t)((uint64_t)q * P::mod(i));
      r[i] = (int32_t)((uint32_t)cc & MASK);
      cc >>= 28;
    }
    cc += (int64_t)c - (int64_t)((uint64_t)q * P::mod(N - 1));
    r[N - 1] = (int32_t)cc;  // < 4p / 2^(28(N-1)): small
    return norm_reduce<false>(r);
  }
  // The value of a signed limb-wise sum of small multiples of field elements: s[i] = sum_t c_t a_t[i] with every a_t in [0, 2p) and
  // sum |c_t| <= 2000, so V = sum s[i] 2^(28 i) has |V| < 2^12 p and every |s[i]| < 2^40.  ONE pass: the quotient is estimated BEFORE
  // any carry runs, in double precision from the three top columns -- d = s[N-1] 2^56 + s[N-2] 2^28 + s[N-3] is V / 2^(28 (N - 3)) up to
  // ~2^12 from the columns below (|s[N-4]| < 2^40 weighs 2^-28) and 2^43 from rounding, against p / 2^(28 (N - 3)) > 2^72: x = d * RECIP3_D is V / p to within 2^-28 --
  // and q = floor(x - 1/2) puts V - q p in (0.49 p, 1.51 p): inside [0, 2p) with no correction step, for negative V as well (no offset
  // K p).  Then one signed carry chain over s[i] - q p_i.  (Before: a chain with K p added, the estimate, a second chain, and the two
  // chains of norm_reduce: 134 instructions at 11 limbs against 75 now.)  The LIN instruction of the pairing VM, the small-coefficient
  // entries of the mat-vec and the Fq2 accumulator use it.
  PCD_HD static Fp from_signed_sum(const int64_t* s) {
    const double d = ((double)s[N - 1] * 268435456.0 + (double)s[N - 2]) * 268435456.0 + (double)s[N - 3];
    const int32_t q = (int32_t)__builtin_floor(d * P::RECIP3_D - 0.5);
    Fp o;
    int64_t cc = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
      cc += s[i] - (int64_t)q * (int64_t)(int32_t)P::mod(i);   // (int32 factors: one v_mad_i64_i32)
      o.v[i] = (uint32_t)cc & MASK;
      cc >>= 28;
    }
    cc += s[N - 1] - (int64_t)q * (int64_t)(int32_t)P::mod(N - 1);
    o.v[N - 1] = (uint32_t)cc;   // the top limb of a value in (0, 2p): non-negative and small
    return o;
  }
  // the inverse (zero maps to zero): divsteps -- 25x (753 bits) / 7x (298 bits) fewer instructions than a^(p-2), which matters wherever
  // ONE lane inverts on the critical path: the affine images of a proof's three points, batch normalisations, the pairing's inverses
  PCD_HD Fp inv() const { return inv_gcd(); }
  // a^(p-2): the cross-check of inv_gcd (tests/hostcheck)
  PCD_HD Fp inv_fermat() const {
    Fp r = one();
    bool started = false;
    for (int i = P::N32 * 32 - 1; i >= 0; i--) {
      if (started) r = r.sqr();
      if ((P::modm2(i >> 5) >> (i & 31)) & 1) { r = started ? r * *this : *this; started = true; }
    }
    return r;
  }
  // The inverse by divsteps (Bernstein-Yang, "Fast constant-time gcd computation and modular inversion", 2019): 28 division steps at a
  // time on the low limbs of (f, g) = (p, a) give a 2x2 transition matrix with entries of at most 28 bits, which is then applied to
  // the full-width f, g (exactly: the low 28 bits cancel) and to d, e with d a = f, e a = g (mod p) (a multiple of p makes those
  // divisible by 2^28 as well).  After (49 bits + 57) / 17 steps g = 0 and f = +-1, so a^-1 = +-d.  No branch depends on the data:
  // every lane of a wave walks the same ~1 000 instructions per batch -- 45 field products' worth for 753 bits, against 940 for
  // a^(p-2) -- which is what makes one inversion per lane and per few dozen additions affordable (msm.hip.h, the pair tree).
  // Montgomery form in and out: the integer inverse of a R is a^-1 R^-1, one product with R^3 puts R back.  Zero maps to zero.
  PCD_HD Fp inv_gcd() const {
    constexpr int STEPS = (49 * P::BITS + 57) / 17, BATCHES = (STEPS + 27) / 28;
    constexpr uint32_t PINV = (0u - P::INV) & MASK;  // p^-1 mod 2^28
    const Fp xc = canonical();
    int32_t f[N], g[N], d[N], e[N];
#pragma unroll
    for (int i = 0; i < N; i++) { f[i] = (int32_t)P::mod(i); g[i] = (int32_t)xc.v[i]; d[i] = 0; e[i] = 0; }
    e[0] = 1;
    int32_t eta = -1;  // -delta
    for (int b = 0; b < BATCHES; b++) {
      uint32_t u = 1, v = 0, q = 0, r = 1, ff = (uint32_t)f[0], gg = (uint32_t)g[0];
#pragma unroll 4
      for (int i = 0; i < 28; i++) {
        uint32_t c1 = (uint32_t)(eta >> 31);
        const uint32_t c2 = 0u - (gg & 1u);
        const uint32_t x = (ff ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
        gg += x & c2; q += y & c2; r += z & c2;
        c1 &= c2;
        eta = (eta ^ (int32_t)c1) - (int32_t)(c1 + 1u);
        ff += gg & c1; u += q & c1; v += r & c1;
        gg >>= 1; u <<= 1; v <<= 1;
      }
      const int64_t U = (int32_t)u, V = (int32_t)v, Q = (int32_t)q, R = (int32_t)r;
      {  // (f, g) <- t (f, g) / 2^28
        int64_t cf = U * f[0] + V * g[0], cg = Q * f[0] + R * g[0];
        cf >>= 28; cg >>= 28;
#pragma unroll
        for (int i = 1; i < N; i++) {
          cf += U * f[i] + V * g[i];
          cg += Q * f[i] + R * g[i];
          f[i - 1] = (int32_t)((uint32_t)cf & MASK); cf >>= 28;
          g[i - 1] = (int32_t)((uint32_t)cg & MASK); cg >>= 28;
        }
        f[N - 1] = (int32_t)cf; g[N - 1] = (int32_t)cg;
      }
      {  // (d, e) <- t (d, e) / 2^28 mod p, kept in (-2p, p)
        const int32_t sd = d[N - 1] >> 31, se = e[N - 1] >> 31;
        int32_t md = ((int32_t)u & sd) + ((int32_t)v & se), me = ((int32_t)q & sd) + ((int32_t)r & se);
        int64_t cd = U * d[0] + V * e[0], ce = Q * d[0] + R * e[0];
        md -= (int32_t)((PINV * (uint32_t)cd + (uint32_t)md) & MASK);
        me -= (int32_t)((PINV * (uint32_t)ce + (uint32_t)me) & MASK);
        cd += (int64_t)P::mod(0) * md; ce += (int64_t)P::mod(0) * me;
        cd >>= 28; ce >>= 28;
#pragma unroll
        for (int i = 1; i < N; i++) {
          cd += U * d[i] + V * e[i] + (int64_t)P::mod(i) * md;
          ce += Q * d[i] + R * e[i] + (int64_t)P::mod(i) * me;
          d[i - 1] = (int32_t)((uint32_t)cd & MASK); cd >>= 28;
          e[i - 1] = (int32_t)((uint32_t)ce & MASK); ce >>= 28;
        }
        d[N - 1] = (int32_t)cd; e[N - 1] = (int32_t)ce;
      }
    }
    // +-d with the sign of f, from (-2p, 2p) into [0, 2p)
    const int32_t sf = f[N - 1] >> 31;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++) { const int32_t x = ((d[i] ^ sf) - sf) + c; d[i] = (int32_t)((uint32_t)x & MASK); c = x >> 28; }
    d[N - 1] = ((d[N - 1] ^ sf) - sf) + c;
    const uint32_t neg = (uint32_t)(d[N - 1] >> 31);
    Fp o;
    c = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++) { const int32_t x = d[i] + (int32_t)(P::mod2(i) & neg) + c; o.v[i] = (uint32_t)x & MASK; c = x >> 28; }
    o.v[N - 1] = (uint32_t)(d[N - 1] + (int32_t)(P::mod2(N - 1) & neg) + c);
    return o * (r2() * r2());
  }
  PCD_HD Fp pow_u64(uint64_t e) const {
    Fp r = one(), b = *this;
    while (e) { if (e & 1) r = r * b; e >>= 1; if (e) b = b.sqr(); }
    return r;
  }
  PCD_HD static Fp from_u64(uint64_t x) {  // small integer -> internal Montgomery form
    Fp r = zero();
    r.v[0] = (uint32_t)x & MASK;
    r.v[1] = (uint32_t)(x >> 28) & MASK;
    r.v[2] = (uint32_t)(x >> 56);
    return r * r2();
  }

  // ---- device memory (internal image)
  PCD_HD static Fp load(const uint32_t* p) {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = p[i];
    return r;
  }
  PCD_HD void store(uint32_t* p) const {
#pragma unroll
    for (int i = 0; i < N; i++) p[i] = v[i];
  }

  // ---- C-ABI image: N32 little-endian 32-bit words (== L u64 limbs)
  PCD_HD static Fp unpack32(const uint32_t* w) {  // plain integer < 2^(28 N), no Montgomery change
    Fp r;
#pragma unroll
    for (int j = 0; j < N; j++) {
      const int bit = 28 * j, q = bit >> 5, s = bit & 31;
      uint32_t lo = (q < P::N32) ? (w[q] >> s) : 0u;
      if (s > 4 && q + 1 < P::N32) lo |= w[q + 1] << (32 - s);
      r.v[j] = lo & MASK;
    }
    return r;
  }
  PCD_HD void pack32(uint32_t* w) const {  // limbs must be normalised and the value < 2^(32 N32)
#pragma unroll
    for (int k = 0; k < P::N32; k++) {
      const int bit = 32 * k, j = bit / 28, s = bit - 28 * j;  // word k = bits [32k, 32k+32)
      uint64_t x = (j < N) ? ((uint64_t)v[j] >> s) : 0;
      if (j + 1 < N) x |= (uint64_t)v[j + 1] << (28 - s);
      if (j + 2 < N && 56 - s < 32) x |= (uint64_t)v[j + 2] << (56 - s);
      w[k] = (uint32_t)x;
    }
  }
  // ABI Montgomery (x R, R = 2^(32 N32)) <-> internal (x R')
  PCD_HD static Fp from_abi(const uint32_t* w) {
    Fp c;
    for (int i = 0; i < N; i++) c.v[i] = P::cin(i);
    return unpack32(w) * c;
  }
  PCD_HD void to_abi(uint32_t* w) const {
    Fp c;
    for (int i = 0; i < N; i++) c.v[i] = P::cout(i);
    (*this * c).canonical().pack32(w);
  }
  // internal -> canonical integer words (`into_repr()`), and back
  PCD_HD void to_canonical_words(uint32_t* w) const {
    Fp o = zero();
    o.v[0] = 1;
    (*this * o).canonical().pack32(w);
  }
  PCD_HD static Fp from_canonical_words(const uint32_t* w) { return unpack32(w) * r2(); }
};

// ------------------------------------------------------------------------------------------------
// F[u]/(u^2 - NR)   (ark-ff Fp2; G2 coordinates of MNT4)
template <class F, unsigned NR>
struct Fp2 {
  typedef typename F::Params Params;
  typedef F Base;
  static constexpr int DEG = 2;
  static constexpr unsigned NONRESIDUE = NR;
  static constexpr int WORDS = 2 * F::WORDS;
  static constexpr int ABI_WORDS = 2 * F::ABI_WORDS;
  F c0, c1;
  PCD_HD static Fp2 zero() { return {F::zero(), F::zero()}; }
  PCD_HD static Fp2 one() { return {F::one(), F::zero()}; }
  PCD_HD bool is_raw_zero() const { return c0.is_raw_zero() && c1.is_raw_zero(); }
  PCD_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  PCD_HD bool operator==(const Fp2& b) const { return c0 == b.c0 && c1 == b.c1; }
  PCD_HD bool operator!=(const Fp2& b) const { return !(*this == b); }
  PCD_HD Fp2 operator+(const Fp2& b) const { return {c0 + b.c0, c1 + b.c1}; }
  PCD_HD Fp2 operator-(const Fp2& b) const { return {c0 - b.c0, c1 - b.c1}; }
  PCD_HD Fp2 neg() const { return {c0.neg(), c1.neg()}; }
  PCD_HD Fp2 dbl() const { return {c0.dbl(), c1.dbl()}; }
  // (not inlined, like Fp::mul: bounds the code size of the G2 point kernels; operands by reference --
  // large by-value aggregates passed on the stack miscompiled for Fp3 on gfx950 / ROCm 7.2)
  // schoolbook with lazy reduction: (a0 b0 + nr a1 b1) + (a0 b1 + a1 b0) u, one Montgomery reduction per coefficient
  __host__ __device__ __noinline__ static void mul_call(Fp2& o, const Fp2& a, const Fp2& b) { mul_impl(o, a, b); }
  __host__ __device__ __noinline__ static void sqr_call(Fp2& o, const Fp2& a) { sqr_impl(o, a); }
  PCD_HD static void mul(Fp2& o, const Fp2& a, const Fp2& b) { if constexpr (F::INLINE_ARITH) mul_impl(o, a, b); else mul_call(o, a, b); }
  PCD_HD static void sqr_(Fp2& o, const Fp2& a) { if constexpr (F::INLINE_ARITH) sqr_impl(o, a); else sqr_call(o, a); }
  PCD_HD static void mul_impl(Fp2& o, const Fp2& a, const Fp2& b) {
    F na1 = a.c1.mul_small(NR);
    F r0 = F::dot2(a.c0, b.c0, na1, b.c1);
    F r1 = F::dot2(a.c0, b.c1, a.c1, b.c0);
    o.c0 = r0;
    o.c1 = r1;
  }
  PCD_HD static void sqr_impl(Fp2& o, const Fp2& a) {
    F na1 = a.c1.mul_small(NR);
    F r0 = F::dot2(a.c0, a.c0, na1, a.c1);
    F r1 = (a.c0 * a.c1).dbl();
    o.c0 = r0;
    o.c1 = r1;
  }
  PCD_HD Fp2 operator*(const Fp2& b) const { Fp2 o; mul(o, *this, b); return o; }
  PCD_HD Fp2 sqr() const { Fp2 o; sqr_(o, *this); return o; }
  PCD_HD Fp2 mul_small(unsigned k) const { return {c0.mul_small(k), c1.mul_small(k)}; }
  PCD_HD Fp2 mul_base(const F& k) const { return {c0 * k, c1 * k}; }
  PCD_HD Fp2 inv() const {
    F n = (c0.sqr() - c1.sqr().mul_small(NR)).inv();
    return {c0 * n, (c1 * n).neg()};
  }
  PCD_HD static Fp2 load(const uint32_t* p) { return {F::load(p), F::load(p + F::WORDS)}; }
  PCD_HD void store(uint32_t* p) const { c0.store(p); c1.store(p + F::WORDS); }
  PCD_HD static Fp2 from_abi(const uint32_t* w) { return {F::from_abi(w), F::from_abi(w + F::ABI_WORDS)}; }
  PCD_HD void to_abi(uint32_t* w) const { c0.to_abi(w); c1.to_abi(w + F::ABI_WORDS); }
};

// The same field with its two coefficients SPLIT OVER A PAIR OF ADJACENT LANES (lane parity = coefficient index): every
// lane of the pair holds one base-field element per Fq2 value, so a Jacobian accumulator over the 753-bit Fq2 costs 81
// registers per lane instead of 162 -- the unsplit form does not fit the register file and its point kernels run out of
// scratch memory (15 % of the mad roofline against 41 % for the 753-bit G1).  Additions are coefficient-wise; a product
// exchanges the partner's coefficients (54 DPP moves) and each lane computes one output coefficient with one fused
// two-term product:   lane 0:  a0 b0 + (nr a1) b1      lane 1:  a0 b1 + a1 b0.
// Memory image unchanged (c0 || c1): load / store address the lane's own half.  Device only; both lanes of a pair must
// follow the same control flow (they work on the same point).
template <class F, unsigned NR>
struct Fp2S {
  typedef typename F::Params Params;
  typedef F Base;
  static constexpr int DEG = 2;
  static constexpr int WORDS = 2 * F::WORDS;
  static constexpr int ABI_WORDS = 2 * F::ABI_WORDS;
  static constexpr int LANES = 2;
  F c;
  PCD_DEV static unsigned parity() { return threadIdx.x & 1u; }
  PCD_DEV static F partner(const F& a) { F r;
#pragma unroll
    for (int i = 0; i < F::N; i++) r.v[i] = (uint32_t)__shfl_xor((int)a.v[i], 1, 64);
    return r; }
  PCD_DEV static bool both(bool b) { const int o = __shfl_xor((int)b, 1, 64); return b & (o != 0); }  // (no short circuit: both lanes shuffle)
  PCD_DEV static Fp2S zero() { return {F::zero()}; }
  PCD_DEV static Fp2S one() { F o = F::one(), z = F::zero(); return {parity() ? z : o}; }
  PCD_DEV bool is_raw_zero() const { return both(c.is_raw_zero()); }
  PCD_DEV bool is_zero() const { return both(c.is_zero()); }
  PCD_DEV bool operator==(const Fp2S& b) const { return (*this - b).is_zero(); }
  PCD_DEV bool operator!=(const Fp2S& b) const { return !(*this == b); }
  PCD_DEV Fp2S operator+(const Fp2S& b) const { return {c + b.c}; }
  PCD_DEV Fp2S operator-(const Fp2S& b) const { return {c - b.c}; }
  PCD_DEV Fp2S neg() const { return {c.neg()}; }
  PCD_DEV Fp2S dbl() const { return {c.dbl()}; }
  PCD_DEV Fp2S mul_small(unsigned k) const { return {c.mul_small(k)}; }
  PCD_DEV Fp2S operator*(const Fp2S& b) const {
    if constexpr (F::MAILBOX) return mb_mul(b);
    const F pa = partner(c), pb = partner(b.c);
    const F npa = pa.mul_small(NR);
    const bool odd = parity() != 0;
    F x, y;  // lane 0: x = a0, y = nr a1;  lane 1: x = a0 (partner's), y = a1
#pragma unroll
    for (int i = 0; i < F::N; i++) { x.v[i] = odd ? pa.v[i] : c.v[i]; y.v[i] = odd ? c.v[i] : npa.v[i]; }
    return {F::dot2(x, b.c, y, pb)};
  }
  PCD_DEV Fp2S sqr() const { return *this * *this; }
  // Mailbox form (F = the MB variant of Fp): every lane posts its coefficient of a and of b in its LDS slots; the non-inlined body
  // reads its own and its partner's (the exchange that is 54 DPP moves above costs nothing extra here) and posts the result.
  __device__ __noinline__ static void mb_mul_call(typename F::MbPtr mb) {
    const unsigned l = threadIdx.x, pl = l ^ 1u;
    const bool odd = (l & 1u) != 0;
    F::mb_sync();  // the partner's posts (made by the caller) are ordered before this lane's reads of them
    const F a0 = F::mb_get_lane(mb, 0, odd ? pl : l), a1 = F::mb_get_lane(mb, 0, odd ? l : pl);
    const F b = F::mb_get_lane(mb, 1, l), pb = F::mb_get_lane(mb, 1, pl);
    F::mb_sync();  // every lane of the pair has read both slots before either posts its result over slot 0
    const F y = a1.mul_small_var(odd ? 1u : NR);  // lane 0: nr a1;  lane 1: a1
    F o;
    F::template dot_impl<2>(o, a0, b, y, pb, a0, a0);
    F::mb_put(mb, 0, o);
  }
  PCD_DEV Fp2S mb_mul(const Fp2S& b) const {
    const typename F::MbPtr mb = F::mb_base();
    F::mb_put(mb, 0, c); F::mb_put(mb, 1, b.c);
    mb_mul_call(mb);
    F r = F::mb_get(mb, 0);
    F::mb_sync();  // (the next posts into these slots stay behind the partners' reads of this result round)
    return {r};
  }
  PCD_DEV static Fp2S load(const uint32_t* p) { return {F::load(p + parity() * F::WORDS)}; }
  PCD_DEV void store(uint32_t* p) const { c.store(p + parity() * F::WORDS); }
  PCD_DEV static Fp2S from_abi(const uint32_t* w) { return {F::from_abi(w + parity() * F::ABI_WORDS)}; }
  PCD_DEV void to_abi(uint32_t* w) const { c.to_abi(w + parity() * F::ABI_WORDS); }
};

// F[u]/(u^3 - NR)   (ark-ff Fp3; G2 coordinates of MNT6)
template <class F, unsigned NR>
struct Fp3 {
  typedef typename F::Params Params;
  typedef F Base;
  static constexpr int DEG = 3;
  static constexpr int WORDS = 3 * F::WORDS;
  static constexpr int ABI_WORDS = 3 * F::ABI_WORDS;
  F c0, c1, c2;
  PCD_HD static Fp3 zero() { return {F::zero(), F::zero(), F::zero()}; }
  PCD_HD static Fp3 one() { return {F::one(), F::zero(), F::zero()}; }
  PCD_HD bool is_raw_zero() const { return c0.is_raw_zero() && c1.is_raw_zero() && c2.is_raw_zero(); }
  PCD_HD bool is_zero() const { return c0.is_zero() && c1.is_zero() && c2.is_zero(); }
  PCD_HD bool operator==(const Fp3& b) const { return c0 == b.c0 && c1 == b.c1 && c2 == b.c2; }
  PCD_HD bool operator!=(const Fp3& b) const { return !(*this == b); }
  PCD_HD Fp3 operator+(const Fp3& b) const { return {c0 + b.c0, c1 + b.c1, c2 + b.c2}; }
  PCD_HD Fp3 operator-(const Fp3& b) const { return {c0 - b.c0, c1 - b.c1, c2 - b.c2}; }
  PCD_HD Fp3 neg() const { return {c0.neg(), c1.neg(), c2.neg()}; }
  PCD_HD Fp3 dbl() const { return {c0.dbl(), c1.dbl(), c2.dbl()}; }
  // schoolbook with lazy reduction, one Montgomery reduction per coefficient:
  //   c0 = a0 b0 + nr (a1 b2 + a2 b1),  c1 = a0 b1 + a1 b0 + nr a2 b2,  c2 = a0 b2 + a1 b1 + a2 b0
  __host__ __device__ __noinline__ static void mul(Fp3& o, const Fp3& a, const Fp3& b) { mul_impl(o, a, b); }
  __host__ __device__ __noinline__ static void sqr_(Fp3& o, const Fp3& a) { sqr_impl(o, a); }
  PCD_HD static void mul_impl(Fp3& o, const Fp3& a, const Fp3& b) {
    F na1 = a.c1.mul_small(NR), na2 = a.c2.mul_small(NR);
    F r0 = F::dot3(a.c0, b.c0, na1, b.c2, na2, b.c1);
    F r1 = F::dot3(a.c0, b.c1, a.c1, b.c0, na2, b.c2);
    F r2 = F::dot3(a.c0, b.c2, a.c1, b.c1, a.c2, b.c0);
    o.c0 = r0;
    o.c1 = r1;
    o.c2 = r2;
  }
  //   c0 = a0^2 + 2 nr a1 a2,  c1 = 2 a0 a1 + nr a2^2,  c2 = a1^2 + 2 a0 a2
  PCD_HD static void sqr_impl(Fp3& o, const Fp3& a) {
    F d0 = a.c0.dbl(), na2 = a.c2.mul_small(NR), dna1 = a.c1.mul_small(2 * NR);
    F r0 = F::dot2(a.c0, a.c0, dna1, a.c2);
    F r1 = F::dot2(d0, a.c1, na2, a.c2);
    F r2 = F::dot2(a.c1, a.c1, d0, a.c2);
    o.c0 = r0;
    o.c1 = r1;
    o.c2 = r2;
  }
  // (always through the non-inlined copies: with the products inlined the MNT6 G2 point kernels take > 15 min to compile)
  PCD_HD Fp3 operator*(const Fp3& b) const { Fp3 o; mul(o, *this, b); return o; }
  PCD_HD Fp3 sqr() const { Fp3 o; sqr_(o, *this); return o; }
  PCD_HD Fp3 mul_small(unsigned k) const { return {c0.mul_small(k), c1.mul_small(k), c2.mul_small(k)}; }
  PCD_HD Fp3 mul_base(const F& k) const { return {c0 * k, c1 * k, c2 * k}; }
  PCD_HD Fp3 inv() const {
    F t0 = c0.sqr() - (c1 * c2).mul_small(NR);
    F t1 = c2.sqr().mul_small(NR) - c0 * c1;
    F t2 = c1.sqr() - c0 * c2;
    F n = (c0 * t0 + (c2 * t1 + c1 * t2).mul_small(NR)).inv();
    return {t0 * n, t1 * n, t2 * n};
  }
  PCD_HD static Fp3 load(const uint32_t* p) { return {F::load(p), F::load(p + F::WORDS), F::load(p + 2 * F::WORDS)}; }
  PCD_HD void store(uint32_t* p) const { c0.store(p); c1.store(p + F::WORDS); c2.store(p + 2 * F::WORDS); }
  PCD_HD static Fp3 from_abi(const uint32_t* w) { return {F::from_abi(w), F::from_abi(w + F::ABI_WORDS), F::from_abi(w + 2 * F::ABI_WORDS)}; }
  PCD_HD void to_abi(uint32_t* w) const { c0.to_abi(w); c1.to_abi(w + F::ABI_WORDS); c2.to_abi(w + 2 * F::ABI_WORDS); }
};

// The cubic extension with its three coefficients SPLIT OVER A TRIPLE OF ADJACENT LANES (lane index mod 3 = coefficient index;
// a 64-lane wave holds 21 triples, lane 63 idles): every lane keeps ONE base-field element per Fq3 value, so a Jacobian
// accumulator costs 3 N registers per lane instead of 9 N.  The unsplit form does not fit the register file (298-bit: 99 of the
// accumulator alone plus a 66-register point; its products were function calls through scratch, ~20 % of the mad roofline).
// Additions are coefficient-wise; a product fetches the partners' coefficients (4 N ds_bpermute) and each lane computes ONE output
// coefficient with one fused three-term product:
//   lane 0:  a0 b0 + (nr a1) b2 + (nr a2) b1      lane 1:  a1 b0 + a0 b1 + (nr a2) b2      lane 2:  a2 b0 + a1 b1 + a0 b2
// Memory image unchanged (c0 || c1 || c2): load / store address the lane's own third.  Device only; the three lanes of a triple
// must follow the same control flow (they work on the same point).
template <class F, unsigned NR>
struct Fp3S {
  typedef typename F::Params Params;
  typedef F Base;
  static constexpr int DEG = 3;
  static constexpr int WORDS = 3 * F::WORDS;
  static constexpr int ABI_WORDS = 3 * F::ABI_WORDS;
  static constexpr int LANES = 3;
  F c;
  PCD_DEV static unsigned role() { return (threadIdx.x & 63u) % 3u; }
  PCD_DEV static int lane_next() { const unsigned l = threadIdx.x & 63u; return (int)l + (l % 3u == 2u ? -2 : 1); }   // holder of coefficient (role + 1) mod 3
  PCD_DEV static int lane_next2() { const unsigned l = threadIdx.x & 63u; return (int)l + (l % 3u == 0u ? 2 : -1); }  // ... (role + 2) mod 3
  PCD_DEV static F from_lane(const F& a, int lane) { F r;
#pragma unroll
    for (int i = 0; i < F::N; i++) r.v[i] = (uint32_t)__shfl((int)a.v[i], lane, 64);
    return r; }
  PCD_DEV static F sel(bool t, const F& a, const F& b) { F r;
#pragma unroll
    for (int i = 0; i < F::N; i++) r.v[i] = t ? a.v[i] : b.v[i];
    return r; }
#if defined(PCD_FP3S_DPP) && PCD_FP3S_DPP
  // EXPERIMENT (round 6; measured and left off: DESIGN.md section 4): both partners' coefficients by DPP wave shifts instead of
  // two ds_bpermute round trips -- lane + 1 / + 2 (wave_shl:1 once / twice) for the roles whose partner sits above, lane - 1 / - 2
  // (wave_shr:1) for those below; 4 DPP moves + 2 selects per limb, no LDS traffic, no waitcnt.  A triple's lanes are active together,
  // so every shifted value that is USED comes from an active lane of the same triple.
  PCD_DEV static void partners(const F& a, F& an, F& an2) {
    const unsigned k = role();
#pragma unroll
    for (int i = 0; i < F::N; i++) {
      const int v = (int)a.v[i];
      const int s1 = __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false);    // wave_shl:1 -- from lane + 1
      const int s2 = __builtin_amdgcn_update_dpp(0, s1, 0x130, 0xf, 0xf, false);   //              from lane + 2
      const int r1 = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false);    // wave_shr:1 -- from lane - 1
      const int r2 = __builtin_amdgcn_update_dpp(0, r1, 0x138, 0xf, 0xf, false);   //              from lane - 2
      an.v[i] = (uint32_t)(k == 2 ? r2 : s1);
      an2.v[i] = (uint32_t)(k == 0 ? s2 : r1);
    }
  }
#else
  PCD_DEV static void partners(const F& a, F& an, F& an2) { an = from_lane(a, lane_next()); an2 = from_lane(a, lane_next2()); }
#endif
  PCD_DEV static bool all3(bool b) {  // (no short circuit: all three lanes shuffle)
    const int o1 = __shfl((int)b, lane_next(), 64), o2 = __shfl((int)b, lane_next2(), 64);
    return b & (o1 != 0) & (o2 != 0);
  }
  PCD_DEV static Fp3S zero() { return {F::zero()}; }
  PCD_DEV static Fp3S one() { F o = F::one(), z = F::zero(); return {sel(role() == 0, o, z)}; }
  PCD_DEV bool is_raw_zero() const { return all3(c.is_raw_zero()); }
  PCD_DEV bool is_zero() const { return all3(c.is_zero()); }
  PCD_DEV bool operator==(const Fp3S& b) const { return (*this - b).is_zero(); }
  PCD_DEV bool operator!=(const Fp3S& b) const { return !(*this == b); }
  PCD_DEV Fp3S operator+(const Fp3S& b) const { return {c + b.c}; }
  PCD_DEV Fp3S operator-(const Fp3S& b) const { return {c - b.c}; }
  PCD_DEV Fp3S neg() const { return {c.neg()}; }
  PCD_DEV Fp3S dbl() const { return {c.dbl()}; }
  PCD_DEV Fp3S mul_small(unsigned k) const { return {c.mul_small(k)}; }
  // With own = a_k, an = a_(k+1), an2 = a_(k+2) (indices mod 3) and the same for b, output coefficient k is
  //   own * y1 + x2 * y2 + x3 * y3   with   (y1, y2, y3) = (b_k, b_(k+2), b_(k+1)) rotated by the role,
  //   x2 = an (times nr on lanes 0, 1),  x3 = an2 (times nr on lane 0)
  PCD_DEV Fp3S operator*(const Fp3S& b) const {
    if constexpr (F::MAILBOX) return mb_mul(b);
    const unsigned k = role();
    F an, an2, bn, bn2;
    partners(c, an, an2);
    partners(b.c, bn, bn2);
    const F x2 = sel(k < 2, an.mul_small(NR), an), x3 = sel(k == 0, an2.mul_small(NR), an2);
    const F y1 = sel(k == 0, b.c, sel(k == 1, bn2, bn));
    const F y2 = sel(k == 0, bn2, sel(k == 1, bn, b.c));
    const F y3 = sel(k == 0, bn, sel(k == 1, b.c, bn2));
    return {F::dot3(c, y1, x2, y2, x3, y3)};
  }
  //   lane 0:  a0 a0 + (2 nr a1) a2      lane 1:  (2 a0) a1 + (nr a2) a2      lane 2:  a1 a1 + (2 a0) a2
  // Mailbox forms: coefficients are posted in the lanes' LDS slots and the non-inlined bodies read the partners' directly (no
  // ds_bpermute exchange, no data selects -- the role picks addresses)
  __device__ __noinline__ static void mb_mul_call(typename F::MbPtr mb) {
    // l0: the lane holding coefficient 0 of this triple (lane 63 has no triple: it idles in every kernel; should it ever come here it
    // reads lane 61's triple instead of indexing past the 64 slots)
    const unsigned l = threadIdx.x & 63u, k = l % 3u, l0 = l == 63u ? 60u : l - k;
    const unsigned k1 = k == 2 ? 0 : k + 1, k2 = k == 0 ? 2 : k - 1;    // (k + 1) mod 3, (k + 2) mod 3
    F::mb_sync();  // the partners' posts (made by the caller) are ordered before this lane's reads of them
    const F a = F::mb_get_lane(mb, 0, l), an = F::mb_get_lane(mb, 0, l0 + k1), an2 = F::mb_get_lane(mb, 0, l0 + k2);
    // own a_k meets b_0, a_(k+1) meets b_2, a_(k+2) meets b_1 on every lane (the lane formulas above)
    const F y1 = F::mb_get_lane(mb, 1, l0), y2 = F::mb_get_lane(mb, 1, l0 + 2), y3 = F::mb_get_lane(mb, 1, l0 + 1);
    F::mb_sync();  // every lane of the triple has read all six slots before any posts its result over slot 0
    const F x2 = an.mul_small_var(k < 2 ? NR : 1u), x3 = an2.mul_small_var(k == 0 ? NR : 1u);
    F o;
    F::template dot_impl<3>(o, a, y1, x2, y2, x3, y3);
    F::mb_put(mb, 0, o);
  }
  __device__ __noinline__ static void mb_sqr_call(typename F::MbPtr mb) {
    const unsigned l = threadIdx.x & 63u, k = l % 3u, l0 = l == 63u ? 60u : l - k;
    const unsigned k1 = k == 2 ? 0 : k + 1, k2 = k == 0 ? 2 : k - 1;
    F::mb_sync();
    const F a = F::mb_get_lane(mb, 0, l), an = F::mb_get_lane(mb, 0, l0 + k1), an2 = F::mb_get_lane(mb, 0, l0 + k2);
    F::mb_sync();
    //   lane 0:  a0 a0 + (2 nr a1) a2      lane 1:  (2 a0) a1 + (nr a2) a2      lane 2:  a1 a1 + (2 a0) a2
    const F x1 = sel(k == 0, a, an2).mul_small_var(k == 1 ? 2u : 1u);
    const F y1 = sel(k == 2, an2, a);
    const F x2 = an.mul_small_var(k == 0 ? 2 * NR : (k == 1 ? NR : 2));
    const F y2 = sel(k == 0, an2, sel(k == 1, an, a));
    F o;
    F::template dot_impl<2>(o, x1, y1, x2, y2, x1, x1);
    F::mb_put(mb, 0, o);
  }
  PCD_DEV Fp3S mb_mul(const Fp3S& b) const {
    const typename F::MbPtr mb = F::mb_base();
    F::mb_put(mb, 0, c); F::mb_put(mb, 1, b.c);
    mb_mul_call(mb);
    F r = F::mb_get(mb, 0);
    F::mb_sync();  // (the next posts into these slots stay behind the partners' reads of this result round)
    return {r};
  }
  PCD_DEV Fp3S mb_sqr() const {
    const typename F::MbPtr mb = F::mb_base();
    F::mb_put(mb, 0, c);
    mb_sqr_call(mb);
    F r = F::mb_get(mb, 0);
    F::mb_sync();
    return {r};
  }
  PCD_DEV Fp3S sqr() const {
    if constexpr (F::MAILBOX) return mb_sqr();
    const unsigned k = role();
    F an, an2;
    partners(c, an, an2);
    const F x1 = sel(k == 0, c, sel(k == 1, an2.dbl(), an2));
    const F y1 = sel(k == 2, an2, c);
    const F x2 = an.mul_small_var(k == 0 ? 2 * NR : (k == 1 ? NR : 2));
    const F y2 = sel(k == 0, an2, sel(k == 1, an, c));
    return {F::dot2(x1, y1, x2, y2)};
  }
  // x * (a u^2) for the twist coefficient (0, 0, a):  (nr a c1, nr a c2, a c0) -- every lane takes its successor's coefficient
  PCD_DEV Fp3S mul_by_au2(unsigned a) const { return {from_lane(c, lane_next()).mul_small_var(role() < 2 ? a * NR : a)}; }
  PCD_DEV static Fp3S load(const uint32_t* p) { return {F::load(p + role() * F::WORDS)}; }
  PCD_DEV void store(uint32_t* p) const { c.store(p + role() * F::WORDS); }
  PCD_DEV static Fp3S from_abi(const uint32_t* w) { return {F::from_abi(w + role() * F::ABI_WORDS)}; }
  PCD_DEV void to_abi(uint32_t* w) const { c.to_abi(w + role() * F::ABI_WORDS); }
};

}  // namespace pcd
