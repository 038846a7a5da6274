// One object per curve (compile with -DPCD_CURVE_IDX=0..3): pairing kernels (pairing.hip.h).
#include "common.h"
#include "pairing.hip.h"
#include "pairing_vm.hip.h"

namespace pcd {

#if PCD_CURVE_IDX == 0
typedef PC_MNT4_298 PCT;
typedef vmgen::MNT4_298 VG;
#define VM_TABLE(x) vmgen::MNT4_298_##x
#elif PCD_CURVE_IDX == 1
typedef PC_MNT6_298 PCT;
typedef vmgen::MNT6_298 VG;
#define VM_TABLE(x) vmgen::MNT6_298_##x
#elif PCD_CURVE_IDX == 2
typedef PC_MNT4_753 PCT;
typedef vmgen::MNT4_753 VG;
#define VM_TABLE(x) vmgen::MNT4_753_##x
#elif PCD_CURVE_IDX == 3
typedef PC_MNT6_753 PCT;
typedef vmgen::MNT6_753 VG;
#define VM_TABLE(x) vmgen::MNT6_753_##x
#else
#error "PCD_CURVE_IDX must be 0..3"
#endif

namespace {

typedef Pairing<PCT> PE;
typedef typename PE::Fq Fq;
typedef typename PE::E E;
typedef typename PE::Fqk Fqk;
constexpr int A1A = Aff<Fq>::ABI_WORDS, A2A = Aff<E>::ABI_WORDS, GW = Fqk::WORDS, GWA = Fqk::ABI_WORDS;

// one lane per pair: f_i = miller_loop(P_i, Q_i); points arrive in the C-ABI image
__global__ void __launch_bounds__(64) miller_kernel(const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2, uint32_t n,
                                                    uint32_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fqk f = PE::miller_loop(Aff<Fq>::from_abi(g1 + (size_t)i * A1A), Aff<E>::from_abi(g2 + (size_t)i * A2A));
  f.store(out + (size_t)i * GW);
}
// one lane per group of `per` consecutive Miller values: their product, then one final exponentiation
__global__ void __launch_bounds__(64) final_exp_kernel(const uint32_t* __restrict__ fs, uint32_t groups, uint32_t per,
                                                       uint32_t* __restrict__ out) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= groups) return;
  typename PE::Frob t;
  frob_init<Fq, PE::K, PCT::NR>(t);
  Fqk f = Fqk::one();
  for (uint32_t i = 0; i < per; i++) f = f * Fqk::load(fs + ((size_t)g * per + i) * GW);
  PE::final_exponentiation(f, t).to_abi(out + (size_t)g * GWA);
}

// ---- one wave per pairing (pairing_vm.hip.h): the latency form, used for small batches -------------------------------------------
typedef VmWave<Fq, VG> VM;
constexpr int KK = PE::K, DD = PE::K / 2;
// flat coefficient j (of v^j) <-> tower position: c_(j mod 2), coefficient j / 2 of the twist field
PCD_DEV int tower_word(int j, int words) { return ((j & 1) * DD + (j >> 1)) * words; }

// f_i = the Miller value of pair i up to a factor in Fq* (the lines are evaluated times pz^3: see tools/gen_pairing_vm.py) and
// UN-INVERTED when the loop count is negative -- vm_final_exp_kernel kills the one and conjugates for the other.  g1z (nullable):
// the Z coordinates of the G1 points, which then are Jacobian (X, Y, Z) with X, Y in g1 -- no inversion for a point that was just summed
__global__ void __launch_bounds__(64) vm_miller_kernel(const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g1z, const uint32_t* __restrict__ g2,
                                                       uint32_t n, uint32_t* __restrict__ out, const VmTables tb) {
  extern __shared__ __attribute__((aligned(16))) uint32_t vm_lds[];
  const uint32_t pair = blockIdx.x, lane = threadIdx.x;
  VM vm;
  vm.init((VM::Lds)vm_lds, tb);
  // lanes 0, 1: the G1 coordinates, lane 2: its Z; lanes 3 .. 3 + 2 DD - 1: the G2 coefficients (x then y); (0, 0) is the point at infinity
  bool nz = false;
  if (lane < 3u + 2u * DD) {
    const uint32_t* src = lane < 2 ? g1 + (size_t)pair * A1A + lane * Fq::ABI_WORDS
                        : lane == 2 ? (g1z ? g1z + (size_t)pair * Fq::ABI_WORDS : nullptr) : g2 + (size_t)pair * A2A + (lane - 3) * Fq::ABI_WORDS;
    Fq v = Fq::one();
    if (src) {
      for (int i = 0; i < Fq::ABI_WORDS; i++) nz = nz || src[i] != 0;
      v = Fq::from_abi(src);
    } else nz = true;
    int r;
    if (lane == 0) r = VG::R_PX0;
    else if (lane == 1) r = VG::R_PY0;
    else if (lane == 2) r = VG::R_PZ0;
    else { const int c = (int)lane - 3; r = (c < DD ? VG::R_QX0 : VG::R_QY0) + (c % DD); }  // (qx0, qx2, .. and qy0, qy2, .. are consecutive registers)
    vm.set_reg(r, v);
  }
  const unsigned long long m = __ballot(nz);
  const bool inf = (m & 3ull) == 0 || (m & 4ull) == 0 || (m >> 3) == 0;   // (0, 0) in G1, Z = 0, or (0, 0) in G2
  __syncthreads();
  if (inf) {  // e(O, Q) = e(P, O) = 1
    if (lane < (uint32_t)KK) (lane == 0 ? Fq::one() : Fq::zero()).store(out + (size_t)pair * GW + tower_word((int)lane, Fq::WORDS));
    return;
  }
  vm.run_flat(tb.flat, 0, tb.nflat);
  if (lane < (uint32_t)KK) vm.get_state(VG::S_F0 + (int)lane).store(out + (size_t)pair * GW + tower_word((int)lane, Fq::WORDS));
}
// one wave per group of `per` consecutive Miller values: their product, then the final exponentiation
__global__ void __launch_bounds__(64) vm_final_exp_kernel(const uint32_t* __restrict__ fs, uint32_t groups, uint32_t per,
                                                          uint32_t* __restrict__ out, const VmTables tb) {
  extern __shared__ __attribute__((aligned(16))) uint32_t vm_lds[];
  const uint32_t g = blockIdx.x, lane = threadIdx.x;
  VM vm;
  vm.init((VM::Lds)vm_lds, tb);
  if (lane < (uint32_t)KK) {
    Fq v = lane == 0 ? Fq::one() : Fq::zero();
    if (per) v = Fq::load(fs + (size_t)g * per * GW + tower_word((int)lane, Fq::WORDS));
    vm.set_state(VG::S_ACC0 + (int)lane, v);
  }
  __syncthreads();
  for (uint32_t i = 1; i < per; i++) {
    if (lane < (uint32_t)KK) vm.set_reg(VG::R_G0 + (int)lane, Fq::load(fs + ((size_t)g * per + i) * GW + tower_word((int)lane, Fq::WORDS)));
    __syncthreads();
    vm.run_flat(tb.flat, tb.alone_off, tb.alone_len);   // fe_mul
  }
  vm.run_flat(tb.flat, 0, tb.nflat);  // norm, the one field inversion (sliding window over p - 2), easy part, w0 power (sliding window), last product
  if (lane < (uint32_t)KK) vm.get_state(VG::S_ACC0 + (int)lane).to_abi(out + (size_t)g * GWA + tower_word((int)lane, Fq::ABI_WORDS));
}

// gt_out[g] = final_exp(prod_{i < per} miller(P_{g per + i}, Q_{g per + i})), g < groups.
// Up to VM_MAX_PAIRS pairs: one wave per pairing (a Miller loop's dependent chain is ~15x shorter); beyond, one lane per pairing
// (64 pairings per wave: the throughput form for batches that fill the chip anyway).  g1z_dev (nullable, wave form only): Z
// coordinates of Jacobian G1 points.
constexpr uint32_t VM_MAX_PAIRS = PCD_VM_MAX_PAIRS;  // (common.h: the host side decides by the same number whether to ask for Jacobian G1 inputs)
hipError_t multi_pairing(hipStream_t st, const uint32_t* g1_dev, const uint32_t* g1z_dev, const uint32_t* g2_dev, uint32_t groups, uint32_t per,
                         uint32_t* scratch, uint32_t* gt_out, const VmCurveTables* vm) {
  const uint32_t n = groups * per;
  if (vm && n <= VM_MAX_PAIRS) {
    if (n) hipLaunchKernelGGL(vm_miller_kernel, dim3(n), dim3(64), (size_t)VM::lds_words(vm->miller) * 4, st, g1_dev, g1z_dev, g2_dev, n, scratch, vm->miller);
    if (groups) hipLaunchKernelGGL(vm_final_exp_kernel, dim3(groups), dim3(64), (size_t)VM::lds_words(vm->final_exp) * 4, st, scratch, groups, per, gt_out, vm->final_exp);
    return hipGetLastError();
  }
  if (g1z_dev) return hipErrorInvalidValue;  // (the lane-per-pairing kernels take affine points)
  if (n) hipLaunchKernelGGL(miller_kernel, dim3((n + 63) / 64), dim3(64), 0, st, g1_dev, g2_dev, n, scratch);
  if (groups) hipLaunchKernelGGL(final_exp_kernel, dim3((groups + 63) / 64), dim3(64), 0, st, scratch, groups, per, gt_out);
  return hipGetLastError();
}

// out[i] = k_i * P_i as a Jacobian point in the C-ABI image; k_i: `kwords` canonical u32 words; P_i affine, C-ABI image ((0, 0) =
// infinity).  One lane per product (the random-linear-combination batch verification scales each proof's A and C by a 128-bit
// challenge: a few hundred group operations of latency next to Miller loops of tens of thousands).
typedef typename PCT::G1 G1C;
__global__ void __launch_bounds__(64) g1_scale_kernel(const uint32_t* __restrict__ g1, const uint32_t* __restrict__ k, uint32_t kwords, uint32_t n,
                                                      uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Aff<Fq> p = Aff<Fq>::from_abi(g1 + (size_t)i * A1A);
  Jac<Fq> r = Jac<Fq>::infinity();
  if (!p.is_inf()) {
    const Jac<Fq> pj = {p.x, p.y, Fq::one()};
    r = EC<G1C>::mul(pj, k + (size_t)i * kwords, (int)kwords);
  }
  if (r.is_inf()) r = Jac<Fq>::infinity();
  r.to_abi(out + (size_t)i * Jac<Fq>::ABI_WORDS);
}
hipError_t g1_scale(hipStream_t st, const uint32_t* g1_dev, const uint32_t* k_dev, uint32_t kwords, uint32_t n, uint32_t* out_dev) {
  if (n) hipLaunchKernelGGL(g1_scale_kernel, dim3((n + 63) / 64), dim3(64), 0, st, g1_dev, k_dev, kwords, n, out_dev);
  return hipGetLastError();
}

}  // namespace

#define PCD_CAT_(a, b) a##b
#define PCD_CAT(a, b) PCD_CAT_(a, b)
// the generated program tables of this curve -> one device block; `out` points into it
static hipError_t vm_upload(hipStream_t st, void** block, VmCurveTables* out) {
  struct Part { const void* src; size_t bytes; };
  const Part parts[] = {
      {VM_TABLE(miller_code), sizeof(VM_TABLE(miller_code))}, {VM_TABLE(miller_flat), sizeof(VM_TABLE(miller_flat))},
      {VM_TABLE(final_exp_code), sizeof(VM_TABLE(final_exp_code))}, {VM_TABLE(final_exp_flat), sizeof(VM_TABLE(final_exp_flat))},
      {VM_TABLE(consts), sizeof(VM_TABLE(consts))}};
  constexpr int NP = 5;
  size_t off[NP + 1] = {0};
  for (int i = 0; i < NP; i++) off[i + 1] = off[i] + ((parts[i].bytes + 15) & ~(size_t)15);
  void* blk = nullptr;
  hipError_t e = hipMalloc(&blk, off[NP]);
  if (e != hipSuccess) return e;
  char* d = (char*)blk;
  for (int i = 0; i < NP && e == hipSuccess; i++) e = hipMemcpyAsync(d + off[i], parts[i].src, parts[i].bytes, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { (void)hipFree(blk); return e; }  // nothing is published: the next pairing call uploads again
  auto at = [&](int i) { return (const uint32_t*)(d + off[i]); };
  // (progs / steps / script stay on the host: the device walks the flat records)
  out->miller = {nullptr, nullptr, at(0), at(4), nullptr, 0, 0, (uint32_t)(parts[0].bytes / 4), 0, at(1), VM_TABLE(miller_flat_script_len), 0, 0};
  out->final_exp = {nullptr, nullptr, at(2), at(4), nullptr, 0, 0, (uint32_t)(parts[2].bytes / 4), 0, at(3), VM_TABLE(final_exp_flat_script_len),
                    VM_TABLE(final_exp_flat_FE_MUL_off), VM_TABLE(final_exp_flat_FE_MUL_len)};
  *block = blk;  // block and tables become visible together, only after every copy has landed
  return hipSuccess;
}
const PairingEntry* PCD_CAT(pcd_pairing_entry_, PCD_CURVE_IDX)() {
  static const PairingEntry e = {GWA, GW, multi_pairing, vm_upload};
  return &e;
}
typedef hipError_t (*G1ScaleFn)(hipStream_t, const uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*);
G1ScaleFn PCD_CAT(pcd_g1_scale_entry_, PCD_CURVE_IDX)() { return g1_scale; }

}  // namespace pcd
