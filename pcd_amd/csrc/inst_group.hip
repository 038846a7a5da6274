// One object per group (compile with -DPCD_GROUP_IDX=0..7, idx = 2 * curve_id + (group_id - 1)):
// MSM driver + small point utilities instantiated for that group.
#include "common.h"
#include "fixed_base.hip.h"

namespace pcd {

#if PCD_GROUP_IDX == 0
typedef G1_MNT4_298 GT;
#elif PCD_GROUP_IDX == 1
typedef G2_MNT4_298 GT;
#elif PCD_GROUP_IDX == 2
typedef G1_MNT6_298 GT;
#elif PCD_GROUP_IDX == 3
typedef G2_MNT6_298 GT;
#elif PCD_GROUP_IDX == 4
typedef G1_MNT4_753 GT;
#elif PCD_GROUP_IDX == 5
typedef G2_MNT4_753 GT;
#elif PCD_GROUP_IDX == 6
typedef G1_MNT6_753 GT;
#elif PCD_GROUP_IDX == 7
typedef G2_MNT6_753 GT;
#else
#error "PCD_GROUP_IDX must be 0..7"
#endif

namespace {

typedef typename GT::F F;

// the point sums run in the lane-split form of the group (see MsmItems): per-lane scratch and latency shrink with it
typedef MsmItems<GT> IT;
typedef typename IT::GA GA;
typedef typename GA::F FA;

// host-facing utilities: inputs and outputs in the C-ABI image
__global__ void __launch_bounds__(64) points_sum_kernel(const uint32_t* __restrict__ in_abi, uint32_t n, uint32_t* __restrict__ scratch,
                                                        uint32_t* __restrict__ out_abi) {
  // one workgroup: strided partial sums per item, then a tree through `scratch` (64 Jacobian points, device image)
  if (blockIdx.x != 0) return;
  constexpr uint32_t PW = IT::PER_WAVE;
  const bool live = !IT::idle();
  const uint32_t it = IT::local();
  Jac<FA> acc = Jac<FA>::infinity();
  if (live) {
    for (uint32_t i = it; i < n; i += PW) acc = EC<GA>::add(acc, Jac<FA>::from_abi(in_abi + (size_t)i * Jac<F>::ABI_WORDS));
    acc.store(scratch + (size_t)it * Jac<F>::WORDS);
  }
  __syncthreads();
  for (uint32_t s = 32; s > 0; s >>= 1) {
    if (live && it < s && it + s < PW && it + s < n) {  // (items >= n hold the identity)
      acc = EC<GA>::add(acc, Jac<FA>::load(scratch + (size_t)(it + s) * Jac<F>::WORDS));
      acc.store(scratch + (size_t)it * Jac<F>::WORDS);
    }
    __syncthreads();
  }
  if (!live || it != 0) return;
  if (acc.is_inf()) acc = Jac<FA>::infinity();
  acc.to_abi(out_abi);
}
__global__ void __launch_bounds__(64) to_affine_kernel(const uint32_t* __restrict__ in_abi, uint32_t n, uint32_t* __restrict__ out_abi) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff<F> a = EC<GT>::to_affine(Jac<F>::from_abi(in_abi + (size_t)i * Jac<F>::ABI_WORDS));
  a.to_abi(out_abi + (size_t)i * Aff<F>::ABI_WORDS);
}

__global__ void __launch_bounds__(64) jac_sum_parts_kernel(const uint32_t* __restrict__ in, size_t part_stride_words, uint32_t parts, uint32_t slots,
                                                           uint32_t* __restrict__ out) {
  if (IT::idle()) return;
  const uint32_t s = IT::item();  // one item per slot
  if (s >= slots) return;
  Jac<FA> acc = Jac<FA>::load(in + (size_t)s * Jac<F>::WORDS);
  for (uint32_t g = 1; g < parts; g++) acc = EC<GA>::add(acc, Jac<FA>::load(in + (size_t)g * part_stride_words + (size_t)s * Jac<F>::WORDS));
  acc.store(out + (size_t)s * Jac<F>::WORDS);
}
hipError_t jac_sum_parts_entry(hipStream_t st, const uint32_t* in, size_t part_stride_words, uint32_t parts, uint32_t slots, uint32_t* out) {
  if (slots == 0) return hipSuccess;
  hipLaunchKernelGGL(jac_sum_parts_kernel, dim3(IT::grid(slots)), dim3(64), 0, st, in, part_stride_words, parts, slots, out);
  return hipGetLastError();
}

hipError_t msm_entry(MsmWorkspace& ws, hipStream_t st, const MsmBasesView& bases, const uint32_t* scalars, uint32_t n,
                     uint32_t* out_dev, int c, uint32_t chunk, int sort_mode, MsmTimings* tm, MsmSharedSort* share, int share_role) {
  return msm_run<GT>(ws, st, bases, scalars, n, out_dev, c, chunk, sort_mode, tm, share, share_role);
}
hipError_t precompute_entry(hipStream_t st, uint32_t* pts, uint32_t n, int groups, int shift) {
  return msm_precompute<GT>(st, pts, n, groups, shift);
}
hipError_t points_in_entry(hipStream_t st, const uint32_t* abi, uint32_t n, uint32_t* out) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL((points_abi_to_internal_kernel<GT>), dim3((n + 63) / 64), dim3(64), 0, st, abi, n, out);
  return hipGetLastError();
}
hipError_t jac_out_entry(hipStream_t st, const uint32_t* in, uint32_t n, uint32_t* abi) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL((jac_internal_to_abi_kernel<GT>), dim3((n + 63) / 64), dim3(64), 0, st, in, n, abi);
  return hipGetLastError();
}
hipError_t points_sum_entry(hipStream_t st, const uint32_t* jac, uint32_t n, uint32_t* scratch, uint32_t* out) {
  hipLaunchKernelGGL(points_sum_kernel, dim3(1), dim3(64), 0, st, jac, n, scratch, out);
  return hipGetLastError();
}
hipError_t to_affine_entry(hipStream_t st, const uint32_t* jac, uint32_t n, uint32_t* aff) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(to_affine_kernel, dim3((n + 63) / 64), dim3(64), 0, st, jac, n, aff);
  return hipGetLastError();
}

constexpr int FB_NWIN = (GT::FR::BITS + FB_WINDOW - 1) / FB_WINDOW;
constexpr size_t FB_TABLE_WORDS = ((size_t)FB_NWIN << FB_WINDOW) * Aff<F>::WORDS + (size_t)FB_NWIN * Jac<F>::WORDS;
hipError_t fixed_base_entry(hipStream_t st, const uint32_t* base_abi, const uint32_t* scalars, uint32_t n, uint32_t* table_scratch,
                            uint32_t* jac_scratch, uint32_t* out_abi, uint8_t* out_inf) {
  uint32_t* bj = table_scratch + ((size_t)FB_NWIN << FB_WINDOW) * Aff<F>::WORDS;
  return fixed_base_run<GT>(st, base_abi, scalars, n, bj, table_scratch, jac_scratch, out_abi, out_inf);
}

hipError_t fb_tables_entry(hipStream_t st, const uint32_t* bases_abi, uint32_t ni, uint32_t* tables) {
  return fb_tables_build<GT>(st, bases_abi, ni, FB_TABLE_WORDS, tables);
}
hipError_t fb_inputs_entry(hipStream_t st, const uint32_t* tables, const uint32_t* abc0_abi, uint32_t ni, const uint32_t* scalars, uint32_t k,
                           uint32_t* scratch, uint32_t* out_abi, uint8_t* out_inf, uint32_t* out_z_abi, uint32_t out_stride) {
  return fb_inputs_run<GT>(st, tables, FB_TABLE_WORDS, abc0_abi, ni, scalars, k, scratch, out_abi, out_inf, out_z_abi, out_stride);
}

}  // namespace

#define PCD_CAT_(a, b) a##b
#define PCD_CAT(a, b) PCD_CAT_(a, b)
const GroupEntry* PCD_CAT(pcd_group_entry_, PCD_GROUP_IDX)() {
  static const GroupEntry e = {Aff<F>::WORDS, MsmBaseStride<GT>::value, Aff<F>::ABI_WORDS, GT::FR::N32, GT::FR::BITS, msm_entry, precompute_entry,
                               points_in_entry, jac_out_entry, points_sum_entry, jac_sum_parts_entry, to_affine_entry, FB_TABLE_WORDS, fixed_base_entry,
                               fb_tables_entry, fb_inputs_entry};
  return &e;
}

}  // namespace pcd
