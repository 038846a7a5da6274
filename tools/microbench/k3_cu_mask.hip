// K3 micro-benchmark: what a CU mask on a stream buys the OTHER streams.
//  (1) which physical CUs (XCC, SE, CU) a queue created by hipExtStreamCreateWithCUMask runs on, for three ways of clearing 8 / 16
//      of the 256 mask bits -- the bit -> CU map decides which clearing leaves one CU free in EVERY XCD (an unmasked kernel's
//      workgroups are dealt to the XCDs round-robin, so each XCD needs a free CU of its own);
//  (2) the latency of a chain of 20 dependent one-wave launches on an unmasked high-priority stream while a register-heavy kernel with
//      long-lived workgroups fills the device from (a) an unmasked stream, (b) a masked stream -- the situation of a bucket-reduction
//      tail behind another MSM's accumulation waves (profiles/DESIGN_history_r01-r05.md section 7).
// Build: hipcc --offload-arch=gfx950 -O3 k3_cu_mask.hip -o ../../build/k3_cu_mask
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(64) where_kernel(uint32_t* out, int spin) {
  // HW_REG_HW_ID (4): CU_ID [11:8], SH_ID [12], SE_ID [15:13];  HW_REG_XCC_ID (20): [3:0]
  uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
  uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
  uint64_t t0 = wall_clock64();
  while (wall_clock64() - t0 < (uint64_t)spin) {}
  if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | (hw & 0xFFFFu);
}

// long-lived, register-heavy workgroups: 2 waves per SIMD at most (like msm_accumulate_kernel at 206 registers)
__global__ void __launch_bounds__(64, 2) hog_kernel(uint32_t* out, int iters) {
  uint32_t v[160];
#pragma unroll
  for (int i = 0; i < 160; i++) v[i] = threadIdx.x * 2654435761u + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 160; i++) v[i] = v[i] * v[(i + 7) % 160] + v[(i + 13) % 160];
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 160; i++) s ^= v[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(64) tiny_kernel(uint32_t* p) {
  uint32_t v[120];
#pragma unroll
  for (int i = 0; i < 120; i++) v[i] = p[0] + i;
  for (int it = 0; it < 40; it++) {
#pragma unroll
    for (int i = 0; i < 120; i++) v[i] = v[i] * v[(i + 7) % 120] + 1;
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 120; i++) s ^= v[i];
  p[0] = s | 1;
}

static void make_mask(uint32_t* m, int kind) {
  for (int i = 0; i < 8; i++) m[i] = 0xFFFFFFFFu;
  if (kind == 1) m[7] &= 0x00FFFFFFu;                                     // the last 8 bits (248..255)
  if (kind == 2) for (int i = 0; i < 8; i++) m[i] &= 0x7FFFFFFFu;         // bit 31 of every word (31, 63, ..)
  if (kind == 3) m[0] &= 0xFFFFFF00u;                                     // the first 8 bits
  if (kind == 4) m[7] &= 0x0000FFFFu;                                     // the last 16 bits
}

int main() {
  uint32_t* d;
  CK(hipMalloc(&d, 1 << 24));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  for (int kind = 0; kind <= 4; kind++) {
    uint32_t mask[8];
    make_mask(mask, kind);
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, 8, mask));
    const int n = 8192;
    hipLaunchKernelGGL(where_kernel, dim3(n), dim3(64), 0, s, d, 20000);
    CK(hipStreamSynchronize(s));
    std::vector<uint32_t> h(n);
    CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
    std::set<uint32_t> cus;
    int per_xcc[16] = {0};
    for (int i = 0; i < n; i++) cus.insert(((h[i] >> 16) << 16) | (h[i] & 0xFF00u));
    for (uint32_t c : cus) per_xcc[(c >> 16) & 15]++;
    printf("mask kind %d: %zu distinct CUs; per XCC:", kind, cus.size());
    for (int x = 0; x < 8; x++) printf(" %d", per_xcc[x]);
    printf("\n");
    if (kind == 0) {  // the (SE, SH, CU) ids present in XCC 0, for reference
      printf("  XCC 0 ids (se.sh.cu):");
      for (uint32_t c : cus) if (((c >> 16) & 15) == 0) printf(" %u.%u.%u", (c >> 13) & 7, (c >> 12) & 1, (c >> 8) & 15);
      printf("\n");
    }
    CK(hipStreamDestroy(s));
  }
  // (2) chain latency behind a hog
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t side;
  CK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, hi));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int kind = -1; kind <= 4; kind++) {
    if (kind == 3) continue;
    hipStream_t hs = nullptr;
    uint32_t mask[8];
    if (kind >= 0) { make_mask(mask, kind); CK(hipExtStreamCreateWithCUMask(&hs, 8, mask)); }
    for (int rep = 0; rep < 3; rep++) {
      // hog: 3 rounds of 2048 workgroups of ~0.5 ms each
      if (kind >= 0) hipLaunchKernelGGL(hog_kernel, dim3(2048 * 3), dim3(64), 0, hs, d + 4096, 1200);
      CK(hipEventRecord(e0, side));
      for (int k = 0; k < 20; k++) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, side, d);
      CK(hipEventRecord(e1, side));
      CK(hipEventSynchronize(e1));
      float chain_ms = 0;
      CK(hipEventElapsedTime(&chain_ms, e0, e1));
      float hog_ms = 0;
      if (kind >= 0) {
        CK(hipStreamSynchronize(hs));
        hipEvent_t h0, h1;
        CK(hipEventCreate(&h0)); CK(hipEventCreate(&h1));
        CK(hipEventRecord(h0, hs));
        hipLaunchKernelGGL(hog_kernel, dim3(2048 * 3), dim3(64), 0, hs, d + 4096, 1200);
        CK(hipEventRecord(h1, hs));
        CK(hipEventSynchronize(h1));
        CK(hipEventElapsedTime(&hog_ms, h0, h1));
      }
      printf("hog mask kind %2d: chain of 20 one-wave launches %.3f ms (hog alone %.3f ms)\n", kind, chain_ms, hog_ms);
    }
    if (hs) CK(hipStreamDestroy(hs));
  }
  return 0;
}
