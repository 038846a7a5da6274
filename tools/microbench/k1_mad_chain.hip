// K1 micro-benchmark: v_mad_u64_u32 throughput as a function of the number of INDEPENDENT accumulator chains per lane (1, 2, 4, 8)
// and of waves per SIMD (1, 2, 4) on gfx950.  A Montgomery product in product-scanning form is ONE chain (acc += a_i * b_j over a
// whole column, columns linked by the carry); this tells how much splitting a column over several accumulators can buy.
// Build: hipcc --offload-arch=gfx950 -O3 k1_mad_chain.hip -o ../../build/k1_mad_chain
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP8(X) X X X X X X X X

template <int CHAINS>
__global__ void __launch_bounds__(256) chain_kernel(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  uint64_t c0 = a, c1 = b, c2 = a + 1, c3 = b + 1, c4 = a + 2, c5 = b + 2, c6 = a + 3, c7 = b + 3;
  for (int it = 0; it < iters; it++) {
    if (CHAINS == 1) {
      REP8(asm volatile(
          "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
          "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
          : "+v"(c0) : "v"(a), "v"(b) : "vcc");)
    } else if (CHAINS == 2) {
      REP8(asm volatile(
          "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n"
          "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n"
          : "+v"(c0), "+v"(c1) : "v"(a), "v"(b) : "vcc");)
    } else if (CHAINS == 4) {
      REP8(asm volatile(
          "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
          "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "vcc");)
    } else {
      REP8(asm volatile(
          "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
          "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : "vcc");)
    }
  }
  uint64_t s = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

template <int CHAINS>
int run(uint32_t* dout, int waves_per_simd) {
  int blocks = 256 * waves_per_simd, iters = 2000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(chain_kernel<CHAINS>, dim3(blocks), dim3(256), 0, 0, dout, 10, 1u);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(chain_kernel<CHAINS>, dim3(blocks), dim3(256), 0, 0, dout, iters, (uint32_t)r);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double lane_ops = (double)blocks * 256 * iters * 64.0;
  printf("v_mad_u64_u32 chains=%d waves/SIMD=%d  %8.3f ms  %10.3e lane-mads/s\n", CHAINS, waves_per_simd, best, lane_ops / (best * 1e-3));
  return 0;
}

int main() {
  uint32_t* dout;
  CK(hipMalloc(&dout, 256 * 8 * 256 * 4 * 2));
  for (int w : {1, 2, 4}) { if (run<1>(dout, w) || run<2>(dout, w) || run<4>(dout, w) || run<8>(dout, w)) return 1; }
  return 0;
}
