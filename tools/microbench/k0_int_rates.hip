// K0 micro-benchmark: issue rates of the integer / fp64 instructions a wide-integer Montgomery
// multiplier can be built from, on gfx950.  Prints ops/s per instruction and per-CU-cycle cost.
// Build: hipcc --offload-arch=gfx950 -O3 k0_int_rates.hip -o ../../build/k0_int_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int KIND>
__global__ void __launch_bounds__(256) rate_kernel(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  uint64_t c0 = a, c1 = b, c2 = a + 1, c3 = b + 1, c4 = a + 2, c5 = b + 2, c6 = a + 3, c7 = b + 3;
  double d0 = a, d1 = b, d2 = 1.5, d3 = 2.5, d4 = 3.5, d5 = 4.5, d6 = 5.5, d7 = 6.5, da = 1.0000001, db = 1e-9;
  uint32_t x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3;
  for (int it = 0; it < iters; it++) {
    if (KIND == 0) {  // v_mad_u64_u32, 8 independent chains x 8
      REP8(asm volatile(
          "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n"
          "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
          "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n"
          "v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : "vcc");)
    } else if (KIND == 1) {  // v_mul_lo_u32
      REP8(asm volatile(
          "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
          "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));)
    } else if (KIND == 2) {  // v_mul_hi_u32
      REP8(asm volatile(
          "v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n"
          "v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));)
    } else if (KIND == 3) {  // v_mad_u32_u24
      REP8(asm volatile(
          "v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n"
          "v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
    } else if (KIND == 4) {  // v_mul_hi_u32_u24
      REP8(asm volatile(
          "v_mul_hi_u32_u24 %0, %0, %8\n v_mul_hi_u32_u24 %1, %1, %8\n v_mul_hi_u32_u24 %2, %2, %8\n v_mul_hi_u32_u24 %3, %3, %8\n"
          "v_mul_hi_u32_u24 %4, %4, %8\n v_mul_hi_u32_u24 %5, %5, %8\n v_mul_hi_u32_u24 %6, %6, %8\n v_mul_hi_u32_u24 %7, %7, %8\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));)
    } else if (KIND == 5) {  // v_add_co_u32 / v_addc_co_u32 pairs (count both)
      REP8(asm volatile(
          "v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_addc_co_u32 %2, vcc, %2, %8, vcc\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n"
          "v_addc_co_u32 %4, vcc, %4, %8, vcc\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n v_addc_co_u32 %6, vcc, %6, %8, vcc\n v_addc_co_u32 %7, vcc, %7, %8, vcc\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");)
    } else if (KIND == 6) {  // v_fma_f64
      REP8(asm volatile(
          "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
          "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
          : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(da), "v"(db));)
    } else if (KIND == 7) {  // v_lshl_add_u64
      REP8(asm volatile(
          "v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n"
          "v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8\n"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(c0 ^ 5));)
    } else if (KIND == 8) {  // v_add_u32 (full-rate reference)
      REP8(asm volatile(
          "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
          "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));)
    } else if (KIND == 9) {  // v_mul_u32_u24 (low 32 of 24x24)
      REP8(asm volatile(
          "v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n"
          "v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));)
    } else if (KIND == 10) {  // v_mad_u64_u32 with SGPR multiplier operand (modulus limb in SGPR)
      REP8(asm volatile(
          "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n"
          "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
          "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n"
          "v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "s"(seed) : "vcc");)
    } else if (KIND == 11) {  // v_mad_i32_i24 / packed? : v_pk_mul_lo_u16 as a cheap 16-bit multiplier
      REP8(asm volatile(
          "v_pk_mad_u16 %0, %0, %8, %9\n v_pk_mad_u16 %1, %1, %8, %9\n v_pk_mad_u16 %2, %2, %8, %9\n v_pk_mad_u16 %3, %3, %8, %9\n"
          "v_pk_mad_u16 %4, %4, %8, %9\n v_pk_mad_u16 %5, %5, %8, %9\n v_pk_mad_u16 %6, %6, %8, %9\n v_pk_mad_u16 %7, %7, %8, %9\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
    }
  }
  uint64_t s = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
  double ds = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
  uint32_t xs = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32) ^ (uint32_t)ds ^ xs;
}

template <int KIND>
int run(const char* name, uint32_t* dout, int waves_per_simd) {
  int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block = one per SIMD) x waves_per_simd
  int iters = 2000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, dout, 10, 1u);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, dout, iters, (uint32_t)r);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double lane_ops = (double)blocks * 256 * iters * 64.0;  // 64 instructions per iteration per lane
  double rate = lane_ops / (best * 1e-3);
  // cycles per wave-instruction per SIMD at 2.4 GHz: SIMDs = 1024
  double wave_instr = (double)blocks * 4 * iters * 64.0;
  double cyc = best * 1e-3 * 2.4e9 / (wave_instr / 1024.0);
  printf("%-22s waves/SIMD=%d  %8.3f ms  %10.3e lane-ops/s  %6.2f cyc/wave-instr/SIMD (at 2.4 GHz)\n", name, waves_per_simd, best, rate, cyc);
  return 0;
}

int main() {
  uint32_t* dout;
  CK(hipMalloc(&dout, 256 * 8 * 256 * 4 * 2));
  for (int w : {1, 2, 4}) {
    run<8>("v_add_u32", dout, w);
    run<0>("v_mad_u64_u32", dout, w);
    run<10>("v_mad_u64_u32(sgpr)", dout, w);
    run<1>("v_mul_lo_u32", dout, w);
    run<2>("v_mul_hi_u32", dout, w);
    run<3>("v_mad_u32_u24", dout, w);
    run<9>("v_mul_u32_u24", dout, w);
    run<4>("v_mul_hi_u32_u24", dout, w);
    run<5>("v_addc_co_u32", dout, w);
    run<7>("v_lshl_add_u64", dout, w);
    run<6>("v_fma_f64", dout, w);
    run<11>("v_pk_mad_u16", dout, w);
  }
  return 0;
}
