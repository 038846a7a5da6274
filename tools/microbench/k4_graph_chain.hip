// Microbenchmark (round 6): what a chain of short DEPENDENT kernels costs per link on one stream -- launched one by one, and replayed as a
// hipGraph captured from the same launches.  The bucket reduction of an MSM is such a chain (14 pair levels of 15-30 us each, a dozen kernels
// of ~5 us); the question is whether a graph would shorten the device-side dispatch between links.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/k4_graph_chain.hip -o build/k4_graph_chain && build/k4_graph_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void link(unsigned* p, int spin) {
  unsigned v = p[threadIdx.x];
  for (int i = 0; i < spin; i++) v = v * 1664525u + 1013904223u;
  p[threadIdx.x] = v;
}
int main() {
  unsigned* d; CK(hipMalloc(&d, 64 * 4)); CK(hipMemset(d, 1, 64 * 4));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int LINKS = 30, REPS = 50;
  for (int spin : {0, 2000, 8000}) {   // ~1 us, ~5 us, ~20 us of dependent work per link
    std::vector<float> plain, graph;
    for (int w = 0; w < 3; w++) { for (int k = 0; k < LINKS; k++) hipLaunchKernelGGL(link, dim3(1), dim3(64), 0, st, d, spin); }
    CK(hipStreamSynchronize(st));
    for (int r = 0; r < REPS; r++) {
      CK(hipEventRecord(a, st));
      for (int k = 0; k < LINKS; k++) hipLaunchKernelGGL(link, dim3(1), dim3(64), 0, st, d, spin);
      CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); plain.push_back(ms);
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int k = 0; k < LINKS; k++) hipLaunchKernelGGL(link, dim3(1), dim3(64), 0, st, d, spin);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; w++) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    for (int r = 0; r < REPS; r++) {
      CK(hipEventRecord(a, st));
      CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); graph.push_back(ms);
    }
    std::sort(plain.begin(), plain.end()); std::sort(graph.begin(), graph.end());
    printf("spin %5d: %d dependent links  plain launches %.1f us (%.2f us a link)   graph replay %.1f us (%.2f us a link)\n", spin, LINKS,
           plain[REPS / 2] * 1e3, plain[REPS / 2] * 1e3 / LINKS, graph[REPS / 2] * 1e3, graph[REPS / 2] * 1e3 / LINKS);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
