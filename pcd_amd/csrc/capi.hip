// extern "C" surface of libpcdhip.so (include/pcdhip.h).  Host orchestration only: every arithmetic
// step is a HIP kernel from msm.hip.h / fft.hip.h / inst_*.hip.  There is no CPU fallback anywhere in this
// library -- if no GPU is usable, pcdhip_init fails with PCDHIP_E_NO_DEVICE and nothing else can be called.
#include <string.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <new>

#include "common.h"

namespace pcd {

#define PCD_DECL_G(k) const GroupEntry* pcd_group_entry_##k();
PCD_DECL_G(0) PCD_DECL_G(1) PCD_DECL_G(2) PCD_DECL_G(3) PCD_DECL_G(4) PCD_DECL_G(5) PCD_DECL_G(6) PCD_DECL_G(7)
#define PCD_DECL_F(k) const FieldEntry* pcd_field_entry_##k();
PCD_DECL_F(0) PCD_DECL_F(1) PCD_DECL_F(2) PCD_DECL_F(3)
#define PCD_DECL_C(k) const CurveEntry* pcd_curve_entry_##k();
PCD_DECL_C(0) PCD_DECL_C(1) PCD_DECL_C(2) PCD_DECL_C(3)
#define PCD_DECL_P(k) const PairingEntry* pcd_pairing_entry_##k();
PCD_DECL_P(0) PCD_DECL_P(1) PCD_DECL_P(2) PCD_DECL_P(3)

const GroupEntry& group_entry(int curve_id, int group_id) {
  typedef const GroupEntry* (*Fn)();
  static const Fn tab[8] = {pcd_group_entry_0, pcd_group_entry_1, pcd_group_entry_2, pcd_group_entry_3,
                            pcd_group_entry_4, pcd_group_entry_5, pcd_group_entry_6, pcd_group_entry_7};
  return *tab[curve_id * 2 + (group_id - 1)]();
}
const FieldEntry& field_entry(int field_id) {
  typedef const FieldEntry* (*Fn)();
  static const Fn tab[4] = {pcd_field_entry_0, pcd_field_entry_1, pcd_field_entry_2, pcd_field_entry_3};
  return *tab[field_id]();
}
const CurveEntry& curve_entry(int curve_id) {
  typedef const CurveEntry* (*Fn)();
  static const Fn tab[4] = {pcd_curve_entry_0, pcd_curve_entry_1, pcd_curve_entry_2, pcd_curve_entry_3};
  return *tab[curve_id]();
}

typedef hipError_t (*G1ScaleFn)(hipStream_t, const uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*);
G1ScaleFn pcd_g1_scale_entry_0(); G1ScaleFn pcd_g1_scale_entry_1(); G1ScaleFn pcd_g1_scale_entry_2(); G1ScaleFn pcd_g1_scale_entry_3();
G1ScaleFn g1_scale_entry(int curve_id) {
  typedef G1ScaleFn (*Fn)();
  static const Fn tab[4] = {pcd_g1_scale_entry_0, pcd_g1_scale_entry_1, pcd_g1_scale_entry_2, pcd_g1_scale_entry_3};
  return tab[curve_id]();
}

const PairingEntry& pairing_entry(int curve_id) {
  typedef const PairingEntry* (*Fn)();
  static const Fn tab[4] = {pcd_pairing_entry_0, pcd_pairing_entry_1, pcd_pairing_entry_2, pcd_pairing_entry_3};
  return *tab[curve_id]();
}

}  // namespace pcd

using namespace pcd;

namespace {

const int kFieldLimbs[4] = {5, 5, 12, 12};
const int kCurveFq[4] = {0, 1, 2, 3};
const int kCurveFr[4] = {1, 0, 3, 2};
const int kCurveG2Deg[4] = {2, 3, 2, 3};

bool valid_curve(int c) { return c >= 0 && c < 4; }
bool valid_field(int f) { return f >= 0 && f < 4; }
bool valid_group(int g) { return g == 1 || g == 2; }

// nothing may throw across the C ABI: bodies that use std containers run inside guarded()
template <class Fn>
int guarded(Fn&& body) {
  try { return body(); }
  catch (const std::bad_alloc&) { return PCDHIP_E_OOM; }
  catch (...) { return PCDHIP_E_HIP; }
}

int fail(pcdhip_ctx* ctx, hipError_t e) {
  if (ctx) { try { ctx->last_hip_error = hipGetErrorString(e); } catch (...) {} }
  if (e == hipErrorOutOfMemory) return PCDHIP_E_OOM;
  if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return PCDHIP_E_NO_DEVICE;
  if (e == hipErrorInvalidValue) return PCDHIP_E_ARG;
  return PCDHIP_E_HIP;
}
#define TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(ctx, e_); } while (0)
#define BIND() do { hipError_t e_ = hipSetDevice(ctx->device); if (e_ != hipSuccess) return fail(ctx, e_); } while (0)

enum { AUX_FFT_X = 0, AUX_FFT_TMP, AUX_A, AUX_B, AUX_C, AUX_Z, AUX_CSR_RP, AUX_CSR_COL, AUX_CSR_COEF, AUX_SCAL, AUX_OUT,
       AUX_G16, AUX_Z_CANON, AUX_H_CANON, AUX_MISC, AUX_FB_TABLE, AUX_FB_JAC, AUX_FB_OUT };

// Evaluation domain as ark-poly `GeneralEvaluationDomain::new(min_size)` picks it: radix-2 when 2^ceil(log2 min_size)
// fits the field's 2-adicity, otherwise the mixed-radix size 2^a q^b (b <= 2) of `best_mixed_domain_size`.
struct Dom { uint32_t n, m; int a; };
const uint32_t kSmallSubgroupBase[4] = {7, 0, 5, 0};  // ark-ff SMALL_SUBGROUP_BASE (adicity 2) of the two help fields

int split_domain(int field_id, size_t n, Dom* d) {  // n = m * 2^a with m in {1, q, q^2}?
  if (n == 0 || n >= (1ull << 31)) return PCDHIP_E_ARG;
  const uint32_t q = kSmallSubgroupBase[field_id];
  uint32_t m = 1;
  size_t r = n;
  for (int b = 0; b < 2 && q && r % q == 0; b++) { r /= q; m *= q; }
  if (r & (r - 1)) return PCDHIP_E_SIZE_UNSUPPORTED;
  int a = 0;
  while (((size_t)1 << a) < r) a++;
  if (a > field_entry(field_id).two_adicity) return PCDHIP_E_SIZE_UNSUPPORTED;
  d->n = (uint32_t)n; d->m = m; d->a = a;
  return PCDHIP_OK;
}
int pick_domain(int field_id, size_t min_size, Dom* d) {
  const FieldEntry& fe = field_entry(field_id);
  int log_n = 0;
  while (((size_t)1 << log_n) < min_size) log_n++;
  if (log_n <= fe.two_adicity) { d->n = 1u << log_n; d->m = 1; d->a = log_n; return PCDHIP_OK; }
  const uint32_t q = kSmallSubgroupBase[field_id];
  if (!q) return PCDHIP_E_SIZE_UNSUPPORTED;
  size_t best = 0;
  for (int b = 0; b <= 2; b++) {
    size_t r = 1;
    for (int i = 0; i < b; i++) r *= q;
    const size_t mm = r;
    int aa = 0;
    while (r < min_size) { r *= 2; aa++; }
    if (aa <= fe.two_adicity && r < (1ull << 31) && (best == 0 || r < best)) { best = r; d->n = (uint32_t)r; d->m = (uint32_t)mm; d->a = aa; }
  }
  return best ? PCDHIP_OK : PCDHIP_E_SIZE_UNSUPPORTED;
}

// affine points flagged as the point at infinity become literal zeros (0, 0) -- the device encoding of the identity
// (not on any of the eight curves: b != 0); `flags` = one byte per point, already on the device
__global__ void __launch_bounds__(256) zero_flagged_kernel(uint32_t* __restrict__ pts, const uint8_t* __restrict__ flags, size_t n_words,
                                                            uint32_t words_per_point) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_words && flags[i / words_per_point]) pts[i] = 0;
}
hipError_t zero_flagged(hipStream_t st, uint32_t* pts_dev, uint8_t* flags_dev, const uint8_t* flags_host, size_t n, size_t point_bytes) {
  if (!flags_host || !n) return hipSuccess;
  hipError_t e = hipMemcpyAsync(flags_dev, flags_host, n, hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return e;
  const size_t nw = n * (point_bytes / 4);
  hipLaunchKernelGGL(zero_flagged_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, pts_dev, flags_dev, nw, (uint32_t)(point_bytes / 4));
  return hipGetLastError();
}

int get_tables(pcdhip_ctx* ctx, int field_id, int log_n, const FftTables** out);
int get_mixed_tables(pcdhip_ctx* ctx, int field_id, const Dom& d, const FftTables** out) {
  uint64_t key = (1ull << 63) | ((uint64_t)field_id << 32) | d.n;
  auto it = ctx->fft_tables.find(key);
  if (it == ctx->fft_tables.end()) {
    FftTables t;
    TRY(field_entry(field_id).mixed_make_tables(ctx->stream, d.n, d.m, &t));
    try { it = ctx->fft_tables.emplace(key, t).first; } catch (...) { return PCDHIP_E_OOM; }
  }
  *out = &it->second;
  return PCDHIP_OK;
}
// v (d.n elements, device image) transformed in place
int domain_transform(pcdhip_ctx* ctx, int field_id, const Dom& d, uint32_t* v, uint32_t* tmp, int inverse, int coset, float* pass_ms,
                     int* npasses) {
  const FieldEntry& fe = field_entry(field_id);
  const FftTables* t2;
  int rc = get_tables(ctx, field_id, d.a, &t2);
  if (rc) return rc;
  if (d.m == 1) { TRY(fe.fft_run(ctx->stream, *t2, v, tmp, d.a, inverse, coset, pass_ms, npasses)); return PCDHIP_OK; }
  const FftTables* t;
  rc = get_mixed_tables(ctx, field_id, d, &t);
  if (rc) return rc;
  TRY(fe.mixed_run(ctx->stream, *t, *t2, v, tmp, d.m, d.a, inverse, coset));
  if (npasses) *npasses = 0;
  return PCDHIP_OK;
}

int get_tables(pcdhip_ctx* ctx, int field_id, int log_n, const FftTables** out) {
  uint64_t key = ((uint64_t)field_id << 32) | (uint32_t)log_n;
  auto it = ctx->fft_tables.find(key);
  if (it == ctx->fft_tables.end()) {
    FftTables t;
    TRY(field_entry(field_id).fft_make_tables(ctx->stream, log_n, &t));
    try { it = ctx->fft_tables.emplace(key, t).first; } catch (...) { return PCDHIP_E_OOM; }
  }
  *out = &it->second;
  return PCDHIP_OK;
}

}  // namespace

extern "C" {

const char* pcdhip_strerror(int code) {
  switch (code) {
    case PCDHIP_OK: return "ok";
    case PCDHIP_E_ARG: return "invalid argument";
    case PCDHIP_E_SIZE_UNSUPPORTED: return "size not supported by this build (e.g. log_n above the field's 2-adicity)";
    case PCDHIP_E_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case PCDHIP_E_OOM: return "out of device memory";
    case PCDHIP_E_HIP: return "HIP runtime error (see pcdhip_last_hip_error)";
    case PCDHIP_E_PREV_TICKET: return "the previous ticket of this slot had an unreduced scalar (this submission was enqueued)";
    default: return "unknown error";
  }
}

int pcdhip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int pcdhip_field_limbs(int field_id) { return valid_field(field_id) ? kFieldLimbs[field_id] : PCDHIP_E_ARG; }
int pcdhip_curve_base_field(int curve_id) { return valid_curve(curve_id) ? kCurveFq[curve_id] : PCDHIP_E_ARG; }
int pcdhip_curve_scalar_field(int curve_id) { return valid_curve(curve_id) ? kCurveFr[curve_id] : PCDHIP_E_ARG; }
int pcdhip_point_limbs(int curve_id, int group_id) {
  if (!valid_curve(curve_id) || !valid_group(group_id)) return PCDHIP_E_ARG;
  int deg = group_id == 1 ? 1 : kCurveG2Deg[curve_id];
  return 2 * deg * kFieldLimbs[kCurveFq[curve_id]];
}

int pcdhip_init(int device_id, pcdhip_ctx** out) {
  if (!out) return PCDHIP_E_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return PCDHIP_E_NO_DEVICE;
  if (device_id < 0 || device_id >= n) return PCDHIP_E_ARG;
  if (hipSetDevice(device_id) != hipSuccess) return PCDHIP_E_NO_DEVICE;
  pcdhip_ctx* ctx = new (std::nothrow) pcdhip_ctx();
  if (!ctx) return PCDHIP_E_OOM;
  ctx->device = device_id;
  // the context's own stream gets the highest priority: inside a proof it carries the witness map, whose small kernels
  // must not queue behind the MSMs of the other streams (the h MSM waits for it)
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  if (hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, greatest) != hipSuccess ||
      hipEventCreate(&ctx->t0) != hipSuccess || hipEventCreate(&ctx->t1) != hipSuccess) {
    delete ctx;
    return PCDHIP_E_HIP;
  }
  *out = ctx;
  return PCDHIP_OK;
}

int pcdhip_init_devices(const int* device_ids, int n_dev, pcdhip_ctx** out) {
  if (!device_ids || n_dev < 1 || n_dev > 64 || !out) return PCDHIP_E_ARG;
  int rc = pcdhip_init(device_ids[0], out);
  if (rc || n_dev == 1) return rc;
  pcdhip_ctx* ctx = *out;
  *out = nullptr;
  rc = guarded([&]() -> int {
    ctx->peers.push_back(ctx);
    for (int i = 1; i < n_dev; i++) {
      pcdhip_ctx* p = nullptr;
      int r = pcdhip_init(device_ids[i], &p);
      if (r) return r;
      ctx->peers.push_back(p);
    }
    // direct xGMI copies between the devices of the context where the platform allows them (the partial results and the
    // slices of h travel device to device; without peer access the runtime stages them through the host)
    for (pcdhip_ctx* a : ctx->peers)
      for (pcdhip_ctx* b : ctx->peers) {
        if (a->device == b->device) continue;
        int can = 0;
        if (hipSetDevice(a->device) != hipSuccess || hipDeviceCanAccessPeer(&can, a->device, b->device) != hipSuccess || !can) continue;
        if (hipDeviceEnablePeerAccess(b->device, 0) != hipSuccess) (void)hipGetLastError();  // (already enabled: fine)
      }
    return PCDHIP_OK;
  });
  if (rc) { pcdhip_destroy(ctx); return rc; }
  *out = ctx;
  return PCDHIP_OK;
}
int pcdhip_ctx_devices(const pcdhip_ctx* ctx) { return !ctx ? PCDHIP_E_ARG : ctx->peers.empty() ? 1 : (int)ctx->peers.size(); }

void pcdhip_destroy(pcdhip_ctx* ctx) {
  if (!ctx) return;
  for (size_t g = 1; g < ctx->peers.size(); g++) pcdhip_destroy(ctx->peers[g]);
  ctx->peers.clear();
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ctx->msm_ws.release();
  ctx->aux_ws.release();
  for (int k = 0; k < 6; k++) {
    ctx->g16_ws[k].release();
    if (ctx->g16_streams[k]) (void)hipStreamDestroy(ctx->g16_streams[k]);
    if (ctx->g16_begin[k]) (void)hipEventDestroy(ctx->g16_begin[k]);
    if (ctx->g16_end[k]) (void)hipEventDestroy(ctx->g16_end[k]);
  }
  if (ctx->lane_stream) (void)hipStreamDestroy(ctx->lane_stream);
  if (ctx->g16_ready) (void)hipEventDestroy(ctx->g16_ready);
  if (ctx->g16_share.ready) (void)hipEventDestroy(ctx->g16_share.ready);
  if (ctx->g16_share_b.ready) (void)hipEventDestroy(ctx->g16_share_b.ready);
  for (auto& kv : ctx->fft_tables) {
    (void)hipFree(kv.second.tw_fwd); (void)hipFree(kv.second.tw_inv);
    (void)hipFree(kv.second.coset); (void)hipFree(kv.second.coset_inv_scaled);
    (void)hipFree(kv.second.tw0_fwd); (void)hipFree(kv.second.tw0_inv);
  }
  if (ctx->xstream_ev) (void)hipEventDestroy(ctx->xstream_ev);
  if (ctx->wm_ev) (void)hipEventDestroy(ctx->wm_ev);
  for (int c = 0; c < 4; c++) if (ctx->vm_block[c]) (void)hipFree(ctx->vm_block[c]);
  for (int k = 0; k < pcdhip_ctx::PIPE_SLOTS; k++) if (ctx->pipe_done[k]) (void)hipEventDestroy(ctx->pipe_done[k]);
  if (ctx->pipe_host) (void)hipHostFree(ctx->pipe_host);
  if (ctx->count_host) (void)hipHostFree(ctx->count_host);
  if (ctx->count_ev) (void)hipEventDestroy(ctx->count_ev);
  if (ctx->t0) (void)hipEventDestroy(ctx->t0);
  if (ctx->t1) (void)hipEventDestroy(ctx->t1);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int pcdhip_host_alloc(size_t bytes, void** out) {
  if (!out) return PCDHIP_E_ARG;
  *out = nullptr;
  hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
  if (e == hipErrorOutOfMemory) return PCDHIP_E_OOM;
  return e == hipSuccess ? PCDHIP_OK : PCDHIP_E_HIP;
}
void pcdhip_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}
int pcdhip_sync(pcdhip_ctx* ctx) {
  if (!ctx) return PCDHIP_E_ARG;
  BIND();
  TRY(hipStreamSynchronize(ctx->stream));
  return PCDHIP_OK;
}
const char* pcdhip_last_hip_error(pcdhip_ctx* ctx) { return ctx ? ctx->last_hip_error.c_str() : ""; }

int pcdhip_timer_start(pcdhip_ctx* ctx) {
  if (!ctx) return PCDHIP_E_ARG;
  BIND();
  TRY(hipEventRecord(ctx->t0, ctx->stream));
  return PCDHIP_OK;
}
int pcdhip_timer_stop(pcdhip_ctx* ctx, float* out_ms) {
  if (!ctx || !out_ms) return PCDHIP_E_ARG;
  BIND();
  TRY(hipEventRecord(ctx->t1, ctx->stream));
  TRY(hipEventSynchronize(ctx->t1));
  TRY(hipEventElapsedTime(out_ms, ctx->t0, ctx->t1));
  return PCDHIP_OK;
}

// ------------------------------------------------------------------------------------------------ buffers
int pcdhip_buf_alloc(pcdhip_ctx* ctx, int field_id, size_t n, pcdhip_buf** out) {
  if (!ctx || !out || !valid_field(field_id)) return PCDHIP_E_ARG;
  BIND();
  pcdhip_buf* b = new (std::nothrow) pcdhip_buf();
  if (!b) return PCDHIP_E_OOM;
  b->field_id = field_id;
  b->n = n;
  b->dptr = nullptr;
  hipError_t e = hipMalloc(&b->dptr, std::max<size_t>(n, 1) * kFieldLimbs[field_id] * 8);
  if (e != hipSuccess) { delete b; return fail(ctx, e); }
  *out = b;
  return PCDHIP_OK;
}
int pcdhip_buf_upload(pcdhip_ctx* ctx, int field_id, const uint64_t* host, size_t n, pcdhip_buf** out) {
  if (!host && n) return PCDHIP_E_ARG;
  int rc = pcdhip_buf_alloc(ctx, field_id, n, out);
  if (rc) return rc;
  hipError_t e = hipMemcpyAsync((*out)->dptr, host, n * kFieldLimbs[field_id] * 8, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { pcdhip_buf_free(ctx, *out); *out = nullptr; return fail(ctx, e); }
  return PCDHIP_OK;
}
int pcdhip_buf_download(pcdhip_ctx* ctx, const pcdhip_buf* buf, uint64_t* host, size_t n) {
  if (!ctx || !buf || !host || n > buf->n) return PCDHIP_E_ARG;
  BIND();
  TRY(hipMemcpyAsync(host, buf->dptr, n * kFieldLimbs[buf->field_id] * 8, hipMemcpyDeviceToHost, ctx->stream));
  TRY(hipStreamSynchronize(ctx->stream));
  return PCDHIP_OK;
}
void pcdhip_buf_free(pcdhip_ctx* ctx, pcdhip_buf* buf) {
  if (!buf) return;
  if (ctx) (void)hipSetDevice(ctx->device);
  (void)hipFree(buf->dptr);
  delete buf;
}

// ------------------------------------------------------------------------------------------------ MSM
// contiguous range [lo, hi) of part g out of `parts` (sizes differ by at most one) -- the point-range sharding of SURVEY.md 8e
static void shard_range(size_t n, size_t g, size_t parts, size_t* lo, size_t* hi) {
  const size_t base = n / parts, rem = n % parts;
  *lo = g * base + std::min(g, rem);
  *hi = *lo + base + (g < rem ? 1 : 0);
}
static int bases_upload_single(pcdhip_ctx* ctx, int curve_id, int group_id, const uint64_t* xy, const uint8_t* inf, size_t n, pcdhip_bases** out);

int pcdhip_bases_upload(pcdhip_ctx* ctx, int curve_id, int group_id, const uint64_t* xy, const uint8_t* inf, size_t n,
                        pcdhip_bases** out) {
  if (!ctx || !out || !valid_curve(curve_id) || !valid_group(group_id) || (!xy && n) || n >= (1ull << 31)) return PCDHIP_E_ARG;
  if (ctx->peers.size() <= 1) return bases_upload_single(ctx, curve_id, group_id, xy, inf, n, out);
  // multi-device context: one ordinary handle per device, each over its point range
  *out = nullptr;
  return guarded([&]() -> int {
    pcdhip_bases* b = new pcdhip_bases();
    b->curve_id = curve_id; b->group_id = group_id; b->n = n; b->dptr = nullptr; b->c = 0; b->groups = 1;
    const size_t G = ctx->peers.size(), pl = (size_t)pcdhip_point_limbs(curve_id, group_id);
    b->shard_lo.resize(G + 1);
    for (size_t g = 0; g < G; g++) {
      size_t lo, hi;
      shard_range(n, g, G, &lo, &hi);
      b->shard_lo[g] = lo; b->shard_lo[g + 1] = hi;
      pcdhip_ctx* C = ctx->peers[g];
      C->precompute = ctx->precompute; C->precompute_budget = ctx->precompute_budget; C->msm_c = ctx->msm_c;
      pcdhip_bases* sh = nullptr;
      int rc = bases_upload_single(C, curve_id, group_id, xy + lo * pl, inf ? inf + lo : nullptr, hi - lo, &sh);
      if (rc) { pcdhip_bases_free(ctx, b); return rc; }
      b->shards.push_back(sh);
    }
    *out = b;
    return PCDHIP_OK;
  });
}
static int bases_upload_single(pcdhip_ctx* ctx, int curve_id, int group_id, const uint64_t* xy, const uint8_t* inf, size_t n, pcdhip_bases** out) {
  BIND();
  const GroupEntry& ge = group_entry(curve_id, group_id);
  const size_t abi_b = (size_t)ge.point_abi_words * 4, int_b = (size_t)ge.base_stride_words * 4;
  pcdhip_bases* b = new (std::nothrow) pcdhip_bases();
  if (!b) return PCDHIP_E_OOM;
  b->curve_id = curve_id; b->group_id = group_id; b->n = n; b->dptr = nullptr; b->c = 0; b->groups = 1;
  // Precomputed window-shifted copies (HBM capacity traded for the serial window combine): group g
  // holds 2^(c Wg g) P_i.  Full precomputation (Wg = 1) needs W copies of the query vector.
  if (ctx->precompute != 0 && n >= 64) {
    const bool full = ctx->precompute < 0;
    const int c = ctx->msm_c ? ctx->msm_c : std::max(6, msm_pick_window(n, ge.scalar_bits, full ? 1 : 0) + ctx->msm_c_bias);
    const int W = msm_num_windows(ge.scalar_bits, c);
    b->c = c;
    b->groups = full ? W : std::min(W, ctx->precompute);
  }
  hipError_t e = hipErrorOutOfMemory;
  while (true) {
    const size_t want = std::max<size_t>(n, 1) * int_b * b->groups;
    // (the caller's budget for one vector counts like the device running out: a host that keeps several keys resident sets it)
    e = (ctx->precompute_budget && want > ctx->precompute_budget && b->groups > 1) ? hipErrorOutOfMemory : hipMalloc(&b->dptr, want);
    if (e == hipSuccess || b->groups == 1) break;
    (void)hipGetLastError();
    b->groups = (b->groups + 1) / 2;  // not enough HBM for this many copies: fewer groups, more bucket windows
  }
  if (e != hipSuccess) { (void)hipGetLastError(); delete b; return fail(ctx, e); }   // (the sticky error must not surface at a later launch's hipGetLastError)
  if (b->groups == 1) b->c = 0;
  // stage the C-ABI image, rewrite flagged points to (0, 0) (not on any of the curves: b != 0), convert
  e = ctx->aux_ws.ensure(AUX_MISC, std::max<size_t>(n, 1) * (abi_b + 1) + 64);
  char* stage = (char*)ctx->aux_ws.buf[AUX_MISC];
  if (e == hipSuccess) e = hipMemcpyAsync(stage, xy, n * abi_b, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = zero_flagged(ctx->stream, (uint32_t*)stage, (uint8_t*)stage + n * abi_b, inf, n, abi_b);
  if (e == hipSuccess) e = ge.points_in(ctx->stream, (const uint32_t*)stage, (uint32_t)n, b->dptr);
  if (e == hipSuccess && b->groups > 1) {
    const int W = msm_num_windows(ge.scalar_bits, b->c);
    const int Wg = (W + b->groups - 1) / b->groups;
    e = ge.precompute(ctx->stream, b->dptr, (uint32_t)n, b->groups, b->c * Wg);
  }
  // the points at infinity as a bitmap: their entries never enter an MSM's bucket lists (msm_base_is_inf)
  if (e == hipSuccess && inf) {
    std::vector<uint32_t> bits((n + 31) / 32 + 1, 0u);
    bool any = false;
    for (size_t i = 0; i < n; i++) if (inf[i]) { bits[i >> 5] |= 1u << (i & 31); any = true; }
    if (any) {
      e = hipMalloc((void**)&b->inf_bits, bits.size() * 4);
      if (e == hipSuccess) e = hipMemcpyAsync(b->inf_bits, bits.data(), bits.size() * 4, hipMemcpyHostToDevice, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // (`bits` is a local)
    }
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { (void)hipFree(b->dptr); (void)hipFree(b->inf_bits); delete b; return fail(ctx, e); }
  *out = b;
  return PCDHIP_OK;
}
int pcdhip_set_precompute(pcdhip_ctx* ctx, int mode) {
  if (!ctx || mode < -1 || mode == 1) return PCDHIP_E_ARG;
  ctx->precompute = mode;
  return PCDHIP_OK;
}
int pcdhip_set_precompute_budget(pcdhip_ctx* ctx, size_t bytes_per_vector) {
  if (!ctx) return PCDHIP_E_ARG;
  ctx->precompute_budget = bytes_per_vector;
  return PCDHIP_OK;
}
int pcdhip_get_precompute_budget(pcdhip_ctx* ctx, size_t* bytes_per_vector) {
  if (!ctx || !bytes_per_vector) return PCDHIP_E_ARG;
  *bytes_per_vector = ctx->precompute_budget;
  return PCDHIP_OK;
}
void pcdhip_bases_free(pcdhip_ctx* ctx, pcdhip_bases* bases) {
  if (!bases) return;
  for (size_t g = 0; g < bases->shards.size(); g++)
    pcdhip_bases_free(ctx && g < ctx->peers.size() ? ctx->peers[g] : ctx, bases->shards[g]);
  bases->shards.clear();
  if (ctx) (void)hipSetDevice(ctx->device);
  (void)hipFree(bases->dptr);
  (void)hipFree(bases->inf_bits);
  delete bases;
}
int pcdhip_bases_info(const pcdhip_bases* bases, size_t n, int* window_bits, int* windows, int* copies) {
  if (!bases || !window_bits || !windows || !copies) return PCDHIP_E_ARG;
  if (!bases->shards.empty()) return pcdhip_bases_info(bases->shards[0], 0, window_bits, windows, copies);  // (the plan of device 0's shard)
  const GroupEntry& ge = group_entry(bases->curve_id, bases->group_id);
  const int c = bases->groups > 1 ? bases->c : msm_pick_window(n ? n : bases->n, ge.scalar_bits, 0);
  *window_bits = c;
  *windows = msm_num_windows(ge.scalar_bits, c);
  *copies = bases->groups;
  return PCDHIP_OK;
}
int pcdhip_g16_pk_info(const pcdhip_g16_pk* pk, int window_bits[5], int windows[5]) {
  if (!pk || !window_bits || !windows) return PCDHIP_E_ARG;
  if (!pk->shards.empty()) return pcdhip_g16_pk_info(pk->shards[0], window_bits, windows);  // (the plan of device 0's shard)
  const pcdhip_bases* q[5] = {pk->a_query, pk->b_g1_query, pk->b_g2_query, pk->l_query, pk->h_query};
  for (int i = 0; i < 5; i++) {
    int copies = 0;
    int rc = q[i] ? pcdhip_bases_info(q[i], 0, &window_bits[i], &windows[i], &copies) : PCDHIP_E_ARG;
    if (rc) return rc;
  }
  return PCDHIP_OK;
}
// bytes of device memory a key's base vectors hold: out[0] the five queries with their window-shifted copies (every shard of a multi-device
// key), out[1] the second layout of the assignment queries for a shorter window (0 when not built), out[2] / out[3] the copies per point of the
// a query in the two layouts (device 0's shard)
int pcdhip_g16_pk_memory(const pcdhip_g16_pk* pk, uint64_t out[4]) {
  if (!pk || !out) return PCDHIP_E_ARG;
  out[0] = out[1] = out[2] = out[3] = 0;
  auto bytes = [](const pcdhip_bases* b) -> uint64_t {
    if (!b || !b->dptr) return 0;
    return (uint64_t)std::max<size_t>(b->n, 1) * group_entry(b->curve_id, b->group_id).base_stride_words * 4 * (uint64_t)b->groups + (b->inf_bits ? (b->n + 7) / 8 : 0);
  };
  std::vector<const pcdhip_g16_pk*> parts;
  if (pk->shards.empty()) parts.push_back(pk); else for (const pcdhip_g16_pk* s : pk->shards) parts.push_back(s);
  for (const pcdhip_g16_pk* s : parts) {
    out[0] += bytes(s->a_query) + bytes(s->b_g1_query) + bytes(s->b_g2_query) + bytes(s->l_query) + bytes(s->h_query);
    out[1] += bytes(s->a_sparse) + bytes(s->b_g1_sparse) + bytes(s->b_g2_sparse) + bytes(s->l_sparse);
  }
  out[2] = parts[0]->a_query ? (uint64_t)parts[0]->a_query->groups : 0;
  out[3] = parts[0]->a_sparse ? (uint64_t)parts[0]->a_sparse->groups : 0;
  return PCDHIP_OK;
}
int pcdhip_stream_wait(pcdhip_ctx* ctx, void* other_stream, int direction) {
  if (!ctx || direction < 0 || direction > 1) return PCDHIP_E_ARG;
  BIND();
  if (!ctx->xstream_ev) TRY(hipEventCreateWithFlags(&ctx->xstream_ev, hipEventDisableTiming));
  hipStream_t other = (hipStream_t)other_stream;
  TRY(hipEventRecord(ctx->xstream_ev, direction == 0 ? other : ctx->stream));
  TRY(hipStreamWaitEvent(direction == 0 ? ctx->stream : other, ctx->xstream_ev, 0));
  return PCDHIP_OK;
}
int pcdhip_msm_config(pcdhip_ctx* ctx, int window_bits, int chunk) {
  if (!ctx || window_bits < 0 || window_bits > 24 || chunk < 0) return PCDHIP_E_ARG;
  ctx->msm_c = window_bits;
  ctx->msm_chunk = (uint32_t)chunk;
  return PCDHIP_OK;
}
int pcdhip_msm_set_sort(pcdhip_ctx* ctx, int mode) {
  if (!ctx || mode < 0 || mode > 2) return PCDHIP_E_ARG;
  ctx->msm_sort = (ctx->msm_sort & ~15) | mode;  // (the higher bits carry pcdhip_msm_set_accumulate's choice: msm.hip.h msm_run)
  return PCDHIP_OK;
}
int pcdhip_msm_set_accumulate(pcdhip_ctx* ctx, int mode, int chunk, int min_pairs) {
  if (!ctx || mode < 0 || mode > 2 || chunk < 0 || chunk > 1024 || (chunk && chunk < 2) || min_pairs < 0 || min_pairs > 255) return PCDHIP_E_ARG;
  ctx->msm_sort = (ctx->msm_sort & 15) | (mode << 4) | (chunk << 8) | (min_pairs << 20);
  return PCDHIP_OK;
}
int pcdhip_groth16_set_assembly(pcdhip_ctx* ctx, int mode) {
  if (!ctx || mode < 0 || mode > 2) return PCDHIP_E_ARG;
  ctx->g16_assembly = mode;
  return PCDHIP_OK;
}
int pcdhip_groth16_set_sparse_window(pcdhip_ctx* ctx, int bits) {
  if (!ctx || bits < -1 || (bits > 0 && (bits < 6 || bits > 22))) return PCDHIP_E_ARG;
  ctx->g16_sparse_window = bits;
  return PCDHIP_OK;
}
int pcdhip_groth16_last_plan(pcdhip_ctx* ctx, uint32_t out[2]) {
  if (!ctx || !out) return PCDHIP_E_ARG;
  out[0] = (uint32_t)ctx->g16_last_sparse; out[1] = ctx->g16_last_general;
  return PCDHIP_OK;
}
static int drop_side_streams(pcdhip_ctx* ctx, bool partial);
static bool pipe_pending(const pcdhip_ctx* ctx);
int pcdhip_groth16_set_schedule(pcdhip_ctx* ctx, int mode) {
  if (!ctx || mode < 0 || mode > 2) return PCDHIP_E_ARG;
  if (pipe_pending(ctx)) return PCDHIP_E_ARG;
  std::vector<pcdhip_ctx*> all = ctx->peers.empty() ? std::vector<pcdhip_ctx*>{ctx} : ctx->peers;
  int first = PCDHIP_OK;
  for (pcdhip_ctx* p : all) {
    if (p->g16_schedule == mode) continue;
    p->g16_schedule = mode;
    const int rc = drop_side_streams(p, false);  // (their priorities depend on the schedule)
    if (rc && !first) first = rc;                // (every peer gets the new schedule before the first failure is reported: ADVICE r05)
  }
  return first;
}
int pcdhip_set_lane_reserve(pcdhip_ctx* ctx, int cus) {
  if (!ctx || cus < -1 || cus > 128) return PCDHIP_E_ARG;
  if (pipe_pending(ctx)) return PCDHIP_E_ARG;
  std::vector<pcdhip_ctx*> all = ctx->peers.empty() ? std::vector<pcdhip_ctx*>{ctx} : ctx->peers;
  int first = PCDHIP_OK;
  for (pcdhip_ctx* p : all) {
    if (p->lane_reserve == cus) continue;
    p->lane_reserve = cus;
    const int rc = drop_side_streams(p, false);
    if (rc && !first) first = rc;
  }
  return first;
}
int pcdhip_groth16_set_witness_split(pcdhip_ctx* ctx, int on) {
  if (!ctx) return PCDHIP_E_ARG;
  ctx->wm_split = on != 0;
  return PCDHIP_OK;
}
int pcdhip_msm_profile(pcdhip_ctx* ctx, int on) {
  if (!ctx) return PCDHIP_E_ARG;
  ctx->msm_profile = on != 0;
  return PCDHIP_OK;
}
int pcdhip_msm_last_plan(pcdhip_ctx* ctx, uint32_t out[2]) {
  if (!ctx || !out) return PCDHIP_E_ARG;
  out[0] = ctx->msm_tm.entries; out[1] = ctx->msm_tm.chunk;
  return PCDHIP_OK;
}
// The roof the MSM, FFT and pairing kernels are priced against, measured on the device at hand: v_mad_u64_u32 issue rate with four waves
// per SIMD, eight independent accumulator chains per lane (tools/microbench/k0_int_rates.hip found 3.42e13 lane-mads/s that way in round 1;
// boxes of the pool differ by ~10 % in sustained clock, and a fraction against a constant moves with the box).
namespace {
__global__ void __launch_bounds__(256) mad_rate_kernel(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  uint64_t c0 = a, c1 = b, c2 = a + 1, c3 = b + 1, c4 = a + 2, c5 = b + 2, c6 = a + 3, c7 = b + 3;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++)
      asm volatile(
          "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n"
          "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
          "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n"
          "v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : "vcc");
  }
  const uint64_t x = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)x ^ (uint32_t)(x >> 32);
}
}  // namespace
int pcdhip_mad_rate(pcdhip_ctx* ctx, double* out_lane_mads_per_s) {
  return guarded([&]() -> int {
  if (!ctx || !out_lane_mads_per_s) return PCDHIP_E_ARG;
  BIND();
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || cus <= 0) cus = 256;
  const int blocks = cus * 4, iters = 1000;  // one workgroup of four waves per SIMD quartet, four of them per CU: four waves per SIMD
  TRY(ctx->aux_ws.ensure(AUX_SCAL, (size_t)blocks * 256 * 4));
  uint32_t* out = (uint32_t*)ctx->aux_ws.buf[AUX_SCAL];
  EventSet<2> ev;
  TRY(ev.create());
  float best = 0;
  for (int r = 0; r < 9; r++) {  // (the first pass warms up; the best of eight: the clock needs a few milliseconds of load to settle)
    TRY(hipEventRecord(ev[0], ctx->stream));
    hipLaunchKernelGGL(mad_rate_kernel, dim3(blocks), dim3(256), 0, ctx->stream, out, r ? iters : 10, (uint32_t)r);
    TRY(hipEventRecord(ev[1], ctx->stream));
    TRY(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    TRY(hipEventElapsedTime(&ms, ev[0], ev[1]));
    if (r && (best == 0 || ms < best)) best = ms;
  }
  TRY(hipGetLastError());
  *out_lane_mads_per_s = (double)blocks * 256.0 * iters * 64.0 / ((double)best * 1e-3);
  return PCDHIP_OK;
  });
}
int pcdhip_msm_last_timings(pcdhip_ctx* ctx, float out_ms[8]) {
  if (!ctx || !out_ms) return PCDHIP_E_ARG;
  const MsmTimings& t = ctx->msm_tm;
  float v[8] = {t.digits, t.scan, t.scatter, t.accumulate, t.fixup, t.tail, t.horner, t.total};
  memcpy(out_ms, v, sizeof v);
  return PCDHIP_OK;
}

static int msm_common(pcdhip_ctx* ctx, const pcdhip_bases* bases, size_t offset, const uint32_t* scalars_dev, size_t n,
                      uint64_t* out_xyz, bool out_on_device = false) {
  const GroupEntry& ge = group_entry(bases->curve_id, bases->group_id);
  const size_t jac_b = (size_t)ge.point_words / 2 * 3 * 4, jac_abi_b = (size_t)ge.point_abi_words / 2 * 3 * 4;
  TRY(ctx->msm_ws.ensure(WS_OUT, jac_b + jac_abi_b + 64));
  uint32_t* out_dev = (uint32_t*)ctx->msm_ws.buf[WS_OUT];
  uint32_t* out_abi = out_dev + jac_b / 4;
  TRY(ge.msm(ctx->msm_ws, ctx->stream, bases->view(offset), scalars_dev, (uint32_t)n, out_dev, ctx->msm_c, ctx->msm_chunk,
             ctx->msm_sort, ctx->msm_profile ? &ctx->msm_tm : nullptr, nullptr, MSM_SHARE_NONE));
  if (out_on_device) { TRY(ge.jac_out(ctx->stream, out_dev, 1, (uint32_t*)out_xyz)); return PCDHIP_OK; }  // asynchronous
  TRY(ge.jac_out(ctx->stream, out_dev, 1, out_abi));
  TRY(hipMemcpyAsync(out_xyz, out_abi, jac_abi_b, hipMemcpyDeviceToHost, ctx->stream));
  uint32_t too_wide = 0;  // a scalar that is not a reduced canonical value (>= 2^bits): the digits would silently drop its top
  if (ctx->msm_ws.last_err_dev) TRY(hipMemcpyAsync(&too_wide, ctx->msm_ws.last_err_dev, 4, hipMemcpyDeviceToHost, ctx->stream));
  TRY(hipStreamSynchronize(ctx->stream));
  return too_wide ? PCDHIP_E_ARG : PCDHIP_OK;
}

// the context's side streams (created on first use): 0 and 1 with the highest priority, 2 with the default one, 3..5 with the lowest
// ... and the accumulate lane: a stream whose queue may use all but `lane_reserve` compute units.  The mask bits are dealt to the XCDs
// round-robin (bit i -> XCD i mod 8: tools/microbench/k3_cu_mask.hip, profiles/r05_k3_cu_mask.txt), so clearing the LAST 8 k bits leaves
// k CUs free in every XCD -- an unmasked kernel's workgroups are dealt to the XCDs round-robin as well, so each needs a free CU of its own.
static int lane_reserve_of(const pcdhip_ctx* ctx) {
  int r = ctx->lane_reserve;
  if (r < 0) { const char* e = getenv("PCDHIP_LANE_RESERVE"); r = e ? atoi(e) : 8; }
  return r < 0 ? 0 : r;
}
// CU-masked streams are made once per (device, reserve) and NEVER destroyed: on this stack (ROCm 7.2) hipStreamDestroy of a stream created
// by hipExtStreamCreateWithCUMask stalls for good now and then once such streams have been created and destroyed before in the process
// (round 5: a soak that toggled the schedule per proof hung in the second round of destructions, profiles/r05_stress.log).  Contexts
// of one device share the lane: stream order only adds dependencies between their accumulations, and the mode is not the default.
static hipStream_t masked_lane_of(int device, int cus, int reserve, hipError_t* err) {
  static std::mutex mu;
  static std::map<std::pair<int, int>, hipStream_t> pool;
  std::lock_guard<std::mutex> lock(mu);
  auto it = pool.find({device, reserve});
  if (it != pool.end()) { *err = hipSuccess; return it->second; }
  std::vector<uint32_t> mask((cus + 31) / 32, 0u);
  for (int i = 0; i < cus - reserve; i++) mask[i / 32] |= 1u << (i % 32);
  hipStream_t s = nullptr;
  *err = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
  if (*err == hipSuccess) pool[{device, reserve}] = s;
  return s;
}
static int drop_side_streams(pcdhip_ctx* ctx, bool partial);
static int make_side_streams(pcdhip_ctx* ctx);
static int ensure_side_streams(pcdhip_ctx* ctx) {
  if (ctx->g16_ready) return PCDHIP_OK;
  const int rc = make_side_streams(ctx);
  // (g16_ready is set last, so a failure partway -- a CU-masked stream the runtime refuses, the sixth stream -- used to leave
  //  the streams and events made so far in the context, to be overwritten and leaked by the next call: destroy them here instead)
  if (rc) (void)drop_side_streams(ctx, true);
  return rc;
}
static int make_side_streams(pcdhip_ctx* ctx) {
  int least = 0, greatest = 0;
  TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
  const bool lane_mode = ctx->g16_schedule == 2;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || cus <= 0) cus = 256;
  const int normal = (least + greatest) / 2;
  for (int k = 0; k < 6; k++) {
    // The runtime multiplexes the streams of ONE priority over a few hardware queues (three for the high priority on this stack), and two
    // streams that share a queue run in submission order: with four high-priority streams the sort of l' sat in the queue of the B MSM, behind
    // that MSM's wait for its accumulation on the lane -- l' and h started 3 ms late (profiles/r05_pt_lane_aliased.txt: q5 carries s4 AND s5).
    // Lane mode therefore spreads its six side streams two per priority (with the context's own stream: three at the normal one); the other
    // schedules keep round 4's mix (2 high, 1 normal, 3 low).
    int prio = lane_mode ? (k < 2 ? greatest : k < 4 ? normal : least) : (k < 2 ? greatest : k == 2 ? normal : least);
    static const char* side_prio = getenv("PCDHIP_SIDE_PRIO");  // developer knob: "low" = every side stream at the lowest priority, "mid" = none above normal
    if (side_prio && !strcmp(side_prio, "low")) prio = least;
    if (side_prio && !strcmp(side_prio, "mid")) prio = k < 3 ? normal : least;
    TRY(hipStreamCreateWithPriority(&ctx->g16_streams[k], hipStreamNonBlocking, prio));
    TRY(hipEventCreate(&ctx->g16_begin[k]));
    TRY(hipEventCreate(&ctx->g16_end[k]));
  }
  if (lane_mode) {
    int reserve = lane_reserve_of(ctx);
    if (reserve >= cus) reserve = 0;
    if (reserve > 0) {
      hipError_t e = hipSuccess;
      ctx->lane.stream = masked_lane_of(ctx->device, cus, reserve, &e);
      TRY(e);
    } else {
      TRY(hipStreamCreateWithPriority(&ctx->lane_stream, hipStreamNonBlocking, least));   // (owned: an ordinary stream can be destroyed)
      ctx->lane.stream = ctx->lane_stream;
    }
    ctx->lane.cus = cus - reserve;
  }
  TRY(hipEventCreateWithFlags(&ctx->g16_ready, hipEventDisableTiming));
  return PCDHIP_OK;
}
// (a change of schedule or of the lane's reserve takes effect on streams made afresh)
static int drop_side_streams(pcdhip_ctx* ctx, bool partial) {
  if (!ctx->g16_ready && !partial) return PCDHIP_OK;
  BIND();
  if (!partial) TRY(hipDeviceSynchronize());   // (partial: nothing has been enqueued on streams that were never handed out)
  for (int k = 0; k < 6; k++) {
    if (ctx->g16_streams[k]) { (void)hipStreamDestroy(ctx->g16_streams[k]); ctx->g16_streams[k] = nullptr; }
    if (ctx->g16_begin[k]) { (void)hipEventDestroy(ctx->g16_begin[k]); ctx->g16_begin[k] = nullptr; }
    if (ctx->g16_end[k]) { (void)hipEventDestroy(ctx->g16_end[k]); ctx->g16_end[k] = nullptr; }
  }
  if (ctx->lane_stream) { (void)hipStreamDestroy(ctx->lane_stream); ctx->lane_stream = nullptr; }   // (only the unmasked lane is owned)
  ctx->lane = MsmLane();
  if (ctx->g16_ready) (void)hipEventDestroy(ctx->g16_ready);
  ctx->g16_ready = nullptr;
  return PCDHIP_OK;
}
static bool pipe_pending(const pcdhip_ctx* ctx) {
  for (int k = 0; k < pcdhip_ctx::PIPE_SLOTS; k++) if (ctx->pipe_busy[k]) return true;
  return false;
}

static int msm_submit_common(pcdhip_ctx* ctx, const pcdhip_bases* bases, size_t offset, const pcdhip_buf* scalars, size_t scalar_offset, size_t n,
                             uint64_t* out_xyz_device, size_t out_slot_stride, int* ticket) {
  if (!ctx || !bases || !scalars || !ticket || !bases->shards.empty()) return PCDHIP_E_ARG;
  if (offset + n > bases->n || scalar_offset + n > scalars->n || n >= (1ull << 31)) return PCDHIP_E_ARG;
  if (scalars->field_id != kCurveFr[bases->curve_id]) return PCDHIP_E_ARG;
  BIND();
  int rc = ensure_side_streams(ctx);
  if (rc) return rc;
  int slot = -1;
  for (int t = 0; t < pcdhip_ctx::PIPE_SLOTS; t++) {
    const int k = (ctx->pipe_next + t) % pcdhip_ctx::PIPE_SLOTS;
    if (!ctx->pipe_busy[k]) { slot = k; break; }
  }
  if (slot < 0) return PCDHIP_E_ARG;  // PIPE_SLOTS submissions are in flight: collect one first
  ctx->pipe_next = (slot + 1) % pcdhip_ctx::PIPE_SLOTS;
  if (!ctx->pipe_host) TRY(hipHostMalloc((void**)&ctx->pipe_host, pcdhip_ctx::PIPE_SLOTS * pcdhip_ctx::PIPE_HOST_WORDS * 8, hipHostMallocDefault));
  if (!ctx->pipe_done[slot]) TRY(hipEventCreateWithFlags(&ctx->pipe_done[slot], hipEventDisableTiming));
  bool prev_bad = false;
  if (ctx->pipe_partial[slot]) {
    // the slot's previous ticket was handed to another stream (pcdhip_msm_ticket_wait), so nobody has looked at its error word and its
    // device-to-host copy may still be in flight: wait for that MSM (long finished in a pipelined loop: it was submitted PIPE_SLOTS
    // submissions ago and its gather has been consumed) before the word is reset, and report an unreduced scalar HERE
    TRY(hipEventSynchronize(ctx->pipe_done[slot]));
    ctx->pipe_partial[slot] = false;
    // (a code of its own, and the new submission goes ahead -- the caller learns that an EARLIER result was wrong without losing
    //  this one or mistaking the code for a refusal)
    prev_bad = (uint32_t)ctx->pipe_host[(size_t)slot * pcdhip_ctx::PIPE_HOST_WORDS + pcdhip_ctx::PIPE_HOST_WORDS - 1] != 0;
  }
  const GroupEntry& ge = group_entry(bases->curve_id, bases->group_id);
  const size_t jac_b = (size_t)ge.point_words / 2 * 3 * 4, jac_abi_b = (size_t)ge.point_abi_words / 2 * 3 * 4;
  MsmWorkspace& ws = ctx->g16_ws[2 + slot];
  hipStream_t sk = ctx->g16_streams[2 + slot];
  TRY(ws.ensure(WS_OUT, jac_b + jac_abi_b + 64));
  uint32_t* out_dev = (uint32_t*)ws.buf[WS_OUT];
  uint32_t* out_abi = out_xyz_device ? (uint32_t*)((char*)out_xyz_device + (size_t)slot * out_slot_stride) : out_dev + jac_b / 4;
  // ordered behind whatever the context's own stream has queued so far (uploads of the operands; for the device-result form also
  // the previous reader of `out_xyz_device`, which pcdhip_stream_wait put in front of the context's stream)
  TRY(hipEventRecord(ctx->g16_ready, ctx->stream));
  TRY(hipStreamWaitEvent(sk, ctx->g16_ready, 0));
  const size_t sw = (size_t)kFieldLimbs[scalars->field_id] * 2;
  // (the accumulations of the MSMs in flight run one after the other on the lane; sorts and bucket reductions beside them)
  ws.lane = ctx->g16_schedule == 2 && ctx->pipe_lane && ctx->lane.stream ? &ctx->lane : nullptr;
  const hipError_t me = ge.msm(ws, sk, bases->view(offset), scalars->dptr + scalar_offset * sw, (uint32_t)n, out_dev, ctx->msm_c, ctx->msm_chunk, ctx->msm_sort,
                               nullptr, nullptr, MSM_SHARE_NONE);
  ws.lane = nullptr;
  TRY(me);
  TRY(ge.jac_out(sk, out_dev, 1, out_abi));
  uint64_t* host = ctx->pipe_host + (size_t)slot * pcdhip_ctx::PIPE_HOST_WORDS;
  host[pcdhip_ctx::PIPE_HOST_WORDS - 1] = 0;
  if (!out_xyz_device) TRY(hipMemcpyAsync(host, out_abi, jac_abi_b, hipMemcpyDeviceToHost, sk));
  if (ws.last_err_dev) TRY(hipMemcpyAsync(&host[pcdhip_ctx::PIPE_HOST_WORDS - 1], ws.last_err_dev, 4, hipMemcpyDeviceToHost, sk));
  TRY(hipEventRecord(ctx->pipe_done[slot], sk));
  ctx->pipe_busy[slot] = true;
  ctx->pipe_out_bytes[slot] = out_xyz_device ? 0 : jac_abi_b;
  *ticket = slot;
  return prev_bad ? PCDHIP_E_PREV_TICKET : PCDHIP_OK;
}
int pcdhip_msm_ticket_status(pcdhip_ctx* ctx, int slot) {
  if (!ctx || slot < 0 || slot >= pcdhip_ctx::PIPE_SLOTS || ctx->pipe_busy[slot] || !ctx->pipe_partial[slot]) return PCDHIP_E_ARG;
  BIND();
  TRY(hipEventSynchronize(ctx->pipe_done[slot]));
  ctx->pipe_partial[slot] = false;
  return (uint32_t)ctx->pipe_host[(size_t)slot * pcdhip_ctx::PIPE_HOST_WORDS + pcdhip_ctx::PIPE_HOST_WORDS - 1] ? PCDHIP_E_PREV_TICKET : PCDHIP_OK;
}
int pcdhip_msm_submit(pcdhip_ctx* ctx, const pcdhip_bases* bases, size_t offset, const pcdhip_buf* scalars, size_t scalar_offset, size_t n,
                      int* ticket) {
  // (PCDHIP_E_PREV_TICKET belongs to the released-ticket form alone.  A plain submission that lands on a slot whose last
  //  user was released by pcdhip_msm_ticket_wait clears that slot's deferred word without reporting it -- callers of this entry point
  //  treat any non-zero code as a refusal and would drop a ticket that IS in flight, leaving the slot busy for good.  A caller that
  //  mixes both forms polls pcdhip_msm_ticket_status before it switches.)
  const int rc = msm_submit_common(ctx, bases, offset, scalars, scalar_offset, n, nullptr, 0, ticket);
  return rc == PCDHIP_E_PREV_TICKET ? PCDHIP_OK : rc;
}
int pcdhip_msm_submit_partial(pcdhip_ctx* ctx, const pcdhip_bases* bases, size_t offset, const pcdhip_buf* scalars, size_t scalar_offset, size_t n,
                              uint64_t* out_xyz_device_slots, size_t slot_stride_bytes, int* ticket) {
  if (!out_xyz_device_slots || (slot_stride_bytes & 7)) return PCDHIP_E_ARG;
  return msm_submit_common(ctx, bases, offset, scalars, scalar_offset, n, out_xyz_device_slots, slot_stride_bytes, ticket);
}
int pcdhip_msm_ticket_wait(pcdhip_ctx* ctx, int ticket, void* other_stream) {
  if (!ctx || ticket < 0 || ticket >= pcdhip_ctx::PIPE_SLOTS || !ctx->pipe_busy[ticket]) return PCDHIP_E_ARG;
  BIND();
  TRY(hipStreamWaitEvent((hipStream_t)other_stream, ctx->pipe_done[ticket], 0));
  ctx->pipe_busy[ticket] = false;  // only once the other stream IS ordered behind the MSM: a failed wait keeps the ticket
  ctx->pipe_partial[ticket] = true;
  return PCDHIP_OK;
}
int pcdhip_msm_collect(pcdhip_ctx* ctx, int ticket, uint64_t* out_xyz) {
  if (!ctx || ticket < 0 || ticket >= pcdhip_ctx::PIPE_SLOTS || !ctx->pipe_busy[ticket] || !out_xyz) return PCDHIP_E_ARG;
  BIND();
  ctx->pipe_busy[ticket] = false;
  TRY(hipEventSynchronize(ctx->pipe_done[ticket]));
  const uint64_t* host = ctx->pipe_host + (size_t)ticket * pcdhip_ctx::PIPE_HOST_WORDS;
  memcpy(out_xyz, host, ctx->pipe_out_bytes[ticket]);
  return (uint32_t)host[pcdhip_ctx::PIPE_HOST_WORDS - 1] ? PCDHIP_E_ARG : PCDHIP_OK;
}

// MSM over bases sharded across the devices of the context: every device runs the whole pipeline on the part of [offset, offset + n)
// that falls into its point range (scalars: the matching slice of the host vector), the partial results travel device to device
// into device 0's gather buffer and one wave sums them there.  No host synchronisation before the result is read back.
// (every error exit drains the peer streams before the frames that outstanding copies point into -- the error words, the caller's
//  scalars -- go away)
static void drain_peers(pcdhip_ctx* ctx) {
  for (pcdhip_ctx* C : ctx->peers) { if (hipSetDevice(C->device) == hipSuccess) (void)hipStreamSynchronize(C->stream); }
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
}
static int msm_sharded_impl(pcdhip_ctx* ctx, const pcdhip_bases* bases, size_t offset, const uint64_t* scalars, size_t n, uint64_t* out_xyz,
                            uint32_t* too_wide);
static int msm_sharded(pcdhip_ctx* ctx, const pcdhip_bases* bases, size_t offset, const uint64_t* scalars, size_t n, uint64_t* out_xyz) {
  if (ctx->peers.size() != bases->shards.size() || ctx->peers.size() > 64) return PCDHIP_E_ARG;
  uint32_t too_wide[64] = {0};
  const int rc = msm_sharded_impl(ctx, bases, offset, scalars, n, out_xyz, too_wide);
  if (rc) drain_peers(ctx);
  return rc;
}
static int msm_sharded_impl(pcdhip_ctx* ctx, const pcdhip_bases* bases, size_t offset, const uint64_t* scalars, size_t n, uint64_t* out_xyz,
                            uint32_t* too_wide) {
  const size_t G = bases->shards.size();
  const GroupEntry& ge = group_entry(bases->curve_id, bases->group_id);
  const size_t jb = (size_t)ge.point_abi_words / 2 * 3 * 4, ji = (size_t)ge.point_words / 2 * 3 * 4;
  const size_t sl = (size_t)kFieldLimbs[kCurveFr[bases->curve_id]];
  BIND();
  TRY(ctx->aux_ws.ensure(AUX_MISC, (G + 1) * jb + 64 * ji + 64));
  uint32_t* d = (uint32_t*)ctx->aux_ws.buf[AUX_MISC];  // result | gathered partials | tree scratch
  uint32_t* gather = d + jb / 4;
  uint32_t* scratch = d + (G + 1) * jb / 4;
  for (size_t g = 0; g < G; g++) {
    const size_t lo = std::max(offset, bases->shard_lo[g]), hi = std::min(offset + n, bases->shard_lo[g + 1]);
    pcdhip_ctx* C = ctx->peers[g];
    if (hi <= lo) {  // nothing of this MSM on device g: its partial is the identity (Z = 0)
      TRY(hipSetDevice(ctx->device));
      TRY(hipMemsetAsync((char*)gather + g * jb, 0, jb, ctx->stream));
      continue;
    }
    const pcdhip_bases* sh = bases->shards[g];
    const size_t ng = hi - lo;
    C->msm_sort = ctx->msm_sort;
    {
      pcdhip_ctx* ctx = C;  // (TRY / BIND report into the device's own context)
      BIND();
      const size_t sbytes = ng * sl * 8;
      TRY(C->msm_ws.ensure(WS_SCAL, std::max<size_t>(sbytes, 8)));
      TRY(C->msm_ws.ensure(WS_OUT, ji + jb + 64));
      TRY(hipMemcpyAsync(C->msm_ws.buf[WS_SCAL], scalars + (lo - offset) * sl, sbytes, hipMemcpyHostToDevice, C->stream));
      uint32_t* out_dev = (uint32_t*)C->msm_ws.buf[WS_OUT];
      uint32_t* out_abi = out_dev + ji / 4;
      TRY(ge.msm(C->msm_ws, C->stream, sh->view(lo - bases->shard_lo[g]), (const uint32_t*)C->msm_ws.buf[WS_SCAL], (uint32_t)ng, out_dev, C->msm_c,
                 C->msm_chunk, C->msm_sort, nullptr, nullptr, MSM_SHARE_NONE));
      TRY(ge.jac_out(C->stream, out_dev, 1, out_abi));
      if (C->msm_ws.last_err_dev) TRY(hipMemcpyAsync(&too_wide[g], C->msm_ws.last_err_dev, 4, hipMemcpyDeviceToHost, C->stream));
    }
    const uint32_t* out_abi = (const uint32_t*)C->msm_ws.buf[WS_OUT] + ji / 4;
    TRY(hipSetDevice(C->device));
    TRY(hipMemcpyPeerAsync((char*)gather + g * jb, ctx->device, out_abi, C->device, jb, C->stream));
    if (C != ctx) {
      if (!C->xstream_ev) TRY(hipEventCreateWithFlags(&C->xstream_ev, hipEventDisableTiming));
      TRY(hipEventRecord(C->xstream_ev, C->stream));
      TRY(hipSetDevice(ctx->device));
      TRY(hipStreamWaitEvent(ctx->stream, C->xstream_ev, 0));
    }
  }
  BIND();
  TRY(ge.points_sum(ctx->stream, gather, (uint32_t)G, scratch, d));
  TRY(hipMemcpyAsync(out_xyz, d, jb, hipMemcpyDeviceToHost, ctx->stream));
  for (size_t g = 1; g < G; g++) { TRY(hipSetDevice(ctx->peers[g]->device)); TRY(hipStreamSynchronize(ctx->peers[g]->stream)); }
  BIND();
  TRY(hipStreamSynchronize(ctx->stream));
  for (size_t g = 0; g < G; g++) if (too_wide[g]) return PCDHIP_E_ARG;
  return PCDHIP_OK;
}

int pcdhip_msm_dev(pcdhip_ctx* ctx, const pcdhip_bases* bases, size_t offset, const pcdhip_buf* scalars, size_t scalar_offset,
                   size_t n, uint64_t* out_xyz) {
  if (!ctx || !bases || !scalars || !out_xyz) return PCDHIP_E_ARG;
  if (!bases->shards.empty()) return PCDHIP_E_ARG;  // sharded bases take host scalars (pcdhip_msm): a pcdhip_buf lives on one device
  if (offset + n > bases->n || scalar_offset + n > scalars->n || n >= (1ull << 31)) return PCDHIP_E_ARG;
  if (scalars->field_id != kCurveFr[bases->curve_id]) return PCDHIP_E_ARG;
  BIND();
  const size_t sw = (size_t)kFieldLimbs[scalars->field_id] * 2;
  return msm_common(ctx, bases, offset, scalars->dptr + scalar_offset * sw, n, out_xyz);
}

int pcdhip_msm_dev_partial(pcdhip_ctx* ctx, const pcdhip_bases* bases, size_t offset, const pcdhip_buf* scalars, size_t scalar_offset,
                           size_t n, uint64_t* out_xyz_device) {
  if (!ctx || !bases || !scalars || !out_xyz_device || !bases->shards.empty()) return PCDHIP_E_ARG;
  if (offset + n > bases->n || scalar_offset + n > scalars->n || n >= (1ull << 31)) return PCDHIP_E_ARG;
  if (scalars->field_id != kCurveFr[bases->curve_id]) return PCDHIP_E_ARG;
  BIND();
  const size_t sw = (size_t)kFieldLimbs[scalars->field_id] * 2;
  return msm_common(ctx, bases, offset, scalars->dptr + scalar_offset * sw, n, out_xyz_device, true);
}

int pcdhip_msm(pcdhip_ctx* ctx, const pcdhip_bases* bases, size_t offset, const uint64_t* scalars, size_t n, uint64_t* out_xyz) {
  if (!ctx || !bases || (!scalars && n) || !out_xyz) return PCDHIP_E_ARG;
  if (offset + n > bases->n || n >= (1ull << 31)) return PCDHIP_E_ARG;
  if (!bases->shards.empty()) return msm_sharded(ctx, bases, offset, scalars, n, out_xyz);
  BIND();
  const size_t sbytes = n * kFieldLimbs[kCurveFr[bases->curve_id]] * 8;
  TRY(ctx->msm_ws.ensure(WS_SCAL, std::max<size_t>(sbytes, 8)));
  TRY(hipMemcpyAsync(ctx->msm_ws.buf[WS_SCAL], scalars, sbytes, hipMemcpyHostToDevice, ctx->stream));
  return msm_common(ctx, bases, offset, (const uint32_t*)ctx->msm_ws.buf[WS_SCAL], n, out_xyz);
}

static int points_sum_common(pcdhip_ctx* ctx, int curve_id, int group_id, const uint64_t* xyz, size_t n, uint64_t* out_xyz, bool in_on_device) {
  if (!ctx || !valid_curve(curve_id) || !valid_group(group_id) || (!xyz && n) || !out_xyz || (n >> 24)) return PCDHIP_E_ARG;
  BIND();
  const GroupEntry& ge = group_entry(curve_id, group_id);
  const size_t jb = (size_t)ge.point_abi_words / 2 * 3 * 4, ji = (size_t)ge.point_words / 2 * 3 * 4;
  TRY(ctx->aux_ws.ensure(AUX_MISC, (n + 1) * jb + 64 * ji + 64));
  uint32_t* d = (uint32_t*)ctx->aux_ws.buf[AUX_MISC];  // result | staged input | tree scratch
  uint32_t* scratch = d + (n + 1) * jb / 4;
  const uint32_t* in = (const uint32_t*)xyz;
  if (!in_on_device) {
    TRY(hipMemcpyAsync(d + jb / 4, xyz, n * jb, hipMemcpyHostToDevice, ctx->stream));
    in = d + jb / 4;
  }
  TRY(ge.points_sum(ctx->stream, in, (uint32_t)n, scratch, d));
  TRY(hipMemcpyAsync(out_xyz, d, jb, hipMemcpyDeviceToHost, ctx->stream));
  TRY(hipStreamSynchronize(ctx->stream));
  return PCDHIP_OK;
}
int pcdhip_points_sum(pcdhip_ctx* ctx, int curve_id, int group_id, const uint64_t* xyz, size_t n, uint64_t* out_xyz) {
  return points_sum_common(ctx, curve_id, group_id, xyz, n, out_xyz, false);
}
int pcdhip_points_sum_dev(pcdhip_ctx* ctx, int curve_id, int group_id, const uint64_t* xyz_device, size_t n, uint64_t* out_xyz) {
  return points_sum_common(ctx, curve_id, group_id, xyz_device, n, out_xyz, true);
}

int pcdhip_to_affine(pcdhip_ctx* ctx, int curve_id, int group_id, const uint64_t* xyz, size_t n, uint64_t* out_xy, uint8_t* out_inf) {
  if (!ctx || !valid_curve(curve_id) || !valid_group(group_id) || (!xyz && n) || !out_xy) return PCDHIP_E_ARG;
  BIND();
  const GroupEntry& ge = group_entry(curve_id, group_id);
  const size_t ab = (size_t)ge.point_abi_words * 4, jb = ab / 2 * 3;
  TRY(ctx->aux_ws.ensure(AUX_MISC, n * (jb + ab) + 64));
  uint32_t* dj = (uint32_t*)ctx->aux_ws.buf[AUX_MISC];
  uint32_t* da = dj + n * jb / 4;
  TRY(hipMemcpyAsync(dj, xyz, n * jb, hipMemcpyHostToDevice, ctx->stream));
  TRY(ge.to_affine(ctx->stream, dj, (uint32_t)n, da));
  TRY(hipMemcpyAsync(out_xy, da, n * ab, hipMemcpyDeviceToHost, ctx->stream));
  TRY(hipStreamSynchronize(ctx->stream));
  if (out_inf) {
    for (size_t i = 0; i < n; i++) {
      const uint64_t* p = out_xy + i * (ab / 8);
      uint64_t o = 0;
      for (size_t k = 0; k < ab / 8; k++) o |= p[k];
      out_inf[i] = (o == 0) ? 1 : 0;
    }
  }
  return PCDHIP_OK;
}

// ------------------------------------------------------------------------------------------------ FFT
static int fft_dev_general(pcdhip_ctx* ctx, pcdhip_buf* data, const Dom& d, int inverse, int coset) {
  inverse = inverse ? 1 : 0;  // (the higher bits of this argument are internal flags of the pass driver: never from the ABI)
  coset = coset ? 1 : 0;
  const FieldEntry& fe = field_entry(data->field_id);
  if (data->n < d.n) return PCDHIP_E_ARG;
  const size_t vb = (size_t)d.n * fe.words * 4;
  TRY(ctx->aux_ws.ensure(AUX_FFT_X, vb));
  TRY(ctx->aux_ws.ensure(AUX_FFT_TMP, vb));
  uint32_t* x = (uint32_t*)ctx->aux_ws.buf[AUX_FFT_X];
  // C-ABI image -> device image, transform, and back (the witness-map pipeline keeps vectors in the device image
  // between its seven transforms and pays neither conversion)
  TRY(fe.convert(ctx->stream, data->dptr, x, d.n, 0));
  int rc = domain_transform(ctx, data->field_id, d, x, (uint32_t*)ctx->aux_ws.buf[AUX_FFT_TMP], inverse, coset, ctx->fft_ms, &ctx->fft_passes);
  if (rc) return rc;
  TRY(fe.convert(ctx->stream, x, data->dptr, d.n, 1));
  TRY(hipStreamSynchronize(ctx->stream));
  return PCDHIP_OK;
}
int pcdhip_fft_dev(pcdhip_ctx* ctx, pcdhip_buf* data, uint32_t log_n, int inverse, int coset) {
  if (!ctx || !data || log_n > 30) return PCDHIP_E_ARG;
  if ((int)log_n > field_entry(data->field_id).two_adicity) return PCDHIP_E_SIZE_UNSUPPORTED;
  BIND();
  Dom d = {1u << log_n, 1, (int)log_n};
  return fft_dev_general(ctx, data, d, inverse, coset);
}
static int fft_host_general(pcdhip_ctx* ctx, int field_id, uint64_t* data, const Dom& d, int inverse, int coset) {
  const FieldEntry& fe = field_entry(field_id);
  const size_t bytes = (size_t)d.n * fe.abi_words * 4;
  TRY(ctx->aux_ws.ensure(AUX_SCAL, bytes));
  pcdhip_buf tmp;
  tmp.field_id = field_id; tmp.n = d.n; tmp.dptr = (uint32_t*)ctx->aux_ws.buf[AUX_SCAL];
  TRY(hipMemcpyAsync(tmp.dptr, data, bytes, hipMemcpyHostToDevice, ctx->stream));
  int rc = fft_dev_general(ctx, &tmp, d, inverse, coset);
  if (rc) return rc;
  TRY(hipMemcpyAsync(data, tmp.dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
  TRY(hipStreamSynchronize(ctx->stream));
  return PCDHIP_OK;
}
int pcdhip_fft(pcdhip_ctx* ctx, int field_id, uint64_t* data, uint32_t log_n, int inverse, int coset) {
  if (!ctx || !data || !valid_field(field_id) || log_n > 30) return PCDHIP_E_ARG;
  if ((int)log_n > field_entry(field_id).two_adicity) return PCDHIP_E_SIZE_UNSUPPORTED;
  BIND();
  Dom d = {1u << log_n, 1, (int)log_n};
  return fft_host_general(ctx, field_id, data, d, inverse, coset);
}
int pcdhip_fft_general(pcdhip_ctx* ctx, int field_id, uint64_t* data, size_t n, int inverse, int coset) {
  if (!ctx || !data || !valid_field(field_id)) return PCDHIP_E_ARG;
  Dom d;
  int rc = split_domain(field_id, n, &d);
  if (rc) return rc;
  BIND();
  return fft_host_general(ctx, field_id, data, d, inverse, coset);
}
// Several transforms of ONE host vector with one trip over PCIe (round 5): pcdhip_fft moves the vector both ways per
// transform (2^20 elements over the 298-bit field: 42 MB each way for 0.3 ms of passes), and what ark-poly's callers do with a polynomial
// is usually a chain -- Marlin's `ifft` of evaluations followed by `coset_fft` on a larger domain, the witness map's `ifft; coset_fft` --
// that seam S2 (rust/src/s2.rs) can hand over whole.  ops[i]: bit 0 inverse, bit 1 coset; all on the domain of n = 2^a q^b elements.
int pcdhip_fft_seq(pcdhip_ctx* ctx, int field_id, uint64_t* data, size_t n, const int* ops, int n_ops) {
  if (!ctx || !data || !valid_field(field_id) || !ops || n_ops < 1 || n_ops > 64) return PCDHIP_E_ARG;
  for (int i = 0; i < n_ops; i++) if (ops[i] < 0 || ops[i] > 3) return PCDHIP_E_ARG;
  Dom d;
  int rc = split_domain(field_id, n, &d);
  if (rc) return rc;
  BIND();
  const FieldEntry& fe = field_entry(field_id);
  const size_t bytes = (size_t)d.n * fe.abi_words * 4, vb = (size_t)d.n * fe.words * 4;
  TRY(ctx->aux_ws.ensure(AUX_SCAL, bytes));
  TRY(ctx->aux_ws.ensure(AUX_FFT_X, vb));
  TRY(ctx->aux_ws.ensure(AUX_FFT_TMP, vb));
  uint32_t* abi = (uint32_t*)ctx->aux_ws.buf[AUX_SCAL];
  uint32_t* x = (uint32_t*)ctx->aux_ws.buf[AUX_FFT_X];
  TRY(hipMemcpyAsync(abi, data, bytes, hipMemcpyHostToDevice, ctx->stream));
  TRY(fe.convert(ctx->stream, abi, x, d.n, 0));
  for (int i = 0; i < n_ops; i++) {  // (the vector stays in the device image between the transforms: neither conversion in between)
    rc = domain_transform(ctx, field_id, d, x, (uint32_t*)ctx->aux_ws.buf[AUX_FFT_TMP], ops[i] & 1, (ops[i] >> 1) & 1, ctx->fft_ms, &ctx->fft_passes);
    if (rc) return rc;
  }
  TRY(fe.convert(ctx->stream, x, abi, d.n, 1));
  TRY(hipMemcpyAsync(data, abi, bytes, hipMemcpyDeviceToHost, ctx->stream));
  TRY(hipStreamSynchronize(ctx->stream));
  return PCDHIP_OK;
}
size_t pcdhip_domain_size(int field_id, size_t min_size) {
  if (!valid_field(field_id) || min_size == 0) return 0;
  Dom d;
  return pick_domain(field_id, min_size, &d) == PCDHIP_OK ? d.n : 0;
}
int pcdhip_fft_last_timings(pcdhip_ctx* ctx, float out_ms[8]) {
  if (!ctx || !out_ms) return PCDHIP_E_ARG;
  memcpy(out_ms, ctx->fft_ms, sizeof ctx->fft_ms);
  return ctx->fft_passes;
}

// ------------------------------------------------------------------------------------------------ witness map
namespace {

// device layout of one CSR matrix inside `d`: row_ptr | coeff (device image) | col | nl | long_rows | lc  (DevCsr, common.h)
size_t csr_bytes(const pcdhip_csr* m, const FieldEntry& fe, size_t off[6]) {
  const uint64_t nnz = m->row_ptr[m->num_rows];
  auto up8 = [](size_t x) { return (x + 7) / 8 * 8; };
  off[0] = 0;
  off[1] = (m->num_rows + 1) * 8;
  off[2] = off[1] + up8(nnz * fe.words * 4);
  off[3] = off[2] + up8(nnz * 4);
  off[4] = off[3] + up8((m->num_rows + 1) * 4);
  off[5] = off[4] + up8((m->num_rows + 1) * 4);   // (at most every row is long)
  return off[5] + up8(nnz + 8);
}
// host-side shape check of a caller's CSR matrix: row_ptr starts at 0 and never decreases, every column index is below
// `num_cols` (the kernels index the assignment with it) -- PCDHIP_E_ARG instead of an out-of-bounds device read
int validate_csr(const pcdhip_csr* m, size_t num_cols) {
  if (!m || !m->row_ptr || (m->num_rows >> 31) || m->row_ptr[0] != 0) return PCDHIP_E_ARG;
  for (uint64_t r = 0; r < m->num_rows; r++) if (m->row_ptr[r + 1] < m->row_ptr[r]) return PCDHIP_E_ARG;
  const uint64_t nnz = m->row_ptr[m->num_rows];
  if (nnz >> 31) return PCDHIP_E_ARG;
  if (nnz && (!m->col || !m->coeff)) return PCDHIP_E_ARG;
  uint32_t worst = 0;
  for (uint64_t k = 0; k < nnz; k++) worst = std::max(worst, m->col[k]);
  return (nnz && worst >= num_cols) ? PCDHIP_E_ARG : PCDHIP_OK;
}
// The small integers among the coefficients, recognised on the host by their C-ABI image (65 candidates, -32 .. 32, keyed by their
// first limb).  Real constraint systems are mostly such coefficients; the mat-vec kernels add instead of multiplying for them.
struct SmallCoeffTable {
  size_t limbs;
  std::vector<uint64_t> img;              // 65 x limbs
  std::vector<std::pair<uint64_t, int>> key;  // (first limb, c) sorted
  SmallCoeffTable(const FieldEntry& fe) : limbs((size_t)fe.abi_words / 2), img(65 * limbs) {
    for (int c = -32; c <= 32; c++) {
      fe.small_abi(c, (uint32_t*)&img[(size_t)(c + 32) * limbs]);
      key.push_back({img[(size_t)(c + 32) * limbs], c});
    }
    std::sort(key.begin(), key.end());
  }
  // the small integer equal to the coefficient at `w`, or 127
  int classify(const uint64_t* w) const {
    auto it = std::lower_bound(key.begin(), key.end(), std::make_pair(w[0], -64));
    for (; it != key.end() && it->first == w[0]; ++it)
      if (memcmp(w, &img[(size_t)(it->second + 32) * limbs], limbs * 8) == 0) return it->second;
    return 127;
  }
};
int upload_csr_to(pcdhip_ctx* ctx, const pcdhip_csr* m, const FieldEntry& fe, size_t num_cols, char* d, DevCsr* out) try {
  int vrc = validate_csr(m, num_cols);
  if (vrc) return vrc;
  const uint64_t nnz = m->row_ptr[m->num_rows], rows = m->num_rows;
  size_t off[6];
  csr_bytes(m, fe, off);
  // classify, and reorder every row: light entries (small integer coefficients) first
  const size_t limbs = (size_t)fe.abi_words / 2;
  const SmallCoeffTable table(fe);
  std::vector<uint32_t> col(nnz), nl(rows + 1, 0), long_rows;
  std::vector<int8_t> lc(nnz + 8, 0);
  std::vector<uint64_t> heavy(nnz * limbs, 0);   // ABI coefficients in the new order (light slots stay zero)
  for (uint64_t r = 0; r < rows; r++) {
    const uint64_t lo = m->row_ptr[r], hi = m->row_ptr[r + 1];
    uint64_t front = lo, back = hi;
    for (uint64_t k = lo; k < hi; k++) {
      const int c = table.classify(m->coeff + k * limbs);
      if (c != 127) { col[front] = m->col[k]; lc[front] = (int8_t)c; front++; }
      else { back--; col[back] = m->col[k]; lc[back] = 127; memcpy(&heavy[back * limbs], m->coeff + k * limbs, limbs * 8); }
    }
    nl[r] = (uint32_t)(front - lo);
    if (hi - lo > SPMV_LONG_ROW) long_rows.push_back((uint32_t)r);
  }
  TRY(hipMemcpyAsync(d + off[0], m->row_ptr, (rows + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  TRY(hipMemcpyAsync(d + off[3], nl.data(), (rows + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  if (!long_rows.empty()) TRY(hipMemcpyAsync(d + off[4], long_rows.data(), long_rows.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  if (nnz) {
    TRY(ctx->aux_ws.ensure(AUX_SCAL, nnz * fe.abi_words * 4));
    TRY(hipMemcpyAsync(ctx->aux_ws.buf[AUX_SCAL], heavy.data(), nnz * fe.abi_words * 4, hipMemcpyHostToDevice, ctx->stream));
    TRY(fe.convert(ctx->stream, (const uint32_t*)ctx->aux_ws.buf[AUX_SCAL], (uint32_t*)(d + off[1]), (uint32_t)nnz, 0));
    TRY(hipMemcpyAsync(d + off[2], col.data(), nnz * 4, hipMemcpyHostToDevice, ctx->stream));
    TRY(hipMemcpyAsync(d + off[5], lc.data(), nnz, hipMemcpyHostToDevice, ctx->stream));
  }
  TRY(hipStreamSynchronize(ctx->stream));  // the host vectors above and the staging slot (reused by the next matrix) are done with
  out->rp = (const uint64_t*)(d + off[0]);
  out->coeff = (const uint32_t*)(d + off[1]);
  out->col = (const uint32_t*)(d + off[2]);
  out->nl = (const uint32_t*)(d + off[3]);
  out->long_rows = (const uint32_t*)(d + off[4]);
  out->lc = (const int8_t*)(d + off[5]);
  out->n_long = (uint32_t)long_rows.size();
  out->rows = (uint32_t)m->num_rows;
  return PCDHIP_OK;
} catch (const std::bad_alloc&) { return PCDHIP_E_OOM; }
int upload_csr(pcdhip_ctx* ctx, int slot, const pcdhip_csr* m, const FieldEntry& fe, size_t num_cols, DevCsr* out) {
  if (!m || !m->row_ptr) return PCDHIP_E_ARG;
  size_t off[6];
  TRY(ctx->aux_ws.ensure(slot, csr_bytes(m, fe, off) + 64));
  return upload_csr_to(ctx, m, fe, num_cols, (char*)ctx->aux_ws.buf[slot], out);
}

// h (d.n elements, device image) left in aux slot AUX_A; z_dev: m elements on device; mats: A, B, C on device
// ifft then coset_fft of v (tmp: ping-pong partner), result in v.  On a radix-2 domain the two transforms are chained without the copy back
// that an odd number of passes otherwise ends with (flag 4 of `inverse`: the first leaves its result in tmp, the second starts there and
// its last pass lands in v): six such copies of n elements per witness map gone (10 % of a 298-bit transform at 2^20).
int chain_transforms(pcdhip_ctx* ctx, int field_id, const Dom& d, uint32_t* v, uint32_t* tmp) {
  int rc;
  if (d.m != 1) {
    rc = domain_transform(ctx, field_id, d, v, tmp, 1, 0, nullptr, nullptr); if (rc) return rc;
    return domain_transform(ctx, field_id, d, v, tmp, 0, 1, nullptr, nullptr);
  }
  int P = 0;
  rc = domain_transform(ctx, field_id, d, v, tmp, 1 | 4, 0, nullptr, &P); if (rc) return rc;
  uint32_t *s2 = (P & 1) ? tmp : v, *t2 = (P & 1) ? v : tmp;
  return domain_transform(ctx, field_id, d, s2, t2, 0 | 4, 1, nullptr, &P);  // P odd: ends in v; P even: stays in v
}
// One of the three independent chains of the witness map on `ctx`'s device and stream: v = M z (k == 0: plus the input-consistency
// rows), then ifft and coset_fft over the domain d, in ctx's own AUX_A / AUX_B / AUX_C buffer `slot` (tmp: AUX_FFT_TMP).
int witness_chain_dev(pcdhip_ctx* ctx, int field_id, const DevCsr& mat, int k, const uint32_t* z_dev, size_t num_inputs, const Dom& d, int slot,
                      bool transforms = true) {
  const FieldEntry& fe = field_entry(field_id);
  const size_t vb = (size_t)d.n * fe.words * 4;
  TRY(ctx->aux_ws.ensure(slot, vb));
  TRY(ctx->aux_ws.ensure(AUX_FFT_TMP, vb));
  uint32_t* v = (uint32_t*)ctx->aux_ws.buf[slot];
  uint32_t* tmp = (uint32_t*)ctx->aux_ws.buf[AUX_FFT_TMP];
  TRY(fe.spmv(ctx->stream, mat, z_dev, (uint32_t)num_inputs, k == 0 ? 1 : 0, d.n, v));
  if (!transforms) return PCDHIP_OK;
  return chain_transforms(ctx, field_id, d, v, tmp);
}
// h = coset_ifft((a o b - c) / Z) from the three transformed chains in ctx's AUX_A, AUX_B, AUX_C; h ends up in AUX_A
int witness_finish_dev(pcdhip_ctx* ctx, int field_id, const Dom& d, const uint32_t* b_at = nullptr, const uint32_t* c_at = nullptr) {
  const FieldEntry& fe = field_entry(field_id);
  uint32_t* a = (uint32_t*)ctx->aux_ws.buf[AUX_A];
  const uint32_t* b = b_at ? b_at : (const uint32_t*)ctx->aux_ws.buf[AUX_B];
  const uint32_t* c = c_at ? c_at : (const uint32_t*)ctx->aux_ws.buf[AUX_C];
  uint32_t* tmp = (uint32_t*)ctx->aux_ws.buf[AUX_FFT_TMP];
  int rc;
  const FftTables* t;
  if (d.m == 1) {
    rc = get_tables(ctx, field_id, d.a, &t); if (rc) return rc;
    TRY(fe.mul_sub_divz(ctx->stream, *t, a, b, c, d.a));
  } else {
    rc = get_mixed_tables(ctx, field_id, d, &t); if (rc) return rc;
    TRY(fe.mixed_mul_sub_divz(ctx->stream, *t, a, b, c, d.n));
  }
  return domain_transform(ctx, field_id, d, a, tmp, 1, 1, nullptr, nullptr);
}
int witness_map_dev(pcdhip_ctx* ctx, int field_id, const DevCsr mats[3], const uint32_t* z_dev, size_t num_inputs, Dom* dom_out,
                    hipEvent_t after_spmv = nullptr) {
  const FieldEntry& fe = field_entry(field_id);
  if (mats[0].rows != mats[1].rows || mats[0].rows != mats[2].rows) return PCDHIP_E_ARG;
  Dom d;
  int rc = pick_domain(field_id, (size_t)mats[0].rows + num_inputs, &d);
  if (rc) return rc;
  const uint32_t n = d.n;
  const size_t vb = (size_t)n * fe.words * 4;
  hipStream_t st = ctx->stream;
  if (d.m == 1) {
    // radix-2 domain: the three chains live back to back in AUX_A (a | b | c) and every pass of their transforms is ONE launch for all
    // three (grid.y): 6 launches instead of 18 -- inside a proof every launch of the map queues behind the MSMs' resident accumulation
    // waves, so the map's length is its launch count as much as its work
    TRY(ctx->aux_ws.ensure(AUX_A, 3 * vb));
    TRY(ctx->aux_ws.ensure(AUX_FFT_TMP, 3 * vb));
    uint32_t* a = (uint32_t*)ctx->aux_ws.buf[AUX_A];
    uint32_t* tmp = (uint32_t*)ctx->aux_ws.buf[AUX_FFT_TMP];
    const size_t ew = (size_t)n * fe.words;
    TRY(fe.spmv3(st, mats, z_dev, (uint32_t)num_inputs, n, a, ew));
    if (after_spmv) TRY(hipEventRecord(after_spmv, st));
    const FftTables* t;
    rc = get_tables(ctx, field_id, d.a, &t);
    if (rc) return rc;
    int P = 0;
    TRY(fe.fft_run_batched(st, *t, a, tmp, d.a, 1 | 4, 0, &P, 3));                 // ifft, result left where the last pass wrote it
    uint32_t *s2 = (P & 1) ? tmp : a, *t2 = (P & 1) ? a : tmp;
    TRY(fe.fft_run_batched(st, *t, s2, t2, d.a, 0 | 4, 1, &P, 3));                 // coset fft from there: lands in a | b | c
    rc = witness_finish_dev(ctx, field_id, d, a + ew, a + 2 * ew);
    if (rc) return rc;
    *dom_out = d;
    return PCDHIP_OK;
  }
  TRY(ctx->aux_ws.ensure(AUX_A, vb));
  TRY(ctx->aux_ws.ensure(AUX_B, vb));
  TRY(ctx->aux_ws.ensure(AUX_C, vb));
  TRY(ctx->aux_ws.ensure(AUX_FFT_TMP, vb));
  uint32_t *a = (uint32_t*)ctx->aux_ws.buf[AUX_A], *b = (uint32_t*)ctx->aux_ws.buf[AUX_B], *c = (uint32_t*)ctx->aux_ws.buf[AUX_C];
  uint32_t* tmp = (uint32_t*)ctx->aux_ws.buf[AUX_FFT_TMP];
  uint32_t* vecs[3] = {a, b, c};
  for (int k = 0; k < 3; k++)
    TRY(fe.spmv(st, mats[k], z_dev, (uint32_t)num_inputs, k == 0 ? 1 : 0, n, vecs[k]));
  if (after_spmv) TRY(hipEventRecord(after_spmv, st));
  // 3 x (ifft, coset_fft), pointwise, coset_ifft
  for (uint32_t* v : vecs) { rc = chain_transforms(ctx, field_id, d, v, tmp); if (rc) return rc; }
  rc = witness_finish_dev(ctx, field_id, d);
  if (rc) return rc;
  *dom_out = d;
  return PCDHIP_OK;
}

int upload_three(pcdhip_ctx* ctx, const pcdhip_csr* A, const pcdhip_csr* B, const pcdhip_csr* C, const FieldEntry& fe, size_t num_vars,
                 DevCsr out[3]) {
  const pcdhip_csr* ms[3] = {A, B, C};
  const int slots[3] = {AUX_CSR_RP, AUX_CSR_COL, AUX_CSR_COEF};  // one aux slot per matrix
  for (int k = 0; k < 3; k++) { int rc = upload_csr(ctx, slots[k], ms[k], fe, num_vars, &out[k]); if (rc) return rc; }
  return PCDHIP_OK;
}
}  // namespace

int pcdhip_groth16_witness_map(pcdhip_ctx* ctx, int field_id, const pcdhip_csr* A, const pcdhip_csr* B, const pcdhip_csr* C,
                               const uint64_t* z, size_t num_vars, size_t num_inputs, uint64_t* h_out) {
  if (!ctx || !valid_field(field_id) || !A || !B || !C || !z || !h_out || num_inputs == 0 || num_inputs > num_vars) return PCDHIP_E_ARG;
  if (num_vars >> 31) return PCDHIP_E_ARG;
  BIND();
  const FieldEntry& fe = field_entry(field_id);
  TRY(ctx->aux_ws.ensure(AUX_Z, num_vars * fe.words * 4));
  TRY(ctx->aux_ws.ensure(AUX_Z_CANON, num_vars * fe.abi_words * 4));
  TRY(hipMemcpyAsync(ctx->aux_ws.buf[AUX_Z_CANON], z, num_vars * fe.abi_words * 4, hipMemcpyHostToDevice, ctx->stream));
  TRY(fe.convert(ctx->stream, (const uint32_t*)ctx->aux_ws.buf[AUX_Z_CANON], (uint32_t*)ctx->aux_ws.buf[AUX_Z], (uint32_t)num_vars, 0));
  Dom dom;
  DevCsr mats[3];
  int rc = upload_three(ctx, A, B, C, fe, num_vars, mats);
  if (rc) return rc;
  rc = witness_map_dev(ctx, field_id, mats, (const uint32_t*)ctx->aux_ws.buf[AUX_Z], num_inputs, &dom);
  if (rc) return rc;
  const uint32_t n = dom.n;
  TRY(ctx->aux_ws.ensure(AUX_H_CANON, (size_t)n * fe.abi_words * 4));
  TRY(fe.convert(ctx->stream, (const uint32_t*)ctx->aux_ws.buf[AUX_A], (uint32_t*)ctx->aux_ws.buf[AUX_H_CANON], n, 1));
  TRY(hipMemcpyAsync(h_out, ctx->aux_ws.buf[AUX_H_CANON], (size_t)n * fe.abi_words * 4, hipMemcpyDeviceToHost, ctx->stream));
  TRY(hipStreamSynchronize(ctx->stream));
  return PCDHIP_OK;
}

// The witness map ALONE over the matrices resident with a key (pcdhip_g16_pk_set_r1cs), nothing else running: what
// pcdhip_groth16_prove queues on the context's stream while its MSMs run on the side streams.  out_ms = device time of
// [the three mat-vecs, the seven transforms + the pointwise step, both].  h_out may be null (timing only).
int pcdhip_g16_witness_map_resident(pcdhip_ctx* ctx, const pcdhip_g16_pk* pk, const uint64_t* z, uint64_t* h_out, float out_ms[3]) {
  if (!ctx || !pk || !z) return PCDHIP_E_ARG;
  if (!pk->shards.empty()) pk = pk->shards[0];
  if (!pk->r1cs_dev) return PCDHIP_E_ARG;
  BIND();
  const int fr = kCurveFr[pk->curve_id];
  const FieldEntry& fe = field_entry(fr);
  const size_t m = pk->num_vars;
  TRY(ctx->aux_ws.ensure(AUX_Z, m * fe.words * 4));
  TRY(ctx->aux_ws.ensure(AUX_Z_CANON, m * fe.abi_words * 4));
  TRY(hipMemcpyAsync(ctx->aux_ws.buf[AUX_Z_CANON], z, m * fe.abi_words * 4, hipMemcpyHostToDevice, ctx->stream));
  TRY(fe.convert(ctx->stream, (const uint32_t*)ctx->aux_ws.buf[AUX_Z_CANON], (uint32_t*)ctx->aux_ws.buf[AUX_Z], (uint32_t)m, 0));
  DevCsr mats[3];
  for (int k = 0; k < 3; k++) mats[k] = pk->mats[k];
  EventSet<3> ev;
  TRY(ev.create());
  TRY(hipEventRecord(ev[0], ctx->stream));
  Dom dom;
  int rc = witness_map_dev(ctx, fr, mats, (const uint32_t*)ctx->aux_ws.buf[AUX_Z], pk->num_inputs, &dom, ev[1]);
  if (rc) return rc;
  TRY(hipEventRecord(ev[2], ctx->stream));
  if (h_out) {
    TRY(ctx->aux_ws.ensure(AUX_H_CANON, (size_t)dom.n * fe.abi_words * 4));
    TRY(fe.convert(ctx->stream, (const uint32_t*)ctx->aux_ws.buf[AUX_A], (uint32_t*)ctx->aux_ws.buf[AUX_H_CANON], dom.n, 1));
    TRY(hipMemcpyAsync(h_out, ctx->aux_ws.buf[AUX_H_CANON], (size_t)dom.n * fe.abi_words * 4, hipMemcpyDeviceToHost, ctx->stream));
  }
  TRY(hipStreamSynchronize(ctx->stream));
  if (out_ms) {
    (void)hipEventElapsedTime(&out_ms[0], ev[0], ev[1]);
    (void)hipEventElapsedTime(&out_ms[1], ev[1], ev[2]);
    (void)hipEventElapsedTime(&out_ms[2], ev[0], ev[2]);
  }
  return PCDHIP_OK;
}

// ------------------------------------------------------------------------------------------------ Groth16
int pcdhip_g16_pk_upload(pcdhip_ctx* ctx, const pcdhip_g16_pk_host* h, pcdhip_g16_pk** out) {
  return guarded([&]() -> int {
  if (!ctx || !h || !out || !valid_curve((int)h->curve_id)) return PCDHIP_E_ARG;
  if (!h->alpha_g1 || !h->beta_g1 || !h->delta_g1 || !h->beta_g2 || !h->delta_g2 || !h->a_query || !h->b_g1_query ||
      !h->b_g2_query || (!h->h_query && h->h_len) || (!h->l_query && h->l_len))
    return PCDHIP_E_ARG;
  if (h->num_vars < 1 || h->num_inputs < 1 || h->num_inputs > h->num_vars || h->l_len != h->num_vars - h->num_inputs) return PCDHIP_E_ARG;
  BIND();
  const int cid = (int)h->curve_id;
  const size_t m = h->num_vars, ni = h->num_inputs;
  // delta is appended to the a / b / l queries: r*delta, s*delta and -rs*delta then ride inside the MSMs as
  // one more (base, scalar) pair instead of being serial scalar multiplications in the assembly.
  // Every a / b / l query gets four trailing slots matching the scalar tail [r, s, -rs, 1] that follows the
  // assignment: delta sits in the slot whose scalar the query needs, the vk point (alpha / beta) in the last one,
  // the remaining slots hold the point at infinity.  See inst_g16.hip.
  struct HostQuery { std::vector<uint64_t> pts; std::vector<uint8_t> inf; size_t n = 0; int group = 1; };
  auto with_tail = [&](int group, const uint64_t* q, const uint8_t* inf, size_t n, const uint64_t* delta, int delta_slot, const uint64_t* vk_point) {
    HostQuery hq;
    const size_t pl = (size_t)pcdhip_point_limbs(cid, group);
    hq.group = group; hq.n = n + 4;
    hq.pts.assign((n + 4) * pl, 0);
    hq.inf.assign(n + 4, 1);
    if (n) memcpy(hq.pts.data(), q, n * pl * 8);
    for (size_t i = 0; i < n; i++) hq.inf[i] = inf ? inf[i] : 0;
    memcpy(hq.pts.data() + (n + delta_slot) * pl, delta, pl * 8);
    hq.inf[n + delta_slot] = 0;
    if (vk_point) { memcpy(hq.pts.data() + (n + 3) * pl, vk_point, pl * 8); hq.inf[n + 3] = 0; }
    return hq;
  };
  HostQuery qa = with_tail(1, h->a_query, h->a_inf, m, h->delta_g1, 0, h->alpha_g1);          // r * delta + alpha
  HostQuery qb1 = with_tail(1, h->b_g1_query, h->b_g1_inf, m, h->delta_g1, 1, h->beta_g1);     // s * delta + beta
  HostQuery qb2 = with_tail(2, h->b_g2_query, h->b_g2_inf, m, h->delta_g2, 1, h->beta_g2);
  HostQuery ql;
  {  // l: padded in front with num_inputs points at infinity, so that it is indexed by the variable like a / b (one
     // sort of the assignment's digits then serves all four MSMs);  -rs * delta in slot 2
    const size_t pl = (size_t)pcdhip_point_limbs(cid, 1);
    std::vector<uint64_t> tmp(m * pl, 0);
    std::vector<uint8_t> tinf(m, 1);
    if (h->l_len) memcpy(tmp.data() + ni * pl, h->l_query, h->l_len * pl * 8);
    for (size_t i = 0; i < h->l_len; i++) tinf[ni + i] = h->l_inf ? h->l_inf[i] : 0;
    ql = with_tail(1, tmp.data(), tinf.data(), m, h->delta_g1, 2, nullptr);
  }
  // one key per device: the whole queries on an ordinary context, the entry range [lo, hi) of the a' / b' / l' queries and the
  // range [hlo, hhi) of the h query on device g of a multi-device context
  auto upload_range = [&](pcdhip_ctx* C, size_t lo, size_t hi, size_t hlo, size_t hhi, pcdhip_g16_pk** res) -> int {
    pcdhip_g16_pk* pk = new pcdhip_g16_pk();
    pk->curve_id = cid; pk->num_vars = m; pk->num_inputs = ni; pk->domain_size = h->domain_size; pk->h_len = h->h_len;
    C->precompute = ctx->precompute; C->precompute_budget = ctx->precompute_budget; C->msm_c = ctx->msm_c;
    auto up = [&](const HostQuery& q, pcdhip_bases** dst) -> int {
      const size_t pl = (size_t)pcdhip_point_limbs(cid, q.group);
      return bases_upload_single(C, cid, q.group, q.pts.data() + lo * pl, q.inf.data() + lo, hi - lo, dst);
    };
    for (size_t i = lo; i < hi && i < m; i++) { pk->a_inf_count += qa.inf[i] != 0; pk->b_inf_count += qb2.inf[i] != 0; }
    // Window bits of a key's queries: one less than the lone MSM's choice for large queries.  A proof runs six MSMs at once and is bound by
    // the sum of their kernels' work; the bucket reductions and fix-ups (proportional to 2^c, a third of that sum at the lone optimum) count
    // in full there, while the lone MSM hides part of them behind its own latency.  Same box, tools/ab_window_step.py: MNT4-298 main proof
    // 17.5 -> 16.8 ms (c = 20 -> 19; 18: 18.2), MNT4-753 160.3 -> 153.0 ms (21 -> 20; 19: 155.8); flat at the help proofs' 2^16 (left alone).
    struct BiasGuard { pcdhip_ctx* c; ~BiasGuard() { c->msm_c_bias = 0; } } bias_guard{C};
    C->msm_c_bias = (hi - lo >= ((size_t)1 << 18)) ? -1 : 0;
    pk->b_inf_same = memcmp(qb1.inf.data() + lo, qb2.inf.data() + lo, hi - lo) == 0;
    int rc = up(qa, &pk->a_query);
    rc = rc ? rc : up(qb1, &pk->b_g1_query);
    rc = rc ? rc : up(qb2, &pk->b_g2_query);
    rc = rc ? rc : up(ql, &pk->l_query);
    const size_t pl1 = (size_t)pcdhip_point_limbs(cid, 1);
    rc = rc ? rc : bases_upload_single(C, cid, 1, h->h_query ? h->h_query + hlo * pl1 : nullptr, h->h_inf ? h->h_inf + hlo : nullptr, hhi - hlo, &pk->h_query);
    if (rc) { pcdhip_g16_pk_free(C, pk); return rc; }
    // A second layout of the four queries over the assignment, for a smaller window (round 5).  The window of a key is fixed by its window-shifted
    // copies, and it is chosen for a DENSE scalar vector: c = 19 / 20 at 2^20 entries, one bucket window of 2^18 / 2^19.  A witness-like
    // assignment leaves a few per cent general scalars, but the fix-up, the first reduction level (two additions per bucket) and the 19 levels
    // over that bucket window cost every one of its four MSMs the same as a dense list: ~1.7 ms of device-filling work per G1 MSM of a proof
    // that takes 8.3 (profiles/r05_witness_like_critical_path.txt).  With copies for a window four bits shorter as well -- what the picker
    // chooses for a sixteenth of the entries -- pcdhip_groth16_prove counts the general scalars and takes the copies that fit the list.
    // Whole keys on an ordinary context only; automatic mode builds them for large keys whose extra copies take at most a quarter of the memory that
    // is free at upload (9.2 GB for a 298-bit key of 2^20 entries, 54 GB for a 753-bit one; pcdhip_set_precompute_budget applies to them as to any vector).
    if (C == ctx && ctx->peers.size() <= 1 && ctx->g16_sparse_window != 0 && pk->a_query && pk->a_query->groups > 1 &&
        (ctx->g16_sparse_window > 0 || hi - lo >= ((size_t)1 << 18) || (field_entry(kCurveFr[cid]).abi_words <= 12 && hi - lo >= ((size_t)1 << 14)))) {
      const int cs = ctx->g16_sparse_window > 0 ? ctx->g16_sparse_window : std::max(8, pk->a_query->c - 4);
      const int Ws = (group_entry(cid, 1).scalar_bits + 1 + cs - 1) / cs;
      const size_t extra = (size_t)Ws * (hi - lo) * (3 * (size_t)group_entry(cid, 1).point_words + (size_t)group_entry(cid, 2).point_words) * 4;
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
      if (cs != pk->a_query->c && (ctx->g16_sparse_window > 0 || extra <= free_b / 4)) {
        const int saved_c = C->msm_c, saved_pre = C->precompute;
        auto drop_sparse = [&]() {
          pcdhip_bases_free(C, pk->a_sparse); pcdhip_bases_free(C, pk->b_g1_sparse); pcdhip_bases_free(C, pk->b_g2_sparse); pcdhip_bases_free(C, pk->l_sparse);
          pk->a_sparse = pk->b_g1_sparse = pk->b_g2_sparse = pk->l_sparse = nullptr;
        };
        // The four layouts must agree in window AND in the number of copies (they share one sorted entry list: MsmSharedSort), while memory (the
        // device's, or pcdhip_set_precompute_budget's bound per vector) grants the wide G2 vector fewer copies than the G1 ones: upload with all
        // copies, then bring everyone down to the smallest count granted (`precompute` = k copies).
        auto up_sparse = [&](int copies) -> int {
          C->msm_c = cs; C->msm_c_bias = 0; C->precompute = copies;
          // (the G2 vector first: it is the widest, so the count it is granted is the one the G1 vectors are then asked for)
          int rs = up(qb2, &pk->b_g2_sparse);
          if (!rs && pk->b_g2_sparse->groups >= 2 && pk->b_g2_sparse->groups < Ws) C->precompute = pk->b_g2_sparse->groups;
          rs = rs ? rs : up(qa, &pk->a_sparse);
          rs = rs ? rs : up(qb1, &pk->b_g1_sparse);
          rs = rs ? rs : up(ql, &pk->l_sparse);
          C->msm_c = saved_c; C->precompute = saved_pre;
          return rs;
        };
        auto sparse_whole = [&]() {
          return pk->a_sparse && pk->b_g1_sparse && pk->b_g2_sparse && pk->l_sparse && pk->a_sparse->groups >= 2 &&
                 pk->a_sparse->c == cs && pk->b_g1_sparse->c == cs && pk->b_g2_sparse->c == cs && pk->l_sparse->c == cs &&
                 pk->a_sparse->groups == pk->b_g1_sparse->groups && pk->a_sparse->groups == pk->b_g2_sparse->groups && pk->a_sparse->groups == pk->l_sparse->groups;
        };
        auto try_sparse = [&]() -> bool {
          int rs = up_sparse(-1);
          if (!rs && !sparse_whole() && pk->a_sparse && pk->b_g1_sparse && pk->b_g2_sparse && pk->l_sparse) {
            const int k = std::min(std::min(pk->a_sparse->groups, pk->b_g1_sparse->groups), std::min(pk->b_g2_sparse->groups, pk->l_sparse->groups));
            drop_sparse();
            if (k >= 2) rs = up_sparse(k);
          }
          if (rs || !sparse_whole()) { drop_sparse(); (void)hipGetLastError(); return false; }   // (the key works without them)
          return true;
        };
        bool have = try_sparse();
        // A window asked for OUTRIGHT (bits > 0) when the ordinary copies have already taken the memory: fewer ordinary copies + the second layout
        // instead of all ordinary copies and none -- the ordinary queries are uploaded again with half the copies (a Horner combine over the
        // windows that share a copy comes back for dense assignments) until the second layout fits beside them.
        for (int round = 0; !have && ctx->g16_sparse_window > 0 && round < 4; round++) {
          const int k = std::min(std::min(pk->a_query->groups, pk->b_g1_query->groups), std::min(pk->b_g2_query->groups, pk->l_query->groups)) / 2;
          if (k < 2) break;
          pcdhip_bases_free(C, pk->a_query); pcdhip_bases_free(C, pk->b_g1_query); pcdhip_bases_free(C, pk->b_g2_query); pcdhip_bases_free(C, pk->l_query);
          pk->a_query = pk->b_g1_query = pk->b_g2_query = pk->l_query = nullptr;
          C->precompute = k; C->msm_c_bias = (hi - lo >= ((size_t)1 << 18)) ? -1 : 0;
          int ro = up(qa, &pk->a_query);
          ro = ro ? ro : up(qb1, &pk->b_g1_query);
          ro = ro ? ro : up(qb2, &pk->b_g2_query);
          ro = ro ? ro : up(ql, &pk->l_query);
          C->precompute = saved_pre;
          if (ro) { pcdhip_g16_pk_free(C, pk); return ro; }
          have = try_sparse();
        }
      }
    }
    *res = pk;
    return PCDHIP_OK;
  };
  *out = nullptr;
  if (ctx->peers.size() <= 1) return upload_range(ctx, 0, m + 4, 0, h->h_len, out);
  pcdhip_g16_pk* parent = new pcdhip_g16_pk();
  parent->curve_id = cid; parent->num_vars = m; parent->num_inputs = ni; parent->domain_size = h->domain_size; parent->h_len = h->h_len;
  const size_t G = ctx->peers.size();
  parent->lo.resize(G + 1); parent->hlo.resize(G + 1);
  for (size_t g = 0; g < G; g++) {
    size_t lo, hi, hlo, hhi;
    shard_range(m + 4, g, G, &lo, &hi);
    shard_range(h->h_len, g, G, &hlo, &hhi);
    parent->lo[g] = lo; parent->lo[g + 1] = hi; parent->hlo[g] = hlo; parent->hlo[g + 1] = hhi;
    pcdhip_g16_pk* sh = nullptr;
    int rc = upload_range(ctx->peers[g], lo, hi, hlo, hhi, &sh);
    if (rc) { pcdhip_g16_pk_free(ctx, parent); return rc; }
    parent->shards.push_back(sh);
  }
  *out = parent;
  return PCDHIP_OK;
  });
}
int pcdhip_g16_pk_set_r1cs(pcdhip_ctx* ctx, pcdhip_g16_pk* pk, const pcdhip_csr* A, const pcdhip_csr* B, const pcdhip_csr* C) {
  if (!ctx || !pk || !A || !B || !C) return PCDHIP_E_ARG;
  if (!pk->shards.empty()) {
    // the witness map's three chains (a, b, c: mat-vec, ifft, coset fft each) run on the first three devices of a sharded key
    // (SURVEY.md 8e), the pointwise step and the last transform on device 0: the matrices are resident wherever a chain runs
    int rc = pcdhip_g16_pk_set_r1cs(ctx->peers.empty() ? ctx : ctx->peers[0], pk->shards[0], A, B, C);
    for (size_t g = 1; !rc && g < 3 && g < pk->shards.size() && g < ctx->peers.size(); g++)
      rc = pcdhip_g16_pk_set_r1cs(ctx->peers[g], pk->shards[g], A, B, C);
    return rc;
  }
  if (A->num_rows != B->num_rows || A->num_rows != C->num_rows || (A->num_rows >> 31)) return PCDHIP_E_ARG;
  if (!A->row_ptr || !B->row_ptr || !C->row_ptr) return PCDHIP_E_ARG;
  BIND();
  const FieldEntry& fe = field_entry(kCurveFr[pk->curve_id]);
  const pcdhip_csr* ms[3] = {A, B, C};
  for (int k = 0; k < 3; k++) { int rc = validate_csr(ms[k], pk->num_vars); if (rc) return rc; }
  size_t total = 0, base[3], off[6];
  for (int k = 0; k < 3; k++) { base[k] = total; total += csr_bytes(ms[k], fe, off) + 64; }
  if (pk->r1cs_dev) { (void)hipFree(pk->r1cs_dev); pk->r1cs_dev = nullptr; }
  TRY(hipMalloc(&pk->r1cs_dev, total + 64));
  for (int k = 0; k < 3; k++) {
    DevCsr dc;
    int rc = upload_csr_to(ctx, ms[k], fe, pk->num_vars, (char*)pk->r1cs_dev + base[k], &dc);
    if (rc) return rc;
    pk->mats[k] = dc;
  }
  pk->rows = (uint32_t)A->num_rows;
  TRY(hipStreamSynchronize(ctx->stream));
  return PCDHIP_OK;
}
void pcdhip_g16_pk_free(pcdhip_ctx* ctx, pcdhip_g16_pk* pk) {
  if (!pk) return;
  for (size_t g = 0; g < pk->shards.size(); g++) pcdhip_g16_pk_free(ctx && g < ctx->peers.size() ? ctx->peers[g] : ctx, pk->shards[g]);
  pk->shards.clear();
  if (ctx) (void)hipSetDevice(ctx->device);
  if (pk->r1cs_dev) (void)hipFree(pk->r1cs_dev);
  pcdhip_bases_free(ctx, pk->a_query); pcdhip_bases_free(ctx, pk->b_g1_query); pcdhip_bases_free(ctx, pk->b_g2_query);
  pcdhip_bases_free(ctx, pk->h_query); pcdhip_bases_free(ctx, pk->l_query);
  pcdhip_bases_free(ctx, pk->a_sparse); pcdhip_bases_free(ctx, pk->b_g1_sparse); pcdhip_bases_free(ctx, pk->b_g2_sparse); pcdhip_bases_free(ctx, pk->l_sparse);
  delete pk;
}

namespace {
// scalars of `can` (n canonical values of sw words) that are neither 0 nor 1 (grid-stride, one atomic per workgroup: the count is read back by the
// host before the proof's MSMs are launched, so its latency is on the proof's path -- one atomic per wave on one address took 110 us at 2^20)
__global__ void __launch_bounds__(256) count_general_kernel(const uint32_t* __restrict__ can, uint32_t n, int sw, uint32_t* __restrict__ out) {
  __shared__ uint32_t block_count;
  if (threadIdx.x == 0) block_count = 0;
  __syncthreads();
  uint32_t mine = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t* p = can + (size_t)i * sw;
    uint32_t hi = 0;
    for (int k = 1; k < sw; k++) hi |= p[k];
    mine += (hi != 0 || p[0] > 1u) ? 1u : 0u;
  }
  if (mine) atomicAdd(&block_count, mine);
  __syncthreads();
  if (threadIdx.x == 0 && block_count) atomicAdd(out, block_count);
}
// One device's share of a Groth16 proof: the scalar vectors (assignment z followed by the tail [r, s, -rs, 1], canonical words,
// and their s / r multiples), the MSMs over an entry range of the device's queries, their results.  An ordinary context runs one
// of these over the whole key; a multi-device context one per device over that device's ranges.
struct G16Run {
  pcdhip_ctx* ctx;            // the device's context
  const pcdhip_g16_pk* pk;    // the device's key (whole queries, or this device's shard)
  int cid = 0;
  size_t m = 0, sw = 0, sb = 0, j1 = 0, j2 = 0;
  uint32_t *z_dev = nullptr, *z_can = nullptr, *sz_can = nullptr, *rz_can = nullptr, *rs_dev = nullptr, *t1 = nullptr, *h_can = nullptr;
  uint32_t *msm_g1 = nullptr, *msm_g2 = nullptr, *mul_scratch = nullptr, *proof_dev = nullptr;
  struct Job { const GroupEntry* ge; MsmBasesView bv; const uint32_t* sc; uint32_t n; uint32_t* out; int tslot; const uint32_t* k; uint32_t* kout; int share; };
  static constexpr int SHARE_B = 4;   // Job::share = role (MSM_SHARE_*) | SHARE_B: the sort shared by the two B MSMs (g16_share_b)
  Job jobs[6];
  int nj = 0;
  bool use_lane = false;      // the jobs' accumulate kernels run one after the other on the context's lane (pcdhip_groth16_set_schedule 2)
  bool sparse = false;        // the assignment MSMs take the key's sparse-window copies (decide_sparse)
  uint32_t* general_dev = nullptr;
  uint32_t* slot(int i) const { return (uint32_t*)((char*)msm_g1 + (size_t)i * j1); }
  size_t partial_bytes() const { return 6 * j1 + j2; }  // msm_g1 (six slots) and msm_g2 are contiguous

  // buffers, z -> device image and canonical words (plain, times s, times r), the scalar tails; ends with g16_ready recorded
  // (need_scaled: the folded assembly multiplies every scalar by s / r -- the chained form does not read those copies, 2 n products less)
  int prepare(const uint64_t* z, const uint64_t* r_mont, const uint64_t* s_mont, size_t n_dom, bool need_scaled = true) {
    BIND();
    cid = pk->curve_id;
    const FieldEntry& fe = field_entry(kCurveFr[cid]);
    m = pk->num_vars;
    sw = (size_t)fe.abi_words;  // words of a canonical scalar == words of an ABI element
    sb = sw * 4;
    hipStream_t st = ctx->stream;
    TRY(ctx->aux_ws.ensure(AUX_Z, m * fe.words * 4));
    TRY(ctx->aux_ws.ensure(AUX_SCAL, m * sb));
    TRY(ctx->aux_ws.ensure(AUX_Z_CANON, 3 * (m + 4) * sb + 2 * sb + 64));
    TRY(ctx->aux_ws.ensure(AUX_H_CANON, n_dom * sb));
    z_dev = (uint32_t*)ctx->aux_ws.buf[AUX_Z];
    uint32_t* z_abi = (uint32_t*)ctx->aux_ws.buf[AUX_SCAL];
    z_can = (uint32_t*)ctx->aux_ws.buf[AUX_Z_CANON];
    sz_can = z_can + (m + 4) * sw;
    rz_can = sz_can + (m + 4) * sw;
    rs_dev = rz_can + (m + 4) * sw;  // r, s (C-ABI Montgomery)
    general_dev = rs_dev + 2 * sw;   // one word: the count of general scalars (decide_sparse)
    h_can = (uint32_t*)ctx->aux_ws.buf[AUX_H_CANON];
    TRY(hipMemcpyAsync(z_abi, z, m * sb, hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(rs_dev, r_mont, sb, hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(rs_dev + sw, s_mont, sb, hipMemcpyHostToDevice, st));
    TRY(fe.convert(st, z_abi, z_dev, (uint32_t)m, 0));
    TRY(fe.scale_canon(st, z_dev, nullptr, z_can, (uint32_t)m, 1));
    if (need_scaled) {
      TRY(fe.scale_canon(st, z_dev, rs_dev + sw, sz_can, (uint32_t)m, 1));
      TRY(fe.scale_canon(st, z_dev, rs_dev, rz_can, (uint32_t)m, 1));
    }
    const GroupEntry& g1 = group_entry(cid, 1);
    const GroupEntry& g2 = group_entry(cid, 2);
    j1 = (size_t)g1.point_words / 2 * 3 * 4;
    j2 = (size_t)g2.point_words / 2 * 3 * 4;
    const CurveEntry& ce = curve_entry(cid);
    TRY(ctx->aux_ws.ensure(AUX_G16, 6 * j1 + j2 + 6 * 400 * j1 + ce.proof_abi_bytes + 256));  // (400 Jacobian slots per one-point product: inst_g16.hip)
    char* gbase = (char*)ctx->aux_ws.buf[AUX_G16];
    msm_g1 = (uint32_t*)gbase;  // h, l', A, s*A, r*B_1, B_1
    msm_g2 = (uint32_t*)(gbase + 6 * j1);
    mul_scratch = (uint32_t*)(gbase + 6 * j1 + j2);
    proof_dev = (uint32_t*)(gbase + 6 * j1 + j2 + 6 * 400 * j1);
    t1 = z_can + m * sw;  // [r, s, -rs, 1] canonical
    TRY(ce.prepare_scalars(st, rs_dev, t1, sz_can + m * sw, rz_can + m * sw));
    { int rc = ensure_side_streams(ctx); if (rc) return rc; }
    if (pipe_pending(ctx)) return PCDHIP_E_ARG;  // submitted MSMs still own side-stream workspaces: collect them first
    TRY(hipEventRecord(ctx->g16_ready, st));  // the scalars z || t (and their scaled copies) are ready
    if (!ctx->g16_share.ready) TRY(hipEventCreateWithFlags(&ctx->g16_share.ready, hipEventDisableTiming));
    if (!ctx->g16_share_b.ready) TRY(hipEventCreateWithFlags(&ctx->g16_share_b.ready, hipEventDisableTiming));
    ctx->g16_share.valid = false;
    ctx->g16_share_b.valid = false;
    nj = 0;
    use_lane = ctx->g16_schedule == 2 && ctx->lane.stream;
    return PCDHIP_OK;
  }
  // Which copies the four assignment MSMs run on: the sparse-window ones when at most an eighth of the assignment is general (neither 0 nor 1).
  // One small kernel over the canonical scalars and a 4-byte read-back: the only point where pcdhip_groth16_prove waits for the device before
  // its last launch (the upload is done by then; everything behind it is queued within 0.4 ms, profiles/r05_prove_host_enqueue.txt).
  int decide_sparse(bool folded) {
    sparse = false;
    ctx->g16_last_sparse = 0; ctx->g16_last_general = 0;
    if (folded || !pk->a_sparse) return PCDHIP_OK;   // (the folded form multiplies every scalar by s / r: nothing stays 0 / 1 but the zeros)
    BIND();
    hipStream_t st = ctx->stream;
    TRY(hipMemsetAsync(general_dev, 0, 4, st));
    hipLaunchKernelGGL(count_general_kernel, dim3((unsigned)std::min<size_t>((m + 255) / 256, 1024)), dim3(256), 0, st, z_can, (uint32_t)m, (int)sw, general_dev);
    // (a page-locked word and an event of its own instead of a pageable copy + a synchronisation of the whole stream -- the host
    //  waits for exactly the count kernel and its 4-byte copy, whatever else a caller has queued behind them on this stream)
    if (!ctx->count_host) TRY(hipHostMalloc((void**)&ctx->count_host, 64, hipHostMallocDefault));
    if (!ctx->count_ev) TRY(hipEventCreateWithFlags(&ctx->count_ev, hipEventDisableTiming));
    TRY(hipMemcpyAsync(ctx->count_host, general_dev, 4, hipMemcpyDeviceToHost, st));
    TRY(hipEventRecord(ctx->count_ev, st));
    TRY(hipEventSynchronize(ctx->count_ev));
    const uint32_t general = *(volatile uint32_t*)ctx->count_host;
    ctx->g16_last_general = general;
    sparse = (uint64_t)general * 8 <= m;
    ctx->g16_last_sparse = sparse ? 1 : 0;
    return PCDHIP_OK;
  }
  // the s z / r z copies of the folded form, when prepare() was told to leave them out (the form is chosen after the count)
  int scale_for_fold() {
    BIND();
    const FieldEntry& fe = field_entry(kCurveFr[cid]);
    hipStream_t st = ctx->stream;
    TRY(fe.scale_canon(st, z_dev, rs_dev + sw, sz_can, (uint32_t)m, 1));
    TRY(fe.scale_canon(st, z_dev, rs_dev, rz_can, (uint32_t)m, 1));
    TRY(hipEventRecord(ctx->g16_ready, st));
    return PCDHIP_OK;
  }
  int launch(int k, hipEvent_t after) {
    BIND();
    const CurveEntry& ce = curve_entry(cid);
    hipStream_t sk = ctx->g16_streams[k];
    TRY(hipStreamWaitEvent(sk, after, 0));
    TRY(hipEventRecord(ctx->g16_begin[k], sk));
    // (schedule 2: the accumulation goes to the lane, in the order the jobs are launched)
    ctx->g16_ws[k].lane = use_lane ? &ctx->lane : nullptr;
    const hipError_t me = jobs[k].ge->msm(ctx->g16_ws[k], sk, jobs[k].bv, jobs[k].sc, jobs[k].n, jobs[k].out, ctx->msm_c, ctx->msm_chunk, ctx->msm_sort, nullptr,
                        (jobs[k].share & 3) ? ((jobs[k].share & SHARE_B) ? &ctx->g16_share_b : &ctx->g16_share) : nullptr, jobs[k].share & 3);
    ctx->g16_ws[k].lane = nullptr;
    TRY(me);
    if (jobs[k].k) TRY(ce.scale_g1(sk, jobs[k].out, jobs[k].k, mul_scratch + (size_t)k * 400 * (j1 / 4), jobs[k].kout));
    TRY(hipEventRecord(ctx->g16_end[k], sk));
    static const bool serial = getenv("PCDHIP_G16_SERIAL") != nullptr;  // developer knob: every MSM of a proof with the device to itself (kernel traces)
    if (serial) TRY(hipDeviceSynchronize());
    return PCDHIP_OK;
  }
  // The MSMs that take the assignment (l', A, B_1 on G1; B on G2) over the entries [lo, lo + cnt) of z || t, which are the
  // entries [0, cnt) of this device's queries.  a', b1', b2', l' are indexed alike and take the same scalars: the first job sorts, the
  // others reuse its list.  The two variable-base products of the assembly, s*A and r*B_1, are either one-lane products queued right
  // behind the A / B_1 MSMs on their high-priority streams (whole-key runs only), or two more MSMs over the same bases with every
  // scalar scaled by s / r (`folded`; the form that also adds up across devices).
  int launch_assignment(size_t lo, size_t cnt, bool folded, hipEvent_t after = nullptr) {
    const GroupEntry& g1 = group_entry(cid, 1);
    const GroupEntry& g2 = group_entry(cid, 2);
    const int PRODUCE = MSM_SHARE_PRODUCE, CONSUME = MSM_SHARE_CONSUME, NONE = MSM_SHARE_NONE;
    const uint32_t n = (uint32_t)cnt;
    const uint32_t *zc = z_can + lo * sw, *szc = sz_can + lo * sw, *rzc = rz_can + lo * sw;
    // Which MSMs share a sort.  The a / b queries of a real key are full of points at infinity (a variable that no row of A / B mentions),
    // and an MSM that sorts for itself leaves their entries out of its bucket lists (MsmBasesView::inf_bits).  A sort that is SHARED must
    // keep every entry its consumers need: the list made for A and l' drops nothing (views without the bitmap); the two B MSMs, which flag
    // the same entries, get a sort of their own when enough of them are infinite to pay for it (a sort is ~15 % of a G1 MSM, ~5 % of a G2
    // one; the accumulations shrink by the infinite fraction).
    auto plain = [](MsmBasesView v) { v.inf_bits = nullptr; return v; };
    const MsmBasesView va = (sparse ? pk->a_sparse : pk->a_query)->view(0), vb1 = (sparse ? pk->b_g1_sparse : pk->b_g1_query)->view(0),
                       vb2 = (sparse ? pk->b_g2_sparse : pk->b_g2_query)->view(0), vl = (sparse ? pk->l_sparse : pk->l_query)->view(0);
    const bool sparse_b = pk->b_inf_count * 16 > cnt;
    if (folded) {
      // (A leaves the shared list for a sort of its own when enough of its entries are infinite AND the MSM is large: at 2^16 the sort costs
      //  what the shorter accumulation saves)
      const bool a_alone = pk->a_inf_count * 16 > cnt && cnt >= ((size_t)1 << 17);   // (measured, folded form: 298-bit 2^20 neutral, 753-bit 2^20 185 -> 180 ms)
      if (sparse_b) {
        jobs[nj++] = {&g2, vb2, zc, n, msm_g2, 5, nullptr, nullptr, NONE};                                   // B (heaviest: high priority), its own sort
        if (a_alone) jobs[nj++] = {&g1, va, zc, n, slot(2), 3, nullptr, nullptr, NONE};                      // A
        else jobs[nj++] = {&g1, plain(va), zc, n, slot(2), 3, nullptr, nullptr, PRODUCE};
      } else {
        jobs[nj++] = {&g2, plain(vb2), zc, n, msm_g2, 5, nullptr, nullptr, PRODUCE};                         // B
        if (a_alone) jobs[nj++] = {&g1, va, zc, n, slot(2), 3, nullptr, nullptr, NONE};                      // A
        else jobs[nj++] = {&g1, plain(va), zc, n, slot(2), 3, nullptr, nullptr, CONSUME};
      }
      jobs[nj++] = {&g1, va, szc, n, slot(3), 3, nullptr, nullptr, NONE};                                    // s * A
      jobs[nj++] = {&g1, vb1, rzc, n, slot(4), 4, nullptr, nullptr, NONE};                                   // r * B_1
      jobs[nj++] = {&g1, plain(vl), zc, n, slot(1), 2, nullptr, nullptr, (sparse_b && a_alone) ? NONE : CONSUME};   // l' (with -rs delta)
    } else {
      // (A sorts for itself when enough of its entries are infinite -- l' then does too: a sort is HBM and atomics and overlaps the other
      //  MSMs' multiply-adds, the accumulation it shortens does not; measured on the bench's 298-bit main proof: 17.2 -> 17.0 ms)
      const bool sparse_a = pk->a_inf_count * 16 > cnt;
      if (sparse_a) jobs[nj++] = {&g1, va, zc, n, slot(2), 3, t1 + sw, slot(3), NONE};                       // A, then s * A
      else jobs[nj++] = {&g1, plain(va), zc, n, slot(2), 3, t1 + sw, slot(3), PRODUCE};
      if (sparse_b && pk->b_inf_same) {
        jobs[nj++] = {&g1, vb1, zc, n, slot(5), 4, t1, slot(4), PRODUCE | SHARE_B};                          // B_1, then r * B_1
        // (b_g1 and b_g2 flag the same entries -- compared byte for byte at upload -- so B names B_1's bitmap: the shared list's identity,
        //  which msm_run checks before it consumes a filtered list)
        MsmBasesView vb2s = vb2;
        vb2s.inf_bits = vb1.inf_bits;
        jobs[nj++] = {&g2, vb2s, zc, n, msm_g2, 5, nullptr, nullptr, CONSUME | SHARE_B};                     // B
      } else if (sparse_b) {
        jobs[nj++] = {&g1, vb1, zc, n, slot(5), 4, t1, slot(4), NONE};
        jobs[nj++] = {&g2, vb2, zc, n, msm_g2, 5, nullptr, nullptr, NONE};
      } else {   // (the full list: made by A, or by B_1 when A sorts for itself)
        jobs[nj++] = {&g1, plain(vb1), zc, n, slot(5), 4, t1, slot(4), sparse_a ? PRODUCE : CONSUME};
        jobs[nj++] = {&g2, plain(vb2), zc, n, msm_g2, 5, nullptr, nullptr, CONSUME};
      }
      jobs[nj++] = {&g1, plain(vl), zc, n, slot(1), 2, nullptr, nullptr, (sparse_a && sparse_b) ? NONE : CONSUME};   // l'
    }
    // (round 5's diagnosis of what a proof waits for left jobs out here behind an environment variable -- profiles/r05_witness_like_critical_path.txt;
    //  a switch that makes proofs wrong does not stay in the library)
    for (int k = 0; k < nj; k++) { int rc = launch(k, after ? after : ctx->g16_ready); if (rc) return rc; }
    return PCDHIP_OK;
  }
  // ... behind `gate`: with the lane only the accumulations wait for it (the sorts start at once), otherwise the MSMs as a whole
  int launch_assignment_gated(size_t lo, size_t cnt, bool folded, hipEvent_t gate) {
    if (!use_lane) return launch_assignment(lo, cnt, folded, gate);
    BIND();
    TRY(hipStreamWaitEvent(ctx->lane.stream, gate, 0));
    return launch_assignment(lo, cnt, folded);
  }
  // the h MSM over the entries [hlo, hlo + cnt) of h (entries [0, cnt) of this device's h query), once `after` has fired
  int launch_h(size_t hlo, size_t cnt, hipEvent_t after) {
    const GroupEntry& g1 = group_entry(cid, 1);
    jobs[nj] = {&g1, pk->h_query->view(0), h_can + hlo * sw, (uint32_t)cnt, slot(0), 1, nullptr, nullptr, MSM_SHARE_NONE};
    return launch(nj++, after);
  }
  // the context's stream waits for every job
  int join() {
    BIND();
    for (int k = 0; k < nj; k++) TRY(hipStreamWaitEvent(ctx->stream, ctx->g16_end[k], 0));
    return PCDHIP_OK;
  }
};

int finish_proof(pcdhip_ctx* ctx, int cid, const uint32_t* proof_dev, uint64_t* proof_out, uint8_t* inf_out) {
  const GroupEntry& g1 = group_entry(cid, 1);
  const GroupEntry& g2 = group_entry(cid, 2);
  TRY(hipMemcpyAsync(proof_out, proof_dev, curve_entry(cid).proof_abi_bytes, hipMemcpyDeviceToHost, ctx->stream));
  TRY(hipStreamSynchronize(ctx->stream));
  if (inf_out) {
    const size_t a1 = (size_t)g1.point_abi_words * 4, a2 = (size_t)g2.point_abi_words * 4;
    const size_t offs[3] = {0, a1 / 8, (a1 + a2) / 8}, lens[3] = {a1 / 8, a2 / 8, a1 / 8};
    for (int i = 0; i < 3; i++) {
      uint64_t o = 0;
      for (size_t k = 0; k < lens[i]; k++) o |= proof_out[offs[i] + k];
      inf_out[i] = (o == 0) ? 1 : 0;
    }
  }
  return PCDHIP_OK;
}

// A proof over a key sharded across the devices of the context (BASELINE configs[4]: the merge node's MSMs on all GPUs).  Every
// device gets z and runs the five assignment MSMs over its entry range with the two assembly products folded in (partial sums add
// up across devices; one-lane products of partial results would not); device 0 computes h and hands each device its slice over
// xGMI; the partial results (six G1 points and one G2 point per device) travel to device 0, one lane per MSM sums them, and the
// ordinary assembly kernel finishes.  No host synchronisation until the proof is read back.
int prove_sharded_impl(pcdhip_ctx* ctx, const pcdhip_g16_pk* pk, const pcdhip_csr* A, const pcdhip_csr* B, const pcdhip_csr* C, const uint64_t* z,
                       const uint64_t* r_mont, const uint64_t* s_mont, uint64_t* proof_out, uint8_t* inf_out);
int prove_sharded(pcdhip_ctx* ctx, const pcdhip_g16_pk* pk, const pcdhip_csr* A, const pcdhip_csr* B, const pcdhip_csr* C, const uint64_t* z,
                  const uint64_t* r_mont, const uint64_t* s_mont, uint64_t* proof_out, uint8_t* inf_out) {
  const int rc = prove_sharded_impl(ctx, pk, A, B, C, z, r_mont, s_mont, proof_out, inf_out);
  if (rc) drain_peers(ctx);  // no device is left reading the caller's z / r / s, or writing into this call's buffers
  return rc;
}
int prove_sharded_impl(pcdhip_ctx* ctx, const pcdhip_g16_pk* pk, const pcdhip_csr* A, const pcdhip_csr* B, const pcdhip_csr* C, const uint64_t* z,
                       const uint64_t* r_mont, const uint64_t* s_mont, uint64_t* proof_out, uint8_t* inf_out) {
  const size_t G = pk->shards.size();
  if (ctx->peers.size() != G) return PCDHIP_E_ARG;
  ctx->g16_last_sparse = 0; ctx->g16_last_general = 0;   // (a sharded proof folds the assembly products in: it never counts, never takes a second layout)
  const int cid = pk->curve_id, fr = kCurveFr[cid];
  const FieldEntry& fe = field_entry(fr);
  const size_t m = pk->num_vars, ni = pk->num_inputs;
  const pcdhip_g16_pk* pk0 = pk->shards[0];
  uint32_t rows = 0;
  if (A && B && C) rows = (uint32_t)A->num_rows;
  else if (!A && !B && !C && pk0->r1cs_dev) rows = pk0->rows;
  else return PCDHIP_E_ARG;
  Dom dom;
  int rc = pick_domain(fr, (size_t)rows + ni, &dom);
  if (rc) return rc;
  const size_t n = dom.n, hl = std::min<size_t>(pk->h_len, n);
  BIND();
  EventSet<3> ev;
  TRY(ev.create());
  TRY(hipEventRecord(ev[0], ctx->stream));
  const bool split3 = G >= 3 && !(A && B && C) && ctx->wm_split && pk->shards[1]->r1cs_dev && pk->shards[2]->r1cs_dev &&
                      pk->shards[1]->rows == rows && pk->shards[2]->rows == rows;
  // (pcdhip_groth16_set_schedule 1, off by default: measured slower) a device that carries a chain of the witness map starts its MSMs only
  // when that chain is through
  const bool map_first = ctx->g16_schedule == 1 || ctx->g16_schedule == 2;  // (2: only the accumulate lane of such a device waits for its chain)
  auto carries_chain = [&](size_t g) { return g == 0 || (split3 && g <= 2); };
  std::vector<G16Run> runs(G);
  for (size_t g = 0; g < G; g++) {
    runs[g].ctx = ctx->peers[g];
    runs[g].pk = pk->shards[g];
    ctx->peers[g]->msm_sort = ctx->msm_sort;
    rc = runs[g].prepare(z, r_mont, s_mont, n);
    if (rc) return rc;
    if (map_first && carries_chain(g)) continue;
    rc = runs[g].launch_assignment(pk->lo[g], pk->lo[g + 1] - pk->lo[g], true);
    if (rc) return rc;
  }
  // h on device 0 while the MSMs above run everywhere
  BIND();
  DevCsr mats[3];
  if (A && B && C) { TRY(hipStreamSynchronize(ctx->stream)); rc = upload_three(ctx, A, B, C, fe, m, mats); if (rc) return rc; }
  else for (int k = 0; k < 3; k++) mats[k] = pk0->mats[k];
  if (split3) {
    // chains b and c on devices 1 and 2 (their z is resident since prepare()), each into its own AUX_A, then device to device into
    // device 0's AUX_B / AUX_C; chain a on device 0 meanwhile; the pointwise step and the last transform on device 0
    const size_t vb = (size_t)dom.n * fe.words * 4;
    TRY(ctx->aux_ws.ensure(AUX_B, vb));
    TRY(ctx->aux_ws.ensure(AUX_C, vb));
    for (size_t g = 1; g <= 2; g++) {
      pcdhip_ctx* Cg = ctx->peers[g];
      TRY(hipSetDevice(Cg->device));
      rc = witness_chain_dev(Cg, fr, pk->shards[g]->mats[g], (int)g, runs[g].z_dev, ni, dom, AUX_A);
      if (rc) return rc;
      TRY(hipMemcpyPeerAsync(ctx->aux_ws.buf[g == 1 ? AUX_B : AUX_C], ctx->device, Cg->aux_ws.buf[AUX_A], Cg->device, vb, Cg->stream));
      if (!Cg->wm_ev) TRY(hipEventCreateWithFlags(&Cg->wm_ev, hipEventDisableTiming));
      TRY(hipEventRecord(Cg->wm_ev, Cg->stream));
      if (map_first) { rc = runs[g].launch_assignment_gated(pk->lo[g], pk->lo[g + 1] - pk->lo[g], true, Cg->wm_ev); if (rc) return rc; }
    }
    BIND();
    rc = witness_chain_dev(ctx, fr, mats[0], 0, runs[0].z_dev, ni, dom, AUX_A);
    if (rc) return rc;
    for (size_t g = 1; g <= 2; g++) TRY(hipStreamWaitEvent(ctx->stream, ctx->peers[g]->wm_ev, 0));
    rc = witness_finish_dev(ctx, fr, dom);
    if (rc) return rc;
  } else {
    Dom dom_used;
    rc = witness_map_dev(ctx, fr, mats, runs[0].z_dev, ni, &dom_used);
    if (rc) return rc;
    if (dom_used.n != dom.n) return PCDHIP_E_ARG;
  }
  TRY(fe.convert(ctx->stream, (const uint32_t*)ctx->aux_ws.buf[AUX_A], runs[0].h_can, (uint32_t)n, 2));
  TRY(hipEventRecord(ev[1], ctx->stream));
  if (map_first) { rc = runs[0].launch_assignment_gated(pk->lo[0], pk->lo[1] - pk->lo[0], true, ev[1]); if (rc) return rc; BIND(); }
  const size_t sw = runs[0].sw, sb = runs[0].sb;
  for (size_t g = 0; g < G; g++) {
    const size_t hlo = std::min(pk->hlo[g], hl), hhi = std::min(pk->hlo[g + 1], hl);
    pcdhip_ctx* Cg = ctx->peers[g];
    hipEvent_t after = ev[1];
    if (g > 0) {  // this device's slice of h, device to device
      TRY(hipSetDevice(Cg->device));
      TRY(hipStreamWaitEvent(Cg->stream, ev[1], 0));
      if (hhi > hlo)
        TRY(hipMemcpyPeerAsync(runs[g].h_can + hlo * sw, Cg->device, runs[0].h_can + hlo * sw, ctx->device, (hhi - hlo) * sb, Cg->stream));
      if (!Cg->xstream_ev) TRY(hipEventCreateWithFlags(&Cg->xstream_ev, hipEventDisableTiming));
      TRY(hipEventRecord(Cg->xstream_ev, Cg->stream));
      after = Cg->xstream_ev;
    }
    rc = runs[g].launch_h(hlo, hhi - hlo, after);
    if (rc) return rc;
  }
  // partial results -> device 0
  const size_t pb = runs[0].partial_bytes(), j1 = runs[0].j1;
  BIND();
  TRY(ctx->aux_ws.ensure(AUX_FB_JAC, (G + 1) * pb + 64));
  char* gather = (char*)ctx->aux_ws.buf[AUX_FB_JAC];  // G partial blocks, then the summed block
  for (size_t g = 0; g < G; g++) {
    pcdhip_ctx* Cg = ctx->peers[g];
    rc = runs[g].join();
    if (rc) return rc;
    TRY(hipSetDevice(Cg->device));
    TRY(hipMemcpyPeerAsync(gather + g * pb, ctx->device, runs[g].msm_g1, Cg->device, pb, Cg->stream));
    if (g > 0) {
      TRY(hipEventRecord(Cg->xstream_ev, Cg->stream));
      TRY(hipSetDevice(ctx->device));
      TRY(hipStreamWaitEvent(ctx->stream, Cg->xstream_ev, 0));
    }
  }
  BIND();
  uint32_t* sum = (uint32_t*)(gather + G * pb);
  TRY(group_entry(cid, 1).jac_sum_parts(ctx->stream, (const uint32_t*)gather, pb / 4, (uint32_t)G, 5, sum));               // h, l', A, s*A, r*B_1
  TRY(group_entry(cid, 2).jac_sum_parts(ctx->stream, (const uint32_t*)(gather + 6 * j1), pb / 4, (uint32_t)G, 1, sum + 6 * j1 / 4));  // B
  TRY(curve_entry(cid).assemble(ctx->stream, sum, sum + 6 * j1 / 4, runs[0].proof_dev));
  TRY(hipEventRecord(ev[2], ctx->stream));
  rc = finish_proof(ctx, cid, runs[0].proof_dev, proof_out, inf_out);
  if (rc) return rc;
  for (size_t g = 1; g < G; g++) { TRY(hipSetDevice(ctx->peers[g]->device)); TRY(hipStreamSynchronize(ctx->peers[g]->stream)); }
  BIND();
  for (int k = 0; k < 7; k++) ctx->g16_ms[k] = 0;
  (void)hipEventElapsedTime(&ctx->g16_ms[0], ev[0], ev[1]);
  (void)hipEventElapsedTime(&ctx->g16_ms[7], ev[0], ev[2]);
  return PCDHIP_OK;
}
}  // namespace

int pcdhip_groth16_prove(pcdhip_ctx* ctx, const pcdhip_g16_pk* pk, const pcdhip_csr* A, const pcdhip_csr* B, const pcdhip_csr* C,
                         const uint64_t* z, const uint64_t* r_mont, const uint64_t* s_mont, uint64_t* proof_out, uint8_t* inf_out) {
  return guarded([&]() -> int {
  if (!ctx || !pk || !z || !r_mont || !s_mont || !proof_out) return PCDHIP_E_ARG;
  if (A && B && C) { const pcdhip_csr* ms[3] = {A, B, C}; for (int k = 0; k < 3; k++) { int v = validate_csr(ms[k], pk->num_vars); if (v) return v; } }
  if (pipe_pending(ctx)) return PCDHIP_E_ARG;  // submitted MSMs still own side-stream workspaces: collect them first (before anything is queued)
  if (!pk->shards.empty()) return prove_sharded(ctx, pk, A, B, C, z, r_mont, s_mont, proof_out, inf_out);
  if (!ctx->peers.empty() && ctx->peers.size() > 1) return PCDHIP_E_ARG;  // a whole-key handle belongs to an ordinary context
  BIND();
  const int cid = pk->curve_id, fr = kCurveFr[cid];
  const FieldEntry& fe = field_entry(fr);
  const size_t m = pk->num_vars, ni = pk->num_inputs;
  hipStream_t st = ctx->stream;
  EventSet<8> ev;
  TRY(ev.create());
  TRY(hipEventRecord(ev[0], st));
  static const bool host_time = getenv("PCDHIP_G16_HOSTTIME") != nullptr;  // developer knob: where the HOST is while the proof's launches queue up
  const auto ht0 = std::chrono::steady_clock::now();
  double ht[6] = {0, 0, 0, 0, 0, 0};
  auto stamp = [&](int k) { if (host_time) ht[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ht0).count(); };
  int rc = PCDHIP_OK;
  uint32_t rows = 0;
  if (A && B && C) rows = (uint32_t)A->num_rows;
  else if (!A && !B && !C && pk->r1cs_dev) rows = pk->rows;
  else return PCDHIP_E_ARG;
  Dom dom;
  rc = pick_domain(fr, (size_t)rows + ni, &dom);
  if (rc) return rc;
  const size_t n = dom.n;
  // ---- K3/K4/K5 first: the four MSMs that take the assignment (l', A, B_1 on G1; B on G2) do not need h, so they are
  // launched before the witness map and run concurrently with it; the h MSM follows the witness map.  Results stay on the
  // device (device image).
  G16Run run;
  run.ctx = ctx; run.pk = pk;
  // automatic choice of the assembly form: the one-point products cost ~2.5 ms (298-bit) / ~14 ms (753-bit; rounds 1-4: one lane, ~5 / ~75 ms) of
  // latency that hides under the other MSMs of a large proof; two more MSMs cost 2 x 3.5 ms (298-bit, 2^20) / 2 x 1 ms (298-bit, 2^16) / 2 x 8 ms
  // (753-bit, 2^16) of throughput.  Measured (folded vs chained, round 5): 17.6 vs 16.2 ms (298-bit, 2^20), 3.8 vs 5.0 ms (298-bit, 2^16),
  // 202 vs 180 ms (753-bit, 2^20), 29.7 vs 42.0 ms (753-bit, 5 * 2^14).
  const size_t fold_below = fe.abi_words > 12 ? (1u << 18) : (1u << 17);
  bool folded = ctx->g16_assembly == 1 || (ctx->g16_assembly == 0 && m + 4 <= fold_below);
  // (a SMALL proof over the 298-bit fields whose assignment turns out sparse takes the chained form after all: the shorter-window copies make its
  //  four MSMs cheap, and the folded form -- every scalar times s / r -- cannot use them: MNT6-298 at 2^16, witness-like: 3.36 ms folded,
  //  3.08 chained on 12-bit copies; over the 753-bit fields the one-point products (~9 ms) lose: 18.4 against 18.0)
  const bool may_unfold = ctx->g16_assembly == 0 && folded && pk->a_sparse && fe.abi_words <= 12;
  rc = run.prepare(z, r_mont, s_mont, n, folded && !may_unfold);
  if (rc) return rc;
  rc = run.decide_sparse(folded && !may_unfold);
  if (rc) return rc;
  if (may_unfold) {
    if (run.sparse) folded = false;
    else { rc = run.scale_for_fold(); if (rc) return rc; }
  }
  // Schedule (pcdhip_groth16_set_schedule).  Default 0: the four assignment MSMs are launched BEFORE the witness map and run concurrently
  // with it; the h MSM follows the map.  Inside a proof the map's ~30 short dependent kernels wait behind the MSMs' accumulate grids (the 8 ms
  // map of a 753-bit proof ends at 160 ms) and the h MSM runs last -- which LOOKS like a serialised tail, so round 4 built mode 1: the map
  // first with the device to itself, then all five MSMs at once.  Measured on one box (tools/ab_step.py, profiles/r04_ab_prove_schedule.txt):
  // mode 1 is SLOWER -- 23.4 against 20.3 ms (MNT4-298, 2^20), 5.9 against 4.9 ms (MNT6-298, 2^16), 202 against 196.5 ms (MNT4-753, 2^20): the
  // proof is bound by the sum of its MSMs' throughput either way, and mode 1 adds the map's time in front while mode 0 hides it.
  //   Round 5, mode 2: the accumulate LANE.  The map first while the MSMs sort beside it, then the accumulate kernels one after
  // the other on a stream whose CU mask leaves a CU per XCD to everything else, each MSM's fix-up / bucket reduction / one-point product on a
  // hardware queue of its own.  What the measurements say (profiles/r05_*): the accumulate kernels of this proof take 1.29 + 1.08 + 1.66 +
  // 1.67 ms (G1: A, B_1, l', h) + ~3.7 ms (G2) ALONE, the sorts 1.7, fix-ups 1.0, first reduction levels 1.1, the map 1.5, upload and
  // conversions 1.1 -- 15 .. 16 ms of work that all wants the same multiply-add issue slots, and mode 0 takes 15.7 .. 16.0 ms: the concurrency
  // round 4's timeline made look destructive is within a few per cent of the sum of the parts.  The lane cannot beat that sum, and pays for
  // the mask (a masked queue runs an accumulate kernel 15 .. 45 % slower: one CU less in ONE shader engine per XCD unbalances the dispatch) or,
  // unmasked, for its own serialisation (18.0 / 16.4 .. 18.4 ms).  Mode 0 stays the default; what round 5 gained came from doing less work.
  const bool map_first = ctx->g16_schedule >= 1;
  stamp(0);
  if (!map_first) { rc = run.launch_assignment(0, m + 4, folded); if (rc) return rc; }
  stamp(1);
  // ---- K1: h, on the context's stream
  DevCsr mats[3];
  if (A && B && C) { TRY(hipStreamSynchronize(st)); rc = upload_three(ctx, A, B, C, fe, m, mats); if (rc) return rc; }  // (staging slot AUX_SCAL is reused)
  else for (int k = 0; k < 3; k++) mats[k] = pk->mats[k];
  Dom dom_used;
  rc = witness_map_dev(ctx, fr, mats, run.z_dev, ni, &dom_used);
  if (rc) return rc;
  if (dom_used.n != dom.n) return PCDHIP_E_ARG;
  TRY(fe.convert(st, (const uint32_t*)ctx->aux_ws.buf[AUX_A], run.h_can, (uint32_t)n, 2));
  TRY(hipEventRecord(ev[1], st));
  stamp(2);
  // (round 5 also tried the map's launches merely ENQUEUED first with the h MSM right behind it and the assignment MSMs after, nothing
  //  gated -- the critical path of a proof over a witness-like assignment is upload -> map -> h MSM: slower everywhere, 11.2 against 8.0 ms
  //  there, 19.0 against 15.8 ms on a uniform assignment, because the G2 MSM and the one-point products then start last; and mode 0 with only
  //  the map's launches moved ahead of the MSMs' in host order: no difference beyond noise on either assignment, as in round 4:
  //  profiles/r05_ab_prove_schedule3.txt)
  if (map_first) { rc = run.launch_assignment_gated(0, m + 4, folded, ev[1]); if (rc) return rc; }  // (lane: no accumulation under the map; the sorts start at once)
  rc = run.launch_h(0, std::min<size_t>(pk->h_query->n, n), ev[1]);
  if (rc) return rc;
  stamp(3);
  rc = run.join();
  if (rc) return rc;
  TRY(hipEventRecord(ev[6], st));
  // assembly: three additions and three affine conversions (writes the proof in the C-ABI image)
  TRY(curve_entry(cid).assemble(st, run.msm_g1, run.msm_g2, run.proof_dev));
  TRY(hipEventRecord(ev[7], st));
  stamp(4);
  rc = finish_proof(ctx, cid, run.proof_dev, proof_out, inf_out);
  if (rc) return rc;
  stamp(5);
  if (host_time)
    fprintf(stderr, "pcdhip prove host ms: prepared %.3f, assignment MSMs enqueued %.3f, map enqueued %.3f, h enqueued %.3f, all enqueued %.3f, done %.3f\n",
            ht[0], ht[1], ht[2], ht[3], ht[4], ht[5]);
  // [witness_map, msm_h, msm_l, msm_a (A and s*A), msm_b_g1 (B_1 and r*B_1), msm_b_g2 (each on its own stream: they
  //  overlap), assembly, total]
  (void)hipEventElapsedTime(&ctx->g16_ms[0], ev[0], ev[1]);
  for (int k = 1; k <= 5; k++) ctx->g16_ms[k] = 0;
  for (int k = 0; k < run.nj; k++) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->g16_begin[k], ctx->g16_end[k]);
    ctx->g16_ms[run.jobs[k].tslot] = std::max(ctx->g16_ms[run.jobs[k].tslot], ms);
  }
  (void)hipEventElapsedTime(&ctx->g16_ms[6], ev[6], ev[7]);
  (void)hipEventElapsedTime(&ctx->g16_ms[7], ev[0], ev[7]);
  return PCDHIP_OK;
  });
}
// ------------------------------------------------------------------------------------------------ fixed-base batches, setup
namespace {
// out (device, AUX_FB_OUT): n affine points in the C-ABI image followed by n flag bytes; scalars canonical words on the device
int fixed_base_dev(pcdhip_ctx* ctx, const GroupEntry& ge, const uint32_t* base_abi_dev, const uint32_t* scalars_dev, size_t n,
                   uint32_t** out_pts, uint8_t** out_inf) {
  const size_t ab = (size_t)ge.point_abi_words * 4, jb = (size_t)ge.point_words / 2 * 3 * 4;
  TRY(ctx->aux_ws.ensure(AUX_FB_TABLE, ge.fb_table_words * 4));
  TRY(ctx->aux_ws.ensure(AUX_FB_JAC, std::max<size_t>(n, 1) * jb));
  TRY(ctx->aux_ws.ensure(AUX_FB_OUT, std::max<size_t>(n, 1) * (ab + 1) + 64));
  *out_pts = (uint32_t*)ctx->aux_ws.buf[AUX_FB_OUT];
  *out_inf = (uint8_t*)ctx->aux_ws.buf[AUX_FB_OUT] + n * ab;
  TRY(ge.fixed_base(ctx->stream, base_abi_dev, scalars_dev, (uint32_t)n, (uint32_t*)ctx->aux_ws.buf[AUX_FB_TABLE],
                    (uint32_t*)ctx->aux_ws.buf[AUX_FB_JAC], *out_pts, *out_inf));
  return PCDHIP_OK;
}
}  // namespace

int pcdhip_fixed_base_mul(pcdhip_ctx* ctx, int curve_id, int group_id, const uint64_t* base_xy, const uint64_t* scalars, size_t n,
                          uint64_t* out_xy, uint8_t* out_inf) {
  if (!ctx || !valid_curve(curve_id) || !valid_group(group_id) || !base_xy || (n && (!scalars || !out_xy || !out_inf)) || (n >> 31))
    return PCDHIP_E_ARG;
  BIND();
  const GroupEntry& ge = group_entry(curve_id, group_id);
  const size_t ab = (size_t)ge.point_abi_words * 4, sb = (size_t)ge.scalar_words * 4;
  const size_t base_off = (n * sb + 63) / 64 * 64;  // the base point sits behind the scalars
  TRY(ctx->aux_ws.ensure(AUX_SCAL, base_off + ab));
  uint32_t* sc = (uint32_t*)ctx->aux_ws.buf[AUX_SCAL];
  uint32_t* base_dev = (uint32_t*)((char*)sc + base_off);
  if (n) TRY(hipMemcpyAsync(sc, scalars, n * sb, hipMemcpyHostToDevice, ctx->stream));
  TRY(hipMemcpyAsync(base_dev, base_xy, ab, hipMemcpyHostToDevice, ctx->stream));
  uint32_t* pts;
  uint8_t* inf;
  int rc = fixed_base_dev(ctx, ge, base_dev, sc, n, &pts, &inf);
  if (rc) return rc;
  if (n) {
    TRY(hipMemcpyAsync(out_xy, pts, n * ab, hipMemcpyDeviceToHost, ctx->stream));
    TRY(hipMemcpyAsync(out_inf, inf, n, hipMemcpyDeviceToHost, ctx->stream));
  }
  TRY(hipStreamSynchronize(ctx->stream));
  return PCDHIP_OK;
}

namespace {
// transpose of a CSR matrix with `cols` columns, as CSR (host side: an index permutation, no field arithmetic)
struct HostCsrT {
  std::vector<uint64_t> rp;
  std::vector<uint32_t> col;
  std::vector<uint64_t> coeff;
  pcdhip_csr view;
};
int transpose_csr(const pcdhip_csr* m, size_t cols, size_t limbs, HostCsrT* t) {
  if (!m || !m->row_ptr) return PCDHIP_E_ARG;
  const uint64_t nnz = m->row_ptr[m->num_rows];
  if (nnz && (!m->col || !m->coeff)) return PCDHIP_E_ARG;
  t->rp.assign(cols + 1, 0);
  for (uint64_t k = 0; k < nnz; k++) {
    if (m->col[k] >= cols) return PCDHIP_E_ARG;
    t->rp[m->col[k] + 1]++;
  }
  for (size_t c = 0; c < cols; c++) t->rp[c + 1] += t->rp[c];
  t->col.resize(nnz);
  t->coeff.resize(nnz * limbs);
  std::vector<uint64_t> fill(t->rp.begin(), t->rp.end() - 1);
  for (uint64_t r = 0; r < m->num_rows; r++)
    for (uint64_t k = m->row_ptr[r]; k < m->row_ptr[r + 1]; k++) {
      const uint64_t d = fill[m->col[k]]++;
      t->col[d] = (uint32_t)r;
      memcpy(&t->coeff[d * limbs], m->coeff + k * limbs, limbs * 8);
    }
  t->view = {cols, t->rp.data(), t->col.data(), t->coeff.data()};
  return PCDHIP_OK;
}
}  // namespace

int pcdhip_groth16_setup(pcdhip_ctx* ctx, int curve_id, const pcdhip_csr* A, const pcdhip_csr* B, const pcdhip_csr* C, size_t num_vars,
                         size_t num_inputs, const uint64_t* g1_xy, const uint64_t* g2_xy, const uint64_t* toxic, pcdhip_g16_setup_out* out) {
  return guarded([&]() -> int {
  if (!ctx || !valid_curve(curve_id) || !A || !B || !C || !g1_xy || !g2_xy || !toxic || !out) return PCDHIP_E_ARG;
  if (num_inputs < 1 || num_inputs > num_vars || (num_vars >> 31) || A->num_rows != B->num_rows || A->num_rows != C->num_rows) return PCDHIP_E_ARG;
  if (!out->alpha_g1 || !out->beta_g1 || !out->delta_g1 || !out->beta_g2 || !out->gamma_g2 || !out->delta_g2 || !out->a_query ||
      !out->a_inf || !out->b_g1_query || !out->b_g1_inf || !out->b_g2_query || !out->b_g2_inf || !out->gamma_abc_g1 || !out->gamma_abc_inf)
    return PCDHIP_E_ARG;
  BIND();
  const int fr = kCurveFr[curve_id];
  const FieldEntry& fe = field_entry(fr);
  const size_t m = num_vars, ni = num_inputs, nc = A->num_rows, limbs = (size_t)kFieldLimbs[fr];
  if (m > ni && (!out->l_query || !out->l_inf)) return PCDHIP_E_ARG;
  Dom d;
  int rc = pick_domain(fr, nc + ni, &d);
  if (rc) return rc;
  const size_t n = d.n;
  if (n > 1 && (!out->h_query || !out->h_inf)) return PCDHIP_E_ARG;
  hipStream_t st = ctx->stream;
  const size_t sb = (size_t)fe.abi_words * 4, eb = (size_t)fe.words * 4;
  // domain constants (generator, 1/n): the tables of the transforms over the same domain
  const FftTables* t;
  rc = d.m == 1 ? get_tables(ctx, fr, d.a, &t) : get_mixed_tables(ctx, fr, d, &t);
  if (rc) return rc;
  // scalars:  G1 [a (m) | b (m) | gamma_abc, l (m) | h (n - 1) | alpha, beta, delta]   G2 [b (m) | beta, gamma, delta]
  const size_t n1 = 3 * m + (n - 1) + 3, n2 = m + 3;
  TRY(ctx->aux_ws.ensure(AUX_FFT_X, n * eb));
  TRY(ctx->aux_ws.ensure(AUX_A, m * eb));
  TRY(ctx->aux_ws.ensure(AUX_B, m * eb));
  TRY(ctx->aux_ws.ensure(AUX_C, m * eb));
  TRY(ctx->aux_ws.ensure(AUX_Z_CANON, n1 * sb));
  TRY(ctx->aux_ws.ensure(AUX_H_CANON, n2 * sb));
  const GroupEntry& g1 = group_entry(curve_id, 1);
  const GroupEntry& g2 = group_entry(curve_id, 2);
  const size_t a1 = (size_t)g1.point_abi_words * 4, a2 = (size_t)g2.point_abi_words * 4;
  TRY(ctx->aux_ws.ensure(AUX_Z, 5 * sb + (size_t)fe.setup_consts * eb + a1 + a2 + 256));
  uint32_t* toxic_dev = (uint32_t*)ctx->aux_ws.buf[AUX_Z];
  uint32_t* consts_dev = toxic_dev + 5 * fe.abi_words;
  uint32_t* err_dev = consts_dev + (size_t)fe.setup_consts * fe.words;
  uint32_t* g1_dev = err_dev + 16;
  uint32_t* g2_dev = g1_dev + g1.point_abi_words;
  uint32_t* u = (uint32_t*)ctx->aux_ws.buf[AUX_FFT_X];
  uint32_t *at = (uint32_t*)ctx->aux_ws.buf[AUX_A], *bt = (uint32_t*)ctx->aux_ws.buf[AUX_B], *ct = (uint32_t*)ctx->aux_ws.buf[AUX_C];
  uint32_t* s1 = (uint32_t*)ctx->aux_ws.buf[AUX_Z_CANON];
  uint32_t* s2 = (uint32_t*)ctx->aux_ws.buf[AUX_H_CANON];
  TRY(hipMemcpyAsync(toxic_dev, toxic, 5 * sb, hipMemcpyHostToDevice, st));
  TRY(hipMemcpyAsync(g1_dev, g1_xy, a1, hipMemcpyHostToDevice, st));
  TRY(hipMemcpyAsync(g2_dev, g2_xy, a2, hipMemcpyHostToDevice, st));
  TRY(hipMemsetAsync(err_dev, 0, 4, st));
  TRY(fe.setup_scalars(st, t->consts, toxic_dev, (uint32_t)n, (uint32_t)nc, (uint32_t)m, (uint32_t)ni, nullptr, nullptr, nullptr, u, consts_dev,
                       err_dev, nullptr, nullptr, nullptr, nullptr, nullptr, 0));
  // At, Bt, Ct: a_i(tau) = sum_j A[j][i] u_j  -- the transposed matrices times u
  const pcdhip_csr* ms[3] = {A, B, C};
  uint32_t* vecs[3] = {at, bt, ct};
  for (int k = 0; k < 3; k++) {
    HostCsrT tr;
    rc = transpose_csr(ms[k], m, limbs, &tr);
    if (rc) return rc;
    DevCsr dm;
    rc = upload_csr(ctx, AUX_CSR_RP, &tr.view, fe, nc, &dm);
    if (rc) return rc;
    TRY(fe.spmv(st, dm, u, 0, 0, (uint32_t)m, vecs[k]));
    TRY(hipStreamSynchronize(st));  // `tr` and the staging slot are reused by the next matrix
  }
  uint32_t err = 0;
  TRY(hipMemcpyAsync(&err, err_dev, 4, hipMemcpyDeviceToHost, st));
  TRY(hipStreamSynchronize(st));
  if (err) return PCDHIP_E_ARG;  // tau lies in the evaluation domain
  TRY(fe.setup_scalars(st, t->consts, toxic_dev, (uint32_t)n, (uint32_t)nc, (uint32_t)m, (uint32_t)ni, at, bt, ct, u, consts_dev, err_dev, s1,
                       s1 + m * fe.abi_words, s1 + 2 * m * fe.abi_words, s1 + 3 * m * fe.abi_words, s2, 1));
  // alpha, beta, delta | beta, gamma, delta: C-ABI Montgomery -> canonical
  uint32_t* tail1 = s1 + (3 * m + (n - 1)) * fe.abi_words;
  uint32_t* tail2 = s2 + m * fe.abi_words;
  const int idx1[3] = {0, 1, 3}, idx2[3] = {1, 2, 3};
  for (int k = 0; k < 3; k++) {
    TRY(fe.convert(st, toxic_dev + idx1[k] * fe.abi_words, tail1 + k * fe.abi_words, 1, 3));
    TRY(fe.convert(st, toxic_dev + idx2[k] * fe.abi_words, tail2 + k * fe.abi_words, 1, 3));
  }
  // the fixed-base batches, then scatter into the caller's arrays
  uint32_t* pts;
  uint8_t* inf;
  rc = fixed_base_dev(ctx, g1, g1_dev, s1, n1, &pts, &inf);
  if (rc) return rc;
  auto fetch = [&](uint64_t* dst, uint8_t* dst_inf, size_t first, size_t cnt, size_t ab) -> hipError_t {
    if (!cnt) return hipSuccess;
    hipError_t e = hipMemcpyAsync(dst, (char*)pts + first * ab, cnt * ab, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && dst_inf) e = hipMemcpyAsync(dst_inf, inf + first, cnt, hipMemcpyDeviceToHost, st);
    return e;
  };
  TRY(fetch(out->a_query, out->a_inf, 0, m, a1));
  TRY(fetch(out->b_g1_query, out->b_g1_inf, m, m, a1));
  TRY(fetch(out->gamma_abc_g1, out->gamma_abc_inf, 2 * m, ni, a1));
  TRY(fetch(out->l_query, out->l_inf, 2 * m + ni, m - ni, a1));
  TRY(fetch(out->h_query, out->h_inf, 3 * m, n - 1, a1));
  TRY(fetch(out->alpha_g1, nullptr, 3 * m + n - 1, 1, a1));
  TRY(fetch(out->beta_g1, nullptr, 3 * m + n, 1, a1));
  TRY(fetch(out->delta_g1, nullptr, 3 * m + n + 1, 1, a1));
  TRY(hipStreamSynchronize(st));
  rc = fixed_base_dev(ctx, g2, g2_dev, s2, n2, &pts, &inf);
  if (rc) return rc;
  TRY(fetch(out->b_g2_query, out->b_g2_inf, 0, m, a2));
  TRY(fetch(out->beta_g2, nullptr, m, 1, a2));
  TRY(fetch(out->gamma_g2, nullptr, m + 1, 1, a2));
  TRY(fetch(out->delta_g2, nullptr, m + 2, 1, a2));
  TRY(hipStreamSynchronize(st));
  out->domain_size = n;
  return PCDHIP_OK;
  });
}

// ------------------------------------------------------------------------------------------------ pairing
// groups x per pairs -> groups GT elements
// g1_z (nullable; honoured by the wave-per-pairing kernels only -- callers pass it only while ctx->pairing_vm): Z of Jacobian G1 points
static int pairing_groups(pcdhip_ctx* ctx, int curve_id, const uint64_t* g1_xy, const uint8_t* g1_inf, const uint64_t* g2_xy,
                          const uint8_t* g2_inf, size_t groups, size_t per, uint64_t* gt_out, const uint64_t* g1_z = nullptr) {
  const size_t n_pairs = groups * per;
  if (!ctx || !valid_curve(curve_id) || (n_pairs && (!g1_xy || !g2_xy)) || !gt_out || n_pairs >= (1u << 24) || (groups && !per)) return PCDHIP_E_ARG;
  BIND();
  const PairingEntry& pe = pairing_entry(curve_id);
  const size_t w1 = (size_t)pcdhip_point_limbs(curve_id, 1) * 8, w2 = (size_t)pcdhip_point_limbs(curve_id, 2) * 8;
  const size_t gb = (size_t)pe.gt_words * 4, gi = (size_t)pe.gt_internal_words * 4;
  const size_t gt_off = n_pairs * (w1 + w2 + gi), z_off = gt_off + std::max<size_t>(groups, 1) * gb, flag_off = z_off + n_pairs * (w1 / 2);
  TRY(ctx->aux_ws.ensure(AUX_MISC, flag_off + 2 * n_pairs + 256));
  char* d = (char*)ctx->aux_ws.buf[AUX_MISC];
  uint32_t* g1d = (uint32_t*)d;
  uint32_t* g2d = (uint32_t*)(d + n_pairs * w1);
  uint32_t* scr = (uint32_t*)(d + n_pairs * (w1 + w2));
  uint32_t* out = (uint32_t*)(d + gt_off);
  if (n_pairs) {
    TRY(hipMemcpyAsync(g1d, g1_xy, n_pairs * w1, hipMemcpyHostToDevice, ctx->stream));
    TRY(hipMemcpyAsync(g2d, g2_xy, n_pairs * w2, hipMemcpyHostToDevice, ctx->stream));
    // flagged infinities -> (0, 0)
    TRY(zero_flagged(ctx->stream, g1d, (uint8_t*)d + flag_off, g1_inf, n_pairs, w1));
    TRY(zero_flagged(ctx->stream, g2d, (uint8_t*)d + flag_off + n_pairs, g2_inf, n_pairs, w2));
  }
  if (!ctx->vm_block[curve_id] && ctx->pairing_vm) TRY(pe.vm_upload(ctx->stream, &ctx->vm_block[curve_id], &ctx->vm_tables[curve_id]));
  uint32_t* g1zd = nullptr;
  if (g1_z && n_pairs) {
    g1zd = (uint32_t*)(d + z_off);
    TRY(hipMemcpyAsync(g1zd, g1_z, n_pairs * (w1 / 2), hipMemcpyHostToDevice, ctx->stream));
  }
  TRY(pe.multi_pairing(ctx->stream, g1d, g1zd, g2d, (uint32_t)groups, (uint32_t)per, scr, out, ctx->pairing_vm ? &ctx->vm_tables[curve_id] : nullptr));
  TRY(hipMemcpyAsync(gt_out, out, groups * gb, hipMemcpyDeviceToHost, ctx->stream));
  TRY(hipStreamSynchronize(ctx->stream));
  return PCDHIP_OK;
}
int pcdhip_pairing_set_mode(pcdhip_ctx* ctx, int mode) {
  if (!ctx || mode < 0 || mode > 1) return PCDHIP_E_ARG;
  ctx->pairing_vm = mode == 0;
  return PCDHIP_OK;
}
int pcdhip_multi_pairing(pcdhip_ctx* ctx, int curve_id, const uint64_t* g1_xy, const uint8_t* g1_inf, const uint64_t* g2_xy,
                         const uint8_t* g2_inf, size_t n_pairs, uint64_t* gt_out) {
  // (one group holding every pair; with no pair at all the empty product still goes through the final exponentiation)
  if (n_pairs == 0) {
    if (!ctx || !valid_curve(curve_id) || !gt_out) return PCDHIP_E_ARG;
    BIND();
    const PairingEntry& pe = pairing_entry(curve_id);
    TRY(ctx->aux_ws.ensure(AUX_MISC, (size_t)pe.gt_words * 4 + 256));
    uint32_t* out = (uint32_t*)ctx->aux_ws.buf[AUX_MISC];
    if (!ctx->vm_block[curve_id] && ctx->pairing_vm) TRY(pe.vm_upload(ctx->stream, &ctx->vm_block[curve_id], &ctx->vm_tables[curve_id]));
    TRY(pe.multi_pairing(ctx->stream, out, nullptr, out, 1, 0, out, out, ctx->pairing_vm ? &ctx->vm_tables[curve_id] : nullptr));
    TRY(hipMemcpyAsync(gt_out, out, (size_t)pe.gt_words * 4, hipMemcpyDeviceToHost, ctx->stream));
    TRY(hipStreamSynchronize(ctx->stream));
    return PCDHIP_OK;
  }
  return pairing_groups(ctx, curve_id, g1_xy, g1_inf, g2_xy, g2_inf, 1, n_pairs, gt_out);
}

// ---- prepared verifying key (ark-groth16 `prepare_verifying_key` / `SNARK::process_vk`; reference call sites
// src/ec_cycle_pcd/mod.rs:71,371,445,495,553) and the verifications that use it
}  // extern "C"

struct pcdhip_pvk {
  int curve_id = 0;
  size_t num_inputs = 0;
  std::vector<uint64_t> alpha, beta, neg_gamma, neg_delta, gamma_abc, alpha_beta /* e(alpha, beta) */, gt_one;
  std::vector<uint8_t> gamma_abc_inf;
  pcdhip_bases* abc = nullptr;  // gamma_abc_g1 resident (no precomputed copies; used when there are too many inputs for window tables)
  // prepared inputs (fixed_base.hip.h): gamma_abc_g1 in the C-ABI image with flagged points zeroed, then one window table per
  // gamma_abc_g1[j], j >= 1 -- one device block; null when the key has more than PVK_TABLE_INPUTS inputs
  uint32_t* abc_dev = nullptr;
  size_t abc_tables_off = 0;  // u32 words from abc_dev to the tables
};
constexpr size_t PVK_TABLE_INPUTS = 256;

namespace {
// y -> -y of n affine points in the C-ABI image, on the host (the library's own host-callable field templates)
template <class FQ>
void neg_y_words(uint64_t* xy, size_t coeffs_per_coord) {
  typedef Fp<FQ, false> B;
  uint32_t* y = (uint32_t*)xy + coeffs_per_coord * B::ABI_WORDS;
  for (size_t c = 0; c < coeffs_per_coord; c++) B::from_abi(y + c * B::ABI_WORDS).neg().to_abi(y + c * B::ABI_WORDS);
}
void negate_point(int curve_id, int group_id, uint64_t* xy) {
  const size_t deg = group_id == 1 ? 1 : (size_t)kCurveG2Deg[curve_id];
  bool zero = true;  // the point at infinity (zero coordinates) stays what it is
  for (size_t k = 0; k < (size_t)pcdhip_point_limbs(curve_id, group_id); k++) zero = zero && xy[k] == 0;
  if (zero) return;
  switch (curve_id) {
    case 0: neg_y_words<F298A>(xy, deg); break;
    case 1: neg_y_words<F298B>(xy, deg); break;
    case 2: neg_y_words<F753A>(xy, deg); break;
    default: neg_y_words<F753B>(xy, deg); break;
  }
}
// canonical scalars: out = sum_i a_i * b_i mod r  (b_i == nullptr: plain sum); `words` u32 words each
template <class FR>
void lincomb_words(const uint64_t* const* a, const uint64_t* const* b, size_t n, uint64_t* out) {
  typedef Fp<FR, false> B;
  B acc = B::zero();
  for (size_t i = 0; i < n; i++) {
    B x = B::from_canonical_words((const uint32_t*)a[i]);
    if (b && b[i]) x = x * B::from_canonical_words((const uint32_t*)b[i]);
    acc = acc + x;
  }
  acc.to_canonical_words((uint32_t*)out);
}
void scalar_lincomb(int fr, const uint64_t* const* a, const uint64_t* const* b, size_t n, uint64_t* out) {
  switch (fr) {
    case 0: lincomb_words<F298A>(a, b, n, out); break;
    case 1: lincomb_words<F298B>(a, b, n, out); break;
    case 2: lincomb_words<F753A>(a, b, n, out); break;
    default: lincomb_words<F753B>(a, b, n, out); break;
  }
}
// acc_i = gamma_abc[0] + sum_j x_ij gamma_abc[j] for k proofs (Jacobian C-ABI image -> affine + flags), through the resident bases
// acc_z (nullable): when given, non-empty on return iff the accumulations came back in Jacobian form -- (X, Y) in acc_a, Z in acc_z -- which
// happens with window tables and the wave-per-pairing kernels on (they take Jacobian G1 points: no inversion anywhere in a verification)
int prepare_inputs(pcdhip_ctx* ctx, const pcdhip_pvk* pvk, size_t k, const uint64_t* public_inputs, std::vector<uint64_t>* acc_a, std::vector<uint8_t>* acc_inf,
                   std::vector<uint64_t>* acc_z = nullptr) {
  const int cid = pvk->curve_id;
  const size_t l1 = (size_t)pcdhip_point_limbs(cid, 1), sl = (size_t)kFieldLimbs[kCurveFr[cid]], ni = pvk->num_inputs;
  std::vector<uint64_t> acc_j(k * (l1 / 2 * 3)), scal(ni * sl);
  acc_a->assign(k * l1, 0);
  acc_inf->assign(k, 0);
  if (pvk->abc_dev) {  // window tables: every proof's accumulation in one launch, no doubling chains
    const GroupEntry& ge = group_entry(cid, 1);
    const size_t jw = (size_t)ge.point_words / 2 * 3, sb = (ni - 1) * sl * 8;
    // Jacobian accumulations only when the wave-per-pairing kernels will take them: 3 pairings per proof, PCD_VM_MAX_PAIRS per launch
    // (beyond, the lane-per-pairing kernels want affine points: a batch of 1366+ proofs used to fail here)
    const bool jac = acc_z && ctx->pairing_vm && 3 * k <= PCD_VM_MAX_PAIRS;
    TRY(ctx->aux_ws.ensure(AUX_FB_JAC, k * (64 * jw * 4 + sb + l1 * 8 + l1 * 4 + 1) + 64));
    char* d = (char*)ctx->aux_ws.buf[AUX_FB_JAC];
    uint32_t* scratch = (uint32_t*)d;
    uint32_t* scal_d = (uint32_t*)(d + k * 64 * jw * 4);
    uint32_t* out_d = (uint32_t*)((char*)scal_d + k * sb);
    uint32_t* z_d = (uint32_t*)((char*)out_d + k * l1 * 8);
    uint8_t* inf_d = (uint8_t*)z_d + k * l1 * 4;
    if (sb) TRY(hipMemcpyAsync(scal_d, public_inputs, k * sb, hipMemcpyHostToDevice, ctx->stream));
    TRY(ge.fb_inputs(ctx->stream, pvk->abc_dev + pvk->abc_tables_off, pvk->abc_dev, (uint32_t)ni, scal_d, (uint32_t)k, scratch, out_d, inf_d, jac ? z_d : nullptr, 1));
    if (jac) { acc_z->assign(k * (l1 / 2), 0); TRY(hipMemcpyAsync(acc_z->data(), z_d, k * l1 * 4, hipMemcpyDeviceToHost, ctx->stream)); }
    TRY(hipMemcpyAsync(acc_a->data(), out_d, k * l1 * 8, hipMemcpyDeviceToHost, ctx->stream));
    TRY(hipMemcpyAsync(acc_inf->data(), inf_d, k, hipMemcpyDeviceToHost, ctx->stream));
    TRY(hipStreamSynchronize(ctx->stream));
    return PCDHIP_OK;
  }
  for (size_t i = 0; i < k; i++) {
    std::fill(scal.begin(), scal.end(), 0);
    scal[0] = 1;
    if (ni > 1) memcpy(&scal[sl], public_inputs + i * (ni - 1) * sl, (ni - 1) * sl * 8);
    int rc = pcdhip_msm(ctx, pvk->abc, 0, scal.data(), ni, &acc_j[i * (l1 / 2 * 3)]);
    if (rc) return rc;
  }
  return pcdhip_to_affine(ctx, cid, 1, acc_j.data(), k, acc_a->data(), acc_inf->data());
}
// The prepared verification on the wave-per-pairing kernels as ONE trip to the device: a single upload (public inputs, the proofs'
// points, -gamma, -delta, Z = 1), the input accumulation writing its Jacobian result straight into the middle pair's slot, the Miller
// loops, the final exponentiations, one download.  (As separate calls -- prepare_inputs, then pairing_groups -- the same work was
// fourteen stream operations and two synchronisations: 0.33 ms in front of a 1.1 ms Miller loop.)
int verify_prepared_vm(pcdhip_ctx* ctx, const pcdhip_pvk* pvk, size_t k, const uint64_t* public_inputs, const uint64_t* proofs, const uint8_t* proofs_inf,
                       int* ok) {
  const int cid = pvk->curve_id;
  const PairingEntry& pe = pairing_entry(cid);
  const GroupEntry& ge = group_entry(cid, 1);
  const size_t l1 = (size_t)pcdhip_point_limbs(cid, 1), l2 = (size_t)pcdhip_point_limbs(cid, 2), pl = 2 * l1 + l2, lz = l1 / 2;
  const size_t sl = (size_t)kFieldLimbs[kCurveFr[cid]], ni = pvk->num_inputs, np = 3 * k;
  const size_t w1 = l1 * 8, w2 = l2 * 8, wz = lz * 8, sb = (ni - 1) * sl * 8, jw = (size_t)ge.point_words / 2 * 3;
  const size_t gb = (size_t)pe.gt_words * 4, gi = (size_t)pe.gt_internal_words * 4, gw = pvk->alpha_beta.size();
  // the upload: [g1 | g2 | z | scalars], 16-byte aligned parts; behind it on the device: Miller values, GT results, accumulation scratch
  auto up16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
  const size_t o_g1 = 0, o_g2 = o_g1 + up16(np * w1), o_z = o_g2 + up16(np * w2), o_sc = o_z + up16(np * wz), up_bytes = o_sc + up16(k * sb);
  const size_t o_f = up_bytes, o_gt = o_f + up16(np * gi), o_scr = o_gt + up16(k * gb), o_inf = o_scr + up16(k * 64 * jw * 4), total = o_inf + up16(k) + 256;
  TRY(ctx->aux_ws.ensure(AUX_MISC, total));
  char* d = (char*)ctx->aux_ws.buf[AUX_MISC];
  std::vector<uint64_t> host(up_bytes / 8, 0);
  char* h = (char*)host.data();
  for (size_t i = 0; i < k; i++) {
    const uint64_t* pr = proofs + i * pl;
    const bool ia = proofs_inf && proofs_inf[3 * i], ib = proofs_inf && proofs_inf[3 * i + 1], ic = proofs_inf && proofs_inf[3 * i + 2];
    // (a flagged infinity stays (0, 0), which is how the kernels know it)
    if (!ia) memcpy(h + o_g1 + (3 * i) * w1, pr, w1);
    if (!ic) memcpy(h + o_g1 + (3 * i + 2) * w1, pr + l1 + l2, w1);
    if (!ib) memcpy(h + o_g2 + (3 * i) * w2, pr + l1, w2);
    memcpy(h + o_g2 + (3 * i + 1) * w2, pvk->neg_gamma.data(), w2);
    memcpy(h + o_g2 + (3 * i + 2) * w2, pvk->neg_delta.data(), w2);
    memcpy(h + o_z + (3 * i) * wz, pvk->gt_one.data(), wz);       // Z = 1 (the first coefficient of GT's one) for the proof's own points
    memcpy(h + o_z + (3 * i + 2) * wz, pvk->gt_one.data(), wz);
  }
  if (sb) memcpy(h + o_sc, public_inputs, k * sb);
  TRY(hipMemcpyAsync(d, h, up_bytes, hipMemcpyHostToDevice, ctx->stream));
  if (!ctx->vm_block[cid]) TRY(pe.vm_upload(ctx->stream, &ctx->vm_block[cid], &ctx->vm_tables[cid]));
  uint32_t* g1d = (uint32_t*)(d + o_g1);
  uint32_t* g1zd = (uint32_t*)(d + o_z);
  // acc_i -> pair 3 i + 1, Jacobian: (X, Y) into the G1 array, Z into the Z array (an infinite accumulation has Z = 0)
  TRY(ge.fb_inputs(ctx->stream, pvk->abc_dev + pvk->abc_tables_off, pvk->abc_dev, (uint32_t)ni, (const uint32_t*)(d + o_sc), (uint32_t)k, (uint32_t*)(d + o_scr),
                   g1d + w1 / 4, (uint8_t*)(d + o_inf), g1zd + wz / 4, 3));
  TRY(pe.multi_pairing(ctx->stream, g1d, g1zd, (const uint32_t*)(d + o_g2), (uint32_t)k, 3, (uint32_t*)(d + o_f), (uint32_t*)(d + o_gt), &ctx->vm_tables[cid]));
  std::vector<uint64_t> gt(k * gw);
  TRY(hipMemcpyAsync(gt.data(), d + o_gt, k * gb, hipMemcpyDeviceToHost, ctx->stream));
  TRY(hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < k; i++) ok[i] = memcmp(&gt[i * gw], pvk->alpha_beta.data(), gw * 8) == 0 ? 1 : 0;
  return PCDHIP_OK;
}
}  // namespace

extern "C" {

int pcdhip_process_vk(pcdhip_ctx* ctx, int curve_id, const uint64_t* alpha_g1, const uint64_t* beta_g2, const uint64_t* gamma_g2,
                      const uint64_t* delta_g2, const uint64_t* gamma_abc_g1, const uint8_t* gamma_abc_inf, size_t num_inputs, pcdhip_pvk** out) {
  return guarded([&]() -> int {
  if (!ctx || !valid_curve(curve_id) || !alpha_g1 || !beta_g2 || !gamma_g2 || !delta_g2 || !gamma_abc_g1 || num_inputs < 1 || !out) return PCDHIP_E_ARG;
  BIND();
  *out = nullptr;
  const size_t l1 = (size_t)pcdhip_point_limbs(curve_id, 1), l2 = (size_t)pcdhip_point_limbs(curve_id, 2);
  const PairingEntry& pe = pairing_entry(curve_id);
  pcdhip_pvk* pvk = new pcdhip_pvk();
  pvk->curve_id = curve_id; pvk->num_inputs = num_inputs;
  pvk->alpha.assign(alpha_g1, alpha_g1 + l1);
  pvk->beta.assign(beta_g2, beta_g2 + l2);
  pvk->neg_gamma.assign(gamma_g2, gamma_g2 + l2);
  pvk->neg_delta.assign(delta_g2, delta_g2 + l2);
  negate_point(curve_id, 2, pvk->neg_gamma.data());
  negate_point(curve_id, 2, pvk->neg_delta.data());
  pvk->gamma_abc.assign(gamma_abc_g1, gamma_abc_g1 + num_inputs * l1);
  pvk->gamma_abc_inf.assign(num_inputs, 0);
  if (gamma_abc_inf) pvk->gamma_abc_inf.assign(gamma_abc_inf, gamma_abc_inf + num_inputs);
  pvk->alpha_beta.assign(pe.gt_words / 2, 0);
  pvk->gt_one.assign(pe.gt_words / 2, 0);
  const int saved = ctx->precompute;
  ctx->precompute = 0;
  int rc = bases_upload_single(ctx, curve_id, 1, gamma_abc_g1, gamma_abc_inf, num_inputs, &pvk->abc);  // (device 0 of a multi-device context)
  ctx->precompute = saved;
  if (!rc && num_inputs <= PVK_TABLE_INPUTS) {
    const GroupEntry& ge = group_entry(curve_id, 1);
    const size_t abi_w = (size_t)ge.point_abi_words;
    const size_t tab_w = ge.fb_table_words;   // one table + the Jacobians of its doubling chain
    const size_t head = (num_inputs * abi_w + num_inputs / 4 + 64 + 3) / 4 * 4;   // points, then the flag bytes, 16-byte aligned
    hipError_t e = hipMalloc((void**)&pvk->abc_dev, (head + std::max<size_t>(num_inputs - 1, 1) * tab_w) * 4);
    if (e != hipSuccess) {  // no room for the window tables (up to 1.3 GB for a 753-bit key): the plain resident vector serves (MSM path)
      (void)hipGetLastError();
      pvk->abc_dev = nullptr;
    }
  }
  if (!rc && pvk->abc_dev) {
    const GroupEntry& ge = group_entry(curve_id, 1);
    const size_t abi_w = (size_t)ge.point_abi_words;
    const size_t head = (num_inputs * abi_w + num_inputs / 4 + 64 + 3) / 4 * 4;
    pvk->abc_tables_off = head;
    hipError_t e = hipMemcpyAsync(pvk->abc_dev, gamma_abc_g1, num_inputs * abi_w * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = zero_flagged(ctx->stream, pvk->abc_dev, (uint8_t*)(pvk->abc_dev + num_inputs * abi_w), gamma_abc_inf, num_inputs, abi_w * 4);
    if (e == hipSuccess) e = ge.fb_tables(ctx->stream, pvk->abc_dev, (uint32_t)num_inputs, pvk->abc_dev + head);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { pcdhip_pvk_free(ctx, pvk); return fail(ctx, e); }
  }
  rc = rc ? rc : pcdhip_multi_pairing(ctx, curve_id, alpha_g1, nullptr, beta_g2, nullptr, 1, pvk->alpha_beta.data());  // e(alpha, beta), once
  rc = rc ? rc : pcdhip_multi_pairing(ctx, curve_id, nullptr, nullptr, nullptr, nullptr, 0, pvk->gt_one.data());
  if (rc) { pcdhip_pvk_free(ctx, pvk); return rc; }
  *out = pvk;
  return PCDHIP_OK;
  });
}
void pcdhip_pvk_free(pcdhip_ctx* ctx, pcdhip_pvk* pvk) {
  if (!pvk) return;
  pcdhip_bases_free(ctx, pvk->abc);
  if (pvk->abc_dev) { if (ctx) (void)hipSetDevice(ctx->device); (void)hipFree(pvk->abc_dev); }
  delete pvk;
}

// `verify_with_processed_vk` for n proofs under one key: e(A, B) e(acc, -gamma) e(C, -delta) == e(alpha, beta) -- three Miller
// loops and one final exponentiation per proof, every Miller loop of the batch in one launch, one lane per proof for the products
// and final exponentiations.
int pcdhip_groth16_verify_prepared(pcdhip_ctx* ctx, const pcdhip_pvk* pvk, size_t n_proofs, const uint64_t* public_inputs_canonical,
                                   const uint64_t* proofs, const uint8_t* proofs_inf, int* ok) {
  return guarded([&]() -> int {
  if (!ctx || !pvk || (n_proofs && (!proofs || !ok)) || (pvk->num_inputs > 1 && n_proofs && !public_inputs_canonical) || n_proofs >= (1u << 20))
    return PCDHIP_E_ARG;
  if (n_proofs == 0) return PCDHIP_OK;
  BIND();
  const int cid = pvk->curve_id;
  const size_t l1 = (size_t)pcdhip_point_limbs(cid, 1), l2 = (size_t)pcdhip_point_limbs(cid, 2), pl = 2 * l1 + l2, k = n_proofs;
  if (ctx->pairing_vm && pvk->abc_dev && 3 * k <= PCD_VM_MAX_PAIRS) return verify_prepared_vm(ctx, pvk, k, public_inputs_canonical, proofs, proofs_inf, ok);
  std::vector<uint64_t> acc_a, acc_z;
  std::vector<uint8_t> acc_inf;
  int rc = prepare_inputs(ctx, pvk, k, public_inputs_canonical, &acc_a, &acc_inf, &acc_z);
  if (rc) return rc;
  const size_t gw = pvk->alpha_beta.size(), lz = l1 / 2;
  std::vector<uint64_t> gt(k * gw), g1s(3 * k * l1), g2s(3 * k * l2), g1z;
  if (!acc_z.empty()) {  // Z = 1 (the first coefficient of GT's one) for the proof's own points, the accumulation's Z for the middle pair
    g1z.resize(3 * k * lz);
    for (size_t i = 0; i < 3 * k; i++) memcpy(&g1z[i * lz], i % 3 == 1 ? &acc_z[(i / 3) * lz] : pvk->gt_one.data(), lz * 8);
  }
  std::vector<uint8_t> inf1(3 * k, 0), inf2(3 * k, 0);
  for (size_t i = 0; i < k; i++) {
    const uint64_t* pr = proofs + i * pl;
    memcpy(&g1s[(3 * i) * l1], pr, l1 * 8);
    memcpy(&g1s[(3 * i + 1) * l1], &acc_a[i * l1], l1 * 8);
    memcpy(&g1s[(3 * i + 2) * l1], pr + l1 + l2, l1 * 8);
    memcpy(&g2s[(3 * i) * l2], pr + l1, l2 * 8);
    memcpy(&g2s[(3 * i + 1) * l2], pvk->neg_gamma.data(), l2 * 8);
    memcpy(&g2s[(3 * i + 2) * l2], pvk->neg_delta.data(), l2 * 8);
    inf1[3 * i + 1] = acc_inf[i];
    if (proofs_inf) { inf1[3 * i] = proofs_inf[3 * i]; inf2[3 * i] = proofs_inf[3 * i + 1]; inf1[3 * i + 2] = proofs_inf[3 * i + 2]; }
  }
  rc = pairing_groups(ctx, cid, g1s.data(), inf1.data(), g2s.data(), inf2.data(), k, 3, gt.data(), g1z.empty() ? nullptr : g1z.data());
  if (rc) return rc;
  for (size_t i = 0; i < k; i++) ok[i] = memcmp(&gt[i * gw], pvk->alpha_beta.data(), gw * 8) == 0 ? 1 : 0;
  return PCDHIP_OK;
  });
}

// All n proofs at once with ONE final exponentiation: with caller-supplied challenges rho_i (128 bits each, two
// u64 limbs per proof; the library has no RNG) the n equations are raised to rho_i and multiplied,
//   prod_i e(rho_i A_i, B_i) * e(sum_i rho_i acc_i, -gamma) * e(sum_i rho_i C_i, -delta) * e(-(sum_i rho_i) alpha, beta) == 1,
// n + 3 Miller loops in one launch and one final exponentiation.  Sound up to 2^-128 over the choice of rho (a batch that contains an
// invalid proof passes with that probability); when it reports failure, pcdhip_groth16_verify_prepared tells which proof is bad.
int pcdhip_groth16_verify_batch_rlc(pcdhip_ctx* ctx, const pcdhip_pvk* pvk, size_t n_proofs, const uint64_t* public_inputs_canonical,
                                    const uint64_t* proofs, const uint8_t* proofs_inf, const uint64_t* rho, int* all_ok) {
  return guarded([&]() -> int {
  if (!ctx || !pvk || !all_ok || (n_proofs && (!proofs || !rho)) || (pvk->num_inputs > 1 && n_proofs && !public_inputs_canonical) || n_proofs >= (1u << 16))
    return PCDHIP_E_ARG;
  *all_ok = 1;
  if (n_proofs == 0) return PCDHIP_OK;
  for (size_t i = 0; i < n_proofs; i++) if (!(rho[2 * i] | rho[2 * i + 1])) return PCDHIP_E_ARG;  // zero is not a challenge
  // While the batch's 3k Miller loops find a SIMD each (one wave per pairing: pairing_vm.hip.h) the k verifications side by side take the
  // time of one -- less than the scalings of the combination cost (k = 8: 5.5 against 17 ms on MNT4-298) -- and their answer is the
  // deterministic one.  The random linear combination is for batches that would otherwise queue on the chip.
  if (ctx->pairing_vm && 3 * n_proofs <= 1024) {
    std::vector<int> each(n_proofs, 0);
    int rc = pcdhip_groth16_verify_prepared(ctx, pvk, n_proofs, public_inputs_canonical, proofs, proofs_inf, each.data());
    if (rc) return rc;
    for (size_t i = 0; i < n_proofs; i++) if (!each[i]) *all_ok = 0;
    return PCDHIP_OK;
  }
  BIND();
  const int cid = pvk->curve_id, fr = kCurveFr[cid];
  const size_t l1 = (size_t)pcdhip_point_limbs(cid, 1), l2 = (size_t)pcdhip_point_limbs(cid, 2), pl = 2 * l1 + l2, k = n_proofs;
  const size_t sl = (size_t)kFieldLimbs[fr], ni = pvk->num_inputs, j1 = l1 / 2 * 3;
  // challenges as full-width canonical scalars (zero is not a challenge)
  std::vector<uint64_t> rw(k * sl, 0);
  for (size_t i = 0; i < k; i++) { rw[i * sl] = rho[2 * i]; rw[i * sl + 1] = rho[2 * i + 1]; if (!(rho[2 * i] | rho[2 * i + 1])) return PCDHIP_E_ARG; }
  // scalars of the gamma_abc combination: s_0 = sum rho_i, s_j = sum_i rho_i x_ij
  std::vector<uint64_t> s(ni * sl, 0);
  std::vector<const uint64_t*> pa(k), pb(k);
  for (size_t i = 0; i < k; i++) pa[i] = &rw[i * sl];
  scalar_lincomb(fr, pa.data(), nullptr, k, &s[0]);
  for (size_t j = 1; j < ni; j++) {
    for (size_t i = 0; i < k; i++) pb[i] = public_inputs_canonical + (i * (ni - 1) + (j - 1)) * sl;
    scalar_lincomb(fr, pa.data(), pb.data(), k, &s[j * sl]);
  }
  std::vector<uint64_t> sacc_j(j1), sacc(l1);
  uint8_t sacc_inf = 0;
  int rc = pcdhip_msm(ctx, pvk->abc, 0, s.data(), ni, sacc_j.data());
  rc = rc ? rc : pcdhip_to_affine(ctx, cid, 1, sacc_j.data(), 1, sacc.data(), &sacc_inf);
  if (rc) return rc;
  // rho_i A_i, rho_i C_i and (sum rho) alpha: 2k + 1 one-lane products in one launch
  const size_t np = 2 * k + 1;
  std::vector<uint64_t> pts(np * l1), ks(np * sl), prod(np * j1);
  for (size_t i = 0; i < k; i++) {
    const uint64_t* pr = proofs + i * pl;
    memcpy(&pts[i * l1], pr, l1 * 8);
    memcpy(&pts[(k + i) * l1], pr + l1 + l2, l1 * 8);
    if (proofs_inf && proofs_inf[3 * i]) memset(&pts[i * l1], 0, l1 * 8);
    if (proofs_inf && proofs_inf[3 * i + 2]) memset(&pts[(k + i) * l1], 0, l1 * 8);
    memcpy(&ks[i * sl], &rw[i * sl], sl * 8);
    memcpy(&ks[(k + i) * sl], &rw[i * sl], sl * 8);
  }
  memcpy(&pts[2 * k * l1], pvk->alpha.data(), l1 * 8);
  memcpy(&ks[2 * k * sl], &s[0], sl * 8);
  {
    const size_t in_b = np * l1 * 8, k_b = np * sl * 8, out_b = np * j1 * 8;
    TRY(ctx->aux_ws.ensure(AUX_MISC, in_b + k_b + out_b + 64));
    char* d = (char*)ctx->aux_ws.buf[AUX_MISC];
    TRY(hipMemcpyAsync(d, pts.data(), in_b, hipMemcpyHostToDevice, ctx->stream));
    TRY(hipMemcpyAsync(d + in_b, ks.data(), k_b, hipMemcpyHostToDevice, ctx->stream));
    TRY(g1_scale_entry(cid)(ctx->stream, (const uint32_t*)d, (const uint32_t*)(d + in_b), (uint32_t)(sl * 2), (uint32_t)np, (uint32_t*)(d + in_b + k_b)));
    TRY(hipMemcpyAsync(prod.data(), d + in_b + k_b, out_b, hipMemcpyDeviceToHost, ctx->stream));
    TRY(hipStreamSynchronize(ctx->stream));
  }
  std::vector<uint64_t> sc_j(j1), ra(k * l1), sc(l1), sal(l1);
  std::vector<uint8_t> ra_inf(k, 0);
  uint8_t sc_inf = 0, sal_inf = 0;
  rc = pcdhip_points_sum(ctx, cid, 1, &prod[k * j1], k, sc_j.data());
  rc = rc ? rc : pcdhip_to_affine(ctx, cid, 1, sc_j.data(), 1, sc.data(), &sc_inf);
  rc = rc ? rc : pcdhip_to_affine(ctx, cid, 1, prod.data(), k, ra.data(), ra_inf.data());
  rc = rc ? rc : pcdhip_to_affine(ctx, cid, 1, &prod[2 * k * j1], 1, sal.data(), &sal_inf);
  if (rc) return rc;
  negate_point(cid, 1, sal.data());
  // the k + 3 pairs of the single product
  const size_t nq = k + 3;
  std::vector<uint64_t> g1s(nq * l1), g2s(nq * l2), gt(pvk->gt_one.size());
  std::vector<uint8_t> inf1(nq, 0), inf2(nq, 0);
  for (size_t i = 0; i < k; i++) {
    memcpy(&g1s[i * l1], &ra[i * l1], l1 * 8);
    memcpy(&g2s[i * l2], proofs + i * pl + l1, l2 * 8);
    inf1[i] = ra_inf[i];
    if (proofs_inf) inf2[i] = proofs_inf[3 * i + 1];
  }
  memcpy(&g1s[k * l1], sacc.data(), l1 * 8);       memcpy(&g2s[k * l2], pvk->neg_gamma.data(), l2 * 8);       inf1[k] = sacc_inf;
  memcpy(&g1s[(k + 1) * l1], sc.data(), l1 * 8);   memcpy(&g2s[(k + 1) * l2], pvk->neg_delta.data(), l2 * 8);   inf1[k + 1] = sc_inf;
  memcpy(&g1s[(k + 2) * l1], sal.data(), l1 * 8);  memcpy(&g2s[(k + 2) * l2], pvk->beta.data(), l2 * 8);        inf1[k + 2] = sal_inf;
  rc = pairing_groups(ctx, cid, g1s.data(), inf1.data(), g2s.data(), inf2.data(), 1, nq, gt.data());
  if (rc) return rc;
  *all_ok = memcmp(gt.data(), pvk->gt_one.data(), gt.size() * 8) == 0 ? 1 : 0;
  return PCDHIP_OK;
  });
}

// ark-groth16 `Groth16::verify` for n proofs under one key (reference call site mod.rs:239, once per prior message of a merge node):
// process_vk once, then the prepared verification.  Deterministic: ok[i] per proof.
int pcdhip_groth16_verify_batch(pcdhip_ctx* ctx, int curve_id, const uint64_t* alpha_g1, const uint64_t* beta_g2, const uint64_t* gamma_g2,
                                const uint64_t* delta_g2, const uint64_t* gamma_abc_g1, const uint8_t* gamma_abc_inf, size_t num_inputs,
                                size_t n_proofs, const uint64_t* public_inputs_canonical, const uint64_t* proofs, const uint8_t* proofs_inf,
                                int* ok) {
  if (!ctx || !valid_curve(curve_id) || !alpha_g1 || !beta_g2 || !gamma_g2 || !delta_g2 || !gamma_abc_g1 || num_inputs < 1 ||
      (num_inputs > 1 && n_proofs && !public_inputs_canonical) || (n_proofs && (!proofs || !ok)) || n_proofs >= (1u << 20))
    return PCDHIP_E_ARG;
  if (n_proofs == 0) return PCDHIP_OK;
  pcdhip_pvk* pvk = nullptr;
  int rc = pcdhip_process_vk(ctx, curve_id, alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1, gamma_abc_inf, num_inputs, &pvk);
  if (rc) return rc;
  rc = pcdhip_groth16_verify_prepared(ctx, pvk, n_proofs, public_inputs_canonical, proofs, proofs_inf, ok);
  pcdhip_pvk_free(ctx, pvk);
  return rc;
}

int pcdhip_groth16_verify(pcdhip_ctx* ctx, int curve_id, const uint64_t* alpha_g1, const uint64_t* beta_g2, const uint64_t* gamma_g2,
                          const uint64_t* delta_g2, const uint64_t* gamma_abc_g1, const uint8_t* gamma_abc_inf, size_t num_inputs,
                          const uint64_t* public_inputs_canonical, const uint64_t* proof, const uint8_t* proof_inf, int* ok) {
  if (!proof || !ok) return PCDHIP_E_ARG;
  *ok = 0;
  return pcdhip_groth16_verify_batch(ctx, curve_id, alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1, gamma_abc_inf, num_inputs, 1,
                                     public_inputs_canonical, proof, proof_inf, ok);
}

int pcdhip_groth16_last_timings(pcdhip_ctx* ctx, float out_ms[8]) {
  if (!ctx || !out_ms) return PCDHIP_E_ARG;
  memcpy(out_ms, ctx->g16_ms, sizeof ctx->g16_ms);
  return PCDHIP_OK;
}

}  // extern "C"
