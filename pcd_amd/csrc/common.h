// Host-side plumbing shared by the translation units of libpcdhip.so: context, handles and the
// per-field / per-group / per-curve entry tables (each instantiation lives in its own object file so
// the eight group instantiations compile in parallel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/pcdhip.h"
#include "msm.hip.h"
#include "vm_tables.h"
// batches of up to this many pairings run one WAVE per pairing (pairing_vm.hip.h); larger ones one lane per pairing (pairing.hip.h)
#define PCD_VM_MAX_PAIRS 4096u

namespace pcd {

struct FftTables {  // per (field, log_n): powers of w, w^-1, g, g^-1 (*1/n folded in), resident for reuse
  uint32_t *tw_fwd = nullptr, *tw_inv = nullptr, *coset = nullptr, *coset_inv_scaled = nullptr;
  // the inter-pass twiddles of the FIRST pass of a multi-pass transform, laid out in the order that pass stores (fft.hip.h "first-pass twiddles"):
  // tw0[o] = w^((o >> d0) (o mod 2^d0)); null for single-pass sizes
  uint32_t *tw0_fwd = nullptr, *tw0_inv = nullptr;
  uint32_t consts[6 * 32] = {0};  // host copy of w, w^-1, g, g^-1, 1/n, 1/Z(g) (computed once, on the device)
};

}  // namespace pcd

struct pcdhip_buf {
  int field_id;
  size_t n;        // elements
  uint32_t* dptr;  // n * words u32
};
struct pcdhip_bases {
  int curve_id, group_id;
  size_t n;
  uint32_t* dptr;  // groups * n affine points, (0,0) = infinity; group g holds 2^(c * Wg * g) P_i
  int c;           // window bits fixed at upload when groups > 1 (0 otherwise)
  int groups;      // 1 = plain bases
  uint32_t* inf_bits = nullptr;  // device bitmap of the points at infinity (bit i: point i), null when the vector holds none
  pcd::MsmBasesView view(size_t offset) const { return {dptr, (uint32_t)n, (uint32_t)offset, c, groups, inf_bits}; }
  // uploaded through a multi-device context: the vector is cut into contiguous point ranges, shard g (points
  // [shard_lo[g], shard_lo[g + 1])) resident on the context's device g as an ordinary handle; dptr is null then
  std::vector<pcdhip_bases*> shards;
  std::vector<size_t> shard_lo;
};
namespace pcd {
// One constraint matrix on the device (built by upload_csr_to, capi.hip).  Inside every row the entries whose coefficient is a SMALL
// integer (|c| <= 32: the +-1 of linear combinations and booleanity constraints, the 2, 3, 4 of range checks -- most of what
// `cs.finalize()` leaves in a verifier circuit) come first: `nl[r]` of them, coefficient in `lc` (int8); the others follow with
// their coefficients as field elements (device image) in `coeff` (indexed like col; the slots of light entries are unused).  Rows with
// more than SPMV_LONG_ROW entries are listed in `long_rows`: they get a wave each instead of a lane.
constexpr uint32_t SPMV_LONG_ROW = 16;
struct DevCsr {
  const uint64_t* rp; const uint32_t* col; const uint32_t* coeff; uint32_t rows;
  const uint32_t* nl = nullptr; const int8_t* lc = nullptr; const uint32_t* long_rows = nullptr; uint32_t n_long = 0;
};
}
struct pcdhip_g16_pk {
  // constraint matrices kept resident by pcdhip_g16_pk_set_r1cs (fixed per circuit, like the key)
  void* r1cs_dev = nullptr;
  pcd::DevCsr mats[3];
  uint32_t rows = 0;
  int curve_id = 0;
  uint64_t num_vars = 0, num_inputs = 0, domain_size = 0;
  // a / b / l queries carry four trailing slots [delta_r, delta_s, delta_rs, vk point] (see inst_g16.hip)
  pcdhip_bases *a_query = nullptr, *b_g1_query = nullptr, *b_g2_query = nullptr, *h_query = nullptr, *l_query = nullptr;
  // the same a' / b' / l' queries laid out for a SMALLER window (round 5, pcdhip_groth16_set_sparse_window): what a proof over a witness-like
  // assignment -- a few per cent general scalars, the rest zero or one -- runs its four assignment MSMs on; null when not built
  pcdhip_bases *a_sparse = nullptr, *b_g1_sparse = nullptr, *b_g2_sparse = nullptr, *l_sparse = nullptr;
  // multi-device key: shard g holds the entry range [lo[g], lo[g + 1]) of the a' / b' / l' queries (num_vars + 4 entries) and the
  // range [hlo[g], hlo[g + 1]) of the h query on the context's device g; the parent's own query handles are null
  std::vector<pcdhip_g16_pk*> shards;
  std::vector<size_t> lo, hlo;
  uint64_t h_len = 0;
  // points at infinity among this device's entries of the a / b queries (a variable that no row of A / B mentions), and whether b_g1 and
  // b_g2 flag the same entries (they do in any key a setup made): decides which MSMs of a proof share a sort (G16Run::launch_assignment)
  size_t a_inf_count = 0, b_inf_count = 0;
  bool b_inf_same = false;
};
struct pcdhip_ctx {
  // multi-device context (pcdhip_init_devices with more than one id): peers[0] == this, peers[g] = the sub-context of device g;
  // empty for an ordinary context
  std::vector<pcdhip_ctx*> peers;
  int device;
  hipStream_t stream;
  pcd::MsmWorkspace msm_ws;
  pcd::MsmWorkspace aux_ws;  // fft ping-pong, witness-map vectors, groth16 scratch
  // the MSMs of a Groth16 proof run concurrently, each on its own stream with its own workspace: the
  // latency-bound bucket-reduction tail of one overlaps the throughput-bound accumulation of the others.
  // Streams 0 and 1 are created with the highest priority, 2 with the default one, 3..5 with the lowest.
  hipStream_t g16_streams[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  pcd::MsmWorkspace g16_ws[6];
  hipEvent_t g16_ready = nullptr, g16_begin[6] = {nullptr}, g16_end[6] = {nullptr};
  // added to the window bits msm_pick_window chooses for vectors uploaded next (0 outside pcdhip_g16_pk_upload): the lone MSM's best
  // window is one bit too wide inside a proof, whose time is the SUM of its kernels' work -- bucket reductions and fix-ups included
  int msm_c_bias = 0;
  pcd::MsmSharedSort g16_share;  // the sort of the assignment's digits, made once per proof and used by four MSMs
  pcd::MsmSharedSort g16_share_b;  // ... and a second one without the entries of the B queries' points at infinity (b_g1 / b_g2 only)
  std::map<uint64_t, pcd::FftTables> fft_tables;
  int msm_c = 0;
  uint32_t msm_chunk = 0;
  int msm_sort = 0;     // 0 LDS partition sort, 1 single-pass binning, 2 two-pass counting sort
  int precompute = -1;  // bases uploaded through this context: -1 full (one bucket window), 0 none, k > 1 groups
  size_t precompute_budget = 0;  // bytes one base vector may occupy with its copies (0: whatever hipMalloc grants)
  bool msm_profile = false;
  pcd::MsmTimings msm_tm;
  float fft_ms[8] = {0};
  int fft_passes = 0;
  float g16_ms[8] = {0};
  int g16_assembly = 0;  // s*A and r*B_1: 0 automatic, 1 folded into two extra MSMs, 2 chained one-lane products
  int g16_sparse_window = 0;             // pcdhip_groth16_set_sparse_window: 0 off (default: opt-in), -1 automatic (large whole keys, when the copies fit), > 0 that many window bits
  uint32_t g16_last_general = 0;         // general scalars (neither 0 nor 1) of the last proof's assignment, when counted
  int g16_last_sparse = 0;               // 1: the last proof ran its assignment MSMs on the sparse-window copies
  hipEvent_t t0 = nullptr, t1 = nullptr;
  // pcdhip_msm_submit / collect: up to PIPE_SLOTS independent MSMs in flight, each on one of the side streams (g16_streams[2 + slot],
  // with that stream's workspace); results land in page-locked host memory
  static constexpr int PIPE_SLOTS = 4;
  bool pipe_busy[PIPE_SLOTS] = {false, false, false, false};
  bool pipe_partial[PIPE_SLOTS] = {};   // the slot's last ticket was released by pcdhip_msm_ticket_wait: its error word is checked at the next submit
  hipEvent_t pipe_done[PIPE_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
  uint64_t* pipe_host = nullptr;  // PIPE_SLOTS x PIPE_HOST_WORDS u64
  uint32_t* count_host = nullptr; // one page-locked word + an event: the count of general scalars of a proof's assignment (G16Run::decide_sparse)
  hipEvent_t count_ev = nullptr;
  static constexpr size_t PIPE_HOST_WORDS = 256;  // >= one Jacobian point in the C-ABI image (216 u64 for Fq3-753) + the error word
  size_t pipe_out_bytes[PIPE_SLOTS] = {0, 0, 0, 0};
  int pipe_next = 0;
  // program tables of the wave-per-pairing VM (pairing_vm.hip.h), uploaded per curve on first use
  void* vm_block[4] = {nullptr, nullptr, nullptr, nullptr};
  pcd::VmCurveTables vm_tables[4] = {};
  bool pairing_vm = true;  // small batches of pairings run one wave per pairing (pcdhip_pairing_set_mode)
  hipEvent_t xstream_ev = nullptr;  // pcdhip_stream_wait: ordering against a caller-owned stream (e.g. the RCCL stream)
  hipEvent_t wm_ev = nullptr;       // sharded prove: this device's chain of the witness map has landed in device 0's buffers
  bool wm_split = true;             // pcdhip_groth16_set_witness_split
  int g16_schedule = 0;             // pcdhip_groth16_set_schedule: 0 assignment MSMs first, the map under them (default: fastest, measured again in round 5); 1 the map first, then all MSMs at once; 2 the map first, then the accumulate lane
  // the accumulate lane (msm.hip.h MsmLane; schedule 2 only): one stream confined by a CU mask to all but `lane_reserve` compute units, for
  // the accumulate kernels of the MSMs that share the device (a proof's five; submitted MSMs).  The masked stream comes from a process-wide
  // pool and is never destroyed (capi.hip masked_lane_of); `lane_stream` is the owned, unmasked one of reserve 0.
  hipStream_t lane_stream = nullptr;
  pcd::MsmLane lane;
  int lane_reserve = -1;            // CUs the lane leaves to the other streams (pcdhip_set_lane_reserve; -1: default, PCDHIP_LANE_RESERVE or 8; 0: no mask)
  bool pipe_lane = true;            // pcdhip_msm_submit under schedule 2: accumulations on the lane
  std::string last_hip_error;
};

namespace pcd {

// ---- per-group entries (inst_group.hip, one object per group) ------------------------------------
typedef hipError_t (*MsmFn)(MsmWorkspace&, hipStream_t, const MsmBasesView& bases, const uint32_t* scalars, uint32_t n,
                            uint32_t* out_dev, int c, uint32_t chunk, int sort_mode, MsmTimings* tm, MsmSharedSort* share,
                            int share_role);
typedef hipError_t (*PrecomputeFn)(hipStream_t, uint32_t* pts, uint32_t n, int groups, int shift);
typedef hipError_t (*PointsSumFn)(hipStream_t, const uint32_t* jac_dev, uint32_t n, uint32_t* scratch /* 64 Jacobians */, uint32_t* out_dev);
typedef hipError_t (*ToAffineFn)(hipStream_t, const uint32_t* jac_dev, uint32_t n, uint32_t* aff_dev);
struct GroupEntry {
  int point_words;      // u32 words per affine point, device-internal image (Jacobian = 3/2 of it)
  int base_stride_words;  // u32 words between consecutive points of a resident base array (>= point_words: records may be padded)
  int point_abi_words;  // u32 words per affine point at the C-ABI
  int scalar_words;     // u32 words per canonical scalar
  int scalar_bits;
  MsmFn msm;
  PrecomputeFn precompute;
  hipError_t (*points_in)(hipStream_t, const uint32_t* abi_dev, uint32_t n, uint32_t* internal_dev);  // affine, ABI -> internal
  hipError_t (*jac_out)(hipStream_t, const uint32_t* internal_dev, uint32_t n, uint32_t* abi_dev);   // Jacobian, internal -> ABI
  PointsSumFn points_sum;  // ABI in, ABI out
  // out[s] = sum over g < parts of in[g * part_stride_words + s * (Jacobian words)], s < slots; device image in and out (the
  // combine step of a proof sharded over devices: every slot is one MSM result, every part one device's partial)
  hipError_t (*jac_sum_parts)(hipStream_t, const uint32_t* in, size_t part_stride_words, uint32_t parts, uint32_t slots, uint32_t* out);
  ToAffineFn to_affine;    // ABI in, ABI out
  // out[i] = k_i * base (fixed_base.hip.h): base / out in the C-ABI affine image, scalars canonical words; scratch sizes in
  // u32 words: fb_table_words (window table + per-window powers), 3/2 * point_words per scalar for the Jacobian results
  size_t fb_table_words;
  hipError_t (*fixed_base)(hipStream_t, const uint32_t* base_abi, const uint32_t* scalars, uint32_t n, uint32_t* table_scratch,
                           uint32_t* jac_scratch, uint32_t* out_abi, uint8_t* out_inf);
  // prepared public inputs (fixed_base.hip.h): window tables of the bases 1 .. ni - 1 (fb_table_words each, consecutive), then
  // acc_i = base_0 + sum_j x_ij base_j for k proofs in one launch (scratch: k x 64 Jacobians), C-ABI affine out -- or, with out_z_abi,
  // the Jacobian (X, Y) in out_abi and Z in out_z_abi (no inversion); proof i's result lands out_stride points after proof i - 1's
  hipError_t (*fb_tables)(hipStream_t, const uint32_t* bases_abi, uint32_t ni, uint32_t* tables);
  hipError_t (*fb_inputs)(hipStream_t, const uint32_t* tables, const uint32_t* base0_abi, uint32_t ni, const uint32_t* scalars, uint32_t k,
                          uint32_t* scratch, uint32_t* out_abi, uint8_t* out_inf, uint32_t* out_z_abi, uint32_t out_stride);
};
const GroupEntry& group_entry(int curve_id, int group_id);  // group_id 1 / 2

// ---- per-field entries (inst_field.hip) ------------------------------------------------------------
struct FieldEntry {
  int words;      // u32 words per element, device-internal image
  int abi_words;  // u32 words per element at the C-ABI
  int two_adicity;
  // fills tables for a domain of 2^log_n (allocates into `t`)
  hipError_t (*fft_make_tables)(hipStream_t, int log_n, FftTables* t);
  // x -> result in `x` (uses `tmp` as the ping-pong partner; both n elements)
  hipError_t (*fft_run)(hipStream_t, const FftTables& t, uint32_t* x, uint32_t* tmp, int log_n, int inverse, int coset,
                        float* pass_ms, int* npasses);
  // mode 0: ABI Montgomery -> internal, 1: internal -> ABI Montgomery, 2: internal -> canonical words,
  // 3: ABI Montgomery -> canonical words
  hipError_t (*convert)(hipStream_t, const uint32_t* in, uint32_t* out, uint32_t n, int mode);
  // a[i] = <A_i, z> for the rows, a[nc + j] = z[j] for inputs when `append_inputs`, zero padding to n
  hipError_t (*spmv)(hipStream_t, const DevCsr& m, const uint32_t* z, uint32_t num_inputs, int append_inputs, uint32_t n, uint32_t* out);
  // the C-ABI image of the small integer c (|c| <= 32) as a field element: what upload_csr_to looks for among the coefficients (host)
  void (*small_abi)(int c, uint32_t* out_abi_words);
  // a = (a * b - c) / Z(g) on the coset of size 2^log_n
  hipError_t (*mul_sub_divz)(hipStream_t, const FftTables& t, uint32_t* a, const uint32_t* b, const uint32_t* c, int log_n);
  // mixed-radix domain n = m * 2^a (ark-poly MixedRadixEvaluationDomain): tables, transform, pointwise step
  hipError_t (*mixed_make_tables)(hipStream_t, uint32_t n, uint32_t m, FftTables* t);
  hipError_t (*mixed_run)(hipStream_t, const FftTables& t, const FftTables& t2, uint32_t* x, uint32_t* tmp, uint32_t m, int a,
                          int inverse, int coset);
  hipError_t (*mixed_mul_sub_divz)(hipStream_t, const FftTables& t, uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t n);
  // out = canonical words of in[i] * k (k: ABI Montgomery element on the device, or null for 1); element 0 read as 1 if asked
  hipError_t (*scale_canon)(hipStream_t, const uint32_t* in_internal, const uint32_t* k_abi, uint32_t* out, uint32_t n, int first_is_one);
  // Groth16 generator scalars (inst_field.hip): phase 0 writes the Lagrange coefficients u (n elements) at tau and the
  // constants block (setup_consts elements); phase 1 turns the transposed mat-vecs At, Bt, Ct (m elements) into the canonical
  // scalars of the a / b / (gamma_abc | l) / h queries.  domain_consts = FftTables::consts of the domain.
  int setup_consts;
  hipError_t (*setup_scalars)(hipStream_t, const void* domain_consts, const uint32_t* toxic_abi, uint32_t n, uint32_t nc, uint32_t m,
                              uint32_t ni, const uint32_t* at, const uint32_t* bt, const uint32_t* ct, uint32_t* u, uint32_t* consts_dev,
                              uint32_t* err_dev, uint32_t* a_can, uint32_t* b_can, uint32_t* t_can, uint32_t* h_can, uint32_t* b2_can,
                              int phase);
  // fft_run over `batch` vectors laid back to back in x (and in tmp): one launch per pass for all of them (the three chains of a witness map)
  hipError_t (*fft_run_batched)(hipStream_t, const FftTables& t, uint32_t* x, uint32_t* tmp, int log_n, int inverse, int coset, int* npasses,
                                uint32_t batch);
  // the three mat-vecs of a witness map in one launch (two with long rows): out, out + stride, out + 2 stride; matrix 0 appends the inputs
  hipError_t (*spmv3)(hipStream_t, const DevCsr mats[3], const uint32_t* z, uint32_t num_inputs, uint32_t n, uint32_t* out, size_t stride_words);
};
const FieldEntry& field_entry(int field_id);

// ---- per-curve entries (inst_g16.hip) --------------------------------------------------------------
struct CurveEntry {
  // three scalar tails of 4 canonical words-vectors each, from ABI Montgomery (r, s) on the device:
  //   t1 = [r, s, -rs, 1]   ts = s * t1   tr = r * t1
  hipError_t (*prepare_scalars)(hipStream_t, const uint32_t* rs_dev, uint32_t* t1, uint32_t* ts, uint32_t* tr);
  size_t proof_abi_bytes;
  // msm_g1: Jacobian points on device (internal image) in the order h, l', A, s*A, r*B1;  msm_g2: B (G2)
  // proof_out: C-ABI affine A || B || C with C = s*A + r*B1 + l' + h
  hipError_t (*assemble)(hipStream_t, const uint32_t* msm_g1, const uint32_t* msm_g2, uint32_t* proof_out);
  // out = k * in for one G1 point (device image; k canonical words); one lane, meant to overlap other work
  hipError_t (*scale_g1)(hipStream_t, const uint32_t* in, const uint32_t* k, uint32_t* scratch, uint32_t* out);
};
const CurveEntry& curve_entry(int curve_id);

// ---- per-curve pairing entries (inst_pairing.hip) -----------------------------------------------------
struct PairingEntry {
  int gt_words;           // u32 words of one GT element (Fq4 / Fq6) at the C-ABI
  int gt_internal_words;  // ... in the device image (scratch sizing)
  // gt_out[g] = final_exp(prod_{i < per} miller(P_{g per + i}, Q_{g per + i})) for g < groups; scratch: groups * per GT elements;
  // vm: this device's copy of the curve's VM program tables (pairing_vm.hip.h; null = the lane-per-pairing kernels only)
  // g1z_dev (nullable; wave-per-pairing form only): the Z coordinates of the G1 points, which are then Jacobian (X, Y in g1_dev)
  hipError_t (*multi_pairing)(hipStream_t, const uint32_t* g1_dev, const uint32_t* g1z_dev, const uint32_t* g2_dev, uint32_t groups, uint32_t per,
                              uint32_t* scratch, uint32_t* gt_out, const VmCurveTables* vm);
  hipError_t (*vm_upload)(hipStream_t, void** block, VmCurveTables* out);
};
const PairingEntry& pairing_entry(int curve_id);

}  // namespace pcd
