/* pcdhip.h -- C ABI of libpcdhip.so: MI355X (gfx950) prover arithmetic for arkworks-style PCD.
 *
 * Drop-in boundary for the hot path behind `ECCyclePCD::prove`
 * (/root/reference src/ec_cycle_pcd/mod.rs:92-181): the two calls `IC::MainSNARK::prove`
 * (mod.rs:171) and `IC::HelpSNARK::prove` (mod.rs:179) spend their time in upstream
 * ark-ec `VariableBaseMSM::multi_scalar_mul`, ark-poly `Radix2EvaluationDomain::{fft,ifft,
 * coset_fft,coset_ifft}_in_place` and ark-groth16 `R1CSToQAP::witness_map` / `create_proof`
 * (Cargo.toml:17-19,39: git dependencies, not vendored).  Each entry point below names the
 * upstream function it replaces; INTEGRATION.md shows the Rust `extern "C"` binding and the
 * `SNARK` impl (the plug-in seam `ECCyclePCDConfig`, mod.rs:24-33) that calls them.
 *
 * Encodings (exactly the in-memory image of the upstream types, so the Rust side is a memcpy):
 *   field element   L little-endian uint64_t limbs (L = 5 for the 298-bit fields, 12 for the
 *                   753-bit fields), Montgomery form with R = 2^(64 L)       [ark-ff Fp320/Fp768]
 *   scalar (MSM)    L limbs, canonical (non-Montgomery, reduced: < r)        [`into_repr()`]
 *   Fq2 / Fq3       consecutive base-field elements c0, c1 (, c2)
 *   affine point    x || y, infinity in a separate byte array (nullable = no point at infinity)
 *   Jacobian point  X || Y || Z, Z = 0 means infinity                        [GroupProjective]
 * Ownership: the caller owns every host buffer for the duration of the call only; device
 * objects are opaque handles with explicit free functions.  No RNG inside the library: the
 * Groth16 blinding factors r, s are inputs.  Thread-safety: one pcdhip_ctx per host thread;
 * handles may be shared by contexts on the same device once created.
 * Every function returns 0 on success or a negative PCDHIP_E_* code; nothing aborts or throws.
 * There is NO CPU fallback: without a usable GPU every call fails with PCDHIP_E_NO_DEVICE.
 */
#ifndef PCDHIP_H
#define PCDHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { PCDHIP_MNT4_298 = 0, PCDHIP_MNT6_298 = 1, PCDHIP_MNT4_753 = 2, PCDHIP_MNT6_753 = 3 };  /* curve_id */
/* field_id: 0 = MNT4-298 Fq (= MNT6-298 Fr), 1 = MNT4-298 Fr (= MNT6-298 Fq),
 *           2 = MNT4-753 Fq (= MNT6-753 Fr), 3 = MNT4-753 Fr (= MNT6-753 Fq) */
enum { PCDHIP_F298A = 0, PCDHIP_F298B = 1, PCDHIP_F753A = 2, PCDHIP_F753B = 3 };
enum { PCDHIP_G1 = 1, PCDHIP_G2 = 2 };  /* group_id */

enum {
  PCDHIP_OK = 0,
  PCDHIP_E_ARG = -1,              /* null pointer / bad id / inconsistent sizes */
  PCDHIP_E_SIZE_UNSUPPORTED = -2, /* e.g. log_n above the field's 2-adicity (mixed-radix domain needed) */
  PCDHIP_E_NO_DEVICE = -3,
  PCDHIP_E_OOM = -4,
  PCDHIP_E_HIP = -5,              /* any other HIP runtime failure; see pcdhip_last_hip_error */
  PCDHIP_E_PREV_TICKET = -6       /* pcdhip_msm_submit_partial only: the PREVIOUS ticket of the slot had an unreduced scalar (its result is
                                     wrong); THIS submission was enqueued all the same and *ticket is valid */
};

typedef struct pcdhip_ctx pcdhip_ctx;
typedef struct pcdhip_bases pcdhip_bases;     /* device-resident affine base points (one query vector) */
typedef struct pcdhip_buf pcdhip_buf;         /* device-resident vector of field elements / scalars */
typedef struct pcdhip_g16_pk pcdhip_g16_pk;   /* device-resident Groth16 proving key */

const char* pcdhip_strerror(int code);
int pcdhip_device_count(void);
/* One context = one device + one HIP stream + reusable workspaces. */
int pcdhip_init(int device_id, pcdhip_ctx** out);
/* a context over a device LIST (one host thread drives all of them).  n_dev = 1 is pcdhip_init.  With more devices
 * the context shards by contiguous point range: pcdhip_bases_upload places range g of the vector on device g and
 * pcdhip_msm (host scalars) runs one MSM per device, gathers the partial results on device 0 over xGMI (peer copies) and sums them
 * there; pcdhip_g16_pk_upload shards every query the same way and pcdhip_groth16_prove runs the five MSMs of the proof on every
 * device over its range (the witness map on device 0, slices of h handed device to device) -- BASELINE configs[4]'s merge node,
 * "its MSMs use all GPUs".  Results are bit-identical to the single-device ones.  Independent PCD DAG branches need no such
 * context: one ordinary context per (host thread, device), N threads x N contexts (pcd_amd/dag.py).  The same id may appear more
 * than once (two logical shards on one device; the 1-GPU tests do that).  Device-resident scalars (pcdhip_msm_dev*) and the
 * FFT / pairing / setup entry points address device 0 only; handles created through a multi-device context belong to it. */
int pcdhip_init_devices(const int* device_ids, int n_dev, pcdhip_ctx** out);
int pcdhip_ctx_devices(const pcdhip_ctx* ctx);  /* number of devices of the context */
void pcdhip_destroy(pcdhip_ctx* ctx);
int pcdhip_sync(pcdhip_ctx* ctx);
/* Page-locked host memory for the buffers that cross PCIe on every proof (the assignment z: 42 MB at 2^20 x 298 bits):
 * host-to-device copies from it run at full PCIe rate and without a staging copy.  Optional -- every entry point
 * accepts ordinary host pointers. */
int pcdhip_host_alloc(size_t bytes, void** out);
void pcdhip_host_free(void* p);
const char* pcdhip_last_hip_error(pcdhip_ctx* ctx);
/* Static facts (no GPU needed). */
int pcdhip_field_limbs(int field_id);                 /* L */
int pcdhip_curve_base_field(int curve_id);            /* field_id of Fq */
int pcdhip_curve_scalar_field(int curve_id);          /* field_id of Fr */
int pcdhip_point_limbs(int curve_id, int group_id);   /* uint64 limbs per affine point (x||y) */

/* ---- device vectors ------------------------------------------------------------------------- */
int pcdhip_buf_upload(pcdhip_ctx* ctx, int field_id, const uint64_t* host, size_t n, pcdhip_buf** out);
int pcdhip_buf_alloc(pcdhip_ctx* ctx, int field_id, size_t n, pcdhip_buf** out);
int pcdhip_buf_download(pcdhip_ctx* ctx, const pcdhip_buf* buf, uint64_t* host, size_t n);
void pcdhip_buf_free(pcdhip_ctx* ctx, pcdhip_buf* buf);

/* ---- K3/K4: variable-base MSM ----------------------------------------------------------------
 * Replaces ark-ec `VariableBaseMSM::multi_scalar_mul(&bases[offset..offset+n], &scalars)`.
 * Bases are uploaded once (proving-key residency) and addressed by handle. */
int pcdhip_bases_upload(pcdhip_ctx* ctx, int curve_id, int group_id, const uint64_t* xy_mont,
                        const uint8_t* inf_flags, size_t n, pcdhip_bases** out);
void pcdhip_bases_free(pcdhip_ctx* ctx, pcdhip_bases* bases);
int pcdhip_msm(pcdhip_ctx* ctx, const pcdhip_bases* bases, size_t offset, const uint64_t* scalars_canonical,
               size_t n, uint64_t* out_xyz_mont);
/* Same with scalars already resident on the device (element `scalar_offset` onwards). */
int pcdhip_msm_dev(pcdhip_ctx* ctx, const pcdhip_bases* bases, size_t offset, const pcdhip_buf* scalars,
                   size_t scalar_offset, size_t n, uint64_t* out_xyz_mont);
/* Throughput form of pcdhip_msm_dev for a caller with several INDEPENDENT MSMs to make (the commitments of a Marlin round over one
 * resident committer key, BASELINE configs[3]; consecutive steps of a benchmark): submit enqueues the MSM on one of the context's four
 * side streams (own stream and workspace, round robin) and returns at once with a ticket; collect waits for that submission and
 * returns its result (PCDHIP_E_ARG as from pcdhip_msm_dev for an unreduced scalar).  The latency-bound bucket reduction of one MSM
 * then overlaps the sort and accumulation of the next.  At most four tickets may be outstanding (a fifth submit fails with
 * PCDHIP_E_ARG); pcdhip_groth16_prove refuses to run while any is (it uses the same side streams).  Not for sharded bases. */
int pcdhip_msm_submit(pcdhip_ctx* ctx, const pcdhip_bases* bases, size_t offset, const pcdhip_buf* scalars, size_t scalar_offset, size_t n,
                      int* ticket);
int pcdhip_msm_collect(pcdhip_ctx* ctx, int ticket, uint64_t* out_xyz_mont);
/* The same for the shard of a process-per-GPU run: the Jacobian partial is left in device memory of the caller
 * -- slot `*ticket` (0..3) of `out_xyz_device_slots`, four slots `slot_stride_bytes` apart: the send buffers of the RCCL
 * all-gather -- instead of coming back to the host, and pcdhip_msm_ticket_wait makes `other_stream` (a hipStream_t: the stream the
 * collective is enqueued on) wait for that MSM -- no host wait -- and frees the ticket.  Before a slot is written again its reader
 * must have been ordered in front of the context's stream (pcdhip_stream_wait direction 0), which every submission follows.
 * Several shard MSMs then overlap each other and the exchange of earlier ones, like pcdhip_msm_submit / collect on one GPU.
 * An unreduced scalar (PCDHIP_E_ARG from pcdhip_msm_dev) is reported on this path by the NEXT submission that reuses the ticket's slot, as
 * PCDHIP_E_PREV_TICKET: a code of its own, so that a pipelined caller can tell "an earlier MSM of this slot was wrong" from "this submission
 * was refused" -- the new submission IS enqueued and *ticket valid in that case.  pcdhip_msm_ticket_status polls a released ticket's word
 * without submitting (e.g. after the last step of a loop).  pcdhip_msm_submit never returns PCDHIP_E_PREV_TICKET: when it reuses a slot
 * whose last user was released by pcdhip_msm_ticket_wait it clears that slot's deferred word unreported (poll pcdhip_msm_ticket_status
 * first when mixing the two forms). */
int pcdhip_msm_submit_partial(pcdhip_ctx* ctx, const pcdhip_bases* bases, size_t offset, const pcdhip_buf* scalars, size_t scalar_offset,
                              size_t n, uint64_t* out_xyz_device_slots, size_t slot_stride_bytes, int* ticket);
int pcdhip_msm_ticket_wait(pcdhip_ctx* ctx, int ticket, void* other_stream);
/* The error word of the last MSM that used `slot` (0 .. 3) and was released by pcdhip_msm_ticket_wait: waits for that MSM, then
 * PCDHIP_OK, or PCDHIP_E_PREV_TICKET when one of its scalars was not reduced.  PCDHIP_E_ARG for a slot that is busy or was never released. */
int pcdhip_msm_ticket_status(pcdhip_ctx* ctx, int slot);
/* Precomputed window-shifted copies of bases uploaded AFTER this call through this context
 * (HBM capacity traded against the serial window combine; the proving key of a PCD is fixed for the whole
 * computation, so the one-time cost amortises over every step):
 *   -1 (default) one copy per scalar window: a single bucket window, no doubling chain at all;
 *    0           none (W bucket windows + Horner combine, as upstream);   k >= 2  k copies.
 * Falls back to fewer copies when device memory does not suffice (the Horner combine comes back for the windows that share a
 * copy); pcdhip_bases_info reports the number of copies a handle actually holds, so the fallback is never silent. */
int pcdhip_set_precompute(pcdhip_ctx* ctx, int mode);
/* Upper bound, in bytes, on what ONE base vector uploaded afterwards through this context may occupy with its copies (0, the
 * default: no bound other than the device's free memory).  A vector whose copies would exceed it gets fewer copies exactly as if
 * hipMalloc had failed (halving until it fits; at worst the plain vector, which is always accepted).  For hosts that keep several
 * proving keys resident: the full set of copies of ONE MNT4-753 key at 2^22 constraints is ~195 GB (BASELINE configs[4]). */
int pcdhip_set_precompute_budget(pcdhip_ctx* ctx, size_t bytes_per_vector);
/* The bound currently set (a layer that changes it for one upload puts the host's own setting back afterwards: rust/src/s2.rs). */
int pcdhip_get_precompute_budget(pcdhip_ctx* ctx, size_t* bytes_per_vector);
/* The plan an MSM of n pairs over these bases runs with (n = 0: the whole vector): signed-digit window bits c, scalar
 * windows W = ceil((bits + 1) / c), and how many window-shifted copies of the vector are resident (1 = none).  bench.py
 * prices the accumulate kernel's executed multiply-adds from it. */
int pcdhip_bases_info(const pcdhip_bases* bases, size_t n, int* window_bits, int* windows, int* copies);
/* Order the context's stream against a caller-owned HIP stream without a host wait (the RCCL stream of the exchange step,
 * SURVEY.md 8e): direction 0 -- work queued on the context's stream from now on waits for everything already queued on
 * `other_stream`; direction 1 -- `other_stream` waits for the context's stream.  `other_stream` is a hipStream_t (NULL =
 * the legacy default stream) of the same device and HIP runtime. */
int pcdhip_stream_wait(pcdhip_ctx* ctx, void* other_stream, int direction);
/* Tuning / introspection: window bits (0 = automatic), sorted entries per lane (0 = default). */
int pcdhip_msm_config(pcdhip_ctx* ctx, int window_bits, int chunk);
/* Sorting strategy of the (bucket, base) entries: 0 (default) MSD partition through LDS (per-workgroup bin
 * histograms, one global atomic per workgroup and bin, per-bin LDS counting sort); 1 single-pass binning into
 * per-bucket slots with an on-device fallback when a bucket overflows; 2 two-pass counting sort with one global
 * atomic per entry (also what small inputs use). */
int pcdhip_msm_set_sort(pcdhip_ctx* ctx, int mode);
/* Form of the bucket accumulation.  mode 0 (default) and 1: running sums (Jacobian / XYZZ mixed additions); 2: for the 753-bit G1
 * groups a PAIR TREE of affine additions whose inversions are shared (5M + 1S per addition instead of 7M + 4S, one divstep inversion per
 * lane and level; msm.hip.h msm_pair_tree_kernel) -- measured -9 % on the accumulation and -5 % on an MSM of 2^20 uniform scalars,
 * +5 % on witness-like ones, hence opt-in; other groups ignore the setting.  chunk (0 = by size, else 2 .. 1024 entries
 * per lane; the per-wave scratch grows with it: ~12 GB for the whole chip at the default cap of 640) and min_pairs (0 = default 12: the
 * smallest batch an inversion is spent on) are tuning and test knobs.  Results are identical in every mode. */
int pcdhip_msm_set_accumulate(pcdhip_ctx* ctx, int mode, int chunk, int min_pairs);
/* Per-stage device time of the last MSM (HIP events on the context's stream), milliseconds:
 * [digits, scan, scatter, accumulate, fixup, tail, horner, total]; enable with on != 0. */
int pcdhip_msm_profile(pcdhip_ctx* ctx, int on);
int pcdhip_msm_last_timings(pcdhip_ctx* ctx, float out_ms[8]);
/* ... and what the device planned for that MSM: out[0] = entries of the sorted (bucket, base) list -- n x windows less the zero digits, the
 * scalars equal to zero or one and the bases at infinity --, out[1] = entries per lane of the accumulate kernel chosen for it. */
int pcdhip_msm_last_plan(pcdhip_ctx* ctx, uint32_t out[2]);
/* Measurement aid: the v_mad_u64_u32 issue rate of the context's device in lane-operations per second (four waves per SIMD, eight independent
 * chains per lane, ~1 ms) -- the roof bench.py prices the integer kernels against (`roofline_int.peak_live`), measured on the box at hand. */
int pcdhip_mad_rate(pcdhip_ctx* ctx, double* out_lane_mads_per_s);
/* Sum of n Jacobian points (the multi-GPU combine step after the all-gather of partial results). */
int pcdhip_points_sum(pcdhip_ctx* ctx, int curve_id, int group_id, const uint64_t* xyz_mont, size_t n,
                      uint64_t* out_xyz_mont);
/* The same exchange without host round trips (bench.py / pcd_amd.dist at N > 1): pcdhip_msm_dev_partial leaves the Jacobian
 * result (X||Y||Z Montgomery limbs) in DEVICE memory owned by the caller -- the send buffer of the RCCL all-gather --
 * asynchronously on the context's stream (pcdhip_sync before another stream reads it); pcdhip_points_sum_dev sums n
 * Jacobian points that already sit in device memory (the receive buffer) and returns the sum to the host. */
int pcdhip_msm_dev_partial(pcdhip_ctx* ctx, const pcdhip_bases* bases, size_t offset, const pcdhip_buf* scalars,
                           size_t scalar_offset, size_t n, uint64_t* out_xyz_device);
int pcdhip_points_sum_dev(pcdhip_ctx* ctx, int curve_id, int group_id, const uint64_t* xyz_device, size_t n,
                          uint64_t* out_xyz_mont);
/* Jacobian -> affine (x||y, flag) for n points, as `into_affine()`. */
int pcdhip_to_affine(pcdhip_ctx* ctx, int curve_id, int group_id, const uint64_t* xyz_mont, size_t n,
                     uint64_t* out_xy_mont, uint8_t* out_inf);

/* ---- K2: radix-2 FFT --------------------------------------------------------------------------
 * Replaces ark-poly Radix2EvaluationDomain::{fft, ifft, coset_fft, coset_ifft}_in_place on a
 * domain of size 2^log_n: (inverse, coset) = (0,0) fft, (1,0) ifft, (0,1) coset_fft, (1,1) coset_ifft.
 * In place, natural order in and out, Montgomery form. */
int pcdhip_fft(pcdhip_ctx* ctx, int field_id, uint64_t* data_mont, uint32_t log_n, int inverse, int coset);
int pcdhip_fft_dev(pcdhip_ctx* ctx, pcdhip_buf* data, uint32_t log_n, int inverse, int coset);
/* K2m: ark-poly `GeneralEvaluationDomain` transforms on a domain of n = 2^a q^b elements (b <= 2; q = 7 for field
 * F298A, 5 for F753A: `MixedRadixEvaluationDomain`, what upstream selects for the HELP proof of a PCD step once its
 * circuit exceeds 2^17 / 2^15 rows); b = 0 is the radix-2 domain.  group_gen = GENERATOR^((p-1)/n). */
int pcdhip_fft_general(pcdhip_ctx* ctx, int field_id, uint64_t* data_mont, size_t n, int inverse, int coset);
/* n_ops transforms of ONE host vector, one after the other, with a single trip over PCIe (seam S2: ark-poly callers chain transforms on a
 * polynomial -- `ifft` then `coset_fft` in R1CSToQAP::witness_map and in Marlin's AHP rounds, reference path tests/mnt4_marlin.rs:141-204 --
 * and pcdhip_fft pays the 2 x n x 40 / 96 bytes per call).  ops[i]: bit 0 = inverse, bit 1 = coset; every transform over the domain of
 * n = 2^a q^b elements (GeneralEvaluationDomain).  Same results as n_ops calls of pcdhip_fft_general. */
int pcdhip_fft_seq(pcdhip_ctx* ctx, int field_id, uint64_t* data_mont, size_t n, const int* ops, int n_ops);
/* Size `GeneralEvaluationDomain::new(min_size)` would pick (0 if none exists): the witness map uses it. */
size_t pcdhip_domain_size(int field_id, size_t min_size);
/* Per-pass device time of the last transform; returns the number of passes written (<= 8). */
int pcdhip_fft_last_timings(pcdhip_ctx* ctx, float out_ms[8]);

/* ---- K1: Groth16 witness map -------------------------------------------------------------------
 * Replaces ark-groth16 `R1CSToQAP::witness_map` (libsnark reduction): h = (A z o B z - C z) / Z on
 * the domain `GeneralEvaluationDomain::new(num_constraints + num_inputs)` (radix-2, or mixed-radix for the help
 * fields beyond their 2-adicity; n = pcdhip_domain_size(...)); writes n elements of h. */
typedef struct {
  uint64_t num_rows;        /* = num_constraints */
  const uint64_t* row_ptr;  /* num_rows + 1 */
  const uint32_t* col;      /* nnz column (variable) indices */
  const uint64_t* coeff;    /* nnz coefficients, Montgomery limbs */
} pcdhip_csr;
int pcdhip_groth16_witness_map(pcdhip_ctx* ctx, int field_id, const pcdhip_csr* A, const pcdhip_csr* B,
                               const pcdhip_csr* C, const uint64_t* z_mont, size_t num_vars, size_t num_inputs,
                               uint64_t* h_out_mont);

/* ---- K1+K3+K4+K5: Groth16 prover arithmetic ----------------------------------------------------
 * Replaces the body of ark-groth16 `create_proof` after constraint synthesis (which stays in the
 * Rust host): witness map + 4 G1 MSMs + 1 G2 MSM + assembly with the caller's r, s. */
typedef struct {
  uint32_t curve_id, _pad;
  uint64_t num_vars;     /* m, including the leading 1 */
  uint64_t num_inputs;   /* including the leading 1 */
  uint64_t domain_size;  /* n */
  const uint64_t *alpha_g1, *beta_g1, *delta_g1;  /* vk.alpha_g1, pk.beta_g1, pk.delta_g1 */
  const uint64_t *beta_g2, *delta_g2;             /* vk.beta_g2, vk.delta_g2 */
  const uint64_t* a_query;    const uint8_t* a_inf;      /* m points */
  const uint64_t* b_g1_query; const uint8_t* b_g1_inf;   /* m */
  const uint64_t* b_g2_query; const uint8_t* b_g2_inf;   /* m */
  const uint64_t* h_query;    const uint8_t* h_inf;  uint64_t h_len;   /* n - 1 */
  const uint64_t* l_query;    const uint8_t* l_inf;  uint64_t l_len;   /* m - num_inputs */
} pcdhip_g16_pk_host;
int pcdhip_g16_pk_upload(pcdhip_ctx* ctx, const pcdhip_g16_pk_host* host, pcdhip_g16_pk** out);
void pcdhip_g16_pk_free(pcdhip_ctx* ctx, pcdhip_g16_pk* pk);
/* The MSM plan of a resident key's queries, in the order a, b_g1, b_g2, l, h: window bits and scalar windows (what a bench needs to count
 * the multiply-adds a proof executes; a key's large queries run one bit under the lone MSM's window, pcdhip_g16_pk_upload). */
int pcdhip_g16_pk_info(const pcdhip_g16_pk* pk, int window_bits[5], int windows[5]);
/* Device memory of a resident key's base vectors, in bytes: out[0] the five queries with their window-shifted copies (summed over the shards of a
 * multi-device key), out[1] the second layout for a shorter window (pcdhip_groth16_set_sparse_window; 0 when not built), out[2] / out[3] the
 * copies per point of the a query in the two layouts.  What a host that keeps several keys resident budgets with (T1). */
int pcdhip_g16_pk_memory(const pcdhip_g16_pk* pk, uint64_t out[4]);
/* Keep the circuit's constraint matrices (`ConstraintMatrices` from `cs.to_matrices()`, fixed per circuit like
 * the key) resident on the device; pcdhip_groth16_prove then accepts A = B = C = NULL and only the assignment
 * z crosses PCIe per proof. */
int pcdhip_g16_pk_set_r1cs(pcdhip_ctx* ctx, pcdhip_g16_pk* pk, const pcdhip_csr* A, const pcdhip_csr* B, const pcdhip_csr* C);
/* `R1CSToQAP::witness_map` alone over the matrices resident with `pk`, with nothing else on the device: h_out (nullable) =
 * the n coefficients of h as from pcdhip_groth16_witness_map; out_ms (nullable) = device milliseconds of [the three mat-vecs,
 * the seven transforms + the pointwise step, both].  pcdhip_groth16_prove overlaps the same work with four of its MSMs. */
int pcdhip_g16_witness_map_resident(pcdhip_ctx* ctx, const pcdhip_g16_pk* pk, const uint64_t* z_mont, uint64_t* h_out_mont,
                                    float out_ms[3]);
/* proof_out = A (G1 x||y) || B (G2 x||y) || C (G1 x||y), affine Montgomery; inf_out[3]. */
int pcdhip_groth16_prove(pcdhip_ctx* ctx, const pcdhip_g16_pk* pk, const pcdhip_csr* A, const pcdhip_csr* B,
                         const pcdhip_csr* C, const uint64_t* z_mont, const uint64_t* r_mont, const uint64_t* s_mont,
                         uint64_t* proof_out, uint8_t* inf_out);
/* Device time of the stages of the last prove, milliseconds:
 * [witness_map, msm_h, msm_l, msm_a, msm_b_g1, msm_b_g2, assembly, total]. */
int pcdhip_groth16_last_timings(pcdhip_ctx* ctx, float out_ms[8]);
/* How pcdhip_groth16_prove obtains the two variable-base products s*A and r*B_1 of upstream's assembly
 * (ark-groth16 create_proof, reached from src/ec_cycle_pcd/mod.rs:171,179); the proof is the same either way:
 *   2  one-lane windowed products queued behind the A / B_1 MSMs on high-priority streams, hidden under the longer
 *      MSMs (B in G2, h, l) of the same proof;
 *   1  folded into two more MSMs over the a / b_g1 bases with every scalar multiplied by s / r;
 *   0  (default) automatic: 2 for large proofs, 1 for small ones (<= 2^17 variables, 2^18 over the 753-bit fields). */
int pcdhip_groth16_set_assembly(pcdhip_ctx* ctx, int mode);
/* A window per PROOF for the four MSMs over the assignment (round 5).  The window of an MSM over a resident key is fixed by the key's
 * window-shifted copies and chosen for a dense scalar vector; the assignment of a verifier circuit (reference src/ec_cycle_pcd/data_structures.rs:269-304:
 * bit decompositions) is zeros and ones with a few per cent general scalars, and the bucket reduction of the dense window then costs more than the
 * additions.  OPT-IN (the second layout more than doubles a key's memory, upload and precompute time, and a host may keep several keys resident):
 * bits: 0 (default) never; -1 automatic: pcdhip_g16_pk_upload also lays the a / b_g1 / b_g2 / l queries out for a window four bits shorter when the key is a
 * whole key of at least 2^18 entries (2^14 over the 298-bit fields) on an ordinary context and the extra copies take at most a quarter of the device memory free at that moment
 * (9.2 GB for a 298-bit key of 2^20 entries, 54 GB for a 753-bit one); 6 .. 22: always, with that window.  Takes effect
 * at the next key upload.  pcdhip_groth16_prove counts the general scalars (neither 0 nor 1) of the assignment on the device and runs the four MSMs
 * on the shorter-window copies when they are at most an eighth of it (chained assembly only -- which a small 298-bit proof with such an assignment
 * then takes instead of the folded form its size would choose); the proof is the same either way.
 * pcdhip_groth16_last_plan: out[0] = 1 when the last proof took them, out[1] = its count of general scalars (0 when not counted). */
int pcdhip_groth16_set_sparse_window(pcdhip_ctx* ctx, int bits);
int pcdhip_groth16_last_plan(pcdhip_ctx* ctx, uint32_t out[2]);
/* Multi-device contexts with at least three devices: where the three independent chains of the witness map (a = A z, b = B z, c = C z:
 * mat-vec, ifft, coset fft each -- ark-groth16 `R1CSToQAP::witness_map`, SURVEY.md 8e "a || b || c on 3 GPUs") run.  1 (default): chain a on
 * device 0, b on device 1, c on device 2 (the matrices are resident on all three, pcdhip_g16_pk_set_r1cs), the two vectors come back
 * device to device, the pointwise step and the last transform run on device 0 -- about 45 % of the map's time leaves the critical path of
 * device 0, which also carries its share of the MSMs.  0: everything on device 0.  The proof is the same either way. */
int pcdhip_groth16_set_witness_split(pcdhip_ctx* ctx, int on);
/* Order of the stages of pcdhip_groth16_prove (what ark-groth16 `create_proof_with_reduction` does one after the other on rayon:
 * witness_map, then the five `VariableBaseMSM::multi_scalar_mul`; reference call sites src/ec_cycle_pcd/mod.rs:171,179).
 *   0 (default)  the four MSMs over the assignment first, each on its own stream, the witness map concurrently with them, the h MSM behind
 *      the map.  Measured in round 5 (profiles/r05_acc_probe_mnt4_298_2p20.txt, r05_pt_serial_lane8.txt): the proof takes the SUM of its kernels' standalone
 *      times to within a few per cent -- every kernel is bound by the same multiply-add issue slots, so no order can beat that sum;
 *   1  the witness map first with the device to itself, then all five MSMs at once (round 4's A/B knob; slower);
 *   2  the accumulate LANE (round 5): the map first while the MSMs sort beside it, then the MSMs' accumulate kernels one
 *      after the other on a stream confined by a CU mask to all but a few compute units (pcdhip_set_lane_reserve), each MSM's fix-up,
 *      bucket reduction and assembly products on a hardware queue of its own under the next accumulation.  A chain of one-wave launches
 *      no longer waits for accumulate workgroups to retire (0.98 ms instead of 20 .. 38 ms behind a device-filling grid,
 *      profiles/r05_k3_cu_mask.txt), but a masked queue runs the accumulate kernels 15 .. 45 % slower than an unmasked one (one CU less
 *      in one shader engine of every XCD unbalances the workgroup dispatch), and unmasked the lane is as fast as mode 0 at best:
 *      main proof MNT4-298 2^20: 15.7 (mode 0) / 18.0 (lane, 8 CUs reserved) / 16.4 .. 18.4 ms (lane, no mask); MNT4-753: 152 / 164 / 156.
 * The proof is the same in every mode.  Not while submitted MSMs are pending (PCDHIP_E_ARG). */
int pcdhip_groth16_set_schedule(pcdhip_ctx* ctx, int mode);
/* Compute units the accumulate lane leaves to the context's other streams (schedule 2; also pcdhip_msm_submit): a multiple of the number
 * of XCDs (8) keeps the same count free in every XCD.  -1 (default): the environment's PCDHIP_LANE_RESERVE, else 8; 0: no mask. */
int pcdhip_set_lane_reserve(pcdhip_ctx* ctx, int cus);

/* ---- the caller side of the path -- key generation -------------------------------------
 * Replaces ark-ec `FixedBaseMSM::{get_window_table, multi_scalar_mul}` + `batch_normalization_into_affine`:
 * out[i] = scalars[i] * base as affine points (x||y Montgomery, flag).  This is the primitive every query of
 * ark-groth16 `generate_parameters` is made of (reference call sites: circuit_specific_setup,
 * src/ec_cycle_pcd/mod.rs:69,78) and the KZG powers of `universal_setup` (mod.rs:346-354). */
int pcdhip_fixed_base_mul(pcdhip_ctx* ctx, int curve_id, int group_id, const uint64_t* base_xy_mont,
                          const uint64_t* scalars_canonical, size_t n, uint64_t* out_xy_mont, uint8_t* out_inf);
/* Replaces ark-groth16 `generate_parameters` (the body of `Groth16::circuit_specific_setup`, mod.rs:69,78) after
 * constraint synthesis, with the caller's randomness: toxic = [alpha, beta, gamma, delta, tau] (5 scalar-field
 * elements, Montgomery; tau must lie outside the evaluation domain, as upstream's
 * `sample_element_outside_domain` guarantees -- PCDHIP_E_ARG otherwise) and the two group generators.
 * QAP evaluation at tau (`instance_map_with_evaluation`: Lagrange coefficients, transposed mat-vecs) and every
 * fixed-base batch run on the device.  All output arrays are caller-allocated:
 * a / b_g1 / b_g2: num_vars points; gamma_abc_g1: num_inputs; l: num_vars - num_inputs;
 * h: n - 1 with n = pcdhip_domain_size(scalar field, num_constraints + num_inputs) (also returned in domain_size). */
typedef struct {
  uint64_t *alpha_g1, *beta_g1, *delta_g1;   /* one G1 point each */
  uint64_t *beta_g2, *gamma_g2, *delta_g2;   /* one G2 point each */
  uint64_t* a_query;      uint8_t* a_inf;
  uint64_t* b_g1_query;   uint8_t* b_g1_inf;
  uint64_t* b_g2_query;   uint8_t* b_g2_inf;
  uint64_t* h_query;      uint8_t* h_inf;
  uint64_t* l_query;      uint8_t* l_inf;
  uint64_t* gamma_abc_g1; uint8_t* gamma_abc_inf;
  uint64_t domain_size;   /* out */
} pcdhip_g16_setup_out;
int pcdhip_groth16_setup(pcdhip_ctx* ctx, int curve_id, const pcdhip_csr* A, const pcdhip_csr* B, const pcdhip_csr* C,
                         size_t num_vars, size_t num_inputs, const uint64_t* g1_xy_mont, const uint64_t* g2_xy_mont,
                         const uint64_t* toxic_mont, pcdhip_g16_setup_out* out);

/* ---- K6: pairing ---------------------------------------------------------------------------------
 * Replaces ark-ec `PairingEngine::product_of_pairings`: gt_out = final_exponentiation(prod_i miller_loop(P_i, Q_i)),
 * an element of Fq4 (MNT4) / Fq6 (MNT6) in tower order c0, c1 over Fq2 / Fq3, Montgomery limbs.  Up to 4096 pairs per call run ONE
 * WAVE PER PAIRING (the independent field products of every curve / tower step on sibling lanes, values in LDS: the latency of a
 * verification is what a merge node waits for); larger batches one lane per pair (64 pairings per wave: throughput). */
int pcdhip_multi_pairing(pcdhip_ctx* ctx, int curve_id, const uint64_t* g1_xy, const uint8_t* g1_inf, const uint64_t* g2_xy,
                         const uint8_t* g2_inf, size_t n_pairs, uint64_t* gt_out);
/* 0 (default): as described above; 1: always the lane-per-pair kernels (A/B measurements, parity tests of both forms).  Applies to
 * every entry point that computes pairings (multi_pairing, the Groth16 verifications, process_vk). */
int pcdhip_pairing_set_mode(pcdhip_ctx* ctx, int mode);
/* Replaces ark-groth16 `Groth16::verify` (reference call site mod.rs:239):
 * e(A,B) == e(alpha,beta) e(gamma_abc[0] + sum_i x_i gamma_abc[i], gamma) e(C,delta); public inputs canonical, without the leading 1. */
int pcdhip_groth16_verify(pcdhip_ctx* ctx, int curve_id, const uint64_t* alpha_g1, const uint64_t* beta_g2, const uint64_t* gamma_g2,
                          const uint64_t* delta_g2, const uint64_t* gamma_abc_g1, const uint8_t* gamma_abc_inf, size_t num_inputs,
                          const uint64_t* public_inputs_canonical, const uint64_t* proof, const uint8_t* proof_inf, int* ok);

/* ---- `process_vk` and the verifications that use it --------------------------------------------------
 * pcdhip_process_vk replaces ark-groth16 `prepare_verifying_key` (`SNARK::process_vk`, reference call sites
 * src/ec_cycle_pcd/mod.rs:71,371,445,495,553): e(alpha, beta) is computed ONCE and kept, gamma and delta are negated, gamma_abc_g1
 * becomes a resident base vector with one 8-bit window table per point (0.86 / 5.2 MB each at 298 / 753 bits, keys of up to 256 public
 * inputs; larger keys keep the plain vector): the input accumulation of a verification is then a few dozen table additions per proof
 * on one workgroup, not a chain of hundreds of doublings.  (The G2 line coefficients upstream also precomputes are not stored: the
 * Miller-loop kernels fuse the G2 steps with the line evaluations, and a coefficient stream would be three times their state.) */
typedef struct pcdhip_pvk pcdhip_pvk;
int pcdhip_process_vk(pcdhip_ctx* ctx, int curve_id, const uint64_t* alpha_g1, const uint64_t* beta_g2, const uint64_t* gamma_g2,
                      const uint64_t* delta_g2, const uint64_t* gamma_abc_g1, const uint8_t* gamma_abc_inf, size_t num_inputs, pcdhip_pvk** out);
void pcdhip_pvk_free(pcdhip_ctx* ctx, pcdhip_pvk* pvk);
/* `Groth16::verify_with_processed_vk` for n_proofs proofs under one key (`ECCyclePCD::verify`, mod.rs:239, once per prior message
 * of a merge node):  e(A, B) e(acc, -gamma) e(C, -delta) == e(alpha, beta)  -- three Miller loops and ONE final exponentiation per
 * proof (upstream's form; four and two without the prepared key), every Miller loop of the batch in one launch (3 n_proofs lanes),
 * the products and final exponentiations in another (one lane per proof); the input combinations acc_i = gamma_abc[0] + sum_j
 * x_ij gamma_abc[j] are n_proofs small MSMs over the resident gamma_abc vector.  Deterministic: ok[i] = 1 / 0 per proof.
 * public inputs: n_proofs x (num_inputs - 1) canonical scalars; proofs: n_proofs x (A || B || C); proofs_inf: n_proofs x 3 or NULL. */
int pcdhip_groth16_verify_prepared(pcdhip_ctx* ctx, const pcdhip_pvk* pvk, size_t n_proofs, const uint64_t* public_inputs_canonical,
                                   const uint64_t* proofs, const uint8_t* proofs_inf, int* ok);
/* The same n_proofs checks folded into ONE product with a SHARED final exponentiation, by a random linear combination with the
 * caller's challenges rho (n_proofs x 2 u64 limbs = 128 bits each, non-zero; the library draws no randomness):
 *   prod_i e(rho_i A_i, B_i) e(sum rho_i acc_i, -gamma) e(sum rho_i C_i, -delta) e(-(sum rho_i) alpha, beta) == 1
 * -- n_proofs + 3 Miller loops in one launch, one final exponentiation.  *all_ok = 1 iff the product is one: every proof valid,
 * or an invalid batch that slipped through with probability 2^-128 over rho.  On 0, pcdhip_groth16_verify_prepared finds the culprit.
 * A batch whose 3 n_proofs pairings all find a SIMD at once (wave-per-pairing mode, n_proofs <= 341) is answered by the n_proofs
 * deterministic checks side by side instead -- they take the time of one, less than the combination's scalings -- with the same
 * meaning of *all_ok (and no slip probability); rho is validated either way. */
int pcdhip_groth16_verify_batch_rlc(pcdhip_ctx* ctx, const pcdhip_pvk* pvk, size_t n_proofs, const uint64_t* public_inputs_canonical,
                                    const uint64_t* proofs, const uint8_t* proofs_inf, const uint64_t* rho, int* all_ok);

/* The native verifications of a merge node (`ECCyclePCD::verify`, mod.rs:239, once per prior
 * message) in one call = pcdhip_process_vk + pcdhip_groth16_verify_prepared + pcdhip_pvk_free.  Same verification key for all; public inputs:
 * n_proofs x (num_inputs - 1) canonical scalars; proofs: n_proofs x (A || B || C); proofs_inf: n_proofs x 3 flags or NULL.
 * ok[i] = 1 / 0 per proof -- deterministic, the same answers as n_proofs calls of pcdhip_groth16_verify. */
int pcdhip_groth16_verify_batch(pcdhip_ctx* ctx, int curve_id, const uint64_t* alpha_g1, const uint64_t* beta_g2,
                                const uint64_t* gamma_g2, const uint64_t* delta_g2, const uint64_t* gamma_abc_g1,
                                const uint8_t* gamma_abc_inf, size_t num_inputs, size_t n_proofs,
                                const uint64_t* public_inputs_canonical, const uint64_t* proofs, const uint8_t* proofs_inf, int* ok);

/* ---- wire format ---------------------------------------------------------------------------
 * Replaces ark-serialize `CanonicalSerialize::{serialize, serialize_uncompressed}` / `CanonicalDeserialize` for
 * `GroupAffine` (G1 and G2 of the four curves), ark-groth16 `Proof` (A || B || C) and `VerifyingKey` (alpha_g1, beta_g2, gamma_g2,
 * delta_g2, u64 length + gamma_abc_g1): what a proof or key looks like when it leaves the process that made it.  Host-side
 * (no GPU, no context).  Field elements: canonical little-endian integers in ceil(bits / 8) bytes (38 / 95), Fq2 / Fq3 as c0, c1
 * (, c2); flags in the top bits of the last byte: 0x80 = y is the larger of (y, -y), 0x40 = infinity; compressed = x with flags,
 * uncompressed = x, y with flags.  Points at this ABI are x || y Montgomery limbs + a flag byte, as everywhere in this header.
 * Reading checks that coordinates are reduced, that the point lies on the curve (compressed: rebuilds y by a square root) and --
 * like upstream's checked `deserialize` -- that a G2 point lies in the prime-order subgroup ([r]Q = O; G1 has cofactor 1):
 * PCDHIP_E_ARG otherwise.  The subgroup test is one scalar multiplication on the host per G2 point (~10 ms at 298 bits, ~0.1 s at
 * 753 bits); pcdhip_deserialize_points_unchecked (= `deserialize_unchecked`: curve check only) is for bulk data a trusted party
 * wrote, e.g. a proving key.  Proofs and verifying keys are always checked.  Sizes are 0 for invalid ids. */
size_t pcdhip_serialized_size(int curve_id, int group_id, int compressed);  /* bytes of one point */
int pcdhip_serialize_points(int curve_id, int group_id, const uint64_t* xy_mont, const uint8_t* inf, size_t n, int compressed, uint8_t* out);
int pcdhip_deserialize_points(int curve_id, int group_id, const uint8_t* in, size_t n, int compressed, uint64_t* xy_mont, uint8_t* inf);
int pcdhip_deserialize_points_unchecked(int curve_id, int group_id, const uint8_t* in, size_t n, int compressed, uint64_t* xy_mont,
                                        uint8_t* inf);
size_t pcdhip_proof_serialized_size(int curve_id, int compressed);
int pcdhip_proof_serialize(int curve_id, const uint64_t* proof, const uint8_t* proof_inf, int compressed, uint8_t* out);
int pcdhip_proof_deserialize(int curve_id, const uint8_t* in, int compressed, uint64_t* proof, uint8_t* proof_inf);
size_t pcdhip_vk_serialized_size(int curve_id, size_t num_inputs, int compressed);
int pcdhip_vk_serialize(int curve_id, const uint64_t* alpha_g1, const uint64_t* beta_g2, const uint64_t* gamma_g2, const uint64_t* delta_g2,
                        const uint64_t* gamma_abc_g1, const uint8_t* gamma_abc_inf, size_t num_inputs, int compressed, uint8_t* out);
/* gamma_abc arrays hold up to max_inputs points; the count found is returned in num_inputs */
int pcdhip_vk_deserialize(int curve_id, const uint8_t* in, size_t in_len, int compressed, uint64_t* alpha_g1, uint64_t* beta_g2, uint64_t* gamma_g2,
                          uint64_t* delta_g2, uint64_t* gamma_abc_g1, uint8_t* gamma_abc_inf, size_t max_inputs, size_t* num_inputs);

/* ---- timing helpers (HIP events on the context's stream, for bench.py) ------------------------- */
int pcdhip_timer_start(pcdhip_ctx* ctx);
int pcdhip_timer_stop(pcdhip_ctx* ctx, float* out_ms);

#ifdef __cplusplus
}
#endif
#endif /* PCDHIP_H */
