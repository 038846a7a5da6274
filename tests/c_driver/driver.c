/* A plain-C caller of include/pcdhip.h, driving pcdhip_groth16_prove exactly as the Rust shim (rust/src/prover.rs) does: points
 * repacked as x || y limbs plus infinity bytes, the assignment as Montgomery limbs, r then s, constraint matrices made resident
 * first -- no Python, no ctypes between the caller and the library.  Input: a blob written by tests/test_c_driver.py
 * (u64 header, then the arrays in the order read below); exit code 0 iff the proof equals the expected bytes and a second
 * entry point (pcdhip_msm over the key's a_query) equals its expected point.  Test infrastructure. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pcdhip.h"

static unsigned char* blob;
static size_t pos;
static const void* take(size_t bytes) { const void* p = blob + pos; pos += (bytes + 7) / 8 * 8; return p; }

/* Seam S2 (rust/src/s2.rs, the hooks of the ark-ec / ark-poly forks): the call sequence a Marlin prover makes through
 * `VariableBaseMSM::multi_scalar_mul` and `Radix2EvaluationDomain::{fft, ifft}_in_place` (/root/reference tests/mnt4_marlin.rs:72-75) --
 * ONE upload of the committer key, MSMs over PREFIXES of it with host scalars (`powers_of_g[..deg + 1]`), transforms IN PLACE on the
 * caller's own vector -- each result compared with the oracle's.  Blob: header {curve, n_points, n_msms, log_n, field_id, 0, 0, 0},
 * points xy, inf, then per MSM {len (u64), scalars, expected affine xy}, then the vector, its fft and the ifft of the vector. */
static int run_s2(long len) {
  const uint64_t* h = (const uint64_t*)take(8 * 8);
  const int curve = (int)h[0], field = (int)h[4];
  const size_t n = (size_t)h[1], k = (size_t)h[2];
  const uint32_t log_n = (uint32_t)h[3];
  const size_t L = (size_t)pcdhip_field_limbs(pcdhip_curve_scalar_field(curve)), l1 = (size_t)pcdhip_point_limbs(curve, PCDHIP_G1);
  const uint64_t* xy = (const uint64_t*)take(n * l1 * 8);
  const uint8_t* inf = (const uint8_t*)take(n);
  pcdhip_ctx* ctx = NULL;
  int rc = pcdhip_init(0, &ctx);
  if (rc) { fprintf(stderr, "pcdhip_init: %s\n", pcdhip_strerror(rc)); return 3; }
  pcdhip_bases* key = NULL;
  rc = pcdhip_bases_upload(ctx, curve, PCDHIP_G1, xy, inf, n, &key);   /* once per committer key */
  if (rc) { fprintf(stderr, "upload: %s\n", pcdhip_strerror(rc)); return 4; }
  int bad = 0;
  uint64_t* xyz = (uint64_t*)calloc(l1 / 2 * 3, 8);
  uint64_t* aff = (uint64_t*)calloc(l1, 8);
  for (size_t i = 0; i < k; i++) {
    const size_t m = (size_t)*(const uint64_t*)take(8);
    const uint64_t* sc = (const uint64_t*)take(m * L * 8);
    const uint64_t* want = (const uint64_t*)take(l1 * 8);
    uint8_t pinf = 0;
    rc = pcdhip_msm(ctx, key, 0, sc, m, xyz);                           /* a prefix of the resident key, host scalars */
    if (!rc) rc = pcdhip_to_affine(ctx, curve, PCDHIP_G1, xyz, 1, aff, &pinf);
    if (rc) { fprintf(stderr, "msm %zu: %s\n", i, pcdhip_strerror(rc)); return 5; }
    if (memcmp(aff, want, l1 * 8) != 0 || pinf != 0) { fprintf(stderr, "prefix MSM %zu (len %zu) differs\n", i, m); bad = 1; }
  }
  const size_t nn = (size_t)1 << log_n, lf = (size_t)pcdhip_field_limbs(field);
  const uint64_t* v = (const uint64_t*)take(nn * lf * 8);
  const uint64_t* want_f = (const uint64_t*)take(nn * lf * 8);
  const uint64_t* want_i = (const uint64_t*)take(nn * lf * 8);
  if (pos > (size_t)len) { fprintf(stderr, "blob too short\n"); return 2; }
  uint64_t* w = (uint64_t*)malloc(nn * lf * 8);
  memcpy(w, v, nn * lf * 8);
  rc = pcdhip_fft(ctx, field, w, log_n, 0, 0);                          /* fft_in_place on the vector's own storage */
  if (rc || memcmp(w, want_f, nn * lf * 8) != 0) { fprintf(stderr, "fft differs (%s)\n", pcdhip_strerror(rc)); bad = 1; }
  rc = pcdhip_fft(ctx, field, w, log_n, 1, 0);                          /* ifft_in_place brings the vector back */
  if (rc || memcmp(w, v, nn * lf * 8) != 0) { fprintf(stderr, "ifft(fft(v)) differs (%s)\n", pcdhip_strerror(rc)); bad = 1; }
  memcpy(w, v, nn * lf * 8);
  rc = pcdhip_fft(ctx, field, w, log_n, 1, 0);
  if (rc || memcmp(w, want_i, nn * lf * 8) != 0) { fprintf(stderr, "ifft differs (%s)\n", pcdhip_strerror(rc)); bad = 1; }
  /* the chain form (rust/src/s2.rs fft_chain): fft then ifft in ONE call, one trip over PCIe -- the vector comes back unchanged; fft alone
     through the same entry point equals pcdhip_fft's answer */
  { const int ops2[2] = {0, 1}, ops1[1] = {0};
    memcpy(w, v, nn * lf * 8);
    rc = pcdhip_fft_seq(ctx, field, w, nn, ops2, 2);
    if (rc || memcmp(w, v, nn * lf * 8) != 0) { fprintf(stderr, "fft_seq(fft, ifft) differs (%s)\n", pcdhip_strerror(rc)); bad = 1; }
    rc = pcdhip_fft_seq(ctx, field, w, nn, ops1, 1);
    if (rc || memcmp(w, want_f, nn * lf * 8) != 0) { fprintf(stderr, "fft_seq(fft) differs (%s)\n", pcdhip_strerror(rc)); bad = 1; }
    bad |= pcdhip_fft_seq(ctx, field, w, nn, ops1, 0) != PCDHIP_E_ARG; }
  /* what the hook answers "not mine" for must come back as a code, never an abort: a transform beyond what the library builds */
  { const int e = pcdhip_fft(ctx, field, w, 31, 0, 0); bad |= !(e == PCDHIP_E_SIZE_UNSUPPORTED || e == PCDHIP_E_ARG); }
  pcdhip_bases_free(ctx, key);
  pcdhip_destroy(ctx);
  printf(bad ? "MISMATCH\n" : "c driver ok: S2 sequence (one upload, prefix MSMs, in-place transforms) equals the oracle\n");
  return bad ? 1 : 0;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: driver <blob> [s2]\n"); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("blob"); return 2; }
  fseek(f, 0, SEEK_END);
  long len = ftell(f);
  fseek(f, 0, SEEK_SET);
  blob = (unsigned char*)malloc((size_t)len);
  if (!blob || fread(blob, 1, (size_t)len, f) != (size_t)len) { fprintf(stderr, "short read\n"); return 2; }
  fclose(f);
  if (argc >= 3 && strcmp(argv[2], "s2") == 0) return run_s2(len);
  const uint64_t* h = (const uint64_t*)take(12 * 8);
  const uint64_t curve = h[0], m = h[1], ni = h[2], dom = h[3], h_len = h[4], l_len = h[5], nc = h[6], nnz_a = h[7], nnz_b = h[8], nnz_c = h[9];
  const size_t L = (size_t)pcdhip_field_limbs(pcdhip_curve_scalar_field((int)curve));
  const size_t l1 = (size_t)pcdhip_point_limbs((int)curve, PCDHIP_G1), l2 = (size_t)pcdhip_point_limbs((int)curve, PCDHIP_G2);

  pcdhip_g16_pk_host k;
  memset(&k, 0, sizeof k);
  k.curve_id = (uint32_t)curve; k.num_vars = m; k.num_inputs = ni; k.domain_size = dom; k.h_len = h_len; k.l_len = l_len;
  k.alpha_g1 = (const uint64_t*)take(l1 * 8); k.beta_g1 = (const uint64_t*)take(l1 * 8); k.delta_g1 = (const uint64_t*)take(l1 * 8);
  k.beta_g2 = (const uint64_t*)take(l2 * 8); k.delta_g2 = (const uint64_t*)take(l2 * 8);
  k.a_query = (const uint64_t*)take(m * l1 * 8); k.a_inf = (const uint8_t*)take(m);
  k.b_g1_query = (const uint64_t*)take(m * l1 * 8); k.b_g1_inf = (const uint8_t*)take(m);
  k.b_g2_query = (const uint64_t*)take(m * l2 * 8); k.b_g2_inf = (const uint8_t*)take(m);
  k.h_query = (const uint64_t*)take(h_len * l1 * 8); k.h_inf = (const uint8_t*)take(h_len);
  k.l_query = (const uint64_t*)take(l_len * l1 * 8); k.l_inf = (const uint8_t*)take(l_len);
  pcdhip_csr A, B, C;
  const uint64_t nnz[3] = {nnz_a, nnz_b, nnz_c};
  pcdhip_csr* ms[3] = {&A, &B, &C};
  for (int i = 0; i < 3; i++) {
    ms[i]->num_rows = nc;
    ms[i]->row_ptr = (const uint64_t*)take((nc + 1) * 8);
    ms[i]->col = (const uint32_t*)take(nnz[i] * 4);
    ms[i]->coeff = (const uint64_t*)take(nnz[i] * L * 8);
  }
  const uint64_t* z = (const uint64_t*)take(m * L * 8);
  const uint64_t* r = (const uint64_t*)take(L * 8);
  const uint64_t* s = (const uint64_t*)take(L * 8);
  const size_t pw = 2 * l1 + l2;
  const uint64_t* want = (const uint64_t*)take(pw * 8);
  const uint8_t* want_inf = (const uint8_t*)take(3);
  const uint64_t* msm_scalars = (const uint64_t*)take(m * L * 8);
  const uint64_t* msm_want_xy = (const uint64_t*)take(l1 * 8);
  if (pos > (size_t)len) { fprintf(stderr, "blob too short\n"); return 2; }

  pcdhip_ctx* ctx = NULL;
  int rc = pcdhip_init(0, &ctx);
  if (rc) { fprintf(stderr, "pcdhip_init: %s\n", pcdhip_strerror(rc)); return 3; }
  pcdhip_g16_pk* pk = NULL;
  rc = pcdhip_g16_pk_upload(ctx, &k, &pk);
  if (!rc) rc = pcdhip_g16_pk_set_r1cs(ctx, pk, &A, &B, &C);
  uint64_t* proof = (uint64_t*)calloc(pw, 8);
  uint8_t inf[3] = {0, 0, 0};
  if (!rc) rc = pcdhip_groth16_prove(ctx, pk, NULL, NULL, NULL, z, r, s, proof, inf);
  if (rc) { fprintf(stderr, "prove: %s (%s)\n", pcdhip_strerror(rc), pcdhip_last_hip_error(ctx)); return 4; }
  int bad = memcmp(proof, want, pw * 8) != 0 || memcmp(inf, want_inf, 3) != 0;
  /* the same proof from a key that also carries its assignment queries for a shorter window (pcdhip_groth16_set_sparse_window): the prover
   * counts the general scalars of z and picks -- the bytes do not depend on which copies it took */
  {
    pcdhip_g16_pk* pk2 = NULL;
    uint32_t plan[2] = {7, 7};
    uint64_t* proof2 = (uint64_t*)calloc(pw, 8);
    uint8_t inf2[3] = {0, 0, 0};
    rc = pcdhip_groth16_set_sparse_window(ctx, 8);
    if (!rc) rc = pcdhip_g16_pk_upload(ctx, &k, &pk2);
    if (!rc) rc = pcdhip_g16_pk_set_r1cs(ctx, pk2, &A, &B, &C);
    if (!rc) rc = pcdhip_groth16_set_assembly(ctx, 2);
    if (!rc) rc = pcdhip_groth16_prove(ctx, pk2, NULL, NULL, NULL, z, r, s, proof2, inf2);
    if (!rc) rc = pcdhip_groth16_last_plan(ctx, plan);
    if (rc) { fprintf(stderr, "prove (sparse-window key): %s (%s)\n", pcdhip_strerror(rc), pcdhip_last_hip_error(ctx)); return 6; }
    bad |= memcmp(proof2, want, pw * 8) != 0 || memcmp(inf2, want_inf, 3) != 0 || plan[0] > 1 || plan[1] > m;
    {  /* what the two keys hold (pcdhip_g16_pk_memory): the second key carries the second layout, the first does not (opt-in);
        * and the budget getter hands back what was set (a layer that changes it for one upload restores the host's setting) */
      uint64_t mem1[4] = {0, 0, 0, 0}, mem2[4] = {0, 0, 0, 0};
      size_t budget = 1;
      bad |= pcdhip_g16_pk_memory(pk, mem1) != PCDHIP_OK || pcdhip_g16_pk_memory(pk2, mem2) != PCDHIP_OK || pcdhip_g16_pk_memory(NULL, mem1) != PCDHIP_E_ARG;
      bad |= mem1[0] == 0 || mem1[1] != 0 || mem1[3] != 0 || mem2[0] == 0 || ((mem2[1] != 0) != (mem2[3] >= 2)) || (plan[0] == 1 && mem2[1] == 0);
      bad |= pcdhip_set_precompute_budget(ctx, (size_t)3 << 30) != PCDHIP_OK || pcdhip_get_precompute_budget(ctx, &budget) != PCDHIP_OK || budget != (size_t)3 << 30;
      bad |= pcdhip_set_precompute_budget(ctx, 0) != PCDHIP_OK || pcdhip_get_precompute_budget(ctx, NULL) != PCDHIP_E_ARG;
    }
    bad |= pcdhip_groth16_set_sparse_window(ctx, 3) != PCDHIP_E_ARG || pcdhip_groth16_last_plan(ctx, NULL) != PCDHIP_E_ARG;
    (void)pcdhip_groth16_set_sparse_window(ctx, 0);
    (void)pcdhip_groth16_set_assembly(ctx, 0);
    pcdhip_g16_pk_free(ctx, pk2);
    free(proof2);
  }
  /* the same key material through the MSM entry point */
  pcdhip_bases* bases = NULL;
  uint64_t* xyz = (uint64_t*)calloc(l1 / 2 * 3, 8);
  uint64_t* xy = (uint64_t*)calloc(l1, 8);
  uint8_t pinf = 0;
  rc = pcdhip_bases_upload(ctx, (int)curve, PCDHIP_G1, k.a_query, k.a_inf, m, &bases);
  if (!rc) rc = pcdhip_msm(ctx, bases, 0, msm_scalars, m, xyz);
  if (!rc) rc = pcdhip_to_affine(ctx, (int)curve, PCDHIP_G1, xyz, 1, xy, &pinf);
  if (rc) { fprintf(stderr, "msm: %s\n", pcdhip_strerror(rc)); return 5; }
  bad |= memcmp(xy, msm_want_xy, l1 * 8) != 0 || pinf != 0;
  /* error behaviour from C: bad arguments are codes, never aborts */
  bad |= pcdhip_msm(ctx, bases, 1, msm_scalars, m, xyz) != PCDHIP_E_ARG;
  bad |= pcdhip_groth16_prove(ctx, pk, &A, NULL, NULL, z, r, s, proof, inf) != PCDHIP_E_ARG;
  pcdhip_bases_free(ctx, bases);
  pcdhip_g16_pk_free(ctx, pk);
  pcdhip_destroy(ctx);
  printf(bad ? "MISMATCH\n" : "c driver ok: proof and MSM equal the expected bytes\n");
  return bad ? 1 : 0;
}
