/* The C-ABI used directly, no Python: set up an SRS from a trapdoor, upload 2^n lookups, prove the range-check
 * Lasso argument on the GPU, verify on the host with the pairing check, then tamper with the proof.
 *   gcc -O2 -Iinclude examples/lasso_c_abi.c -Lhalo2-lasso_amd -llasso_hip -Wl,-rpath,$PWD/halo2-lasso_amd -o /tmp/lasso_c_abi
 * This is what a host written in the reference's language would do through its FFI. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lasso_hip.h"

#define CHECK(call)                                                                  \
  do {                                                                               \
    lh_status rc_ = (call);                                                          \
    if (rc_ != LH_OK) {                                                              \
      fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, lh_last_error());               \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

static uint64_t rng_state = 0x4C4153534F00ull;
static uint64_t next_u64(void) { /* splitmix64 */
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? (size_t)atoi(argv[1]) : 16, l = 16, c = 2;
  const size_t nv = n > l ? n : l, N = (size_t)1 << n;
  lh_ctx* ctx = NULL;
  CHECK(lh_ctx_create(0, &ctx));

  /* trapdoor: any nonzero field elements; a 62-bit value in the low limb is a valid Montgomery representative */
  lh_fr* ss = calloc(nv, sizeof(lh_fr));
  for (size_t i = 0; i < nv; i++) ((uint64_t*)&ss[i])[0] = (next_u64() >> 2) | 1;
  lh_srs* srs = NULL;
  lh_mkzg_vp* vp = NULL;
  CHECK(lh_mkzg_setup(ctx, ss, nv, &srs));
  CHECK(lh_mkzg_vp_setup(ss, nv, &vp));

  /* range-check table: value = dim0 + 2^16 dim1, both limbs looked up in the identity subtable */
  lh_lasso_table table;
  memset(&table, 0, sizeof table);
  table.num_chunks = (uint32_t)c, table.chunk_bits = (uint32_t)l, table.num_memories = 2, table.num_terms = 2;
  lh_fr one_mont = {{0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full}};
  lh_fr coeff[2];
  uint64_t c1[2] = {1, 1ull << 16};
  for (int j = 0; j < 2; j++) {
    table.memory_chunk[j] = (uint32_t)j, table.memory_subtable[j] = LH_SUBTABLE_IDENTITY;
    table.g_num_factors[j] = 1, table.g_factor[j][0] = (uint8_t)j;
  }
  /* Fr::from(u64) on the device gives the Montgomery form of the two coefficients */
  uint64_t* d_u64 = NULL;
  lh_fr* d_fr = NULL;
  CHECK(lh_alloc(ctx, sizeof c1, (void**)&d_u64));
  CHECK(lh_alloc(ctx, sizeof coeff, (void**)&d_fr));
  CHECK(lh_upload(ctx, d_u64, c1, sizeof c1));
  CHECK(lh_fr_from_u64(ctx, d_u64, 2, d_fr));
  CHECK(lh_download(ctx, coeff, d_fr, sizeof coeff));
  if (memcmp(&coeff[0], &one_mont, sizeof(lh_fr)) != 0) {
    fprintf(stderr, "unexpected Montgomery form of 1\n");
    return 1;
  }
  table.g_coeff[0] = coeff[0], table.g_coeff[1] = coeff[1];

  uint32_t* h_dim = malloc(N * sizeof(uint32_t));
  const uint32_t* d_dims[2];
  for (size_t j = 0; j < c; j++) {
    for (size_t i = 0; i < N; i++) h_dim[i] = (uint32_t)(next_u64() & 0xffff);
    void* d = NULL;
    CHECK(lh_alloc(ctx, N * sizeof(uint32_t), &d));
    CHECK(lh_upload(ctx, d, h_dim, N * sizeof(uint32_t)));
    d_dims[j] = d;
  }

  lh_transcript* t = NULL;
  CHECK(lh_keccak_transcript_new(&t));
  CHECK(lh_lasso_prove(ctx, srs, &table, n, d_dims, t));
  const uint8_t* proof = NULL;
  size_t proof_len = 0;
  CHECK(lh_keccak_transcript_proof(t, &proof, &proof_len));
  printf("2^%zu range-check lookups proved: %zu proof bytes\n", n, proof_len);

  lh_transcript* r = NULL;
  CHECK(lh_keccak_transcript_from_proof(proof, proof_len, &r));
  CHECK(lh_lasso_verify(vp, &table, n, r));
  size_t left = 1;
  CHECK(lh_keccak_transcript_remaining(r, &left));
  printf("verified on the host, %zu bytes left unread\n", left);
  lh_keccak_transcript_free(r);

  uint8_t* bad = malloc(proof_len);
  memcpy(bad, proof, proof_len);
  bad[proof_len / 2] ^= 1;
  CHECK(lh_keccak_transcript_from_proof(bad, proof_len, &r));
  lh_status rc = lh_lasso_verify(vp, &table, n, r);
  printf("tampered proof -> status %d (%s)\n", rc, lh_last_error());
  lh_keccak_transcript_free(r);

  lh_keccak_transcript_free(t);
  lh_mkzg_vp_free(vp);
  lh_srs_free(ctx, srs);
  lh_ctx_destroy(ctx);
  return (left == 0 && rc != LH_OK) ? 0 : 1;
}
