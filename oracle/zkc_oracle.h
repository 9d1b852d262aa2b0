/* oracle/zkc_oracle.h -- TEST INFRASTRUCTURE ONLY (see bn254.h header note).
 * CPU restatement of the zkCensus hot path: witness (circuit/census.circom:49-115 + circomlib 2.0.5 templates),
 * Groth16 prove (snarkjs 0.7.0 groth16_prove algorithm) and verify. */
#ifndef ZKC_ORACLE_H
#define ZKC_ORACLE_H
#include "bn254.h"

#define ZKO_NPUB 8
/* flat input order = signal declaration order of census.circom:51-67:
 * electionId[2], nullifier, availableWeight, voteHash[2], sikRoot, censusRoot, address, password, signature,
 * voteWeight, censusSiblings[nLevels+1], sikSiblings[nLevels+1]  (334 values for nLevels=160), 4 x u64 LE, standard form */
static inline int zko_n_inputs(int nLevels) { return 12 + 2 * (nLevels + 1); }
int zko_n_wires(int nLevels);

/* witness error codes (the reference wasm raises exception code 4 naming the failing template/line) */
enum { ZKO_OK = 0, ZKO_ERR_WEIGHT = 1,        /* census.circom:72  checkWeight.out === 1 */
       ZKO_ERR_SIK_ROOT = 2,                  /* census.circom:79-90  sikVerifier root */
       ZKO_ERR_CENSUS_ROOT = 3,               /* census.circom:92-103 censusVerifier root */
       ZKO_ERR_NULLIFIER = 4,                 /* census.circom:111-114 */
       ZKO_ERR_LAST_SIBLING = 5,              /* SMTLevIns (smtlevins.circom:93 via smtverifier.circom:70): censusSiblings[nLevels] must be 0 */
       ZKO_ERR_INPUT_RANGE = 6,               /* an input value >= r */
       ZKO_ERR_SIK_LAST_SIBLING = 7 };        /* the same assert in sikVerifier: sikSiblings[nLevels] must be 0.  With several violations the status is the
                                                 assert the reference's wasm reaches first: 1, 7, 2, 5, 3, 4 in that order */

void zko_poseidon(uint64_t out[4], const uint64_t *in /* n x 4 */, int n);   /* n = 2,3,4 ; standard form in/out */
int  zko_witness(int nLevels, const uint64_t *inputs, uint64_t *wires);      /* wires: zko_n_wires x 4 u64, standard form */


/* ---- Groth16 (binary interfaces: every field element is 32 bytes little-endian, STANDARD form;
 *      G1 affine = x||y (64 B), G2 affine = x.c0||x.c1||y.c0||y.c1 (128 B); all-zero = infinity) ---- */
/* vk layout: alpha1(64) beta2(128) gamma2(128) delta2(128) IC[nPub+1](64 each) */
int  zko_groth16_verify(const uint8_t *vk, int nPub, const uint8_t *pub /* nPub x 32 */, const uint8_t *proof /* A64 B128 C64 */);
void zko_ntt(uint64_t *data /* n x 4, standard form, in place */, int logn, int inverse);
void zko_root_of_unity(uint64_t out[4], int logn);          /* 5^((r-1)/2^logn), ffjavascript's Fr.w[logn] */
void zko_msm_g1(uint8_t out[64], const uint8_t *bases /* n x 64 */, const uint8_t *scalars /* n x 32 */, size_t n);
void zko_msm_g2(uint8_t out[128], const uint8_t *bases /* n x 128 */, const uint8_t *scalars, size_t n);
void zko_g1_mul(uint8_t out[64], const uint8_t base[64], const uint8_t k[32]);
/* .zkey (snarkjs groth16 format) */
typedef struct {
    uint32_t nVars, nPublic, domainSize, nCoeffs;
    const uint8_t *alpha1, *beta1, *beta2, *gamma2, *delta1, *delta2;   /* Montgomery, as stored */
    const uint8_t *ic, *coeffs, *pointsA, *pointsB1, *pointsB2, *pointsC, *pointsH;
} zko_zkey_t;
int  zko_zkey_parse(const uint8_t *buf, size_t len, zko_zkey_t *z);       /* 0 ok */
/* stage outputs for parity tests */
int  zko_build_abc(const zko_zkey_t *z, const uint64_t *wtns, uint64_t *A, uint64_t *B, uint64_t *C); /* each domainSize x 4, standard form */
int  zko_h_evals(const zko_zkey_t *z, const uint64_t *wtns, uint64_t *P);  /* domainSize x 4: (A'B'-C') on the odd coset, standard form */
/* full prove: proof = A(64) B(128) C(64) standard-form affine; pub = nPublic x 32 */
int  zko_groth16_prove(const uint8_t *zkey, size_t len, const uint64_t *wtns, uint32_t nWtns, const uint8_t r[32], const uint8_t s[32],
                       uint8_t proof[256], uint8_t *pub);
int  zko_zkey_vk(const uint8_t *zkey, size_t len, uint8_t *vk_out /* 448 + 64*(nPub+1) */);

#endif
