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
       ZKO_ERR_LAST_SIBLING = 5,              /* SMTLevIns: siblings[nLevels] must be 0 */
       ZKO_ERR_INPUT_RANGE = 6 };             /* an input value >= r */

void zko_poseidon(uint64_t out[4], const uint64_t *in /* n x 4 */, int n);   /* n = 2,3,4 ; standard form in/out */
int  zko_witness(int nLevels, const uint64_t *inputs, uint64_t *wires);      /* wires: zko_n_wires x 4 u64, standard form */

#endif
