/* zkcensus.h -- C ABI of libzkcensus.so, the MI355X-native Groth16 prover for the Vocdoni zkCensus circuit.
 *
 * This is the drop-in boundary.  The reference reaches the hot path through
 *   TS : snarkjs  groth16.fullProve(inputs, wasmFile, zkeyFile)          ts_inputs/src/example.ts:358-362
 *   Go : prover.Prove(zkey, wasm, inputs) / proof.Verify(vkey)            zk_census_test.go:89,122
 *        -> go-rapidsnark (cgo) -> rapidsnark `groth16_prover(...)`
 * Every entry point below takes plain pointers and sizes; all field elements crossing the ABI are 32-byte
 * little-endian integers in STANDARD (non-Montgomery) form unless a comment says otherwise; points are affine
 * (G1 = x||y, G2 = x.c0||x.c1||y.c0||y.c1; all-zero = infinity).  No exceptions cross the ABI; functions return 0 on
 * success and a ZKC_ERR_* code otherwise, with text available from zkc_last_error().
 */
#ifndef ZKCENSUS_H
#define ZKCENSUS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct zkc_ctx zkc_ctx;     /* one per (process, GPU): HIP stream, Poseidon tables, witness template */
typedef struct zkc_zkey zkc_zkey;   /* a proving key made device-resident by zkc_zkey_load */

enum {
    ZKC_OK = 0,
    ZKC_ERR_GENERIC = 1,              /* rapidsnark PROVER_ERROR */
    ZKC_ERR_SHORT_BUFFER = 2,         /* rapidsnark PROVER_ERROR_SHORT_BUFFER: required sizes written back */
    ZKC_ERR_INVALID_WITNESS_LENGTH = 3,
    ZKC_ERR_BAD_ARG = 4,
    ZKC_ERR_FORMAT = 5,               /* malformed .zkey / .wtns / JSON */
    ZKC_ERR_HIP = 6,
    ZKC_ERR_WITNESS = 7               /* at least one voter failed a circuit assert: see per-voter status */
};
/* per-voter witness status (the reference wasm raises "Assert Failed" naming these template lines) */
enum {
    ZKC_W_OK = 0, ZKC_W_ERR_WEIGHT = 1 /* census.circom:72 */, ZKC_W_ERR_SIK_ROOT = 2 /* :90 */,
    ZKC_W_ERR_CENSUS_ROOT = 3 /* :103 */, ZKC_W_ERR_NULLIFIER = 4 /* :114 */, ZKC_W_ERR_LAST_SIBLING = 5 /* smtlevins */,
    ZKC_W_ERR_INPUT_RANGE = 6
};

/* ---- context ---- */
int  zkc_ctx_create(int hip_device, zkc_ctx** out);
void zkc_ctx_destroy(zkc_ctx* ctx);
const char* zkc_last_error(const zkc_ctx* ctx);        /* ctx may be NULL: last error of a failed zkc_ctx_create */
void* zkc_ctx_stream(zkc_ctx* ctx);                    /* the hipStream_t every kernel of this ctx is launched on */

/* ---- circuit shape: ZkFranchiseProofCircuit(nLevels), circuit/census.circom:49 ---- */
int zkc_circuit_n_inputs(int nLevels);                 /* 334 for nLevels = 160 */
int zkc_circuit_n_wires(int nLevels);                  /* 82754 for nLevels = 160 */

/* ---- a1: witness calculation (replaces wtns.calculate / CalculateWTNSBin) ----
 * inputs : B x n_inputs x 32 B, census.circom:51-67 declaration order:
 *          electionId[2], nullifier, availableWeight, voteHash[2], sikRoot, censusRoot, address, password, signature,
 *          voteWeight, censusSiblings[nLevels+1], sikSiblings[nLevels+1]
 * wtns   : B x n_wires x 32 B in the reference circuit.wasm's wire order (what .wtns section 2 holds)
 * status : B x int32 (ZKC_W_*).  Returns ZKC_ERR_WITNESS if any voter failed; the others are still valid. */
int zkc_witness(zkc_ctx* ctx, int nLevels, const void* inputs, int B, void* wtns, int32_t* status);
/* same with device-resident buffers (hipMalloc'ed or torch tensors), asynchronous on zkc_ctx_stream */
int zkc_witness_dev(zkc_ctx* ctx, int nLevels, const void* d_inputs, int B, void* d_wtns, int32_t* d_status /* B */);

/* ---- f3: TEST-ONLY trusted setup with known toxic waste (stand-in for circuit/circuit-compiler.sh:99-136, whose
 * output proving_key.zkey is a missing blob).  Reads an iden3 .r1cs, writes a snarkjs-format Groth16 .zkey and a
 * verification_key.json.  Host only; never use the result outside tests and benchmarks. */
int zkc_setup_from_r1cs(const char* r1cs_path, uint64_t seed, const char* zkey_path, const char* vkey_json_path,
                        char* err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif
