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
 *
 * Threads: every entry point may be called from any thread.  Calls that take the same zkc_ctx (directly or through a zkc_zkey /
 * zkc_msm made on it) are serialised by a mutex inside the context -- one GPU pipeline per context; different contexts (one per
 * GPU) run concurrently.  groth16_prover is re-entrant the way rapidsnark's is (callable from concurrent goroutines): its callers share the
 * process-wide proving service (zkc_service_default), which coalesces them into pipeline passes.  zkc_last_error(ctx) is the last error of that context (read it before
 * another thread's call on the same context fails); zkc_verify_last_error is per thread.
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
/* per-voter witness status.  The reference's witness calculator stops at the first assert it reaches and throws "Assert Failed." with circom's trace
 * of template lines (snarkjs witness_calculator.js exceptionHandler code 4); the values name those sites, and when a voter violates several the status
 * is the one the wasm reaches first: it runs census.circom top to bottom (:72 weight, :79-90 sikVerifier, :92-103 censusVerifier, :114 nullifier) and, in an
 * SMTVerifier, the SMTLevIns assert (smtverifier.circom:70 -> smtlevins.circom:93) before the root comparison (:134) -- i.e. 1, 7, 2, 5, 3, 4.
 * tests/golden/witness_vectors.json holds the wasm's message for every single, pair and triple of violations. */
enum {
    ZKC_W_OK = 0, ZKC_W_ERR_WEIGHT = 1 /* census.circom:72 */, ZKC_W_ERR_SIK_ROOT = 2 /* :90 via smtverifier.circom:134 */,
    ZKC_W_ERR_CENSUS_ROOT = 3 /* :103 via :134 */, ZKC_W_ERR_NULLIFIER = 4 /* :114 */, ZKC_W_ERR_LAST_SIBLING = 5 /* :103 via smtlevins.circom:93: censusSiblings[nLevels] != 0 */,
    ZKC_W_ERR_INPUT_RANGE = 6 /* a 32-byte input value >= r: this boundary's own check, made before any assert (snarkjs reduces decimal strings mod r first, and so do the hosts above it) */,
    ZKC_W_ERR_SIK_LAST_SIBLING = 7 /* :90 via smtlevins.circom:93: sikSiblings[nLevels] != 0 */
};
/* The Error.message snarkjs throws for that status: "Assert Failed.\nError in template <T> line: <n>\n..." (innermost template first).  At nLevels 160 the
 * template instance numbers of the committed dev/160 circuit.wasm are included (ForceEqualIfEnabled_159, SMTLevIns_80, SMTVerifier_160, ZkFranchiseProofCircuit_234)
 * and the text is byte-equal to the reference's; other depths would need the circuit compiled to know them and get the names without numbers.  NULL for 0 / unknown. */
const char* zkc_witness_status_text(int nLevels, int32_t status);

/* ---- context ----
 * One context per (process, GPU).  SURVEY.md 8b sketched zkc_ctx_create(device_ids[], n); the build runs one process per GPU instead
 * (bench.py / torch.distributed: independent proofs shard with no data-path collective), so a context names exactly one device; a host that
 * wants several GPUs in one process creates one context per device and calls them from separate threads (contexts do not share state). */
int  zkc_ctx_create(int hip_device, zkc_ctx** out);
void zkc_ctx_destroy(zkc_ctx* ctx);
const char* zkc_last_error(const zkc_ctx* ctx);        /* ctx may be NULL: last error of a failed zkc_ctx_create */
void* zkc_ctx_stream(zkc_ctx* ctx);                    /* the hipStream_t every kernel of this ctx is launched on */

/* ---- circuit shape: ZkFranchiseProofCircuit(nLevels), circuit/census.circom:49 ---- */
int zkc_circuit_n_inputs(int nLevels);                 /* 334 for nLevels = 160 */
int zkc_circuit_n_wires(int nLevels);                  /* 82754 for nLevels = 160 */
/* Circuit selection the way snarkjs callers name a circuit: by its witness-calculator wasm (groth16.fullProve(input, wasmFile, zkey),
 * ts_inputs/src/example.ts:358-362).  Hashes the image and returns nLevels of the native circuit it stands for -- 160 for
 * sha256 80a73567...c139 (artifacts/zkCensus/dev/circuits-info.md:7) -- or -1 for a wasm this build has no native witness generator for.
 * sha256_hex (may be NULL) receives the 64-digit hash + NUL either way. */
int zkc_circuit_nlevels_from_wasm(const void* wasm, size_t len, char sha256_hex[65]);
void zkc_sha256(const void* data, size_t len, uint8_t out[32]);

/* ---- a1: witness calculation (replaces wtns.calculate / CalculateWTNSBin) ----
 * inputs : B x n_inputs x 32 B, census.circom:51-67 declaration order:
 *          electionId[2], nullifier, availableWeight, voteHash[2], sikRoot, censusRoot, address, password, signature,
 *          voteWeight, censusSiblings[nLevels+1], sikSiblings[nLevels+1]
 * wtns   : B x n_wires x 32 B in the reference circuit.wasm's wire order (what .wtns section 2 holds)
 * status : B x int32 (ZKC_W_*).  Returns ZKC_ERR_WITNESS if any voter failed; the others are still valid. */
int zkc_witness(zkc_ctx* ctx, int nLevels, const void* inputs, int B, void* wtns, int32_t* status);
/* The inputs as the reference hands them over -- the TEXT of inputs_example.json (zk_census_test.go:85-89: prover.Prove's third argument is that file image;
 * internal/inputs.go:14-31 is its schema; ts_inputs/src/example.ts:358 passes the same object to groth16.fullProve) -> the flat block above.  Reads the object the way
 * circom_runtime 0.1.22's witness calculator does: the 12 names in any order, values as decimal strings, "0x" hex strings or integer literals of any length (with a sign),
 * nested arrays flattened, everything reduced mod r.  Host only.  Returns ZKC_OK; ZKC_ERR_FORMAT: not a JSON object (err = where); ZKC_ERR_GENERIC with circom_runtime's
 * message in err: "Signal <name> not found\n", "Too many values for input signal <name>\n", "Not enough values for input signal <name>\n", "Not all inputs have been set.
 * Only <k> out of <n>", "Cannot convert <text> to a BigInt".  One extension: sibling lists shorter than nLevels + 1 are padded with zeros (the generators pad,
 * internal/inputs.go:90-97; a caller that holds an arbo proof need not).  out: zkc_circuit_n_inputs(nLevels) x 32 B. */
int zkc_inputs_from_json(const char* json, size_t len, int nLevels, void* out, char* err, size_t errlen);
/* same with device-resident buffers (hipMalloc'ed or torch tensors), asynchronous on zkc_ctx_stream */
int zkc_witness_dev(zkc_ctx* ctx, int nLevels, const void* d_inputs, int B, void* d_wtns, int32_t* d_status /* B */);

/* ---- f2: proving key residency.  Parses a snarkjs-format Groth16 .zkey (the reference's proving_key.zkey format,
 * circuit/circuit-compiler.sh:112-131), builds the CSR of section 4 and uploads + pre-shifts the MSM bases. ---- */
int  zkc_zkey_load(zkc_ctx* ctx, const void* zkey_bytes, size_t len, zkc_zkey** out);
void zkc_zkey_free(zkc_zkey* zk);
int  zkc_zkey_info(const zkc_zkey* zk, uint32_t* nVars, uint32_t* nPublic, uint32_t* domainSize);
int  zkc_zkey_pass_info(const zkc_zkey* zk, int* pass_size, int* lanes);      /* proofs per pipeline pass (a batch call of B voters is cut into ceil(B / pass_size) equal passes) and pipeline lanes the passes rotate over */
int  zkc_zkey_header_info(const void* zkey_bytes, size_t len, uint32_t* nVars, uint32_t* nPublic, uint32_t* domainSize);   /* from the file image alone: host only, nothing is loaded */
int  zkc_zkey_sha256(const zkc_zkey* zk, uint8_t out[32]);      /* of the .zkey image it was loaded from (circuits-info.md:5 publishes this hash) */
/* cheap identity of a .zkey image for resident-key caches: a SAMPLED hash (SHA-256 over the header, the IC points, the ends of every section and a
 * 64-byte block of every 64 KB), not the SHA-256 of the image -- two images that differ only in unsampled bytes share it.  The proving service,
 * the N-API addon and groth16.py compare it on every call; the service also compares the full SHA-256 once per (image, resident key).  Host only. */
int  zkc_zkey_fingerprint(const void* zkey_bytes, size_t len, uint8_t out[32]);

/* ---- a2-a7: Groth16 prove (replaces snarkjs groth16.prove / rapidsnark groth16_prover internals).
 * wtns   : nWitness x 32 B standard form (the payload of .wtns section 2), host (zkc_prove) or device (zkc_prove_dev)
 * r, s   : the two blinding scalars, 32 B LE, < field order.  snarkjs/rapidsnark draw them at random; they are explicit
 *          here so that identical (zkey, wtns, r, s) gives identical bytes on every backend (SURVEY.md hard part 3).
 *          zkc_random_scalars fills n x 32 B with scalars uniform in [0, r) from the OS generator (rejection sampling).
 * proof  : A (64) | B (128) | C (64), affine, standard form.   public_out : nPublic x 32 B (may be NULL). */
void zkc_random_scalars(uint8_t* out, size_t n);
int zkc_prove(zkc_zkey* zk, const void* wtns, uint32_t nWitness, const uint8_t r[32], const uint8_t s[32],
              uint8_t proof[256], uint8_t* public_out);
int zkc_prove_dev(zkc_zkey* zk, const void* d_wtns, uint32_t nWitness, const uint8_t r[32], const uint8_t s[32],
                  uint8_t proof[256], uint8_t* public_out);

/* batch form: B witnesses resident in HBM (B x nWitness x 32 B), rs = B x 64 B (r || s per proof), outputs on the host:
 * proofs B x 256 B, publics B x nPublic x 32 B (may be NULL).  Up to ZKC_INFLIGHT (default 64 for a census key, 96 otherwise) proofs share one MSM
 * pipeline pass and the passes of a call rotate over ZKC_LANES pipeline lanes (default 4 for a census key; zkc_zkey_pass_info); the work space belongs to the
 * CONTEXT, is shared by its keys and grows with what calls put in flight (a single-proof caller touches one lane with room for four proofs, 0.6 GB at nLevels 160; a
 * 1 024-voter call four lanes of 64, 37 GB).  Mirrors what a rapidsnark / snarkjs caller would loop over (zk_census_test.go:89 per voter). */
int zkc_prove_batch_dev(zkc_zkey* zk, const void* d_wtns, uint32_t nWitness, int B, const uint8_t* rs, uint8_t* proofs, uint8_t* publics);

/* groth16.fullProve (ts_inputs/src/example.ts:358-362) for a batch, everything on the device: B input blocks (zkc_circuit_n_inputs x 32 B
 * each, census.circom:51-67 order) -> witnesses (left in d_wtns: B x nWires x 32 B), per-voter circuit status in d_status (ZKC_W_*; a voter
 * whose inputs fail a circuit assert still yields bytes in `proofs`, to be discarded by the caller) and the proofs.  The witness kernels
 * of one pass run underneath the MSMs of the previous one.  Needs a key of ZkFranchiseProofCircuit(nLevels) shape. */
int zkc_fullprove_batch_dev(zkc_zkey* zk, const void* d_inputs, int B, void* d_wtns, int32_t* d_status, const uint8_t* rs, uint8_t* proofs, uint8_t* publics);

/* The same batch calls in two halves, for a caller that keeps the GPU fed with batch after batch (bench.py does; the proving service below does it for per-voter callers):
 * begin validates, enqueues every pipeline pass and returns without waiting for the GPU; finish waits for that call and copies its proofs (B x 256 B) and public
 * signals (B x nPublic x 32 B, may be NULL) out.  slot = 0 or 1: two calls may be in flight on one key, and while call k drains (bucket reduction, blinding, copies)
 * the witness kernels and transforms of call k + 1 already run.  A slot must be finished before it is begun again; d_inputs / d_wtns / d_status belong to the call
 * until its finish returns (use two sets of buffers); rs is copied by begin.  d_inputs == NULL: the witnesses are given in d_wtns (zkc_prove_batch_dev's form). */
int zkc_batch_begin(zkc_zkey* zk, int slot, const void* d_inputs, int B, void* d_wtns, int32_t* d_status, const uint8_t* rs);
int zkc_batch_finish(zkc_zkey* zk, int slot, uint8_t* proofs, uint8_t* publics);

/* ---- e: several GPUs from ONE host process (SURVEY.md 8b's zkc_ctx_create(device_ids[], n), 8e's "one host thread + one HIP stream set per
 * device").  The reference's hosts are single processes (the Go loop over prover.Prove, zk_census_test.go:89; Node's groth16.fullProve,
 * ts_inputs/src/example.ts:358-362): a pool owns one context and one resident key per listed device and splits a batch into contiguous blocks
 * (device g of G proves voters [g B/G, (g+1) B/G), sizes differing by at most one), one host thread per device, no exchange between devices.
 * bench.py reaches the same split with one process per GPU over torch.distributed. ---- */
typedef struct zkc_pool zkc_pool;
int  zkc_pool_create(const int* hip_devices, int n, zkc_pool** out);
void zkc_pool_destroy(zkc_pool* pool);
int  zkc_pool_size(const zkc_pool* pool);
zkc_ctx*  zkc_pool_ctx(zkc_pool* pool, int i);           /* the i-th device's context / resident key (NULL before zkc_pool_zkey_load) */
zkc_zkey* zkc_pool_zkey(zkc_pool* pool, int i);
const char* zkc_pool_last_error(const zkc_pool* pool); /* pool may be NULL: last error of a failed zkc_pool_create */
int  zkc_pool_zkey_load(zkc_pool* pool, const void* zkey_bytes, size_t len);   /* every device or none */
/* groth16.fullProve for B voters, HOST buffers: inputs B x zkc_circuit_n_inputs x 32 B; rs B x 64 B or NULL (drawn uniform in Fr);
 * proofs B x 256 B; publics B x nPublic x 32 B or NULL; status B x int32 (ZKC_W_*) or NULL.  ZKC_ERR_WITNESS: every device finished and at
 * least one voter failed a circuit assert (the other proofs are valid). */
int  zkc_pool_fullprove_batch(zkc_pool* pool, const void* inputs, int B, const uint8_t* rs, uint8_t* proofs, uint8_t* publics, int32_t* status);

/* ---- the reference's own call shape, one voter per call, made fast: a submission queue behind the single-proof entry points.
 * prover.Prove is called per voter, from a loop or from goroutines (zk_census_test.go:89), groth16.fullProve per ballot
 * (ts_inputs/src/example.ts:358-362).  A service owns one worker thread per pipeline lane of a GPU ($ZKC_SERVICE_WORKERS, default 4; passes of up to $ZKC_SERVICE_PASS = 64
 * proofs); callers enqueue one voter each and the workers form pipeline passes out of whoever is waiting: a lone caller is served at once, concurrent callers share passes
 * (64 concurrent callers: 2 800-3 000 proofs/s on one MI355X, 256: 3 100-3 300 -- the rate of a 1 024-voter batch call -- instead of 64 x the single-proof latency).
 * Devices: hip_devices[n], or n = 0: $ZKC_DEVICE ("2", "0,1,2,3", "all"), unset = every visible device; a device is brought up (context, key
 * tables) only when the queue is long enough to pay for it, and loads its key before it takes requests.  A device keeps up to $ZKC_SERVICE_KEYS keys resident
 * (default 4, ~2.5 GB of tables each at nLevels 160, ONE 37 GB set of lane work space per device shared by them; least recently used out first): callers with different keys -- one per environment and depth,
 * circuit/circuit-compiler.sh:15,82 -- share a GPU without reloads, each batch of one key.  Key identity: the service keeps its own copy of
 * every .zkey image its devices may hold; a request finds its image through zkc_zkey_fingerprint (a SAMPLED hash) and, the first time a given caller
 * buffer (pointer, length) shows up, through the SHA-256 of the whole image -- so the caller's .zkey buffer need only stay valid during the call itself.
 * The blocking calls return the voter's own result: ZKC_OK, ZKC_ERR_WITNESS (status = ZKC_W_*: that voter failed a circuit assert; other
 * callers of the same pass are not affected), or an error with text in err.  rs = r || s (64 B) or NULL (drawn uniform in Fr).
 * The submit calls return at once; `done` runs on a service thread when the proof (or error) is in the caller's buffers, which -- like the
 * inputs / witness, but not the .zkey image -- must stay valid until then.  zkc_service_default(): the process-wide instance groth16_prover uses. */
typedef struct zkc_service zkc_service;
typedef void (*zkc_done_fn)(void* user, int rc, int32_t witness_status, const char* error_text /* valid during the call */);
int  zkc_service_create(const int* hip_devices, int n, zkc_service** out);
void zkc_service_destroy(zkc_service* svc);               /* waits for the batches in flight; queued requests fail with ZKC_ERR_GENERIC */
zkc_service* zkc_service_default(void);                   /* NULL when no GPU is visible (zkc_service_last_error) */
const char* zkc_service_last_error(void);                 /* per thread */
int zkc_service_fullprove(zkc_service* svc, const void* zkey, size_t zkey_len, int nLevels, const void* inputs /* n_inputs x 32 B */, const uint8_t* rs,
                          uint8_t proof[256], uint8_t* publics /* nPublic x 32 B or NULL */, int32_t* status, char* err, size_t errlen);
int zkc_service_prove(zkc_service* svc, const void* zkey, size_t zkey_len, const void* wtns /* nWitness x 32 B, host */, uint32_t nWitness, const uint8_t* rs,
                      uint8_t proof[256], uint8_t* publics, char* err, size_t errlen);
/* prover.Prove(zkey, wasm, inputs) with the reference's three byte slices (zk_census_test.go:81-89): the circuit is named by the witness-calculator image as the reference's
 * callers name it (its SHA-256 selects the native generator, zkc_circuit_nlevels_from_wasm; remembered per buffer, so the 3 MB hash is not taken per call) -- wasm NULL: by the
 * key's own shape -- and the inputs are the JSON text (zkc_inputs_from_json).  Otherwise zkc_service_fullprove. */
int zkc_service_fullprove_json(zkc_service* svc, const void* zkey, size_t zkey_len, const void* wasm, size_t wasm_len, const char* inputs_json, size_t inputs_len,
                               const uint8_t* rs, uint8_t proof[256], uint8_t* publics, int32_t* status, char* err, size_t errlen);
int zkc_service_submit_fullprove(zkc_service* svc, const void* zkey, size_t zkey_len, int nLevels, const void* inputs, const uint8_t* rs,
                                 uint8_t proof[256], uint8_t* publics, zkc_done_fn done, void* user);
int zkc_service_submit_prove(zkc_service* svc, const void* zkey, size_t zkey_len, const void* wtns, uint32_t nWitness, const uint8_t* rs,
                             uint8_t proof[256], uint8_t* publics, zkc_done_fn done, void* user);
/* out[0] requests accepted, [1] batches run, [2] largest batch, [3] key loads, [4] devices listed, [5] devices that ran a batch, [6] requests failed
 * wholesale (HIP / key errors), [7] requests waiting now */
int zkc_service_stats(zkc_service* svc, uint64_t out[8]);
/* where the workers' time went, microseconds summed over all batches: out[0] uploads before the GPU is free, [1] waiting for the GPU (the other worker's
 * batch), [2] key check / load + top-up upload, [3] the batch call itself, [4] handing results back; [5] proofs, [6] batches, [7] keys evicted (a device keeps
 * $ZKC_SERVICE_KEYS keys resident, default 4, least recently used out first; the reference has one key per environment and depth, circuit/circuit-compiler.sh:15,82) */
int zkc_service_timing(zkc_service* svc, uint64_t out[8]);
/* what the service holds in memory NOW (bytes): out[0] resident keys over all devices, [1] their constant tables (pre-shifted bases, matrices, twiddles, folding tables),
 * [2] the per-pass work space of the devices' pipeline lanes -- ONE set per device, shared by every key resident on it -- [3] the largest key's tables, [4] the largest device's
 * work space, [5] the workers' device staging (inputs, witnesses), [6] pinned host memory (staging + witness slots), [7] work-space reservations that failed so far (the lanes
 * then grow on demand).  INTEGRATION.md section 1 has the figures at nLevels 160. */
int zkc_service_memory(zkc_service* svc, uint64_t out[8]);

/* ---- the rapidsnark entry point (go-rapidsnark prover.h `groth16_prover`, reached from prover.Prove at
 * zk_census_test.go:89): whole .zkey and .wtns file images in, NUL-terminated proof / public-signal JSON out.
 * Returns 0 OK, 1 ERROR, 2 SHORT_BUFFER (required sizes written back; nothing is proved, so a size query is cheap),
 * 3 INVALID_WITNESS_LENGTH.  r, s are random, uniform in Fr.  Thread-safe / re-entrant the way rapidsnark's is, and concurrent callers
 * (goroutines) are coalesced: every call goes through zkc_service_default() above. */
int groth16_prover(const void* zkey_buffer, unsigned long zkey_size, const void* wtns_buffer, unsigned long wtns_size,
                   char* proof_buffer, unsigned long* proof_size, char* public_buffer, unsigned long* public_size,
                   char* error_msg, unsigned long error_msg_maxsize);

/* The same from the circuit INPUTS: what a cgo prover.Prove(zkey, wasm, inputs) calls in place of wasmer's witness calculator followed by groth16_prover
 * (zk_census_test.go:89).  Three file images in -- .zkey, circuit.wasm (names the circuit; may be NULL: by the key's shape), inputs JSON --
 * proof / public-signal JSON out, rapidsnark's return codes and buffer protocol.  A voter whose inputs fail a circuit assert: 1 with the wasm's own message
 * ("Assert Failed.\nError in template ...") in error_msg; a malformed inputs object: 1 with circom_runtime's message. */
int groth16_fullprove(const void* zkey_buffer, unsigned long zkey_size, const void* wasm_buffer, unsigned long wasm_size, const char* inputs_json, unsigned long inputs_size,
                      char* proof_buffer, unsigned long* proof_size, char* public_buffer, unsigned long* public_size,
                      char* error_msg, unsigned long error_msg_maxsize);

/* ---- a9: verification on the CPU (replaces proof.Verify(vkey) zk_census_test.go:122 / snarkjs groth16.verify).
 * zkc_verify takes the texts of verification_key.json, signals.json and proof.json: 1 valid, 0 invalid, <0 = -ZKC_ERR_*.
 * [r5] The three texts are parsed as JSON and must have the reference's shapes (objects / arrays of decimal strings, pi_a 3, pi_b 3 x 2, IC nPublic + 1; protocol
 * "groth16" and curve "bn128" where present): anything Go's json.Unmarshal (prover.ParseProof, zk_census_test.go:118) or JSON.parse refuses is -ZKC_ERR_FORMAT, never 1.
 * zkc_verify_last_error() is the text of the calling thread's LAST verify call (empty after a plain 0 or 1).
 * Never a positive value other than 1.  Proof points must be on their curves and B in the order-r subgroup of the twist (both
 * entry points, single and batch, apply the same membership checks); JSON points must have z = 1 (or 0 = infinity).
 * zkc_verify_bin takes vk = alpha1(64) beta2(128) gamma2(128) delta2(128) IC[nPublic+1](64 each), standard form.
 * [r5] About 1.4 ms per proof on the GPU boxes' hosts (csrc/zkc_pairing.h); the latest eight verification keys are kept ready by their bytes (their checks and
 * line coefficients are computed on first use, ~6 ms), any thread may call. */
int zkc_verify(const char* vkey_json, const char* public_json, const char* proof_json);
int zkc_verify_bin(const uint8_t* vk, int nPublic, const uint8_t* pub, const uint8_t* proof);

/* e(P, Q) as snarkjs / ffjavascript compute and print it (optimal ate, libff's final exponentiation): what verification_key.json carries
 * as vk_alphabeta_12 (artifacts/zkCensus/dev/160/verification_key.json:52).  g1 64 B, g2 128 B standard form; out = 12 x 32 B standard form
 * in the nesting order of that JSON member.  Host only. */
int zkc_pairing_bin(const uint8_t g1[64], const uint8_t g2[128], uint8_t out[384]);

/* ---- f4: batch verification of N proofs under one key (what a vote-counting node does after zk_census_test.go:103-124 per vote).
 * One random-linear-combination pairing check: N + 3 Miller loops and one final exponentiation; the G1 scalar multiplications run on
 * the GPU of `ctx`, and from 128 proofs on so do the Miller loops and the G2 membership tests (csrc/zkc_pairing_dev.hip: 8 192 proofs in 10 ms; below that, host
 * threads, sixteen pairs per shared accumulator).  vk as for zkc_verify_bin; pubs: N x nPublic x 32 B; proofs: N x 256 B (standard
 * form).  seed32: 32 bytes of FRESH randomness for the weights (NULL: taken from the OS); soundness error about 2^-128.
 * Returns 1 when every proof is valid, 0 when at least one is not (verify singly to find it), <0 = -ZKC_ERR_*.
 * Environment: ZKC_VERIFY_BATCH_GPU=0 / 1 keeps the Miller loops on host threads / sends them to the GPU whatever N; ZKC_VERIFY_TRACE=1 prints the time of each phase. */
int zkc_verify_batch(zkc_ctx* ctx, const uint8_t* vk, int nPublic, const uint8_t* pubs, const uint8_t* proofs, int N, const uint8_t* seed32);
const char* zkc_verify_last_error(void);

/* ---- a8: artifact codecs.  proof/public JSON exactly as snarkjs prints them (proof.json, signals.json);
 * .wtns = iden3 binfile "wtns" v2 (section 1: n8, prime, nWitness; section 2: nWitness x 32 B LE). */
int zkc_proof_to_json(const uint8_t proof[256], const uint8_t* pub, int nPublic, char* proof_buf, unsigned long* proof_size,
                      char* public_buf, unsigned long* public_size);
/* The way back (prover.ParseProof, zk_census_test.go:118, and the vkey []byte of (*Proof).Verify, :122): the JSON texts -> the binary forms zkc_verify_bin /
 * zkc_verify_batch take, under the same strict reading as zkc_verify.  zkc_proof_from_json: *nPublic in = room in pub (32 B each), out = signals in the document;
 * 1 = parsed, 0 = well-formed documents with a value that is no encoding (json.Unmarshal takes it, no verifier will), <0 = -ZKC_ERR_FORMAT (a document Unmarshal
 * refuses) / -ZKC_ERR_SHORT_BUFFER / -ZKC_ERR_BAD_ARG, text in zkc_verify_last_error().  zkc_vkey_from_json: *vk_size in = room, out = 448 + 64 (nPublic + 1). */
int zkc_proof_from_json(const char* proof_json, const char* public_json, uint8_t proof[256], uint8_t* pub, int* nPublic);
int zkc_vkey_from_json(const char* vkey_json, uint8_t* vk, unsigned long* vk_size, int* nPublic);
int zkc_wtns_parse(const void* wtns_bytes, unsigned long size, const uint8_t** payload, uint32_t* nWitness);
unsigned long zkc_wtns_write(const void* payload, uint32_t nWitness, void* out, unsigned long out_size);   /* returns bytes needed/written */

/* ---- the NTT and G1 MSM engines on their own (what ffjavascript's Fr.fft / Fr.ifft and G1.multiExpAffine are to snarkjs'
 * groth16.prove, ts_inputs/src/example.ts:358).  Used by SURVEY.md 8(d) config 5 (ii): 2^20-point synthetic stress (tools/stress.py).
 * zkc_ntt_dev        : nvec contiguous vectors of 2^logn Fr elements in Montgomery form (R = 2^256), natural order in and out,
 *                      d_src != d_dst; inverse != 0 includes the 1/n factor.
 * zkc_g1_mul_batch_dev: d_out[i] = k_i * base (scalars 32 B standard form on the device; points affine standard form, 64 B).
 * zkc_msm_g1_load_dev : n bases (device, affine standard form; checked to be on the curve) -> resident pre-shifted window tables.
 * zkc_msm_g1_dev      : sum_i s_i P_i, scalars n x 32 B standard form on the device; out = affine standard form (zero = infinity). */
typedef struct zkc_msm zkc_msm;
int zkc_ntt_dev(zkc_ctx* ctx, const void* d_src, void* d_dst, int logn, int nvec, int inverse);
int zkc_g1_mul_batch_dev(zkc_ctx* ctx, const uint8_t base_std[64], const void* d_scalars, uint32_t n, void* d_out);
int zkc_msm_g1_load_dev(zkc_ctx* ctx, const void* d_bases_std, uint32_t n, zkc_msm** out);
int zkc_msm_g1_dev(zkc_msm* m, const void* d_scalars, uint8_t out[64]);
void zkc_msm_g1_free(zkc_msm* m);

/* ---- test hooks (stage outputs for parity tests against the oracle; not part of the drop-in surface) ----
 * zkc_debug_stage: stage 0 -> A_w | B_w | C_w after buildABC (3 x domainSize x 32 B, Montgomery form);
 *                  stage 1 -> joinABC output (A'B' - C') on the odd coset (domainSize x 32 B, standard form).
 * zkc_msm_debug  : one MSM over zkey section which (0=A 1=B1 2=B2 3=C 4=H) with caller scalars (device, standard form);
 *                  host_out = affine point in standard form (64 B, or 128 B for B2). */
int zkc_debug_stage(zkc_zkey* zk, const void* d_wtns, int stage, void* host_out);
unsigned long long zkc_debug_early_retries(void);      /* calls of given witnesses that were laid out from their sibling wires, refused by the fold check and proved again from their fold flags (process-wide) */
int zkc_msm_debug(zkc_zkey* zk, int which, const void* d_scalars, uint32_t count, void* host_out);

/* ---- f1: the census / voter generator (internal/helpers.go:36-85 GenTree -- arbo.NewTree{Poseidon}, Add, GenProof, zero padding -- and internal/inputs.go:33-98
 * MockInputs; ts_inputs/src/inputs.ts:38-88).  arbo tree semantics: leaf = H(key, value, 1), node = H(left, right), path bit i = bit i (LSB first) of the key, an empty
 * subtree is 0, a subtree holding one leaf is that leaf's hash.  The trie is split on the host, every hash runs on the GPU (leaves in one launch, inner nodes one launch per
 * depth), sibling lists are written on the device: 8 192 voters in tens of milliseconds.  All values 32-byte little-endian, standard form, below r; host buffers.
 * zkc_poseidon_batch   : n_inputs in {2, 3, 4}; inputs B x n_inputs x 32 B, out B x 32 B.
 * zkc_smt_build        : one tree over n (key, value) pairs with distinct keys -> root (32 B), siblings (n x (nLevels + 1) x 32 B, zero-padded, may be NULL: leaf i's sibling
 *                        at level l in slot i (nLevels + 1) + l) and depths (levels above leaf i; may be NULL).  ZKC_ERR_BAD_ARG: duplicate keys, or two keys that share
 *                        their first nLevels path bits.
 * zkc_census_inputs    : the circuit inputs of n voters of one election: SIK = H(address, password, signature), nullifier = H(signature, password, electionId[0],
 *                        electionId[1]), census tree address -> available_weight, SIK tree address -> SIK, both roots and every voter's two sibling lists, as n blocks of
 *                        zkc_circuit_n_inputs(nLevels) x 32 B (the layout zkc_witness takes) into inputs_out (host) and / or d_inputs_out (device), either may be NULL.
 *                        vote_hash: n x 2 x 32 B; election_id: 2 x 32 B; roots_out (may be NULL): census root | SIK root. ---- */
int zkc_poseidon_batch(zkc_ctx* ctx, int n_inputs, const void* inputs, size_t B, void* out);
int zkc_smt_build(zkc_ctx* ctx, const void* keys, const void* values, size_t n, int nLevels, uint8_t root[32], void* siblings, int32_t* depths);
int zkc_census_inputs(zkc_ctx* ctx, size_t n, int nLevels, const uint8_t election_id[64], const void* address, const void* password, const void* signature,
                      const void* available_weight, const void* vote_weight, const void* vote_hash, void* inputs_out, void* d_inputs_out, uint8_t* roots_out);

/* ---- measurement: HIP-event timing per kernel category on zkc_ctx_stream (bit i of mask enables category i) ----
 * 0 witness, 1 buildABC mat-vec, 2 NTT+joinABC, 3 MSM digits+sort+offsets, 4 MSM bucket accumulation G1, 5 same G2,
 * 6 MSM heavy+reduce+final, 7 (no timing) bytes = (scalar, base) pairs x 96 B that entered the G1 MSMs after constant folding and
 * launches = group additions the G1 bucket accumulation actually performed (non-zero digits).  zkc_profile_read returns the summed duration, the number of bracketed launches and the
 * ALGORITHMIC bytes those launches processed. */
int zkc_profile_enable(zkc_ctx* ctx, uint32_t mask);
int zkc_profile_read(zkc_ctx* ctx, int category, double* total_ms, uint64_t* launches, uint64_t* alg_bytes);

/* ---- f3: TEST-ONLY trusted setup with known toxic waste (stand-in for circuit/circuit-compiler.sh:99-136, whose
 * output proving_key.zkey is a missing blob).  Reads an iden3 .r1cs, writes a snarkjs-format Groth16 .zkey and a
 * verification_key.json.  Host only; never use the result outside tests and benchmarks. */
int zkc_setup_from_r1cs(const char* r1cs_path, uint64_t seed, const char* zkey_path, const char* vkey_json_path,
                        char* err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif
