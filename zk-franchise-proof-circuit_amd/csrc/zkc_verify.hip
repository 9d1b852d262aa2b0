// zkc_verify.hip -- a9: Groth16 verification on the CPU (constant work, no GPU), plus the JSON / .wtns codecs (a8) and the
// rapidsnark-shaped `groth16_prover` entry point (product host code).
//
// Mirrors go-rapidsnark/verifier VerifyGroth16 behind dvote's proof.Verify (zk_census_test.go:122) and snarkjs
// groth16.verify: vk_x = IC0 + sum s_i IC_{i+1};  e(-A, B) e(alpha, beta) e(vk_x, gamma) e(C, delta) == 1.
// The pairing lives in zkc_pairing.h (optimal ate, projective sparse lines, shared accumulator, prepared G2 points, cyclotomic final exponentiation).
#include <cstdio>
#include <ctime>
#include <cstring>
#include <string>
#include <vector>
#include <random>
#include <thread>
#include <algorithm>
#include "zkc_prover.h"
#include "zkc_hostparse.h"
#include "zkc_pairing.h"
#include <sys/random.h>
#include <cerrno>
#include <mutex>
#include <memory>
#include <array>

using namespace zkc;
using namespace zkc::pairing;
namespace zkc {      // zkc_pairing_dev.hip
int miller_membership_begin(zkc_ctx* ctx, const G2Affine* h_Q, uint32_t N);
void miller_join(zkc_ctx* ctx);
int miller_product_dev(zkc_ctx* ctx, const G1XYZZ* d_P, uint32_t N, pairing::Fq12* product, int* bad);
}

namespace {

// e(P, Q) as snarkjs stores it in verification_key.json as vk_alphabeta_12 (artifacts/zkCensus/dev/160/verification_key.json:52): ffjavascript / wasmcurves follow
// libff's alt_bn128, whose final exponentiation ends in the Fuentes-Castaneda chunk and so yields the reduced pairing raised to 2x(6x^2 + 3x + 1) -- which is what
// pairing::final_exp returns.  Pinned by tests/test_oracle_pinning.py against the reference's own verification key (alpha, beta -> vk_alphabeta_12).
Fq12 pairing_snarkjs(const G1Affine& P, const G2Affine& Q) { return final_exp(miller(P, Q)); }
bool rd_fq_std(Fq& o, const uint8_t* p) { uint32_t s[8]; memcpy(s, p, 32); if (!fp_std_lt_p<FqParams>(s)) return false; o = fp_from_std<FqParams>(s); return true; }
bool rd_g1_std(G1Affine& o, const uint8_t* p) { return rd_fq_std(o.x, p) && rd_fq_std(o.y, p + 32); }
bool rd_g2_std(G2Affine& o, const uint8_t* p) { return rd_fq_std(o.x.c0, p) && rd_fq_std(o.x.c1, p + 32) && rd_fq_std(o.y.c0, p + 64) && rd_fq_std(o.y.c1, p + 96); }

using parse::dec_of;

thread_local std::string g_err;
int vfail(int code, const std::string& m) { g_err = m; return code; }

}  // namespace

extern "C" const char* zkc_verify_last_error(void) { return g_err.c_str(); }
// e(P, Q) as snarkjs / ffjavascript compute and print it: 12 x 32 B standard form, order c0.a0.(c0,c1) c0.a1 c0.a2 c1.a0 c1.a1 c1.a2
// (the nesting of vk_alphabeta_12 in verification_key.json).  Host only.
extern "C" int zkc_pairing_bin(const uint8_t g1[64], const uint8_t g2[128], uint8_t out[384]) {
    G1Affine P; G2Affine Q;
    if (!g1 || !g2 || !out || !rd_g1_std(P, g1) || !rd_g2_std(Q, g2) || !g1_on_curve(P) || !g2_on_curve(Q)) return ZKC_ERR_BAD_ARG;
    const Fq12 e = pairing_snarkjs(P, Q);
    const Fq2* c[6] = {&e.a.a0, &e.a.a1, &e.a.a2, &e.b.a0, &e.b.a1, &e.b.a2};
    for (int i = 0; i < 6; i++) { uint32_t t[8]; fp_to_std<FqParams>(t, c[i]->c0); memcpy(out + 64 * i, t, 32); fp_to_std<FqParams>(t, c[i]->c1); memcpy(out + 64 * i + 32, t, 32); }
    return ZKC_OK;
}
extern "C" void zkc_sha256(const void* data, size_t len, uint8_t out[32]) { parse::sha256(data, len, out); }
extern "C" int zkc_zkey_sha256(const zkc_zkey* zk, uint8_t out[32]) { if (!zk || !out) return ZKC_ERR_BAD_ARG; memcpy(out, zk->sha256, 32); return ZKC_OK; }
extern "C" int zkc_zkey_fingerprint(const void* zkey_bytes, size_t len, uint8_t out[32]) {
    parse::BinSections bs; std::string perr;
    if (!zkey_bytes || !out || !parse::binfile_sections((const uint8_t*)zkey_bytes, len, "zkey", 1, bs, perr)) return ZKC_ERR_FORMAT;
    parse::zkey_fingerprint((const uint8_t*)zkey_bytes, len, bs, out);
    return ZKC_OK;
}
// The message of the Error snarkjs throws when the witness calculator hits an assert (witness_calculator.js exceptionHandler: "Assert Failed.\n" + the lines
// circom's runtime printed, innermost template first).  Pinned against the wasm's own output in tests/golden/witness_vectors.json.
extern "C" const char* zkc_witness_status_text(int nLevels, int32_t status) {
    struct Site { int32_t st; const char* frames[3][2]; };                        // {template, line}, innermost first
    static const Site sites[] = {
        {ZKC_W_ERR_WEIGHT,           {{"ZkFranchiseProofCircuit", "72"}}},
        {ZKC_W_ERR_SIK_LAST_SIBLING, {{"SMTLevIns", "93"}, {"SMTVerifier", "70"}, {"ZkFranchiseProofCircuit", "90"}}},
        {ZKC_W_ERR_SIK_ROOT,         {{"ForceEqualIfEnabled", "56"}, {"SMTVerifier", "134"}, {"ZkFranchiseProofCircuit", "90"}}},
        {ZKC_W_ERR_LAST_SIBLING,     {{"SMTLevIns", "93"}, {"SMTVerifier", "70"}, {"ZkFranchiseProofCircuit", "103"}}},
        {ZKC_W_ERR_CENSUS_ROOT,      {{"ForceEqualIfEnabled", "56"}, {"SMTVerifier", "134"}, {"ZkFranchiseProofCircuit", "103"}}},
        {ZKC_W_ERR_NULLIFIER,        {{"ForceEqualIfEnabled", "56"}, {"ZkFranchiseProofCircuit", "114"}}},
    };
    static const struct { const char* tmpl; const char* id; } ids160[] = {{"ForceEqualIfEnabled", "_159"}, {"SMTLevIns", "_80"}, {"SMTVerifier", "_160"}, {"ZkFranchiseProofCircuit", "_234"}};
    static const std::vector<std::string> text = [] {
        std::vector<std::string> t(2 * 8);
        for (int with_ids = 0; with_ids < 2; with_ids++) {
            for (const Site& s : sites) {
                std::string m = "Assert Failed.\n";
                for (const auto& f : s.frames) {
                    if (!f[0]) break;
                    m += std::string("Error in template ") + f[0];
                    if (with_ids) for (const auto& k : ids160) if (!strcmp(k.tmpl, f[0])) m += k.id;
                    m += std::string(" line: ") + f[1] + "\n";
                }
                t[with_ids * 8 + s.st] = m;
            }
            t[with_ids * 8 + ZKC_W_ERR_INPUT_RANGE] = "Input value is not below the field order r\n";
        }
        return t;
    }();
    if (status <= 0 || status >= 8) return nullptr;
    return text[(nLevels == 160 ? 8 : 0) + status].c_str();
}
// circuits this build has a native witness generator for, by the sha256 of their circom witness-calculator wasm
// the shape of a key straight from the file image (no GPU, no load): what a host needs to size its buffers before it submits a request
extern "C" int zkc_zkey_header_info(const void* zkey_bytes, size_t len, uint32_t* nVars, uint32_t* nPublic, uint32_t* domainSize) {
    parse::BinSections bs; parse::ZkeyHeader zh; std::string perr;
    if (!zkey_bytes || !parse::binfile_sections((const uint8_t*)zkey_bytes, len, "zkey", 1, bs, perr) || !parse::zkey_check(bs, zh, perr, false)) { g_err = perr; return ZKC_ERR_FORMAT; }
    if (nVars) *nVars = zh.nVars; if (nPublic) *nPublic = zh.nPub; if (domainSize) *domainSize = zh.n;
    return ZKC_OK;
}
extern "C" int zkc_circuit_nlevels_from_wasm(const void* wasm, size_t len, char sha256_hex[65]) {
    static const struct { const char* sha; int nLevels; } known[] = {
        {"80a73567f6a4655d4332301efcff4bc5711bb48176d1c71fdb1e48df222ac139", 160},      // artifacts/zkCensus/dev/circuits-info.md:7
    };
    if (!wasm) return -1;
    uint8_t d[32]; parse::sha256(wasm, len, d); const std::string hex = parse::hex_of(d, 32);
    if (sha256_hex) memcpy(sha256_hex, hex.c_str(), 65);
    for (const auto& k : known) if (hex == k.sha) return k.nLevels;
    return -1;
}

// the circuit a caller names by its witness-calculator image, per call and cheaply: the SHA-256 of a 3 MB wasm is ~2 ms, a proof's share of a GPU pass 0.3 ms, so the answer
// is remembered per (buffer, length) and re-checked against a sampled digest of the buffer (both ends and a 64-byte block of every 64 KB: a buffer re-used for another file
// is hashed again).  -1: unknown wasm
static int nlevels_of_wasm_cached(const void* wasm, size_t len) {
    struct Seen { const void* p; size_t len; uint8_t sample[32]; int nLevels; };
    static std::mutex mu; static std::vector<Seen> seen;
    uint8_t smp[32];
    { parse::Sha256 h; const uint8_t* b = (const uint8_t*)wasm; const uint64_t l64 = len; h.update(&l64, 8); const size_t edge = std::min<size_t>(len, 1024); h.update(b, edge); h.update(b + len - edge, edge);
      for (size_t off = 0; off + 64 <= len; off += 65536) h.update(b + off, 64); h.final(smp); }
    { std::lock_guard<std::mutex> g(mu); for (auto& e : seen) if (e.p == wasm && e.len == len && !memcmp(e.sample, smp, 32)) return e.nLevels; }
    const int nl = zkc_circuit_nlevels_from_wasm(wasm, len, nullptr);
    std::lock_guard<std::mutex> g(mu);
    if (seen.size() >= 16) seen.erase(seen.begin());
    Seen e; e.p = wasm; e.len = len; memcpy(e.sample, smp, 32); e.nLevels = nl; seen.push_back(e);
    return nl;
}
// nLevels of the call: from the wasm's hash when one is given (the reference's callers name the circuit that way: prover.Prove(zkey, wasm, inputs), zk_census_test.go:89),
// else from the key's own shape (wire count and 8 public signals).  < 0: err says why
static int nlevels_of_call(const void* zkey, size_t zkey_len, const void* wasm, size_t wasm_len, std::string& err) {
    if (wasm) {
        const int nl = nlevels_of_wasm_cached(wasm, wasm_len);
        if (nl < 0) err = "the witness calculator (wasm) is not one this build has a native circuit for; the C ABI has no wasm runtime (the N-API surface executes unknown circuits in Node): compute the witness elsewhere and call groth16_prover";
        return nl;
    }
    parse::BinSections bs; parse::ZkeyHeader zh;
    if (!parse::binfile_sections((const uint8_t*)zkey, zkey_len, "zkey", 1, bs, err) || !parse::zkey_check(bs, zh, err, false)) return -1;
    if (zh.nPub == 8) for (int nl = 3; nl <= 253; nl++) if ((uint32_t)zkc_circuit_n_wires(nl) == zh.nVars) return nl;
    err = "no wasm given and the key is not a ZkFranchiseProofCircuit key"; return -1;
}
// inputs_example.json's text -> the flat input block (zkc_hostparse.h circuit_inputs_from_json: circom_runtime's reading and messages)
extern "C" int zkc_inputs_from_json(const char* json, size_t len, int nLevels, void* out, char* err, size_t errlen) {
    auto fail = [&](int code, const std::string& m) { if (err && errlen) snprintf(err, errlen, "%s", m.c_str()); return code; };
    if (!json || !out || nLevels < 3 || nLevels > 253) return fail(ZKC_ERR_BAD_ARG, "zkc_inputs_from_json: bad argument");
    std::string why;
    const int rc = parse::circuit_inputs_from_json(json, len, nLevels, (uint8_t*)out, why);
    if (rc) return fail(rc == 1 ? ZKC_ERR_FORMAT : ZKC_ERR_GENERIC, why);
    if (err && errlen) err[0] = 0;
    return ZKC_OK;
}
// prover.Prove(zkey, wasm, inputs) with the reference's three byte slices (zk_census_test.go:81-89), one voter, through the proving service: witness and proof on the GPU
extern "C" int zkc_service_fullprove_json(zkc_service* svc, const void* zkey, size_t zkey_len, const void* wasm, size_t wasm_len, const char* inputs_json, size_t inputs_len,
                                          const uint8_t* rs, uint8_t proof[256], uint8_t* publics, int32_t* status, char* err, size_t errlen) {
    auto fail = [&](int code, const std::string& m) { if (err && errlen) snprintf(err, errlen, "%s", m.c_str()); return code; };
    if (status) *status = 0;
    if (!svc || !zkey || !inputs_json || !proof) return fail(ZKC_ERR_BAD_ARG, "zkc_service_fullprove_json: bad argument");
    std::string why;
    const int nl = nlevels_of_call(zkey, zkey_len, wasm, wasm_len, why);
    if (nl < 0) return fail(ZKC_ERR_BAD_ARG, why);
    std::vector<uint8_t> flat(32 * (size_t)zkc_circuit_n_inputs(nl));
    const int rc = parse::circuit_inputs_from_json(inputs_json, inputs_len, nl, flat.data(), why);
    if (rc) return fail(rc == 1 ? ZKC_ERR_FORMAT : ZKC_ERR_GENERIC, why);
    return zkc_service_fullprove(svc, zkey, zkey_len, nl, flat.data(), rs, proof, publics, status, err, errlen);
}
// ... and with rapidsnark's conventions for everything else (groth16_prover below): JSON texts out, 0 / 1 / 2, the process-wide service, random (r, s).  What a cgo
// prover.Prove calls instead of wasmer + groth16_prover.  A voter whose inputs fail a circuit assert: 1, error_msg = the wasm's message.
extern "C" int groth16_fullprove(const void* zkey_buffer, unsigned long zkey_size, const void* wasm_buffer, unsigned long wasm_size, const char* inputs_json, unsigned long inputs_size,
                                 char* proof_buffer, unsigned long* proof_size, char* public_buffer, unsigned long* public_size, char* error_msg, unsigned long error_msg_maxsize) {
    auto err = [&](int code, const std::string& m) { if (error_msg && error_msg_maxsize) snprintf(error_msg, error_msg_maxsize, "%s", m.c_str()); return code; };
    if (!zkey_buffer || !inputs_json || !proof_size || !public_size) return err(ZKC_ERR_GENERIC, "groth16_fullprove: null argument");
    parse::BinSections bs; parse::ZkeyHeader zh; std::string perr;
    if (!parse::binfile_sections((const uint8_t*)zkey_buffer, zkey_size, "zkey", 1, bs, perr) || !parse::zkey_check(bs, zh, perr, false)) return err(ZKC_ERR_GENERIC, perr);
    const unsigned long need_proof = 8 * 80 + 128, need_public = (unsigned long)zh.nPub * 80 + 8;
    if (!proof_buffer || !public_buffer || *proof_size < need_proof || *public_size < need_public) {
        *proof_size = need_proof; *public_size = need_public;
        return err(ZKC_ERR_SHORT_BUFFER, "Proof or public signals buffer is too short");
    }
    zkc_service* svc = zkc_service_default();
    if (!svc) return err(ZKC_ERR_GENERIC, zkc_service_last_error());
    uint8_t proof[256]; std::vector<uint8_t> pub(32 * (size_t)zh.nPub + 1); char etext[512] = {0}; int32_t st = 0;
    int rc = zkc_service_fullprove_json(svc, zkey_buffer, zkey_size, wasm_buffer, wasm_size, inputs_json, inputs_size, nullptr, proof, pub.data(), &st, etext, sizeof etext);
    if (rc) return err(ZKC_ERR_GENERIC, etext);
    rc = zkc_proof_to_json(proof, pub.data(), (int)zh.nPub, proof_buffer, proof_size, public_buffer, public_size);
    if (rc == ZKC_ERR_SHORT_BUFFER) return err(ZKC_ERR_SHORT_BUFFER, "Proof or public signals buffer is too short");
    return rc;
}

// ---- a verification key made ready once: points read and checked (on the curve, G2 points in the order-r subgroup), gamma and delta prepared into their line
// coefficients, the Miller value of (alpha, beta) computed.  A node verifies every ballot of an election under ONE key (zk_census_test.go:103-124 per vote), so the
// latest few keys are kept by their bytes; the three [r]Q checks and three preparations were a third of a verification. ----
namespace {
struct VkReady {
    std::vector<uint8_t> bytes; int nPublic = 0;
    G1Affine alpha; G2Affine beta, gamma, delta; std::vector<G1Affine> ic;
    G2Prepared pgamma, pdelta, pbeta; Fq12 m_alpha_beta;
    std::vector<G1Affine> ic_mult;                 // k IC_j for k = 1..15, j = 1..nPublic (row j - 1): the public-input combination takes one addition per 4 bits of a signal
};
std::mutex g_vk_mu; std::vector<std::shared_ptr<const VkReady>> g_vk_ready;      // most recent first, at most 8
std::shared_ptr<const VkReady> vk_ready(const uint8_t* vk, int nPublic, int* code) {
    const size_t len = 448 + 64 * ((size_t)nPublic + 1);
    {
        std::lock_guard<std::mutex> g(g_vk_mu);
        for (size_t i = 0; i < g_vk_ready.size(); i++)
            if (g_vk_ready[i]->bytes.size() == len && !memcmp(g_vk_ready[i]->bytes.data(), vk, len)) {
                auto hit = g_vk_ready[i];
                if (i) { g_vk_ready.erase(g_vk_ready.begin() + (long)i); g_vk_ready.insert(g_vk_ready.begin(), hit); }
                return hit;
            }
    }
    auto r = std::make_shared<VkReady>(); r->bytes.assign(vk, vk + len); r->nPublic = nPublic; r->ic.resize((size_t)nPublic + 1);
    if (!rd_g1_std(r->alpha, vk) || !rd_g2_std(r->beta, vk + 64) || !rd_g2_std(r->gamma, vk + 192) || !rd_g2_std(r->delta, vk + 320)) { *code = vfail(-ZKC_ERR_FORMAT, "verification key coordinate >= q"); return nullptr; }
    if (!g1_on_curve(r->alpha) || !g2_in_subgroup(r->beta) || !g2_in_subgroup(r->gamma) || !g2_in_subgroup(r->delta)) { *code = vfail(-ZKC_ERR_FORMAT, "verification key point not on the curve / not in the order-r subgroup"); return nullptr; }
    for (int j = 0; j <= nPublic; j++) if (!rd_g1_std(r->ic[j], vk + 448 + 64 * (size_t)j) || !g1_on_curve(r->ic[j])) { *code = vfail(-ZKC_ERR_FORMAT, "IC point invalid"); return nullptr; }
    r->pgamma = prepare_g2(r->gamma); r->pdelta = prepare_g2(r->delta); r->pbeta = prepare_g2(r->beta);
    const Pair ab{r->alpha, &r->pbeta}; r->m_alpha_beta = multi_miller(&ab, 1);
    if (nPublic <= 64) {
        r->ic_mult.resize(15 * (size_t)nPublic);
        for (int j = 0; j < nPublic; j++) {
            G1XYZZ m = G1XYZZ::from_affine(r->ic[j + 1]);
            for (int k = 0; k < 15; k++) { r->ic_mult[15 * (size_t)j + k] = xyzz_to_affine_gcd(m); m = xyzz_add_affine(m, r->ic[j + 1]); }
        }
    }
    std::lock_guard<std::mutex> g(g_vk_mu);
    g_vk_ready.insert(g_vk_ready.begin(), r); if (g_vk_ready.size() > 8) g_vk_ready.pop_back();
    return r;
}
// sum_j k_j P_j over a handful of points (the public-input combination vk_x): one doubling chain shared by all scalars, mixed additions
G1XYZZ g1_sum_of_products(const G1Affine* pts, const uint32_t (*k)[8], int n) {
    int top = -1;
    for (int b = 255; b >= 0 && top < 0; b--) for (int j = 0; j < n; j++) if ((k[j][b >> 5] >> (b & 31)) & 1) { top = b; break; }
    G1XYZZ acc = G1XYZZ::inf();
    for (int b = top; b >= 0; b--) {
        acc = xyzz_dbl(acc);
        for (int j = 0; j < n; j++) if ((k[j][b >> 5] >> (b & 31)) & 1) acc = xyzz_add_affine(acc, pts[j]);
    }
    return acc;
}
}  // namespace

// vk: alpha1(64) beta2(128) gamma2(128) delta2(128) IC[nPublic+1](64 each); pub: nPublic x 32; proof: A(64) B(128) C(64); standard form
extern "C" int zkc_verify_bin(const uint8_t* vk, int nPublic, const uint8_t* pub, const uint8_t* proof) {
    g_err.clear();
    if (!vk || !pub || !proof || nPublic < 0 || nPublic > 4096) return vfail(-ZKC_ERR_BAD_ARG, "zkc_verify_bin: bad argument");
    int code = 0;
    const std::shared_ptr<const VkReady> V = vk_ready(vk, nPublic, &code);
    if (!V) return code;
    G1Affine A, C; G2Affine B;
    if (!rd_g1_std(A, proof) || !rd_g2_std(B, proof + 64) || !rd_g1_std(C, proof + 192)) return 0;
    // B must lie in the order-r subgroup of the twist (G2 has a cofactor): go-rapidsnark's bn256 unmarshalling and this library's batch verifier
    // reject such points too, so the two entry points agree on crafted proofs
    if (!g1_on_curve(A) || !g1_on_curve(C) || !g2_in_subgroup(B)) return 0;
    std::vector<std::array<uint32_t, 8>> k((size_t)nPublic);
    for (int i = 0; i < nPublic; i++) {
        memcpy(k[i].data(), pub + 32 * i, 32);
        if (!fp_std_lt_p<FrParams>(k[i].data())) return 0;                         // snarkjs: public input not in field -> invalid
    }
    G1XYZZ sum = G1XYZZ::inf();
    if (!V->ic_mult.empty()) {
        for (int w = 63; w >= 0; w--) {
            if (!sum.is_inf()) for (int d = 0; d < 4; d++) sum = xyzz_dbl(sum);
            for (int j = 0; j < nPublic; j++) { const uint32_t dg = (k[j][w >> 3] >> (4 * (w & 7))) & 15u; if (dg) sum = xyzz_add_affine(sum, V->ic_mult[15 * (size_t)j + dg - 1]); }
        }
    } else sum = g1_sum_of_products(V->ic.data() + 1, (const uint32_t (*)[8])k.data(), nPublic);
    const G1Affine vkx = xyzz_to_affine_gcd(xyzz_add_affine(sum, V->ic[0]));
    const G2Prepared pB = prepare_g2(B);
    const Pair pairs[3] = {{affine_neg(A), &pB}, {vkx, &V->pgamma}, {C, &V->pdelta}};
    return is_one12(final_exp(multi_miller(pairs, 3) * V->m_alpha_beta)) ? 1 : 0;
}

// ---- f4: batch verification (the step on the other side of the path, zk_census_test.go:103-124 run per vote) ----
// N proofs under one key are folded into one pairing-product check with random 128-bit weights rho_i:
//     prod_i e(-rho_i A_i, B_i) * e((sum rho_i) alpha, beta) * e(sum_i rho_i vk_x_i, gamma) * e(sum_i rho_i C_i, delta) == 1
// i.e. N + 3 Miller loops and ONE final exponentiation instead of 4 N and N.  The G1 work (rho_i A_i for every proof and the MSM
// sum rho_i C_i) runs on the GPU with the prover's double-and-add / group-sum kernels; so do the N Miller loops and the membership tests of the B_i from 128
// proofs on (zkc_pairing_dev.hip: one lane per pair writes its lines, a product tree per loop step, the host finishes the accumulator); smaller batches keep them on
// host threads, sixteen pairs per shared accumulator.
// A cheating prover passes with probability about 2^-128 provided the weights are unpredictable to it: `seed32` must be fresh
// randomness (NULL: std::random_device).  Each B_i is checked to lie in the order-r subgroup of the twist (G2 has a cofactor), as
// zkc_verify_bin does.  Returns 1 all valid / 0 at least one invalid / <0 = -ZKC_ERR_*.
namespace {
struct Xoshiro { uint64_t s[4]; uint64_t next() { auto rotl = [](uint64_t x, int k) { return (x << k) | (x >> (64 - k)); };
    const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17; s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45); return r; } };
}
extern "C" int zkc_verify_batch(zkc_ctx* ctx, const uint8_t* vk, int nPublic, const uint8_t* pubs, const uint8_t* proofs, int N, const uint8_t* seed32) {
    g_err.clear();
    if (!ctx || !vk || !pubs || !proofs || nPublic < 0 || N <= 0) return vfail(-ZKC_ERR_BAD_ARG, "zkc_verify_batch: bad argument");
    if (nPublic > 4096) return vfail(-ZKC_ERR_BAD_ARG, "zkc_verify_batch: bad argument");
    int code = 0;
    const std::shared_ptr<const VkReady> V = vk_ready(vk, nPublic, &code);
    if (!V) return code;
    const std::vector<G1Affine>& ic = V->ic;
    const bool vtrace = getenv("ZKC_VERIFY_TRACE") != nullptr; double vt0 = 0, vt1 = 0, vt2 = 0, vt3 = 0;
    auto vnow = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    vt0 = vnow();
    Xoshiro rng;
    if (seed32) memcpy(rng.s, seed32, 32); else { std::random_device rd; for (auto& x : rng.s) x = ((uint64_t)rd() << 32) | rd(); }
    if (!(rng.s[0] | rng.s[1] | rng.s[2] | rng.s[3])) rng.s[0] = 1;
    // ---- parse, per-proof membership checks, weights ----
    std::vector<G1Affine> pts(2 * (size_t)N); std::vector<G2Affine> Bs(N); std::vector<uint32_t> rho(8 * 2 * (size_t)N, 0);
    std::vector<Fr> xsum(nPublic, Fr::zero()); Fr rsum = Fr::zero();
    for (int i = 0; i < N; i++) {                                        // the weights first, from the one generator: the same whatever the number of parsing threads
        uint32_t* r = rho.data() + 8 * (size_t)i;
        const uint64_t lo = rng.next(), hi = rng.next(); r[0] = (uint32_t)lo; r[1] = (uint32_t)(lo >> 32); r[2] = (uint32_t)hi; r[3] = (uint32_t)(hi >> 32);
        memcpy(rho.data() + 8 * ((size_t)N + i), r, 32);
    }
    auto parse_range = [&](int lo, int hi, std::vector<Fr>& xs, Fr& rs) -> bool {
        for (int i = lo; i < hi; i++) {
            const uint8_t* pr = proofs + 256 * (size_t)i;
            if (!rd_g1_std(pts[i], pr) || !rd_g2_std(Bs[i], pr + 64) || !rd_g1_std(pts[N + i], pr + 192)) return false;
            if (!g1_on_curve(pts[i]) || !g1_on_curve(pts[N + i]) || !g2_on_curve(Bs[i])) return false;
            const Fr rm = fp_from_std<FrParams>(rho.data() + 8 * (size_t)i); rs = rs + rm;
            for (int j = 0; j < nPublic; j++) {
                uint32_t k[8]; memcpy(k, pubs + 32 * ((size_t)i * nPublic + j), 32);
                if (!fp_std_lt_p<FrParams>(k)) return false;
                xs[j] = xs[j] + rm * fp_from_std<FrParams>(k);
            }
        }
        return true;
    };
    {
        const unsigned np = N >= 4096 ? std::max(1u, std::min({std::thread::hardware_concurrency(), N >= 32768 ? 16u : 8u})) : 1u;      // a microsecond per proof: worth threads from a few thousand on
        std::vector<std::vector<Fr>> xs(np, std::vector<Fr>(nPublic, Fr::zero())); std::vector<Fr> rs(np, Fr::zero()); std::vector<char> okp(np, 1);
        std::vector<std::thread> th;
        auto run = [&](unsigned t) { okp[t] = parse_range((int)((size_t)N * t / np), (int)((size_t)N * (t + 1) / np), xs[t], rs[t]) ? 1 : 0; };
        for (unsigned t = 1; t < np; t++) th.emplace_back(run, t);
        run(0); for (auto& x : th) x.join();
        for (unsigned t = 0; t < np; t++) { if (!okp[t]) return 0; rsum = rsum + rs[t]; for (int j = 0; j < nPublic; j++) xsum[j] = xsum[j] + xs[t][j]; }
    }
    vt1 = vnow();
    const char* gpu_e = getenv("ZKC_VERIFY_BATCH_GPU"); const int gpu_env = gpu_e ? atoi(gpu_e) : -1;
    const bool on_gpu = gpu_env < 0 ? N >= 128 : gpu_env != 0;
    Fq12 gpu_product = one12(); int gpu_bad = 0;
    // groups: N singletons (rho_i A_i), then the rho_i C_i in runs of 64 -- a group is summed by ONE lane, and one lane adding all N of them was 100 ms at N = 8192
    const uint32_t ncg = ((uint32_t)N + 63) / 64, ngroups = (uint32_t)N + ncg;
    std::vector<G1XYZZ> gout(ngroups);
    {
        ZKC_LOCK(ctx);
        if (hipSetDevice(ctx->device) != hipSuccess) return vfail(-ZKC_ERR_HIP, "zkc_verify_batch: hipSetDevice failed");     // never a positive code: 1 means "all valid"
        std::vector<uint32_t> idx(2 * (size_t)N), gs((size_t)ngroups + 1);
        for (size_t i = 0; i < idx.size(); i++) idx[i] = (uint32_t)i;
        for (int i = 0; i < N; i++) gs[i] = (uint32_t)i;
        for (uint32_t g = 0; g <= ncg; g++) gs[(size_t)N + g] = (uint32_t)N + std::min(64 * g, (uint32_t)N);
        // device buffers from the context's verifier work space (kept between calls while small: zkc_internal.h)
        const int rc = [&]() -> int {
            void *d_pts, *d_rho, *d_idx, *d_gs, *d_tmp, *d_gout; int e;
            if ((e = zkc_vws(ctx, zkc_ctx::VWS_PTS, pts.size() * sizeof(G1Affine), &d_pts)) || (e = zkc_vws(ctx, zkc_ctx::VWS_RHO, rho.size() * 4, &d_rho)) ||
                (e = zkc_vws(ctx, zkc_ctx::VWS_IDX, idx.size() * 4, &d_idx)) || (e = zkc_vws(ctx, zkc_ctx::VWS_GS, gs.size() * 4, &d_gs)) ||
                (e = zkc_vws(ctx, zkc_ctx::VWS_FOLD_TMP, 2 * (size_t)N * sizeof(G1XYZZ), &d_tmp)) || (e = zkc_vws(ctx, zkc_ctx::VWS_FOLD_OUT, (size_t)ngroups * sizeof(G1XYZZ), &d_gout))) return e;
            if (on_gpu && (e = miller_membership_begin(ctx, Bs.data(), (uint32_t)N))) return e;       // the B_i go up; their membership tests (second stream) and the lines of their Miller loops (third) start beside all that follows
            struct Join { zkc_ctx* c; bool armed; ~Join() { if (armed) miller_join(c); } } join{ctx, on_gpu};      // whatever happens below, that kernel is through before the buffers can be trimmed
            ZKC_HIP_CHECK(ctx, hipMemcpyAsync(d_pts, pts.data(), pts.size() * sizeof(G1Affine), hipMemcpyHostToDevice, ctx->stream));
            ZKC_HIP_CHECK(ctx, hipMemcpyAsync(d_rho, rho.data(), rho.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            ZKC_HIP_CHECK(ctx, hipMemcpyAsync(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            ZKC_HIP_CHECK(ctx, hipMemcpyAsync(d_gs, gs.data(), gs.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            if ((e = fold_group_sums_g1_ws(ctx, (const G1Affine*)d_pts, (const uint32_t*)d_rho, (const uint32_t*)d_idx, 2 * (uint32_t)N, (const uint32_t*)d_gs, ngroups,
                                           (G1XYZZ*)d_tmp, (G1XYZZ*)d_gout, gout.data()))) return e;
            if (on_gpu && (e = miller_product_dev(ctx, (const G1XYZZ*)d_gout, (uint32_t)N, &gpu_product, &gpu_bad))) return e;
            return ZKC_OK;
        }();
        zkc_verify_ws_trim(ctx, (size_t)256 << 20);
        if (rc) return vfail(-rc, std::string("zkc_verify_batch: ") + zkc_last_error(ctx));
    }
    vt2 = vnow();
    // ---- vk_x side: (sum rho) IC0 + sum_j (sum_i rho_i x_ij) IC_j ----
    std::vector<std::array<uint32_t, 8>> ks((size_t)nPublic + 1);
    fp_to_std<FrParams>(ks[0].data(), rsum);
    for (int j = 0; j < nPublic; j++) fp_to_std<FrParams>(ks[j + 1].data(), xsum[j]);
    const G1XYZZ vx = g1_sum_of_products(ic.data(), (const uint32_t (*)[8])ks.data(), nPublic + 1);
    const G1Affine ralpha = xyzz_to_affine_gcd(g1_sum_of_products(&V->alpha, (const uint32_t (*)[8])ks.data(), 1));
    // ---- Miller loops on host threads (and the subgroup check of every B_i): a thread's pairs share one accumulator, sixteen at a time ----
    const unsigned nthr = std::max(1u, std::min({std::thread::hardware_concurrency(), 32u, ((unsigned)N + 7) / 8}));
    std::vector<Fq12> part(nthr, one12()); std::vector<int> bad(nthr, 0);
    auto work = [&](unsigned t) {
        constexpr int CH = 16;
        const int lo = (int)((size_t)N * t / nthr), hi = (int)((size_t)N * (t + 1) / nthr);
        Fq12 f = one12(); G2Prepared prep[CH]; Pair pairs[CH];
        for (int i0 = lo; i0 < hi; i0 += CH) {
            const int n = std::min(CH, hi - i0);
            for (int k = 0; k < n; k++) {
                const int i = i0 + k;
                if (!g2_in_subgroup(Bs[i])) { bad[t] = 1; return; }
                prep[k] = prepare_g2(Bs[i]);
                pairs[k] = {affine_neg(xyzz_to_affine_gcd(gout[i])), &prep[k]};
            }
            f = f * multi_miller(pairs, (size_t)n);
        }
        part[t] = f;
    };
    if (on_gpu) { if (gpu_bad) return 0; part.assign(1, gpu_product); }
    else {
        std::vector<std::thread> th; for (unsigned t = 1; t < nthr; t++) th.emplace_back(work, t);
        work(0); for (auto& x : th) x.join();
        for (unsigned t = 0; t < nthr; t++) if (bad[t]) return 0;
    }
    G1XYZZ csum = G1XYZZ::inf(); for (uint32_t g = 0; g < ncg; g++) csum = xyzz_add(csum, gout[(size_t)N + g]);
    const Pair tail[3] = {{ralpha, &V->pbeta}, {xyzz_to_affine_gcd(vx), &V->pgamma}, {xyzz_to_affine_gcd(csum), &V->pdelta}};
    Fq12 f = multi_miller(tail, 3);
    for (const Fq12& x : part) f = f * x;
    const int verdict = is_one12(final_exp(f)) ? 1 : 0;
    vt3 = vnow();
    if (vtrace) fprintf(stderr, "zkc_verify_batch N=%d: parse %.2f ms, device %.2f ms, host tail %.2f ms\n", N, vt1 - vt0, vt2 - vt1, vt3 - vt2);
    return verdict;
}

// JSON surface: the three artifact files of the reference (verification_key.json, signals.json, proof.json). 1 valid / 0 invalid / <0 error
extern "C" int zkc_verify(const char* vkey_json, const char* public_json, const char* proof_json) {
    g_err.clear();                                       // the text belongs to THIS call: a plain invalid proof (0) leaves it empty
    if (!vkey_json || !public_json || !proof_json) return vfail(-ZKC_ERR_BAD_ARG, "zkc_verify: bad argument");
    std::vector<uint8_t> vkb, pubb, prb; int np = 0; std::string perr;
    const int rc = parse::verify_inputs_from_json(vkey_json, public_json, proof_json, vkb, pubb, prb, np, perr);
    if (rc < 0) return vfail(-ZKC_ERR_FORMAT, perr);
    if (rc == 0) return 0;
    return zkc_verify_bin(vkb.data(), np, pubb.data(), prb.data());
}

// prover.ParseProof (zk_census_test.go:118) at the C ABI: proof.json + signals.json texts -> the binary forms every other entry point takes.  1 parsed / 0 well-formed
// documents with a value that is no encoding / <0 = -ZKC_ERR_* (text in zkc_verify_last_error)
extern "C" int zkc_proof_from_json(const char* proof_json, const char* public_json, uint8_t proof[256], uint8_t* pub, int* nPublic) {
    g_err.clear();
    if (!proof_json || !public_json || !nPublic) return vfail(-ZKC_ERR_BAD_ARG, "zkc_proof_from_json: bad argument");
    std::vector<uint8_t> pubb, prb; int np = 0; std::string perr;
    const int rc = parse::proof_from_json(public_json, proof_json, pubb, prb, np, perr);
    if (rc < 0) return vfail(-ZKC_ERR_FORMAT, perr);
    const int cap = *nPublic; *nPublic = np;
    if (rc == 0) return 0;
    if ((np && !pub) || cap < np) return vfail(-ZKC_ERR_SHORT_BUFFER, "zkc_proof_from_json: room for fewer public signals than the document holds");
    if (proof) memcpy(proof, prb.data(), 256);
    if (np) memcpy(pub, pubb.data(), 32ull * np);
    return 1;
}
// verification_key.json text -> the layout zkc_verify_bin / zkc_verify_batch take.  1 parsed / <0 = -ZKC_ERR_*; *vk_size: in = room, out = 448 + 64 (nPublic + 1)
extern "C" int zkc_vkey_from_json(const char* vkey_json, uint8_t* vk, unsigned long* vk_size, int* nPublic) {
    g_err.clear();
    if (!vkey_json || !vk_size) return vfail(-ZKC_ERR_BAD_ARG, "zkc_vkey_from_json: bad argument");
    std::vector<uint8_t> vkb; int nIC = 0; std::string perr;
    if (parse::vkey_from_json(vkey_json, vkb, nIC, perr) < 0) return vfail(-ZKC_ERR_FORMAT, perr);
    const unsigned long room = *vk_size; *vk_size = (unsigned long)vkb.size();
    if (nPublic) *nPublic = nIC - 1;
    if (!vk || room < vkb.size()) return vfail(-ZKC_ERR_SHORT_BUFFER, "zkc_vkey_from_json: buffer too small (size written back)");
    memcpy(vk, vkb.data(), vkb.size());
    return 1;
}

// proof 256 B + public signals -> the JSON texts snarkjs / rapidsnark emit (a8); returns needed size when the buffer is short
extern "C" int zkc_proof_to_json(const uint8_t proof[256], const uint8_t* pub, int nPublic, char* proof_buf, unsigned long* proof_size,
                                 char* public_buf, unsigned long* public_size) {
    if (!proof || (!pub && nPublic) || !proof_size || !public_size) return ZKC_ERR_BAD_ARG;
    auto d = [&](int off) { return "\"" + dec_of(proof + off) + "\""; };
    std::string pj = "{\"pi_a\":[" + d(0) + "," + d(32) + ",\"1\"],\"pi_b\":[[" + d(64) + "," + d(96) + "],[" + d(128) + "," + d(160) + "],[\"1\",\"0\"]],\"pi_c\":[" +
                     d(192) + "," + d(224) + ",\"1\"],\"protocol\":\"groth16\",\"curve\":\"bn128\"}";
    std::string sj = "[";
    for (int i = 0; i < nPublic; i++) sj += (i ? ",\"" : "\"") + dec_of(pub + 32 * i) + "\"";
    sj += "]";
    const bool shortbuf = !proof_buf || !public_buf || *proof_size < pj.size() + 1 || *public_size < sj.size() + 1;
    if (shortbuf) {      // report sizes that hold ANY proof of this shape (77 decimal digits per coordinate): r, s differ between calls
        *proof_size = 8 * 80 + 128; *public_size = (unsigned long)nPublic * 80 + 8;
        return ZKC_ERR_SHORT_BUFFER;
    }
    *proof_size = pj.size() + 1; *public_size = sj.size() + 1;
    memcpy(proof_buf, pj.c_str(), pj.size() + 1); memcpy(public_buf, sj.c_str(), sj.size() + 1);
    return ZKC_OK;
}

// .wtns (iden3 binfile) -> pointer to the nWitness x 32 B payload inside the buffer
extern "C" int zkc_wtns_parse(const void* wtns_bytes, unsigned long size, const uint8_t** payload, uint32_t* nWitness) {
    return parse::wtns_view((const uint8_t*)wtns_bytes, size, payload, nWitness) ? ZKC_OK : ZKC_ERR_FORMAT;
}
extern "C" unsigned long zkc_wtns_write(const void* payload, uint32_t nWitness, void* out, unsigned long out_size) {
    const unsigned long need = 12 + 12 + 40 + 12 + 32ul * nWitness;
    if (!out || out_size < need) return need;
    uint8_t* o = (uint8_t*)out; uint32_t v;
    memcpy(o, "wtns", 4); v = 2; memcpy(o + 4, &v, 4); memcpy(o + 8, &v, 4);
    uint64_t sz = 40; v = 1; memcpy(o + 12, &v, 4); memcpy(o + 16, &sz, 8);
    v = 32; memcpy(o + 24, &v, 4); memcpy(o + 28, FrParams::p, 32); memcpy(o + 60, &nWitness, 4);
    sz = 32ull * nWitness; v = 2; memcpy(o + 64, &v, 4); memcpy(o + 68, &sz, 8); memcpy(o + 76, payload, sz);
    return need;
}

// n scalars uniform in [0, r): 254 random bits from the OS generator, rejected when >= r (what snarkjs' Fr.random and rapidsnark do).  The bytes
// come from getrandom(2) in bulk: std::random_device costs a system call or an RDSEED per 32-bit word, 0.25 ms per scalar pair -- half a second
// for the 2048 scalars of a 1024-voter batch whose blinding the caller leaves to the library.
static void os_random(uint8_t* p, size_t n) {
    while (n) {
        const ssize_t got = getrandom(p, n < 4096 ? n : 4096, 0);
        if (got > 0) { p += got; n -= (size_t)got; continue; }
        if (got < 0 && errno == EINTR) continue;
        std::random_device rd;                                      // no getrandom (seccomp, ancient kernel): word by word
        for (size_t i = 0; i < n; i++) p[i] = (uint8_t)rd();
        return;
    }
}
extern "C" void zkc_random_scalars(uint8_t* out, size_t n) {
    std::vector<uint8_t> pool;
    size_t have = 0;
    for (size_t i = 0; i < n; i++) {
        uint32_t t[8];
        do {
            if (have == 0) { pool.resize(32 * std::min<size_t>(n - i + 8, 4096)); os_random(pool.data(), pool.size()); have = pool.size() / 32; }
            memcpy(t, pool.data() + 32 * (--have), 32); t[7] &= 0x3fffffffu;
        } while (!fp_std_lt_p<FrParams>(t));
        memcpy(out + 32 * i, t, 32);
    }
    if (!pool.empty()) memset(pool.data(), 0, pool.size());          // unused draws do not linger
}

// rapidsnark's entry point (prover.h), byte for byte: whole .zkey and .wtns buffers in, NUL-terminated JSON out.
// r and s are drawn from the OS generator like the reference provers do.  Re-entrant, and concurrent callers (goroutines over prover.Prove,
// zk_census_test.go:89) are coalesced: the call enqueues its witness with the process-wide proving service (zkc_service.hip; devices from
// $ZKC_DEVICE) and waits for its own proof.  The resident key is identified per call by the sampled fingerprint of the .zkey image and, once per
// (image, resident key), by the SHA-256 of the whole image (the identity the reference publishes for its keys,
// artifacts/zkCensus/dev/circuits-info.md:5).  When a buffer is too short the required sizes are written back WITHOUT proving (they depend only on nPublic).
extern "C" int groth16_prover(const void* zkey_buffer, unsigned long zkey_size, const void* wtns_buffer, unsigned long wtns_size,
                              char* proof_buffer, unsigned long* proof_size, char* public_buffer, unsigned long* public_size,
                              char* error_msg, unsigned long error_msg_maxsize) {
    auto err = [&](int code, const std::string& m) { if (error_msg && error_msg_maxsize) snprintf(error_msg, error_msg_maxsize, "%s", m.c_str()); return code; };
    if (!zkey_buffer || !wtns_buffer || !proof_size || !public_size) return err(ZKC_ERR_GENERIC, "groth16_prover: null argument");
    // everything that does not need the GPU first: file shapes and buffer sizes
    parse::BinSections bs; parse::ZkeyHeader zh; std::string perr;
    if (!parse::binfile_sections((const uint8_t*)zkey_buffer, zkey_size, "zkey", 1, bs, perr) || !parse::zkey_check(bs, zh, perr, false)) return err(ZKC_ERR_GENERIC, perr);      // the coefficient scan is the loader's
    const uint8_t* payload; uint32_t nw;
    if (zkc_wtns_parse(wtns_buffer, wtns_size, &payload, &nw)) return err(ZKC_ERR_GENERIC, "Invalid witness file");
    if (nw != zh.nVars) return err(ZKC_ERR_INVALID_WITNESS_LENGTH, "Invalid witness length. Circuit: " + std::to_string(zh.nVars) + ", witness: " + std::to_string(nw));
    const unsigned long need_proof = 8 * 80 + 128, need_public = (unsigned long)zh.nPub * 80 + 8;      // hold ANY proof of this shape (77 decimal digits per coordinate)
    if (!proof_buffer || !public_buffer || *proof_size < need_proof || *public_size < need_public) {
        *proof_size = need_proof; *public_size = need_public;
        return err(ZKC_ERR_SHORT_BUFFER, "Proof or public signals buffer is too short");
    }
    zkc_service* svc = zkc_service_default();
    if (!svc) return err(ZKC_ERR_GENERIC, zkc_service_last_error());
    uint8_t proof[256]; std::vector<uint8_t> pub(32 * (size_t)zh.nPub + 1); char etext[512] = {0};
    int rc = zkc_service_prove(svc, zkey_buffer, zkey_size, payload, nw, nullptr, proof, pub.data(), etext, sizeof etext);
    if (rc) return err(rc == ZKC_ERR_INVALID_WITNESS_LENGTH ? rc : ZKC_ERR_GENERIC, etext);
    rc = zkc_proof_to_json(proof, pub.data(), (int)zh.nPub, proof_buffer, proof_size, public_buffer, public_size);
    if (rc == ZKC_ERR_SHORT_BUFFER) return err(ZKC_ERR_SHORT_BUFFER, "Proof or public signals buffer is too short");
    return rc;
}
