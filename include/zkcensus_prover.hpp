// zkcensus_prover.hpp -- the reference's HOST interface for this path, in C++ over the C ABI of zkcensus.h (header only; link libzkcensus.so).
//
// The reference's host side is Go (zk_census_test.go:11 imports go.vocdoni.io/dvote/crypto/zk/prover; internal/*.go hold the input generator) and this image has no Go
// toolchain, so the same interface is restated here with the same names, argument meaning and error behaviour, and tests/host/reference_test_shape.cc reads like
// zk_census_test.go.  INTEGRATION.md section 1 holds the cgo binding a maintainer of the Go package would add instead.
//
//   prover::Prove(zkey, wasm, inputs)            zk_census_test.go:89    three file images in, a Proof out
//   Proof::Bytes()                               zk_census_test.go:93    proof.json / signals.json texts, byte-equal in form to the committed artifacts
//   prover::ParseProof(proofData, pubSignals)    zk_census_test.go:118
//   Proof::Verify(vkey)                          zk_census_test.go:122
//   internal::BigToFF / BytesToArbo / GenTree    internal/helpers.go:16-85
//   internal::MockInputs, circuitInputs::Bytes   internal/inputs.go:33-105
//
// Go returns (value, error); these throw zkcensus::Error with the library's text instead.  The package itself (go.vocdoni.io/dvote v1.7.1-0.20230811121242-379d7356fa06, go.mod:7) is not vendored in the
// reference: the names and shapes are the ones its call sites and committed artifacts show.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "zkcensus.h"

namespace zkcensus {
struct Error : std::runtime_error { using std::runtime_error::runtime_error; };
using ByteSlice = std::string;                                     // a Go []byte: a file image or a JSON text
using Big = std::array<uint8_t, 32>;                               // a *big.Int below 2^256, little endian (the C ABI's standard form)

namespace detail {
inline std::string dec(const uint8_t le[32]) {                     // (*big.Int).String()
    uint32_t s[8]; memcpy(s, le, 32); std::string out; bool nz = true;
    while (nz) {
        uint64_t rem = 0; nz = false;
        for (int i = 7; i >= 0; i--) { const uint64_t cur = (rem << 32) | s[i]; s[i] = (uint32_t)(cur / 10); rem = cur % 10; if (s[i]) nz = true; }
        out.push_back((char)('0' + rem));
    }
    return std::string(out.rbegin(), out.rend());
}
inline Big big_u64(uint64_t v) { Big b{}; memcpy(b.data(), &v, 8); return b; }
// new(big.Int).SetBytes(b) mod r for a big-endian byte string of any length (a 64-byte signature is 512 bits): double-and-add over the bits, 4 x 64-bit limbs
inline Big mod_r_be(const uint8_t* p, size_t n) {
    static const uint64_t R[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    uint64_t a[4] = {0, 0, 0, 0};
    auto reduce = [&] {                                            // a < 2r on entry
        uint64_t d[4]; unsigned __int128 br = 0;
        for (int i = 0; i < 4; i++) { const unsigned __int128 x = (unsigned __int128)a[i] - R[i] - (uint64_t)br; d[i] = (uint64_t)x; br = (x >> 64) & 1; }
        if (!br) memcpy(a, d, 32);
    };
    for (size_t i = 0; i < n; i++)
        for (int bit = 7; bit >= 0; bit--) {
            uint64_t c = (p[i] >> bit) & 1;
            for (int k = 0; k < 4; k++) { const uint64_t nc = a[k] >> 63; a[k] = (a[k] << 1) | c; c = nc; }      // a < r < 2^254: no bit is lost
            reduce();
        }
    Big out; memcpy(out.data(), a, 32); return out;
}
inline void check(int rc, zkc_ctx* ctx, const char* what) {
    if (rc != ZKC_OK) { const char* m = zkc_last_error(ctx); throw Error(std::string(what) + ": " + (m && *m ? m : "error " + std::to_string(rc))); }
}
inline zkc_ctx* context() {                                        // one device context per process for the generator's hashing (device 0)
    static zkc_ctx* ctx = [] { zkc_ctx* c = nullptr; const int rc = zkc_ctx_create(0, &c); if (rc != ZKC_OK) { const char* m = zkc_last_error(nullptr); throw Error(std::string("zkc_ctx_create: ") + (m ? m : "")); } return c; }();
    return ctx;
}
inline std::string quoted_list(const std::vector<std::string>& v) {
    std::string s = "["; for (size_t i = 0; i < v.size(); i++) { s += i ? ",\"" : "\""; s += v[i]; s += "\""; } return s + "]";
}
}  // namespace detail

namespace prover {
struct ProofData {                                                 // proof.json: pi_a, pi_b, pi_c as decimal strings
    std::vector<std::string> A; std::vector<std::vector<std::string>> B; std::vector<std::string> C;
};
struct Proof {
    ProofData Data; std::vector<std::string> PubSignals;
    ByteSlice raw_proof, raw_signals;                                  // set instead of Data when a document parsed but holds a value that is no encoding (Verify refuses it)
    // (*Proof).Bytes(): json.Marshal of Data and of PubSignals -- compact, members pi_a, pi_b, pi_c in that order, nothing else (artifacts/.../proof.json, signals.json)
    std::pair<ByteSlice, ByteSlice> Bytes() const {
        if (Data.A.empty()) return {raw_proof, raw_signals};
        std::string p = "{\"pi_a\":" + detail::quoted_list(Data.A) + ",\"pi_b\":[";
        for (size_t i = 0; i < Data.B.size(); i++) { if (i) p += ","; p += detail::quoted_list(Data.B[i]); }
        p += "],\"pi_c\":" + detail::quoted_list(Data.C) + "}";
        return {p, detail::quoted_list(PubSignals)};
    }
    // (*Proof).Verify(vkey): the file image of verification_key.json; returns on a valid proof, throws otherwise
    void Verify(const ByteSlice& vkey) const {
        const auto t = Bytes();
        const int rc = zkc_verify(vkey.c_str(), t.second.c_str(), t.first.c_str());
        if (rc == 1) return;
        if (rc == 0) throw Error("proof verification failed");
        const char* m = zkc_verify_last_error(); throw Error(std::string("verify: ") + (m ? m : ""));
    }
};
// prover.ParseProof: both documents read as encoding/json would (strict JSON, pi_a / pi_c three strings, pi_b three pairs, signals an array of strings)
inline Proof ParseProof(const ByteSlice& proofData, const ByteSlice& pubSignals) {
    uint8_t bin[256]; std::vector<uint8_t> pub(32 * 4096); int n = 4096;
    const int rc = zkc_proof_from_json(proofData.c_str(), pubSignals.c_str(), bin, pub.data(), &n);
    if (rc < 0) { const char* m = zkc_verify_last_error(); throw Error(std::string("parsing proof: ") + (m ? m : "")); }
    Proof p;
    if (rc == 0) { p.raw_proof = proofData; p.raw_signals = pubSignals; return p; }
    auto d = [&](int off) { return detail::dec(bin + off); };
    p.Data.A = {d(0), d(32), "1"}; p.Data.B = {{d(64), d(96)}, {d(128), d(160)}, {"1", "0"}}; p.Data.C = {d(192), d(224), "1"};
    for (int i = 0; i < n; i++) p.PubSignals.push_back(detail::dec(pub.data() + 32 * i));
    return p;
}
// prover.Prove(zkey, wasm, inputs): the three file images of proving_key.zkey, circuit.wasm and inputs_example.json.  An empty wasm names the circuit by the
// key's own shape (include/zkcensus.h groth16_fullprove).  Throws with the witness calculator's message when the inputs fail a circuit assert.
inline Proof Prove(const ByteSlice& zkey, const ByteSlice& wasm, const ByteSlice& inputs) {
    std::vector<char> pb(2048), ub(2048); char err[512] = {0};
    for (int attempt = 0; attempt < 2; attempt++) {
        unsigned long ps = pb.size(), us = ub.size();
        const int rc = groth16_fullprove(zkey.data(), zkey.size(), wasm.empty() ? nullptr : wasm.data(), wasm.size(), inputs.data(), inputs.size(),
                                         pb.data(), &ps, ub.data(), &us, err, sizeof err);
        if (rc == 0) return ParseProof(pb.data(), ub.data());
        if (rc == 2 && attempt == 0) { pb.assign(ps, 0); ub.assign(us, 0); continue; }                  // PROVER_ERROR_SHORT_BUFFER: sizes written back
        throw Error(std::string(err[0] ? err : "groth16_fullprove failed"));
    }
    throw Error("groth16_fullprove: short buffer");
}
}  // namespace prover

namespace internal {
// internal/helpers.go:16-26
inline Big BigToFF(const uint8_t* big_endian, size_t n) { return detail::mod_r_be(big_endian, n); }
// internal/helpers.go:28-34: sha256, each half read as a little-endian integer
inline std::array<Big, 2> BytesToArbo(const uint8_t* input, size_t n) {
    uint8_t h[32]; zkc_sha256(input, n, h);
    std::array<Big, 2> out{}; memcpy(out[0].data(), h, 16); memcpy(out[1].data(), h + 16, 16); return out;
}
struct Tree { Big root; int nSiblings; std::vector<Big> siblings; };
// internal/helpers.go:36-85 GenTree: an arbo Poseidon tree holding (key, value) and n - 1 leaves of random 20-byte keys with value 1; the proof of `key`, its siblings
// zero-padded to nLevels (the reference fixes 160).  Keys and values are arbo byte strings: little-endian integers.  Hashing and the sibling walk run on the GPU
// (zkc_smt_build); pebbledb has no part in it.
inline Tree GenTree(const uint8_t key[20], const Big& value, int n, int nLevels = 160) {
    std::mt19937_64 rng{std::random_device{}()};
    std::vector<uint8_t> keys(32ull * n, 0), vals(32ull * n, 0);
    memcpy(keys.data(), key, 20); memcpy(vals.data(), value.data(), 32);
    for (int i = 1; i < n; i++) { for (int k = 0; k < 20; k++) keys[32ull * i + k] = (uint8_t)rng(); vals[32ull * i] = 1; }
    Tree t; t.siblings.assign(nLevels, Big{});
    std::vector<uint8_t> sib(32ull * n * (nLevels + 1)); std::vector<int32_t> depth(n);
    detail::check(zkc_smt_build(detail::context(), keys.data(), vals.data(), n, nLevels, t.root.data(), sib.data(), depth.data()), detail::context(), "GenTree");
    for (int l = 0; l < nLevels; l++) memcpy(t.siblings[l].data(), sib.data() + 32ull * l, 32);
    t.nSiblings = depth[0];
    return t;
}
// internal/inputs.go:14-31
struct circuitInputs {
    std::vector<std::string> ElectionId; std::string Nullifier, AvailableWeight; std::vector<std::string> VoteHash; std::string SikRoot, CensusRoot;
    std::string Address, Password, Signature, VoteWeight; std::vector<std::string> CensusSiblings, SikSiblings;
    // internal/inputs.go:100-103: json.MarshalIndent(inputs, "", "\t")
    ByteSlice Bytes() const {
        std::string s = "{\n";
        auto str = [&](const char* k, const std::string& v, bool last = false) { s += std::string("\t\"") + k + "\": \"" + v + "\"" + (last ? "\n" : ",\n"); };
        auto arr = [&](const char* k, const std::vector<std::string>& v, bool last = false) {
            s += std::string("\t\"") + k + "\": [\n";
            for (size_t i = 0; i < v.size(); i++) s += "\t\t\"" + v[i] + "\"" + (i + 1 < v.size() ? ",\n" : "\n");
            s += last ? "\t]\n" : "\t],\n";
        };
        arr("electionId", ElectionId); str("nullifier", Nullifier); str("availableWeight", AvailableWeight); arr("voteHash", VoteHash); str("sikRoot", SikRoot);
        str("censusRoot", CensusRoot); str("address", Address); str("password", Password); str("signature", Signature); str("voteWeight", VoteWeight);
        arr("censusSiblings", CensusSiblings); arr("sikSiblings", SikSiblings, true);
        return s + "}";
    }
};
// internal/inputs.go:33-98 MockInputs.  An account is a random 20-byte address and a random 64-byte signature (the SIK signature without its recovery byte, ts_inputs/src/inputs.ts:6-13) here: deriving them (secp256k1, vocdoni's SIK payload)
// is the node's business, and the circuit sees only the field elements.  Everything from there on follows the reference: password "password123", weight 10 of which
// 5 are spent, SIK = H(address, password, signature), nullifier = H(signature, password, electionId), one tree each of nKeys leaves, siblings padded to nLevels + 1.
// (The reference ignores nKeys and builds 10-leaf trees; 10 is what its only caller passes.)
inline circuitInputs MockInputs(int nLevels, int nKeys) {
    zkc_ctx* ctx = detail::context();
    std::random_device rd; uint8_t address[20], signature[64];
    for (auto& b : address) b = (uint8_t)rd();
    for (auto& b : signature) b = (uint8_t)rd();
    const char* password = "password123";
    const Big availableWeight = detail::big_u64(10);
    Big addr{}; memcpy(addr.data(), address, 20);                                                       // arbo.BytesToBigInt: little endian
    const Big pw = BigToFF((const uint8_t*)password, strlen(password)), sig = BigToFF(signature, sizeof signature);
    static const uint8_t electionId[32] = {0x7f, 0xae, 0xab, 0x7a, 0x7d, 0x25, 0x05, 0x27, 0xd6, 0x14, 0xe9, 0x52, 0xae, 0x8e, 0x44, 0x68,
                                           0x25, 0xbd, 0x11, 0x24, 0xc6, 0xde, 0xf4, 0x10, 0x84, 0x4c, 0x7c, 0x38, 0x3d, 0x15, 0x19, 0xa6};      // internal/inputs.go:57
    const auto ffElectionId = BytesToArbo(electionId, 32);
    uint8_t in3[96], in4[128]; Big sik, nullifier;
    memcpy(in3, addr.data(), 32); memcpy(in3 + 32, pw.data(), 32); memcpy(in3 + 64, sig.data(), 32);
    detail::check(zkc_poseidon_batch(ctx, 3, in3, 1, sik.data()), ctx, "AccountSIK");
    memcpy(in4, sig.data(), 32); memcpy(in4 + 32, pw.data(), 32); memcpy(in4 + 64, ffElectionId[0].data(), 32); memcpy(in4 + 96, ffElectionId[1].data(), 32);
    detail::check(zkc_poseidon_batch(ctx, 4, in4, 1, nullifier.data()), ctx, "AccountSIKnullifier");
    const Tree census = GenTree(address, availableWeight, nKeys, nLevels), siks = GenTree(address, sik, nKeys, nLevels);
    const uint8_t weightBytes[1] = {10};                                                                 // availableWeight.Bytes()
    const auto voteHash = BytesToArbo(weightBytes, 1);
    circuitInputs ci;
    ci.ElectionId = {detail::dec(ffElectionId[0].data()), detail::dec(ffElectionId[1].data())};
    ci.Nullifier = detail::dec(nullifier.data()); ci.AvailableWeight = "10";
    ci.VoteHash = {detail::dec(voteHash[0].data()), detail::dec(voteHash[1].data())};
    ci.SikRoot = detail::dec(siks.root.data()); ci.CensusRoot = detail::dec(census.root.data());
    ci.Address = detail::dec(addr.data()); ci.Password = detail::dec(pw.data()); ci.Signature = detail::dec(sig.data()); ci.VoteWeight = "5";
    for (auto& s : census.siblings) ci.CensusSiblings.push_back(detail::dec(s.data()));
    ci.CensusSiblings.push_back("0");
    for (auto& s : siks.siblings) ci.SikSiblings.push_back(detail::dec(s.data()));
    ci.SikSiblings.push_back("0");
    return ci;
}
}  // namespace internal
}  // namespace zkcensus
