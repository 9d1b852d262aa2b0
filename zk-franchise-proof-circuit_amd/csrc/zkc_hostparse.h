// zkc_hostparse.h -- the host-only parsers of libzkcensus (product code): iden3 binfile section tables, the Groth16 .zkey header
// checks, .wtns, the decimal-string JSON shapes of proof.json / signals.json / verification_key.json, and SHA-256 (key-cache
// identity, circuit selection by wasm hash: artifacts/zkCensus/dev/circuits-info.md:5-7).
//
// Plain C++17, no HIP: the same header is compiled into libzkcensus.so by hipcc and, with -fsanitize=address,undefined, into
// tests/host/parse_asan.cc, which feeds it truncated / oversized / mutated files (tests/test_host_parsers_asan.py).  Every read
// below is preceded by a bounds check that cannot wrap; nothing here allocates in proportion to an untrusted length field.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>
#include "zkc_json.h"

namespace zkc { namespace parse {

static constexpr uint32_t kFqP[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
static constexpr uint32_t kFrP[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};

inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

// ---- iden3 binfile: magic(4) version(u32) nSections(u32) then per section id(u32) size(u64) payload ----
struct BinSections { const uint8_t* sec[16]; uint64_t ssz[16]; };
inline bool binfile_sections(const uint8_t* buf, size_t len, const char magic[4], uint32_t version, BinSections& out, std::string& err) {
    for (int i = 0; i < 16; i++) { out.sec[i] = nullptr; out.ssz[i] = 0; }
    if (!buf || len < 12 || memcmp(buf, magic, 4) != 0 || rd32(buf + 4) != version) { err = std::string("not a ") + std::string(magic, 4) + " v" + std::to_string(version) + " file"; return false; }
    size_t p = 12;
    for (uint32_t i = 0, ns = rd32(buf + 8); i < ns; i++) {
        if (len - p < 12) { err = "truncated section table"; return false; }
        const uint32_t id = rd32(buf + p); const uint64_t sz = rd64(buf + p + 4); p += 12;
        if (sz > len - p) { err = "truncated section " + std::to_string(id); return false; }        // no wrap: p <= len
        if (id < 16) { out.sec[id] = buf + p; out.ssz[id] = sz; }
        p += (size_t)sz;
    }
    return true;
}

// ---- Groth16 .zkey (snarkjs zkey format v1; circuit/circuit-compiler.sh:112-131 writes it) ----
struct ZkeyHeader { uint32_t nVars, nPub, n, logn, nCoeffs; };
inline bool zkey_check(const BinSections& s, ZkeyHeader& h, std::string& err, bool scan_coefficients = true) {
    for (int i = 1; i <= 9; i++) if (!s.sec[i]) { err = "zkey: missing section " + std::to_string(i); return false; }
    if (s.ssz[1] < 4 || rd32(s.sec[1]) != 1) { err = "zkey: protocol is not groth16"; return false; }
    // section 2: n8q(4) q(32) n8r(4) r(32) nVars nPub domainSize(4 each) alpha1 beta1(64 each) beta2 gamma2(128 each) delta1(64) delta2(128) = 660 bytes
    if (s.ssz[2] < 660) { err = "zkey: header section too short"; return false; }
    const uint8_t* b = s.sec[2];
    if (rd32(b) != 32 || memcmp(b + 4, kFqP, 32) != 0 || rd32(b + 36) != 32 || memcmp(b + 40, kFrP, 32) != 0) { err = "zkey: curve is not bn128"; return false; }
    h.nVars = rd32(b + 72); h.nPub = rd32(b + 76); h.n = rd32(b + 80); h.logn = 0;
    while (h.logn < 31 && (1u << h.logn) < h.n) h.logn++;
    if (h.n < 4 || (1u << h.logn) != h.n || h.logn > 28) { err = "zkey: domain size must be a power of two in [4, 2^28]"; return false; }
    if (h.nVars == 0 || h.nPub >= h.nVars) { err = "zkey: nVars / nPublic out of range"; return false; }
    const uint64_t nv = h.nVars, np = h.nPub, nc = nv - np - 1, n = h.n;
    if (s.ssz[3] != 64 * (np + 1) || s.ssz[5] != 64 * nv || s.ssz[6] != 64 * nv || s.ssz[7] != 128 * nv || s.ssz[8] != 64 * nc || s.ssz[9] != 64 * n) {
        err = "zkey: section sizes do not match the header"; return false;
    }
    if (s.ssz[4] < 4) { err = "zkey: coefficient section too short"; return false; }
    h.nCoeffs = rd32(s.sec[4]);
    if (s.ssz[4] != 4 + 44ull * h.nCoeffs) { err = "zkey: coefficient section size"; return false; }
    const uint8_t* c = s.sec[4] + 4;
    for (uint32_t i = 0; scan_coefficients && i < h.nCoeffs; i++) {
        const uint32_t m = rd32(c + 44ull * i), cc = rd32(c + 44ull * i + 4), w = rd32(c + 44ull * i + 8);
        if (m > 1 || cc >= h.n || w >= h.nVars) { err = "zkey: coefficient out of range"; return false; }
        const uint8_t* v = c + 44ull * i + 12; int k = 7;                          // the value is a field element: little endian, below r
        while (k >= 0 && rd32(v + 4 * k) == kFrP[k]) k--;
        if (k < 0 || rd32(v + 4 * k) > kFrP[k]) { err = "zkey: coefficient value is not a reduced field element"; return false; }
    }
    return true;
}

// ---- .wtns (iden3 binfile "wtns" v2): section 1 = n8(4) prime(32) nWitness(4); section 2 = nWitness x 32 B ----
inline bool wtns_view(const uint8_t* buf, size_t len, const uint8_t** payload, uint32_t* nWitness) {
    BinSections s; std::string err;
    if (!binfile_sections(buf, len, "wtns", 2, s, err)) return false;
    if (!s.sec[1] || !s.sec[2] || s.ssz[1] < 40) return false;
    if (rd32(s.sec[1]) != 32 || memcmp(s.sec[1] + 4, kFrP, 32) != 0) return false;
    const uint32_t nw = rd32(s.sec[1] + 36);
    if (s.ssz[2] != 32ull * nw) return false;
    if (payload) *payload = s.sec[2];
    if (nWitness) *nWitness = nw;
    return true;
}

// ---- decimal strings <-> 32-byte little-endian integers ----
inline bool dec_to_std(const std::string& d, uint32_t out[8]) {          // false: not a decimal number, or >= 2^256
    uint32_t t[8] = {0}; if (d.empty() || d.size() > 80) return false;
    for (char ch : d) {
        if (ch < '0' || ch > '9') return false;
        uint64_t c = (uint64_t)(ch - '0');
        for (int j = 0; j < 8; j++) { c += (uint64_t)t[j] * 10; t[j] = (uint32_t)c; c >>= 32; }
        if (c) return false;
    }
    memcpy(out, t, 32); return true;
}
inline std::string dec_of(const uint8_t* p) {
    uint32_t s[8]; memcpy(s, p, 32); std::string out; bool nz = true;
    while (nz) { uint64_t rem = 0; nz = false; for (int i = 7; i >= 0; i--) { uint64_t cur = (rem << 32) | s[i]; s[i] = (uint32_t)(cur / 10); rem = cur % 10; if (s[i]) nz = true; } out.push_back((char)('0' + rem)); }
    return std::string(out.rbegin(), out.rend());
}
// JSON projective points as snarkjs writes them: G1 [x, y, z], G2 [[x0,x1],[y0,y1],[z0,z1]] with z = 1 (affine) or 0 (infinity), every coordinate a decimal STRING.
// Any other z is rejected: the artifacts never carry one, and silently treating it as affine would accept a different point.
// Each returns 1 = read, 0 = the right shape with a value that is no point encoding (not decimal, >= 2^256, another z): an invalid proof; -1 = the wrong SHAPE (not an
// array of the right length of strings): a malformed document, which encoding/json would refuse to unmarshal (zk_census_test.go:118)
inline int strs_of(const json::Value& v, size_t n, const std::string** out) {
    if (v.type != json::Value::Array || v.a.size() != n) return -1;
    for (size_t i = 0; i < n; i++) { if (v.a[i].type != json::Value::String) return -1; out[i] = &v.a[i].s; }
    return 1;
}
inline int put_g1_json(const json::Value& v, uint8_t* out) {
    const std::string* c[3]; if (strs_of(v, 3, c) < 0) return -1;
    uint32_t z[8]; if (!dec_to_std(*c[2], z)) return 0;
    uint32_t hi = 0; for (int i = 1; i < 8; i++) hi |= z[i];
    if (hi || z[0] > 1) return 0;
    if (z[0] == 0) { memset(out, 0, 64); return 1; }
    uint32_t s[8]; if (!dec_to_std(*c[0], s)) return 0; memcpy(out, s, 32);
    if (!dec_to_std(*c[1], s)) return 0; memcpy(out + 32, s, 32); return 1;
}
inline int put_g2_json(const json::Value& v, uint8_t* out) {
    if (v.type != json::Value::Array || v.a.size() != 3) return -1;
    const std::string* c[6];
    for (int k = 0; k < 3; k++) if (strs_of(v.a[k], 2, c + 2 * k) < 0) return -1;
    uint32_t z0[8], z1[8]; if (!dec_to_std(*c[4], z0) || !dec_to_std(*c[5], z1)) return 0;
    uint32_t hi = 0; for (int i = 1; i < 8; i++) hi |= z0[i]; for (int i = 0; i < 8; i++) hi |= z1[i];
    if (hi || z0[0] > 1) return 0;
    if (z0[0] == 0) { memset(out, 0, 128); return 1; }
    for (int k = 0; k < 4; k++) { uint32_t s[8]; if (!dec_to_std(*c[k], s)) return 0; memcpy(out + 32 * k, s, 32); }
    return 1;
}
// "protocol" / "curve" as snarkjs writes them into verification_key.json and proof.json: where present they must say groth16 / bn128 (go-rapidsnark's types carry
// both; a key or proof of another scheme or curve is not something this verifier can speak to)
inline bool scheme_ok(const json::Value& doc, const char* what, std::string& err) {
    const json::Value* p = doc.find("protocol"); const json::Value* c = doc.find("curve");
    if (p && (p->type != json::Value::String || p->s != "groth16")) { err = std::string(what) + ": protocol is not \"groth16\""; return false; }
    if (c && (c->type != json::Value::String || (c->s != "bn128" && c->s != "bn254" && c->s != "BN128" && c->s != "BN254" && c->s != "altbn128"))) { err = std::string(what) + ": curve is not \"bn128\""; return false; }
    return true;
}
// proof.json + signals.json texts -> proof (256 B) and public signals (np x 32 B), what prover.ParseProof reads (zk_census_test.go:118): both PARSED as JSON
// (zkc_json.h), shapes exactly the reference's.  returns 1 ok, 0 = well-formed documents whose VALUES are no valid encoding (json.Unmarshal takes them, no verifier
// does), -1 = a malformed document (err set)
inline int proof_from_json(const std::string& pj, const std::string& pr, std::vector<uint8_t>& pubb, std::vector<uint8_t>& prb, int& nPublic, std::string& err) {
    json::Value jp, jr; std::string perr;
    if (!json::parse(pj.data(), pj.size(), jp, perr)) { err = "public signals: " + perr; return -1; }
    if (!json::parse(pr.data(), pr.size(), jr, perr)) { err = "proof: " + perr; return -1; }
    if (jr.type != json::Value::Object) { err = "proof JSON: expected an object"; return -1; }
    if (jp.type != json::Value::Array) { err = "public signals JSON: expected an array of decimal strings"; return -1; }
    for (auto& x : jp.a) if (x.type != json::Value::String) { err = "public signals JSON: expected an array of decimal strings"; return -1; }
    if (!scheme_ok(jr, "proof JSON", err)) return -1;
    const json::Value *pa = jr.find("pi_a"), *pb = jr.find("pi_b"), *pc = jr.find("pi_c");
    if (!pa || !pb || !pc) { err = "proof JSON: missing member"; return -1; }
    const size_t np = jp.a.size();
    if (np > 4096) { err = "public signals JSON: too many signals"; return -1; }
    pubb.assign(32 * np + 1, 0); prb.assign(256, 0); nPublic = (int)np;
    const int ra = put_g1_json(*pa, prb.data()), rb = put_g2_json(*pb, prb.data() + 64), rc = put_g1_json(*pc, prb.data() + 192);
    if (ra < 0 || rb < 0 || rc < 0) { err = "proof JSON: pi_a / pi_c must be three decimal strings, pi_b three pairs"; return -1; }
    for (size_t i = 0; i < np; i++) { uint32_t s[8]; if (!dec_to_std(jp.a[i].s, s)) return 0; memcpy(pubb.data() + 32 * i, s, 32); }
    if (ra == 0 || rb == 0 || rc == 0) return 0;
    return 1;
}
// verification_key.json text -> the binary layout of zkc_verify_bin (alpha1 beta2 gamma2 delta2 IC[nIC]); 1 ok, -1 malformed (err set)
inline int vkey_from_json(const std::string& vk, std::vector<uint8_t>& vkb, int& nIC, std::string& err) {
    json::Value jv; std::string perr;
    if (!json::parse(vk.data(), vk.size(), jv, perr)) { err = "verification key: " + perr; return -1; }
    if (jv.type != json::Value::Object) { err = "verification key JSON: expected an object"; return -1; }
    if (!scheme_ok(jv, "verification key JSON", err)) return -1;
    const json::Value *a1 = jv.find("vk_alpha_1"), *b2 = jv.find("vk_beta_2"), *g2 = jv.find("vk_gamma_2"), *d2 = jv.find("vk_delta_2"), *ic = jv.find("IC"), *npub = jv.find("nPublic");
    if (!a1 || !b2 || !g2 || !d2 || !ic) { err = "verification key JSON: missing member"; return -1; }
    if (ic->type != json::Value::Array || ic->a.empty() || ic->a.size() > 4097) { err = "verification key: IC length does not match the public signals"; return -1; }
    const size_t n = ic->a.size();
    if (npub && (npub->type != json::Value::Number || npub->s != std::to_string(n - 1))) { err = "verification key: nPublic does not match the public signals"; return -1; }
    vkb.assign(448 + 64 * n, 0);
    if (put_g1_json(*a1, vkb.data()) != 1 || put_g2_json(*b2, vkb.data() + 64) != 1 || put_g2_json(*g2, vkb.data() + 192) != 1 || put_g2_json(*d2, vkb.data() + 320) != 1) {
        err = "verification key JSON: bad point"; return -1;
    }
    for (size_t i = 0; i < n; i++) if (put_g1_json(ic->a[i], vkb.data() + 448 + 64 * i) != 1) { err = "verification key JSON: bad IC point"; return -1; }
    nIC = (int)n;
    return 1;
}
// verification_key.json + signals.json + proof.json texts -> the binary layouts of zkc_verify_bin.  [r5] The three are PARSED as JSON (zkc_json.h) and must have the
// reference's shapes exactly: round 4 collected quoted strings and ignored everything between them, so a document with stray tokens, a damaged member name or a missing
// comma still verified where prover.ParseProof's json.Unmarshal (zk_census_test.go:118) and snarkjs's JSON.parse refuse it.
// returns 1 ok, 0 = well-formed documents whose VALUES are no valid encoding (an invalid proof), -1 = a malformed document (err set)
inline int verify_inputs_from_json(const std::string& vk, const std::string& pj, const std::string& pr, std::vector<uint8_t>& vkb, std::vector<uint8_t>& pubb,
                                   std::vector<uint8_t>& prb, int& nPublic, std::string& err) {
    int nIC = 0;
    if (vkey_from_json(vk, vkb, nIC, err) < 0) return -1;
    const int rp = proof_from_json(pj, pr, pubb, prb, nPublic, err);
    if (rp < 0) return -1;
    if (nIC != nPublic + 1) { err = "verification key: IC length does not match the public signals"; return -1; }
    return rp;
}

// ---- circuit inputs as the reference hands them over: the text of inputs_example.json (zk_census_test.go:85-89, prover.Prove's third argument; internal/inputs.go:14-31
// is its schema) -> the flat block of the C ABI, census.circom:51-67 declaration order, 32-byte little-endian values reduced mod r.  What circom_runtime 0.1.22's
// witness calculator does with the parsed object (witness_calculator.js _doCalculateWitness: flatArray, normalize = BigInt(v) mod r), with its messages:
//   a name that is no input signal of the circuit      "Signal <name> not found\n"
//   more values than the signal has                    "Too many values for input signal <name>\n"
//   fewer (other than a sibling list, see below)       "Not enough values for input signal <name>\n"
//   signals left unset at the end                      "Not all inputs have been set. Only <k> out of <n>"
//   a value BigInt() would not take                    "Cannot convert <text> to a BigInt"
// Values: decimal strings (what the reference's generators write, internal/inputs.go:82-97), "0x" hex strings and integer literals of any length (not rounded through a
// double), with a sign; nested arrays are flattened.  One deliberate extension, kept from rounds 1-4's hosts: censusSiblings / sikSiblings shorter than nLevels + 1 are
// padded with zeros (an arbo proof has as many siblings as the leaf is deep; the generators pad, inputs.go:90-97, callers need not).
inline const char* const* circuit_input_names() {
    static const char* const k[12] = {"electionId", "nullifier", "availableWeight", "voteHash", "sikRoot", "censusRoot", "address", "password", "signature", "voteWeight", "censusSiblings", "sikSiblings"};
    return k;
}
inline void fr_addmod(uint64_t a[4], const uint64_t b[4]) {                      // a = a + b mod r, a, b < r
    static const uint64_t R[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    unsigned __int128 c = 0; uint64_t t[4];
    for (int i = 0; i < 4; i++) { c += (unsigned __int128)a[i] + b[i]; t[i] = (uint64_t)c; c >>= 64; }
    uint64_t d[4]; unsigned __int128 br = 0;
    for (int i = 0; i < 4; i++) { const unsigned __int128 x = (unsigned __int128)t[i] - R[i] - (uint64_t)br; d[i] = (uint64_t)x; br = (x >> 64) & 1; }
    const bool ge = c != 0 || br == 0;
    for (int i = 0; i < 4; i++) a[i] = ge ? d[i] : t[i];
}
// text of an integer (decimal, or 0x / 0X hex; optional sign) -> its residue mod r as 32 little-endian bytes; false: BigInt() would throw
inline bool integer_mod_r(const std::string& txt, uint8_t out[32]) {
    size_t i = 0; bool neg = false;
    while (i < txt.size() && (txt[i] == ' ' || txt[i] == '\t' || txt[i] == '\n' || txt[i] == '\r')) i++;      // BigInt("  12 ") = 12n
    size_t e = txt.size(); while (e > i && (txt[e - 1] == ' ' || txt[e - 1] == '\t' || txt[e - 1] == '\n' || txt[e - 1] == '\r')) e--;
    if (i == e) { memset(out, 0, 32); return true; }                             // BigInt("") = 0n
    bool sign = false;
    if (txt[i] == '-' || txt[i] == '+') { sign = true; neg = txt[i] == '-'; i++; }
    const bool hex = e - i > 2 && txt[i] == '0' && (txt[i + 1] == 'x' || txt[i + 1] == 'X');
    if (hex) { if (sign) return false; i += 2; }                                   // BigInt("-0x1") throws
    if (i == e) return false;
    uint64_t acc[4] = {0, 0, 0, 0};
    if (!hex && e - i <= 77) {
        // what every real input is: a decimal of at most 77 digits (10^77 < 2^256).  Nineteen digits at a time into a 64-bit chunk, acc = acc * 10^k + chunk over four limbs,
        // then at most five subtractions of r (2^256 / r < 5.3).  (The digit-by-digit loop below costs 385 modular additions per 77-digit value: 4 us, and a voter has twenty.)
        static const uint64_t P10[20] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull, 1000000000ull, 10000000000ull, 100000000000ull, 1000000000000ull,
                                         10000000000000ull, 100000000000000ull, 1000000000000000ull, 10000000000000000ull, 100000000000000000ull, 1000000000000000000ull, 10000000000000000000ull};
        static const uint64_t R[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
        while (i < e) {
            const size_t k = std::min<size_t>(19, e - i); uint64_t chunk = 0;
            for (size_t j = 0; j < k; j++) { const char ch = txt[i + j]; if (ch < '0' || ch > '9') return false; chunk = chunk * 10 + (uint64_t)(ch - '0'); }
            unsigned __int128 c = chunk;
            for (int l = 0; l < 4; l++) { c += (unsigned __int128)acc[l] * P10[k]; acc[l] = (uint64_t)c; c >>= 64; }
            i += k;
        }
        for (;;) {
            uint64_t d[4]; unsigned __int128 br = 0;
            for (int l = 0; l < 4; l++) { const unsigned __int128 x = (unsigned __int128)acc[l] - R[l] - (uint64_t)br; d[l] = (uint64_t)x; br = (x >> 64) & 1; }
            if (br) break;
            memcpy(acc, d, 32);
        }
    }
    for (; i < e; i++) {
        const char ch = txt[i]; int dgt;
        if (ch >= '0' && ch <= '9') dgt = ch - '0';
        else if (hex && ch >= 'a' && ch <= 'f') dgt = ch - 'a' + 10;
        else if (hex && ch >= 'A' && ch <= 'F') dgt = ch - 'A' + 10;
        else return false;
        uint64_t x2[4], x8[4]; memcpy(x2, acc, 32); fr_addmod(x2, acc);          // 2 acc
        memcpy(x8, x2, 32); fr_addmod(x8, x2); fr_addmod(x8, x8);               // 8 acc
        if (hex) { fr_addmod(x8, x8); memcpy(acc, x8, 32); }                     // 16 acc
        else { fr_addmod(x8, x2); memcpy(acc, x8, 32); }                         // 10 acc
        const uint64_t dd[4] = {(uint64_t)dgt, 0, 0, 0}; fr_addmod(acc, dd);
    }
    if (neg && (acc[0] | acc[1] | acc[2] | acc[3])) {
        static const uint64_t R[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
        unsigned __int128 br = 0;
        for (int k = 0; k < 4; k++) { const unsigned __int128 x = (unsigned __int128)R[k] - acc[k] - (uint64_t)br; acc[k] = (uint64_t)x; br = (x >> 64) & 1; }
    }
    memcpy(out, acc, 32); return true;
}
// a JSON number token that is an integer (no fraction, no exponent other than one that leaves an integer: kept simple -- digits only) -> true
inline bool flatten_input_value(const json::Value& v, std::vector<const json::Value*>& out, int depth = 0) {
    if (v.type == json::Value::Array) { if (depth > 8) return false; for (auto& x : v.a) if (!flatten_input_value(x, out, depth + 1)) return false; return true; }
    out.push_back(&v); return true;
}
// returns ZKC_OK-style 0 on success; 1 = malformed JSON / not an object, 2 = a circuit-input error (err = circom_runtime's message)
inline int circuit_inputs_from_json(const char* text, size_t len, int nLevels, uint8_t* out, std::string& err) {
    json::Value doc; std::string perr;
    if (!json::parse(text, len, doc, perr)) { err = perr; return 1; }
    if (doc.type != json::Value::Object) { err = "JSON: the circuit inputs must be an object"; return 1; }
    const char* const* names = circuit_input_names();
    const size_t nsib = (size_t)nLevels + 1, total = 12 + 2 * nsib;
    size_t size_of[12], off_of[12]; { size_t o = 0; for (int k = 0; k < 12; k++) { size_of[k] = (k == 0 || k == 3) ? 2 : k >= 10 ? nsib : 1; off_of[k] = o; o += size_of[k]; } }
    bool seen[12] = {false}; size_t set = 0;
    memset(out, 0, 32 * total);
    for (auto& kv : doc.o) {
        int k = -1; for (int i = 0; i < 12; i++) if (kv.first == names[i]) k = i;
        if (k < 0) { err = "Signal " + kv.first + " not found\n"; return 2; }
        std::vector<const json::Value*> vals;
        if (!flatten_input_value(kv.second, vals)) { err = "Too many values for input signal " + kv.first + "\n"; return 2; }
        if (vals.size() > size_of[k]) { err = "Too many values for input signal " + kv.first + "\n"; return 2; }
        if (vals.size() < size_of[k] && k < 10) { err = "Not enough values for input signal " + kv.first + "\n"; return 2; }
        if (seen[k]) memset(out + 32 * off_of[k], 0, 32 * size_of[k]);         // a repeated name: the last one stands (JSON.parse)
        for (size_t i = 0; i < vals.size(); i++) {
            const json::Value& x = *vals[i]; bool ok;
            if (x.type == json::Value::String) ok = integer_mod_r(x.s, out + 32 * (off_of[k] + i));
            else if (x.type == json::Value::Number) { ok = x.s.find_first_of(".eE") == std::string::npos && integer_mod_r(x.s, out + 32 * (off_of[k] + i)); }
            else if (x.type == json::Value::Bool) { memset(out + 32 * (off_of[k] + i), 0, 32); out[32 * (off_of[k] + i)] = x.b ? 1 : 0; ok = true; }      // BigInt(true) = 1n
            else ok = false;
            if (!ok) { err = "Cannot convert " + (x.type == json::Value::Null ? std::string("null") : x.type == json::Value::Object ? std::string("[object Object]") : x.s) + " to a BigInt"; return 2; }
        }
        if (!seen[k]) { seen[k] = true; set += size_of[k]; }
    }
    if (set < total) { err = "Not all inputs have been set. Only " + std::to_string(set) + " out of " + std::to_string(total); return 2; }
    return 0;
}

// ---- SHA-256 (FIPS 180-4) ----
// With the SHA extensions of the host CPU (detected once through cpuid) a block takes ~40 cycles instead of ~600: the per-call key fingerprint drops
// from 0.6 ms to 0.05 ms and the SHA-256 of a whole 55 MB key image from 0.27 s to 0.03 s.  Same digests either way (tests/test_host_abi_cpu.py).
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define ZKC_SHA_NI 1
}}  // leave zkc::parse for the system headers
#include <immintrin.h>
#include <cpuid.h>
namespace zkc { namespace parse {
inline bool sha_ni_available() {
    static const bool ok = [] {
        unsigned a, b, c, d;
        if (!__get_cpuid(1, &a, &b, &c, &d) || !(c & (1u << 19)) || !(c & (1u << 9))) return false;       // SSE4.1, SSSE3
        if (!__get_cpuid_count(7, 0, &a, &b, &c, &d)) return false;
        return (b & (1u << 29)) != 0 && getenv("ZKC_NO_SHA_NI") == nullptr;                                // SHA
    }();
    return ok;
}
// state h[8] += compression of n consecutive 64-byte blocks at p
__attribute__((target("sha,sse4.1,ssse3"))) inline void sha256_blocks_ni(uint32_t h[8], const uint8_t* p, size_t n) {
    alignas(16) static const uint32_t K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u,
        0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
        0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u,
        0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
        0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
        0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bll, 0x0405060700010203ll);
    // the instruction wants the state as (A B E F) and (C D G H), high lane first
    __m128i t = _mm_loadu_si128((const __m128i*)h), s1 = _mm_loadu_si128((const __m128i*)(h + 4));
    t = _mm_shuffle_epi32(t, 0xb1);                      // C D A B -> ... (words D C B A in memory order a b c d)
    s1 = _mm_shuffle_epi32(s1, 0x1b);
    __m128i s0 = _mm_alignr_epi8(t, s1, 8);              // A B E F
    s1 = _mm_blend_epi16(s1, t, 0xf0);                   // C D G H
    for (; n; n--, p += 64) {
        const __m128i save0 = s0, save1 = s1;
        __m128i m[4];
        for (int i = 0; i < 4; i++) m[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 16 * i)), bswap);
        for (int r = 0; r < 16; r++) {                   // four rounds per step
            __m128i w = m[r & 3];
            __m128i wk = _mm_add_epi32(w, _mm_load_si128((const __m128i*)(K + 4 * r)));
            s1 = _mm_sha256rnds2_epu32(s1, s0, wk);
            s0 = _mm_sha256rnds2_epu32(s0, s1, _mm_shuffle_epi32(wk, 0x0e));
            if (r < 12) {                                // schedule words 16 + 4r .. 19 + 4r into m[r & 3]
                __m128i x = _mm_sha256msg1_epu32(m[r & 3], m[(r + 1) & 3]);
                x = _mm_add_epi32(x, _mm_alignr_epi8(m[(r + 3) & 3], m[(r + 2) & 3], 4));
                m[r & 3] = _mm_sha256msg2_epu32(x, m[(r + 3) & 3]);
            }
        }
        s0 = _mm_add_epi32(s0, save0); s1 = _mm_add_epi32(s1, save1);
    }
    t = _mm_shuffle_epi32(s0, 0x1b);                     // F E B A
    s1 = _mm_shuffle_epi32(s1, 0xb1);                    // D C H G
    s0 = _mm_blend_epi16(t, s1, 0xf0);                   // D C B A -> memory a b c d
    s1 = _mm_alignr_epi8(s1, t, 8);                      // H G F E -> memory e f g h
    _mm_storeu_si128((__m128i*)h, s0); _mm_storeu_si128((__m128i*)(h + 4), s1);
}
#endif
struct Sha256 {
    uint32_t h[8]; uint8_t buf[64]; uint64_t total = 0; size_t fill = 0; bool ni = true;      // ni = false: portable rounds only (tests compare the two)
    explicit Sha256(bool allow_ni = true) : ni(allow_ni) { static const uint32_t iv[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u}; memcpy(h, iv, 32); }
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u,
            0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
            0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u,
            0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
            0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
            0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
        for (int i = 16; i < 64; i++) { const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10); w[i] = w[i - 16] + s0 + w[i - 7] + s1; }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void block1(const uint8_t* p) {
#ifdef ZKC_SHA_NI
        if (ni && sha_ni_available()) { sha256_blocks_ni(h, p, 1); return; }
#endif
        block(p);
    }
    void update(const void* data, size_t n) {
        const uint8_t* p = (const uint8_t*)data; total += n;
        if (fill) { const size_t k = n < 64 - fill ? n : 64 - fill; memcpy(buf + fill, p, k); fill += k; p += k; n -= k; if (fill == 64) { block1(buf); fill = 0; } }
#ifdef ZKC_SHA_NI
        if (ni && n >= 64 && sha_ni_available()) { const size_t nb = n / 64; sha256_blocks_ni(h, p, nb); p += 64 * nb; n -= 64 * nb; }
#endif
        for (; n >= 64; p += 64, n -= 64) block(p);
        if (n) { memcpy(buf, p, n); fill = n; }
    }
    void final(uint8_t out[32]) {
        const uint64_t bits = total * 8; uint8_t pad[72] = {0x80}; const size_t padlen = (fill < 56 ? 56 : 120) - fill;
        uint8_t lenb[8]; for (int i = 0; i < 8; i++) lenb[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(pad, padlen); update(lenb, 8);
        for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
    }
};
inline void sha256(const void* data, size_t n, uint8_t out[32]) { Sha256 s; s.update(data, n); s.final(out); }
// Identity of a .zkey image for the resident-key caches, cheap enough to take on every call (a 55 MB key hashes in 0.27 s, a proof takes 6 ms):
// SHA-256 over the file length, the whole header section (it holds alpha, beta, gamma, delta of the ceremony: 660 bytes that differ between any two
// real keys), the IC section, the first and last 4 KB of every other section and one 64-byte block out of every 64 KB of the file.  Two images
// with equal fingerprints differ only if someone crafted them to; the full SHA-256 (what circuits-info.md publishes) is taken once, at load.
inline void zkey_fingerprint(const uint8_t* buf, size_t len, const BinSections& bs, uint8_t out[32]) {
    Sha256 s; const uint64_t l64 = len; s.update(&l64, 8);
    for (int i = 1; i < 16; i++) if (bs.sec[i]) {
        s.update(&bs.ssz[i], 8);
        if (i == 2 || i == 3 || bs.ssz[i] <= 8192) s.update(bs.sec[i], (size_t)bs.ssz[i]);
        else { s.update(bs.sec[i], 4096); s.update(bs.sec[i] + bs.ssz[i] - 4096, 4096); }
    }
    for (size_t off = 0; off + 64 <= len; off += 65536) s.update(buf + off, 64);
    s.final(out);
}
inline std::string hex_of(const uint8_t* p, size_t n) { static const char* d = "0123456789abcdef"; std::string s; for (size_t i = 0; i < n; i++) { s.push_back(d[p[i] >> 4]); s.push_back(d[p[i] & 15]); } return s; }

}}  // namespace zkc::parse
