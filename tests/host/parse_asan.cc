// parse_asan.cc -- sanitizer harness for the host-only parsers of libzkcensus (zkc_hostparse.h): built by
// tests/test_host_parsers_asan.py with g++ -fsanitize=address,undefined -fno-sanitize-recover=all and run on the CPU.
// It assembles a tiny but well-formed .zkey / .wtns / JSON triple, checks that they parse, then feeds the parsers every truncation,
// hostile length fields (2^64 - 1, sizes that wrap p + sz), n = 1 / 3 domains, out-of-range coefficients and some thousands of random byte
// mutations.  After each successful parse it touches every byte the loader would read (header points, sections 3-9), so an
// accepted-but-short file shows up as an ASan report.  Exit code 0 = no report and every hostile file rejected.
#include "../../zk-franchise-proof-circuit_amd/csrc/zkc_hostparse.h"
#include <cstdio>
#include <cstdlib>
using namespace zkc::parse;

static void put32(std::vector<uint8_t>& v, uint32_t x) { for (int i = 0; i < 4; i++) v.push_back((uint8_t)(x >> (8 * i))); }
static void put64(std::vector<uint8_t>& v, uint64_t x) { for (int i = 0; i < 8; i++) v.push_back((uint8_t)(x >> (8 * i))); }
static void putn(std::vector<uint8_t>& v, const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; v.insert(v.end(), b, b + n); }
static std::vector<uint8_t> binfile(const char* magic, uint32_t ver, const std::vector<std::pair<uint32_t, std::vector<uint8_t>>>& secs) {
    std::vector<uint8_t> f; putn(f, magic, 4); put32(f, ver); put32(f, (uint32_t)secs.size());
    for (auto& s : secs) { put32(f, s.first); put64(f, s.second.size()); putn(f, s.second.data(), s.second.size()); }
    return f;
}
static std::vector<uint8_t> make_zkey(uint32_t nVars, uint32_t nPub, uint32_t n, uint32_t nCoeffs) {
    std::vector<std::pair<uint32_t, std::vector<uint8_t>>> s;
    std::vector<uint8_t> s1; put32(s1, 1); s.push_back({1, s1});
    std::vector<uint8_t> h; put32(h, 32); putn(h, kFqP, 32); put32(h, 32); putn(h, kFrP, 32); put32(h, nVars); put32(h, nPub); put32(h, n);
    h.resize(h.size() + 64 + 64 + 128 + 128 + 64 + 128, 7); s.push_back({2, h});
    s.push_back({3, std::vector<uint8_t>(64 * (size_t)(nPub + 1), 1)});
    std::vector<uint8_t> c; put32(c, nCoeffs);
    for (uint32_t i = 0; i < nCoeffs; i++) { put32(c, i & 1); put32(c, i % n); put32(c, i % nVars); c.resize(c.size() + 32, 9); }
    s.push_back({4, c});
    s.push_back({5, std::vector<uint8_t>(64 * (size_t)nVars, 2)}); s.push_back({6, std::vector<uint8_t>(64 * (size_t)nVars, 3)});
    s.push_back({7, std::vector<uint8_t>(128 * (size_t)nVars, 4)}); s.push_back({8, std::vector<uint8_t>(64 * (size_t)(nVars - nPub - 1), 5)});
    s.push_back({9, std::vector<uint8_t>(64 * (size_t)n, 6)}); s.push_back({10, std::vector<uint8_t>(68, 0)});
    return binfile("zkey", 1, s);
}
static volatile uint64_t sink;
static void touch(const uint8_t* p, uint64_t n) { uint64_t a = 0; for (uint64_t i = 0; i < n; i++) a += p[i]; sink += a; }
// what zkc_zkey_load does with a file the parser accepted
static bool load_like(const std::vector<uint8_t>& f) {
    std::vector<uint8_t> copy(f);                       // exact-size heap block: any over-read is out of bounds for ASan
    BinSections bs; ZkeyHeader zh; std::string err;
    if (!binfile_sections(copy.data(), copy.size(), "zkey", 1, bs, err) || !zkey_check(bs, zh, err)) return false;
    touch(bs.sec[2], 660);
    touch(bs.sec[3], 64ull * (zh.nPub + 1)); touch(bs.sec[4], 4 + 44ull * zh.nCoeffs);
    touch(bs.sec[5], 64ull * zh.nVars); touch(bs.sec[6], 64ull * zh.nVars); touch(bs.sec[7], 128ull * zh.nVars);
    touch(bs.sec[8], 64ull * (zh.nVars - zh.nPub - 1)); touch(bs.sec[9], 64ull * zh.n);
    std::vector<uint32_t> half(zh.n / 2); half[0] = 1;  // the twiddle table the loader fills: n >= 4 guaranteed by the parser
    uint8_t d[32]; sha256(copy.data(), copy.size(), d); sink += d[0];
    zkey_fingerprint(copy.data(), copy.size(), bs, d); sink += d[1];
    return true;
}
static bool wtns_like(const std::vector<uint8_t>& f) {
    std::vector<uint8_t> copy(f); const uint8_t* pl; uint32_t nw;
    if (!wtns_view(copy.data(), copy.size(), &pl, &nw)) return false;
    touch(pl, 32ull * nw); return true;
}
static uint64_t rng_s = 0x9e3779b97f4a7c15ull;
static uint64_t rnd() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return rng_s; }
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main() {
    const std::vector<uint8_t> good = make_zkey(6, 2, 8, 11);
    CHECK(load_like(good));
    // every truncation is rejected (or, when it still parses, read in bounds)
    for (size_t n = 0; n < good.size(); n++) { std::vector<uint8_t> t(good.begin(), good.begin() + n); CHECK(!load_like(t)); }
    // hostile section sizes: all ones, and values that wrap p + sz
    for (uint64_t sz : {~0ull, ~0ull - 11, ~0ull - 23, 1ull << 63, (unsigned long long)good.size()}) {
        std::vector<uint8_t> t(good); memcpy(t.data() + 16, &sz, 8); CHECK(!load_like(t));
        for (size_t at = 16; at + 8 <= t.size(); at += 97) { std::vector<uint8_t> u(good); memcpy(u.data() + at, &sz, 8); (void)load_like(u); }
    }
    CHECK(!load_like(make_zkey(6, 2, 1, 0))); CHECK(!load_like(make_zkey(6, 2, 2, 0))); CHECK(!load_like(make_zkey(6, 2, 3, 0))); CHECK(!load_like(make_zkey(6, 2, 12, 3)));
    CHECK(load_like(make_zkey(6, 2, 4, 0))); CHECK(load_like(make_zkey(3, 2, 4, 5)));
    { std::vector<uint8_t> t = make_zkey(6, 2, 8, 0); CHECK(load_like(t)); }
    { // nPub >= nVars, nVars = 0: header fields patched in place (section 2 starts after sec 1: 12 + 12 + 4 + 12)
        const size_t h = 12 + 12 + 4 + 12;
        std::vector<uint8_t> t(good); uint32_t v = 0; memcpy(t.data() + h + 72, &v, 4); CHECK(!load_like(t));
        t = good; v = 6; memcpy(t.data() + h + 76, &v, 4); CHECK(!load_like(t));
        t = good; v = 0xffffffffu; memcpy(t.data() + h + 76, &v, 4); CHECK(!load_like(t));
        t = good; v = 0x80000000u; memcpy(t.data() + h + 80, &v, 4); CHECK(!load_like(t));
    }
    { // coefficient pointing outside the matrix / the witness; nCoeffs that does not match the section
        BinSections bs; std::string err; std::vector<uint8_t> t(good); CHECK(binfile_sections(t.data(), t.size(), "zkey", 1, bs, err));
        const size_t c0 = (size_t)(bs.sec[4] - t.data());
        uint32_t v = 2; memcpy(t.data() + c0 + 4, &v, 4); CHECK(!load_like(t));
        t = good; v = 8; memcpy(t.data() + c0 + 8, &v, 4); CHECK(!load_like(t));
        t = good; v = 6; memcpy(t.data() + c0 + 12, &v, 4); CHECK(!load_like(t));
        t = good; v = 0x5d1745d2u; memcpy(t.data() + c0, &v, 4); CHECK(!load_like(t));       // 4 + 44 * nCoeffs wraps 32 bits
        t = good; memset(t.data() + c0 + 4 + 12, 0xff, 32); CHECK(!load_like(t));              // a coefficient value that is not a reduced field element
        t = good; memcpy(t.data() + c0 + 4 + 12, kFrP, 32); CHECK(!load_like(t));               // ... exactly r
        { t = good; uint32_t rm1[8]; memcpy(rm1, kFrP, 32); rm1[0] -= 1; memcpy(t.data() + c0 + 4 + 12, rm1, 32); CHECK(load_like(t)); }      // r - 1 is one
    }
    for (int it = 0; it < 20000; it++) {                 // random mutations: accepted or not, never out of bounds
        std::vector<uint8_t> t(good); const int k = 1 + (int)(rnd() % 4);
        for (int j = 0; j < k; j++) t[rnd() % (it % 3 ? 200 : t.size())] = (uint8_t)rnd();
        (void)load_like(t);
    }
    // ---- .wtns ----
    std::vector<uint8_t> w1; put32(w1, 32); putn(w1, kFrP, 32); put32(w1, 5);
    const std::vector<uint8_t> wt = binfile("wtns", 2, {{1, w1}, {2, std::vector<uint8_t>(160, 1)}});
    CHECK(wtns_like(wt));
    for (size_t n = 0; n < wt.size(); n++) { std::vector<uint8_t> t(wt.begin(), wt.begin() + n); CHECK(!wtns_like(t)); }
    { std::vector<uint8_t> s1(8, 0); CHECK(!wtns_like(binfile("wtns", 2, {{1, s1}, {2, std::vector<uint8_t>(160, 1)}}))); }   // section 1 shorter than 40 bytes
    for (uint64_t sz : {~0ull, ~0ull - 11, 1ull << 63}) { std::vector<uint8_t> t(wt); memcpy(t.data() + 16, &sz, 8); CHECK(!wtns_like(t)); }
    for (int it = 0; it < 20000; it++) { std::vector<uint8_t> t(wt); t[rnd() % t.size()] = (uint8_t)rnd(); t[rnd() % 80] = (uint8_t)rnd(); (void)wtns_like(t); }
    // ---- JSON shapes ----
    const std::string vk = "{\"vk_alpha_1\":[\"1\",\"2\",\"1\"],\"vk_beta_2\":[[\"1\",\"2\"],[\"3\",\"4\"],[\"1\",\"0\"]],\"vk_gamma_2\":[[\"1\",\"2\"],[\"3\",\"4\"],[\"1\",\"0\"]],"
                           "\"vk_delta_2\":[[\"1\",\"2\"],[\"3\",\"4\"],[\"1\",\"0\"]],\"IC\":[[\"1\",\"2\",\"1\"],[\"1\",\"2\",\"1\"]]}";
    const std::string pub = "[\"5\"]", pr = "{\"pi_a\":[\"1\",\"2\",\"1\"],\"pi_b\":[[\"1\",\"2\"],[\"3\",\"4\"],[\"1\",\"0\"]],\"pi_c\":[\"1\",\"2\",\"1\"]}";
    std::vector<uint8_t> a, b, c; int np; std::string err;
    CHECK(verify_inputs_from_json(vk, pub, pr, a, b, c, np, err) == 1 && np == 1);
    { std::string bad = pr; bad.replace(bad.find("\"1\"]"), 3, "\"2\""); CHECK(verify_inputs_from_json(vk, pub, bad, a, b, c, np, err) == 0); }       // z = 2 is not affine
    CHECK(verify_inputs_from_json(vk, "[\"5\",\"6\"]", pr, a, b, c, np, err) == -1);                                                                 // IC length mismatch
    CHECK(verify_inputs_from_json(vk, "[\"" + std::string(90, '9') + "\"]", pr, a, b, c, np, err) == 0);                                              // >= 2^256
    for (size_t n = 0; n < vk.size(); n += 3) (void)verify_inputs_from_json(vk.substr(0, n), pub, pr, a, b, c, np, err);
    for (size_t n = 0; n < pr.size(); n++) (void)verify_inputs_from_json(vk, pub, pr.substr(0, n), a, b, c, np, err);
    for (int it = 0; it < 20000; it++) {
        std::string v2 = vk, p2 = pr, s2 = pub;
        v2[rnd() % v2.size()] = (char)rnd(); p2[rnd() % p2.size()] = (char)rnd(); if (it & 1) s2[rnd() % s2.size()] = (char)rnd();
        (void)verify_inputs_from_json(v2, s2, p2, a, b, c, np, err);
    }
    // the two halves on their own (zkc_proof_from_json / zkc_vkey_from_json at the C ABI): codes, and the same truncation sweep
    { int n2 = 0, nic = 0; CHECK(proof_from_json(pub, pr, b, c, n2, err) == 1 && n2 == 1 && c.size() == 256); CHECK(vkey_from_json(vk, a, nic, err) == 1 && nic == 2 && a.size() == 448 + 128);
      CHECK(proof_from_json("[5]", pr, b, c, n2, err) == -1); CHECK(proof_from_json(pub, "[]", b, c, n2, err) == -1); CHECK(vkey_from_json("[]", a, nic, err) == -1);
      for (size_t n = 0; n < pr.size(); n++) CHECK(proof_from_json(pub, pr.substr(0, n), b, c, n2, err) == -1);
      for (size_t n = 0; n < vk.size(); n++) CHECK(vkey_from_json(vk.substr(0, n), a, nic, err) == -1); }
    // ---- [r5] the strict JSON reader and the circuit-inputs reader (zkc_json.h, circuit_inputs_from_json): truncations, mutations, deep nesting, long numbers ----
    {
        std::string in = "{\"electionId\":[\"1\",\"2\"],\"nullifier\":\"3\",\"availableWeight\":\"4\",\"voteHash\":[\"5\",\"6\"],\"sikRoot\":\"7\",\"censusRoot\":\"8\",\"address\":\"0x9\",\"password\":-10,"
                         "\"signature\":\"21888242871839275222246405745257275088548364400416034343698204186575808495618\",\"voteWeight\":1,\"censusSiblings\":[\"1\",[\"2\",\"3\"]],\"sikSiblings\":[]}";
        std::vector<uint8_t> blk(32 * (12 + 2 * 11)); std::string e;
        CHECK(circuit_inputs_from_json(in.data(), in.size(), 10, blk.data(), e) == 0);
        CHECK(blk[32 * 8] == 9 && blk[32 * 10] == 1 && blk[32 * 11] == 1 && blk[32 * 12] == 1 && blk[32 * 14] == 3);      // hex, r + 1 = 1, number, nested list
        { uint32_t w[8]; memcpy(w, blk.data() + 32 * 9, 32); CHECK(w[0] == kFrP[0] - 10 && w[7] == kFrP[7]); }            // -10 = r - 10
        for (size_t n = 0; n < in.size(); n++) CHECK(circuit_inputs_from_json(in.data(), n, 10, blk.data(), e) != 0);
        for (int it = 0; it < 30000; it++) {
            std::string m = in; const int k = 1 + (int)(rnd() % 3);
            for (int j = 0; j < k; j++) { const size_t at = rnd() % m.size(); if (rnd() & 1) m[at] = (char)rnd(); else m.insert(at, 1, "[]{}\",:\\0-e."[rnd() % 13]); }
            (void)circuit_inputs_from_json(m.data(), m.size(), 10, blk.data(), e);
        }
        std::string deep(100000, '['); CHECK(circuit_inputs_from_json(deep.data(), deep.size(), 10, blk.data(), e) == 1);
        std::string longnum = "{\"voteWeight\":" + std::string(200000, '7') + "}"; CHECK(circuit_inputs_from_json(longnum.data(), longnum.size(), 10, blk.data(), e) == 2);      // parses; the other signals are missing
        zkc::json::Value v; std::string je;
        const char* okdoc = "{\"a\":[1,2.5e3,-0,true,false,null,\"\\u00e9\\ud83d\\ude00\\n\"]}";
        CHECK(zkc::json::parse(okdoc, strlen(okdoc), v, je) && v.o.size() == 1 && v.o[0].second.a.size() == 7 && v.o[0].second.a[6].s == "\xc3\xa9\xf0\x9f\x98\x80\n");
        const char* bad[] = {"", " ", "{", "}", "[1,]", "{\"a\":1,}", "{a:1}", "01", "1.", ".5", "-", "+1", "1e", "\"\\x\"", "\"\t\"", "nul", "tru", "[1 2]", "{\"a\" 1}", "[1]]", "NaN", "Infinity", "'a'", "\"\xff\"", "\"\xc0\x80\"", "\"\xed\xa0\x80\""};
        for (const char* b : bad) { zkc::json::Value x; CHECK(!zkc::json::parse(b, strlen(b), x, je)); }
    }
    { uint8_t d[32]; sha256("abc", 3, d); CHECK(hex_of(d, 32) == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad");
      std::string m(1000, 'a'); sha256(m.data(), m.size(), d); CHECK(hex_of(d, 32) == "41edece42d63e8d9bf515a9ba6932e1c20cbc9f5a5d134645adb5db1b9737ea3"); }
    // the SHA-extension rounds against the portable ones: every length around the block boundaries, split updates
    { std::vector<uint8_t> m(70000); uint32_t x = 12345; for (auto& b : m) { x = x * 1664525u + 1013904223u; b = (uint8_t)(x >> 24); }
      for (size_t len : {0ul, 1ul, 55ul, 56ul, 63ul, 64ul, 65ul, 119ul, 120ul, 127ul, 128ul, 129ul, 1000ul, 4096ul, 65536ul, 69999ul}) {
          uint8_t a[32], b[32]; Sha256 p(false); p.update(m.data(), len); p.final(a);
          Sha256 q(true); const size_t cut = len / 3; q.update(m.data(), cut); q.update(m.data() + cut, len - cut); q.final(b);
          CHECK(memcmp(a, b, 32) == 0); } }
    printf("host parsers: ok\n");
    return 0;
}
