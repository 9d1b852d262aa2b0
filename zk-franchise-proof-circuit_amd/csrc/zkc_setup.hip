// zkc_setup.hip -- TEST-ONLY Groth16 trusted setup with KNOWN toxic waste (product host code, no GPU work).
//
// Stand-in for the reference's ceremony `snarkjs groth16 setup / zkey contribute / beacon`
// (circuit/circuit-compiler.sh:99-136), needed because proving_key.zkey is a missing blob and unreproducible.
// Reads an iden3 .r1cs (written by r1cs.py), derives tau, alpha, beta, gamma, delta from a seed and writes a
// snarkjs-format Groth16 .zkey plus verification_key.json.  NOT for production keys: whoever knows
// the seed can forge proofs.  Knowing the waste also gives tests an exponent-space closed form for every MSM.
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <functional>
#include "zkc_curve.h"
#include "zkc_fixedbase.h"
#include "../../include/zkcensus.h"

#include "zkc_hostparse.h"
extern "C" int zkc_pairing_bin(const uint8_t g1[64], const uint8_t g2[128], uint8_t out[384]);
using namespace zkc;

namespace {

const uint32_t G2X0[8] = {0xd992f6edu, 0x46debd5cu, 0xf75edaddu, 0x674322d4u, 0x5e5c4479u, 0x426a0066u, 0x121f1e76u, 0x1800deefu};
const uint32_t G2X1[8] = {0xaef312c2u, 0x97e485b7u, 0x35a9e712u, 0xf1aa4933u, 0x31fb5d25u, 0x7260bfb7u, 0x920d483au, 0x198e9393u};
const uint32_t G2Y0[8] = {0x66fa7daau, 0x4ce6cc01u, 0x0c43d37bu, 0xe3d1e769u, 0x8dcb408fu, 0x4aab7180u, 0xdb8c6debu, 0x12c85ea5u};
const uint32_t G2Y1[8] = {0xd122975bu, 0x55acdadcu, 0x70b38ef3u, 0xbc4b3133u, 0x690c3395u, 0xec9e99adu, 0x585ff075u, 0x090689d0u};

struct Rng {   // splitmix64
    uint64_t s;
    uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
    Fr fr() {      // uniform-ish non-zero element: 253 random bits (< r)
        uint32_t v[8];
        for (int i = 0; i < 4; i++) { uint64_t x = next(); v[2 * i] = (uint32_t)x; v[2 * i + 1] = (uint32_t)(x >> 32); }
        v[7] &= 0x1fffffffu; if (!(v[0] | v[1])) v[0] = 1;
        return fp_from_std<FrParams>(v);
    }
};

Fr fr_pow(Fr a, uint64_t e) { Fr r = Fr::one(); while (e) { if (e & 1) r = r * a; a = a * a; e >>= 1; } return r; }
Fr fr_root_of_unity(int logn) {          // 5^((r-1)/2^logn), ffjavascript's Fr.w[logn]
    uint32_t e[8]; for (int i = 0; i < 8; i++) e[i] = FrParams::p[i]; e[0] -= 1;
    for (int i = 0; i < 8; i++) e[i] = (e[i] >> 28) | (i < 7 ? e[i + 1] << 4 : 0);
    Fr g = fp_from_u32<FrParams>(5), w = Fr::one();
    for (int i = 255; i >= 0; i--) { w = w * w; if ((e[i >> 5] >> (i & 31)) & 1) w = w * g; }
    for (int i = 28; i > logn; i--) w = w * w;
    return w;
}
void batch_inverse(std::vector<Fr>& v) {  // in place; zeros stay zero
    std::vector<Fr> pre(v.size()); Fr acc = Fr::one();
    for (size_t i = 0; i < v.size(); i++) { pre[i] = acc; if (!v[i].is_zero()) acc = acc * v[i]; }
    Fr ai = fp_inv<FrParams>(acc);
    for (size_t i = v.size(); i-- > 0;) { if (v[i].is_zero()) continue; Fr t = ai * pre[i]; ai = ai * v[i]; v[i] = t; }
}

struct Term { uint32_t wire; Fr coef; };
struct Cons { std::vector<Term> a, b, c; };

uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
void put32(std::vector<uint8_t>& o, uint32_t v) { uint8_t b[4]; memcpy(b, &v, 4); o.insert(o.end(), b, b + 4); }
void put_raw(std::vector<uint8_t>& o, const void* p, size_t n) { o.insert(o.end(), (const uint8_t*)p, (const uint8_t*)p + n); }
void put_fq(std::vector<uint8_t>& o, const Fq& a) { put_raw(o, a.v, 32); }                     // Montgomery, as .zkey stores points
void put_g1(std::vector<uint8_t>& o, const G1Affine& p) { put_fq(o, p.x); put_fq(o, p.y); }
void put_g2(std::vector<uint8_t>& o, const G2Affine& p) { put_fq(o, p.x.c0); put_fq(o, p.x.c1); put_fq(o, p.y.c0); put_fq(o, p.y.c1); }

std::string dec(const uint32_t s_in[8]) {
    uint32_t s[8]; memcpy(s, s_in, 32); std::string out;
    bool nz = true;
    while (nz) {
        uint64_t rem = 0; nz = false;
        for (int i = 7; i >= 0; i--) { uint64_t cur = (rem << 32) | s[i]; s[i] = (uint32_t)(cur / 10); rem = cur % 10; if (s[i]) nz = true; }
        out.push_back((char)('0' + rem));
    }
    return std::string(out.rbegin(), out.rend());
}
std::string dec_fq(const Fq& a) { uint32_t s[8]; fp_to_std<FqParams>(s, a); return dec(s); }
std::string json_g1(const G1Affine& p) { return "[\n  \"" + dec_fq(p.x) + "\",\n  \"" + dec_fq(p.y) + "\",\n  \"1\"\n ]"; }
std::string json_g2(const G2Affine& p) {
    return "[\n  [\n   \"" + dec_fq(p.x.c0) + "\",\n   \"" + dec_fq(p.x.c1) + "\"\n  ],\n  [\n   \"" + dec_fq(p.y.c0) + "\",\n   \"" + dec_fq(p.y.c1) +
           "\"\n  ],\n  [\n   \"1\",\n   \"0\"\n  ]\n ]";
}

int fail(char* err, size_t errlen, const std::string& m) { if (err && errlen) { snprintf(err, errlen, "%s", m.c_str()); } return ZKC_ERR_FORMAT; }

}  // namespace

extern "C" int zkc_setup_from_r1cs(const char* r1cs_path, uint64_t seed, const char* zkey_path, const char* vkey_json_path,
                                   char* err, size_t errlen) {
    // ---- read .r1cs ----
    FILE* f = fopen(r1cs_path, "rb"); if (!f) return fail(err, errlen, std::string("cannot open ") + r1cs_path);
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf((size_t)sz); if (fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) { fclose(f); return fail(err, errlen, "short read"); } fclose(f);
    if (sz < 12 || memcmp(buf.data(), "r1cs", 4)) return fail(err, errlen, "not an r1cs file");
    uint32_t nsec = rd32(&buf[8]); size_t p = 12; const uint8_t *s1 = nullptr, *s2 = nullptr; uint64_t s2sz = 0;
    for (uint32_t i = 0; i < nsec; i++) { uint32_t id = rd32(&buf[p]); uint64_t n = rd64(&buf[p + 4]); p += 12; if (id == 1) s1 = &buf[p]; if (id == 2) { s2 = &buf[p]; s2sz = n; } p += n; }
    if (!s1 || !s2 || rd32(s1) != 32) return fail(err, errlen, "bad r1cs header");
    for (int i = 0; i < 8; i++) if (rd32(s1 + 4 + 4 * i) != FrParams::p[i]) return fail(err, errlen, "r1cs prime is not BN254 r");
    const uint32_t nWires = rd32(s1 + 36), nPubOut = rd32(s1 + 40), nPubIn = rd32(s1 + 44), nCons = rd32(s1 + 60);
    const uint32_t nPub = nPubOut + nPubIn;
    std::vector<Cons> cons(nCons);
    {
        const uint8_t* q = s2; const uint8_t* end = s2 + s2sz;
        for (uint32_t k = 0; k < nCons; k++) {
            std::vector<Term>* v[3] = {&cons[k].a, &cons[k].b, &cons[k].c};
            for (int m = 0; m < 3; m++) {
                if (q + 4 > end) return fail(err, errlen, "r1cs constraints truncated");
                uint32_t n = rd32(q); q += 4; v[m]->resize(n);
                for (uint32_t t = 0; t < n; t++) { uint32_t s[8]; (*v[m])[t].wire = rd32(q); memcpy(s, q + 4, 32); (*v[m])[t].coef = fp_from_std<FrParams>(s); q += 36; }
            }
        }
    }
    uint32_t logn = 0; while ((1u << logn) < nCons + nPub + 1) logn++;
    const uint32_t n = 1u << logn;
    // ---- toxic waste ----
    Rng rng{seed};
    const Fr tau = rng.fr(), alpha = rng.fr(), beta = rng.fr(), gamma = rng.fr(), delta = rng.fr();
    const Fr w = fr_root_of_unity((int)logn), g = fr_root_of_unity((int)logn + 1);
    const Fr ninv = fp_inv<FrParams>(fp_from_u32<FrParams>(n));
    const Fr tn = fr_pow(tau, n), zt = tn - Fr::one();                     // Z(tau) = tau^n - 1
    // Lagrange basis at tau over H: L_c = Z(tau) w^c / (n (tau - w^c));  over the odd coset gH: L'_c = (-tau^n - 1) w^c / (n (tau/g - w^c))
    std::vector<Fr> wp(n), lag(n), lagc(n);
    wp[0] = Fr::one(); for (uint32_t i = 1; i < n; i++) wp[i] = wp[i - 1] * w;
    const Fr tg = tau * fp_inv<FrParams>(g);
    for (uint32_t i = 0; i < n; i++) { lag[i] = tau - wp[i]; lagc[i] = tg - wp[i]; }
    batch_inverse(lag); batch_inverse(lagc);
    const Fr zc = Fr::zero() - tn - Fr::one();
    for (uint32_t i = 0; i < n; i++) { lag[i] = lag[i] * wp[i] * zt * ninv; lagc[i] = lagc[i] * wp[i] * zc * ninv; }
    // ---- QAP polynomials at tau ----
    std::vector<Fr> u(nWires, Fr::zero()), v(nWires, Fr::zero()), ww(nWires, Fr::zero());
    for (uint32_t k = 0; k < nCons; k++) {
        for (auto& t : cons[k].a) u[t.wire] = u[t.wire] + t.coef * lag[k];
        for (auto& t : cons[k].b) v[t.wire] = v[t.wire] + t.coef * lag[k];
        for (auto& t : cons[k].c) ww[t.wire] = ww[t.wire] + t.coef * lag[k];
    }
    for (uint32_t i = 0; i <= nPub; i++) u[i] = u[i] + lag[nCons + i];     // snarkjs' extra rows A[nCons+i][i] = 1
    // ---- points ----
    G1Affine G1{Fq::one(), fp_from_u32<FqParams>(2)};
    G2Affine G2{{fp_from_std<FqParams>(G2X0), fp_from_std<FqParams>(G2X1)}, {fp_from_std<FqParams>(G2Y0), fp_from_std<FqParams>(G2Y1)}};
    FixedBase<Fq> fb1(G1); FixedBase<Fq2> fb2(G2);
    const Fr dinv = fp_inv<FrParams>(delta), ginv = fp_inv<FrParams>(gamma);
    std::vector<G1Affine> pA(nWires), pB1(nWires), pC(nWires), pH(n); std::vector<G2Affine> pB2(nWires);
    parallel_for(nWires, [&](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            pA[i] = fb1.mul(u[i]); pB1[i] = fb1.mul(v[i]); pB2[i] = fb2.mul(v[i]);
            Fr k = (beta * u[i] + alpha * v[i] + ww[i]) * (i <= nPub ? ginv : dinv);
            pC[i] = fb1.mul(k);
        }
    });
    const Fr hk = zt * dinv * fp_inv<FrParams>(Fr::zero() - fp_from_u32<FrParams>(2));   // Z(tau) / (-2 delta)
    parallel_for(n, [&](size_t a, size_t b) { for (size_t i = a; i < b; i++) pH[i] = fb1.mul(lagc[i] * hk); });
    const G1Affine alpha1 = fb1.mul(alpha), beta1 = fb1.mul(beta), delta1 = fb1.mul(delta);
    const G2Affine beta2 = fb2.mul(beta), gamma2 = fb2.mul(gamma), delta2 = fb2.mul(delta);
    // ---- .zkey ----
    std::vector<std::vector<uint8_t>> sec(11);
    put32(sec[1], 1);
    put32(sec[2], 32); put_raw(sec[2], FqParams::p, 32); put32(sec[2], 32); put_raw(sec[2], FrParams::p, 32);
    put32(sec[2], nWires); put32(sec[2], nPub); put32(sec[2], n);
    put_g1(sec[2], alpha1); put_g1(sec[2], beta1); put_g2(sec[2], beta2); put_g2(sec[2], gamma2); put_g1(sec[2], delta1); put_g2(sec[2], delta2);
    for (uint32_t i = 0; i <= nPub; i++) put_g1(sec[3], pC[i]);
    {
        uint32_t ncoef = nPub + 1; for (auto& c : cons) ncoef += (uint32_t)(c.a.size() + c.b.size());
        put32(sec[4], ncoef);
        Fr r2; for (int i = 0; i < 8; i++) r2.v[i] = FrParams::r2[i];
        auto put_coef = [&](uint32_t m, uint32_t c, uint32_t s, const Fr& val) { put32(sec[4], m); put32(sec[4], c); put32(sec[4], s); Fr dm = val * r2; put_raw(sec[4], dm.v, 32); };
        for (uint32_t k = 0; k < nCons; k++) { for (auto& t : cons[k].a) put_coef(0, k, t.wire, t.coef); for (auto& t : cons[k].b) put_coef(1, k, t.wire, t.coef); }
        for (uint32_t i = 0; i <= nPub; i++) put_coef(0, nCons + i, i, Fr::one());
    }
    for (uint32_t i = 0; i < nWires; i++) { put_g1(sec[5], pA[i]); put_g1(sec[6], pB1[i]); put_g2(sec[7], pB2[i]); }
    for (uint32_t i = nPub + 1; i < nWires; i++) put_g1(sec[8], pC[i]);
    for (uint32_t i = 0; i < n; i++) put_g1(sec[9], pH[i]);
    sec[10].assign(64, 0); put32(sec[10], 0);                              // circuit hash (unused here), 0 contributions
    FILE* o = fopen(zkey_path, "wb"); if (!o) return fail(err, errlen, std::string("cannot write ") + zkey_path);
    fwrite("zkey", 1, 4, o); uint32_t ver = 1, ns = 10; fwrite(&ver, 4, 1, o); fwrite(&ns, 4, 1, o);
    for (uint32_t id = 1; id <= 10; id++) { uint64_t len = sec[id].size(); fwrite(&id, 4, 1, o); fwrite(&len, 8, 1, o); fwrite(sec[id].data(), 1, len, o); }
    fclose(o);
    // ---- verification_key.json (members and order of artifacts/zkCensus/dev/160/verification_key.json, vk_alphabeta_12 = e(alpha1, beta2)
    //      as snarkjs' `zkey export verificationkey` prints it, circuit/circuit-compiler.sh:133-134) ----
    if (vkey_json_path) {
        std::string j = "{\n \"protocol\": \"groth16\",\n \"curve\": \"bn128\",\n \"nPublic\": " + std::to_string(nPub) + ",\n";
        j += " \"vk_alpha_1\": " + json_g1(alpha1) + ",\n \"vk_beta_2\": " + json_g2(beta2) + ",\n \"vk_gamma_2\": " + json_g2(gamma2) + ",\n \"vk_delta_2\": " + json_g2(delta2) + ",\n";
        {
            uint8_t a[64], b[128], e[384]; uint32_t t[8];
            fp_to_std<FqParams>(t, alpha1.x); memcpy(a, t, 32); fp_to_std<FqParams>(t, alpha1.y); memcpy(a + 32, t, 32);
            const Fq* bc[4] = {&beta2.x.c0, &beta2.x.c1, &beta2.y.c0, &beta2.y.c1};
            for (int i = 0; i < 4; i++) { fp_to_std<FqParams>(t, *bc[i]); memcpy(b + 32 * i, t, 32); }
            if (zkc_pairing_bin(a, b, e) != ZKC_OK) return fail(err, errlen, "pairing e(alpha, beta) failed");
            j += " \"vk_alphabeta_12\": [\n";
            for (int h = 0; h < 2; h++) {
                j += "  [\n";
                for (int k = 0; k < 3; k++) j += "   [\"" + zkc::parse::dec_of(e + 64 * (3 * h + k)) + "\", \"" + zkc::parse::dec_of(e + 64 * (3 * h + k) + 32) + "\"]" + (k < 2 ? ",\n" : "\n");
                j += h == 0 ? "  ],\n" : "  ]\n";
            }
            j += " ],\n";
        }
        j += " \"IC\": [\n";
        for (uint32_t i = 0; i <= nPub; i++) j += "  " + json_g1(pC[i]) + (i < nPub ? ",\n" : "\n");
        j += " ]\n}\n";
        FILE* v = fopen(vkey_json_path, "wb"); if (!v) return fail(err, errlen, std::string("cannot write ") + vkey_json_path);
        fwrite(j.data(), 1, j.size(), v); fclose(v);
    }
    return ZKC_OK;
}
