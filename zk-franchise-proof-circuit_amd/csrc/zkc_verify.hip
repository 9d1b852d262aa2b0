// zkc_verify.hip -- a9: Groth16 verification on the CPU (constant work, no GPU), plus the JSON / .wtns codecs (a8) and the
// rapidsnark-shaped `groth16_prover` entry point (product host code).
//
// Mirrors go-rapidsnark/verifier VerifyGroth16 behind dvote's proof.Verify (zk_census_test.go:122) and snarkjs
// groth16.verify: vk_x = IC0 + sum s_i IC_{i+1};  e(-A, B) e(alpha, beta) e(vk_x, gamma) e(C, delta) == 1.
// The pairing is the plain ate pairing over T = 6x^2 with affine line functions and a straight final exponentiation:
// verification is a handful of milliseconds per proof and is not on the throughput path.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <random>
#include <thread>
#include <algorithm>
#include "zkc_prover.h"

using namespace zkc;

namespace {

// ---------------- Fq6 = Fq2[v]/(v^3 - xi), Fq12 = Fq6[w]/(w^2 - v), xi = 9 + u ----------------
Fq2 mul_xi(const Fq2& a) {
    Fq t0 = fp_dbl(fp_dbl(fp_dbl(a.c0))) + a.c0, t1 = fp_dbl(fp_dbl(fp_dbl(a.c1))) + a.c1;   // 9a
    return {t0 - a.c1, t1 + a.c0};
}
struct Fq6 { Fq2 a0, a1, a2; };
struct Fq12 { Fq6 a, b; };
Fq6 operator+(const Fq6& x, const Fq6& y) { return {x.a0 + y.a0, x.a1 + y.a1, x.a2 + y.a2}; }
Fq6 operator-(const Fq6& x, const Fq6& y) { return {x.a0 - y.a0, x.a1 - y.a1, x.a2 - y.a2}; }
Fq6 neg6(const Fq6& x) { return {fp_neg(x.a0), fp_neg(x.a1), fp_neg(x.a2)}; }
Fq6 operator*(const Fq6& x, const Fq6& y) {
    return {x.a0 * y.a0 + mul_xi(x.a1 * y.a2 + x.a2 * y.a1), x.a0 * y.a1 + x.a1 * y.a0 + mul_xi(x.a2 * y.a2), x.a0 * y.a2 + x.a1 * y.a1 + x.a2 * y.a0};
}
Fq6 mul_v(const Fq6& x) { return {mul_xi(x.a2), x.a0, x.a1}; }
Fq6 inv6(const Fq6& x) {
    Fq2 c0 = fp_sqr(x.a0) - mul_xi(x.a1 * x.a2), c1 = mul_xi(fp_sqr(x.a2)) - x.a0 * x.a1, c2 = fp_sqr(x.a1) - x.a0 * x.a2;
    Fq2 t = fp_inv(mul_xi(x.a2 * c1 + x.a1 * c2) + x.a0 * c0);
    return {c0 * t, c1 * t, c2 * t};
}
Fq12 one12() { Fq12 r{}; r.a.a0 = Fq2::one(); r.a.a1 = r.a.a2 = r.b.a0 = r.b.a1 = r.b.a2 = Fq2::zero(); return r; }
Fq12 operator*(const Fq12& x, const Fq12& y) { return {x.a * y.a + mul_v(x.b * y.b), x.a * y.b + x.b * y.a}; }
Fq12 conj12(const Fq12& x) { return {x.a, neg6(x.b)}; }
Fq12 inv12(const Fq12& x) { Fq6 t = inv6(x.a * x.a - mul_v(x.b * x.b)); return {x.a * t, neg6(x.b * t)}; }
bool is_one12(const Fq12& x) { Fq12 o = one12(); return memcmp(&x, &o, sizeof o) == 0; }

// line through twist points with slope lam evaluated at P in G1:  yP + (-lam xP) w + (lam xT - yT) w^3
Fq12 line_eval(const Fq2& lam, const Fq2& xT, const Fq2& yT, const G1Affine& P) {
    Fq12 l{}; l.a.a0 = {P.y, Fq::zero()}; l.a.a1 = l.a.a2 = l.b.a2 = Fq2::zero();
    l.b.a0 = fp_neg(lam * Fq2{P.x, Fq::zero()}); l.b.a1 = lam * xT - yT;
    return l;
}
Fq12 miller(const G1Affine& P, const G2Affine& Q) {
    static const uint64_t T[2] = {0xf83e9682e87cfd46ull, 0x6f4d8248eeb859fbull};   // 6 x^2, x = 4965661367192848881
    Fq12 f = one12();
    if (P.is_inf() || Q.is_inf()) return f;
    Fq2 xR = Q.x, yR = Q.y; bool rinf = false;
    for (int i = 125; i >= 0; i--) {
        f = f * f;
        if (!rinf) {
            Fq2 x2 = fp_sqr(xR), lam = (fp_dbl(x2) + x2) * fp_inv(fp_dbl(yR));
            f = f * line_eval(lam, xR, yR, P);
            Fq2 x3 = fp_sqr(lam) - fp_dbl(xR), y3 = lam * (xR - x3) - yR;
            xR = x3; yR = y3;
        }
        if ((T[i >> 6] >> (i & 63)) & 1) {
            if (rinf) { xR = Q.x; yR = Q.y; rinf = false; continue; }
            Fq2 dx = Q.x - xR;
            if (dx.is_zero()) { rinf = true; continue; }              // vertical line: killed by the final exponentiation
            Fq2 lam = (Q.y - yR) * fp_inv(dx);
            f = f * line_eval(lam, xR, yR, P);
            Fq2 x3 = fp_sqr(lam) - xR - Q.x, y3 = lam * (xR - x3) - yR;
            xR = x3; yR = y3;
        }
    }
    return f;
}
Fq12 final_exp(const Fq12& f) {          // (q^12 - 1)/r = (q^6 - 1) * ((q^6 + 1)/r); f^(q^6) = conj(f)
    static const uint64_t E[20] = {
        0x5250a54036e3f812ull, 0xa5635f1596789051ull, 0xd1138bf54d5bd1d4ull, 0xa8ce2533be36c7a2ull, 0x94f69f6b84e09bf6ull,
        0x42ad1f5e50ef3644ull, 0x0fcc420e48c3454cull, 0x758e4408ecc9952cull, 0xc901bf1887c6042cull, 0xa733cd65b14bb3b5ull,
        0xdf6d76bdcf51b0d8ull, 0xca64c0fd82eb59e1ull, 0x1d2e5726e39276a1ull, 0xc2d1ea74a391cae9ull, 0x07409206c82d647eull,
        0x051c6d1aa5afdd17ull, 0xb37f601919667af5ull, 0x150e578c5084015bull, 0xfbdea556c23998e4ull, 0x000fd14cc52f5b83ull};
    Fq12 b = conj12(f) * inv12(f), r = one12();
    for (int k = 1267; k >= 0; k--) { r = r * r; if ((E[k >> 6] >> (k & 63)) & 1) r = r * b; }
    return r;
}
bool g1_on_curve(const G1Affine& a) { return a.is_inf() || fp_sqr(a.y) == fp_sqr(a.x) * a.x + fp_from_u32<FqParams>(3); }
bool g2_on_curve(const G2Affine& a) {
    if (a.is_inf()) return true;
    static const Fq2 B = Fq2{fp_from_u32<FqParams>(3), Fq::zero()} * fp_inv(Fq2{fp_from_u32<FqParams>(9), Fq::one()});
    return fp_sqr(a.y) == fp_sqr(a.x) * a.x + B;
}
bool rd_fq_std(Fq& o, const uint8_t* p) { uint32_t s[8]; memcpy(s, p, 32); if (!fp_std_lt_p<FqParams>(s)) return false; o = fp_from_std<FqParams>(s); return true; }
bool rd_g1_std(G1Affine& o, const uint8_t* p) { return rd_fq_std(o.x, p) && rd_fq_std(o.y, p + 32); }
bool rd_g2_std(G2Affine& o, const uint8_t* p) { return rd_fq_std(o.x.c0, p) && rd_fq_std(o.x.c1, p + 32) && rd_fq_std(o.y.c0, p + 64) && rd_fq_std(o.y.c1, p + 96); }

// ---------------- minimal JSON reader for the three artifact shapes (arrays of decimal strings) ----------------
struct JTok { const char* p; const char* e; };
void skip_ws(JTok& t) { while (t.p < t.e && (*t.p == ' ' || *t.p == '\n' || *t.p == '\r' || *t.p == '\t' || *t.p == ',')) t.p++; }
bool dec_to_std(const std::string& d, uint32_t out[8], bool reduce_mod_r) {
    uint32_t t[8] = {0}; if (d.empty()) return false;
    for (char ch : d) {
        if (ch < '0' || ch > '9') return false;
        uint64_t c = (uint64_t)(ch - '0');
        for (int j = 0; j < 8; j++) { c += (uint64_t)t[j] * 10; t[j] = (uint32_t)c; c >>= 32; }
        if (c) { if (!reduce_mod_r) return false; return false; }
    }
    memcpy(out, t, 32); return true;
}
// collects every decimal string found under key `key` (flattened, document order)
bool json_strings_under(const std::string& js, const char* key, std::vector<std::string>& out) {
    size_t k = key ? js.find(std::string("\"") + key + "\"") : 0;
    if (k == std::string::npos) return false;
    size_t p = key ? js.find(':', k) : 0; if (p == std::string::npos) return false;
    if (key) p++;
    while (p < js.size() && (js[p] == ' ' || js[p] == '\n' || js[p] == '\t' || js[p] == '\r')) p++;
    if (p >= js.size()) return false;
    if (js[p] == '"') { size_t e = js.find('"', p + 1); if (e == std::string::npos) return false; out.push_back(js.substr(p + 1, e - p - 1)); return true; }
    if (js[p] != '[') return false;
    int depth = 0;
    for (; p < js.size(); p++) {
        if (js[p] == '[') depth++;
        else if (js[p] == ']') { if (--depth == 0) return true; }
        else if (js[p] == '"') { size_t e = js.find('"', p + 1); if (e == std::string::npos) return false; out.push_back(js.substr(p + 1, e - p - 1)); p = e; }
    }
    return false;
}
std::string dec_of(const uint8_t* p) {
    uint32_t s[8]; memcpy(s, p, 32); std::string out; bool nz = true;
    while (nz) { uint64_t rem = 0; nz = false; for (int i = 7; i >= 0; i--) { uint64_t cur = (rem << 32) | s[i]; s[i] = (uint32_t)(cur / 10); rem = cur % 10; if (s[i]) nz = true; } out.push_back((char)('0' + rem)); }
    return std::string(out.rbegin(), out.rend());
}
bool put_g1_json(const std::vector<std::string>& v, size_t at, uint8_t* out) {   // [x, y, z]
    uint32_t z[8]; if (at + 3 > v.size() || !dec_to_std(v[at + 2], z, false)) return false;
    bool inf = true; for (int i = 0; i < 8; i++) inf &= z[i] == 0;
    if (inf) { memset(out, 0, 64); return true; }
    uint32_t s[8]; if (!dec_to_std(v[at], s, false)) return false; memcpy(out, s, 32);
    if (!dec_to_std(v[at + 1], s, false)) return false; memcpy(out + 32, s, 32); return true;
}
bool put_g2_json(const std::vector<std::string>& v, size_t at, uint8_t* out) {   // [[x0,x1],[y0,y1],[z0,z1]]
    uint32_t z0[8], z1[8]; if (at + 6 > v.size() || !dec_to_std(v[at + 4], z0, false) || !dec_to_std(v[at + 5], z1, false)) return false;
    bool inf = true; for (int i = 0; i < 8; i++) inf &= (z0[i] | z1[i]) == 0;
    if (inf) { memset(out, 0, 128); return true; }
    for (int k = 0; k < 4; k++) { uint32_t s[8]; if (!dec_to_std(v[at + k], s, false)) return false; memcpy(out + 32 * k, s, 32); }
    return true;
}

thread_local std::string g_err;
int vfail(int code, const std::string& m) { g_err = m; return code; }

}  // namespace

extern "C" const char* zkc_verify_last_error(void) { return g_err.c_str(); }

// vk: alpha1(64) beta2(128) gamma2(128) delta2(128) IC[nPublic+1](64 each); pub: nPublic x 32; proof: A(64) B(128) C(64); standard form
extern "C" int zkc_verify_bin(const uint8_t* vk, int nPublic, const uint8_t* pub, const uint8_t* proof) {
    if (!vk || !pub || !proof || nPublic < 0) return vfail(-ZKC_ERR_BAD_ARG, "zkc_verify_bin: bad argument");
    G1Affine alpha, A, C; G2Affine beta, gamma, delta, B;
    if (!rd_g1_std(alpha, vk) || !rd_g2_std(beta, vk + 64) || !rd_g2_std(gamma, vk + 192) || !rd_g2_std(delta, vk + 320)) return vfail(-ZKC_ERR_FORMAT, "verification key coordinate >= q");
    if (!rd_g1_std(A, proof) || !rd_g2_std(B, proof + 64) || !rd_g1_std(C, proof + 192)) return 0;
    if (!g1_on_curve(A) || !g1_on_curve(C) || !g2_on_curve(B)) return 0;
    G1Affine ic; if (!rd_g1_std(ic, vk + 448)) return vfail(-ZKC_ERR_FORMAT, "IC coordinate >= q");
    G1XYZZ acc = G1XYZZ::from_affine(ic);
    for (int i = 0; i < nPublic; i++) {
        uint32_t k[8]; memcpy(k, pub + 32 * i, 32);
        if (!fp_std_lt_p<FrParams>(k)) return 0;                                   // snarkjs: public input not in field -> invalid
        if (!rd_g1_std(ic, vk + 448 + 64 * (i + 1))) return vfail(-ZKC_ERR_FORMAT, "IC coordinate >= q");
        acc = xyzz_add(acc, xyzz_mul(G1XYZZ::from_affine(ic), k));
    }
    const G1Affine vkx = xyzz_to_affine(acc);
    Fq12 f = miller(affine_neg(A), B) * miller(alpha, beta) * miller(vkx, gamma) * miller(C, delta);
    return is_one12(final_exp(f)) ? 1 : 0;
}

// ---- f4: batch verification (the step on the other side of the path, zk_census_test.go:103-124 run per vote) ----
// N proofs under one key are folded into one pairing-product check with random 128-bit weights rho_i:
//     prod_i e(-rho_i A_i, B_i) * e((sum rho_i) alpha, beta) * e(sum_i rho_i vk_x_i, gamma) * e(sum_i rho_i C_i, delta) == 1
// i.e. N + 3 Miller loops and ONE final exponentiation instead of 4 N and N.  The G1 work (rho_i A_i for every proof and the MSM
// sum rho_i C_i) runs on the GPU with the prover's double-and-add / group-sum kernels; Miller loops run on host threads.
// A cheating prover passes with probability about 2^-128 provided the weights are unpredictable to it: `seed32` must be fresh
// randomness (NULL: std::random_device).  Each B_i is checked to lie in the order-r subgroup of the twist (G2 has a cofactor; the
// single-proof check, like snarkjs, does not need that).  Returns 1 all valid / 0 at least one invalid / <0 = -ZKC_ERR_*.
namespace {
struct Xoshiro { uint64_t s[4]; uint64_t next() { auto rotl = [](uint64_t x, int k) { return (x << k) | (x >> (64 - k)); };
    const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17; s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45); return r; } };
}
extern "C" int zkc_verify_batch(zkc_ctx* ctx, const uint8_t* vk, int nPublic, const uint8_t* pubs, const uint8_t* proofs, int N, const uint8_t* seed32) {
    if (!ctx || !vk || !pubs || !proofs || nPublic < 0 || N <= 0) return vfail(-ZKC_ERR_BAD_ARG, "zkc_verify_batch: bad argument");
    G1Affine alpha; G2Affine beta, gamma, delta; std::vector<G1Affine> ic(nPublic + 1);
    if (!rd_g1_std(alpha, vk) || !rd_g2_std(beta, vk + 64) || !rd_g2_std(gamma, vk + 192) || !rd_g2_std(delta, vk + 320)) return vfail(-ZKC_ERR_FORMAT, "verification key coordinate >= q");
    for (int j = 0; j <= nPublic; j++) if (!rd_g1_std(ic[j], vk + 448 + 64 * (size_t)j)) return vfail(-ZKC_ERR_FORMAT, "IC coordinate >= q");
    Xoshiro rng;
    if (seed32) memcpy(rng.s, seed32, 32); else { std::random_device rd; for (auto& x : rng.s) x = ((uint64_t)rd() << 32) | rd(); }
    if (!(rng.s[0] | rng.s[1] | rng.s[2] | rng.s[3])) rng.s[0] = 1;
    // ---- parse, per-proof membership checks, weights ----
    std::vector<G1Affine> pts(2 * (size_t)N); std::vector<G2Affine> Bs(N); std::vector<uint32_t> rho(8 * 2 * (size_t)N, 0);
    std::vector<Fr> xsum(nPublic, Fr::zero()); Fr rsum = Fr::zero();
    for (int i = 0; i < N; i++) {
        const uint8_t* pr = proofs + 256 * (size_t)i;
        if (!rd_g1_std(pts[i], pr) || !rd_g2_std(Bs[i], pr + 64) || !rd_g1_std(pts[N + i], pr + 192)) return 0;
        if (!g1_on_curve(pts[i]) || !g1_on_curve(pts[N + i]) || !g2_on_curve(Bs[i])) return 0;
        uint32_t* r = rho.data() + 8 * (size_t)i;
        const uint64_t lo = rng.next(), hi = rng.next(); r[0] = (uint32_t)lo; r[1] = (uint32_t)(lo >> 32); r[2] = (uint32_t)hi; r[3] = (uint32_t)(hi >> 32);
        memcpy(rho.data() + 8 * ((size_t)N + i), r, 32);
        const Fr rm = fp_from_std<FrParams>(r); rsum = rsum + rm;
        for (int j = 0; j < nPublic; j++) {
            uint32_t k[8]; memcpy(k, pubs + 32 * ((size_t)i * nPublic + j), 32);
            if (!fp_std_lt_p<FrParams>(k)) return 0;
            xsum[j] = xsum[j] + rm * fp_from_std<FrParams>(k);
        }
    }
    // ---- G1 side on the GPU: rho_i A_i (N single-element groups) and sum rho_i C_i (one group) ----
    std::vector<G1XYZZ> gout(N + 1);
    {
        ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device));
        G1Affine* d_pts = nullptr; uint32_t *d_rho = nullptr, *d_idx = nullptr, *d_gs = nullptr;
        std::vector<uint32_t> idx(2 * (size_t)N), gs(N + 2);
        for (size_t i = 0; i < idx.size(); i++) idx[i] = (uint32_t)i;
        for (int i = 0; i <= N; i++) gs[i] = (uint32_t)i; gs[N + 1] = 2 * (uint32_t)N;
        int rc = ZKC_OK;
        auto cleanup = [&]() { for (void* q : {(void*)d_pts, (void*)d_rho, (void*)d_idx, (void*)d_gs}) if (q) (void)hipFree(q); };
        if (hipMalloc((void**)&d_pts, pts.size() * sizeof(G1Affine)) != hipSuccess || hipMalloc((void**)&d_rho, rho.size() * 4) != hipSuccess ||
            hipMalloc((void**)&d_idx, idx.size() * 4) != hipSuccess || hipMalloc((void**)&d_gs, gs.size() * 4) != hipSuccess) { cleanup(); return vfail(-ZKC_ERR_HIP, "zkc_verify_batch: hipMalloc failed"); }
        if (hipMemcpy(d_pts, pts.data(), pts.size() * sizeof(G1Affine), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_rho, rho.data(), rho.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_gs, gs.data(), gs.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { cleanup(); return vfail(-ZKC_ERR_HIP, "zkc_verify_batch: hipMemcpy failed"); }
        rc = fold_group_sums_g1(ctx, d_pts, d_rho, d_idx, 2 * (uint32_t)N, 0, d_gs, (uint32_t)N + 1, gout.data());
        cleanup();
        if (rc) return vfail(-rc, std::string("zkc_verify_batch: ") + zkc_last_error(ctx));
    }
    // ---- vk_x side: (sum rho) IC0 + sum_j (sum_i rho_i x_ij) IC_j ----
    uint32_t k[8]; fp_to_std<FrParams>(k, rsum);
    G1XYZZ vx = xyzz_mul(G1XYZZ::from_affine(ic[0]), k);
    const G1Affine ralpha = xyzz_to_affine(xyzz_mul(G1XYZZ::from_affine(alpha), k));
    for (int j = 0; j < nPublic; j++) { fp_to_std<FrParams>(k, xsum[j]); vx = xyzz_add(vx, xyzz_mul(G1XYZZ::from_affine(ic[j + 1]), k)); }
    // ---- Miller loops on host threads (and the subgroup check of every B_i) ----
    const unsigned nthr = std::max(1u, std::min({std::thread::hardware_concurrency(), 32u, (unsigned)N}));
    std::vector<Fq12> part(nthr, one12()); std::vector<int> bad(nthr, 0);
    auto work = [&](unsigned t) {
        uint32_t rord[8]; for (int q = 0; q < 8; q++) rord[q] = FrParams::p[q];
        Fq12 f = one12();
        for (int i = (int)t; i < N; i += (int)nthr) {
            if (!xyzz_mul(G2XYZZ::from_affine(Bs[i]), rord).is_inf()) { bad[t] = 1; return; }
            f = f * miller(affine_neg(xyzz_to_affine(gout[i])), Bs[i]);
        }
        part[t] = f;
    };
    std::vector<std::thread> th; for (unsigned t = 1; t < nthr; t++) th.emplace_back(work, t);
    work(0); for (auto& x : th) x.join();
    for (unsigned t = 0; t < nthr; t++) if (bad[t]) return 0;
    Fq12 f = miller(ralpha, beta) * miller(xyzz_to_affine(vx), gamma) * miller(xyzz_to_affine(gout[N]), delta);
    for (unsigned t = 0; t < nthr; t++) f = f * part[t];
    return is_one12(final_exp(f)) ? 1 : 0;
}

// JSON surface: the three artifact files of the reference (verification_key.json, signals.json, proof.json). 1 valid / 0 invalid / <0 error
extern "C" int zkc_verify(const char* vkey_json, const char* public_json, const char* proof_json) {
    if (!vkey_json || !public_json || !proof_json) return vfail(-ZKC_ERR_BAD_ARG, "zkc_verify: bad argument");
    const std::string vk(vkey_json), pj(public_json), pr(proof_json);
    std::vector<std::string> a1, b2, g2, d2, ic, pub, pa, pb, pc;
    if (!json_strings_under(vk, "vk_alpha_1", a1) || !json_strings_under(vk, "vk_beta_2", b2) || !json_strings_under(vk, "vk_gamma_2", g2) ||
        !json_strings_under(vk, "vk_delta_2", d2) || !json_strings_under(vk, "IC", ic)) return vfail(-ZKC_ERR_FORMAT, "verification key JSON: missing member");
    if (!json_strings_under(pj, nullptr, pub)) return vfail(-ZKC_ERR_FORMAT, "public signals JSON: expected an array of decimal strings");
    if (!json_strings_under(pr, "pi_a", pa) || !json_strings_under(pr, "pi_b", pb) || !json_strings_under(pr, "pi_c", pc)) return vfail(-ZKC_ERR_FORMAT, "proof JSON: missing member");
    const int np = (int)pub.size();
    if (ic.size() != 3 * (size_t)(np + 1)) return vfail(-ZKC_ERR_FORMAT, "verification key: IC length does not match the public signals");
    std::vector<uint8_t> vkb(448 + 64 * (size_t)(np + 1)), pubb(32 * (size_t)np + 1), prb(256);
    if (!put_g1_json(a1, 0, vkb.data()) || !put_g2_json(b2, 0, vkb.data() + 64) || !put_g2_json(g2, 0, vkb.data() + 192) || !put_g2_json(d2, 0, vkb.data() + 320))
        return vfail(-ZKC_ERR_FORMAT, "verification key JSON: bad point");
    for (int i = 0; i <= np; i++) if (!put_g1_json(ic, 3 * (size_t)i, vkb.data() + 448 + 64 * i)) return vfail(-ZKC_ERR_FORMAT, "verification key JSON: bad IC point");
    for (int i = 0; i < np; i++) { uint32_t s[8]; if (!dec_to_std(pub[i], s, false)) return 0; memcpy(pubb.data() + 32 * i, s, 32); }
    if (!put_g1_json(pa, 0, prb.data()) || !put_g2_json(pb, 0, prb.data() + 64) || !put_g1_json(pc, 0, prb.data() + 192)) return 0;
    return zkc_verify_bin(vkb.data(), np, pubb.data(), prb.data());
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
    const uint8_t* b = (const uint8_t*)wtns_bytes;
    if (!b || size < 12 || memcmp(b, "wtns", 4)) return ZKC_ERR_FORMAT;
    uint32_t ver, nsec; memcpy(&ver, b + 4, 4); memcpy(&nsec, b + 8, 4); if (ver != 2) return ZKC_ERR_FORMAT;
    size_t p = 12; const uint8_t *s1 = nullptr, *s2 = nullptr; uint64_t z2 = 0;
    for (uint32_t i = 0; i < nsec; i++) {
        if (p + 12 > size) return ZKC_ERR_FORMAT;
        uint32_t id; uint64_t sz; memcpy(&id, b + p, 4); memcpy(&sz, b + p + 4, 8); p += 12;
        if (p + sz > size) return ZKC_ERR_FORMAT;
        if (id == 1) s1 = b + p; if (id == 2) { s2 = b + p; z2 = sz; }
        p += sz;
    }
    if (!s1 || !s2) return ZKC_ERR_FORMAT;
    uint32_t n8, nw; memcpy(&n8, s1, 4); if (n8 != 32 || memcmp(s1 + 4, FrParams::p, 32)) return ZKC_ERR_FORMAT;
    memcpy(&nw, s1 + 36, 4); if (z2 != 32ull * nw) return ZKC_ERR_FORMAT;
    if (payload) *payload = s2; if (nWitness) *nWitness = nw;
    return ZKC_OK;
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

// rapidsnark's entry point (prover.h), byte for byte: whole .zkey and .wtns buffers in, NUL-terminated JSON out.
// r and s are drawn from the OS generator like the reference provers do.  One process-wide context on device
// $ZKC_DEVICE (default 0); the last key stays resident so that repeated calls with the same buffer skip the load.
extern "C" int groth16_prover(const void* zkey_buffer, unsigned long zkey_size, const void* wtns_buffer, unsigned long wtns_size,
                              char* proof_buffer, unsigned long* proof_size, char* public_buffer, unsigned long* public_size,
                              char* error_msg, unsigned long error_msg_maxsize) {
    auto err = [&](int code, const std::string& m) { if (error_msg && error_msg_maxsize) snprintf(error_msg, error_msg_maxsize, "%s", m.c_str()); return code; };
    if (!zkey_buffer || !wtns_buffer || !proof_size || !public_size) return err(ZKC_ERR_GENERIC, "groth16_prover: null argument");
    static zkc_ctx* ctx = nullptr; static zkc_zkey* zk = nullptr; static const void* zk_ptr = nullptr; static unsigned long zk_len = 0; static uint64_t zk_sum = 0;
    if (!ctx) { const char* d = getenv("ZKC_DEVICE"); int rc = zkc_ctx_create(d ? atoi(d) : 0, &ctx); if (rc) { ctx = nullptr; return err(ZKC_ERR_GENERIC, zkc_last_error(nullptr)); } }
    uint64_t sum = 1469598103934665603ull; { const uint8_t* b = (const uint8_t*)zkey_buffer; for (unsigned long i = 0; i < zkey_size; i += 4099) sum = (sum ^ b[i]) * 1099511628211ull; }
    if (!zk || zk_ptr != zkey_buffer || zk_len != zkey_size || zk_sum != sum) {
        if (zk) { zkc_zkey_free(zk); zk = nullptr; }
        int rc = zkc_zkey_load(ctx, zkey_buffer, zkey_size, &zk); if (rc) { zk = nullptr; return err(ZKC_ERR_GENERIC, zkc_last_error(ctx)); }
        zk_ptr = zkey_buffer; zk_len = zkey_size; zk_sum = sum;
    }
    const uint8_t* payload; uint32_t nw;
    if (zkc_wtns_parse(wtns_buffer, wtns_size, &payload, &nw)) return err(ZKC_ERR_GENERIC, "Invalid witness file");
    uint32_t nv, np, dn; zkc_zkey_info(zk, &nv, &np, &dn);
    if (nw != nv) return err(ZKC_ERR_INVALID_WITNESS_LENGTH, "Invalid witness length. Circuit: " + std::to_string(nv) + ", witness: " + std::to_string(nw));
    uint8_t r[32], s[32], proof[256]; std::vector<uint8_t> pub(32 * (size_t)np + 1);
    { std::random_device rd; for (int i = 0; i < 31; i++) { r[i] = (uint8_t)rd(); s[i] = (uint8_t)rd(); } r[31] = s[31] = 0; }    // < 2^248 < field order
    int rc = zkc_prove(zk, payload, nw, r, s, proof, pub.data());
    if (rc) return err(rc == ZKC_ERR_INVALID_WITNESS_LENGTH ? rc : ZKC_ERR_GENERIC, zkc_last_error(ctx));
    rc = zkc_proof_to_json(proof, pub.data(), (int)np, proof_buffer, proof_size, public_buffer, public_size);
    if (rc == ZKC_ERR_SHORT_BUFFER) return err(ZKC_ERR_SHORT_BUFFER, "Proof or public signals buffer is too short");
    return rc;
}
