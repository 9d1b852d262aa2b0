// zkc_prove.hip -- proving-key residency and the Groth16 prove pipeline behind the C ABI (product code).
//
// zkc_zkey_load  parses a snarkjs-format Groth16 .zkey (the reference's proving_key.zkey format),
//                turns section 4 into CSR, uploads the bases and pre-shifts them for the MSM windows.
// zkc_prove_dev  witness (device, standard form) -> proof: buildABC -> 3 x (iNTT, coset shift, NTT) -> joinABC ->
//                5 MSMs -> blinding (a7) on the host with injectable (r, s).
// Mirrors snarkjs groth16.prove (ts_inputs/src/example.ts:358-362 via fullProve) / rapidsnark groth16_prover
// (zk_census_test.go:89).
#include "zkc_prover.h"
#include <cstring>
#include <algorithm>

using namespace zkc;

extern "C" __global__ void zkc_matvec(const uint32_t*, const uint32_t*, const Fr*, const Fr*, Fr*, int);
extern "C" __global__ void zkc_pointwise_mul(const Fr*, const Fr*, Fr*, int);
extern "C" __global__ void zkc_join_abc(const Fr*, const Fr*, const Fr*, uint32_t*, int);

namespace {
uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
G1Affine rd_g1(const uint8_t* p) { G1Affine a; memcpy(a.x.v, p, 32); memcpy(a.y.v, p + 32, 32); return a; }
G2Affine rd_g2(const uint8_t* p) { G2Affine a; memcpy(a.x.c0.v, p, 32); memcpy(a.x.c1.v, p + 32, 32); memcpy(a.y.c0.v, p + 64, 32); memcpy(a.y.c1.v, p + 96, 32); return a; }
Fr fr_root_of_unity(int logn) {
    uint32_t e[8]; for (int i = 0; i < 8; i++) e[i] = FrParams::p[i]; e[0] -= 1;
    for (int i = 0; i < 8; i++) e[i] = (e[i] >> 28) | (i < 7 ? e[i + 1] << 4 : 0);
    Fr g = fp_from_u32<FrParams>(5), w = Fr::one();
    for (int i = 255; i >= 0; i--) { w = w * w; if ((e[i >> 5] >> (i & 31)) & 1) w = w * g; }
    for (int i = 28; i > logn; i--) w = w * w;
    return w;
}
template <class T> int dmalloc(zkc_ctx* ctx, T** p, size_t count) { ZKC_HIP_CHECK(ctx, hipMalloc((void**)p, count * sizeof(T))); return ZKC_OK; }
void g1_to_std(uint8_t* out, const G1Affine& a) { uint32_t s[8]; fp_to_std<FqParams>(s, a.x); memcpy(out, s, 32); fp_to_std<FqParams>(s, a.y); memcpy(out + 32, s, 32); }
void g2_to_std(uint8_t* out, const G2Affine& a) {
    uint32_t s[8]; fp_to_std<FqParams>(s, a.x.c0); memcpy(out, s, 32); fp_to_std<FqParams>(s, a.x.c1); memcpy(out + 32, s, 32);
    fp_to_std<FqParams>(s, a.y.c0); memcpy(out + 64, s, 32); fp_to_std<FqParams>(s, a.y.c1); memcpy(out + 96, s, 32);
}
}  // namespace

extern "C" void zkc_zkey_free(zkc_zkey* zk) {
    if (!zk) return;
    (void)hipSetDevice(zk->ctx->device);
    (void)hipStreamSynchronize(zk->ctx->stream);
    void* ptrs[] = {zk->d_rowptr, zk->d_col, zk->d_val, zk->d_tw_fwd, zk->d_tw_inv, zk->d_coset, zk->d_A, zk->d_B1, zk->d_C, zk->d_H, zk->d_B2,
                    zk->d_a, zk->d_c, zk->d_t, zk->d_p, zk->d_keys, zk->d_vals, zk->d_keys2, zk->d_vals2, zk->d_off, zk->d_heavy, zk->d_sort_tmp,
                    zk->d_buckets, zk->d_partial, zk->d_results};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (zk->h_results) (void)hipHostFree(zk->h_results);
    delete zk;
}

extern "C" int zkc_zkey_load(zkc_ctx* ctx, const void* zkey_bytes, size_t len, zkc_zkey** out) {
    if (!ctx || !zkey_bytes || !out) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "zkc_zkey_load: bad argument");
    ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint8_t* buf = (const uint8_t*)zkey_bytes;
    if (len < 12 || memcmp(buf, "zkey", 4) || rd32(buf + 4) != 1) return zkc_fail(ctx, ZKC_ERR_FORMAT, "not a zkey v1 file");
    const uint8_t* sec[16] = {nullptr}; uint64_t ssz[16] = {0};
    size_t p = 12;
    for (uint32_t i = 0, ns = rd32(buf + 8); i < ns; i++) {
        if (p + 12 > len) return zkc_fail(ctx, ZKC_ERR_FORMAT, "zkey: truncated section table");
        uint32_t id = rd32(buf + p); uint64_t sz = rd64(buf + p + 4); p += 12;
        if (p + sz > len) return zkc_fail(ctx, ZKC_ERR_FORMAT, "zkey: truncated section");
        if (id < 16) { sec[id] = buf + p; ssz[id] = sz; }
        p += sz;
    }
    for (int i = 1; i <= 9; i++) if (!sec[i]) return zkc_fail(ctx, ZKC_ERR_FORMAT, "zkey: missing section " + std::to_string(i));
    if (rd32(sec[1]) != 1) return zkc_fail(ctx, ZKC_ERR_FORMAT, "zkey: protocol is not groth16");
    const uint8_t* h = sec[2];
    if (rd32(h) != 32 || memcmp(h + 4, FqParams::p, 32) || rd32(h + 36) != 32 || memcmp(h + 40, FrParams::p, 32))
        return zkc_fail(ctx, ZKC_ERR_FORMAT, "zkey: curve is not bn128");
    zkc_zkey* zk = new zkc_zkey(); zk->ctx = ctx;
    zk->nVars = rd32(h + 72); zk->nPub = rd32(h + 76); zk->n = rd32(h + 80);
    while ((1u << zk->logn) < zk->n) zk->logn++;
    const uint32_t n = zk->n, nv = zk->nVars, np = zk->nPub, nc = nv - np - 1;
    if ((1u << zk->logn) != n || ssz[3] != 64ull * (np + 1) || ssz[5] != 64ull * nv || ssz[6] != 64ull * nv || ssz[7] != 128ull * nv ||
        ssz[8] != 64ull * nc || ssz[9] != 64ull * n) { delete zk; return zkc_fail(ctx, ZKC_ERR_FORMAT, "zkey: section sizes do not match the header"); }
    zk->alpha1 = rd_g1(h + 84); zk->beta1 = rd_g1(h + 148); zk->beta2 = rd_g2(h + 212); zk->gamma2 = rd_g2(h + 340);
    zk->delta1 = rd_g1(h + 468); zk->delta2 = rd_g2(h + 532);
    for (uint32_t i = 0; i <= np; i++) zk->ic.push_back(rd_g1(sec[3] + 64ull * i));
    zk->nCoeffs = rd32(sec[4]);
    if (ssz[4] != 4 + 44ull * zk->nCoeffs) { delete zk; return zkc_fail(ctx, ZKC_ERR_FORMAT, "zkey: coefficient section size"); }
    int rc = ZKC_OK;
    auto bail = [&](int code) { zkc_zkey_free(zk); return code; };
#define ZKC_UP(dst, src, bytes)                                                                          \
    do { hipError_t _e = hipMemcpy((dst), (src), (bytes), hipMemcpyHostToDevice);                        \
        if (_e != hipSuccess) return bail(zkc_fail(ctx, ZKC_ERR_HIP, std::string("hipMemcpy H2D: ") + hipGetErrorString(_e))); } while (0)
    // ---- section 4 -> CSR (row = matrix * n + constraint) ----
    {
        std::vector<uint32_t> rowptr(2 * (size_t)n + 1, 0), col(zk->nCoeffs); std::vector<Fr> val(zk->nCoeffs);
        const uint8_t* c = sec[4] + 4;
        for (uint32_t i = 0; i < zk->nCoeffs; i++) {
            uint32_t m = rd32(c + 44ull * i), cc = rd32(c + 44ull * i + 4), s = rd32(c + 44ull * i + 8);
            if (m > 1 || cc >= n || s >= nv) return bail(zkc_fail(ctx, ZKC_ERR_FORMAT, "zkey: coefficient out of range"));
            rowptr[(size_t)m * n + cc + 1]++;
        }
        for (size_t r = 0; r < 2 * (size_t)n; r++) rowptr[r + 1] += rowptr[r];
        std::vector<uint32_t> fill(rowptr.begin(), rowptr.end() - 1);
        for (uint32_t i = 0; i < zk->nCoeffs; i++) {
            uint32_t m = rd32(c + 44ull * i), cc = rd32(c + 44ull * i + 4), s = rd32(c + 44ull * i + 8);
            uint32_t k = fill[(size_t)m * n + cc]++;
            col[k] = s; memcpy(val[k].v, c + 44ull * i + 12, 32);
        }
        if ((rc = dmalloc(ctx, &zk->d_rowptr, rowptr.size())) || (rc = dmalloc(ctx, &zk->d_col, col.size() + 1)) || (rc = dmalloc(ctx, &zk->d_val, val.size() + 1))) return bail(rc);
        ZKC_UP(zk->d_rowptr, rowptr.data(), rowptr.size() * 4);
        ZKC_UP(zk->d_col, col.data(), col.size() * 4);
        ZKC_UP(zk->d_val, val.data(), val.size() * sizeof(Fr));
    }
    // ---- twiddles and the coset/1-over-n scale ----
    {
        const Fr w = fr_root_of_unity((int)zk->logn), g = fr_root_of_unity((int)zk->logn + 1);
        const Fr wi = fp_inv<FrParams>(w), ninv = fp_inv<FrParams>(fp_from_u32<FrParams>(n));
        std::vector<Fr> f(n / 2), b(n / 2), cs(n);
        f[0] = b[0] = Fr::one(); for (uint32_t i = 1; i < n / 2; i++) { f[i] = f[i - 1] * w; b[i] = b[i - 1] * wi; }
        cs[0] = ninv; for (uint32_t i = 1; i < n; i++) cs[i] = cs[i - 1] * g;
        if ((rc = dmalloc(ctx, &zk->d_tw_fwd, n / 2)) || (rc = dmalloc(ctx, &zk->d_tw_inv, n / 2)) || (rc = dmalloc(ctx, &zk->d_coset, n))) return bail(rc);
        ZKC_UP(zk->d_tw_fwd, f.data(), f.size() * sizeof(Fr));
        ZKC_UP(zk->d_tw_inv, b.data(), b.size() * sizeof(Fr));
        ZKC_UP(zk->d_coset, cs.data(), cs.size() * sizeof(Fr));
    }
    // ---- bases: window 0 = the zkey points as stored (affine, Montgomery), windows 1.. pre-shifted on the device ----
    if ((rc = dmalloc(ctx, &zk->d_A, (size_t)MSM_NW * nv)) || (rc = dmalloc(ctx, &zk->d_B1, (size_t)MSM_NW * nv)) || (rc = dmalloc(ctx, &zk->d_B2, (size_t)MSM_NW * nv)) ||
        (rc = dmalloc(ctx, &zk->d_C, (size_t)MSM_NW * nc)) || (rc = dmalloc(ctx, &zk->d_H, (size_t)MSM_NW * n))) return bail(rc);
    ZKC_UP(zk->d_A, sec[5], 64ull * nv); ZKC_UP(zk->d_B1, sec[6], 64ull * nv);
    ZKC_UP(zk->d_B2, sec[7], 128ull * nv); ZKC_UP(zk->d_C, sec[8], 64ull * nc);
    ZKC_UP(zk->d_H, sec[9], 64ull * n);
    if ((rc = msm_precompute_g1(ctx, nullptr, nv, zk->d_A)) || (rc = msm_precompute_g1(ctx, nullptr, nv, zk->d_B1)) || (rc = msm_precompute_g2(ctx, nullptr, nv, zk->d_B2)) ||
        (rc = msm_precompute_g1(ctx, nullptr, nc, zk->d_C)) || (rc = msm_precompute_g1(ctx, nullptr, n, zk->d_H))) return bail(rc);
    // ---- work buffers ----
    const size_t maxpts = std::max<size_t>(nv, n), total = maxpts * MSM_NW;
    if ((rc = dmalloc(ctx, &zk->d_a, 2 * (size_t)n)) || (rc = dmalloc(ctx, &zk->d_c, n)) || (rc = dmalloc(ctx, &zk->d_t, n)) || (rc = dmalloc(ctx, &zk->d_p, 8 * (size_t)n)) ||
        (rc = dmalloc(ctx, &zk->d_keys, total)) || (rc = dmalloc(ctx, &zk->d_vals, total)) || (rc = dmalloc(ctx, &zk->d_keys2, total)) || (rc = dmalloc(ctx, &zk->d_vals2, total)) ||
        (rc = dmalloc(ctx, &zk->d_off, MSM_NB + 2)) || (rc = dmalloc(ctx, &zk->d_heavy, MSM_MAX_HEAVY + 1))) return bail(rc);
    zk->d_b = zk->d_a + n;
    ZKC_HIP_CHECK(ctx, hipMalloc(&zk->d_buckets, (size_t)MSM_NB * sizeof(G2XYZZ)));
    ZKC_HIP_CHECK(ctx, hipMalloc(&zk->d_partial, (size_t)(MSM_NB / MSM_GROUP / 64 + 1) * sizeof(G2XYZZ)));
    ZKC_HIP_CHECK(ctx, hipMalloc(&zk->d_results, 8 * sizeof(G2XYZZ)));
    ZKC_HIP_CHECK(ctx, hipHostMalloc(&zk->h_results, 8 * sizeof(G2XYZZ)));
    ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *out = zk;
    return ZKC_OK;
}

extern "C" int zkc_zkey_info(const zkc_zkey* zk, uint32_t* nVars, uint32_t* nPublic, uint32_t* domainSize) {
    if (!zk) return ZKC_ERR_BAD_ARG;
    if (nVars) *nVars = zk->nVars; if (nPublic) *nPublic = zk->nPub; if (domainSize) *domainSize = zk->n;
    return ZKC_OK;
}

// stages a2..a4: leaves (A'B' - C') on the odd coset in zk->d_p (standard form) and the intermediate vectors in d_a/d_b/d_c
static int h_evals_dev(zkc_zkey* zk, const uint32_t* d_wtns) {
    zkc_ctx* ctx = zk->ctx; const uint32_t n = zk->n; hipStream_t st = ctx->stream;
    {
    zkc_prof_scope _ps(ctx, ZKC_PROF_MATVEC, (uint64_t)zk->nCoeffs * 68 + 3ull * n * 32);
    hipLaunchKernelGGL(zkc_matvec, dim3((2 * n + 255) / 256), dim3(256), 0, st, zk->d_rowptr, zk->d_col, zk->d_val, (const Fr*)d_wtns, zk->d_a, (int)(2 * n));
    ZKC_HIP_CHECK(ctx, hipGetLastError());
    hipLaunchKernelGGL(zkc_pointwise_mul, dim3((n + 255) / 256), dim3(256), 0, st, zk->d_a, zk->d_b, zk->d_c, (int)n);
    ZKC_HIP_CHECK(ctx, hipGetLastError());
    }
    zkc_prof_scope _pn(ctx, ZKC_PROF_NTT, 6ull * 2 * n * 32 + 4ull * n * 32);   // 6 transforms r+w, joinABC
    Fr* v[3] = {zk->d_a, zk->d_b, zk->d_c};
    for (int k = 0; k < 3; k++) {
        int rc = ntt_run(ctx, v[k], zk->d_t, zk->d_tw_inv, zk->d_coset, (int)zk->logn); if (rc) return rc;
        rc = ntt_run(ctx, zk->d_t, v[k], zk->d_tw_fwd, nullptr, (int)zk->logn); if (rc) return rc;
    }
    hipLaunchKernelGGL(zkc_join_abc, dim3((n + 255) / 256), dim3(256), 0, st, zk->d_a, zk->d_b, zk->d_c, zk->d_p, (int)n);
    ZKC_HIP_CHECK(ctx, hipGetLastError());
    return ZKC_OK;
}

extern "C" int zkc_debug_stage(zkc_zkey* zk, const void* d_wtns, int stage, void* host_out) {
    // test hook: stage 0 -> A_w | B_w | C_w (3n Fr, Montgomery) after buildABC; stage 1 -> joinABC output (n x 32 B standard)
    if (!zk || !d_wtns || !host_out) return ZKC_ERR_BAD_ARG;
    zkc_ctx* ctx = zk->ctx; const uint32_t n = zk->n;
    ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (stage == 0) {
        hipLaunchKernelGGL(zkc_matvec, dim3((2 * n + 255) / 256), dim3(256), 0, ctx->stream, zk->d_rowptr, zk->d_col, zk->d_val, (const Fr*)d_wtns, zk->d_a, (int)(2 * n));
        hipLaunchKernelGGL(zkc_pointwise_mul, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, zk->d_a, zk->d_b, zk->d_c, (int)n);
        ZKC_HIP_CHECK(ctx, hipGetLastError());
        ZKC_HIP_CHECK(ctx, hipMemcpyAsync(host_out, zk->d_a, 64ull * n, hipMemcpyDeviceToHost, ctx->stream));
        ZKC_HIP_CHECK(ctx, hipMemcpyAsync((uint8_t*)host_out + 64ull * n, zk->d_c, 32ull * n, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        int rc = h_evals_dev(zk, (const uint32_t*)d_wtns); if (rc) return rc;
        ZKC_HIP_CHECK(ctx, hipMemcpyAsync(host_out, zk->d_p, 32ull * n, hipMemcpyDeviceToHost, ctx->stream));
    }
    ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ZKC_OK;
}

extern "C" int zkc_msm_debug(zkc_zkey* zk, int which, const void* d_scalars, uint32_t count, void* host_out) {
    // test hook: one MSM over a zkey section (0=A 1=B1 2=B2 3=C 4=H) with caller scalars; host_out = affine standard form
    if (!zk || !d_scalars || !host_out || which < 0 || which > 4) return ZKC_ERR_BAD_ARG;
    zkc_ctx* ctx = zk->ctx;
    ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t full = which == 3 ? zk->nVars - zk->nPub - 1 : which == 4 ? zk->n : zk->nVars;
    if (count != full) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "zkc_msm_debug: count must equal the section size");
    int rc;
    if (which == 2) rc = msm_g2_run(zk, zk->d_B2, (const uint32_t*)d_scalars, count, 0);
    else rc = msm_g1_run(zk, which == 0 ? zk->d_A : which == 1 ? zk->d_B1 : which == 3 ? zk->d_C : zk->d_H, (const uint32_t*)d_scalars, count, 0);
    if (rc) return rc;
    ZKC_HIP_CHECK(ctx, hipMemcpyAsync(zk->h_results, zk->d_results, sizeof(G2XYZZ), hipMemcpyDeviceToHost, ctx->stream));
    ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (which == 2) g2_to_std((uint8_t*)host_out, xyzz_to_affine(*(G2XYZZ*)zk->h_results));
    else g1_to_std((uint8_t*)host_out, xyzz_to_affine(*(G1XYZZ*)zk->h_results));
    return ZKC_OK;
}

// a7: piA = alpha + A + r delta ; piB = beta + B + s delta ; piC = C + H + s piA + r piB1 - r s delta   (host, constant work)
static void finalize_proof(const zkc_zkey* zk, const G2XYZZ* res, const uint8_t r32[32], const uint8_t s32[32], uint8_t proof[256]) {
    uint32_t rk[8], sk[8]; memcpy(rk, r32, 32); memcpy(sk, s32, 32);
    const G1XYZZ A = *(const G1XYZZ*)&res[0], B1 = *(const G1XYZZ*)&res[1], C = *(const G1XYZZ*)&res[3], H = *(const G1XYZZ*)&res[4];
    const G2XYZZ B2 = res[2];
    const G1XYZZ d1 = G1XYZZ::from_affine(zk->delta1); const G2XYZZ d2 = G2XYZZ::from_affine(zk->delta2);
    G1XYZZ piA = xyzz_add(xyzz_add_affine(A, zk->alpha1), xyzz_mul(d1, rk));
    G2XYZZ piB = xyzz_add(xyzz_add_affine(B2, zk->beta2), xyzz_mul(d2, sk));
    G1XYZZ piB1 = xyzz_add(xyzz_add_affine(B1, zk->beta1), xyzz_mul(d1, sk));
    Fr rf = fp_from_std<FrParams>(rk), sf = fp_from_std<FrParams>(sk);
    uint32_t nrs[8]; fp_to_std<FrParams>(nrs, Fr::zero() - rf * sf);
    G1XYZZ piC = xyzz_add(xyzz_add(C, H), xyzz_add(xyzz_add(xyzz_mul(piA, sk), xyzz_mul(piB1, rk)), xyzz_mul(d1, nrs)));
    g1_to_std(proof, xyzz_to_affine(piA)); g2_to_std(proof + 64, xyzz_to_affine(piB)); g1_to_std(proof + 192, xyzz_to_affine(piC));
}

extern "C" int zkc_prove_dev(zkc_zkey* zk, const void* d_wtns, uint32_t nWitness, const uint8_t r32[32], const uint8_t s32[32],
                             uint8_t proof[256], uint8_t* public_out) {
    if (!zk || !d_wtns || !r32 || !s32 || !proof) return zkc_fail(zk ? zk->ctx : nullptr, ZKC_ERR_BAD_ARG, "zkc_prove_dev: bad argument");
    zkc_ctx* ctx = zk->ctx;
    if (nWitness != zk->nVars) return zkc_fail(ctx, ZKC_ERR_INVALID_WITNESS_LENGTH, "Invalid witness length. Circuit: " + std::to_string(zk->nVars) + ", witness: " + std::to_string(nWitness));
    uint32_t t[8]; memcpy(t, r32, 32); if (!fp_std_lt_p<FrParams>(t)) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "r >= field order");
    memcpy(t, s32, 32); if (!fp_std_lt_p<FrParams>(t)) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "s >= field order");
    ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t* w = (const uint32_t*)d_wtns;
    int rc = h_evals_dev(zk, w); if (rc) return rc;
    if ((rc = msm_g1_run(zk, zk->d_A, w, zk->nVars, 0))) return rc;
    if ((rc = msm_g1_run(zk, zk->d_B1, w, zk->nVars, 1))) return rc;
    if ((rc = msm_g2_run(zk, zk->d_B2, w, zk->nVars, 2))) return rc;
    if ((rc = msm_g1_run(zk, zk->d_C, w + 8ull * (zk->nPub + 1), zk->nVars - zk->nPub - 1, 3))) return rc;
    if ((rc = msm_g1_run(zk, zk->d_H, zk->d_p, zk->n, 4))) return rc;
    ZKC_HIP_CHECK(ctx, hipMemcpyAsync(zk->h_results, zk->d_results, 5 * sizeof(G2XYZZ), hipMemcpyDeviceToHost, ctx->stream));
    if (public_out) ZKC_HIP_CHECK(ctx, hipMemcpyAsync(public_out, w + 8, 32ull * zk->nPub, hipMemcpyDeviceToHost, ctx->stream));
    ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    finalize_proof(zk, (const G2XYZZ*)zk->h_results, r32, s32, proof);
    return ZKC_OK;
}

extern "C" int zkc_prove(zkc_zkey* zk, const void* wtns, uint32_t nWitness, const uint8_t r32[32], const uint8_t s32[32],
                         uint8_t proof[256], uint8_t* public_out) {
    if (!zk || !wtns) return zkc_fail(zk ? zk->ctx : nullptr, ZKC_ERR_BAD_ARG, "zkc_prove: bad argument");
    zkc_ctx* ctx = zk->ctx;
    if (nWitness != zk->nVars) return zkc_fail(ctx, ZKC_ERR_INVALID_WITNESS_LENGTH, "Invalid witness length. Circuit: " + std::to_string(zk->nVars) + ", witness: " + std::to_string(nWitness));
    ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int rc = zkc_ensure(ctx, &ctx->d_scratch_out, &ctx->scratch_out_sz, 32ull * nWitness); if (rc) return rc;
    ZKC_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_scratch_out, wtns, 32ull * nWitness, hipMemcpyHostToDevice, ctx->stream));
    return zkc_prove_dev(zk, ctx->d_scratch_out, nWitness, r32, s32, proof, public_out);
}
