// zkc_generic.hip -- the prover's NTT and G1 MSM engines as stand-alone device entry points (no .zkey): what ffjavascript's
// `Fr.fft / Fr.ifft` and `G1.multiExpAffine` are to snarkjs (ts_inputs/src/example.ts:358 -> groth16.prove).  They exist for
// the circuit cannot reach a 2^20 domain, so the "large census" stress figures are a synthetic
// 2^20-point MSM over bases k_i G and a 2^20 NTT (tools/stress.py), checked against the oracle in exponent space.
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include "zkc_prover.h"

using namespace zkc;

namespace {
Fr root_of_unity(int logn) {            // 5^((r-1)/2^28) squared down to order 2^logn
    uint32_t e[8]; for (int i = 0; i < 8; i++) e[i] = FrParams::p[i]; e[0] -= 1;
    for (int i = 0; i < 8; i++) e[i] = (e[i] >> 28) | (i < 7 ? e[i + 1] << 4 : 0);
    Fr g = fp_from_u32<FrParams>(5), w = Fr::one();
    for (int i = 255; i >= 0; i--) { w = w * w; if ((e[i >> 5] >> (i & 31)) & 1) w = w * g; }
    for (int i = 28; i > logn; i--) w = w * w;
    return w;
}

__global__ void __launch_bounds__(256) zkc_fill_fr(Fr* __restrict__ dst, Fr v, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = v;
}
// affine standard form (x, y little endian; all zero = infinity)  <->  affine Montgomery
__global__ void __launch_bounds__(256) zkc_g1_std_to_mont(const uint32_t* __restrict__ in, G1Affine* __restrict__ out, uint32_t n, uint32_t* __restrict__ bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x[8], y[8];
    for (int k = 0; k < 8; k++) { x[k] = in[16 * (size_t)i + k]; y[k] = in[16 * (size_t)i + 8 + k]; }
    if (!fp_std_lt_p<FqParams>(x) || !fp_std_lt_p<FqParams>(y)) { atomicOr(bad, 1u); return; }
    G1Affine a; a.x = fp_from_std<FqParams>(x); a.y = fp_from_std<FqParams>(y);
    if (!a.is_inf() && !(fp_sqr(a.y) == fp_sqr(a.x) * a.x + fp_from_u32<FqParams>(3))) atomicOr(bad, 2u);
    out[i] = a;
}
__global__ void __launch_bounds__(64) zkc_g1_xyzz_to_std(const G1XYZZ* __restrict__ in, uint32_t* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const G1Affine a = xyzz_to_affine(in[i]);
    uint32_t s[8];
    fp_to_std<FqParams>(s, a.x); for (int k = 0; k < 8; k++) out[16 * (size_t)i + k] = s[k];
    fp_to_std<FqParams>(s, a.y); for (int k = 0; k < 8; k++) out[16 * (size_t)i + 8 + k] = s[k];
}
// out[i] = k[i] * P by double-and-add (P: one affine Montgomery point)
__global__ void __launch_bounds__(64) zkc_g1_mul_same_base(G1Affine p, const uint32_t* __restrict__ scalars, uint32_t n, G1XYZZ* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k[8]; for (int q = 0; q < 8; q++) k[q] = scalars[8 * (size_t)i + q];
    out[i] = xyzz_mul(G1XYZZ::from_affine(p), k);
}
}  // namespace

struct zkc_msm {
    zkc_zkey zk;                 // only ctx and d_g1 are used by the MSM pipeline
    MsmWork w; uint32_t n = 0; int c = 0; MsmJobList jl;
};

// In-place-capable NTT over BN254 Fr on `nvec` contiguous vectors of 2^logn elements in MONTGOMERY form (R = 2^256), natural order in
// and out; inverse != 0 computes the inverse transform including the 1/n factor.  d_src != d_dst.  3 <= logn <= 27.
extern "C" int zkc_ntt_dev(zkc_ctx* ctx, const void* d_src, void* d_dst, int logn, int nvec, int inverse) {
    if (!ctx || !d_src || !d_dst || d_src == d_dst || logn < 3 || logn > 27 || nvec <= 0) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "zkc_ntt_dev: bad argument");
    ZKC_LOCK(ctx);
    ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t n = 1u << logn;
    zkc_ctx::TwiddleSet& t = ctx->ntt_tw[logn];     // owned by the context (its lock is held): released with it, never inherited by another device's context
    if (!t.fwd) {
        const Fr w = root_of_unity(logn), wi = fp_inv<FrParams>(w);
        std::vector<Fr> f(n / 2), b(n / 2);
        f[0] = b[0] = Fr::one(); for (uint32_t i = 1; i < n / 2; i++) { f[i] = f[i - 1] * w; b[i] = b[i - 1] * wi; }
        Fr* d_tmp = nullptr; int rc;
        ZKC_HIP_CHECK(ctx, hipMalloc((void**)&d_tmp, (size_t)(n / 2) * sizeof(Fr)));
        ZKC_HIP_CHECK(ctx, hipMalloc(&t.ninv, (size_t)n * sizeof(Fr)));
        ZKC_HIP_CHECK(ctx, hipMemcpy(d_tmp, f.data(), f.size() * sizeof(Fr), hipMemcpyHostToDevice));
        if ((rc = ntt_make_tw29(ctx, d_tmp, n / 2, &t.fwd))) return rc;
        ZKC_HIP_CHECK(ctx, hipMemcpy(d_tmp, b.data(), b.size() * sizeof(Fr), hipMemcpyHostToDevice));
        if ((rc = ntt_make_tw29(ctx, d_tmp, n / 2, &t.inv))) return rc;
        ZKC_HIP_CHECK(ctx, hipFree(d_tmp));
        hipLaunchKernelGGL(zkc_fill_fr, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (Fr*)t.ninv, fp_inv<FrParams>(fp_from_u32<FrParams>(n)), n);
        ZKC_HIP_CHECK(ctx, hipGetLastError());
    }
    zkc_prof_scope _pn(ctx, ZKC_PROF_NTT, (uint64_t)nvec * 2ull * n * 32, ctx->stream);
    return ntt_run(ctx, ctx->stream, (const Fr*)d_src, (Fr*)d_dst, inverse ? t.inv : t.fwd, inverse ? (const Fr*)t.ninv : nullptr, logn, nvec);
}

// d_out[i] = k_i * base for n scalars (device, standard form 32 B each); base and outputs are affine points in standard form (64 B).
extern "C" int zkc_g1_mul_batch_dev(zkc_ctx* ctx, const uint8_t base_std[64], const void* d_scalars, uint32_t n, void* d_out) {
    if (!ctx || !base_std || !d_scalars || !d_out || n == 0) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "zkc_g1_mul_batch_dev: bad argument");
    ZKC_LOCK(ctx);
    ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    uint32_t x[8], y[8]; memcpy(x, base_std, 32); memcpy(y, base_std + 32, 32);
    if (!fp_std_lt_p<FqParams>(x) || !fp_std_lt_p<FqParams>(y)) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "zkc_g1_mul_batch_dev: base coordinate >= q");
    G1Affine p; p.x = fp_from_std<FqParams>(x); p.y = fp_from_std<FqParams>(y);
    G1XYZZ* d_tmp = nullptr;
    ZKC_HIP_CHECK(ctx, hipMalloc((void**)&d_tmp, (size_t)n * sizeof(G1XYZZ)));
    hipLaunchKernelGGL(zkc_g1_mul_same_base, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, p, (const uint32_t*)d_scalars, n, d_tmp);
    hipLaunchKernelGGL(zkc_g1_xyzz_to_std, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, d_tmp, (uint32_t*)d_out, n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_tmp);
    if (e != hipSuccess) return zkc_fail(ctx, ZKC_ERR_HIP, std::string("zkc_g1_mul_batch_dev: ") + hipGetErrorString(e));
    return ZKC_OK;
}

// A fixed set of n G1 bases (device, affine standard form, 64 B each) made resident as pre-shifted window tables.
extern "C" int zkc_msm_g1_load_dev(zkc_ctx* ctx, const void* d_bases_std, uint32_t n, zkc_msm** out) {
    if (!ctx || !d_bases_std || !out || n == 0 || n > (1u << 21)) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "zkc_msm_g1_load_dev: bad argument (1 <= n <= 2^21)");
    ZKC_LOCK(ctx);
    ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    zkc_msm* m = new zkc_msm(); m->zk.ctx = ctx; m->n = n; m->c = msm_c_for(n);
    const int nw = msm_nw(m->c);
    uint32_t* d_bad = nullptr; uint32_t bad = 0; int rc = ZKC_OK;
    auto bail = [&](int code) { if (d_bad) (void)hipFree(d_bad); if (m->zk.d_g1) (void)hipFree(m->zk.d_g1); m->zk.d_g1 = nullptr; msm_work_free(m->w); delete m; return code; };
    if (hipMalloc((void**)&m->zk.d_g1, (size_t)nw * n * sizeof(G1Affine)) != hipSuccess || hipMalloc((void**)&d_bad, 4) != hipSuccess || hipMemset(d_bad, 0, 4) != hipSuccess)
        return bail(zkc_fail(ctx, ZKC_ERR_HIP, "zkc_msm_g1_load_dev: hipMalloc failed"));
    hipLaunchKernelGGL(zkc_g1_std_to_mont, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const uint32_t*)d_bases_std, m->zk.d_g1, n, d_bad);
    if (hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
        return bail(zkc_fail(ctx, ZKC_ERR_HIP, "zkc_msm_g1_load_dev: base conversion failed"));
    if (bad) return bail(zkc_fail(ctx, ZKC_ERR_FORMAT, bad & 1 ? "zkc_msm_g1_load_dev: base coordinate >= q" : "zkc_msm_g1_load_dev: base not on the curve"));
    if ((rc = msm_precompute_g1(ctx, n, m->zk.d_g1, m->c))) return bail(rc);
    if ((rc = msm_work_alloc(ctx, m->w, (size_t)nw * n, (size_t)msm_half(m->c), 1, false))) return bail(rc);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return bail(zkc_fail(ctx, ZKC_ERR_HIP, "zkc_msm_g1_load_dev: table build failed"));
    (void)hipFree(d_bad);
    *out = m;
    return ZKC_OK;
}
// sum_i s_i P_i over the resident bases; d_scalars: n x 32 B standard form (device); out: affine standard form (all zero = infinity)
extern "C" int zkc_msm_g1_dev(zkc_msm* m, const void* d_scalars, uint8_t out[64]) {
    if (!m || !d_scalars || !out) return ZKC_ERR_BAD_ARG;
    zkc_ctx* ctx = m->zk.ctx;
    ZKC_LOCK(ctx);
    ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    m->jl.clear(); m->jl.add((const uint32_t*)d_scalars, nullptr, m->n, 0, m->n, 0, m->c);
    int rc = msm_pass_g1(&m->zk, m->w, m->jl, 0, true, ctx->stream); if (rc) return rc;
    ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const G1Affine a = xyzz_to_affine(*(const G1XYZZ*)m->w.h_results);
    uint32_t s[8]; fp_to_std<FqParams>(s, a.x); memcpy(out, s, 32); fp_to_std<FqParams>(s, a.y); memcpy(out + 32, s, 32);
    return ZKC_OK;
}
extern "C" void zkc_msm_g1_free(zkc_msm* m) {
    if (!m) return;
    ZKC_LOCK(m->zk.ctx);
    (void)hipSetDevice(m->zk.ctx->device); (void)hipStreamSynchronize(m->zk.ctx->stream);
    if (m->zk.d_g1) (void)hipFree(m->zk.d_g1);
    m->zk.d_g1 = nullptr; msm_work_free(m->w);
    delete m;
}
