// zkc_msm.hip -- K4/K5/K6/K8: BN254 G1 / G2 multi-scalar multiplication (Pippenger bucket method) for MI355X.
//
// Replaces G1.multiExpAffine / G2.multiExpAffine of ffjavascript/wasmcurves (snarkjs groth16_prove.js, reached from
// ts_inputs/src/example.ts:358-362) and rapidsnark's multiexp (zk_census_test.go:89).
//
// The proving key is constant for the life of a context, so zkc_zkey_load pre-shifts every base once:
// T[w][i] = 2^(c w) P_i.  Per MSM:
//   K4  zkc_msm_digits   scalar -> 20 signed 13-bit digits; emits (bucket = w*4096 + |d|-1, entry = w*count+i | sign)
//       rocPRIM radix sort of the (bucket, entry) pairs, zkc_msm_offsets = bucket segment boundaries
//   K5  zkc_msm_accumulate  one lane per bucket, XYZZ += affine (8M+2S) over its segment, 64-B gathers from T;
//       zkc_msm_heavy       buckets with more than MSM_HEAVY entries (witness bits -> digit 1) : one block each
//   K6  zkc_msm_reduce   running sums over groups of 32 buckets, group offset by a small double-and-add, block tree
//       zkc_msm_final    sums the block partials.  No window doublings are left because the bases are pre-shifted.
#include "zkc_prover.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>

namespace zkc {

template <class F> struct PointIO;
template <> struct PointIO<Fq> {
    static __device__ __forceinline__ Affine<Fq> load(const Affine<Fq>* p) {
        const uint4* d = reinterpret_cast<const uint4*>(p); uint4 a = d[0], b = d[1], c = d[2], e = d[3];
        Affine<Fq> r;
        r.x.v[0] = a.x; r.x.v[1] = a.y; r.x.v[2] = a.z; r.x.v[3] = a.w; r.x.v[4] = b.x; r.x.v[5] = b.y; r.x.v[6] = b.z; r.x.v[7] = b.w;
        r.y.v[0] = c.x; r.y.v[1] = c.y; r.y.v[2] = c.z; r.y.v[3] = c.w; r.y.v[4] = e.x; r.y.v[5] = e.y; r.y.v[6] = e.z; r.y.v[7] = e.w;
        return r;
    }
};
template <> struct PointIO<Fq2> {
    static __device__ __forceinline__ Affine<Fq2> load(const Affine<Fq2>* p) {
        const Affine<Fq>* q = reinterpret_cast<const Affine<Fq>*>(p);
        Affine<Fq> lo = PointIO<Fq>::load(q), hi = PointIO<Fq>::load(q + 1);
        return {{lo.x, lo.y}, {hi.x, hi.y}};        // memory order x.c0, x.c1, y.c0, y.c1
    }
};

// ---- K4 ----
extern "C" __global__ void __launch_bounds__(256)
zkc_msm_digits(const uint32_t* __restrict__ scalars, uint32_t count, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint4* sp = reinterpret_cast<const uint4*>(scalars + 8 * (size_t)i); uint4 a = sp[0], b = sp[1];
    const uint32_t s[9] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, 0};
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < MSM_NW; w++) {
        const int bit = w * MSM_C, li = bit >> 5, sh = bit & 31;
        uint64_t two = (li < 8) ? ((uint64_t)s[li] | ((uint64_t)s[li + 1] << 32)) : 0;
        uint32_t d = (uint32_t)((two >> sh) & ((1u << MSM_C) - 1)) + carry;
        uint32_t neg = 0;
        if (d > (uint32_t)MSM_HALF) { d = (1u << MSM_C) - d; neg = 1; carry = 1; } else carry = 0;
        keys[(size_t)w * count + i] = d ? (uint32_t)(w * MSM_HALF) + d - 1 : (uint32_t)MSM_NB;
        vals[(size_t)w * count + i] = ((uint32_t)w * count + i) | (neg << 31);
    }
}
extern "C" __global__ void __launch_bounds__(256)
zkc_msm_offsets(const uint32_t* __restrict__ keys_sorted, uint32_t total, uint32_t* __restrict__ off, uint32_t* __restrict__ heavy_count) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) *heavy_count = 0;
    if (b > (uint32_t)MSM_NB) return;
    uint32_t lo = 0, hi = total;                       // first position with key >= b
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (keys_sorted[mid] < b) lo = mid + 1; else hi = mid; }
    off[b] = lo;
}

// cold-path group operations are kept out of line (one copy per field) to bound code size and compile time
template <class F> __device__ __noinline__ XYZZ<F> add_ni(const XYZZ<F>& a, const XYZZ<F>& b) { return xyzz_add(a, b); }
template <class F> __device__ __noinline__ XYZZ<F> dbl_ni(const XYZZ<F>& a) { return xyzz_dbl(a); }

// ---- K5 ----

template <class F>
__global__ void __launch_bounds__(128)
zkc_msm_accumulate(const Affine<F>* __restrict__ table, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ off,
                   XYZZ<F>* __restrict__ buckets, uint32_t* __restrict__ heavy, uint32_t* __restrict__ heavy_count) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= (uint32_t)MSM_NB) return;
    const uint32_t s = off[b], e = off[b + 1];
    XYZZ<F> acc = XYZZ<F>::inf();
    if (e - s > (uint32_t)MSM_HEAVY) {
        uint32_t k = atomicAdd(heavy_count, 1u);
        if (k < (uint32_t)MSM_MAX_HEAVY) { heavy[k] = b; buckets[b] = acc; return; }
        // list full: fall through and do it here (slow but correct)
    }
    for (uint32_t j = s; j < e; j++) {
        const uint32_t v = vals[j];
        Affine<F> p = PointIO<F>::load(table + (v & 0x7fffffffu));
        if (v >> 31) p.y = fp_neg(p.y);
        acc = xyzz_add_affine(acc, p);
    }
    buckets[b] = acc;
}
template <class F>
__global__ void __launch_bounds__(256)
zkc_msm_heavy(const Affine<F>* __restrict__ table, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ off,
              XYZZ<F>* __restrict__ buckets, const uint32_t* __restrict__ heavy, const uint32_t* __restrict__ heavy_count) {
    extern __shared__ uint4 lds4[];
    XYZZ<F>* sh = reinterpret_cast<XYZZ<F>*>(lds4);
    uint32_t nh = *heavy_count; if (nh > (uint32_t)MSM_MAX_HEAVY) nh = MSM_MAX_HEAVY;
    for (uint32_t h = blockIdx.x; h < nh; h += gridDim.x) {
        const uint32_t b = heavy[h], s = off[b], e = off[b + 1];
        XYZZ<F> acc = XYZZ<F>::inf();
        for (uint32_t j = s + threadIdx.x; j < e; j += blockDim.x) {
            const uint32_t v = vals[j];
            Affine<F> p = PointIO<F>::load(table + (v & 0x7fffffffu));
            if (v >> 31) p.y = fp_neg(p.y);
            acc = xyzz_add_affine(acc, p);
        }
        sh[threadIdx.x] = acc; __syncthreads();
        for (int st = blockDim.x / 2; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) sh[threadIdx.x] = add_ni(sh[threadIdx.x], sh[threadIdx.x + st]);
            __syncthreads();
        }
        if (threadIdx.x == 0) buckets[b] = sh[0];
        __syncthreads();
    }
}

// ---- K6 ----  sum over all windows of sum_d d * B[w][d]; thread = group of MSM_GROUP consecutive buckets of one window
template <class F>
__global__ void __launch_bounds__(64)
zkc_msm_reduce(const XYZZ<F>* __restrict__ buckets, XYZZ<F>* __restrict__ partial) {
    extern __shared__ uint4 lds4[];
    XYZZ<F>* sh = reinterpret_cast<XYZZ<F>*>(lds4);
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;           // group id
    const uint32_t ngroups = MSM_NB / MSM_GROUP;
    XYZZ<F> contrib = XYZZ<F>::inf();
    if (g < ngroups) {
        const uint32_t first = g * MSM_GROUP;                            // bucket index; digit value = (first % HALF) + k + 1
        const uint32_t base = first % MSM_HALF;                          // digits base+1 .. base+GROUP
        XYZZ<F> run = XYZZ<F>::inf(), loc = XYZZ<F>::inf();
        for (int k = MSM_GROUP - 1; k >= 0; k--) { run = add_ni(run, buckets[first + k]); loc = add_ni(loc, run); }
        // contribution = loc + base * run   (base < 2^12)
        XYZZ<F> sc = XYZZ<F>::inf();
        for (int bit = MSM_C - 2; bit >= 0; bit--) { sc = dbl_ni(sc); if ((base >> bit) & 1) sc = add_ni(sc, run); }
        contrib = add_ni(loc, sc);
    }
    sh[threadIdx.x] = contrib; __syncthreads();
    for (int st = blockDim.x / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] = add_ni(sh[threadIdx.x], sh[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
template <class F>
__global__ void __launch_bounds__(64)
zkc_msm_final(const XYZZ<F>* __restrict__ partial, int nparts, XYZZ<F>* __restrict__ result) {
    extern __shared__ uint4 lds4[];
    XYZZ<F>* sh = reinterpret_cast<XYZZ<F>*>(lds4);
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) acc = add_ni(acc, partial[i]);
    sh[threadIdx.x] = acc; __syncthreads();
    for (int st = blockDim.x / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] = add_ni(sh[threadIdx.x], sh[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) *result = sh[0];
}

// ---- one-time base table: table[w][i] = 2^c * table[w-1][i] ----
template <class F>
__global__ void __launch_bounds__(128)
zkc_msm_shift_bases(const Affine<F>* __restrict__ prev, Affine<F>* __restrict__ next, uint32_t count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Affine<F> a = PointIO<F>::load(prev + i);
    XYZZ<F> p = xyzz_dbl_affine(a);
    for (int k = 1; k < MSM_C; k++) p = dbl_ni(p);
    next[i] = xyzz_to_affine(p);
}

template <class F>
static int precompute(zkc_ctx* ctx, uint32_t count, Affine<F>* d_table) {
    for (int w = 1; w < MSM_NW; w++) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(zkc_msm_shift_bases<F>), dim3((count + 127) / 128), dim3(128), 0, ctx->stream,
                           d_table + (size_t)(w - 1) * count, d_table + (size_t)w * count, count);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return zkc_fail(ctx, ZKC_ERR_HIP, std::string("zkc_msm_shift_bases: ") + hipGetErrorString(e));
    }
    return ZKC_OK;
}
int msm_precompute_g1(zkc_ctx* ctx, const G1Affine*, uint32_t count, G1Affine* d_table) { return precompute<Fq>(ctx, count, d_table); }
int msm_precompute_g2(zkc_ctx* ctx, const G2Affine*, uint32_t count, G2Affine* d_table) { return precompute<Fq2>(ctx, count, d_table); }

static const bool g_debug_sync = getenv("ZKC_DEBUG_SYNC") != nullptr;   // serialise + log every launch (diagnostics only)
#define ZKC_LAUNCH_CHECK(zk, name)                                                                         \
    do { hipError_t _e = hipGetLastError(); if (_e != hipSuccess)                                          \
        return zkc_fail((zk)->ctx, ZKC_ERR_HIP, std::string(name ": ") + hipGetErrorString(_e));           \
        if (g_debug_sync) { fprintf(stderr, "[zkc] %s launched\n", name); fflush(stderr);                  \
            _e = hipStreamSynchronize((zk)->ctx->stream);                                                  \
            fprintf(stderr, "[zkc] %s done: %s\n", name, hipGetErrorString(_e)); fflush(stderr); } } while (0)

template <class F>
static int msm_run(zkc_zkey* zk, const Affine<F>* table, const uint32_t* d_scalars, uint32_t count, int slot) {
    zkc_ctx* ctx = zk->ctx; hipStream_t st = ctx->stream;
    const uint32_t total = count * MSM_NW;
    constexpr bool kG2 = sizeof(F) == sizeof(Fq2);
    const uint64_t alg_bytes = (uint64_t)count * (sizeof(Affine<F>) + 32);      // bases + scalars of this MSM
    uint32_t* heavy_count = zk->d_heavy + MSM_MAX_HEAVY;
    {
    zkc_prof_scope _ps(ctx, ZKC_PROF_MSM_SORT, 0);
    hipLaunchKernelGGL(zkc_msm_digits, dim3((count + 255) / 256), dim3(256), 0, st, d_scalars, count, zk->d_keys, zk->d_vals);
    ZKC_LAUNCH_CHECK(zk, "zkc_msm_digits");
    int end_bit = 1; while ((1u << end_bit) <= (uint32_t)MSM_NB) end_bit++;
    size_t need = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, need, zk->d_keys, zk->d_keys2, zk->d_vals, zk->d_vals2, total, 0, end_bit, st);
    if (e != hipSuccess) return zkc_fail(ctx, ZKC_ERR_HIP, "radix_sort_pairs(size)");
    int rc = zkc_ensure(ctx, &zk->d_sort_tmp, &zk->sort_tmp_sz, need); if (rc) return rc;
    e = rocprim::radix_sort_pairs(zk->d_sort_tmp, need, zk->d_keys, zk->d_keys2, zk->d_vals, zk->d_vals2, total, 0, end_bit, st);
    if (e != hipSuccess) return zkc_fail(ctx, ZKC_ERR_HIP, "radix_sort_pairs");
    ZKC_LAUNCH_CHECK(zk, "radix_sort_pairs");
    hipLaunchKernelGGL(zkc_msm_offsets, dim3((MSM_NB + 1 + 255) / 256), dim3(256), 0, st, zk->d_keys2, total, zk->d_off, heavy_count);
    ZKC_LAUNCH_CHECK(zk, "zkc_msm_offsets");
    }
    XYZZ<F>* buckets = reinterpret_cast<XYZZ<F>*>(zk->d_buckets);
    XYZZ<F>* partial = reinterpret_cast<XYZZ<F>*>(zk->d_partial);
    XYZZ<F>* results = reinterpret_cast<XYZZ<F>*>(reinterpret_cast<uint8_t*>(zk->d_results) + (size_t)slot * sizeof(XYZZ<Fq2>));
    {
    zkc_prof_scope _ps(ctx, kG2 ? ZKC_PROF_MSM_ACC_G2 : ZKC_PROF_MSM_ACC_G1, alg_bytes);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(zkc_msm_accumulate<F>), dim3((MSM_NB + 127) / 128), dim3(128), 0, st, table, zk->d_vals2, zk->d_off,
                       buckets, zk->d_heavy, heavy_count);
    ZKC_LAUNCH_CHECK(zk, "zkc_msm_accumulate");
    }
    zkc_prof_scope _pr(ctx, ZKC_PROF_MSM_REDUCE, 0);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(zkc_msm_heavy<F>), dim3(256), dim3(256), 256 * sizeof(XYZZ<F>), st, table, zk->d_vals2, zk->d_off, buckets,
                       zk->d_heavy, heavy_count);
    ZKC_LAUNCH_CHECK(zk, "zkc_msm_heavy");
    const int ngroups = MSM_NB / MSM_GROUP, nblocks = (ngroups + 63) / 64;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(zkc_msm_reduce<F>), dim3(nblocks), dim3(64), 64 * sizeof(XYZZ<F>), st, buckets, partial);
    ZKC_LAUNCH_CHECK(zk, "zkc_msm_reduce");
    hipLaunchKernelGGL(HIP_KERNEL_NAME(zkc_msm_final<F>), dim3(1), dim3(64), 64 * sizeof(XYZZ<F>), st, partial, nblocks, results);
    ZKC_LAUNCH_CHECK(zk, "zkc_msm_final");
    return ZKC_OK;
}
int msm_g1_run(zkc_zkey* zk, const G1Affine* table, const uint32_t* d_scalars, uint32_t count, int slot) { return msm_run<Fq>(zk, table, d_scalars, count, slot); }
int msm_g2_run(zkc_zkey* zk, const G2Affine* table, const uint32_t* d_scalars, uint32_t count, int slot) { return msm_run<Fq2>(zk, table, d_scalars, count, slot); }

}  // namespace zkc
