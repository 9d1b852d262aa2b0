// zkc_msm.hip -- K4/K5/K6/K8: BN254 G1 / G2 multi-scalar multiplication (Pippenger bucket method) for MI355X.
//
// Replaces G1.multiExpAffine / G2.multiExpAffine of ffjavascript/wasmcurves (snarkjs groth16_prove.js, reached from
// ts_inputs/src/example.ts:358-362) and rapidsnark's multiexp (zk_census_test.go:89).
//
// The proving key is constant for the life of a context, so zkc_zkey_load pre-shifts every base once
// (T[w][i] = 2^(c w) P_i): windows combine with plain additions, no doublings, and a digit d of ANY window lands in
// the same bucket d -- a job has 2^(c-1) signed-digit buckets in total (c = 17 for H, 12 for the witness sections).
// A pipeline PASS runs a list of jobs (sections x proofs in flight) through the same launches:
//   K4  zkc_msm_sort.hip    scalar -> signed c-bit digits -> entries (sign | table row) grouped by bucket inside each job's own region
//       (two counting passes per job, hand-written: no key array, no device-wide sort), bucket boundaries, segments of <= 128 entries
//       and their order by decreasing length
//   K5  zkc_msm_accumulate29[_g2]  one lane per SEGMENT (<= 128 entries of one bucket): XYZZ += affine (8M + 2S)
//       in radix-2^29 coordinates (zkc_f29*.h) with 64-byte gathers from T.  Cutting buckets into segments keeps lanes
//       balanced when many scalars repeat (witness bits; the circuit has only ~4.5k distinct values among 82k wires).
//   K6  zkc_msm_merge       buckets of more than 8 segments: a wave each, shuffle tree
//       zkc_msm_window[29]  one wave per virtual window (4096 / 2048 consecutive buckets of a job in a full pass, 1024 / 256 or 256 / 64 in smaller ones): per-lane running
//       sums, a suffix scan across the 64 lanes in LDS, x per, tree sum  ->  W = sum_j j B_j and S = sum_j B_j
//       zkc_msm_final       one workgroup per job: sum_k W_k + vw sum_k k S_k over its virtual windows.
#include <cstdio>
#include <cstddef>
#include <ctime>
#include <cstdlib>
#include <cstring>
#include <string.h>
#include <vector>
#include "zkc_prover.h"
#include "zkc_f29.h"
#include "zkc_f29_g1.h"
#include "zkc_f29_g2.h"

namespace zkc {

template <class F> struct PointIO;
template <class PP> struct PointIO<Fp<PP>> {
    typedef Fp<PP> Fq;
    static __device__ __forceinline__ Affine<Fq> load(const Affine<Fq>* p) {
        const uint4* d = reinterpret_cast<const uint4*>(p); uint4 a = d[0], b = d[1], c = d[2], e = d[3];
        Affine<Fq> r;
        r.x.v[0] = a.x; r.x.v[1] = a.y; r.x.v[2] = a.z; r.x.v[3] = a.w; r.x.v[4] = b.x; r.x.v[5] = b.y; r.x.v[6] = b.z; r.x.v[7] = b.w;
        r.y.v[0] = c.x; r.y.v[1] = c.y; r.y.v[2] = c.z; r.y.v[3] = c.w; r.y.v[4] = e.x; r.y.v[5] = e.y; r.y.v[6] = e.z; r.y.v[7] = e.w;
        return r;
    }
};
template <class BB> struct PointIO<Fq2T<BB>> {
    typedef Fq2T<BB> Fq2; typedef BB Fq;
    static __device__ __forceinline__ Affine<Fq2> load(const Affine<Fq2>* p) {
        const Affine<Fq>* q = reinterpret_cast<const Affine<Fq>*>(p);
        Affine<Fq> lo = PointIO<Fq>::load(q), hi = PointIO<Fq>::load(q + 1);
        return {{lo.x, lo.y}, {hi.x, hi.y}};        // memory order x.c0, x.c1, y.c0, y.c1
    }
};

// A bucket of L entries cut into k = ceil(L / seg) segments is split EVENLY: segment i covers [floor(i L / k), floor((i + 1) L / k)).
__device__ __forceinline__ void msm_seg_range(uint32_t L, uint32_t k, uint32_t i, uint32_t& lo, uint32_t& hi) {
    lo = (uint32_t)(((uint64_t)i * L) / k); hi = (uint32_t)(((uint64_t)(i + 1) * L) / k);
}

// ---- K5, G1: the same segment walk with the accumulator kept in radix 2^29 (zkc_f29.h, zkc_f29_g1.h) ----
template <int MINW>
__global__ void __launch_bounds__(128, MINW)
zkc_msm_accumulate29(const Affine<Fq>* __restrict__ table_all, const MsmJobList* __restrict__ jlp, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ off,
                     const uint32_t* __restrict__ bcnt, const uint32_t* __restrict__ segoff, const uint32_t* __restrict__ seg2bucket, const uint32_t* __restrict__ perm, uint32_t nbuckets,
                     XYZZ<Fq>* __restrict__ partial, uint32_t max_segments) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t nseg = segoff[nbuckets]; if (nseg > max_segments) nseg = max_segments;
    if (gid >= nseg) return;
    const uint32_t s = perm[gid];                       // segments by decreasing length: the lanes of a wave finish together
    const uint32_t b = seg2bucket[s];
    uint32_t lo, hi; msm_seg_range(bcnt[b], segoff[b + 1] - segoff[b], s - segoff[b], lo, hi);
    const uint32_t start = off[b] + lo, end = off[b] + hi;
    uint32_t bd, bj; jlp->decode(b, bd, bj);
    const Affine<Fq>* __restrict__ table = table_all + jlp->job[bj].tbl_off;      // a segment belongs to one job
    constexpr uint32_t rowmask = 0x7fffffffu;                                     // entry word: sign | table row
    Acc29 acc; bool inf = true;
    uint32_t v = vals[start];
    Affine<Fq> p = PointIO<Fq>::load(table + (v & rowmask));
    for (uint32_t j = start; j < end; j++) {
        const uint32_t vn = (j + 1 < end) ? vals[j + 1] : v;
        Affine<Fq> pn = PointIO<Fq>::load(table + (vn & rowmask));      // next gather in flight during this addition
        if (!p.is_inf()) {
            if (v >> 31) p.y = fp_neg(p.y);
            uint32_t x2[9], y2[9];
            f29_from_fp_shl5(x2, p.x.v); f29_from_fp_shl5(y2, p.y.v);
            bool same_y = false;
            if (inf) {
                f29_mul<FqParams>(acc.X, x2, F29K<FqParams>::one.l); f29_mul<FqParams>(acc.Y, y2, F29K<FqParams>::one.l);
#pragma unroll
                for (int k = 0; k < 9; k++) acc.ZZ[k] = acc.ZZZ[k] = F29K<FqParams>::one.l[k];
                inf = false;
            } else if (!f29_madd(acc, x2, y2, same_y)) {
                if (same_y) {                                               // the bucket holds this very point: double it (rare; generic code)
                    const XYZZ<Fq> d = xyzz_dbl_affine(p);
                    f29_enter_fq(acc.X, d.X.v); f29_enter_fq(acc.Y, d.Y.v); f29_enter_fq(acc.ZZ, d.ZZ.v); f29_enter_fq(acc.ZZZ, d.ZZZ.v);
                } else inf = true;                                          // P + (-P)
            }
        }
        v = vn; p = pn;
    }
    // [r2] tried and not kept: leaving the segment sum in its nine-limb form (144 B) for the merge / window kernels instead of this round trip through the
    // canonical form (970 instructions here, 110 to slice it again there: 0.7 % of a pass on paper) -- 3097 / 3091 against 3090 / 3104 proofs/s, nothing
    XYZZ<Fq> out = XYZZ<Fq>::inf();
    if (!inf) { out.X = f29_to_fp<FqParams>(acc.X); out.Y = f29_to_fp<FqParams>(acc.Y); out.ZZ = f29_to_fp<FqParams>(acc.ZZ); out.ZZZ = f29_to_fp<FqParams>(acc.ZZZ); }
    partial[s] = out;
}

// ---- K5, G2: radix-2^29 accumulator over Fq2 (zkc_f29_g2.h).  The table is a second copy of the pre-shifted G2 bases in R' form:
// 60 words per point = x (9 + 9 limbs, word 18 = 1 for the point at infinity, word 19 pad), y, -y, so a signed digit only picks
// which 80-byte chunk to load. ----
constexpr int G2T29_WORDS = 60;
__global__ void __launch_bounds__(128)
zkc_g2_table29(const Affine<Fq2>* __restrict__ tbl, uint32_t* __restrict__ out, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const Affine<Fq2> a = PointIO<Fq2>::load(tbl + i);
    uint32_t* o = out + i * G2T29_WORDS;
    const Fq2 ny = fp_neg(a.y);
    f29_enter_fq(o, a.x.c0.v); f29_enter_fq(o + 9, a.x.c1.v); o[18] = a.is_inf() ? 1u : 0u; o[19] = 0;
    f29_enter_fq(o + 20, a.y.c0.v); f29_enter_fq(o + 29, a.y.c1.v); o[38] = o[39] = 0;
    f29_enter_fq(o + 40, ny.c0.v); f29_enter_fq(o + 49, ny.c1.v); o[58] = o[59] = 0;
}
int msm_g2_table29(zkc_ctx* ctx, const G2Affine* d_table, uint32_t* d_out, size_t count) {
    hipLaunchKernelGGL(zkc_g2_table29, dim3((unsigned)((count + 127) / 128)), dim3(128), 0, ctx->stream, d_table, d_out, count);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return zkc_fail(ctx, ZKC_ERR_HIP, std::string("zkc_g2_table29: ") + hipGetErrorString(e));
    return ZKC_OK;
}
struct G2Chunk { uint32_t w[20]; };
__device__ __forceinline__ G2Chunk g2_chunk_load(const uint32_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p); G2Chunk c;
#pragma unroll
    for (int i = 0; i < 5; i++) { const uint4 v = q[i]; c.w[4 * i] = v.x; c.w[4 * i + 1] = v.y; c.w[4 * i + 2] = v.z; c.w[4 * i + 3] = v.w; }
    return c;
}
template <int MINW>
__global__ void __launch_bounds__(128, MINW)
zkc_msm_accumulate29_g2(const uint32_t* __restrict__ table29_all, const MsmJobList* __restrict__ jlp, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ off,
                        const uint32_t* __restrict__ bcnt, const uint32_t* __restrict__ segoff, const uint32_t* __restrict__ seg2bucket, const uint32_t* __restrict__ perm, uint32_t nbuckets,
                        XYZZ<Fq2>* __restrict__ partial, uint32_t max_segments) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t nseg = segoff[nbuckets]; if (nseg > max_segments) nseg = max_segments;
    if (gid >= nseg) return;
    const uint32_t s = perm[gid];                       // segments by decreasing length: the lanes of a wave finish together
    const uint32_t b = seg2bucket[s];
    uint32_t lo, hi; msm_seg_range(bcnt[b], segoff[b + 1] - segoff[b], s - segoff[b], lo, hi);
    const uint32_t start = off[b] + lo, end = off[b] + hi;
    uint32_t bd, bj; jlp->decode(b, bd, bj);
    const uint32_t* __restrict__ table29 = table29_all + (size_t)jlp->job[bj].tbl_off * G2T29_WORDS;
    constexpr uint32_t rowmask = 0x7fffffffu;
    Acc29G2 acc; bool inf = true;
    uint32_t v = vals[start];
    const uint32_t* pp = table29 + (size_t)(v & rowmask) * G2T29_WORDS;
    G2Chunk cx = g2_chunk_load(pp), cy = g2_chunk_load(pp + ((v >> 31) ? 40 : 20));
    for (uint32_t j = start; j < end; j++) {
        const uint32_t vn = (j + 1 < end) ? vals[j + 1] : v;
        const uint32_t* pn = table29 + (size_t)(vn & rowmask) * G2T29_WORDS;
        const G2Chunk nx = g2_chunk_load(pn), ny = g2_chunk_load(pn + ((vn >> 31) ? 40 : 20));      // next gather in flight during this addition
        if (!cx.w[18]) {
            F2x29 x2, y2;
#pragma unroll
            for (int k = 0; k < 9; k++) { x2.c0[k] = cx.w[k]; x2.c1[k] = cx.w[9 + k]; y2.c0[k] = cy.w[k]; y2.c1[k] = cy.w[9 + k]; }
            bool same_y = false;
            if (inf) {
                acc.X = x2; acc.Y = y2;
#pragma unroll
                for (int k = 0; k < 9; k++) { acc.ZZ.c0[k] = acc.ZZZ.c0[k] = F29K<FqParams>::one.l[k]; acc.ZZ.c1[k] = acc.ZZZ.c1[k] = 0; }
                inf = false;
            } else if (!f29g2_madd(acc, x2, y2, same_y)) {
                if (same_y) {                                               // the bucket holds this very point: double it (rare; generic code)
                    Affine<Fq2> a; a.x = {f29_to_fp<FqParams>(x2.c0), f29_to_fp<FqParams>(x2.c1)}; a.y = {f29_to_fp<FqParams>(y2.c0), f29_to_fp<FqParams>(y2.c1)};
                    const XYZZ<Fq2> d = xyzz_dbl_affine(a);
                    f29_enter_fq(acc.X.c0, d.X.c0.v); f29_enter_fq(acc.X.c1, d.X.c1.v); f29_enter_fq(acc.Y.c0, d.Y.c0.v); f29_enter_fq(acc.Y.c1, d.Y.c1.v);
                    f29_enter_fq(acc.ZZ.c0, d.ZZ.c0.v); f29_enter_fq(acc.ZZ.c1, d.ZZ.c1.v); f29_enter_fq(acc.ZZZ.c0, d.ZZZ.c0.v); f29_enter_fq(acc.ZZZ.c1, d.ZZZ.c1.v);
                } else inf = true;                                          // P + (-P)
            }
        }
        v = vn; cx = nx; cy = ny;
    }
    XYZZ<Fq2> out = XYZZ<Fq2>::inf();
    if (!inf) {
        out.X = {f29_to_fp<FqParams>(acc.X.c0), f29_to_fp<FqParams>(acc.X.c1)}; out.Y = {f29_to_fp<FqParams>(acc.Y.c0), f29_to_fp<FqParams>(acc.Y.c1)};
        out.ZZ = {f29_to_fp<FqParams>(acc.ZZ.c0), f29_to_fp<FqParams>(acc.ZZ.c1)}; out.ZZZ = {f29_to_fp<FqParams>(acc.ZZZ.c0), f29_to_fp<FqParams>(acc.ZZZ.c1)};
    }
    partial[s] = out;
}

// ---- [r4] K5, G2, with the NEXT entry's table row fetched by the LDS-DMA path (global_load_lds_dwordx4: memory -> LDS without passing through VGPRs).
// The kernel above keeps the next entry's two 80-byte chunks in 40 registers for the whole of the current addition (367 VGPRs: one wave per SIMD, VALU-busy 0.70 -- the
// one heavy kernel of a pass with idle issue slots).  Here the prefetch lives in LDS (160 bytes per lane, laid out [piece][lane]: lane i of a wave writes
// M0 base + 16 i), the lane copies it into registers only when the addition starts, and the register budget is capped for two waves per SIMD.  One wave per workgroup:
// nothing is shared, no barrier.
constexpr int G2DMA_T = 64;
template <int MINW>
__global__ void __launch_bounds__(G2DMA_T, MINW)
zkc_msm_accumulate29_g2_dma(const uint32_t* __restrict__ table29_all, const MsmJobList* __restrict__ jlp, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ off,
                            const uint32_t* __restrict__ bcnt, const uint32_t* __restrict__ segoff, const uint32_t* __restrict__ seg2bucket, const uint32_t* __restrict__ perm, uint32_t nbuckets,
                            XYZZ<Fq2>* __restrict__ partial, uint32_t max_segments) {
    __shared__ uint4 stage[10][G2DMA_T];
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x;
    uint32_t nseg = segoff[nbuckets]; if (nseg > max_segments) nseg = max_segments;
    if (gid >= nseg) return;
    const uint32_t s = perm[gid];
    const uint32_t b = seg2bucket[s];
    uint32_t lo, hi; msm_seg_range(bcnt[b], segoff[b + 1] - segoff[b], s - segoff[b], lo, hi);
    const uint32_t start = off[b] + lo, end = off[b] + hi;
    uint32_t bd, bj; jlp->decode(b, bd, bj);
    const uint32_t* __restrict__ table29 = table29_all + (size_t)jlp->job[bj].tbl_off * G2T29_WORDS;
    constexpr uint32_t rowmask = 0x7fffffffu;
    typedef const __attribute__((address_space(1))) void* gptr; typedef __attribute__((address_space(3))) void* lptr;
    auto fetch = [&](uint32_t v) {
        const uint32_t* px = table29 + (size_t)(v & rowmask) * G2T29_WORDS;
        const uint32_t* py = px + ((v >> 31) ? 40 : 20);
#pragma unroll
        for (int k = 0; k < 5; k++) __builtin_amdgcn_global_load_lds((gptr)(px + 4 * k), (lptr)&stage[k][0], 16, 0, 0);
#pragma unroll
        for (int k = 0; k < 5; k++) __builtin_amdgcn_global_load_lds((gptr)(py + 4 * k), (lptr)&stage[5 + k][0], 16, 0, 0);
    };
    Acc29G2 acc; bool inf = true;
    uint32_t v = vals[start];
    fetch(v);
    for (uint32_t j = start; j < end; j++) {
        const uint32_t vn = (j + 1 < end) ? vals[j + 1] : v;
        F2x29 x2, y2; uint32_t isinf;
        {
            uint32_t w[40];
#pragma unroll
            for (int k = 0; k < 10; k++) { const uint4 q = stage[k][lane]; w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w; }
#pragma unroll
            for (int k = 0; k < 9; k++) { x2.c0[k] = w[k]; x2.c1[k] = w[9 + k]; y2.c0[k] = w[20 + k]; y2.c1[k] = w[29 + k]; }
            isinf = w[18];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the row is in registers before its LDS slot is handed to the next fetch
        if (j + 1 < end) fetch(vn);                                    // in flight during this addition
        if (!isinf) {
            bool same_y = false;
            if (inf) {
                acc.X = x2; acc.Y = y2;
#pragma unroll
                for (int k = 0; k < 9; k++) { acc.ZZ.c0[k] = acc.ZZZ.c0[k] = F29K<FqParams>::one.l[k]; acc.ZZ.c1[k] = acc.ZZZ.c1[k] = 0; }
                inf = false;
            } else if (!f29g2_madd_lean(acc, x2, y2, same_y)) {
                if (same_y) {                                               // the bucket holds this very point: double it (rare; generic code)
                    Affine<Fq2> a; a.x = {f29_to_fp<FqParams>(x2.c0), f29_to_fp<FqParams>(x2.c1)}; a.y = {f29_to_fp<FqParams>(y2.c0), f29_to_fp<FqParams>(y2.c1)};
                    const XYZZ<Fq2> d = xyzz_dbl_affine(a);
                    f29_enter_fq(acc.X.c0, d.X.c0.v); f29_enter_fq(acc.X.c1, d.X.c1.v); f29_enter_fq(acc.Y.c0, d.Y.c0.v); f29_enter_fq(acc.Y.c1, d.Y.c1.v);
                    f29_enter_fq(acc.ZZ.c0, d.ZZ.c0.v); f29_enter_fq(acc.ZZ.c1, d.ZZ.c1.v); f29_enter_fq(acc.ZZZ.c0, d.ZZZ.c0.v); f29_enter_fq(acc.ZZZ.c1, d.ZZZ.c1.v);
                } else inf = true;                                          // P + (-P)
            }
        }
        v = vn;
    }
    XYZZ<Fq2> out = XYZZ<Fq2>::inf();
    if (!inf) {
        out.X = {f29_to_fp<FqParams>(acc.X.c0), f29_to_fp<FqParams>(acc.X.c1)}; out.Y = {f29_to_fp<FqParams>(acc.Y.c0), f29_to_fp<FqParams>(acc.Y.c1)};
        out.ZZ = {f29_to_fp<FqParams>(acc.ZZ.c0), f29_to_fp<FqParams>(acc.ZZ.c1)}; out.ZZZ = {f29_to_fp<FqParams>(acc.ZZZ.c0), f29_to_fp<FqParams>(acc.ZZZ.c1)};
    }
    partial[s] = out;
}

// ---- [r3] K5 for a SMALL G2 pass (one to four proofs): half a WAVE per bucket.  With a lane per segment of 16 entries a bucket of ~100 entries is 16 mixed additions in a row,
// up to eight full ones in the window walk and a merge pass for the heavier buckets -- the longest link of a lone proof's G2 chain (0.9 of 1.7 ms).  Here 32 lanes stride over
// the bucket's entries (three or four additions in a row for 100 entries) and a five-step butterfly sums them; 2048 buckets are 1024 waves, one per SIMD, one round.  No segment
// lists.  A bucket of more than 128 entries (the wires that are 1: thousands of entries in digit 1 of the lowest window) is cut into up to `nslice` slices, a half-wave each
// (grid.y), whose sums are the bucket's "segments": the merge kernel adds them.  What the window kernel reads: partial[b nslice + y], segoff[b] = b nslice,
// segcnt[b] = slices in use.  A full pass (192 k buckets) keeps the lane-per-segment kernel: there every lane has work for the whole walk.
__global__ void __launch_bounds__(64)
zkc_msm_bucketwave_g2(const uint32_t* __restrict__ table29_all, const MsmJobList* __restrict__ jlp, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ off,
                      const uint32_t* __restrict__ bcnt, uint32_t* __restrict__ segoff, uint32_t* __restrict__ segcnt, uint32_t* __restrict__ heavy, uint32_t* __restrict__ heavy_count,
                      XYZZ<Fq2>* __restrict__ partial, uint32_t nbuckets, uint32_t nslice) {
    const uint32_t half = threadIdx.x >> 5, lane = threadIdx.x & 31, y = blockIdx.y;
    const uint32_t b = 2 * blockIdx.x + half;
    const uint32_t cnt = b < nbuckets ? bcnt[b] : 0u;
    uint32_t m = (cnt + 127) / 128; if (m > nslice) m = nslice;               // slices in use
    if (b < nbuckets && y == 0 && lane == 0) {
        segoff[b] = b * nslice; segcnt[b] = m;
        if (m > 1) { const uint32_t h = atomicAdd(heavy_count, 1u); if (h < (uint32_t)MSM_MAX_HEAVY) heavy[h] = b; }
    }
    const uint32_t lo = y < m ? (uint32_t)(((uint64_t)cnt * y) / m) : 0u, hi = y < m ? (uint32_t)(((uint64_t)cnt * (y + 1)) / m) : 0u;
    Acc29G2 acc; f29g2_pt_set_inf(acc); bool inf = true;
    if (__ballot(hi > lo) == 0) return;                                      // neither bucket of this wave has a slice y
    if (hi > lo) {
        uint32_t bd, bj; jlp->decode(b, bd, bj);
        const uint32_t* __restrict__ table29 = table29_all + (size_t)jlp->job[bj].tbl_off * G2T29_WORDS;
        constexpr uint32_t rowmask = 0x7fffffffu;
        const uint32_t start = off[b];
        for (uint32_t e = lo + lane; e < hi; e += 32) {
            const uint32_t v = vals[start + e];
            const uint32_t* pp = table29 + (size_t)(v & rowmask) * G2T29_WORDS;
            const G2Chunk cx = g2_chunk_load(pp), cy = g2_chunk_load(pp + ((v >> 31) ? 40 : 20));
            if (cx.w[18]) continue;                        // base at infinity
            F2x29 x2, y2;
#pragma unroll
            for (int k = 0; k < 9; k++) { x2.c0[k] = cx.w[k]; x2.c1[k] = cx.w[9 + k]; y2.c0[k] = cy.w[k]; y2.c1[k] = cy.w[9 + k]; }
            bool same_y = false;
            if (inf) {
                acc.X = x2; acc.Y = y2;
#pragma unroll
                for (int k = 0; k < 9; k++) { acc.ZZ.c0[k] = acc.ZZZ.c0[k] = F29K<FqParams>::one.l[k]; acc.ZZ.c1[k] = acc.ZZZ.c1[k] = 0; }
                inf = false;
            } else if (!f29g2_madd(acc, x2, y2, same_y)) {
                if (same_y) {                              // the lane holds this very point: double it (rare; generic code)
                    Affine<Fq2> a; a.x = {f29_to_fp<FqParams>(x2.c0), f29_to_fp<FqParams>(x2.c1)}; a.y = {f29_to_fp<FqParams>(y2.c0), f29_to_fp<FqParams>(y2.c1)};
                    acc = f29g2_pt_from_xyzz(xyzz_dbl_affine(a));
                } else { f29g2_pt_set_inf(acc); inf = true; }  // P + (-P)
            }
        }
    }
    acc = wave_sum_g2(acc, 16);                            // over the 32 lanes of each half
    if (lane == 0 && hi > lo) partial[(size_t)b * nslice + y] = f29g2_pt_is_inf(acc) ? XYZZ<Fq2>::inf() : f29g2_pt_to_xyzz(acc);
}

// buckets with many segments (repeated witness values; every bucket of a 2^20-point job) are summed by one wave each: lanes stride over
// the segments, then a tree over the lanes in LDS; the result replaces the bucket's first segment.  Radix-2^29 coordinates, one traits
// struct per group.
struct Merge29G1 {
    typedef Fq F; typedef Acc29 Acc;
    static __device__ __forceinline__ void set_inf(Acc& a) { f29_pt_set_inf(a); }
    static __device__ __forceinline__ bool is_inf(const Acc& a) { return f29_pt_is_inf(a); }
    static __device__ __forceinline__ Acc from(const XYZZ<F>& p) { return f29_pt_from_xyzz(p); }
    static __device__ __forceinline__ XYZZ<F> to(const Acc& a) { return f29_pt_to_xyzz(a); }
    static __device__ __forceinline__ void add(Acc& r, const Acc& a, const Acc& b) { f29_pt_add(r, a, b); }
};
struct Merge29G2 {
    typedef Fq2 F; typedef Acc29G2 Acc;
    static __device__ __forceinline__ void set_inf(Acc& a) { f29g2_pt_set_inf(a); }
    static __device__ __forceinline__ bool is_inf(const Acc& a) { return f29g2_pt_is_inf(a); }
    static __device__ __forceinline__ Acc from(const XYZZ<F>& p) { return f29g2_pt_from_xyzz(p); }
    static __device__ __forceinline__ XYZZ<F> to(const Acc& a) { return f29g2_pt_to_xyzz(a); }
    static __device__ __forceinline__ void add(Acc& r, const Acc& a, const Acc& b) { f29g2_pt_add(r, a, b); }
};
template <class G>
__global__ void __launch_bounds__(64)
zkc_msm_merge29(XYZZ<typename G::F>* __restrict__ partial, const uint32_t* __restrict__ segoff, uint32_t* __restrict__ segcnt,
                const uint32_t* __restrict__ heavy, const uint32_t* __restrict__ heavy_count, uint32_t max_segments) {
    __shared__ typename G::Acc sh[64];
    uint32_t nh = *heavy_count; if (nh > (uint32_t)MSM_MAX_HEAVY) nh = MSM_MAX_HEAVY;
    for (uint32_t h = blockIdx.x; h < nh; h += gridDim.x) {
        const uint32_t b = heavy[h], s0 = segoff[b];
        uint32_t s1 = s0 + segcnt[b]; if (s1 > max_segments) s1 = max_segments;
        typename G::Acc acc; G::set_inf(acc);
        for (uint32_t s = s0 + threadIdx.x; s < s1; s += 64) { const XYZZ<typename G::F> p = partial[s]; if (!p.is_inf()) { const typename G::Acc q = G::from(p); G::add(acc, acc, q); } }
        __syncthreads();
        sh[threadIdx.x] = acc; __syncthreads();
        for (int st = 32; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) { typename G::Acc t = sh[threadIdx.x]; G::add(t, t, sh[threadIdx.x + st]); sh[threadIdx.x] = t; }
            __syncthreads();
        }
        if (threadIdx.x == 0) { partial[s0] = G::is_inf(sh[0]) ? XYZZ<typename G::F>::inf() : G::to(sh[0]); segcnt[b] = 1; }
        __syncthreads();
    }
}

// ---- K6 ---- one wave per virtual window of 64 x per consecutive buckets of a job; lane t owns buckets per t .. per t + per - 1.
// Output per virtual window: W = sum_j j * B_j (local weights) and S = sum_j B_j; zkc_msm_final applies the window's offset.
// [r2] tried and not kept: splitting this kernel into its dense half (per-lane sums to memory, 1.49 ms instead of 2.26 on the G1 stream) and a
// lane-per-window fold of the 64 pairs on the blinding stream (1.7 ms, 103 waves).  Same box: pass period 29.39 against 29.53 ms, 3069 / 3062 against
// 3067 / 3051 proofs/s -- the fold and the blinding then run beside the next pass' first NTT kernel, which slows down by what this stream gained.
// K6 for G1 with radix-2^29 coordinates (zkc_f29_g1.h): same walk, 1.7x fewer instructions per group addition than the generic code
// through the out-of-line product.  Partials are read as canonical 8 x u32 points and sliced; W and S leave in canonical form.
__global__ void __launch_bounds__(64)
zkc_msm_window29(const XYZZ<Fq>* __restrict__ partial, const uint32_t* __restrict__ segoff, const uint32_t* __restrict__ segcnt,
                 const MsmWindow* __restrict__ windows, XYZZ<Fq>* __restrict__ wres, uint32_t max_segments) {
    __shared__ Acc29 sh[64];
    const MsmWindow win = windows[blockIdx.x];
    const int PER = (int)win.per;
    const uint32_t first = win.bucket0 + threadIdx.x * PER;
    Acc29 run, loc; f29_pt_set_inf(run); f29_pt_set_inf(loc);
    for (int k = PER - 1; k >= 0; k--) {
        uint32_t s0 = segoff[first + k], s1 = s0 + segcnt[first + k];
        if (s1 > max_segments) s1 = max_segments;
        for (uint32_t s = s0; s < s1; s++) { const Acc29 q = f29_pt_from_xyzz(partial[s]); f29_pt_add(run, run, q); }
        f29_pt_add(loc, loc, run);
    }
    sh[threadIdx.x] = run; __syncthreads();
    for (int o = 1; o < 64; o <<= 1) {                                     // suffix sums R_t across the lanes
        Acc29 v; f29_pt_set_inf(v);
        if ((int)threadIdx.x + o < 64) v = sh[threadIdx.x + o];
        __syncthreads();
        if ((int)threadIdx.x + o < 64) { Acc29 t = sh[threadIdx.x]; f29_pt_add(t, t, v); sh[threadIdx.x] = t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) wres[2 * win.out + 1] = f29_pt_is_inf(sh[0]) ? XYZZ<Fq>::inf() : f29_pt_to_xyzz(sh[0]);      // S = R_0
    Acc29 y; f29_pt_set_inf(y);
    if (threadIdx.x >= 1) { y = sh[threadIdx.x]; if (!f29_pt_is_inf(y)) for (int k = PER; k > 1; k >>= 1) f29_pt_dbl(y, y); }
    f29_pt_add(y, y, loc);
    __syncthreads();
    sh[threadIdx.x] = y; __syncthreads();
    for (int st = 32; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { Acc29 t = sh[threadIdx.x]; f29_pt_add(t, t, sh[threadIdx.x + st]); sh[threadIdx.x] = t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) wres[2 * win.out] = f29_pt_is_inf(sh[0]) ? XYZZ<Fq>::inf() : f29_pt_to_xyzz(sh[0]);
}
// K6 for G2 in radix 2^29 (zkc_f29_g2.h): the generic version needs 256 VGPRs plus 1.4 KB of scratch per lane and was the longest
// link of a single proof's critical path.
__global__ void __launch_bounds__(64)
zkc_msm_window29_g2(const XYZZ<Fq2>* __restrict__ partial, const uint32_t* __restrict__ segoff, const uint32_t* __restrict__ segcnt,
                    const MsmWindow* __restrict__ windows, XYZZ<Fq2>* __restrict__ wres, uint32_t max_segments) {
    __shared__ Acc29G2 sh[64];
    const MsmWindow win = windows[blockIdx.x];
    const int PER = (int)win.per;
    const uint32_t first = win.bucket0 + threadIdx.x * PER;
    Acc29G2 run, loc; f29g2_pt_set_inf(run); f29g2_pt_set_inf(loc);
    for (int k = PER - 1; k >= 0; k--) {
        uint32_t s0 = segoff[first + k], s1 = s0 + segcnt[first + k];
        if (s1 > max_segments) s1 = max_segments;
        for (uint32_t s = s0; s < s1; s++) { const XYZZ<Fq2> pq = partial[s]; if (!pq.is_inf()) { const Acc29G2 q = f29g2_pt_from_xyzz(pq); f29g2_pt_add(run, run, q); } }
        f29g2_pt_add(loc, loc, run);
    }
    sh[threadIdx.x] = run; __syncthreads();
    for (int o = 1; o < 64; o <<= 1) {
        Acc29G2 v; f29g2_pt_set_inf(v);
        if ((int)threadIdx.x + o < 64) v = sh[threadIdx.x + o];
        __syncthreads();
        if ((int)threadIdx.x + o < 64) { Acc29G2 t = sh[threadIdx.x]; f29g2_pt_add(t, t, v); sh[threadIdx.x] = t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) wres[2 * win.out + 1] = f29g2_pt_is_inf(sh[0]) ? XYZZ<Fq2>::inf() : f29g2_pt_to_xyzz(sh[0]);
    Acc29G2 y; f29g2_pt_set_inf(y);
    if (threadIdx.x >= 1) { y = sh[threadIdx.x]; if (!f29g2_pt_is_inf(y)) for (int k = PER; k > 1; k >>= 1) f29g2_pt_dbl(y, y); }
    f29g2_pt_add(y, y, loc);
    __syncthreads();
    sh[threadIdx.x] = y; __syncthreads();
    for (int st = 32; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { Acc29G2 t = sh[threadIdx.x]; f29g2_pt_add(t, t, sh[threadIdx.x + st]); sh[threadIdx.x] = t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) wres[2 * win.out] = f29g2_pt_is_inf(sh[0]) ? XYZZ<Fq2>::inf() : f29g2_pt_to_xyzz(sh[0]);
}
// one workgroup per job: result = sum_k W_k + vw * sum_k k * S_k over the job's virtual windows k (digit = vw k + local index)
// the per-job sum for G1 in radix 2^29, up to MSM_MAX_VW_G1 virtual windows per job; the scans run over the job's own window count (rounded up to a power of two)
__global__ void __launch_bounds__(MSM_MAX_VW_G1)
zkc_msm_final29(const XYZZ<Fq>* __restrict__ wres, const MsmJobList* __restrict__ jl, XYZZ<Fq>* __restrict__ results) {
    __shared__ Acc29 sh[MSM_MAX_VW_G1];
    const int j = blockIdx.x;
    const uint32_t vw = jl->job[j].vw, nvw = (1u << (jl->job[j].c - 1)) / vw, w0 = jl->job[j].win_off;
    int nt = 1; while ((uint32_t)nt < nvw) nt <<= 1;                       // uniform over the workgroup
    Acc29 Wk, Sk; f29_pt_set_inf(Wk); f29_pt_set_inf(Sk);
    if (threadIdx.x < nvw) {
        const XYZZ<Fq> a = wres[2 * (w0 + threadIdx.x)], b = wres[2 * (w0 + threadIdx.x) + 1];
        if (!a.is_inf()) Wk = f29_pt_from_xyzz(a);
        if (!b.is_inf()) Sk = f29_pt_from_xyzz(b);
    }
    sh[threadIdx.x] = Sk; __syncthreads();
    for (int o = 1; o < nt; o <<= 1) {                                     // suffix sums of S over k
        Acc29 v; f29_pt_set_inf(v);
        if ((int)threadIdx.x + o < nt) v = sh[threadIdx.x + o];
        __syncthreads();
        if ((int)threadIdx.x + o < nt) { Acc29 t = sh[threadIdx.x]; f29_pt_add(t, t, v); sh[threadIdx.x] = t; }
        __syncthreads();
    }
    Acc29 y; f29_pt_set_inf(y);
    if (threadIdx.x >= 1 && (int)threadIdx.x < nt) { y = sh[threadIdx.x]; if (!f29_pt_is_inf(y)) for (uint32_t k = vw; k > 1; k >>= 1) f29_pt_dbl(y, y); }   // sum_{k>=1} R_k = sum_k k S_k, times vw
    f29_pt_add(y, y, Wk);
    __syncthreads();
    sh[threadIdx.x] = y; __syncthreads();
    for (int st = nt / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { Acc29 t = sh[threadIdx.x]; f29_pt_add(t, t, sh[threadIdx.x + st]); sh[threadIdx.x] = t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) results[j] = f29_pt_is_inf(sh[0]) ? XYZZ<Fq>::inf() : f29_pt_to_xyzz(sh[0]);
}

// the per-job sum for G2 in radix 2^29 (same scheme as zkc_msm_final)
__global__ void __launch_bounds__(MSM_MAX_VW_PER_JOB)
zkc_msm_final29_g2(const XYZZ<Fq2>* __restrict__ wres, const MsmJobList* __restrict__ jl, XYZZ<Fq2>* __restrict__ results) {
    __shared__ Acc29G2 sh[MSM_MAX_VW_PER_JOB];
    const int j = blockIdx.x;
    const uint32_t vw = jl->job[j].vw, nvw = (1u << (jl->job[j].c - 1)) / vw, w0 = jl->job[j].win_off;
    int nt = 1; while ((uint32_t)nt < nvw) nt <<= 1;                       // uniform over the workgroup: the scans run over the job's own window count
    Acc29G2 Wk, Sk; f29g2_pt_set_inf(Wk); f29g2_pt_set_inf(Sk);
    if (threadIdx.x < nvw) {
        const XYZZ<Fq2> a = wres[2 * (w0 + threadIdx.x)], b = wres[2 * (w0 + threadIdx.x) + 1];
        if (!a.is_inf()) Wk = f29g2_pt_from_xyzz(a);
        if (!b.is_inf()) Sk = f29g2_pt_from_xyzz(b);
    }
    sh[threadIdx.x] = Sk; __syncthreads();
    for (int o = 1; o < nt; o <<= 1) {
        Acc29G2 v; f29g2_pt_set_inf(v);
        if ((int)threadIdx.x + o < nt) v = sh[threadIdx.x + o];
        __syncthreads();
        if ((int)threadIdx.x + o < nt) { Acc29G2 t = sh[threadIdx.x]; f29g2_pt_add(t, t, v); sh[threadIdx.x] = t; }
        __syncthreads();
    }
    Acc29G2 y; f29g2_pt_set_inf(y);
    if (threadIdx.x >= 1 && (int)threadIdx.x < nt) { y = sh[threadIdx.x]; if (!f29g2_pt_is_inf(y)) for (uint32_t k = vw; k > 1; k >>= 1) f29g2_pt_dbl(y, y); }
    f29g2_pt_add(y, y, Wk);
    __syncthreads();
    sh[threadIdx.x] = y; __syncthreads();
    for (int st = nt / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { Acc29G2 t = sh[threadIdx.x]; f29g2_pt_add(t, t, sh[threadIdx.x + st]); sh[threadIdx.x] = t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) results[j] = f29g2_pt_is_inf(sh[0]) ? XYZZ<Fq2>::inf() : f29g2_pt_to_xyzz(sh[0]);
}

// ---- one-time base table: table[w][i] = 2^c * table[w-1][i] ----
template <class F>
__global__ void __launch_bounds__(128)
zkc_msm_shift_bases(const Affine<F>* __restrict__ prev, Affine<F>* __restrict__ next, uint32_t count, int c) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Affine<F> a = PointIO<F>::load(prev + i);
    XYZZ<F> p = xyzz_dbl_affine(a);
    for (int k = 1; k < c; k++) p = xyzz_dbl(p);
    next[i] = xyzz_to_affine(p);
}
template <class F>
static int precompute(zkc_ctx* ctx, uint32_t count, Affine<F>* d_table, int c) {
    for (int w = 1; w < msm_nw(c); w++) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(zkc_msm_shift_bases<F>), dim3((count + 127) / 128), dim3(128), 0, ctx->stream,
                           d_table + (size_t)(w - 1) * count, d_table + (size_t)w * count, count, c);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return zkc_fail(ctx, ZKC_ERR_HIP, std::string("zkc_msm_shift_bases: ") + hipGetErrorString(e));
    }
    return ZKC_OK;
}
int msm_precompute_g1(zkc_ctx* ctx, uint32_t count, G1Affine* d_table, int c) { return precompute<Fq>(ctx, count, d_table, c); }
int msm_precompute_g2(zkc_ctx* ctx, uint32_t count, G2Affine* d_table, int c) { return precompute<Fq2>(ctx, count, d_table, c); }

// ---- constant folding support: out[i] = scalar[wire_i] * P[wire_i - shift] by double-and-add, then per-group sums ----
__device__ __forceinline__ uint32_t limb_of(const uint32_t k[8], int i) {
    uint32_t w = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) w = (q == i) ? k[q] : w;
    return w;
}
template <class F>
__global__ void __launch_bounds__(64)
zkc_fold_mul(const Affine<F>* __restrict__ tbl, const uint32_t* __restrict__ scalars, const uint32_t* __restrict__ wires, uint32_t nw,
             int32_t pt_shift, XYZZ<F>* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nw) return;
    const uint32_t wire = wires[i];
    const uint4* sp = reinterpret_cast<const uint4*>(scalars + 8 * (size_t)wire); uint4 a = sp[0], b = sp[1];
    const uint32_t k[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    Affine<F> p = PointIO<F>::load(tbl + (uint32_t)((int32_t)wire - pt_shift));
    XYZZ<F> r = XYZZ<F>::inf();
    int top = -1;                                                    // the batch verifier's weights are 128 bits wide: start at the scalar's own top bit
#pragma unroll
    for (int q = 7; q >= 0; q--) if (top < 0 && k[q]) top = 32 * q + 31 - __clz(k[q]);
    for (int bit = top; bit >= 0; bit--) {
        r = xyzz_dbl(r);
        if ((limb_of(k, bit >> 5) >> (bit & 31)) & 1) r = xyzz_add_affine(r, p);
    }
    out[i] = r;
}
template <class F>
__global__ void __launch_bounds__(64)
zkc_fold_gsum(const XYZZ<F>* __restrict__ in, const uint32_t* __restrict__ gstart, uint32_t ngroups, XYZZ<F>* __restrict__ out) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t i = gstart[g], e = gstart[g + 1]; i < e; i++) acc = xyzz_add(acc, in[i]);
    out[g] = acc;
}
template <class F>
static int fold_group_sums(zkc_ctx* ctx, const Affine<F>* tbl, const uint32_t* d_scalars, const uint32_t* d_wires, uint32_t nw, int32_t pt_shift,
                           const uint32_t* d_gstart, uint32_t ngroups, XYZZ<F>* h_out) {
    XYZZ<F>*d_tmp = nullptr, *d_out = nullptr;
    const int rc = [&]() -> int {
        ZKC_HIP_CHECK(ctx, hipMalloc(&d_tmp, (size_t)nw * sizeof(XYZZ<F>)));
        ZKC_HIP_CHECK(ctx, hipMalloc(&d_out, (size_t)ngroups * sizeof(XYZZ<F>)));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(zkc_fold_mul<F>), dim3((nw + 63) / 64), dim3(64), 0, ctx->stream, tbl, d_scalars, d_wires, nw, pt_shift, d_tmp);
        ZKC_HIP_CHECK(ctx, hipGetLastError());
        hipLaunchKernelGGL(HIP_KERNEL_NAME(zkc_fold_gsum<F>), dim3((ngroups + 63) / 64), dim3(64), 0, ctx->stream, d_tmp, d_gstart, ngroups, d_out);
        ZKC_HIP_CHECK(ctx, hipGetLastError());
        ZKC_HIP_CHECK(ctx, hipMemcpyAsync(h_out, d_out, (size_t)ngroups * sizeof(XYZZ<F>), hipMemcpyDeviceToHost, ctx->stream));
        ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        return ZKC_OK;
    }();
    if (d_tmp) (void)hipFree(d_tmp);
    if (d_out) (void)hipFree(d_out);
    return rc;
}
int fold_group_sums_g1(zkc_ctx* ctx, const G1Affine* tbl, const uint32_t* s, const uint32_t* w, uint32_t nw, int32_t sh, const uint32_t* gs, uint32_t ng, G1XYZZ* o) {
    return fold_group_sums<Fq>(ctx, tbl, s, w, nw, sh, gs, ng, o);
}
// the batch verifier's form: scratch and sums in buffers of the caller's (the context's verifier work space), the sums left on the device for the Miller kernels and copied to h_out
int fold_group_sums_g1_ws(zkc_ctx* ctx, const G1Affine* tbl, const uint32_t* s, const uint32_t* w, uint32_t nw, const uint32_t* gs, uint32_t ng, G1XYZZ* d_tmp, G1XYZZ* d_out, G1XYZZ* h_out) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(zkc_fold_mul<Fq>), dim3((nw + 63) / 64), dim3(64), 0, ctx->stream, tbl, s, w, nw, 0, d_tmp);
    ZKC_HIP_CHECK(ctx, hipGetLastError());
    hipLaunchKernelGGL(HIP_KERNEL_NAME(zkc_fold_gsum<Fq>), dim3((ng + 63) / 64), dim3(64), 0, ctx->stream, d_tmp, gs, ng, d_out);
    ZKC_HIP_CHECK(ctx, hipGetLastError());
    ZKC_HIP_CHECK(ctx, hipMemcpyAsync(h_out, d_out, (size_t)ng * sizeof(G1XYZZ), hipMemcpyDeviceToHost, ctx->stream));
    ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ZKC_OK;
}
int fold_group_sums_g2(zkc_ctx* ctx, const G2Affine* tbl, const uint32_t* s, const uint32_t* w, uint32_t nw, int32_t sh, const uint32_t* gs, uint32_t ng, G2XYZZ* o) {
    return fold_group_sums<Fq2>(ctx, tbl, s, w, nw, sh, gs, ng, o);
}

// ---- work space ----
static int msm_work_alloc_impl(zkc_ctx* ctx, MsmWork& w, size_t max_entries, size_t max_buckets, int max_jobs, bool g2) {
    w.max_entries = max_entries; w.max_jobs = max_jobs; w.max_buckets = max_buckets; w.xyzz_size = g2 ? sizeof(G2XYZZ) : sizeof(G1XYZZ);
    const size_t nb = max_buckets;
    // segments of a pass: a bucket of c entries is cut into ceil(c / seg) segments, seg = clamp(total / 131072, MSM_SEG_MIN, MSM_SEG) (msm_pass), so there are at most
    // total / seg + nb of them: total / MSM_SEG for a full pass, under 131072 x 17 / 16 whenever seg is below MSM_SEG.  [r5] Rounds 1-4 sized every per-segment array for
    // max_entries / MSM_SEG_MIN -- a full pass cut into 16-entry segments, which msm_pass never does: 4.7 GB of partial sums per lane at 64 proofs in flight where 1.4 are reachable.
    // (32 x 8192: the half-wave-per-bucket form of a small G2 pass wants up to 32 slices for each of its <= 8192 buckets.)
    w.max_segments = std::max<size_t>(max_entries / MSM_SEG + 1, (size_t)32 * 8192) + nb + 64;
    w.max_bins = (size_t)max_jobs << MSM_MAX_HBITS;
    ZKC_HIP_CHECK(ctx, hipMalloc((void**)&w.vals, max_entries * 4 + 16)); ZKC_HIP_CHECK(ctx, hipMalloc((void**)&w.vals2, max_entries * 4 + 16));
    ZKC_HIP_CHECK(ctx, hipMalloc((void**)&w.hist, w.max_bins * 4)); ZKC_HIP_CHECK(ctx, hipMalloc((void**)&w.bin_start, w.max_bins * 4));
    w.max_tilecnt = max_entries / 8 + ((size_t)max_jobs << MSM_MAX_HBITS);       // sum over jobs of 2^hbits x ceil(count / 1024): count <= entries / nw, nw >= 8
    ZKC_HIP_CHECK(ctx, hipMalloc((void**)&w.tilecnt, w.max_tilecnt * 4));
    ZKC_HIP_CHECK(ctx, hipMalloc((void**)&w.off, (nb + 2) * 4)); ZKC_HIP_CHECK(ctx, hipMalloc((void**)&w.bcnt, (nb + 2) * 4)); ZKC_HIP_CHECK(ctx, hipMalloc((void**)&w.segcnt, (nb + 2) * 4));
    ZKC_HIP_CHECK(ctx, hipMalloc((void**)&w.segoff, (nb + 2) * 4)); ZKC_HIP_CHECK(ctx, hipMalloc((void**)&w.seg2bucket, w.max_segments * 4));
    ZKC_HIP_CHECK(ctx, hipMalloc((void**)&w.seglen, w.max_segments * 4)); ZKC_HIP_CHECK(ctx, hipMalloc((void**)&w.perm, w.max_segments * 4));
    w.max_lencnt = (size_t)(MSM_SEG + 1) * (w.max_segments / 256 + 2);
    ZKC_HIP_CHECK(ctx, hipMalloc((void**)&w.lencnt, w.max_lencnt * 4));
    ZKC_HIP_CHECK(ctx, hipMalloc((void**)&w.scan_blk, (std::max(nb + 2, w.max_lencnt) / 1024 + 2) * 4));
    ZKC_HIP_CHECK(ctx, hipMalloc((void**)&w.heavy, (MSM_MAX_HEAVY + 1) * 4));
    ZKC_HIP_CHECK(ctx, hipMalloc((void**)&w.d_jobs, sizeof(MsmJobList)));
    ZKC_HIP_CHECK(ctx, hipMalloc(&w.partial, w.max_segments * w.xyzz_size));
    const size_t max_vw = max_buckets / MSM_VW_MIN + (size_t)max_jobs;
    ZKC_HIP_CHECK(ctx, hipMalloc(&w.wres, 2 * max_vw * w.xyzz_size));
    ZKC_HIP_CHECK(ctx, hipMalloc((void**)&w.d_windows, max_vw * sizeof(MsmWindow))); w.max_windows = max_vw;
    w.max_tiles = max_entries / (8 * MSM_TILE_SCALARS) + (size_t)max_jobs + 1;
    ZKC_HIP_CHECK(ctx, hipMalloc((void**)&w.d_tilejob, w.max_tiles * 2));
    for (int s = 0; s < 2; s++) {
        ZKC_HIP_CHECK(ctx, hipHostMalloc((void**)&w.h_tilejob[s], w.max_tiles * 2));
        ZKC_HIP_CHECK(ctx, hipHostMalloc((void**)&w.h_jobs[s], sizeof(MsmJobList))); ZKC_HIP_CHECK(ctx, hipHostMalloc((void**)&w.h_windows[s], max_vw * sizeof(MsmWindow)));
        ZKC_HIP_CHECK(ctx, hipEventCreateWithFlags(&w.h_ev[s], hipEventDisableTiming));
    }
    ZKC_HIP_CHECK(ctx, hipMalloc(&w.results, 2 * (size_t)max_jobs * w.xyzz_size));
    ZKC_HIP_CHECK(ctx, hipHostMalloc(&w.h_results, (size_t)max_jobs * w.xyzz_size));
    return ZKC_OK;
}
// all or nothing: a failed allocation leaves `w` empty (capacities zero), so that msm_pass' capacity checks refuse it instead of launching on null buffers
int msm_work_alloc(zkc_ctx* ctx, MsmWork& w, size_t max_entries, size_t max_buckets, int max_jobs, bool g2) {
    const int rc = msm_work_alloc_impl(ctx, w, max_entries, max_buckets, max_jobs, g2);
    if (rc != ZKC_OK) msm_work_free(w);
    return rc;
}
void msm_work_free(MsmWork& w) {
    void* p[] = {w.vals, w.vals2, w.hist, w.bin_start, w.tilecnt, w.off, w.bcnt, w.segcnt, w.segoff, w.seg2bucket, w.seglen, w.perm, w.scan_blk, w.lencnt, w.heavy, w.d_jobs, w.d_windows,
                 w.partial, w.wres, w.results};
    for (void* q : p) if (q) (void)hipFree(q);
    if (w.h_results) (void)hipHostFree(w.h_results);
    if (w.d_tilejob) (void)hipFree(w.d_tilejob);
    for (int s = 0; s < 2; s++) { if (w.h_tilejob[s]) (void)hipHostFree(w.h_tilejob[s]); if (w.h_jobs[s]) (void)hipHostFree(w.h_jobs[s]); if (w.h_windows[s]) (void)hipHostFree(w.h_windows[s]); if (w.h_ev[s]) (void)hipEventDestroy(w.h_ev[s]); }
    w = MsmWork();
}

static const bool g_debug_sync = getenv("ZKC_DEBUG_SYNC") != nullptr;   // serialise + log every launch (diagnostics only)
#define ZKC_LAUNCH_CHECK(ctx, name)                                                                        \
    do { hipError_t _e = hipGetLastError(); if (_e != hipSuccess)                                          \
        return zkc_fail((ctx), ZKC_ERR_HIP, std::string(name ": ") + hipGetErrorString(_e));               \
        if (g_debug_sync) { timespec _t0, _t1; clock_gettime(CLOCK_MONOTONIC, &_t0);                        \
            _e = hipStreamSynchronize(st); clock_gettime(CLOCK_MONOTONIC, &_t1);                \
            fprintf(stderr, "[zkc] %-22s %8.3f ms  %s\n", name, (_t1.tv_sec - _t0.tv_sec) * 1e3 + (_t1.tv_nsec - _t0.tv_nsec) * 1e-6, \
                    hipGetErrorString(_e)); fflush(stderr); } } while (0)

template <class F>
static int msm_pass(zkc_zkey* zk, MsmWork& w, const Affine<F>* table, const MsmJobList& jl_in, int slot, bool to_host, hipStream_t st, hipEvent_t ev_sorted, hipEvent_t wait_before_acc = nullptr, hipEvent_t ev_acc = nullptr,
                    hipStream_t st_red = nullptr, hipEvent_t ev_red = nullptr) {
    zkc_ctx* ctx = zk->ctx;
    const int nj = jl_in.njobs;
    if (nj <= 0 || nj > w.max_jobs) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "msm_pass: job count");
    static thread_local MsmJobList jl;                  // the caller's list plus the bucket-id layout of this pass
    jl = jl_in;
    if (!jl.finish()) {
        std::string d = "msm_pass: jobs need at most two window sizes (the larger first) and tables that fit the entry word [" + std::to_string(nj) + " jobs:";
        for (int j = 0; j < nj && j < 16; j++) d += " c" + std::to_string(jl.job[j].c) + "/" + std::to_string(jl.job[j].count) + "/" + std::to_string(jl.job[j].tbl_count);
        return zkc_fail(ctx, ZKC_ERR_BAD_ARG, d + "]");
    }
    const size_t total = jl.total_entries;
    const uint32_t nb = jl.total_buckets;
    if (total > w.max_entries || nb > w.max_buckets) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "msm_pass: too many entries for the work space");
    // entries per accumulation lane: long segments mean fewer partial sums to reduce (best for a full pass), but a lane walks its segment
    // serially, so a small pass (one proof) wants short ones: aim at ~2 waves per SIMD
    const uint32_t seg = (uint32_t)std::min<size_t>(MSM_SEG, std::max<size_t>(MSM_SEG_MIN, total / 131072));
    constexpr bool kG2 = sizeof(F) == sizeof(Fq2);
    static const bool bw_off = [] { const char* e = getenv("ZKC_G2_BUCKET_WAVE"); return e && atoi(e) == 0; }();
    const uint32_t bw_slices = (uint32_t)std::min<size_t>(32, w.max_segments / std::max<uint32_t>(nb, 1u));      // slices a heavy bucket may be cut into: nb x slices partial sums must fit
    const uint32_t* const g2_table29 = (kG2 && jl.job[0].c == (uint32_t)MSM_C_G2_LONE) ? zk->d_g2_29_lone : (kG2 && zk->c_deep && jl.job[0].c == (uint32_t)zk->c_deep) ? zk->d_g2_29_deep : zk->d_g2_29;       // a G2 pass is of one window size
    const bool bucket_wave = kG2 && !bw_off && nb <= 8192 && bw_slices >= 1;         // a small G2 pass: half a wave per bucket (zkc_msm_bucketwave_g2), no segment lists
    uint64_t alg_bytes = 0; uint32_t maxcount = 0;
    uint64_t streamed_bytes = 0;
    for (int j = 0; j < nj; j++) {
        alg_bytes += (uint64_t)jl.job[j].tbl_count * (sizeof(Affine<F>) + 32);     // the whole section, folded or not
        streamed_bytes += (uint64_t)jl.job[j].count * (sizeof(Affine<F>) + 32);     // (scalar, base) pairs that actually enter the MSM
        maxcount = std::max(maxcount, jl.job[j].count);
    }
    const size_t seg_bound = std::min<size_t>(w.max_segments, total / seg + nb);     // launch bound on the number of segments
    {
        zkc_prof_scope _ps(ctx, ZKC_PROF_MSM_SORT, 0, st);
        if (jl.total_windows > w.max_windows) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "msm_pass: too many virtual windows for the work space");
        const int hs = w.h_next; w.h_next ^= 1;
        ZKC_HIP_CHECK(ctx, zkc_wait_event(w.h_ev[hs]));                            // the copy that last used this staging slot has executed (two passes back)
        MsmWindow* wins = w.h_windows[hs]; uint32_t nwin = 0;
        for (int j = 0; j < nj; j++) for (uint32_t k = 0; k < (uint32_t)msm_half((int)jl.job[j].c) / jl.job[j].vw; k++)
            wins[nwin++] = MsmWindow{jl.id_of(k * jl.job[j].vw, (uint32_t)j), jl.job[j].win_off + k, jl.job[j].vw / 64};
        memcpy(w.h_jobs[hs], &jl, offsetof(MsmJobList, job) + (size_t)nj * sizeof(MsmJob)); w.h_jobs[hs]->njobs = jl.njobs;
        memcpy(&w.h_jobs[hs]->njobs, &jl.njobs, sizeof(MsmJobList) - offsetof(MsmJobList, njobs));
        ZKC_HIP_CHECK(ctx, hipMemcpyAsync(w.d_jobs, w.h_jobs[hs], sizeof(MsmJobList), hipMemcpyHostToDevice, st));
        ZKC_HIP_CHECK(ctx, hipMemcpyAsync(w.d_windows, wins, (size_t)nwin * sizeof(MsmWindow), hipMemcpyHostToDevice, st));
        if (jl.total_tiles > w.max_tiles) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "msm_pass: too many tiles for the work space");
        { uint16_t* tj = w.h_tilejob[hs]; uint32_t k = 0; for (int j = 0; j < nj; j++) { const uint32_t nt = (jl.job[j].count + MSM_TILE_SCALARS - 1) / MSM_TILE_SCALARS; for (uint32_t i = 0; i < nt; i++) tj[k++] = (uint16_t)j; } }
        ZKC_HIP_CHECK(ctx, hipMemcpyAsync(w.d_tilejob, w.h_tilejob[hs], (size_t)jl.total_tiles * 2, hipMemcpyHostToDevice, st));
        ZKC_HIP_CHECK(ctx, hipEventRecord(w.h_ev[hs], st));
        unsigned long long* ectr = (!kG2 && ((ctx->prof.mask >> ZKC_PROF_MSM_ACC_G1) & 1)) ? ctx->d_prof_entries : nullptr;      // profiling: real G1 additions of the pass
        int rc = msm_bucket_entries(ctx, w, jl, st, ectr); if (rc) return rc;                  // K4: digits -> entries grouped by bucket (vals2, off, bcnt)
        if (g_debug_sync) { hipError_t _e = hipStreamSynchronize(st); fprintf(stderr, "[zkc] bucket entries: %s\n", hipGetErrorString(_e)); }
        if (!bucket_wave) { rc = msm_build_segments(ctx, w, jl, seg, seg_bound, st); if (rc) return rc; }      // segments of <= seg entries, longest first
        if (g_debug_sync) { hipError_t _e = hipStreamSynchronize(st); fprintf(stderr, "[zkc] segments: %s\n", hipGetErrorString(_e)); }
        if (ev_sorted) ZKC_HIP_CHECK(ctx, hipEventRecord(ev_sorted, st));          // the short kernels of this pass are through: what follows is long-running
    }
    XYZZ<F>* partial = reinterpret_cast<XYZZ<F>*>(w.partial);
    XYZZ<F>* wres = reinterpret_cast<XYZZ<F>*>(w.wres);
    XYZZ<F>* results = reinterpret_cast<XYZZ<F>*>(w.results) + (size_t)slot * w.max_jobs;
    if (wait_before_acc) ZKC_HIP_CHECK(ctx, hipStreamWaitEvent(st, wait_before_acc, 0));      // hold the (VALU-bound) accumulation until the other stream reaches its memory-bound phase
    static const bool acc_chain_on = [] { const char* e = getenv("ZKC_ACC_CHAIN"); return !(e && atoi(e) == 0); }();
    const bool acc_chain = !kG2 && acc_chain_on && total >= ((size_t)1 << 22);       // (a lone proof's 0.2 ms accumulation is not worth an event)
    if (acc_chain) {
        if (!ctx->ev_acc_chain) ZKC_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_acc_chain, hipEventDisableTiming));
        if (ctx->acc_chain_armed) ZKC_HIP_CHECK(ctx, hipStreamWaitEvent(st, ctx->ev_acc_chain, 0));
    }
    {
        zkc_prof_scope _ps(ctx, kG2 ? ZKC_PROF_MSM_ACC_G2 : ZKC_PROF_MSM_ACC_G1, alg_bytes, st);
        if (!kG2 && ((ctx->prof.mask >> ZKC_PROF_MSM_ACC_G1) & 1)) ctx->prof.bytes[ZKC_PROF_MSM_G1_STREAMED] += streamed_bytes;
        if constexpr (kG2) {
          if (bucket_wave) {
            ZKC_HIP_CHECK(ctx, hipMemsetAsync(w.heavy + MSM_MAX_HEAVY, 0, 4, st));
            hipLaunchKernelGGL(zkc_msm_bucketwave_g2, dim3((nb + 1) / 2, bw_slices), dim3(64), 0, st, g2_table29, (const MsmJobList*)w.d_jobs, w.vals2, w.off, w.bcnt, w.segoff, w.segcnt, w.heavy,
                               w.heavy + MSM_MAX_HEAVY, reinterpret_cast<XYZZ<Fq2>*>(partial), nb, bw_slices);
          }
          else {
            static const int g2_form = [] { const char* e = getenv("ZKC_G2_ACC"); return e ? atoi(e) : 0; }();      // 0: registers hold the next row (one wave per SIMD); 1 / 2: LDS-DMA prefetch at one / two waves per SIMD
            if (g2_form == 2)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(zkc_msm_accumulate29_g2_dma<2>), dim3((unsigned)((seg_bound + G2DMA_T - 1) / G2DMA_T)), dim3(G2DMA_T), 0, st,
                                   g2_table29, (const MsmJobList*)w.d_jobs, w.vals2, w.off, w.bcnt, w.segoff, w.seg2bucket, w.perm, nb, reinterpret_cast<XYZZ<Fq2>*>(partial), (uint32_t)w.max_segments);
            else if (g2_form == 1)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(zkc_msm_accumulate29_g2_dma<1>), dim3((unsigned)((seg_bound + G2DMA_T - 1) / G2DMA_T)), dim3(G2DMA_T), 0, st,
                                   g2_table29, (const MsmJobList*)w.d_jobs, w.vals2, w.off, w.bcnt, w.segoff, w.seg2bucket, w.perm, nb, reinterpret_cast<XYZZ<Fq2>*>(partial), (uint32_t)w.max_segments);
            else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(zkc_msm_accumulate29_g2<1>), dim3((unsigned)((seg_bound + 127) / 128)), dim3(128), 0, st,
                               g2_table29, (const MsmJobList*)w.d_jobs, w.vals2, w.off, w.bcnt, w.segoff, w.seg2bucket, w.perm, nb, reinterpret_cast<XYZZ<Fq2>*>(partial), (uint32_t)w.max_segments);
          }
        } else {  // G1: same layout, field type with the inlined product.  160 VGPRs = 3 waves per SIMD; capped at 128 (4 waves) the accumulator spills and the kernel is 3.6x slower
            hipLaunchKernelGGL(HIP_KERNEL_NAME(zkc_msm_accumulate29<2>), dim3((unsigned)((seg_bound + 127) / 128)), dim3(128), 0, st,
                               reinterpret_cast<const Affine<Fq>*>(table), (const MsmJobList*)w.d_jobs, w.vals2, w.off, w.bcnt, w.segoff, w.seg2bucket, w.perm, nb,
                               reinterpret_cast<XYZZ<Fq>*>(partial), (uint32_t)w.max_segments);
            if (acc_chain) { ZKC_HIP_CHECK(ctx, hipEventRecord(ctx->ev_acc_chain, st)); ctx->acc_chain_armed = true; }      // the next G1 accumulation of this context, on whatever lane, starts behind this one
        }
        ZKC_LAUNCH_CHECK(ctx, "zkc_msm_accumulate");
        if (ev_acc) ZKC_HIP_CHECK(ctx, hipEventRecord(ev_acc, st));                 // the long kernel of the pass is through: what follows (bucket reduction, blinding) is the latency-bound tail
    }
    // [r4] the bucket reduction of a full G1 pass on a stream of its own (st_red): a few thousand latency-shaped waves that leave most of the chip's issue slots idle -- the G1 stream
    // goes on to the next pass' buildABC / transforms meanwhile, and takes this pass' work space back only when ev_red says so (the caller waits for it before its next msm_pass)
    if (st_red && st_red != st) {
        if (!ev_acc) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "msm_pass: a reduction stream needs ev_acc");
        ZKC_HIP_CHECK(ctx, hipStreamWaitEvent(st_red, ev_acc, 0));
        st = st_red;
    }
    {
        zkc_prof_scope _pr(ctx, ZKC_PROF_MSM_REDUCE, 0, st);
        if constexpr (kG2) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(zkc_msm_merge29<Merge29G2>), dim3(1024), dim3(64), 0, st, reinterpret_cast<XYZZ<Fq2>*>(partial), w.segoff, w.segcnt, w.heavy,
                               w.heavy + MSM_MAX_HEAVY, (uint32_t)w.max_segments);
        } else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(zkc_msm_merge29<Merge29G1>), dim3(1024), dim3(64), 0, st, reinterpret_cast<XYZZ<Fq>*>(partial), w.segoff, w.segcnt, w.heavy,
                               w.heavy + MSM_MAX_HEAVY, (uint32_t)w.max_segments);
        ZKC_LAUNCH_CHECK(ctx, "zkc_msm_merge");
        if constexpr (kG2)
            hipLaunchKernelGGL(zkc_msm_window29_g2, dim3(jl.total_windows), dim3(64), 0, st, reinterpret_cast<const XYZZ<Fq2>*>(partial), w.segoff, w.segcnt,
                               (const MsmWindow*)w.d_windows, reinterpret_cast<XYZZ<Fq2>*>(wres), (uint32_t)w.max_segments);
        else
            hipLaunchKernelGGL(zkc_msm_window29, dim3(jl.total_windows), dim3(64), 0, st, reinterpret_cast<const XYZZ<Fq>*>(partial), w.segoff, w.segcnt,
                               (const MsmWindow*)w.d_windows, reinterpret_cast<XYZZ<Fq>*>(wres), (uint32_t)w.max_segments);
        ZKC_LAUNCH_CHECK(ctx, "zkc_msm_window");
        static_assert(msm_half(MSM_C_SMALL) / MSM_VW_MIN <= MSM_MAX_VW_PER_JOB, "G2 per-job sum: one lane per virtual window");
        if (kG2 && jl.job[0].c == (uint32_t)MSM_C_G2_LONE && !zk->d_g2_29_lone) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "msm_pass: no 8-bit-window G2 table on this key");
        for (int j = 0; j < nj; j++) if ((1u << (jl.job[j].c - 1)) / jl.job[j].vw > (uint32_t)(kG2 ? MSM_MAX_VW_PER_JOB : MSM_MAX_VW_G1)) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "msm_pass: too many virtual windows in a job for the per-job sum");
        if constexpr (kG2)
            hipLaunchKernelGGL(zkc_msm_final29_g2, dim3(nj), dim3(MSM_MAX_VW_PER_JOB), 0, st, reinterpret_cast<const XYZZ<Fq2>*>(wres), (const MsmJobList*)w.d_jobs,
                               reinterpret_cast<XYZZ<Fq2>*>(results));
        else
            hipLaunchKernelGGL(zkc_msm_final29, dim3(nj), dim3(MSM_MAX_VW_G1), 0, st, reinterpret_cast<const XYZZ<Fq>*>(wres), (const MsmJobList*)w.d_jobs, reinterpret_cast<XYZZ<Fq>*>(results));
        ZKC_LAUNCH_CHECK(ctx, "zkc_msm_final");
    }
    if (to_host) ZKC_HIP_CHECK(ctx, hipMemcpyAsync(w.h_results, results, (size_t)nj * sizeof(XYZZ<F>), hipMemcpyDeviceToHost, st));
    if (ev_red) ZKC_HIP_CHECK(ctx, hipEventRecord(ev_red, st));
    return ZKC_OK;
}
int msm_pass_g1(zkc_zkey* zk, MsmWork& w, const MsmJobList& jl, int slot, bool to_host, hipStream_t st, hipEvent_t ev_sorted, hipEvent_t ev_acc, hipStream_t st_red, hipEvent_t ev_red) {
    return msm_pass<Fq>(zk, w, zk->d_g1, jl, slot, to_host, st, ev_sorted, nullptr, ev_acc, st_red, ev_red);
}
int msm_pass_g2(zkc_zkey* zk, MsmWork& w, const MsmJobList& jl, int slot, bool to_host, hipStream_t st, hipEvent_t wait_before_acc) { return msm_pass<Fq2>(zk, w, zk->d_g2, jl, slot, to_host, st, nullptr, wait_before_acc); }

}  // namespace zkc
