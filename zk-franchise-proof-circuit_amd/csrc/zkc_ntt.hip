// zkc_ntt.hip -- K3/K3a/K7: sparse mat-vec (buildABC), radix-2 NTT over BN254 Fr and the joinABC pointwise pass.
//
// Replaces snarkjs groth16_prove.js buildABC1 / Fr.ifft / batchApplyKey / Fr.fft / joinABC (reached from
// ts_inputs/src/example.ts:358-362) and rapidsnark's equivalents (zk_census_test.go:89).
//
// NTT plan: decimation-in-time over bit-reversed input, log2(n) stages processed in passes of at most 9 stages; a
// pass keeps its tile (2^b "mid" indices x LO_T neighbouring "lo" indices) resident in LDS, so a 2^17 transform
// makes two HBM round trips (9 + 8 stages) instead of seventeen.  The bit reversal is folded into the first pass'
// loads and the 1/n * g^i coset scaling of the inverse transform into the last pass' stores.
// All vectors are Montgomery-form Fr, 32 B per element, natural order in HBM.
#include "zkc_internal.h"
#include "zkc_prover.h"
#include "zkc_f29.h"

namespace zkc {

__device__ __forceinline__ Fr ld_fr(const Fr* p) {
    const uint4* d = reinterpret_cast<const uint4*>(p); uint4 a = d[0], b = d[1];
    Fr r; r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w; return r;
}
__device__ __forceinline__ void st_fr(Fr* p, const Fr& r) {
    uint4* d = reinterpret_cast<uint4*>(p);
    d[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]); d[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}

// ---- buildABC1: rows [0,n) = A, [n,2n) = B ; out = sum coef * w[signal].  coef is stored as val*R^2 (the .zkey
// convention) so one Montgomery product with the standard-form witness lands in Montgomery form. ----
// The matrix is kept in jagged-diagonal order (zkc_zkey_load): rows sorted by length, the k-th coefficients of all rows that have one
// stored contiguously.  Lane r walks row perm[r]: its loads of (col, val) are coalesced across the wave, neighbouring lanes have rows
// of (nearly) equal length (the zkCensus R1CS has rows of 1 to ~120 coefficients), and no per-coefficient product is ever written out.
// Batched over the proofs of a pass: blockIdx.y = proof.
// The first `nlong` rows (more than MATVEC_LONG coefficients; up to 319 in the zkCensus R1CS) get a wave each: a lane walking such a row
// alone is a 319-deep chain of dependent gathers and held the whole kernel for 2 ms.
// [r2] tried and not kept: accumulating a row in the 64-bit column sums of the radix-2^29 product with one reduction per four terms.  It costs
// 218 instructions per term + 323 for the way back to a canonical element, against ~450 per term here; the zkCensus rows hold 1.77 coefficients on
// average (most hold one), so nothing is gained: 1.877 M VALU-busy cycles per pass either way (rocprofv3 SQ counters).
// [r4] Coefficients that are +1 or -1 (276 k of the census circuit's 463 k; 84 % of its non-empty rows hold nothing else) are marked in the two top bits of `col` at key load
// and cost an addition of the wire's MONTGOMERY form, which zkc_wtns_mont makes once per wire (82 754 products per proof) -- the product by the stored val R^2 was doing
// nothing for them but that conversion, once per TERM.  wm == nullptr: no such buffer (more wires than 3 n), every term is a product as before.
constexpr uint32_t MV_UNIT = 0x80000000u, MV_NEG = 0x40000000u, MV_COL = 0x3fffffffu;
extern "C" __global__ void __launch_bounds__(256)
zkc_wtns_mont(const Fr* __restrict__ wtns_std, size_t wtns_stride, Fr* __restrict__ wm, size_t wm_stride, uint32_t nv) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    Fr r2;
#pragma unroll
    for (int k = 0; k < 8; k++) r2.v[k] = FrParams::r2[k];
    st_fr(wm + (size_t)blockIdx.y * wm_stride + i, ld_fr(wtns_std + (size_t)blockIdx.y * wtns_stride + i) * r2);
}
__device__ __forceinline__ Fr mv_term(const Fr* __restrict__ val, const Fr* __restrict__ w, const Fr* __restrict__ wm, uint32_t idx, uint32_t c) {
    if (wm && (c & MV_UNIT)) { const Fr x = ld_fr(wm + (c & MV_COL)); return (c & MV_NEG) ? fp_neg(x) : x; }
    return ld_fr(val + idx) * ld_fr(w + (c & MV_COL));
}
extern "C" __global__ void __launch_bounds__(256)
zkc_matvec_jds(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ rowlen, const uint32_t* __restrict__ jdptr,
               const uint32_t* __restrict__ col, const Fr* __restrict__ val, const Fr* __restrict__ wtns_std, size_t wtns_stride,
               Fr* __restrict__ abc, int n, uint32_t nlong, const Fr* __restrict__ wm_all, size_t wm_stride) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;      // rows [0,n) = A, [n,2n) = B ; abc layout [proof][3][n]
    const Fr* __restrict__ w = wtns_std + (size_t)blockIdx.y * wtns_stride;
    const Fr* __restrict__ wm = wm_all ? wm_all + (size_t)blockIdx.y * wm_stride : nullptr;
    Fr acc = Fr::zero();
    if ((t >> 6) < nlong) {                         // wave-uniform
        const uint32_t r = t >> 6, len = rowlen[r];
        for (uint32_t k = lane; k < len; k += 64) { const uint32_t idx = jdptr[k] + r; acc = acc + mv_term(val, w, wm, idx, col[idx]); }
        for (int d = 32; d > 0; d >>= 1) {
            Fr o;
#pragma unroll
            for (int i = 0; i < 8; i++) o.v[i] = (uint32_t)__shfl_down((int)acc.v[i], d, 64);
            acc = acc + o;
        }
        if (lane == 0) st_fr(abc + (size_t)blockIdx.y * 3 * n + perm[r], acc);
        return;
    }
    const uint32_t r = t - nlong * 63u;             // = nlong + (t - 64 nlong)
    if (r >= 2u * (uint32_t)n) return;
    const uint32_t len = rowlen[r];
    for (uint32_t k = 0; k < len; k++) { const uint32_t idx = jdptr[k] + r; acc = acc + mv_term(val, w, wm, idx, col[idx]); }
    st_fr(abc + (size_t)blockIdx.y * 3 * n + perm[r], acc);
}
// c = a * b
extern "C" __global__ void __launch_bounds__(256)
zkc_pointwise_mul(Fr* __restrict__ abc, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fr* a = abc + (size_t)blockIdx.y * 3 * n;
    if (i < n) st_fr(a + 2 * (size_t)n + i, ld_fr(a + i) * ld_fr(a + n + i));
}
// joinABC: p = a*b - c, written in STANDARD form (the H-MSM reads scalar digits from it).  In radix 2^29: the three operands enter as
// 32 x value (their R' form), (a b + (D - c) 2^261) / 2^261 is one fused product (zkc_f29.h), and a second reduction of the nine limbs alone
// divides by R' once more, which is the way out of Montgomery form; 243 mads per element instead of two out-of-line products.
struct JoinDom { static constexpr L9 D27 = f29_dominator<FrParams>(1u << 29, 1u << 27); };
extern "C" __global__ void __launch_bounds__(256)
zkc_join_abc(const Fr* __restrict__ abc, uint32_t* __restrict__ p_std, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr* a = abc + (size_t)blockIdx.y * 3 * n;
    const Fr av = ld_fr(a + i), bv = ld_fr(a + n + i), cv = ld_fr(a + 2 * (size_t)n + i);
    uint32_t A[9], B[9], C[9], T[9];
    f29_from_fp_shl5(A, av.v); f29_from_fp_shl5(B, bv.v); f29_from_fp_shl5(C, cv.v);
#pragma unroll
    for (int k = 0; k < 9; k++) C[k] = JoinDom::D27.l[k] - C[k];                 // c < 32 p: dominated by D27 (< 43.3 p)
    f29_mul_addhi<FrParams>(T, A, B, C);                                         // (x y - z) 2^261, below 32 * 32 / 169 + 43.3 + 1 < 51 p
    uint64_t col[18];
#pragma unroll
    for (int k = 0; k < 9; k++) { col[k] = T[k]; col[9 + k] = 0; }
    f29_reduce_cols<FrParams>(T, col);                                           // / 2^261: the standard-form value, below 51 p / 2^261 + p < 2 p
    Fr r;
    r.v[0] = f29_word<0>(T); r.v[1] = f29_word<32>(T); r.v[2] = f29_word<64>(T); r.v[3] = f29_word<96>(T);
    r.v[4] = f29_word<128>(T); r.v[5] = f29_word<160>(T); r.v[6] = f29_word<192>(T); r.v[7] = f29_word<224>(T);
    fp_reduce_once<FrParams>(r.v);
    uint4* d = reinterpret_cast<uint4*>(p_std + 8 * ((size_t)blockIdx.y * n + i));
    d[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]); d[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}

// ---- one NTT pass: stages s0+1 .. s0+b of a DIT transform of size 2^logn ----
// index i = (hi | mid | lo), mid = b bits at position s0, lo = s0 bits.  A block owns tile (hi, lo in [lo0, lo0+LO_T)).
// first pass (s0 == 0): loads src[bitrev(i)]; otherwise loads src[i] (src may == dst).
// scale != nullptr: multiply element i by scale[i] when storing (used on the last pass of the inverse transform).
//
// Inside the tile every element is nine 29-bit limbs in R' = 2^261 form (zkc_f29.h): a loaded element enters as 32 x value (< 32 p); a
// butterfly is t = v w (81 + 81 mads, < 1.7 p because the twiddle is < 1.2 p), u + t and u - t + D24, each carried -- no conditional
// subtraction anywhere; magnitudes grow by at most 6.3 p per stage (43 p in the product-free first stage; < 130 p after nine, capacity 169 p) and are brought back below 3 p
// (or multiplied by the scale factor) before the exact division by 32 that returns the 8 x u32 form.  tw29[j] = w^j (or w^-j), j < n/2,
// in that limb form, 12 words per entry (zkc_tw29).
constexpr int NTT_TILE = 1024;        // elements per block tile = 36 KiB of LDS
constexpr int TW29_WORDS = 12;
extern "C" __global__ void __launch_bounds__(256) zkc_tw29(const Fr* __restrict__ tw, uint32_t* __restrict__ out, uint32_t count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const Fr x = ld_fr(tw + i);
    uint32_t t[9], r[9]; f29_from_fp_shl5(t, x.v); f29_mul<FrParams>(r, t, F29K<FrParams>::one.l);       // < 1.2 p
    uint4* o = reinterpret_cast<uint4*>(out + (size_t)TW29_WORDS * i);
    o[0] = make_uint4(r[0], r[1], r[2], r[3]); o[1] = make_uint4(r[4], r[5], r[6], r[7]); o[2] = make_uint4(r[8], 0, 0, 0);
}
struct NttDom { static constexpr L9 D24 = f29_dominator<FrParams>(1u << 29, 1u << 24); static constexpr L9 D27 = f29_dominator<FrParams>(1u << 29, 1u << 27); };
__device__ __forceinline__ void ntt_ld_w(uint32_t w[9], const uint32_t* __restrict__ tw29, size_t e) {
    const uint4* wp = reinterpret_cast<const uint4*>(tw29 + (size_t)TW29_WORDS * e);
    const uint4 w0 = wp[0], w1 = wp[1], w2 = wp[2];
    w[0] = w0.x; w[1] = w0.y; w[2] = w0.z; w[3] = w0.w; w[4] = w1.x; w[5] = w1.y; w[6] = w1.z; w[7] = w1.w; w[8] = w2.x;
}
// one multiply-then-add butterfly on two LDS slots; `plain`: twiddle is 1 and v may be as large as 32 p (freshly loaded)
__device__ __forceinline__ void ntt_bfly(uint32_t* pu, uint32_t* pv, const uint32_t w[9], bool plain, bool carry) {
    uint32_t u[9], v[9], tt[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { u[i] = pu[i]; v[i] = pv[i]; }
    if (plain) {
#pragma unroll
        for (int i = 0; i < 9; i++) { const uint32_t vi = v[i]; v[i] = u[i] + NttDom::D27.l[i] - vi; u[i] += vi; }
    } else {
        f29_mul<FrParams>(tt, v, w);
#pragma unroll
        for (int i = 0; i < 9; i++) { v[i] = u[i] + NttDom::D24.l[i] - tt[i]; u[i] += tt[i]; }
    }
    if (carry) { f29_carry(u); f29_carry(v); }
#pragma unroll
    for (int i = 0; i < 9; i++) { pu[i] = u[i]; pv[i] = v[i]; }
}
// two stages at once on four LDS slots (one LDS round trip, one carry pass for two stages): x1 = v1 wa, x3 = v3 wa, then (p0 +- x1), (p2 +- x3),
// then the second stage pairs (0, 2) with wb and (1, 3) with wc.  Which slots are 0..3 differs between the RN and NR orders; the arithmetic is the
// same.  Limbs: a carried value plus a dominator minus a product is below 2^30.6, and 2^30.6 x 2^29 still fits the column sums of the next product.
__device__ __forceinline__ void ntt_r4(uint32_t* p0, uint32_t* p1, uint32_t* p2, uint32_t* p3, const uint32_t wa[9], const uint32_t wb[9], const uint32_t wc[9], bool swap_mid) {
    uint32_t a[9], b[9], c[9], d[9], t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { a[i] = p0[i]; b[i] = p1[i]; c[i] = p2[i]; d[i] = p3[i]; }
    // first stage: (a, b) and (c, d) are the pairs, same twiddle
    f29_mul<FrParams>(t, b, wa);
#pragma unroll
    for (int i = 0; i < 9; i++) { b[i] = a[i] + NttDom::D24.l[i] - t[i]; a[i] += t[i]; }
    f29_mul<FrParams>(t, d, wa);
#pragma unroll
    for (int i = 0; i < 9; i++) { d[i] = c[i] + NttDom::D24.l[i] - t[i]; c[i] += t[i]; }
    // second stage: (a, c) with wb, (b, d) with wc
    f29_mul<FrParams>(t, c, wb);
#pragma unroll
    for (int i = 0; i < 9; i++) { c[i] = a[i] + NttDom::D24.l[i] - t[i]; a[i] += t[i]; }
    f29_mul<FrParams>(t, d, wc);
#pragma unroll
    for (int i = 0; i < 9; i++) { d[i] = b[i] + NttDom::D24.l[i] - t[i]; b[i] += t[i]; }
    f29_carry(a); f29_carry(b); f29_carry(c); f29_carry(d);
    (void)swap_mid;
#pragma unroll
    for (int i = 0; i < 9; i++) { p0[i] = a[i]; p1[i] = b[i]; p2[i] = c[i]; p3[i] = d[i]; }
}
extern "C" __global__ void __launch_bounds__(256)
zkc_ntt_pass(const Fr* src_all, Fr* dst_all, const uint32_t* __restrict__ tw29, const Fr* __restrict__ scale,      // src_all == dst_all when ntt_pair_run calls it in place: no __restrict__ on the data pointers
             int logn, int s0, int b, int first) {
    const Fr* src = src_all + ((size_t)blockIdx.y << logn);      // blockIdx.y = vector of the batch
    Fr* dst = dst_all + ((size_t)blockIdx.y << logn);
    extern __shared__ uint32_t tile[];                                      // 9 words per element (odd stride: conflict-free)
    const int mid_n = 1 << b;
    const int lo_bits = s0;
    const int lo_t = (NTT_TILE >> b) < (1 << lo_bits) ? (NTT_TILE >> b) : (1 << lo_bits);   // neighbouring lo per tile
    const int tiles_per_hi = (1 << lo_bits) / lo_t;
    const int hi = blockIdx.x / tiles_per_hi, lo0 = (blockIdx.x % tiles_per_hi) * lo_t;
    const int elems = mid_n * lo_t;
    const size_t base = (size_t)hi << (s0 + b);
    // load: LDS slot = mid * lo_t + l  <->  global index base + mid * 2^s0 + lo0 + l
    for (int e = threadIdx.x; e < elems; e += blockDim.x) {
        const int mid = e / lo_t, l = e - mid * lo_t;
        size_t gi = base + ((size_t)mid << s0) + lo0 + l;
        if (first) gi = __brev((unsigned)gi) >> (32 - logn);
        const Fr x = ld_fr(src + gi);
        uint32_t t[9]; f29_from_fp_shl5(t, x.v);
#pragma unroll
        for (int k = 0; k < 9; k++) tile[9 * e + k] = t[k];
    }
    __syncthreads();
    // [r2] stages two at a time where possible (ntt_r4: one LDS round trip and one carry pass for two stages, three twiddles for four
    // butterflies); the twiddle-free stage 1 of a transform and an odd stage left over run one at a time.
    int t = 1;
    auto single = [&](int t1, bool carry) {
        const int half = 1 << (t1 - 1);
        const int s = s0 + t1;                                    // global stage, m = 2^s
        for (int q = threadIdx.x; q < elems / 2; q += blockDim.x) {
            const int l = q % lo_t, pr = q / lo_t;                // pr indexes the (mid) butterfly pair
            const int j = pr & (half - 1), blk = pr >> (t1 - 1);
            const int m0 = (blk << t1) + j, m1 = m0 + half;
            const unsigned k = ((unsigned)j << s0) + lo0 + l;     // butterfly index within the half-block of size 2^(s-1)
            uint32_t w[9]; ntt_ld_w(w, tw29, (size_t)k << (logn - s));
            ntt_bfly(tile + 9 * (m0 * lo_t + l), tile + 9 * (m1 * lo_t + l), w, s == 1, carry);   // s == 1: every twiddle is w^0 = 1, no product (v < 32 p: dominator D27)
        }
        __syncthreads();
    };
    if (s0 == 0) { single(1, true); t = 2; }
    for (; t + 1 <= b; t += 2) {
        const int half = 1 << (t - 1), s = s0 + t;
        for (int q = threadIdx.x; q < elems / 4; q += blockDim.x) {
            const int l = q % lo_t, u = q / lo_t;
            const int j = u & (half - 1), blk = u >> (t - 1);
            const int m = (blk << (t + 1)) + j;
            const unsigned k = ((unsigned)j << s0) + lo0 + l, kh = k + ((unsigned)half << s0);
            uint32_t wa[9], wb[9], wc[9];
            ntt_ld_w(wa, tw29, (size_t)k << (logn - s)); ntt_ld_w(wb, tw29, (size_t)k << (logn - s - 1)); ntt_ld_w(wc, tw29, (size_t)kh << (logn - s - 1));
            uint32_t* p = tile + 9 * (m * lo_t + l);
            ntt_r4(p, p + 9 * half * lo_t, p + 9 * 2 * half * lo_t, p + 9 * 3 * half * lo_t, wa, wb, wc, false);
        }
        __syncthreads();
    }
    if (t <= b) single(t, true);
    for (int e = threadIdx.x; e < elems; e += blockDim.x) {
        const int mid = e / lo_t, l = e - mid * lo_t;
        const size_t gi = base + ((size_t)mid << s0) + lo0 + l;
        uint32_t r[9];
#pragma unroll
        for (int k = 0; k < 9; k++) r[k] = tile[9 * e + k];
        if (scale) { const Fr sc = ld_fr(scale + gi); uint32_t s29[9], o[9]; f29_from_fp_shl5(s29, sc.v); f29_mul<FrParams>(o, r, s29);
#pragma unroll
            for (int k = 0; k < 9; k++) r[k] = o[k]; }
        else f29_reduce_small<FrParams>(r);
        st_fr(dst + gi, f29_to_fp<FrParams>(r));
    }
}

// ================= [round 2] the transform PAIR of the prover without bit reversal and with three HBM round trips =================
// h_evals needs, per vector, iNTT -> multiply by g^i / n -> NTT.  Both transforms use multiply-then-add (Cooley-Tukey) butterflies:
//   inverse : natural input -> BIT-REVERSED output ("NR": spans n/2 ... 1; every butterfly of block i of a stage with 2^q blocks uses the one
//             twiddle w^-(brev_q(i) n / 2^(q+1)))
//   forward : bit-reversed input -> natural output ("RN": spans 1 ... n/2, the stages zkc_ntt_pass runs)
// so position p simply holds X[brev(p)] in between, the scale table is stored bit-reversed once per key, and no load or store of any pass is a
// permutation (round 1 gathered 32-byte elements by bit-reversed index in the first pass of both transforms).  The last nine NR stages, the
// scaling and the first nine RN stages all live on the same 512 consecutive positions, so they share one LDS residency:
//   zkc_ntt_nr_head   NR stages 0 .. logn-10 : tile = (top bits) x 4 neighbouring positions           (HBM round trip 1)
//   zkc_ntt_mid       NR stages logn-9 .. logn-1, x scale_br, RN stages 1 .. 9 on 2 x 512 positions    (HBM round trip 2)
//   zkc_ntt_pass      RN stages 10 .. logn (the existing kernel, first = 0)                            (HBM round trip 3)
// Element format between kernels stays 8 x u32 Montgomery (R = 2^256); inside a tile nine 29-bit limbs (R' = 2^261) as in zkc_ntt_pass.
// NR stages q0 .. q0+b-1 over a tile whose `mid` index (b bits) is the position bits those stages pair up; slot(mid, l) gives the LDS slot.
// prefix = the position bits above mid (q0 of them): block index of stage q0 + r is (prefix << r) | (mid >> (b - r)).
template <int RADIX, class Slot>
__device__ __forceinline__ void ntt_nr_stages(uint32_t* tile, Slot slot, int b, int lo_t, int q0, uint32_t prefix_of_l0, int prefix_per_l, const uint32_t* __restrict__ tw29, int logn, bool fresh) {
    const int mid_n = 1 << b;
    auto tw_of = [&](int q, uint32_t i) -> size_t { return q ? (size_t)((__brev(i) >> (32 - q)) << (logn - q - 1)) : (size_t)0; };      // block i of the stage with 2^q blocks
    int r = 0;
    if (fresh) {                                    // stage 0 of a transform: twiddle 1, operands as large as 32 p -> plain butterflies, alone
        const int span = mid_n >> 1, pairs = span * lo_t;
        for (int x = threadIdx.x; x < pairs; x += blockDim.x) {
            const int l = x % lo_t, j = x / lo_t;
            uint32_t w[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            ntt_bfly(tile + 9 * slot(j, l), tile + 9 * slot(j + span, l), w, true, true);
        }
        __syncthreads();
        r = 1;
    }
    for (; RADIX >= 4 && r + 1 < b; r += 2) {               // stages q, q + 1 together: slots m, m + span/2, m + span, m + 3 span/2
        const int q = q0 + r, span = mid_n >> (r + 1), hspan = span >> 1, units = (mid_n >> 2) * lo_t;
        for (int x = threadIdx.x; x < units; x += blockDim.x) {
            const int l = x % lo_t, u = x / lo_t;
            const int blk = u / hspan, j = u - blk * hspan;
            const int m = blk * 2 * span + j;
            const uint32_t i = ((prefix_of_l0 + (uint32_t)(prefix_per_l * l)) << r) | (uint32_t)blk;
            uint32_t wa[9], wb[9], wc[9];
            ntt_ld_w(wa, tw29, tw_of(q, i)); ntt_ld_w(wb, tw29, tw_of(q + 1, 2 * i)); ntt_ld_w(wc, tw29, tw_of(q + 1, 2 * i + 1));
            // stage q pairs (m, m + span) and (m + hspan, m + span + hspan); stage q + 1 pairs (m, m + hspan) in block 2i and (m + span, m + span + hspan) in block 2i + 1
            // ntt_r4 takes (a, b), (c, d) as first-stage pairs and (a, c), (b, d) as second-stage pairs: a = m, b = m + span, c = m + hspan, d = m + span + hspan
            ntt_r4(tile + 9 * slot(m, l), tile + 9 * slot(m + span, l), tile + 9 * slot(m + hspan, l), tile + 9 * slot(m + span + hspan, l), wa, wb, wc, false);
        }
        __syncthreads();
    }
    for (; r < b; r++) {                            // one stage at a time (the odd one left over, or all of them: the head kernel measured slower two at a time)
        const int q = q0 + r, span = mid_n >> (r + 1), pairs = (mid_n >> 1) * lo_t;
        for (int x = threadIdx.x; x < pairs; x += blockDim.x) {
            const int l = x % lo_t, pr = x / lo_t;
            const int blk = pr / span, j = pr - blk * span;
            const int m0 = blk * 2 * span + j;
            const uint32_t i = ((prefix_of_l0 + (uint32_t)(prefix_per_l * l)) << r) | (uint32_t)blk;
            uint32_t w[9]; ntt_ld_w(w, tw29, tw_of(q, i));
            ntt_bfly(tile + 9 * slot(m0, l), tile + 9 * slot(m0 + span, l), w, false, RADIX >= 4 || (r & 1) == 1 || r == b - 1);
        }
        __syncthreads();
    }
}
template <int RADIX>
__device__ __forceinline__ void ntt_nr_head_body(const Fr* src_all, Fr* dst_all, const uint32_t* __restrict__ tw29, int logn, int q0, int b) {      // src_all == dst_all: in place
    // NR stages q0 .. q0+b-1.  tile: mid = the b position bits below the top q0 (stride 2^sh, sh = logn-q0-b), lo_t = NTT_TILE >> b neighbouring positions starting
    // at lo0; the top q0 bits (`prefix`) are fixed per block.  q0 = 0: the first kernel of a transform (stage 0 is twiddle-free, operands as large as 32 p).
    // [r4] domains above 2^18 run two of these in a row (q0 = 0, then q0 = b of the first): the mid kernel always takes the last nine NR stages.
    const Fr* src = src_all + ((size_t)blockIdx.y << logn);
    Fr* dst = dst_all + ((size_t)blockIdx.y << logn);
    extern __shared__ uint32_t tile[];
    const int mid_n = 1 << b, sh = logn - q0 - b, lo_t = NTT_TILE >> b, elems = mid_n * lo_t;
    const int bpp = (1 << sh) / lo_t;                                    // blocks per prefix
    const uint32_t prefix = blockIdx.x / bpp;
    const size_t base = ((size_t)prefix << (logn - q0)) + (size_t)(blockIdx.x % bpp) * lo_t;
    for (int e = threadIdx.x; e < elems; e += blockDim.x) {
        const int mid = e / lo_t, l = e - mid * lo_t;
        const Fr x = ld_fr(src + (base + ((size_t)mid << sh) + l));
        uint32_t t[9]; f29_from_fp_shl5(t, x.v);
#pragma unroll
        for (int k = 0; k < 9; k++) tile[9 * e + k] = t[k];
    }
    __syncthreads();
    ntt_nr_stages<RADIX>(tile, [lo_t](int mid, int l) { return mid * lo_t + l; }, b, lo_t, q0, prefix, 0, tw29, logn, q0 == 0);
    for (int e = threadIdx.x; e < elems; e += blockDim.x) {
        const int mid = e / lo_t, l = e - mid * lo_t;
        uint32_t r[9];
#pragma unroll
        for (int k = 0; k < 9; k++) r[k] = tile[9 * e + k];
        f29_reduce_small<FrParams>(r);
        st_fr(dst + (base + ((size_t)mid << sh) + l), f29_to_fp<FrParams>(r));
    }
}
extern "C" __global__ void __launch_bounds__(256)
zkc_ntt_nr_head(const Fr* src_all, Fr* dst_all, const uint32_t* __restrict__ tw29, int logn, int q0, int b) { ntt_nr_head_body<1>(src_all, dst_all, tw29, logn, q0, b); }
extern "C" __global__ void __launch_bounds__(256)
zkc_ntt_nr_head_r4(const Fr* src_all, Fr* dst_all, const uint32_t* __restrict__ tw29, int logn, int q0, int b) { ntt_nr_head_body<4>(src_all, dst_all, tw29, logn, q0, b); }
// NR stages logn-9 .. logn-1 of the inverse, the scale, RN stages 1 .. 9 of the forward transform: NTT_TILE consecutive positions = two blocks of 512
extern "C" __global__ void __launch_bounds__(256)
zkc_ntt_mid(Fr* __restrict__ data_all, const uint32_t* __restrict__ tw_inv29, const uint32_t* __restrict__ tw_fwd29, const Fr* __restrict__ scale_br, int logn) {
    Fr* __restrict__ data = data_all + ((size_t)blockIdx.y << logn);
    extern __shared__ uint32_t tile[];
    constexpr int B = 9, MID = 1 << B, NSUB = NTT_TILE / MID;
    const size_t base = (size_t)blockIdx.x * NTT_TILE;
    for (int e = threadIdx.x; e < NTT_TILE; e += blockDim.x) {
        const Fr x = ld_fr(data + base + e);
        uint32_t t[9]; f29_from_fp_shl5(t, x.v);
#pragma unroll
        for (int k = 0; k < 9; k++) tile[9 * e + k] = t[k];
    }
    __syncthreads();
    // slot(mid, l) = l * 512 + mid : sub-block l of the tile, position mid inside it; its prefix = the logn-9 position bits above = blockIdx.x * NSUB + l
    ntt_nr_stages<4>(tile, [](int mid, int l) { return l * MID + mid; }, B, NSUB, logn - B, (uint32_t)blockIdx.x * NSUB, 1, tw_inv29, logn, false);
    // x g^k / n at bit-reversed positions (scale_br[p] = scale[brev(p)]): a product, so the value is back below 2 p
    for (int e = threadIdx.x; e < NTT_TILE; e += blockDim.x) {
        uint32_t r[9], s29[9], o[9];
#pragma unroll
        for (int k = 0; k < 9; k++) r[k] = tile[9 * e + k];
        const Fr sc = ld_fr(scale_br + base + e);
        f29_from_fp_shl5(s29, sc.v); f29_mul<FrParams>(o, r, s29);
#pragma unroll
        for (int k = 0; k < 9; k++) tile[9 * e + k] = o[k];
    }
    __syncthreads();
    // RN stages 1 .. 9 (what zkc_ntt_pass does in its first pass, minus the bit-reversed load): stage t pairs mid and mid + 2^(t-1).
    // Stage 1 alone (twiddle 1, operands below 18 p after the scaling product), then (2,3) (4,5) (6,7) (8,9) two at a time.
    for (int x = threadIdx.x; x < NTT_TILE / 2; x += blockDim.x) {
        const int l = x / (MID / 2), pr = x - l * (MID / 2);
        uint32_t w[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        ntt_bfly(tile + 9 * (l * MID + 2 * pr), tile + 9 * (l * MID + 2 * pr + 1), w, true, true);
    }
    __syncthreads();
    for (int t = 2; t + 1 <= B; t += 2) {
        const int half = 1 << (t - 1);              // stage t: pairs (m, m + half), j < half; stage t + 1: pairs (m, m + 2 half) with j and (m + half, m + 3 half) with j + half
        for (int x = threadIdx.x; x < NTT_TILE / 4; x += blockDim.x) {
            const int l = x / (MID / 4), u = x - l * (MID / 4);
            const int j = u & (half - 1), blk = u >> (t - 1);
            const int m = (blk << (t + 1)) + j;
            uint32_t wa[9], wb[9], wc[9];
            ntt_ld_w(wa, tw_fwd29, (size_t)j << (logn - t)); ntt_ld_w(wb, tw_fwd29, (size_t)j << (logn - t - 1)); ntt_ld_w(wc, tw_fwd29, (size_t)(j + half) << (logn - t - 1));
            uint32_t* p = tile + 9 * (l * MID + m);
            ntt_r4(p, p + 9 * half, p + 9 * 2 * half, p + 9 * 3 * half, wa, wb, wc, false);
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < NTT_TILE; e += blockDim.x) {
        uint32_t r[9];
#pragma unroll
        for (int k = 0; k < 9; k++) r[k] = tile[9 * e + k];
        f29_reduce_small<FrParams>(r);
        st_fr(data + base + e, f29_to_fp<FrParams>(r));
    }
}
extern "C" __global__ void __launch_bounds__(256)
zkc_bitrev_copy(const Fr* __restrict__ src, Fr* __restrict__ dst, int logn) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> logn) return;
    st_fr(dst + i, ld_fr(src + (__brev(i) >> (32 - logn))));
}
int ntt_bitrev_table(zkc_ctx* ctx, const Fr* d_src, Fr** out, int logn) {
    ZKC_HIP_CHECK(ctx, hipMalloc((void**)out, sizeof(Fr) << logn));
    hipLaunchKernelGGL(zkc_bitrev_copy, dim3(((1u << logn) + 255) / 256), dim3(256), 0, ctx->stream, d_src, *out, logn);
    ZKC_HIP_CHECK(ctx, hipGetLastError());
    return ZKC_OK;
}
// the prover's transform pair on `nvec` contiguous vectors: data <- NTT(scale x iNTT(data)), in place (tmp is not needed); 12 <= logn <= 27.
// Up to 2^18: head (NR 0 .. logn-10) -> mid -> tail (RN 10 .. logn), three HBM round trips.  [r4] Above (BASELINE configs[4]: a 2^20 domain, the ceiling of the reference's
// powers of tau, circuit/circuit-compiler.sh:57) the logn-9 head stages and the logn-9 tail stages no longer fit one tile of 1024 elements with whole cache lines per row,
// so each side is two kernels: five round trips, still in place and still without a single permuted access (the stand-alone ntt_run needs six, two of them bit-reversed gathers).
int ntt_pair_run(zkc_ctx* ctx, hipStream_t st, Fr* data, const uint32_t* tw_inv29, const uint32_t* tw_fwd29, const Fr* scale_br, int logn, int nvec) {
    if (logn < 12 || logn > 27) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "ntt_pair_run: 12 <= logn <= 27");
    const int bh = logn - 9;                                     // head: NR stages 0 .. logn-10, tail: RN stages 10 .. logn (bh stages each)
    const int b1 = bh <= 9 ? bh : (bh + 1) / 2, b2 = bh - b1;    // one kernel each side up to 2^18, two above
    const unsigned blocks = (1u << logn) / NTT_TILE;
    // [r4] the head kernel runs its stages two at a time like the other two (one LDS round trip, three twiddles and one carry pass per two stages: 290 instead of ~365
    // instructions per butterfly; alternating on one box 3196 / 3166 against 3175 / 3146 proofs/s, +0.65 %, equal on a second box).  ZKC_NTT_RADIX=1: one stage at a time, the
    // round-2 form ("measured slower two at a time" was true of the round-2 pipeline, whose transforms ran alone).  Three stages at a time on eight slots was built and measured
    // too: 137 instead of 145 instructions per element and stage, but 162-168 VGPRs (three waves per SIMD instead of four) -- 2856 / 2845 against 2997 / 2990 proofs/s, -5 %; removed.
    static const int radix = [] { const char* e = getenv("ZKC_NTT_RADIX"); return e ? atoi(e) : 4; }();
    auto head = radix >= 4 ? zkc_ntt_nr_head_r4 : zkc_ntt_nr_head;
    hipLaunchKernelGGL(head, dim3(blocks, nvec), dim3(256), (size_t)NTT_TILE * 36, st, (const Fr*)data, data, tw_inv29, logn, 0, b1);
    if (b2) hipLaunchKernelGGL(head, dim3(blocks, nvec), dim3(256), (size_t)NTT_TILE * 36, st, (const Fr*)data, data, tw_inv29, logn, b1, b2);
    hipLaunchKernelGGL(zkc_ntt_mid, dim3(blocks, nvec), dim3(256), (size_t)NTT_TILE * 36, st, data, tw_inv29, tw_fwd29, scale_br, logn);
    hipLaunchKernelGGL(zkc_ntt_pass, dim3(blocks, nvec), dim3(256), (size_t)NTT_TILE * 36, st, (const Fr*)data, data, tw_fwd29, (const Fr*)nullptr, logn, 9, b1, 0);
    if (b2) hipLaunchKernelGGL(zkc_ntt_pass, dim3(blocks, nvec), dim3(256), (size_t)NTT_TILE * 36, st, (const Fr*)data, data, tw_fwd29, (const Fr*)nullptr, logn, 9 + b1, b2, 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return zkc_fail(ctx, ZKC_ERR_HIP, std::string("ntt_pair_run: ") + hipGetErrorString(e));
    return ZKC_OK;
}

// twiddle table (n/2 Montgomery-form Fr, device) -> the 12-word limb form the pass kernel reads
int ntt_make_tw29(zkc_ctx* ctx, const Fr* d_tw, uint32_t count, uint32_t** out) {
    ZKC_HIP_CHECK(ctx, hipMalloc((void**)out, (size_t)count * TW29_WORDS * 4));
    hipLaunchKernelGGL(zkc_tw29, dim3((count + 255) / 256), dim3(256), 0, ctx->stream, d_tw, *out, count);
    ZKC_HIP_CHECK(ctx, hipGetLastError());
    ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ZKC_OK;
}

// Full transform of `nvec` contiguous vectors src -> dst (src must differ from dst: the first pass scatters by bit reversal).
int ntt_run(zkc_ctx* ctx, hipStream_t st, const Fr* src, Fr* dst, const uint32_t* tw29, const Fr* scale, int logn, int nvec) {
    int s0 = 0; bool first = true;
    while (s0 < logn) {
        int b = logn - s0 < 9 ? logn - s0 : 9;
        if (logn - s0 > 9 && logn - s0 < 18) b = (logn - s0 + 1) / 2;       // balance the last two passes (17 -> 9 + 8)
        if (b > 9) b = 9;
        const int lo_t = (NTT_TILE >> b) < (1 << s0) ? (NTT_TILE >> b) : (1 << s0);
        const int nblocks = (1 << logn) / ((1 << b) * lo_t);
        const bool last = s0 + b == logn;
        hipLaunchKernelGGL(zkc_ntt_pass, dim3(nblocks, nvec), dim3(256), (size_t)(1 << b) * lo_t * 36, st,
                           first ? src : dst, dst, tw29, last ? scale : nullptr, logn, s0, b, first ? 1 : 0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return zkc_fail(ctx, ZKC_ERR_HIP, std::string("zkc_ntt_pass: ") + hipGetErrorString(e));
        s0 += b; first = false;
    }
    return ZKC_OK;
}

}  // namespace zkc
