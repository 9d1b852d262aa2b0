// zkc_finalize.hip -- K9: the constant-size blinding step of Groth16 on the device (stage a7): one lane per scalar product and then one lane per
// proof element for a pass of proofs (zkc_finalize_products / _combine), one workgroup of four waves for a single proof (zkc_finalize).
//
//   piA = alpha + A + r delta ;  piB = beta2 + B2 + s delta2 ;  piC = C + H + s piA + r piB1 - r s delta
// (snarkjs groth16_prove.js tail / rapidsnark; reached from ts_inputs/src/example.ts:358-362, zk_census_test.go:89).
// Expanded so that nothing depends on piA / piB1:  s piA + r piB1 - rs delta = s A' + s alpha + r B1' + r beta1 + rs delta,
// i.e. two variable-base products and five fixed-base ones in G1 plus one in G2, the fixed-base ones read from 8-bit window tables.
// Runs on the lane's blinding stream so that it overlaps the next pipeline pass.
#include "zkc_prover.h"
#include "zkc_f29_g1.h"
#include "zkc_f29_g2.h"

namespace zkc {

template <class F>
__device__ XYZZ<F> fb_mul(const Affine<F>* __restrict__ tab, const uint32_t k[8]) {
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int w = 0; w < 32; w++) {
        uint32_t limb = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) limb = (q == (w >> 2)) ? k[q] : limb;
        const uint32_t d = (limb >> (8 * (w & 3))) & 0xff;
        if (d) acc = xyzz_add_affine(acc, tab[w * 255 + d - 1]);
    }
    return acc;
}
__device__ void store_fq_std(uint8_t* out, const Fq& a) {
    uint32_t s[8]; fp_to_std<FqParams>(s, a);
    uint4* d = reinterpret_cast<uint4*>(out);
    d[0] = make_uint4(s[0], s[1], s[2], s[3]); d[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// k * P for a 254-bit k: 4-bit fixed windows over radix-2^29 coordinates (zkc_f29_g1.h): 14 additions for the table, then 64 x
// (4 doublings + 1 addition).  tab: 16 entries in LDS owned by the calling lane.
__device__ G1XYZZ var_mul29(const G1XYZZ& P, const uint32_t k[8], Acc29* tab) {
    if (P.is_inf()) return P;
    f29_pt_set_inf(tab[0]); tab[1] = f29_pt_from_xyzz(P);
    for (int i = 2; i < 16; i++) { Acc29 t = tab[i - 1]; f29_pt_add(t, t, tab[1]); tab[i] = t; }
    Acc29 acc; f29_pt_set_inf(acc);
    for (int w = 63; w >= 0; w--) {
        if (!f29_pt_is_inf(acc)) for (int d = 0; d < 4; d++) f29_pt_dbl(acc, acc);
        uint32_t limb = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) limb = (q == (w >> 3)) ? k[q] : limb;
        const uint32_t dg = (limb >> (4 * (w & 7))) & 15u;
        if (dg) f29_pt_add(acc, acc, tab[dg]);
    }
    return f29_pt_is_inf(acc) ? G1XYZZ::inf() : f29_pt_to_xyzz(acc);
}

// the same product with the 16-entry table in global scratch (L2-resident: 2.3 KB per product), for the lane-per-product kernel below
__device__ G1XYZZ var_mul29_g(const G1XYZZ& P, const uint32_t k[8], Acc29* __restrict__ tab) {
    if (P.is_inf()) return P;
    { Acc29 one = f29_pt_from_xyzz(P), t = one; tab[1] = one; for (int i = 2; i < 16; i++) { f29_pt_add(t, t, one); tab[i] = t; } }
    Acc29 acc; f29_pt_set_inf(acc);
    for (int w = 63; w >= 0; w--) {
        if (!f29_pt_is_inf(acc)) for (int d = 0; d < 4; d++) f29_pt_dbl(acc, acc);
        uint32_t limb = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) limb = (q == (w >> 3)) ? k[q] : limb;
        const uint32_t dg = (limb >> (4 * (w & 7))) & 15u;
        if (dg) { const Acc29 e = tab[dg]; f29_pt_add(acc, acc, e); }
    }
    return f29_pt_is_inf(acc) ? G1XYZZ::inf() : f29_pt_to_xyzz(acc);
}

// the voter-independent part of a section's MSM for proof q (constant folding, zkc_prove.hip): base + sum over the folded levels of both trees
template <class P>
__device__ P fold_const(const P* __restrict__ tab, const FinalizeArgs& a, int q) {
    if (a.dc[q] == 255) return P::inf();
    return xyzz_add(tab[0], xyzz_add(tab[1 + a.dc[q]], tab[1 + a.fold_n + a.ds[q]]));
}
// One workgroup of four waves per proof, one task per wave so that the three independent latency chains run side by side:
//   wave 0: s A' and r B1' (lanes 0, 1; variable base), then piC once everything else has arrived
//   wave 1: r delta, s delta, rs delta, s alpha, r beta1 from the 8-bit fixed-base tables (lanes 0..4), lane 0 goes on to piA
//   wave 2: s delta2 in G2 and piB
extern "C" __global__ void __launch_bounds__(256)
zkc_finalize(FinalizeArgs a) {
    __shared__ G1XYZZ sh[8];
    __shared__ Acc29 tab[2][16];
    const int q = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t r[8], s[8];
    { const uint4* p = reinterpret_cast<const uint4*>(a.rs + 64 * (size_t)q); uint4 x = p[0], y = p[1], z = p[2], w = p[3];
      r[0] = x.x; r[1] = x.y; r[2] = x.z; r[3] = x.w; r[4] = y.x; r[5] = y.y; r[6] = y.z; r[7] = y.w;
      s[0] = z.x; s[1] = z.y; s[2] = z.z; s[3] = z.w; s[4] = w.x; s[5] = w.y; s[6] = w.z; s[7] = w.w; }
    const int nq = gridDim.x;                         // results of a pass: H_0 .. H_{nq-1}, then A_q, B1_q, C_q per proof
    uint8_t* out = a.out + 256 * (size_t)q;
    if (wave == 0 && lane < 2) {
        const G1XYZZ P = lane == 0 ? xyzz_add(a.r1[nq + 3 * q + 0], fold_const(a.foldA, a, q)) : xyzz_add(a.r1[nq + 3 * q + 1], fold_const(a.foldB1, a, q));
        sh[lane] = var_mul29(P, lane == 0 ? s : r, tab[lane]);                                 // s A' , r B1'
    } else if (wave == 1 && lane < 5) {
        uint32_t k[8];
        if (lane == 2) { Fr rs = fp_from_std<FrParams>(r) * fp_from_std<FrParams>(s); fp_to_std<FrParams>(k, rs); }
        else {
#pragma unroll
            for (int i = 0; i < 8; i++) k[i] = (lane == 0 || lane == 4) ? r[i] : s[i];
        }
        const G1Affine* tb = lane <= 2 ? a.tblDelta1 : lane == 3 ? a.tblAlpha1 : a.tblBeta1;       // r d, s d, rs d, s alpha, r beta1
        const G1XYZZ v = fb_mul<Fq>(tb, k);
        sh[2 + lane] = v;
        if (lane == 0) {        // piA = A' + alpha + r delta
            const G1XYZZ A = xyzz_add(a.r1[nq + 3 * q + 0], fold_const(a.foldA, a, q));
            G1Affine p = xyzz_to_affine(xyzz_add(xyzz_add_affine(A, a.alpha1), v));
            store_fq_std(out, p.x); store_fq_std(out + 32, p.y);
        }
    } else if (wave == 2 && lane == 0) {     // piB = B2' + beta2 + s delta2
        const G2XYZZ sd = fb_mul<Fq2>(a.tblDelta2, s);
        G2Affine p = xyzz_to_affine(xyzz_add(xyzz_add_affine(xyzz_add(a.r2[q], fold_const(a.foldB2, a, q)), a.beta2), sd));
        store_fq_std(out + 64, p.x.c0); store_fq_std(out + 96, p.x.c1); store_fq_std(out + 128, p.y.c0); store_fq_std(out + 160, p.y.c1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {     // piC = C' + H + s A' + s alpha + r B1' + r beta1 + rs delta
        G1XYZZ c = xyzz_add(xyzz_add(a.r1[nq + 3 * q + 2], fold_const(a.foldC, a, q)), a.r1[q]);
        c = xyzz_add(c, sh[0]); c = xyzz_add(c, sh[5]); c = xyzz_add(c, sh[1]); c = xyzz_add(c, sh[6]); c = xyzz_add(c, sh[4]);
        G1Affine p = xyzz_to_affine(c);
        store_fq_std(out + 192, p.x); store_fq_std(out + 224, p.y);
    }
}

// ---- [r2] the same step with one LANE per scalar product instead of one wave: a pass of 94 proofs was 282 waves with one to five live lanes each
// (every wave-instruction costs the SIMD its four cycles whatever the lanes do: 1.1 % of a pass' VALU cycles, and 3.9 ms at the end of a step);
// it is 22 dense waves now.  Kernel 1, grid (ceil(n / 64), 8 tasks), lane = proof: task 0 s A', 1 r B1' (variable base, table in scratch),
// 2..6 r delta, s delta, rs delta, s alpha, r beta1 (fixed-base tables), 7 s delta2 in G2.  Kernel 2, grid (ceil(n / 64), 3): piA, piC, piB.
__device__ __forceinline__ void load_rs(const FinalizeArgs& a, int q, uint32_t r[8], uint32_t s[8]) {
    const uint4* p = reinterpret_cast<const uint4*>(a.rs + 64 * (size_t)q); const uint4 x = p[0], y = p[1], z = p[2], w = p[3];
    r[0] = x.x; r[1] = x.y; r[2] = x.z; r[3] = x.w; r[4] = y.x; r[5] = y.y; r[6] = y.z; r[7] = y.w;
    s[0] = z.x; s[1] = z.y; s[2] = z.z; s[3] = z.w; s[4] = w.x; s[5] = w.y; s[6] = w.z; s[7] = w.w;
}
extern "C" __global__ void __launch_bounds__(64)
zkc_finalize_products(FinalizeArgs a, int nq) {
    const int q = blockIdx.x * 64 + threadIdx.x, task = blockIdx.y;
    if (q >= nq) return;
    uint32_t r[8], s[8]; load_rs(a, q, r, s);
    G1XYZZ* res1 = reinterpret_cast<G1XYZZ*>(a.scratch);                       // [7][nq] G1 results, then [nq] G2 results, then the tables [2][nq][16]
    G2XYZZ* res2 = reinterpret_cast<G2XYZZ*>(res1 + 7 * (size_t)nq);
    Acc29* tabs = reinterpret_cast<Acc29*>(res2 + nq);
    if (task < 2) {
        const G1XYZZ P = task == 0 ? xyzz_add(a.r1[nq + 3 * q + 0], fold_const(a.foldA, a, q)) : xyzz_add(a.r1[nq + 3 * q + 1], fold_const(a.foldB1, a, q));
        res1[(size_t)task * nq + q] = var_mul29_g(P, task == 0 ? s : r, tabs + ((size_t)task * nq + q) * 16);                 // s A' , r B1'
    } else if (task == 3) {
        return;                                                            // s delta in G1 is not part of any proof element (slot kept so that the task numbers read like the formula)
    } else if (task < 7) {
        uint32_t k[8];
        if (task == 4) { Fr rs = fp_from_std<FrParams>(r) * fp_from_std<FrParams>(s); fp_to_std<FrParams>(k, rs); }
        else {
#pragma unroll
            for (int i = 0; i < 8; i++) k[i] = (task == 2 || task == 6) ? r[i] : s[i];
        }
        const G1Affine* tb = task <= 4 ? a.tblDelta1 : task == 5 ? a.tblAlpha1 : a.tblBeta1;       // r d, s d, rs d, s alpha, r beta1
        res1[(size_t)task * nq + q] = fb_mul<Fq>(tb, k);
    } else {
        res2[q] = fb_mul<Fq2>(a.tblDelta2, s);
    }
}
extern "C" __global__ void __launch_bounds__(64)
zkc_finalize_combine(FinalizeArgs a, int nq) {
    const int q = blockIdx.x * 64 + threadIdx.x, role = blockIdx.y;
    if (q >= nq) return;
    const G1XYZZ* res1 = reinterpret_cast<const G1XYZZ*>(a.scratch);
    const G2XYZZ* res2 = reinterpret_cast<const G2XYZZ*>(res1 + 7 * (size_t)nq);
    uint8_t* out = a.out + 256 * (size_t)q;
    if (role == 0) {            // piA = A' + alpha + r delta
        const G1XYZZ A = xyzz_add(a.r1[nq + 3 * q + 0], fold_const(a.foldA, a, q));
        G1Affine p = xyzz_to_affine(xyzz_add(xyzz_add_affine(A, a.alpha1), res1[2 * (size_t)nq + q]));
        store_fq_std(out, p.x); store_fq_std(out + 32, p.y);
    } else if (role == 1) {     // piC = C' + H + s A' + s alpha + r B1' + r beta1 + rs delta     (same order of additions as the one-wave-per-task kernel)
        G1XYZZ c = xyzz_add(xyzz_add(a.r1[nq + 3 * q + 2], fold_const(a.foldC, a, q)), a.r1[q]);
        c = xyzz_add(c, res1[0 * (size_t)nq + q]); c = xyzz_add(c, res1[5 * (size_t)nq + q]); c = xyzz_add(c, res1[1 * (size_t)nq + q]);
        c = xyzz_add(c, res1[6 * (size_t)nq + q]); c = xyzz_add(c, res1[4 * (size_t)nq + q]);
        G1Affine p = xyzz_to_affine(c);
        store_fq_std(out + 192, p.x); store_fq_std(out + 224, p.y);
    } else {                    // piB = B2' + beta2 + s delta2
        G2Affine p = xyzz_to_affine(xyzz_add(xyzz_add_affine(xyzz_add(a.r2[q], fold_const(a.foldB2, a, q)), a.beta2), res2[q]));
        store_fq_std(out + 64, p.x.c0); store_fq_std(out + 96, p.x.c1); store_fq_std(out + 128, p.y.c0); store_fq_std(out + 160, p.y.c1);
    }
}

// ---- [r3] passes of one or two proofs: no variable-base product at all ----
// A lone proof pays for the LENGTH of the blinding's chains, not for their work: s A' and r B1' were 252 doublings and 78 additions in a row on one lane (0.87 ms of
// a 4.7 ms proof, profiles/r03_single_proof_timeline.txt), beside 32 G2 additions and three square-and-multiply inversions in 8 x u32 code.  Here
//   s A' = s K_A + sum_i (s w_i) A_i ,  r B1' = r K_B1 + sum_i (r w_i) B1_i     (K: the proof's folded constant, i: the wires that stay in its MSMs)
// the two sums are two more jobs of the pass' G1 MSM (zkc_blind_scalars writes s w_i and r w_i; a lone proof leaves the chip idle anyway), and every other
// product has a fixed base: delta, alpha (+ base constant), beta1 (+ base constant), the per-depth suffix constants of both trees, delta2.  A fixed-base product
// is read from a 4-bit window table -- T[base][w][d-1] = d 16^w base, built on the device when the key is loaded (zkc_fb4_build) -- one window per LANE, and summed
// by a six-step butterfly over the wave: six additions deep instead of 32 (or 330).  The inversions run in radix 2^29.
// Table order (G1): 0 delta1, 1 alpha1, 2 beta1, 3 alpha1 + foldA[0], 4 beta1 + foldB1[0], 5 + i: foldA[1 + i] (i < 2 n), 5 + 2 n + i: foldB1[1 + i].
extern "C" __global__ void __launch_bounds__(64) zkc_fb4_build_g1(const G1XYZZ* __restrict__ bases, int nbases, G1XYZZ* __restrict__ out) {
    const int t = blockIdx.x * 64 + threadIdx.x, b = t >> 6, w = t & 63;
    if (b >= nbases) return;
    G1XYZZ* o = out + (size_t)t * FB4_ROW;
    const G1XYZZ B = bases[b];
    if (B.is_inf()) { for (int d = 0; d < FB4_ROW; d++) o[d] = G1XYZZ::inf(); return; }
    Acc29 P = f29_pt_from_xyzz(B);
    for (int i = 0; i < 4 * w; i++) f29_pt_dbl(P, P);
    Acc29 acc = P; o[0] = f29_pt_to_xyzz(acc);
    for (int d = 1; d < FB4_ROW; d++) { f29_pt_add(acc, acc, P); o[d] = f29_pt_to_xyzz(acc); }
}
extern "C" __global__ void __launch_bounds__(64) zkc_fb4_build_g2(G2XYZZ B, G2XYZZ beta2, G2XYZZ beta2_base, G2XYZZ* __restrict__ out) {
    const int w = threadIdx.x;
    if (w == 0) { out[(size_t)FB4_WIN * FB4_ROW] = beta2; out[(size_t)FB4_WIN * FB4_ROW + 1] = beta2_base; }       // what piB adds besides s delta2 and the MSM result
    G2XYZZ* o = out + (size_t)w * FB4_ROW;
    Acc29G2 P = f29g2_pt_from_xyzz(B);
    for (int i = 0; i < 4 * w; i++) f29g2_pt_dbl(P, P);
    Acc29G2 acc = P; o[0] = f29g2_pt_to_xyzz(acc);
    for (int d = 1; d < FB4_ROW; d++) { f29g2_pt_add(acc, acc, P); o[d] = f29g2_pt_to_xyzz(acc); }
}
int fb4_build(zkc_ctx* ctx, hipStream_t st, const G1XYZZ* d_bases, int nbases, G1XYZZ* d_out, const G2XYZZ& delta2, const G2XYZZ& beta2, const G2XYZZ& beta2_base, G2XYZZ* d_out2) {
    hipLaunchKernelGGL(zkc_fb4_build_g1, dim3(nbases), dim3(64), 0, st, d_bases, nbases, d_out);
    hipLaunchKernelGGL(zkc_fb4_build_g2, dim3(1), dim3(64), 0, st, delta2, beta2, beta2_base, d_out2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return zkc_fail(ctx, ZKC_ERR_HIP, std::string("zkc_fb4_build: ") + hipGetErrorString(e));
    return ZKC_OK;
}
// out[sec][idx] = k_sec * w[idx] for the wires idx of the section's list (sec 0: the A list with k = s, sec 1: the B list with k = r), standard form in and out
extern "C" __global__ void __launch_bounds__(256) zkc_blind_scalars(BlindArgs a) {
    const int q = blockIdx.y >> 1, sec = blockIdx.y & 1;
    const uint32_t j = blockIdx.x * 256 + threadIdx.x, cnt = sec ? a.nB[q] : a.nA[q];
    if (j >= cnt) return;
    const uint32_t* map = sec ? a.mapB[q] : a.mapA[q];
    const uint32_t idx = map ? map[j] : j;
    const uint4* kp = reinterpret_cast<const uint4*>(a.rs + 64 * (size_t)q + (sec ? 0 : 32)); const uint4 k0 = kp[0], k1 = kp[1];
    const uint32_t ks[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
    const uint4* wp = reinterpret_cast<const uint4*>(a.w[q] + 8 * (size_t)idx); const uint4 w0 = wp[0], w1 = wp[1];
    const uint32_t ws[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    uint4* o = reinterpret_cast<uint4*>(a.out[q] + ((size_t)sec * a.nv + idx) * 8);
    if ((ws[0] | ws[1] | ws[2] | ws[3] | ws[4] | ws[5] | ws[6] | ws[7]) == 0) { o[0] = make_uint4(0, 0, 0, 0); o[1] = make_uint4(0, 0, 0, 0); return; }
    uint32_t pr[8]; fp_to_std<FrParams>(pr, fp_from_std<FrParams>(ws) * fp_from_std<FrParams>(ks));
    o[0] = make_uint4(pr[0], pr[1], pr[2], pr[3]); o[1] = make_uint4(pr[4], pr[5], pr[6], pr[7]);
}
int blind_scalars_launch(zkc_ctx* ctx, hipStream_t st, const BlindArgs& a, int nproofs) {
    uint32_t mx = 1; for (int q = 0; q < nproofs; q++) mx = std::max(mx, std::max(a.nA[q], a.nB[q]));
    hipLaunchKernelGGL(zkc_blind_scalars, dim3((mx + 255) / 256, 2 * nproofs), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return zkc_fail(ctx, ZKC_ERR_HIP, std::string("zkc_blind_scalars: ") + hipGetErrorString(e));
    return ZKC_OK;
}

__device__ __forceinline__ uint32_t nibble_of(const uint32_t k[8], int i) {
    uint32_t limb = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) limb = (q == (i >> 3)) ? k[q] : limb;
    return (limb >> (4 * (i & 7))) & 15u;
}
__device__ __forceinline__ void shfl_xor_words(uint32_t* dst, const uint32_t* src, int nwords, int mask) {
    for (int k = 0; k < nwords; k++) dst[k] = (uint32_t)__shfl_xor((int)src[k], mask, 64);
}
__device__ __forceinline__ Acc29 load_pt29(const G1XYZZ& p) { Acc29 a; if (p.is_inf()) f29_pt_set_inf(a); else a = f29_pt_from_xyzz(p); return a; }
__device__ __forceinline__ Acc29G2 load_pt29(const G2XYZZ& p) { Acc29G2 a; if (p.is_inf()) f29g2_pt_set_inf(a); else a = f29g2_pt_from_xyzz(p); return a; }
// The three points of a small pass' proof leave the device as XYZZ (512 bytes: piA | piB | piC) and the host divides (prove_batch_finish): one field inversion is
// ~0.15 ms as a chain of 254 dependent squarings on one lane, whatever the arithmetic, and ~10 us on a CPU core.  (The passes of many proofs keep the inversions on the device, one
// per lane: there they cost no latency and the host would pay them serially.)
__device__ __forceinline__ void store_xyzz(uint8_t* out, const Acc29& p) {
    G1XYZZ* o = reinterpret_cast<G1XYZZ*>(out);
    if (f29_pt_is_inf(p)) *o = G1XYZZ::inf(); else *o = f29_pt_to_xyzz(p);
}
__device__ __forceinline__ void store_xyzz(uint8_t* out, const Acc29G2& p) {
    G2XYZZ* o = reinterpret_cast<G2XYZZ*>(out);
    if (f29g2_pt_is_inf(p)) *o = G2XYZZ::inf(); else *o = f29g2_pt_to_xyzz(p);
}
// Three kernels, one wave per workgroup (the additions want more registers than a twelve-wave workgroup leaves them):
//   zkc_blind_tree_g1      grid (10, proofs): tasks 0..7 a fixed-base product each; 8, 9: what does not depend on them (the MSM results and the constants of
//                          piA and piC) summed meanwhile.  Results (radix 2^29) to scratch[proof][task].
//   zkc_blind_tree_g1_out  grid (proofs), two waves: piC = eight operands over eight lanes of wave 0, piA = two over two lanes of wave 1; XYZZ out.
//   zkc_blind_tree_g2      grid (proofs): piB = B2' + beta2 + s delta2.  Depends on nothing in G1: it runs on the G2 stream right behind the G2 MSM.
// MSM results of the pass: r1[q] = H_q, r1[nq + 5 q + {0..4}] = A_q, B1_q, C_q, sum (s w) A, sum (r w) B1 ; r2[q] = B2_q.
extern "C" __global__ void __launch_bounds__(64)
zkc_blind_tree_g1(FinalizeArgs a) {
    const int task = blockIdx.x, q = blockIdx.y, nq = gridDim.y, lane = threadIdx.x;
    uint32_t r[8], s[8]; load_rs(a, q, r, s);
    const bool folded = a.dc[q] != 255; const int n = a.fold_n, dc = a.dc[q], ds = a.ds[q];
    const G1XYZZ* res = a.r1 + nq + 5 * (size_t)q;
    Acc29* outv = reinterpret_cast<Acc29*>(a.scratch) + (size_t)q * 10 + task;
    Acc29 v; f29_pt_set_inf(v);
    if (task < 8) {
        int base; bool use_r;
        switch (task) {
            case 0: base = 0; use_r = true; break;                                   // r delta
            case 1: base = 0; use_r = true; break;                                   // rs delta (scalar replaced below)
            case 2: base = folded ? 3 : 1; use_r = false; break;                     // s (alpha + base constant of A)
            case 3: base = folded ? 5 + dc : -1; use_r = false; break;               // s (census-tree constant of A)
            case 4: base = folded ? 5 + n + ds : -1; use_r = false; break;           // s (sik-tree constant of A)
            case 5: base = folded ? 4 : 2; use_r = true; break;                      // r (beta1 + base constant of B1)
            case 6: base = folded ? 5 + 2 * n + dc : -1; use_r = true; break;
            default: base = folded ? 5 + 3 * n + ds : -1; use_r = true; break;
        }
        uint32_t k[8];
#pragma unroll
        for (int i = 0; i < 8; i++) k[i] = use_r ? r[i] : s[i];
        if (task == 1) { const Fr rs = fp_from_std<FrParams>(r) * fp_from_std<FrParams>(s); fp_to_std<FrParams>(k, rs); }
        if (base >= 0) {
            const uint32_t d = nibble_of(k, lane);
            if (d) v = load_pt29(a.fb4[((size_t)base * FB4_WIN + lane) * FB4_ROW + d - 1]);
            v = wave_sum_g1(v);
        }
    } else if (task == 8) {             // piA without r delta: A + alpha + [foldA constants], four operands over four lanes
        if (lane == 0) v = load_pt29(res[0]);
        else if (lane == 1) v = load_pt29(a.fb4[((size_t)(folded ? 3 : 1) * FB4_WIN) * FB4_ROW]);     // 1 x (alpha + foldA[0]) or 1 x alpha
        else if (lane == 2 && folded) v = load_pt29(a.foldA[1 + dc]);
        else if (lane == 3 && folded) v = load_pt29(a.foldA[1 + n + ds]);
        v = wave_sum_g1(v, 2);
    } else {                            // piC without the fixed-base products: C + [foldC constants] + H + sum (s w) A + sum (r w) B1, seven operands over eight lanes
        if (lane == 0) v = load_pt29(res[2]);
        else if (lane == 1 && folded) v = load_pt29(a.foldC[0]);
        else if (lane == 2 && folded) v = load_pt29(a.foldC[1 + dc]);
        else if (lane == 3 && folded) v = load_pt29(a.foldC[1 + n + ds]);
        else if (lane == 4) v = load_pt29(a.r1[q]);
        else if (lane == 5) v = load_pt29(res[3]);
        else if (lane == 6) v = load_pt29(res[4]);
        v = wave_sum_g1(v, 4);
    }
    if (lane == 0) *outv = v;
}
extern "C" __global__ void __launch_bounds__(128)
zkc_blind_tree_g1_out(FinalizeArgs a) {
    const int q = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const Acc29* sv = reinterpret_cast<const Acc29*>(a.scratch) + (size_t)q * 10;
    uint8_t* out = a.out_xyzz + 512 * (size_t)q;
    Acc29 v; f29_pt_set_inf(v);
    if (wave == 0) {            // piC = [task 9] + rs delta + s (alpha + K_A) (tasks 2, 3, 4) + r (beta1 + K_B1) (5, 6, 7): eight operands over eight lanes
        if (lane == 0) v = sv[9]; else if (lane < 8) v = sv[lane];
        v = wave_sum_g1(v, 4);
        if (lane == 0) store_xyzz(out + 384, v);
    } else {                    // piA = [task 8] + r delta
        if (lane == 0) v = sv[8]; else if (lane == 1) v = sv[0];
        v = wave_sum_g1(v, 1);
        if (lane == 0) store_xyzz(out, v);
    }
}
// piB = B2' + beta2 + s delta2: wave 0 the 64 windows of s delta2, wave 1 the four operands that do not depend on s (the MSM result, beta2 + the base constant,
// the two per-depth constants); one more addition joins them
extern "C" __global__ void __launch_bounds__(128)
zkc_blind_tree_g2(FinalizeArgs a) {
    __shared__ Acc29G2 pre;
    const int q = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool folded = a.dc[q] != 255;
    Acc29G2 v; f29g2_pt_set_inf(v);
    if (wave == 0) {
        uint32_t r[8], s[8]; load_rs(a, q, r, s);
        const uint32_t d = nibble_of(s, lane);
        if (d) v = load_pt29(a.fb4g2[(size_t)lane * FB4_ROW + d - 1]);
        v = wave_sum_g2(v, 32);
    } else {
        if (lane == 0) v = load_pt29(a.r2[q]);
        else if (lane == 1) v = load_pt29(a.fb4g2[(size_t)FB4_WIN * FB4_ROW + (folded ? 1 : 0)]);      // beta2, beta2 + foldB2[0]: two points behind the table
        else if (lane == 2 && folded) v = load_pt29(a.foldB2[1 + a.dc[q]]);
        else if (lane == 3 && folded) v = load_pt29(a.foldB2[1 + a.fold_n + a.ds[q]]);
        v = wave_sum_g2(v, 2);
        if (lane == 0) pre = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) { Acc29G2 t; f29g2_pt_add(t, v, pre); store_xyzz(a.out_xyzz + 512 * (size_t)q + 128, t); }
}
int finalize_tree_g2_launch(zkc_ctx* ctx, hipStream_t st, const FinalizeArgs& a, int nproofs) {
    hipLaunchKernelGGL(zkc_blind_tree_g2, dim3(nproofs), dim3(128), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return zkc_fail(ctx, ZKC_ERR_HIP, std::string("zkc_blind_tree_g2: ") + hipGetErrorString(e));
    return ZKC_OK;
}

size_t finalize_scratch_bytes(int nproofs) { return (size_t)nproofs * (7 * sizeof(G1XYZZ) + sizeof(G2XYZZ) + 2 * 16 * sizeof(Acc29)); }

int finalize_launch(zkc_ctx* ctx, hipStream_t st, const FinalizeArgs& a, int nproofs) {
    static const bool v1 = getenv("ZKC_FINALIZE_WAVES") != nullptr;          // the one-wave-per-task kernel (variable-base products on one lane each)
    if (a.per == 5) {           // the pass carries the two blinding sums as MSM jobs and piB has been written on the G2 stream (zkc_prove.hip)
        hipLaunchKernelGGL(zkc_blind_tree_g1, dim3(10, nproofs), dim3(64), 0, st, a);
        hipLaunchKernelGGL(zkc_blind_tree_g1_out, dim3(nproofs), dim3(128), 0, st, a);
    }
    else if (v1 || !a.scratch || nproofs <= 2) hipLaunchKernelGGL(zkc_finalize, dim3(nproofs), dim3(256), 0, st, a);
    else {
        hipLaunchKernelGGL(zkc_finalize_products, dim3((nproofs + 63) / 64, 8), dim3(64), 0, st, a, nproofs);
        hipLaunchKernelGGL(zkc_finalize_combine, dim3((nproofs + 63) / 64, 3), dim3(64), 0, st, a, nproofs);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return zkc_fail(ctx, ZKC_ERR_HIP, std::string("zkc_finalize: ") + hipGetErrorString(e));
    return ZKC_OK;
}

}  // namespace zkc
