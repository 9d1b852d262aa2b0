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

size_t finalize_scratch_bytes(int nproofs) { return (size_t)nproofs * (7 * sizeof(G1XYZZ) + sizeof(G2XYZZ) + 2 * 16 * sizeof(Acc29)); }

int finalize_launch(zkc_ctx* ctx, hipStream_t st, const FinalizeArgs& a, int nproofs) {
    static const bool v1 = getenv("ZKC_FINALIZE_WAVES") != nullptr;          // the one-wave-per-task kernel (lowest latency for a single proof's own three chains)
    if (v1 || !a.scratch || nproofs <= 2) hipLaunchKernelGGL(zkc_finalize, dim3(nproofs), dim3(256), 0, st, a);
    else {
        hipLaunchKernelGGL(zkc_finalize_products, dim3((nproofs + 63) / 64, 8), dim3(64), 0, st, a, nproofs);
        hipLaunchKernelGGL(zkc_finalize_combine, dim3((nproofs + 63) / 64, 3), dim3(64), 0, st, a, nproofs);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return zkc_fail(ctx, ZKC_ERR_HIP, std::string("zkc_finalize: ") + hipGetErrorString(e));
    return ZKC_OK;
}

}  // namespace zkc
