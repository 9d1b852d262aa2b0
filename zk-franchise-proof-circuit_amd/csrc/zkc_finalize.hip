// zkc_finalize.hip -- K9: the constant-size blinding step of Groth16 on the device (stage a7), one wave per proof.
//
//   piA = alpha + A + r delta ;  piB = beta2 + B2 + s delta2 ;  piC = C + H + s piA + r piB1 - r s delta
// (snarkjs groth16_prove.js tail / rapidsnark; reached from ts_inputs/src/example.ts:358-362, zk_census_test.go:89).
// Expanded so that nothing depends on piA / piB1:  s piA + r piB1 - rs delta = s A' + s alpha + r B1' + r beta1 + rs delta,
// i.e. two variable-base products (lanes 0, 1) and six fixed-base ones read from 8-bit window tables (lanes 2..7).
// Runs on the context's second stream so that it overlaps the next pipeline pass.
#include "zkc_prover.h"

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

extern "C" __global__ void __launch_bounds__(64)
zkc_finalize(FinalizeArgs a) {
    __shared__ G1XYZZ sh[8];
    __shared__ G2XYZZ sh2;
    const int q = blockIdx.x, lane = threadIdx.x;
    uint32_t r[8], s[8];
    { const uint4* p = reinterpret_cast<const uint4*>(a.rs + 64 * (size_t)q); uint4 x = p[0], y = p[1], z = p[2], w = p[3];
      r[0] = x.x; r[1] = x.y; r[2] = x.z; r[3] = x.w; r[4] = y.x; r[5] = y.y; r[6] = y.z; r[7] = y.w;
      s[0] = z.x; s[1] = z.y; s[2] = z.z; s[3] = z.w; s[4] = w.x; s[5] = w.y; s[6] = w.z; s[7] = w.w; }
    const int nq = gridDim.x;                         // results of a pass: H_0 .. H_{nq-1}, then A_q, B1_q, C_q per proof
    const G1XYZZ A = xyzz_add(a.r1[nq + 3 * q + 0], a.kA), B1 = xyzz_add(a.r1[nq + 3 * q + 1], a.kB1);
    if (lane < 2) sh[lane] = xyzz_mul(lane == 0 ? A : B1, lane == 0 ? s : r);            // s A' , r B1'
    else if (lane < 7) {
        uint32_t k[8];
        if (lane == 4) { Fr rs = fp_from_std<FrParams>(r) * fp_from_std<FrParams>(s); fp_to_std<FrParams>(k, rs); }
        else {
#pragma unroll
            for (int i = 0; i < 8; i++) k[i] = (lane == 2 || lane == 6) ? r[i] : s[i];
        }
        const G1Affine* tab = lane <= 4 ? a.tblDelta1 : lane == 5 ? a.tblAlpha1 : a.tblBeta1;   // r d, s d, rs d, s alpha, r beta1
        sh[lane] = fb_mul<Fq>(tab, k);
    } else if (lane == 7) sh2 = fb_mul<Fq2>(a.tblDelta2, s);                                  // s delta2
    __syncthreads();
    uint8_t* out = a.out + 256 * (size_t)q;
    if (lane == 0) {            // piA = A' + alpha + r delta
        G1Affine p = xyzz_to_affine(xyzz_add(xyzz_add_affine(A, a.alpha1), sh[2]));
        store_fq_std(out, p.x); store_fq_std(out + 32, p.y);
    } else if (lane == 1) {     // piC = C' + H + s A' + s alpha + r B1' + r beta1 + rs delta
        G1XYZZ c = xyzz_add(xyzz_add(a.r1[nq + 3 * q + 2], a.kC), a.r1[q]);
        c = xyzz_add(c, sh[0]); c = xyzz_add(c, sh[5]); c = xyzz_add(c, sh[1]); c = xyzz_add(c, sh[6]); c = xyzz_add(c, sh[4]);
        G1Affine p = xyzz_to_affine(c);
        store_fq_std(out + 192, p.x); store_fq_std(out + 224, p.y);
    } else if (lane == 7) {     // piB = B2' + beta2 + s delta2
        G2Affine p = xyzz_to_affine(xyzz_add(xyzz_add_affine(xyzz_add(a.r2[q], a.kB2), a.beta2), sh2));
        store_fq_std(out + 64, p.x.c0); store_fq_std(out + 96, p.x.c1); store_fq_std(out + 128, p.y.c0); store_fq_std(out + 160, p.y.c1);
    }
}

int finalize_launch(zkc_ctx* ctx, hipStream_t st, const FinalizeArgs& a, int nproofs) {
    hipLaunchKernelGGL(zkc_finalize, dim3(nproofs), dim3(64), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return zkc_fail(ctx, ZKC_ERR_HIP, std::string("zkc_finalize: ") + hipGetErrorString(e));
    return ZKC_OK;
}

}  // namespace zkc
