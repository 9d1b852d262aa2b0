// zkc_f29_g1.h -- G1 group operations on radix-2^29 coordinates (zkc_f29.h): the mixed addition of the bucket accumulation (K5) and the
// full addition / doubling of the bucket reduction (K6).  Magnitudes are multiples of p; products contract by 169 = 2^261 / p.
// Dominators: D24 < 6.3 p (dominates carried values below 5.29 p), D25 < 11.6 p (below 10.58 p), D26 < 22.2 p (21.1 p), D27 < 43.3 p (42.3 p);
// "x2"/"x3" variants have limbs 0..7 above 2 or 3 times 2^29 (dominate a sum of two / three carried values).
#pragma once
#include "zkc_f29.h"
#include "zkc_curve.h"

namespace zkc {

struct Acc29 { uint32_t X[9], Y[9], ZZ[9], ZZZ[9]; };
struct Dom29 {
    static constexpr L9 D25 = f29_dominator<FqParams>(1u << 29, 1u << 25);
    static constexpr L9 D24 = f29_dominator<FqParams>(1u << 29, 1u << 24);
    static constexpr L9 D24x3 = f29_dominator<FqParams>(3u << 29, 1u << 24);
    static constexpr L9 D26x2 = f29_dominator<FqParams>(2u << 29, 1u << 26);
    static constexpr L9 D27 = f29_dominator<FqParams>(1u << 29, 1u << 27);
};
// an 8 x u32 element (R = 2^256 form) -> R' form below 1.2 p
ZKC_HD void f29_enter_fq(uint32_t r[9], const uint32_t w[8]) {
    uint32_t t[9]; f29_from_fp_shl5(t, w); f29_mul<FqParams>(r, t, F29K<FqParams>::one.l);
}

// ---- mixed addition acc += (x2, y2); x2, y2 = 32 x table coordinate (< 32 p).  Accumulator invariant: X, Y < 10.5 p carried, ZZ, ZZZ < 4 p.
//   U2 = x2 ZZ, S2 = y2 ZZZ < 1.8 p ; P = U2 - X + D25, R = S2 - Y + D25 < 13.4 p ; PP, RR < 2.1 p ; PPP, Q < 1.2 p
//   X3 = RR - PPP - 2Q + D24x3 < 8.4 p ; W = Q - X3 + D25 < 12.7 p ; Y3 = (R W + (D25 - Y) PPP) / 2^261 + p < 2.2 p ; ZZ PP, ZZZ PPP < 1.1 p
// Returns false and leaves acc alone when the two points share their x coordinate (same_y tells which case).
ZKC_HD bool f29_madd(Acc29& acc, const uint32_t x2[9], const uint32_t y2[9], bool& same_y) {
    typedef FqParams P;
    // every subtraction rides on a reduction: P = x2 ZZ / 2^261 + (D25 - X), likewise R, X3 = R^2 / 2^261 + (D24x3 - PPP - 2Q), and
    // Y3 = (R W + (D25 - Y) PPP) / 2^261 with a single reduction for both products
    uint32_t Pn[9], Rn[9], nX[9], nY[9];
#pragma unroll
    for (int k = 0; k < 9; k++) { nX[k] = Dom29::D25.l[k] - acc.X[k]; nY[k] = Dom29::D25.l[k] - acc.Y[k]; }       // limbs <= 2^30, >= 0
    f29_mul_addhi<P>(Pn, x2, acc.ZZ, nX); f29_mul_addhi<P>(Rn, y2, acc.ZZZ, nY);
    if (f29_is_zero_mod_p<P>(Pn)) { same_y = f29_is_zero_mod_p<P>(Rn); return false; }
    uint32_t PP[9], PPP[9], Q[9], W[9], T[9], V[9];
    f29_sqr<P>(PP, Pn); f29_mul<P>(PPP, Pn, PP); f29_mul<P>(Q, acc.X, PP);
#pragma unroll
    for (int k = 0; k < 9; k++) T[k] = Dom29::D24x3.l[k] - PPP[k] - 2 * Q[k];
    f29_sqr_addhi<P>(acc.X, Rn, T);
    f29_sub(W, Q, acc.X, Dom29::D25);                              // limbs < 1.5 * 2^30: 13.5 + 9 + 2.25 (reduction) < 32 in units of 2^59 per column
    f29_mul2sum<P>(acc.Y, Rn, W, nY, PPP);
    f29_mul<P>(T, acc.ZZ, PP); f29_mul<P>(V, acc.ZZZ, PPP);
#pragma unroll
    for (int k = 0; k < 9; k++) { acc.ZZ[k] = T[k]; acc.ZZZ[k] = V[k]; }
    return true;
}

// ---- bucket reduction: points with all four coordinates carried and below 32 p ("loose": what slicing an 8 x u32 point gives, and what
// the operations below return).  Infinity is ZZ with all limbs zero (a finite point never has ZZ = 0 mod p, and infinity is only ever
// stored as exact zeros).
ZKC_HD bool f29_pt_is_inf(const Acc29& a) { uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) o |= a.ZZ[k]; return o == 0; }
ZKC_HD void f29_pt_set_inf(Acc29& a) {
#pragma unroll
    for (int k = 0; k < 9; k++) a.X[k] = a.Y[k] = a.ZZ[k] = a.ZZZ[k] = 0; }
ZKC_HD Acc29 f29_pt_from_xyzz(const XYZZ<Fq>& p) {       // canonical 8 x u32 -> loose
    Acc29 a; f29_from_fp_shl5(a.X, p.X.v); f29_from_fp_shl5(a.Y, p.Y.v); f29_from_fp_shl5(a.ZZ, p.ZZ.v); f29_from_fp_shl5(a.ZZZ, p.ZZZ.v); return a; }
ZKC_HD XYZZ<Fq> f29_pt_to_xyzz(const Acc29& a) {
    XYZZ<Fq> p; p.X = f29_to_fp<FqParams>(a.X); p.Y = f29_to_fp<FqParams>(a.Y); p.ZZ = f29_to_fp<FqParams>(a.ZZ); p.ZZZ = f29_to_fp<FqParams>(a.ZZZ); return p; }

// doubling (dbl-2008-s-1), a finite.  U = 2Y < 64 p ; V = U^2 < 25.3 p ; W = U V < 10.6 p ; S = X V < 5.8 p ; M = 3 X^2 < 21.2 p ;
// X3 = M^2 - 2S + D26x2 < 25.9 p ; Y3 = M (S - X3 + D27) - W Y + D25 < 18.7 p ; ZZ3 = V ZZ < 5.8 p ; ZZZ3 = W ZZZ < 3.1 p
ZKC_HD void f29_pt_dbl(Acc29& r, const Acc29& a) {
    typedef FqParams P;
    uint32_t U[9], V[9], W[9], S[9], M[9], T[9], X3[9], Y3[9];
    f29_add(U, a.Y, a.Y);
    f29_mul<P>(V, U, U); f29_mul<P>(W, U, V); f29_mul<P>(S, a.X, V);
    f29_sqr<P>(T, a.X);
#pragma unroll
    for (int k = 0; k < 9; k++) M[k] = 3 * T[k];
    f29_carry(M);
    f29_sqr<P>(T, M);
#pragma unroll
    for (int k = 0; k < 9; k++) X3[k] = T[k] + Dom29::D26x2.l[k] - 2 * S[k];
    f29_carry(X3);
    f29_sub(T, S, X3, Dom29::D27);                        // limbs < 1.5 * 2^30, times carried M
    f29_mul<P>(Y3, M, T); f29_mul<P>(T, W, a.Y);
    f29_sub(Y3, Y3, T, Dom29::D25); f29_carry(Y3);
    f29_mul<P>(T, V, a.ZZ); f29_mul<P>(S, W, a.ZZZ);
#pragma unroll
    for (int k = 0; k < 9; k++) { r.X[k] = X3[k]; r.Y[k] = Y3[k]; r.ZZ[k] = T[k]; r.ZZZ[k] = S[k]; }
}
// full addition (add-2008-s), complete.  U1, U2, S1, S2 < 7.1 p ; P, R (+ D25) < 18.7 p ; PP, RR < 3.1 p ; PPP < 1.4 p ; Q < 1.2 p ;
// X3 = RR - PPP - 2Q + D24x3 < 9.4 p ; W = Q - X3 + D25 < 12.8 p ; Y3 = R W - S1 PPP + D24 < 8.7 p ; ZZ3, ZZZ3 < 1.2 p
ZKC_HD void f29_pt_add(Acc29& r, const Acc29& a, const Acc29& b) {
    typedef FqParams P;
    if (f29_pt_is_inf(b)) { r = a; return; }
    if (f29_pt_is_inf(a)) { r = b; return; }
    // as in f29_madd, the subtractions ride on reductions and Y3 needs one reduction for its two products
    uint32_t U1[9], S1[9], Pn[9], Rn[9], T[9], nS1[9];
    f29_mul<P>(U1, a.X, b.ZZ);
#pragma unroll
    for (int k = 0; k < 9; k++) T[k] = Dom29::D25.l[k] - U1[k];
    f29_mul_addhi<P>(Pn, b.X, a.ZZ, T);
    f29_mul<P>(S1, a.Y, b.ZZZ);
#pragma unroll
    for (int k = 0; k < 9; k++) nS1[k] = Dom29::D25.l[k] - S1[k];
    f29_mul_addhi<P>(Rn, b.Y, a.ZZZ, nS1);
    if (f29_is_zero_mod_p<P>(Pn)) {
        if (f29_is_zero_mod_p<P>(Rn)) f29_pt_dbl(r, a); else f29_pt_set_inf(r);
        return;
    }
    uint32_t PP[9], PPP[9], Q[9], X3[9], Y3[9], V[9];
    f29_sqr<P>(PP, Pn); f29_mul<P>(PPP, Pn, PP); f29_mul<P>(Q, U1, PP);
#pragma unroll
    for (int k = 0; k < 9; k++) T[k] = Dom29::D24x3.l[k] - PPP[k] - 2 * Q[k];
    f29_sqr_addhi<P>(X3, Rn, T);
    f29_sub(T, Q, X3, Dom29::D25);
    f29_mul2sum<P>(Y3, Rn, T, nS1, PPP);
    f29_mul<P>(T, a.ZZ, b.ZZ); f29_mul<P>(V, T, PP);
    f29_mul<P>(T, a.ZZZ, b.ZZZ); f29_mul<P>(Q, T, PPP);
#pragma unroll
    for (int k = 0; k < 9; k++) { r.X[k] = X3[k]; r.Y[k] = Y3[k]; r.ZZ[k] = V[k]; r.ZZZ[k] = Q[k]; }
}

// sum over the lanes of a wave (every lane ends up with a representative of the sum)
__device__ inline Acc29 wave_sum_g1(Acc29 p, int top = 32) {
    for (int m = top; m >= 1; m >>= 1) {
        Acc29 o;
#pragma unroll
        for (int k = 0; k < 9; k++) { o.X[k] = (uint32_t)__shfl_xor((int)p.X[k], m, 64); o.Y[k] = (uint32_t)__shfl_xor((int)p.Y[k], m, 64);
                                      o.ZZ[k] = (uint32_t)__shfl_xor((int)p.ZZ[k], m, 64); o.ZZZ[k] = (uint32_t)__shfl_xor((int)p.ZZZ[k], m, 64); }
        Acc29 r; f29_pt_add(r, p, o); p = r;
    }
    return p;
}

}  // namespace zkc
