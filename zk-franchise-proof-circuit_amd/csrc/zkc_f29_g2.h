// zkc_f29_g2.h -- G2 mixed addition over Fq2 = Fq[u]/(u^2+1) with both components in radix 2^29 (zkc_f29.h).
//
// Invariant of every Fq2 value between operations ("tame"): limbs 0..7 < 2^29 (carried) and each component below 10 p.
// Products contract by 169 = 2^261 / p, so with D24 < 6.3 p (dominates carried values below 5.29 p) and D24x2 < 6.3 p
// (dominates a sum of two carried values whose sum is below 5.29 p):
//   mul2(a, b), a < A p, b < B p per component, A B < 278:
//       t0 = a0 b0, t1 = a1 b1 < A B / 169 + 1 (< 2.65 p) ; t2 = (a0 + a1)(b0 + b1) < 4 A B / 169 + 1
//       c0 = t0 - t1 + D24 < 9 p ; c1 = t2 - t0 - t1 + D24x2 < 4 A B / 169 + 7.3 p   (tame when A B <= 100)
//   sqr2(a), a < 14 p: c0 = (a0 + a1)(a0 - a1 + D26) < 7 p, c1 = (2 a0) a1 < 3.4 p: direct product outputs.
// The accumulator's X and Y are brought below 3 p after every addition (f29_reduce_small) so that the dominators of the
// next addition stay small; the table holds x, y and -y below 1.2 p in R' form (zkc_g2_table29), 9 limbs each.
#pragma once
#include "zkc_f29.h"
#include "zkc_curve.h"
#include "zkc_f29_g1.h"

namespace zkc {

struct F2x29 { uint32_t c0[9], c1[9]; };
struct Acc29G2 { F2x29 X, Y, ZZ, ZZZ; };
struct Dom29G2 {
    typedef FqParams P;
    static constexpr L9 D24 = f29_dominator<P>(1u << 29, 1u << 24);
    static constexpr L9 D24x2 = f29_dominator<P>(2u << 29, 1u << 24);
    static constexpr L9 D25 = f29_dominator<P>(1u << 29, 1u << 25);
    static constexpr L9 D26 = f29_dominator<P>(1u << 29, 1u << 26);
    static constexpr L9 D27x3 = f29_dominator<P>(3u << 29, 1u << 27);
    static constexpr L9 D24x3 = f29_dominator<P>(3u << 29, 1u << 24);
};

// (a0 + a1 u)(b0 + b1 u) with u^2 = -1, schoolbook with one reduction per component:
//   c0 = (a0 b0 + (D26 - a1) b1) / 2^261 [+ h0],  c1 = (a0 b1 + a1 b0) / 2^261 [+ h1]
// a carried with components below 21 p, b carried; outputs are direct reduction outputs (carried), below (A + 22.2) B / 169 + 1 [+ h] p.
ZKC_HD void f29g2_mul_addhi(F2x29& r, const F2x29& a, const F2x29& b, const uint32_t h0[9], const uint32_t h1[9]) {
    typedef FqParams P;
    uint32_t na1[9];
#pragma unroll
    for (int k = 0; k < 9; k++) na1[k] = Dom29G2::D26.l[k] - a.c1[k];
    F2x29 o;
    f29_mul2sum_addhi<P>(o.c0, a.c0, b.c0, na1, b.c1, h0);
    f29_mul2sum_addhi<P>(o.c1, a.c0, b.c1, a.c1, b.c0, h1);
    r = o;
}
ZKC_HD void f29g2_mul(F2x29& r, const F2x29& a, const F2x29& b) {
    const uint32_t z[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    f29g2_mul_addhi(r, a, b, z, z);
}
// a carried, each component below 14 p
ZKC_HD void f29g2_sqr_addhi(F2x29& r, const F2x29& a, const uint32_t h0[9], const uint32_t h1[9]) {
    typedef FqParams P;
    uint32_t s[9], d[9], a2[9];
    f29_add(s, a.c0, a.c1); f29_sub(d, a.c0, a.c1, Dom29G2::D26);
#pragma unroll
    for (int k = 0; k < 9; k++) a2[k] = a.c0[k] << 1;
    F2x29 o;
    f29_mul_addhi<P>(o.c0, s, d, h0); f29_mul_addhi<P>(o.c1, a2, a.c1, h1);
    r = o;
}
ZKC_HD void f29g2_sqr(F2x29& r, const F2x29& a) {
    const uint32_t z[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    f29g2_sqr_addhi(r, a, z, z);
}
ZKC_HD void f29g2_neg_hi(F2x29& h, const F2x29& v, const L9& D) {
#pragma unroll
    for (int k = 0; k < 9; k++) { h.c0[k] = D.l[k] - v.c0[k]; h.c1[k] = D.l[k] - v.c1[k]; } }
ZKC_HD bool f29g2_is_zero(const F2x29& a) { return f29_is_zero_mod_p<FqParams>(a.c0) && f29_is_zero_mod_p<FqParams>(a.c1); }

// acc += (x2, y2); x2, y2 tame and below 1.2 p.  Returns false and leaves acc alone when the x coordinates agree.
// Magnitudes: U2, S2 < 7.6 p ; P = U2 - X + D24, R = S2 - Y + D24 < 13.9 p ; PP, RR < 7 p ; PPP = P PP, Q = X PP < 10 p ;
// X3 = RR - PPP - 2Q + D27x3 < 51 p -> < 3 p ; W = Q - X3 + D24 < 16.3 p ; T = R W < 12.7 p, V = Y PPP < 10 p ;
// Y3 = T - V + D25 < 24.3 p -> < 3 p ; ZZ PP, ZZZ PPP < 10 p.
ZKC_HD bool f29g2_madd(Acc29G2& acc, const F2x29& x2, const F2x29& y2, bool& same_y) {
    // with the one-reduction-per-component product every product output is below 4 p (carried): P = x2 ZZ + (D24 - X), R likewise < 11 p ;
    // PP, RR < 3 p ; PPP = P PP, Q = X PP < 2 p ; X3 = RR + (D24x3 - PPP - 2Q) < 10 p -> < 3 p ; W = Q - X3 + D24 < 8.3 p ;
    // Y3 = R W + (D24 - V), V = Y PPP < 2 p: < 9 p -> < 3 p ; ZZ PP, ZZZ PPP < 2 p
    F2x29 Pn, Rn, h;
#pragma unroll
    for (int k = 0; k < 9; k++) { h.c0[k] = Dom29G2::D24.l[k] - acc.X.c0[k]; h.c1[k] = Dom29G2::D24.l[k] - acc.X.c1[k]; }
    f29g2_mul_addhi(Pn, x2, acc.ZZ, h.c0, h.c1);
#pragma unroll
    for (int k = 0; k < 9; k++) { h.c0[k] = Dom29G2::D24.l[k] - acc.Y.c0[k]; h.c1[k] = Dom29G2::D24.l[k] - acc.Y.c1[k]; }
    f29g2_mul_addhi(Rn, y2, acc.ZZZ, h.c0, h.c1);
    if (f29g2_is_zero(Pn)) { same_y = f29g2_is_zero(Rn); return false; }
    F2x29 PP, PPP, Q, W, V;
    f29g2_sqr(PP, Pn); f29g2_mul(PPP, Pn, PP); f29g2_mul(Q, acc.X, PP);
#pragma unroll
    for (int k = 0; k < 9; k++) {
        h.c0[k] = Dom29G2::D24x3.l[k] - PPP.c0[k] - 2 * Q.c0[k];
        h.c1[k] = Dom29G2::D24x3.l[k] - PPP.c1[k] - 2 * Q.c1[k];
    }
    f29g2_sqr_addhi(acc.X, Rn, h.c0, h.c1);
    f29_reduce_small<FqParams>(acc.X.c0); f29_reduce_small<FqParams>(acc.X.c1);
    f29_sub(W.c0, Q.c0, acc.X.c0, Dom29G2::D24); f29_carry(W.c0); f29_sub(W.c1, Q.c1, acc.X.c1, Dom29G2::D24); f29_carry(W.c1);
    f29g2_mul(V, acc.Y, PPP);
#pragma unroll
    for (int k = 0; k < 9; k++) { h.c0[k] = Dom29G2::D24.l[k] - V.c0[k]; h.c1[k] = Dom29G2::D24.l[k] - V.c1[k]; }
    f29g2_mul_addhi(acc.Y, Rn, W, h.c0, h.c1);
    f29_reduce_small<FqParams>(acc.Y.c0); f29_reduce_small<FqParams>(acc.Y.c1);
    f29g2_mul(acc.ZZ, acc.ZZ, PP);
    f29g2_mul(acc.ZZZ, acc.ZZZ, PPP);
    return true;
}

// [r4] the same addition with its ten products in an order that keeps few values alive, and scheduling fences between them so that the compiler does not interleave
// independent products (which is what holds ~330 registers in f29g2_madd: more instruction-level parallelism than ONE wave per SIMD can use, no room for a second wave).
// Alive at the widest point (PPP = P PP): Y, ZZ, ZZZ, R, Q, P, PP, the product's 18 column sums and one negated operand: about 190 registers.  Same operations, same magnitudes.
__device__ __forceinline__ bool f29g2_madd_lean(Acc29G2& acc, const F2x29& x2, const F2x29& y2, bool& same_y) {
#define ZKC_FENCE() __builtin_amdgcn_sched_barrier(0)
    F2x29 Pn, Rn, h;
    f29g2_neg_hi(h, acc.X, Dom29G2::D24);
    f29g2_mul_addhi(Pn, x2, acc.ZZ, h.c0, h.c1); ZKC_FENCE();
    f29g2_neg_hi(h, acc.Y, Dom29G2::D24);
    f29g2_mul_addhi(Rn, y2, acc.ZZZ, h.c0, h.c1); ZKC_FENCE();
    if (f29g2_is_zero(Pn)) { same_y = f29g2_is_zero(Rn); return false; }
    F2x29 PP, PPP, Q, W, V;
    f29g2_sqr(PP, Pn); ZKC_FENCE();
    f29g2_mul(Q, acc.X, PP); ZKC_FENCE();                       // the old X dies here
    f29g2_mul(acc.ZZ, acc.ZZ, PP); ZKC_FENCE();
    f29g2_mul(PPP, Pn, PP); ZKC_FENCE();                        // P and PP die here
    f29g2_mul(acc.ZZZ, acc.ZZZ, PPP); ZKC_FENCE();
    f29g2_mul(V, acc.Y, PPP); ZKC_FENCE();                      // the old Y dies here
#pragma unroll
    for (int k = 0; k < 9; k++) {
        h.c0[k] = Dom29G2::D24x3.l[k] - PPP.c0[k] - 2 * Q.c0[k];
        h.c1[k] = Dom29G2::D24x3.l[k] - PPP.c1[k] - 2 * Q.c1[k];
    }
    f29g2_sqr_addhi(acc.X, Rn, h.c0, h.c1);
    f29_reduce_small<FqParams>(acc.X.c0); f29_reduce_small<FqParams>(acc.X.c1); ZKC_FENCE();
    f29_sub(W.c0, Q.c0, acc.X.c0, Dom29G2::D24); f29_carry(W.c0); f29_sub(W.c1, Q.c1, acc.X.c1, Dom29G2::D24); f29_carry(W.c1);
    f29g2_neg_hi(h, V, Dom29G2::D24);
    f29g2_mul_addhi(acc.Y, Rn, W, h.c0, h.c1);
    f29_reduce_small<FqParams>(acc.Y.c0); f29_reduce_small<FqParams>(acc.Y.c1); ZKC_FENCE();
#undef ZKC_FENCE
    return true;
}

// ---- bucket reduction in G2: points with all coordinates carried and below 10 p per component ("tame": what the operations below return;
// a canonical 8 x u32 point enters by slicing 32 x value and one f29_reduce_small per component).  Infinity is ZZ with all limbs zero.
// Product outputs are below (A + 22.2) B / 169 + 1 [+ addend] for operands below A p, B p, so with tame inputs:
//   full addition: U1, S1 < 3 p ; P, R = ... + (D24 - U1) < 9.3 p ; PP < 4.5 p ; PPP, Q < 2 p ; X3 = RR + (D25x3 - PPP - 2Q) < 17 p -> < 3 p ;
//                  W = Q - X3 + D24 < 8.3 p ; V = S1 PPP < 1.3 p ; Y3 = R W + (D24 - V) < 8.9 p -> < 3 p ; ZZ3, ZZZ3 < 2 p
//   doubling:      U = 2 Y < 6 p ; V = U^2 < 3 p ; W = U V, S = X V < 1.5 p ; M = 3 X^2 < 5.7 p ; X3 = M^2 + (D24x2 - 2 S) < 9.3 p -> < 3 p ;
//                  T = S - X3 + D24 < 7.8 p ; Y3 = M T + (D24 - W Y) < 8.7 p -> < 3 p ; ZZ3 = V ZZ, ZZZ3 = W ZZZ < 2.6 p
struct Dom29G2x { typedef FqParams P;
    static constexpr L9 D25x3 = f29_dominator<P>(3u << 29, 1u << 25);
    static constexpr L9 D24x2b = f29_dominator<P>(2u << 29, 1u << 24);
};
ZKC_HD bool f29g2_pt_is_inf(const Acc29G2& a) { uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) o |= a.ZZ.c0[k] | a.ZZ.c1[k]; return o == 0; }
ZKC_HD void f29g2_pt_set_inf(Acc29G2& a) {
#pragma unroll
    for (int k = 0; k < 9; k++) a.X.c0[k] = a.X.c1[k] = a.Y.c0[k] = a.Y.c1[k] = a.ZZ.c0[k] = a.ZZ.c1[k] = a.ZZZ.c0[k] = a.ZZZ.c1[k] = 0; }
ZKC_HD void f29g2_enter(F2x29& r, const Fq2& a) {
    f29_from_fp_shl5(r.c0, a.c0.v); f29_reduce_small<FqParams>(r.c0); f29_from_fp_shl5(r.c1, a.c1.v); f29_reduce_small<FqParams>(r.c1); }
ZKC_HD Acc29G2 f29g2_pt_from_xyzz(const XYZZ<Fq2>& p) { Acc29G2 a; f29g2_enter(a.X, p.X); f29g2_enter(a.Y, p.Y); f29g2_enter(a.ZZ, p.ZZ); f29g2_enter(a.ZZZ, p.ZZZ); return a; }
ZKC_HD Fq2 f29g2_leave(const F2x29& a) { return {f29_to_fp<FqParams>(a.c0), f29_to_fp<FqParams>(a.c1)}; }
ZKC_HD XYZZ<Fq2> f29g2_pt_to_xyzz(const Acc29G2& a) { return {f29g2_leave(a.X), f29g2_leave(a.Y), f29g2_leave(a.ZZ), f29g2_leave(a.ZZZ)}; }
ZKC_HD void f29g2_reduce(F2x29& a) { f29_reduce_small<FqParams>(a.c0); f29_reduce_small<FqParams>(a.c1); }

ZKC_HD void f29g2_pt_dbl(Acc29G2& r, const Acc29G2& a) {          // a finite
    F2x29 U, V, W, S, M, T, h, X3, Y3;
#pragma unroll
    for (int k = 0; k < 9; k++) { U.c0[k] = a.Y.c0[k] << 1; U.c1[k] = a.Y.c1[k] << 1; }
    f29_carry(U.c0); f29_carry(U.c1);
    f29g2_sqr(V, U); f29g2_mul(W, U, V); f29g2_mul(S, a.X, V);
    f29g2_sqr(T, a.X);
#pragma unroll
    for (int k = 0; k < 9; k++) { M.c0[k] = 3 * T.c0[k]; M.c1[k] = 3 * T.c1[k]; }
    f29_carry(M.c0); f29_carry(M.c1);
#pragma unroll
    for (int k = 0; k < 9; k++) { h.c0[k] = Dom29G2x::D24x2b.l[k] - 2 * S.c0[k]; h.c1[k] = Dom29G2x::D24x2b.l[k] - 2 * S.c1[k]; }
    f29g2_sqr_addhi(X3, M, h.c0, h.c1); f29g2_reduce(X3);
    f29_sub(T.c0, S.c0, X3.c0, Dom29G2::D24); f29_carry(T.c0); f29_sub(T.c1, S.c1, X3.c1, Dom29G2::D24); f29_carry(T.c1);
    f29g2_mul(S, W, a.Y);                                           // W Y
    f29g2_neg_hi(h, S, Dom29G2::D24);
    f29g2_mul_addhi(Y3, M, T, h.c0, h.c1); f29g2_reduce(Y3);
    f29g2_mul(T, V, a.ZZ); f29g2_mul(S, W, a.ZZZ);
    r.X = X3; r.Y = Y3; r.ZZ = T; r.ZZZ = S;
}
ZKC_HD void f29g2_pt_add(Acc29G2& r, const Acc29G2& a, const Acc29G2& b) {
    if (f29g2_pt_is_inf(b)) { r = a; return; }
    if (f29g2_pt_is_inf(a)) { r = b; return; }
    F2x29 U1, S1, Pn, Rn, h, nS1;
    f29g2_mul(U1, a.X, b.ZZ); f29g2_neg_hi(h, U1, Dom29G2::D24);
    f29g2_mul_addhi(Pn, b.X, a.ZZ, h.c0, h.c1);
    f29g2_mul(S1, a.Y, b.ZZZ); f29g2_neg_hi(nS1, S1, Dom29G2::D24);
    f29g2_mul_addhi(Rn, b.Y, a.ZZZ, nS1.c0, nS1.c1);
    if (f29g2_is_zero(Pn)) {
        if (f29g2_is_zero(Rn)) f29g2_pt_dbl(r, a); else f29g2_pt_set_inf(r);
        return;
    }
    F2x29 PP, PPP, Q, X3, Y3, W, V;
    f29g2_sqr(PP, Pn); f29g2_mul(PPP, Pn, PP); f29g2_mul(Q, U1, PP);
#pragma unroll
    for (int k = 0; k < 9; k++) {
        h.c0[k] = Dom29G2x::D25x3.l[k] - PPP.c0[k] - 2 * Q.c0[k];
        h.c1[k] = Dom29G2x::D25x3.l[k] - PPP.c1[k] - 2 * Q.c1[k];
    }
    f29g2_sqr_addhi(X3, Rn, h.c0, h.c1); f29g2_reduce(X3);
    f29_sub(W.c0, Q.c0, X3.c0, Dom29G2::D24); f29_carry(W.c0); f29_sub(W.c1, Q.c1, X3.c1, Dom29G2::D24); f29_carry(W.c1);
    f29g2_mul(V, S1, PPP); f29g2_neg_hi(h, V, Dom29G2::D24);
    f29g2_mul_addhi(Y3, Rn, W, h.c0, h.c1); f29g2_reduce(Y3);
    f29g2_mul(V, a.ZZ, b.ZZ); f29g2_mul(W, V, PP);
    f29g2_mul(V, a.ZZZ, b.ZZZ); f29g2_mul(Q, V, PPP);
    r.X = X3; r.Y = Y3; r.ZZ = W; r.ZZZ = Q;
}

// sum over the lanes of a wave by a butterfly of full additions, `top` = half the group width (32: the whole wave); every lane of a group ends up with a representative of the sum
__device__ inline Acc29G2 wave_sum_g2(Acc29G2 p, int top) {
    for (int m = top; m >= 1; m >>= 1) {
        Acc29G2 o;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            o.X.c0[k] = (uint32_t)__shfl_xor((int)p.X.c0[k], m, 64); o.X.c1[k] = (uint32_t)__shfl_xor((int)p.X.c1[k], m, 64);
            o.Y.c0[k] = (uint32_t)__shfl_xor((int)p.Y.c0[k], m, 64); o.Y.c1[k] = (uint32_t)__shfl_xor((int)p.Y.c1[k], m, 64);
            o.ZZ.c0[k] = (uint32_t)__shfl_xor((int)p.ZZ.c0[k], m, 64); o.ZZ.c1[k] = (uint32_t)__shfl_xor((int)p.ZZ.c1[k], m, 64);
            o.ZZZ.c0[k] = (uint32_t)__shfl_xor((int)p.ZZZ.c0[k], m, 64); o.ZZZ.c1[k] = (uint32_t)__shfl_xor((int)p.ZZZ.c1[k], m, 64);
        }
        Acc29G2 r; f29g2_pt_add(r, p, o); p = r;
    }
    return p;
}

}  // namespace zkc
