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

}  // namespace zkc
