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
};

ZKC_HD void f29g2_mul(F2x29& r, const F2x29& a, const F2x29& b) {
    typedef FqParams P;
    uint32_t t0[9], t1[9], t2[9], sa[9], sb[9];
    f29_mul<P>(t0, a.c0, b.c0); f29_mul<P>(t1, a.c1, b.c1);
    f29_add(sa, a.c0, a.c1); f29_add(sb, b.c0, b.c1);
    f29_mul<P>(t2, sa, sb);
    f29_sub(r.c0, t0, t1, Dom29G2::D24); f29_carry(r.c0);
#pragma unroll
    for (int k = 0; k < 9; k++) r.c1[k] = t2[k] + Dom29G2::D24x2.l[k] - t0[k] - t1[k];
    f29_carry(r.c1);
}
// a carried, each component below 14 p
ZKC_HD void f29g2_sqr(F2x29& r, const F2x29& a) {
    typedef FqParams P;
    uint32_t s[9], d[9], a2[9];
    f29_add(s, a.c0, a.c1); f29_sub(d, a.c0, a.c1, Dom29G2::D26);
#pragma unroll
    for (int k = 0; k < 9; k++) a2[k] = a.c0[k] << 1;
    f29_mul<P>(r.c0, s, d); f29_mul<P>(r.c1, a2, a.c1);
}
ZKC_HD bool f29g2_is_zero(const F2x29& a) { return f29_is_zero_mod_p<FqParams>(a.c0) && f29_is_zero_mod_p<FqParams>(a.c1); }

// acc += (x2, y2); x2, y2 tame and below 1.2 p.  Returns false and leaves acc alone when the x coordinates agree.
// Magnitudes: U2, S2 < 7.6 p ; P = U2 - X + D24, R = S2 - Y + D24 < 13.9 p ; PP, RR < 7 p ; PPP = P PP, Q = X PP < 10 p ;
// X3 = RR - PPP - 2Q + D27x3 < 51 p -> < 3 p ; W = Q - X3 + D24 < 16.3 p ; T = R W < 12.7 p, V = Y PPP < 10 p ;
// Y3 = T - V + D25 < 24.3 p -> < 3 p ; ZZ PP, ZZZ PPP < 10 p.
ZKC_HD bool f29g2_madd(Acc29G2& acc, const F2x29& x2, const F2x29& y2, bool& same_y) {
    F2x29 U, Pn, Rn;
    f29g2_mul(U, x2, acc.ZZ);
    f29_sub(Pn.c0, U.c0, acc.X.c0, Dom29G2::D24); f29_carry(Pn.c0); f29_sub(Pn.c1, U.c1, acc.X.c1, Dom29G2::D24); f29_carry(Pn.c1);
    f29g2_mul(U, y2, acc.ZZZ);
    f29_sub(Rn.c0, U.c0, acc.Y.c0, Dom29G2::D24); f29_carry(Rn.c0); f29_sub(Rn.c1, U.c1, acc.Y.c1, Dom29G2::D24); f29_carry(Rn.c1);
    if (f29g2_is_zero(Pn)) { same_y = f29g2_is_zero(Rn); return false; }
    F2x29 PP, PPP, Q, W;
    f29g2_sqr(PP, Pn); f29g2_mul(PPP, Pn, PP); f29g2_mul(Q, acc.X, PP);
    f29g2_sqr(U, Rn);                                                   // RR
#pragma unroll
    for (int k = 0; k < 9; k++) {
        acc.X.c0[k] = U.c0[k] + Dom29G2::D27x3.l[k] - PPP.c0[k] - 2 * Q.c0[k];
        acc.X.c1[k] = U.c1[k] + Dom29G2::D27x3.l[k] - PPP.c1[k] - 2 * Q.c1[k];
    }
    f29_carry(acc.X.c0); f29_reduce_small<FqParams>(acc.X.c0); f29_carry(acc.X.c1); f29_reduce_small<FqParams>(acc.X.c1);
    f29_sub(W.c0, Q.c0, acc.X.c0, Dom29G2::D24); f29_carry(W.c0); f29_sub(W.c1, Q.c1, acc.X.c1, Dom29G2::D24); f29_carry(W.c1);
    f29g2_mul(U, Rn, W);                                                // T
    f29g2_mul(Q, acc.Y, PPP);                                           // V
    f29_sub(acc.Y.c0, U.c0, Q.c0, Dom29G2::D25); f29_carry(acc.Y.c0); f29_reduce_small<FqParams>(acc.Y.c0);
    f29_sub(acc.Y.c1, U.c1, Q.c1, Dom29G2::D25); f29_carry(acc.Y.c1); f29_reduce_small<FqParams>(acc.Y.c1);
    f29g2_mul(U, acc.ZZ, PP); acc.ZZ = U;
    f29g2_mul(U, acc.ZZZ, PPP); acc.ZZZ = U;
    return true;
}

}  // namespace zkc
