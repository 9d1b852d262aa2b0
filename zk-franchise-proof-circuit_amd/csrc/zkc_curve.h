// zkc_curve.h -- BN254 G1 / G2 group arithmetic (product code, host + device).
//
// Points are accumulated in extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ): the mixed addition
// affine + XYZZ costs 8M + 2S with no inversion, which is what the MSM bucket accumulation (K5/K8) spends its
// time in.  Infinity is ZZ == 0.  Affine points are (x, y) with the all-zero pair standing for infinity, exactly
// the .zkey convention.  G1 is over Fq, G2 over Fq2 = Fq[u]/(u^2+1), both y^2 = x^3 + b (a = 0).
#pragma once
#include "zkc_field.h"

namespace zkc {

template <class B>
struct Fq2T {                       // B[u]/(u^2 + 1)
    B c0, c1;
    ZKC_HD static Fq2T zero() { return {B::zero(), B::zero()}; }
    ZKC_HD static Fq2T one() { return {B::one(), B::zero()}; }
    ZKC_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    ZKC_HD bool operator==(const Fq2T& b) const { return c0 == b.c0 && c1 == b.c1; }
    ZKC_HD bool operator!=(const Fq2T& b) const { return !(*this == b); }
};
template <class B> ZKC_HD Fq2T<B> operator+(const Fq2T<B>& a, const Fq2T<B>& b) { return {a.c0 + b.c0, a.c1 + b.c1}; }
template <class B> ZKC_HD Fq2T<B> operator-(const Fq2T<B>& a, const Fq2T<B>& b) { return {a.c0 - b.c0, a.c1 - b.c1}; }
template <class B> ZKC_HD Fq2T<B> operator*(const Fq2T<B>& a, const Fq2T<B>& b) {      // Karatsuba: 3 base-field multiplications
    B t0 = a.c0 * b.c0, t1 = a.c1 * b.c1;
    B t2 = (a.c0 + a.c1) * (b.c0 + b.c1);
    return {t0 - t1, t2 - t0 - t1};
}
template <class B> ZKC_HD Fq2T<B> fp_sqr(const Fq2T<B>& a) {                        // (c0+c1)(c0-c1), 2 c0 c1
    B t = a.c0 * a.c1;
    return {(a.c0 + a.c1) * (a.c0 - a.c1), t + t};
}
template <class B> ZKC_HD Fq2T<B> fp_dbl(const Fq2T<B>& a) { return {a.c0 + a.c0, a.c1 + a.c1}; }
template <class B> ZKC_HD Fq2T<B> fp_neg(const Fq2T<B>& a) { return {fp_neg(a.c0), fp_neg(a.c1)}; }
template <class B> ZKC_HD Fq2T<B> fp_inv(const Fq2T<B>& a) {
    B n = fp_inv(a.c0 * a.c0 + a.c1 * a.c1);
    return {a.c0 * n, fp_neg(a.c1 * n)};
}
template <class B> ZKC_HD Fq2T<B> fp_inv_gcd(const Fq2T<B>& a) {                      // binary-Euclid inversion of the norm (zkc_field.h): single-lane tails only
    B n = fp_inv_gcd(a.c0 * a.c0 + a.c1 * a.c1);
    return {a.c0 * n, fp_neg(a.c1 * n)};
}
using Fq2 = Fq2T<Fq>;

template <class F>
struct Affine {
    F x, y;
    ZKC_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    ZKC_HD static Affine inf() { return {F::zero(), F::zero()}; }
};
template <class F>
struct XYZZ {
    F X, Y, ZZ, ZZZ;
    ZKC_HD bool is_inf() const { return ZZ.is_zero(); }
    ZKC_HD static XYZZ inf() { return {F::zero(), F::zero(), F::zero(), F::zero()}; }
    ZKC_HD static XYZZ from_affine(const Affine<F>& a) { return a.is_inf() ? inf() : XYZZ{a.x, a.y, F::one(), F::one()}; }
};

// doubling of an affine point (mdbl-2008-s-1)
template <class F>
ZKC_HD XYZZ<F> xyzz_dbl_affine(const Affine<F>& a) {
    if (a.is_inf()) return XYZZ<F>::inf();
    F U = fp_dbl(a.y), V = fp_sqr(U), W = U * V, S = a.x * V;
    F X2 = fp_sqr(a.x), M = fp_dbl(X2) + X2;
    F X3 = fp_sqr(M) - fp_dbl(S);
    F Y3 = M * (S - X3) - W * a.y;
    return {X3, Y3, V, W};
}
// doubling (dbl-2008-s-1)
template <class F>
ZKC_HD XYZZ<F> xyzz_dbl(const XYZZ<F>& p) {
    if (p.is_inf()) return p;
    F U = fp_dbl(p.Y), V = fp_sqr(U), W = U * V, S = p.X * V;
    F X2 = fp_sqr(p.X), M = fp_dbl(X2) + X2;
    F X3 = fp_sqr(M) - fp_dbl(S);
    F Y3 = M * (S - X3) - W * p.Y;
    return {X3, Y3, V * p.ZZ, W * p.ZZZ};
}
// mixed addition p + a (madd-2008-s), complete: handles infinity, p == a and p == -a
template <class F>
ZKC_HD XYZZ<F> xyzz_add_affine(const XYZZ<F>& p, const Affine<F>& a) {
    if (a.is_inf()) return p;
    if (p.is_inf()) return {a.x, a.y, F::one(), F::one()};
    F U2 = a.x * p.ZZ, S2 = a.y * p.ZZZ;
    F P = U2 - p.X, Rr = S2 - p.Y;
    if (P.is_zero()) return Rr.is_zero() ? xyzz_dbl_affine(a) : XYZZ<F>::inf();
    F PP = fp_sqr(P), PPP = P * PP, Q = p.X * PP;
    F X3 = fp_sqr(Rr) - PPP - fp_dbl(Q);
    F Y3 = Rr * (Q - X3) - p.Y * PPP;
    return {X3, Y3, p.ZZ * PP, p.ZZZ * PPP};
}
// general addition (add-2008-s), complete
template <class F>
ZKC_HD XYZZ<F> xyzz_add(const XYZZ<F>& p, const XYZZ<F>& q) {
    if (q.is_inf()) return p;
    if (p.is_inf()) return q;
    F U1 = p.X * q.ZZ, U2 = q.X * p.ZZ, S1 = p.Y * q.ZZZ, S2 = q.Y * p.ZZZ;
    F P = U2 - U1, Rr = S2 - S1;
    if (P.is_zero()) return Rr.is_zero() ? xyzz_dbl(p) : XYZZ<F>::inf();
    F PP = fp_sqr(P), PPP = P * PP, Q = U1 * PP;
    F X3 = fp_sqr(Rr) - PPP - fp_dbl(Q);
    F Y3 = Rr * (Q - X3) - S1 * PPP;
    return {X3, Y3, p.ZZ * q.ZZ * PP, p.ZZZ * q.ZZZ * PPP};
}
template <class F>
ZKC_HD Affine<F> affine_neg(const Affine<F>& a) { return {a.x, fp_neg(a.y)}; }
template <class F>
ZKC_HD XYZZ<F> xyzz_neg(const XYZZ<F>& p) { return {p.X, fp_neg(p.Y), p.ZZ, p.ZZZ}; }
template <class F>
ZKC_HD Affine<F> xyzz_to_affine(const XYZZ<F>& p) {
    if (p.is_inf()) return Affine<F>::inf();
    F zi3 = fp_inv(p.ZZZ);              // 1/ZZZ ; 1/ZZ = ZZZ^-1 * ZZZ / ZZ ... use zi2 = (zi3 * ZZ)^2 since ZZ^3 = ZZZ^2
    F zi = zi3 * p.ZZ;                  // Z^-1  (ZZ = Z^2, ZZZ = Z^3)
    F zi2 = fp_sqr(zi);
    return {p.X * zi2, p.Y * zi3};
}
template <class F>
ZKC_HD Affine<F> xyzz_to_affine_gcd(const XYZZ<F>& p) {       // xyzz_to_affine with fp_inv_gcd: same element, a quarter of the instructions, data-dependent loops
    if (p.is_inf()) return Affine<F>::inf();
    F zi3 = fp_inv_gcd(p.ZZZ);
    F zi = zi3 * p.ZZ;
    F zi2 = fp_sqr(zi);
    return {p.X * zi2, p.Y * zi3};
}
// k * p by double-and-add over a 256-bit standard-form scalar (8 x u32 LE); host-side finalize and setup use this
template <class F>
ZKC_HD XYZZ<F> xyzz_mul(const XYZZ<F>& p, const uint32_t k[8]) {
    XYZZ<F> r = XYZZ<F>::inf();
    for (int i = 255; i >= 0; i--) {
        r = xyzz_dbl(r);
        if ((k[i >> 5] >> (i & 31)) & 1) r = xyzz_add(r, p);
    }
    return r;
}

using G1Affine = Affine<Fq>;
using G2Affine = Affine<Fq2>;
using G1XYZZ = XYZZ<Fq>;
using G2XYZZ = XYZZ<Fq2>;

}  // namespace zkc
