// zkc_pairing.h -- the BN254 pairing of the verifiers (a9 / f4).  The tower arithmetic and the line steps are host + device code (the batch verifier's kernels,
// zkc_pairing_dev.hip, run them on the GPU); the loop drivers, the prepared points, the final exponentiation and the membership tests are host code (zkc_verify.hip).
//
// What go-rapidsnark's verifier does behind (*Proof).Verify (zk_census_test.go:122) and snarkjs behind groth16.verify, restated: optimal ate Miller loop over 6x + 2
// with the two Frobenius steps, homogeneous projective line functions (no inversions), lines multiplied in sparsely, several pairs sharing ONE accumulator (one squaring
// per bit whatever the number of pairs), G2 points "prepared" into their line coefficients (a verification key's gamma and delta once per key), and a final
// exponentiation split into the easy part and the Fuentes-Castaneda hard part over three powers of x.  Rounds 1-4 ran the plain ate loop over 6x^2 with affine lines
// (one Fq2 inversion per step), dense Fq12 products and a 1268-bit square-and-multiply final exponentiation: 30 ms per proof on the GPU boxes' hosts.
//
// Tower: Fq2 = Fq[u]/(u^2 + 1), Fq6 = Fq2[v]/(v^3 - xi), Fq12 = Fq6[w]/(w^2 - v), xi = 9 + u; the twist is y^2 = x^3 + 3/xi (D type), a line lives in the
// coefficients of w^0, w^1, w^3.  The value final_exp() returns is the reduced pairing raised to 2x(6x^2 + 3x + 1) -- exactly what libff / ffjavascript compute and
// what verification_key.json carries as vk_alphabeta_12, which pins all of this (tests/test_oracle_pinning.py through zkc_pairing_bin).
#pragma once
#include <array>
#include <cstring>
#include <vector>
#include "zkc_curve.h"

// host + device, inlined at the compiler's discretion (an Fq12 product is 54 field products: forcing it inline everywhere costs minutes of compile time and buys nothing)
#define ZKC_HDI __host__ __device__ inline

namespace zkc { namespace pairing {

ZKC_HDI Fq2 mul_xi(const Fq2& a) {                                    // (9 + u) a
    const Fq t0 = fp_dbl(fp_dbl(fp_dbl(a.c0))) + a.c0, t1 = fp_dbl(fp_dbl(fp_dbl(a.c1))) + a.c1;
    return {t0 - a.c1, t1 + a.c0};
}
ZKC_HDI Fq2 conj2(const Fq2& a) { return {a.c0, fp_neg(a.c1)}; }
ZKC_HDI Fq2 scale2(const Fq2& a, const Fq& s) { return {a.c0 * s, a.c1 * s}; }
inline Fq2 fq2_pow(const Fq2& a, const uint32_t* e, int nbits) { Fq2 r = Fq2::one(); for (int i = nbits - 1; i >= 0; i--) { r = fp_sqr(r); if ((e[i >> 5] >> (i & 31)) & 1) r = r * a; } return r; }

struct Fq6 { Fq2 a0, a1, a2; };
struct Fq12 { Fq6 a, b; };
ZKC_HDI Fq6 operator+(const Fq6& x, const Fq6& y) { return {x.a0 + y.a0, x.a1 + y.a1, x.a2 + y.a2}; }
ZKC_HDI Fq6 operator-(const Fq6& x, const Fq6& y) { return {x.a0 - y.a0, x.a1 - y.a1, x.a2 - y.a2}; }
ZKC_HDI Fq6 neg6(const Fq6& x) { return {fp_neg(x.a0), fp_neg(x.a1), fp_neg(x.a2)}; }
ZKC_HDI Fq6 mul_v(const Fq6& x) { return {mul_xi(x.a2), x.a0, x.a1}; }
ZKC_HDI Fq6 operator*(const Fq6& x, const Fq6& y) {                   // Karatsuba: 6 products in Fq2
    const Fq2 v0 = x.a0 * y.a0, v1 = x.a1 * y.a1, v2 = x.a2 * y.a2;
    return {v0 + mul_xi((x.a1 + x.a2) * (y.a1 + y.a2) - v1 - v2), (x.a0 + x.a1) * (y.a0 + y.a1) - v0 - v1 + mul_xi(v2), (x.a0 + x.a2) * (y.a0 + y.a2) - v0 - v2 + v1};
}
ZKC_HDI Fq6 mul_by_01(const Fq6& x, const Fq2& b0, const Fq2& b1) {  // x (b0 + b1 v): 5 products
    const Fq2 v0 = x.a0 * b0, v1 = x.a1 * b1;
    return {v0 + mul_xi((x.a1 + x.a2) * b1 - v1), (x.a0 + x.a1) * (b0 + b1) - v0 - v1, (x.a0 + x.a2) * b0 - v0 + v1};
}
ZKC_HDI Fq6 mul_by_fq2(const Fq6& x, const Fq2& b) { return {x.a0 * b, x.a1 * b, x.a2 * b}; }
inline Fq6 inv6(const Fq6& x) {
    const Fq2 c0 = fp_sqr(x.a0) - mul_xi(x.a1 * x.a2), c1 = mul_xi(fp_sqr(x.a2)) - x.a0 * x.a1, c2 = fp_sqr(x.a1) - x.a0 * x.a2;
    const Fq2 t = fp_inv_gcd(mul_xi(x.a2 * c1 + x.a1 * c2) + x.a0 * c0);
    return {c0 * t, c1 * t, c2 * t};
}
ZKC_HDI Fq12 one12() { Fq12 r{}; r.a.a0 = Fq2::one(); r.a.a1 = r.a.a2 = r.b.a0 = r.b.a1 = r.b.a2 = Fq2::zero(); return r; }
ZKC_HDI Fq12 operator*(const Fq12& x, const Fq12& y) {                // Karatsuba: 3 products in Fq6
    const Fq6 aa = x.a * y.a, bb = x.b * y.b;
    return {aa + mul_v(bb), (x.a + x.b) * (y.a + y.b) - aa - bb};
}
ZKC_HDI Fq12 sqr12(const Fq12& x) {                                   // complex squaring: 2 products in Fq6
    const Fq6 ab = x.a * x.b;
    return {(x.a + x.b) * (x.a + mul_v(x.b)) - ab - mul_v(ab), ab + ab};
}
// squaring inside the cyclotomic subgroup (Granger-Scott: Fq12 as three Fq4 = Fq2[y]/(y^2 - xi) squarings, 6 products in Fq2 instead of 12); valid for x with
// x^(q^4 - q^2 + 1) = 1, which everything after the easy part of the final exponentiation satisfies
inline Fq12 cyclotomic_sqr(const Fq12& x) {
    const Fq2 &r0 = x.a.a0, &r4 = x.a.a1, &r3 = x.a.a2, &r2 = x.b.a0, &r1 = x.b.a1, &r5 = x.b.a2;
    auto sq4 = [](const Fq2& a, const Fq2& b, Fq2& t0, Fq2& t1) {      // (a + b y)^2 = t0 + t1 y
        const Fq2 ab = a * b;
        t0 = (a + b) * (mul_xi(b) + a) - ab - mul_xi(ab); t1 = fp_dbl(ab);
    };
    Fq2 t0, t1, t2, t3, t4, t5;
    sq4(r0, r1, t0, t1); sq4(r2, r3, t2, t3); sq4(r4, r5, t4, t5);
    auto three_minus_two = [](const Fq2& t, const Fq2& z) { return fp_dbl(t - z) + t; };      // 3t - 2z
    auto three_plus_two = [](const Fq2& t, const Fq2& z) { return fp_dbl(t + z) + t; };       // 3t + 2z
    Fq12 o;
    o.a.a0 = three_minus_two(t0, r0); o.b.a1 = three_plus_two(t1, r1);
    o.b.a0 = three_plus_two(mul_xi(t5), r2); o.a.a2 = three_minus_two(t4, r3);
    o.a.a1 = three_minus_two(t2, r4); o.b.a2 = three_plus_two(t3, r5);
    return o;
}
ZKC_HDI Fq12 conj12(const Fq12& x) { return {x.a, neg6(x.b)}; }      // x^(q^6); the inverse inside the cyclotomic subgroup
inline Fq12 inv12(const Fq12& x) { const Fq6 t = inv6(x.a * x.a - mul_v(x.b * x.b)); return {x.a * t, neg6(x.b * t)}; }
inline bool is_one12(const Fq12& x) { const Fq12 o = one12(); return memcmp(&x, &o, sizeof o) == 0; }
// f (c0 + d0 w + d1 w^3): the product with a line, 13 products in Fq2 instead of 18
ZKC_HDI Fq12 mul_by_034(const Fq12& f, const Fq2& c0, const Fq2& d0, const Fq2& d1) {
    const Fq6 a = mul_by_fq2(f.a, c0), b = mul_by_01(f.b, d0, d1), e = mul_by_01(f.a + f.b, c0 + d0, d1);
    return {a + mul_v(b), e - a - b};
}

// (c0 + d0 w + d1 w^3)(c0' + d0' w + d1' w^3): two lines into one dense element, 9 products in Fq2 (the first level of the batch verifier's product tree on the GPU)
ZKC_HDI Fq12 mul_034_by_034(const Fq2* l, const Fq2* m) {
    const Fq2 d1d1 = l[2] * m[2];
    Fq12 r;
    r.a.a0 = l[0] * m[0] + mul_xi(d1d1); r.a.a1 = l[1] * m[1]; r.a.a2 = l[1] * m[2] + l[2] * m[1];
    r.b.a0 = l[0] * m[1] + l[1] * m[0]; r.b.a1 = l[0] * m[2] + l[2] * m[0]; r.b.a2 = Fq2::zero();
    return r;
}
ZKC_HDI Fq12 dense_of_034(const Fq2* l) { Fq12 r; r.a.a0 = l[0]; r.a.a1 = r.a.a2 = r.b.a2 = Fq2::zero(); r.b.a0 = l[1]; r.b.a1 = l[2]; return r; }
// ---- the point side of the Miller loop: R <- 2R or R + Q on the twist in homogeneous projective coordinates, and the coefficients (c, d0, d1) of the line through
// the points involved (Costello-Lange-Naehrig formulas, as arkworks' bn / gnark lay them out): the line at P is c yP + d0 xP w + d1 w^3, up to factors in Fq2 ----
struct LinePoint { Fq2 X, Y, Z; };
ZKC_HDI void line_dbl(LinePoint& R, const Fq2& twist_b, const Fq& half, Fq2 out[3]) {
    const Fq2 a = scale2(R.X * R.Y, half), b = fp_sqr(R.Y), c = fp_sqr(R.Z), c3 = fp_dbl(c) + c, e = twist_b * c3, f = fp_dbl(e) + e, g = scale2(b + f, half),
              h = fp_sqr(R.Y + R.Z) - (b + c), i = e - b, j = fp_sqr(R.X), e2 = fp_sqr(e);
    R.X = a * (b - f); R.Y = fp_sqr(g) - (fp_dbl(e2) + e2); R.Z = b * h;
    out[0] = fp_neg(h); out[1] = fp_dbl(j) + j; out[2] = i;
}
ZKC_HDI void line_add(LinePoint& R, const Fq2& xq, const Fq2& yq, Fq2 out[3]) {
    const Fq2 theta = R.Y - yq * R.Z, lambda = R.X - xq * R.Z, c = fp_sqr(theta), d = fp_sqr(lambda), e = lambda * d, f = R.Z * c, g = R.X * d, h = e + f - fp_dbl(g);
    const Fq2 j = theta * xq - lambda * yq;
    R.X = lambda * h; R.Y = theta * (g - h) - e * R.Y; R.Z = R.Z * e;
    out[0] = lambda; out[1] = fp_neg(theta); out[2] = j;
}

// ---- constants derived once: xi^((q-1)/6) and its powers (Frobenius on Fq12 and on the twist), the twist's b, 1/2 ----
struct Consts {
    Fq2 g1[6], g3[6]; Fq g2[6];          // gk[i] = xi^(i (q^k - 1) / 6): w^i -> gk[i] w^i under x -> x^(q^k) (coefficients conjugated for odd k); g2 lies in Fq
    Fq2 twist_b; Fq half;
    Fq2 psi_x, psi_y, psi2_x, psi2_y;     // pi(Q) = (conj(x) psi_x, conj(y) psi_y); pi^2(Q) = (x psi2_x, y psi2_y)
};
inline const Consts& consts() {
    static const Consts C = [] {
        Consts c;
        uint32_t e6[8]; { uint32_t qm1[8]; for (int i = 0; i < 8; i++) qm1[i] = FqParams::p[i]; qm1[0] -= 1;
                          uint64_t rem = 0; for (int i = 7; i >= 0; i--) { const uint64_t cur = (rem << 32) | qm1[i]; e6[i] = (uint32_t)(cur / 6); rem = cur % 6; } }
        const Fq2 xi{fp_from_u32<FqParams>(9), Fq::one()};
        const Fq2 g = fq2_pow(xi, e6, 254);
        c.g1[0] = Fq2::one(); for (int i = 1; i < 6; i++) c.g1[i] = c.g1[i - 1] * g;
        for (int i = 0; i < 6; i++) { const Fq2 n = c.g1[i] * conj2(c.g1[i]); c.g2[i] = n.c0; c.g3[i] = scale2(c.g1[i], n.c0); }      // q^2 - 1 = (q - 1)(q + 1), q^3 - 1 = (q - 1)(q^2 + q + 1)
        c.twist_b = scale2(fp_inv_gcd(xi), fp_from_u32<FqParams>(3));
        c.half = fp_inv_gcd(fp_from_u32<FqParams>(2));
        c.psi_x = c.g1[2]; c.psi_y = c.g1[3];                                       // xi^((q-1)/3), xi^((q-1)/2)
        c.psi2_x = {c.g2[2], Fq::zero()}; c.psi2_y = {c.g2[3], Fq::zero()};
        return c;
    }();
    return C;
}
inline Fq12 frobenius(const Fq12& f, int k) {                       // f^(q^k), k = 1, 2, 3
    const Consts& C = consts();
    const Fq2* c[6] = {&f.a.a0, &f.b.a0, &f.a.a1, &f.b.a1, &f.a.a2, &f.b.a2};       // coefficients of w^0 .. w^5
    Fq2 o[6];
    for (int i = 0; i < 6; i++) {
        const Fq2 x = (k & 1) ? conj2(*c[i]) : *c[i];
        o[i] = k == 1 ? x * C.g1[i] : k == 2 ? scale2(x, C.g2[i]) : x * C.g3[i];
    }
    return {{o[0], o[2], o[4]}, {o[1], o[3], o[5]}};
}

// ---- curve and subgroup membership ----
inline bool g1_on_curve(const G1Affine& a) { return a.is_inf() || fp_sqr(a.y) == fp_sqr(a.x) * a.x + fp_from_u32<FqParams>(3); }
inline bool g2_on_curve(const G2Affine& a) {
    if (a.is_inf()) return true;
    return fp_sqr(a.y) == fp_sqr(a.x) * a.x + consts().twist_b;
}
// membership in G2, the order-r subgroup of the twist, by definition: [r]a = infinity.  Kept as the cross-check of the one below (tests/host/pairing_host.hip)
inline bool g2_in_subgroup_by_order(const G2Affine& a) {
    if (!g2_on_curve(a)) return false;
    if (a.is_inf()) return true;
    int top = 255; while (!((FrParams::p[top >> 5] >> (top & 31)) & 1)) top--;
    G2XYZZ acc = G2XYZZ::from_affine(a);
    for (int i = top - 1; i >= 0; i--) { acc = xyzz_dbl(acc); if ((FrParams::p[i >> 5] >> (i & 31)) & 1) acc = xyzz_add_affine(acc, a); }
    return acc.is_inf();
}
// ... at half the work: psi(a) == [6x^2]a, psi the untwist-Frobenius-twist map.  psi satisfies X^2 - tX + q = 0 on the whole twist, so psi(a) = [u]a gives
// [u^2 - tu + q]a = 0, and with u = t - 1 = 6x^2 that scalar is q - 6x^2 = r: a has order r (or is infinity), and the points of order dividing r in E'(Fq2) are exactly
// G2 (r divides the group order once).  Conversely psi acts on G2 as multiplication by q = t - 1 mod r.  A 127-bit scalar instead of a 254-bit one.
inline bool g2_in_subgroup(const G2Affine& a) {
    if (!g2_on_curve(a)) return false;
    if (a.is_inf()) return true;
    static const uint64_t T[2] = {0xf83e9682e87cfd46ull, 0x6f4d8248eeb859fbull};   // 6 x^2, x = 4965661367192848881; bit 126 is the top one
    G2XYZZ acc = G2XYZZ::from_affine(a);
    for (int i = 125; i >= 0; i--) { acc = xyzz_dbl(acc); if ((T[i >> 6] >> (i & 63)) & 1) acc = xyzz_add_affine(acc, a); }
    if (acc.is_inf()) return false;
    const Consts& C = consts();
    return acc.X == conj2(a.x) * C.psi_x * acc.ZZ && acc.Y == conj2(a.y) * C.psi_y * acc.ZZZ;
}

// ---- Miller loop ----
typedef std::array<Fq2, 3> LineCoeffs;                               // (c, d0, d1): the line is c yP + d0 xP w + d1 w^3
struct G2Prepared { std::vector<LineCoeffs> lines; bool inf = true; };
// 6x + 2 = 29793968203157093288 (x = 4965661367192848881, 65 bits) in signed digits: the non-adjacent form, its top "1 0 -1" folded back to "1 1" so that the loop
// keeps 64 doublings -- 21 additions instead of the 36 set bits.  digit[64] = 1 is the starting point R = Q.
struct AteLoop { int8_t digit[65]; };
inline const AteLoop& ate_loop() {
    static const AteLoop L = [] {
        AteLoop l{}; const unsigned __int128 value = ((unsigned __int128)1 << 64) | 0x9d797039be763ba8ull;
        int8_t d[68] = {0}; int len = 0;
        for (unsigned __int128 n = value; n; n >>= 1, len++)
            if (n & 1) { if ((n & 3) == 1) { d[len] = 1; n -= 1; } else { d[len] = -1; n += 1; } }
        if (len == 66 && d[65] == 1 && d[64] == 0 && d[63] == -1) { d[65] = 0; d[64] = 1; d[63] = 1; len = 65; }      // 2^65 - 2^63 = 2^64 + 2^63
        __int128 back = 0; for (int i = len - 1; i >= 0; i--) back = 2 * back + d[i];
        if (len != 65 || d[64] != 1 || back != (__int128)value) { for (int i = 0; i < 65; i++) d[i] = (int8_t)((value >> i) & 1); }   // plain binary (never taken for this constant)
        for (int i = 0; i < 65; i++) l.digit[i] = d[i];
        return l;
    }();
    return L;
}
// Q -> the coefficients of every line of its Miller loop, in loop order
inline G2Prepared prepare_g2(const G2Affine& Q) {
    G2Prepared out; if (Q.is_inf()) return out;
    out.inf = false; out.lines.reserve(96);
    const Consts& C = consts(); const AteLoop& L = ate_loop();
    LinePoint R{Q.x, Q.y, Fq2::one()};
    auto dbl = [&] { LineCoeffs l; line_dbl(R, C.twist_b, C.half, l.data()); out.lines.push_back(l); };
    auto add = [&](const Fq2& xq, const Fq2& yq) { LineCoeffs l; line_add(R, xq, yq, l.data()); out.lines.push_back(l); };
    const Fq2 nQy = fp_neg(Q.y);
    for (int i = 63; i >= 0; i--) { dbl(); if (L.digit[i] > 0) add(Q.x, Q.y); else if (L.digit[i] < 0) add(Q.x, nQy); }
    add(conj2(Q.x) * C.psi_x, conj2(Q.y) * C.psi_y);                               // + pi(Q)
    add(Q.x * C.psi2_x, fp_neg(Q.y * C.psi2_y));                                   // - pi^2(Q)
    return out;
}
struct Pair { G1Affine P; const G2Prepared* Q; };
// prod_k f_{6x+2, Q_k}(P_k) with its Frobenius lines, all pairs on one accumulator.  Pairs with P or Q at infinity contribute 1.
inline Fq12 multi_miller(const Pair* pairs, size_t n) {
    std::vector<const Pair*> live; for (size_t k = 0; k < n; k++) if (!pairs[k].P.is_inf() && !pairs[k].Q->inf) live.push_back(&pairs[k]);
    Fq12 f = one12(); size_t idx = 0;
    auto ell = [&] { for (const Pair* p : live) { const LineCoeffs& l = p->Q->lines[idx]; f = mul_by_034(f, scale2(l[0], p->P.y), scale2(l[1], p->P.x), l[2]); } idx++; };
    if (live.empty()) return f;
    const AteLoop& L = ate_loop();
    for (int i = 63; i >= 0; i--) { if (i != 63) f = sqr12(f); ell(); if (L.digit[i]) ell(); }
    ell(); ell();
    return f;
}
inline Fq12 miller(const G1Affine& P, const G2Affine& Q) { const G2Prepared q = prepare_g2(Q); const Pair p{P, &q}; return multi_miller(&p, 1); }

// ---- final exponentiation: f -> f^((q^12 - 1)/r * 2x(6x^2 + 3x + 1)) ----
inline Fq12 exp_by_neg_x(const Fq12& f) {                            // f^(-x) inside the cyclotomic subgroup
    static const uint64_t X = 4965661367192848881ull;
    Fq12 r = f;
    for (int i = 61; i >= 0; i--) { r = cyclotomic_sqr(r); if ((X >> i) & 1) r = r * f; }   // bit 62 is the top one
    return conj12(r);
}
inline Fq12 final_exp(const Fq12& f) {
    Fq12 r = conj12(f) * inv12(f);                                   // f^(q^6 - 1)
    r = frobenius(r, 2) * r;                                         // ^(q^2 + 1): r is in the cyclotomic subgroup from here on
    // Fuentes-Castaneda, Knapp, Rodriguez-Henriquez: lambda_0 + lambda_1 q + lambda_2 q^2 + lambda_3 q^3 with
    // lambda_0 = 1 + 6x + 12x^2 + 12x^3, lambda_1 = 4x + 6x^2 + 12x^3, lambda_2 = 6x + 6x^2 + 12x^3, lambda_3 = lambda_1 - 1
    const Fq12 y0 = exp_by_neg_x(r), y1 = cyclotomic_sqr(y0), y2 = cyclotomic_sqr(y1), y3n = y2 * y1, y4 = exp_by_neg_x(y3n), y5 = cyclotomic_sqr(y4), y6n = exp_by_neg_x(y5);
    const Fq12 y3 = conj12(y3n), y6 = conj12(y6n);
    const Fq12 y7 = y6 * y4, y8 = y7 * y3, y9 = y8 * y1, y10 = y8 * y4, y11 = y10 * r;
    const Fq12 y13 = frobenius(y9, 1) * y11, y14 = frobenius(y8, 2) * y13, y15 = frobenius(conj12(r) * y9, 3);
    return y15 * y14;
}
}}  // namespace zkc::pairing
