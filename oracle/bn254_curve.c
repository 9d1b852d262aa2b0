/* oracle/bn254_curve.c -- TEST INFRASTRUCTURE ONLY.  BN254 G1/G2, the Fq2/Fq6/Fq12 tower and a plain ate
 * pairing (Miller loop over T = t-1 = 6x^2, affine line functions, final exponentiation by straight
 * square-and-multiply).  Restates what the reference reaches through go-rapidsnark/verifier
 * (zk_census_test.go:122) and snarkjs groth16.verify; any non-degenerate bilinear pairing decides the
 * Groth16 product check identically, so the simplest correct construction is used.  Pinned by the
 * reference's committed (proof.json, signals.json, verification_key.json) triple. */
#include "bn254.h"
#include <string.h>

/* ---------------- Fq helpers ---------------- */
#define Q (&FQ)
static void q_add(fe_t *o, const fe_t *a, const fe_t *b) { fe_add(o, a, b, Q); }
static void q_sub(fe_t *o, const fe_t *a, const fe_t *b) { fe_sub(o, a, b, Q); }
static void q_mul(fe_t *o, const fe_t *a, const fe_t *b) { fe_mul(o, a, b, Q); }
static void q_sqr(fe_t *o, const fe_t *a) { fe_mul(o, a, a, Q); }
static void q_neg(fe_t *o, const fe_t *a) { fe_neg(o, a, Q); }
static void q_inv(fe_t *o, const fe_t *a) { fe_inv(o, a, Q); }
static void q_dbl(fe_t *o, const fe_t *a) { fe_add(o, a, a, Q); }
static void q_one(fe_t *o) { memcpy(o->v, FQ.r1, 32); }
static void q_zero(fe_t *o) { memset(o, 0, sizeof *o); }

/* ---------------- Fq2 = Fq[u]/(u^2+1) ---------------- */
static void f2_add(fq2_t *o, const fq2_t *a, const fq2_t *b) { q_add(&o->c0, &a->c0, &b->c0); q_add(&o->c1, &a->c1, &b->c1); }
static void f2_sub(fq2_t *o, const fq2_t *a, const fq2_t *b) { q_sub(&o->c0, &a->c0, &b->c0); q_sub(&o->c1, &a->c1, &b->c1); }
static void f2_neg(fq2_t *o, const fq2_t *a) { q_neg(&o->c0, &a->c0); q_neg(&o->c1, &a->c1); }
static void f2_dbl(fq2_t *o, const fq2_t *a) { f2_add(o, a, a); }
static void f2_mul(fq2_t *o, const fq2_t *a, const fq2_t *b) {
    fe_t t0, t1, t2, t3;
    q_mul(&t0, &a->c0, &b->c0); q_mul(&t1, &a->c1, &b->c1);
    q_mul(&t2, &a->c0, &b->c1); q_mul(&t3, &a->c1, &b->c0);
    q_sub(&o->c0, &t0, &t1); q_add(&o->c1, &t2, &t3);
}
static void f2_sqr(fq2_t *o, const fq2_t *a) { fq2_t t = *a; f2_mul(o, &t, &t); }
static void f2_inv(fq2_t *o, const fq2_t *a) {
    fe_t n, t; q_sqr(&n, &a->c0); q_sqr(&t, &a->c1); q_add(&n, &n, &t); q_inv(&n, &n);
    q_mul(&o->c0, &a->c0, &n); q_mul(&t, &a->c1, &n); q_neg(&o->c1, &t);
}
static int f2_is_zero(const fq2_t *a) { return fe_is_zero(&a->c0) && fe_is_zero(&a->c1); }
static int f2_eq(const fq2_t *a, const fq2_t *b) { return fe_eq(&a->c0, &b->c0) && fe_eq(&a->c1, &b->c1); }
static void f2_one(fq2_t *o) { q_one(&o->c0); q_zero(&o->c1); }
static void f2_zero(fq2_t *o) { memset(o, 0, sizeof *o); }
static void f2_mul_xi(fq2_t *o, const fq2_t *a) {   /* (9+u)(c0+c1 u) = 9c0 - c1 + (9c1 + c0) u */
    fe_t n0, n1, t; q_dbl(&t, &a->c0); q_dbl(&t, &t); q_dbl(&t, &t); q_add(&n0, &t, &a->c0); q_sub(&n0, &n0, &a->c1);
    q_dbl(&t, &a->c1); q_dbl(&t, &t); q_dbl(&t, &t); q_add(&n1, &t, &a->c1); q_add(&n1, &n1, &a->c0);
    o->c0 = n0; o->c1 = n1;
}

/* ---------------- curves ---------------- */
static int g1_eq_dummy;
#define CF fe_t
#define CA g1a_t
#define CJ g1j_t
#define PFX(n) g1j_##n
#define c_add q_add
#define c_sub q_sub
#define c_mul q_mul
#define c_sqr q_sqr
#define c_neg q_neg
#define c_inv q_inv
#define c_dbl q_dbl
#define c_one q_one
#define c_zero q_zero
#define c_is_zero fe_is_zero
#define c_eq fe_eq
#include "curve_tmpl.h"
#undef CF
#undef CA
#undef CJ
#undef PFX
#undef c_add
#undef c_sub
#undef c_mul
#undef c_sqr
#undef c_neg
#undef c_inv
#undef c_dbl
#undef c_one
#undef c_zero
#undef c_is_zero
#undef c_eq
#define CF fq2_t
#define CA g2a_t
#define CJ g2j_t
#define PFX(n) g2j_##n
#define c_add f2_add
#define c_sub f2_sub
#define c_mul f2_mul
#define c_sqr f2_sqr
#define c_neg f2_neg
#define c_inv f2_inv
#define c_dbl f2_dbl
#define c_one f2_one
#define c_zero f2_zero
#define c_is_zero f2_is_zero
#define c_eq f2_eq
#include "curve_tmpl.h"

g1a_t G1_GEN; g2a_t G2_GEN;
static fq2_t TWIST_B;    /* 3/(9+u) */
static fe_t G1_B;

void g1a_neg(g1a_t *o, const g1a_t *a) { *o = *a; if (!a->inf) q_neg(&o->y, &a->y); }
int g1a_on_curve(const g1a_t *a) {
    if (a->inf) return 1;
    fe_t l, r; q_sqr(&l, &a->y); q_sqr(&r, &a->x); q_mul(&r, &r, &a->x); q_add(&r, &r, &G1_B); return fe_eq(&l, &r);
}
int g2a_on_curve(const g2a_t *a) {
    if (a->inf) return 1;
    fq2_t l, r; f2_sqr(&l, &a->y); f2_sqr(&r, &a->x); f2_mul(&r, &r, &a->x); f2_add(&r, &r, &TWIST_B); return f2_eq(&l, &r);
}
static void q_from_dec(fe_t *o, const char *s) { uint64_t t[4]; dec_to_u256(s, strlen(s), t); fe_from_u64x4(o, t, Q); }
void zko_curve_init(void) {
    (void)g1_eq_dummy;
    fe_set_u64(&G1_GEN.x, 1, Q); fe_set_u64(&G1_GEN.y, 2, Q); G1_GEN.inf = 0; fe_set_u64(&G1_B, 3, Q);
    q_from_dec(&G2_GEN.x.c0, "10857046999023057135944570762232829481370756359578518086990519993285655852781");
    q_from_dec(&G2_GEN.x.c1, "11559732032986387107991004021392285783925812861821192530917403151452391805634");
    q_from_dec(&G2_GEN.y.c0, "8495653923123431417604973247489272438418190587263600148770280649306958101930");
    q_from_dec(&G2_GEN.y.c1, "4082367875863433681332203403145435568316851327593401208105741076214120093531");
    G2_GEN.inf = 0;
    fq2_t xi, three; fe_set_u64(&xi.c0, 9, Q); fe_set_u64(&xi.c1, 1, Q); fe_set_u64(&three.c0, 3, Q); q_zero(&three.c1);
    f2_inv(&xi, &xi); f2_mul(&TWIST_B, &three, &xi);
}

/* ---------------- Fq6 = Fq2[v]/(v^3 - xi), Fq12 = Fq6[w]/(w^2 - v) ---------------- */
typedef struct { fq2_t a0, a1, a2; } fq6_t;
typedef struct { fq6_t a, b; } fq12_t;
static void f6_add(fq6_t *o, const fq6_t *x, const fq6_t *y) { f2_add(&o->a0, &x->a0, &y->a0); f2_add(&o->a1, &x->a1, &y->a1); f2_add(&o->a2, &x->a2, &y->a2); }
static void f6_sub(fq6_t *o, const fq6_t *x, const fq6_t *y) { f2_sub(&o->a0, &x->a0, &y->a0); f2_sub(&o->a1, &x->a1, &y->a1); f2_sub(&o->a2, &x->a2, &y->a2); }
static void f6_neg(fq6_t *o, const fq6_t *x) { f2_neg(&o->a0, &x->a0); f2_neg(&o->a1, &x->a1); f2_neg(&o->a2, &x->a2); }
static void f6_mul(fq6_t *o, const fq6_t *x, const fq6_t *y) {
    fq2_t t, u, c0, c1, c2;
    f2_mul(&c0, &x->a0, &y->a0); f2_mul(&t, &x->a1, &y->a2); f2_mul(&u, &x->a2, &y->a1); f2_add(&t, &t, &u); f2_mul_xi(&t, &t); f2_add(&c0, &c0, &t);
    f2_mul(&c1, &x->a0, &y->a1); f2_mul(&t, &x->a1, &y->a0); f2_add(&c1, &c1, &t); f2_mul(&t, &x->a2, &y->a2); f2_mul_xi(&t, &t); f2_add(&c1, &c1, &t);
    f2_mul(&c2, &x->a0, &y->a2); f2_mul(&t, &x->a1, &y->a1); f2_add(&c2, &c2, &t); f2_mul(&t, &x->a2, &y->a0); f2_add(&c2, &c2, &t);
    o->a0 = c0; o->a1 = c1; o->a2 = c2;
}
static void f6_mul_v(fq6_t *o, const fq6_t *x) { fq2_t t; f2_mul_xi(&t, &x->a2); fq2_t a0 = x->a0, a1 = x->a1; o->a0 = t; o->a1 = a0; o->a2 = a1; }
static void f6_inv(fq6_t *o, const fq6_t *x) {
    fq2_t c0, c1, c2, t, u;
    f2_sqr(&c0, &x->a0); f2_mul(&t, &x->a1, &x->a2); f2_mul_xi(&t, &t); f2_sub(&c0, &c0, &t);
    f2_sqr(&c1, &x->a2); f2_mul_xi(&c1, &c1); f2_mul(&t, &x->a0, &x->a1); f2_sub(&c1, &c1, &t);
    f2_sqr(&c2, &x->a1); f2_mul(&t, &x->a0, &x->a2); f2_sub(&c2, &c2, &t);
    f2_mul(&t, &x->a2, &c1); f2_mul(&u, &x->a1, &c2); f2_add(&t, &t, &u); f2_mul_xi(&t, &t); f2_mul(&u, &x->a0, &c0); f2_add(&t, &t, &u);
    f2_inv(&t, &t);
    f2_mul(&o->a0, &c0, &t); f2_mul(&o->a1, &c1, &t); f2_mul(&o->a2, &c2, &t);
}
static void f12_one(fq12_t *o) { memset(o, 0, sizeof *o); f2_one(&o->a.a0); }
static void f12_mul(fq12_t *o, const fq12_t *x, const fq12_t *y) {
    fq6_t aa, bb, t, u, ra, rb;
    f6_mul(&aa, &x->a, &y->a); f6_mul(&bb, &x->b, &y->b);
    f6_mul_v(&t, &bb); f6_add(&ra, &aa, &t);
    f6_mul(&t, &x->a, &y->b); f6_mul(&u, &x->b, &y->a); f6_add(&rb, &t, &u);
    o->a = ra; o->b = rb;
}
static void f12_conj(fq12_t *o, const fq12_t *x) { o->a = x->a; f6_neg(&o->b, &x->b); }
static void f12_inv(fq12_t *o, const fq12_t *x) {
    fq6_t t, u; f6_mul(&t, &x->a, &x->a); f6_mul(&u, &x->b, &x->b); f6_mul_v(&u, &u); f6_sub(&t, &t, &u); f6_inv(&t, &t);
    f6_mul(&o->a, &x->a, &t); f6_mul(&u, &x->b, &t); f6_neg(&o->b, &u);
}
static int f12_is_one(const fq12_t *x) { fq12_t one; f12_one(&one); return memcmp(x, &one, sizeof one) == 0; }

/* line through twist points (slope lam on the twist) evaluated at P in G1: yP + (-lam xP) w + (lam xT - yT) w^3 */
static void line_eval(fq12_t *l, const fq2_t *lam, const fq2_t *xT, const fq2_t *yT, const g1a_t *P) {
    memset(l, 0, sizeof *l);
    l->a.a0.c0 = P->y;
    fq2_t t; t.c0 = P->x; q_zero(&t.c1); f2_mul(&t, lam, &t); f2_neg(&l->b.a0, &t);
    f2_mul(&t, lam, xT); f2_sub(&l->b.a1, &t, yT);
}
static void miller(fq12_t *f, const g1a_t *P, const g2a_t *Qp) {
    /* T = 6x^2, x = 4965661367192848881 */
    static const uint64_t T[2] = {0xf83e9682e87cfd46ULL, 0x6f4d8248eeb859fbULL};
    f12_one(f);
    if (P->inf || Qp->inf) return;
    fq2_t xR = Qp->x, yR = Qp->y, lam, t, u, x3, y3; fq12_t l;
    int rinf = 0;
    for (int i = 125; i >= 0; i--) {
        f12_mul(f, f, f);
        if (!rinf) {
            /* tangent */
            f2_sqr(&t, &xR); f2_dbl(&u, &t); f2_add(&t, &t, &u); f2_dbl(&u, &yR); f2_inv(&u, &u); f2_mul(&lam, &t, &u);
            line_eval(&l, &lam, &xR, &yR, P); f12_mul(f, f, &l);
            f2_sqr(&x3, &lam); f2_sub(&x3, &x3, &xR); f2_sub(&x3, &x3, &xR);
            f2_sub(&t, &xR, &x3); f2_mul(&y3, &lam, &t); f2_sub(&y3, &y3, &yR);
            xR = x3; yR = y3;
        }
        if ((T[i >> 6] >> (i & 63)) & 1) {
            if (rinf) { xR = Qp->x; yR = Qp->y; rinf = 0; continue; }
            f2_sub(&t, &Qp->x, &xR);
            if (f2_is_zero(&t)) { rinf = 1; continue; }   /* vertical line: killed by final exponentiation */
            f2_sub(&u, &Qp->y, &yR); f2_inv(&t, &t); f2_mul(&lam, &u, &t);
            line_eval(&l, &lam, &xR, &yR, P); f12_mul(f, f, &l);
            f2_sqr(&x3, &lam); f2_sub(&x3, &x3, &xR); f2_sub(&x3, &x3, &Qp->x);
            f2_sub(&t, &xR, &x3); f2_mul(&y3, &lam, &t); f2_sub(&y3, &y3, &yR);
            xR = x3; yR = y3;
        }
    }
}
static void final_exp(fq12_t *o, const fq12_t *f) {
    /* (q^12-1)/r = (q^6-1) * ((q^6+1)/r);  f^(q^6) = conj(f) */
    static const uint64_t E[20] = {
        0x5250a54036e3f812ULL, 0xa5635f1596789051ULL, 0xd1138bf54d5bd1d4ULL, 0xa8ce2533be36c7a2ULL, 0x94f69f6b84e09bf6ULL,
        0x42ad1f5e50ef3644ULL, 0x0fcc420e48c3454cULL, 0x758e4408ecc9952cULL, 0xc901bf1887c6042cULL, 0xa733cd65b14bb3b5ULL,
        0xdf6d76bdcf51b0d8ULL, 0xca64c0fd82eb59e1ULL, 0x1d2e5726e39276a1ULL, 0xc2d1ea74a391cae9ULL, 0x07409206c82d647eULL,
        0x051c6d1aa5afdd17ULL, 0xb37f601919667af5ULL, 0x150e578c5084015bULL, 0xfbdea556c23998e4ULL, 0x000fd14cc52f5b83ULL};
    fq12_t c, i, b, r; f12_conj(&c, f); f12_inv(&i, f); f12_mul(&b, &c, &i);
    f12_one(&r);
    for (int k = 1267; k >= 0; k--) {
        f12_mul(&r, &r, &r);
        if ((E[k >> 6] >> (k & 63)) & 1) f12_mul(&r, &r, &b);
    }
    *o = r;
}
int pairing_product_is_one(const g1a_t *P, const g2a_t *Qs, int n) {
    fq12_t acc, f; f12_one(&acc);
    for (int i = 0; i < n; i++) { miller(&f, &P[i], &Qs[i]); f12_mul(&acc, &acc, &f); }
    final_exp(&acc, &acc);
    return f12_is_one(&acc);
}
