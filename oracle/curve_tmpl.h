/* oracle/curve_tmpl.h -- TEST INFRASTRUCTURE ONLY.  Short-Weierstrass a=0 Jacobian arithmetic, instantiated
 * twice (G1 over Fq, G2 over Fq2) by bn254_curve.c.  Macros expected: CF (coord type), CA/CJ (affine/jacobian types),
 * PFX(name), and field ops c_add,c_sub,c_mul,c_sqr,c_neg,c_inv,c_is_zero,c_eq,c_one,c_zero,c_dbl. */
void PFX(set_inf)(CJ *p) { c_one(&p->x); c_one(&p->y); c_zero(&p->z); }
static int PFX(is_inf)(const CJ *p) { return c_is_zero(&p->z); }
void PFX(from_affine)(CJ *o, const CA *a) {
    if (a->inf) { PFX(set_inf)(o); return; }
    o->x = a->x; o->y = a->y; c_one(&o->z);
}
void PFX(to_affine)(CA *o, const CJ *p) {
    if (PFX(is_inf)(p)) { memset(o, 0, sizeof *o); o->inf = 1; return; }
    CF zi, zi2, zi3; c_inv(&zi, &p->z); c_sqr(&zi2, &zi); c_mul(&zi3, &zi2, &zi);
    c_mul(&o->x, &p->x, &zi2); c_mul(&o->y, &p->y, &zi3); o->inf = 0;
}
void PFX(dbl)(CJ *o, const CJ *p) {
    if (PFX(is_inf)(p)) { *o = *p; return; }
    CF A, B, C, D, E, F, t, X3, Y3, Z3;
    c_sqr(&A, &p->x); c_sqr(&B, &p->y); c_sqr(&C, &B);
    c_add(&t, &p->x, &B); c_sqr(&t, &t); c_sub(&t, &t, &A); c_sub(&t, &t, &C); c_dbl(&D, &t);
    c_dbl(&E, &A); c_add(&E, &E, &A); c_sqr(&F, &E);
    c_dbl(&t, &D); c_sub(&X3, &F, &t);
    c_sub(&t, &D, &X3); c_mul(&Y3, &E, &t); c_dbl(&t, &C); c_dbl(&t, &t); c_dbl(&t, &t); c_sub(&Y3, &Y3, &t);
    c_mul(&Z3, &p->y, &p->z); c_dbl(&Z3, &Z3);
    o->x = X3; o->y = Y3; o->z = Z3;
}
void PFX(add)(CJ *o, const CJ *p, const CJ *q) {
    if (PFX(is_inf)(p)) { *o = *q; return; }
    if (PFX(is_inf)(q)) { *o = *p; return; }
    CF Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, t, X3, Y3, Z3;
    c_sqr(&Z1Z1, &p->z); c_sqr(&Z2Z2, &q->z);
    c_mul(&U1, &p->x, &Z2Z2); c_mul(&U2, &q->x, &Z1Z1);
    c_mul(&S1, &p->y, &q->z); c_mul(&S1, &S1, &Z2Z2);
    c_mul(&S2, &q->y, &p->z); c_mul(&S2, &S2, &Z1Z1);
    c_sub(&H, &U2, &U1); c_sub(&rr, &S2, &S1);
    if (c_is_zero(&H)) { if (c_is_zero(&rr)) PFX(dbl)(o, p); else PFX(set_inf)(o); return; }
    c_dbl(&rr, &rr);
    c_dbl(&I, &H); c_sqr(&I, &I); c_mul(&J, &H, &I); c_mul(&V, &U1, &I);
    c_sqr(&X3, &rr); c_sub(&X3, &X3, &J); c_dbl(&t, &V); c_sub(&X3, &X3, &t);
    c_sub(&t, &V, &X3); c_mul(&Y3, &rr, &t); c_mul(&t, &S1, &J); c_dbl(&t, &t); c_sub(&Y3, &Y3, &t);
    c_add(&Z3, &p->z, &q->z); c_sqr(&Z3, &Z3); c_sub(&Z3, &Z3, &Z1Z1); c_sub(&Z3, &Z3, &Z2Z2); c_mul(&Z3, &Z3, &H);
    o->x = X3; o->y = Y3; o->z = Z3;
}
void PFX(add_affine)(CJ *o, const CJ *p, const CA *q) {
    if (q->inf) { *o = *p; return; }
    CJ qj; PFX(from_affine)(&qj, q); PFX(add)(o, p, &qj);
}
void PFX(mul)(CJ *o, const CJ *p, const uint64_t k[4]) {
    CJ r, b = *p; PFX(set_inf)(&r);
    for (int i = 255; i >= 0; i--) {
        PFX(dbl)(&r, &r);
        if ((k[i >> 6] >> (i & 63)) & 1) PFX(add)(&r, &r, &b);
    }
    *o = r;
}
