/* oracle/groth16.c -- TEST INFRASTRUCTURE ONLY.
 * Groth16 prove / verify restated from the published snarkjs 0.7.0 algorithm (groth16_prove.js: buildABC1,
 * ifft -> batchApplyKey(inc = w_{2n}) -> fft, joinABC, five multiExpAffine, blinding with r,s) that the reference
 * reaches at ts_inputs/src/example.ts:358-362, and from go-rapidsnark's verifier (zk_census_test.go:122).
 * snarkjs / ffjavascript / rapidsnark are un-vendored, so: verify is pinned by the reference's committed
 * (proof.json, signals.json, verification_key.json); prove is "parity unpinned" against the reference provers
 * (random r,s, proving_key.zkey is a missing blob) and is pinned instead by the toxic-waste closed form of the test
 * setup: tests/closed_form.py evaluates pi_a, pi_b, pi_c as scalars from (seed, .r1cs, witness, r, s) with Lagrange
 * evaluation at tau -- no NTT, no MSM, no .zkey -- and tests/test_closed_form_prover.py requires byte equality with
 * this file's proofs at nLevels 10 and 160 (and with the GPU's). */
#include "zkc_oracle.h"
#include <string.h>
#include <stdlib.h>

#define R (&FR)
#define Q (&FQ)
static void rd_fq(fe_t *o, const uint8_t *p) { uint64_t s[4]; memcpy(s, p, 32); fe_from_u64x4(o, s, Q); }
static void wr_fq(uint8_t *p, const fe_t *a) { uint64_t s[4]; fe_to_u64x4(s, a, Q); memcpy(p, s, 32); }
static int all_zero(const uint8_t *p, size_t n) { for (size_t i = 0; i < n; i++) if (p[i]) return 0; return 1; }
static void rd_g1(g1a_t *o, const uint8_t *p) { if (all_zero(p, 64)) { memset(o, 0, sizeof *o); o->inf = 1; return; } rd_fq(&o->x, p); rd_fq(&o->y, p + 32); o->inf = 0; }
static void rd_g2(g2a_t *o, const uint8_t *p) {
    if (all_zero(p, 128)) { memset(o, 0, sizeof *o); o->inf = 1; return; }
    rd_fq(&o->x.c0, p); rd_fq(&o->x.c1, p + 32); rd_fq(&o->y.c0, p + 64); rd_fq(&o->y.c1, p + 96); o->inf = 0;
}
static void wr_g1(uint8_t *p, const g1a_t *a) { if (a->inf) { memset(p, 0, 64); return; } wr_fq(p, &a->x); wr_fq(p + 32, &a->y); }
static void wr_g2(uint8_t *p, const g2a_t *a) {
    if (a->inf) { memset(p, 0, 128); return; }
    wr_fq(p, &a->x.c0); wr_fq(p + 32, &a->x.c1); wr_fq(p + 64, &a->y.c0); wr_fq(p + 96, &a->y.c1);
}
/* zkey points are stored already in Montgomery form */
static void rdm_g1(g1a_t *o, const uint8_t *p) { if (all_zero(p, 64)) { memset(o, 0, sizeof *o); o->inf = 1; return; } memcpy(o->x.v, p, 32); memcpy(o->y.v, p + 32, 32); o->inf = 0; }
static void rdm_g2(g2a_t *o, const uint8_t *p) {
    if (all_zero(p, 128)) { memset(o, 0, sizeof *o); o->inf = 1; return; }
    memcpy(o->x.c0.v, p, 32); memcpy(o->x.c1.v, p + 32, 32); memcpy(o->y.c0.v, p + 64, 32); memcpy(o->y.c1.v, p + 96, 32); o->inf = 0;
}

int zko_groth16_verify(const uint8_t *vk, int nPub, const uint8_t *pub, const uint8_t *proof) {
    zko_init();
    g1a_t alpha, A, C, ic, P[4]; g2a_t beta, gamma, delta, B, Qs[4];
    rd_g1(&alpha, vk); rd_g2(&beta, vk + 64); rd_g2(&gamma, vk + 192); rd_g2(&delta, vk + 320);
    rd_g1(&A, proof); rd_g2(&B, proof + 64); rd_g1(&C, proof + 192);
    if (!g1a_on_curve(&A) || !g1a_on_curve(&C) || !g2a_on_curve(&B)) return 0;
    g1j_t acc, t; rd_g1(&ic, vk + 448); g1j_from_affine(&acc, &ic);
    for (int i = 0; i < nPub; i++) {
        uint64_t k[4]; memcpy(k, pub + 32 * i, 32);
        if (u256_cmp(k, FR.p) >= 0) return 0;
        rd_g1(&ic, vk + 448 + 64 * (i + 1)); g1j_from_affine(&t, &ic); g1j_mul(&t, &t, k); g1j_add(&acc, &acc, &t);
    }
    g1a_t vkx; g1j_to_affine(&vkx, &acc);
    g1a_neg(&P[0], &A); Qs[0] = B; P[1] = alpha; Qs[1] = beta; P[2] = vkx; Qs[2] = gamma; P[3] = C; Qs[3] = delta;
    return pairing_product_is_one(P, Qs, 4);
}

/* ---- NTT over Fr ---- */
void zko_root_of_unity(uint64_t out[4], int logn) {
    zko_init();
    fe_t g, w; fe_set_u64(&g, 5, R);
    uint64_t e[4]; memcpy(e, FR.p, 32); e[0] -= 1;                 /* r-1 */
    /* shift right by 28 */
    for (int i = 0; i < 4; i++) e[i] = (e[i] >> 28) | (i < 3 ? e[i + 1] << 36 : 0);
    fe_pow(&w, &g, e, R);                                          /* order 2^28 */
    for (int i = 28; i > logn; i--) fe_mul(&w, &w, &w, R);
    fe_to_u64x4(out, &w, R);
}
static void ntt_mont(fe_t *a, int logn, int inverse) {
    size_t n = (size_t)1 << logn;
    for (size_t i = 0, j = 0; i < n; i++) {                        /* bit reversal */
        if (i < j) { fe_t t = a[i]; a[i] = a[j]; a[j] = t; }
        size_t m = n >> 1; while (m && (j & m)) { j ^= m; m >>= 1; } j |= m;
    }
    uint64_t ws[4]; fe_t wn;
    for (int s = 1; s <= logn; s++) {
        zko_root_of_unity(ws, s); fe_from_u64x4(&wn, ws, R);
        if (inverse) fe_inv(&wn, &wn, R);
        size_t m = (size_t)1 << s, h = m >> 1;
        fe_t *tw = malloc(sizeof(fe_t) * h); memcpy(tw[0].v, FR.r1, 32);
        for (size_t k = 1; k < h; k++) fe_mul(&tw[k], &tw[k - 1], &wn, R);
        for (size_t b = 0; b < n; b += m) for (size_t k = 0; k < h; k++) {
            fe_t t, u = a[b + k]; fe_mul(&t, &a[b + k + h], &tw[k], R);
            fe_add(&a[b + k], &u, &t, R); fe_sub(&a[b + k + h], &u, &t, R);
        }
        free(tw);
    }
    if (inverse) { fe_t ninv; fe_set_u64(&ninv, (uint64_t)n, R); fe_inv(&ninv, &ninv, R); for (size_t i = 0; i < n; i++) fe_mul(&a[i], &a[i], &ninv, R); }
}
void zko_ntt(uint64_t *data, int logn, int inverse) {
    zko_init(); size_t n = (size_t)1 << logn; fe_t *a = malloc(sizeof(fe_t) * n);
    for (size_t i = 0; i < n; i++) fe_from_u64x4(&a[i], data + 4 * i, R);
    ntt_mont(a, logn, inverse);
    for (size_t i = 0; i < n; i++) fe_to_u64x4(data + 4 * i, &a[i], R);
    free(a);
}

/* ---- MSM: bucket method (Pippenger), window c, unsigned digits; deliberately simple ---- */
#define MSM_BODY(JT, AT, PFX)                                                                         \
    int c = n < 32 ? 3 : n < 1024 ? 7 : n < 65536 ? 11 : 14; int nw = (254 + c - 1) / c;               \
    size_t nb = ((size_t)1 << c) - 1; JT *bk = malloc(sizeof(JT) * nb); JT total, run, sum; PFX##set_inf(&total); \
    for (int w = nw - 1; w >= 0; w--) {                                                              \
        for (int d = 0; d < c; d++) PFX##dbl(&total, &total);                                        \
        for (size_t b = 0; b < nb; b++) PFX##set_inf(&bk[b]);                                        \
        for (size_t i = 0; i < n; i++) {                                                             \
            if (pts[i].inf) continue; const uint64_t *k = sc + 4 * i; int bit = w * c; uint64_t dgt = 0; \
            for (int t = 0; t < c && bit + t < 256; t++) dgt |= ((k[(bit + t) >> 6] >> ((bit + t) & 63)) & 1) << t; \
            if (dgt) PFX##add_affine(&bk[dgt - 1], &bk[dgt - 1], &pts[i]);                           \
        }                                                                                            \
        PFX##set_inf(&run); PFX##set_inf(&sum);                                                      \
        for (size_t b = nb; b-- > 0;) { PFX##add(&run, &run, &bk[b]); PFX##add(&sum, &sum, &run); }  \
        PFX##add(&total, &total, &sum);                                                              \
    }                                                                                                \
    free(bk);
static void msm_g1(g1j_t *out, const g1a_t *pts, const uint64_t *sc, size_t n) { MSM_BODY(g1j_t, g1a_t, g1j_) *out = total; }
static void msm_g2(g2j_t *out, const g2a_t *pts, const uint64_t *sc, size_t n) { MSM_BODY(g2j_t, g2a_t, g2j_) *out = total; }
void zko_msm_g1(uint8_t out[64], const uint8_t *bases, const uint8_t *scalars, size_t n) {
    zko_init(); g1a_t *p = malloc(sizeof(g1a_t) * (n ? n : 1)); uint64_t *s = malloc(32 * (n ? n : 1));
    for (size_t i = 0; i < n; i++) rd_g1(&p[i], bases + 64 * i);
    memcpy(s, scalars, 32 * n); g1j_t r; msm_g1(&r, p, s, n); g1a_t a; g1j_to_affine(&a, &r); wr_g1(out, &a); free(p); free(s);
}
void zko_msm_g2(uint8_t out[128], const uint8_t *bases, const uint8_t *scalars, size_t n) {
    zko_init(); g2a_t *p = malloc(sizeof(g2a_t) * (n ? n : 1)); uint64_t *s = malloc(32 * (n ? n : 1));
    for (size_t i = 0; i < n; i++) rd_g2(&p[i], bases + 128 * i);
    memcpy(s, scalars, 32 * n); g2j_t r; msm_g2(&r, p, s, n); g2a_t a; g2j_to_affine(&a, &r); wr_g2(out, &a); free(p); free(s);
}
void zko_g1_mul(uint8_t out[64], const uint8_t base[64], const uint8_t k[32]) {
    zko_init(); g1a_t a; rd_g1(&a, base); g1j_t j; g1j_from_affine(&j, &a); uint64_t kk[4]; memcpy(kk, k, 32); g1j_mul(&j, &j, kk); g1j_to_affine(&a, &j); wr_g1(out, &a);
}

/* ---- .zkey ---- */
static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
int zko_zkey_parse(const uint8_t *buf, size_t len, zko_zkey_t *z) {
    memset(z, 0, sizeof *z);
    if (len < 12 || memcmp(buf, "zkey", 4) || rd32(buf + 4) != 1) return -1;
    uint32_t nsec = rd32(buf + 8); size_t p = 12; const uint8_t *sec[16] = {0}; uint64_t ssz[16] = {0};
    for (uint32_t i = 0; i < nsec; i++) {
        if (p + 12 > len) return -2;
        uint32_t id = rd32(buf + p); uint64_t sz = rd64(buf + p + 4); p += 12;
        if (p + sz > len) return -2;
        if (id < 16) { sec[id] = buf + p; ssz[id] = sz; }
        p += sz;
    }
    if (!sec[1] || rd32(sec[1]) != 1) return -3;
    for (int i = 2; i <= 9; i++) if (!sec[i]) return -4;
    const uint8_t *h = sec[2];
    if (rd32(h) != 32 || memcmp(h + 4, FQ.p, 32) || rd32(h + 36) != 32 || memcmp(h + 40, FR.p, 32)) return -5;
    z->nVars = rd32(h + 72); z->nPublic = rd32(h + 76); z->domainSize = rd32(h + 80);
    z->alpha1 = h + 84; z->beta1 = h + 148; z->beta2 = h + 212; z->gamma2 = h + 340; z->delta1 = h + 468; z->delta2 = h + 532;
    z->ic = sec[3]; z->nCoeffs = rd32(sec[4]); z->coeffs = sec[4] + 4;
    z->pointsA = sec[5]; z->pointsB1 = sec[6]; z->pointsB2 = sec[7]; z->pointsC = sec[8]; z->pointsH = sec[9];
    if (ssz[3] != 64ull * (z->nPublic + 1) || ssz[4] != 4 + 44ull * z->nCoeffs || ssz[5] != 64ull * z->nVars || ssz[6] != 64ull * z->nVars ||
        ssz[7] != 128ull * z->nVars || ssz[8] != 64ull * (z->nVars - z->nPublic - 1) || ssz[9] != 64ull * z->domainSize) return -6;
    return 0;
}
static int ilog2(uint32_t n) { int l = 0; while ((1u << l) < n) l++; return l; }
/* buildABC1: out[m][c] += coef * w[s]; C = A*B pointwise (all Montgomery) */
static void build_abc_mont(const zko_zkey_t *z, const fe_t *w, fe_t *A, fe_t *B, fe_t *C) {
    uint32_t n = z->domainSize; memset(A, 0, sizeof(fe_t) * n); memset(B, 0, sizeof(fe_t) * n);
    fe_t one_std = {{1, 0, 0, 0}};
    for (uint32_t i = 0; i < z->nCoeffs; i++) {
        const uint8_t *e = z->coeffs + 44ull * i; uint32_t m = rd32(e), c = rd32(e + 4), s = rd32(e + 8);
        fe_t v, t; memcpy(v.v, e + 12, 32); fe_mul(&v, &v, &one_std, R);    /* stored as v*R^2 -> v*R */
        fe_mul(&t, &v, &w[s], R); fe_t *dst = (m == 0 ? A : B) + c; fe_add(dst, dst, &t, R);
    }
    for (uint32_t i = 0; i < n; i++) fe_mul(&C[i], &A[i], &B[i], R);
}
int zko_build_abc(const zko_zkey_t *z, const uint64_t *wtns, uint64_t *A, uint64_t *B, uint64_t *C) {
    uint32_t n = z->domainSize; fe_t *w = malloc(sizeof(fe_t) * z->nVars), *a = malloc(sizeof(fe_t) * n), *b = malloc(sizeof(fe_t) * n), *c = malloc(sizeof(fe_t) * n);
    for (uint32_t i = 0; i < z->nVars; i++) fe_from_u64x4(&w[i], wtns + 4 * i, R);
    build_abc_mont(z, w, a, b, c);
    for (uint32_t i = 0; i < n; i++) { fe_to_u64x4(A + 4 * i, &a[i], R); fe_to_u64x4(B + 4 * i, &b[i], R); fe_to_u64x4(C + 4 * i, &c[i], R); }
    free(w); free(a); free(b); free(c); return 0;
}
static void h_evals_mont(const zko_zkey_t *z, const fe_t *w, fe_t *P) {
    uint32_t n = z->domainSize; int logn = ilog2(n);
    fe_t *a = malloc(sizeof(fe_t) * n), *b = malloc(sizeof(fe_t) * n), *c = malloc(sizeof(fe_t) * n);
    build_abc_mont(z, w, a, b, c);
    uint64_t incs[4]; fe_t inc; zko_root_of_unity(incs, logn + 1); fe_from_u64x4(&inc, incs, R);
    fe_t *v[3] = {a, b, c};
    for (int k = 0; k < 3; k++) {
        ntt_mont(v[k], logn, 1);
        fe_t f; memcpy(f.v, FR.r1, 32);
        for (uint32_t i = 0; i < n; i++) { fe_mul(&v[k][i], &v[k][i], &f, R); fe_mul(&f, &f, &inc, R); }
        ntt_mont(v[k], logn, 0);
    }
    for (uint32_t i = 0; i < n; i++) { fe_mul(&P[i], &a[i], &b[i], R); fe_sub(&P[i], &P[i], &c[i], R); }
    free(a); free(b); free(c);
}
int zko_h_evals(const zko_zkey_t *z, const uint64_t *wtns, uint64_t *P) {
    uint32_t n = z->domainSize; fe_t *w = malloc(sizeof(fe_t) * z->nVars), *p = malloc(sizeof(fe_t) * n);
    for (uint32_t i = 0; i < z->nVars; i++) fe_from_u64x4(&w[i], wtns + 4 * i, R);
    h_evals_mont(z, w, p);
    for (uint32_t i = 0; i < n; i++) fe_to_u64x4(P + 4 * i, &p[i], R);
    free(w); free(p); return 0;
}
int zko_groth16_prove(const uint8_t *zkey, size_t len, const uint64_t *wtns, uint32_t nWtns, const uint8_t r[32], const uint8_t s[32],
                      uint8_t proof[256], uint8_t *pub) {
    zko_init(); zko_zkey_t z; int rc = zko_zkey_parse(zkey, len, &z); if (rc) return rc;
    if (nWtns != z.nVars) return 3;                                   /* INVALID_WITNESS_LENGTH */
    uint32_t n = z.domainSize, nv = z.nVars, np = z.nPublic;
    fe_t *w = malloc(sizeof(fe_t) * nv), *P = malloc(sizeof(fe_t) * n);
    for (uint32_t i = 0; i < nv; i++) fe_from_u64x4(&w[i], wtns + 4 * i, R);
    h_evals_mont(&z, w, P);
    uint64_t *Ps = malloc(32ull * n); for (uint32_t i = 0; i < n; i++) fe_to_u64x4(Ps + 4 * i, &P[i], R);
    g1a_t *g1 = malloc(sizeof(g1a_t) * (n > nv ? n : nv)); g2a_t *g2 = malloc(sizeof(g2a_t) * nv);
    g1j_t A, B1, C, H, t; g2j_t B2, t2;
    for (uint32_t i = 0; i < nv; i++) rdm_g1(&g1[i], z.pointsA + 64ull * i);
    msm_g1(&A, g1, wtns, nv);
    for (uint32_t i = 0; i < nv; i++) rdm_g1(&g1[i], z.pointsB1 + 64ull * i);
    msm_g1(&B1, g1, wtns, nv);
    for (uint32_t i = 0; i < nv; i++) rdm_g2(&g2[i], z.pointsB2 + 128ull * i);
    msm_g2(&B2, g2, wtns, nv);
    for (uint32_t i = 0; i < nv - np - 1; i++) rdm_g1(&g1[i], z.pointsC + 64ull * i);
    msm_g1(&C, g1, wtns + 4ull * (np + 1), nv - np - 1);
    for (uint32_t i = 0; i < n; i++) rdm_g1(&g1[i], z.pointsH + 64ull * i);
    msm_g1(&H, g1, Ps, n);
    uint64_t rk[4], sk[4]; memcpy(rk, r, 32); memcpy(sk, s, 32);
    g1a_t alpha1, beta1, delta1; g2a_t beta2, delta2;
    rdm_g1(&alpha1, z.alpha1); rdm_g1(&beta1, z.beta1); rdm_g1(&delta1, z.delta1); rdm_g2(&beta2, z.beta2); rdm_g2(&delta2, z.delta2);
    /* piA = alpha + A + r*delta */
    g1j_t piA, piB1, piC, d1; g2j_t piB2, d2;
    g1j_from_affine(&d1, &delta1); g2j_from_affine(&d2, &delta2);
    g1j_add_affine(&piA, &A, &alpha1); g1j_mul(&t, &d1, rk); g1j_add(&piA, &piA, &t);
    g2j_add_affine(&piB2, &B2, &beta2); g2j_mul(&t2, &d2, sk); g2j_add(&piB2, &piB2, &t2);
    g1j_add_affine(&piB1, &B1, &beta1); g1j_mul(&t, &d1, sk); g1j_add(&piB1, &piB1, &t);
    /* piC = C + H + s*piA + r*piB1 - r*s*delta */
    g1j_add(&piC, &C, &H); g1j_mul(&t, &piA, sk); g1j_add(&piC, &piC, &t); g1j_mul(&t, &piB1, rk); g1j_add(&piC, &piC, &t);
    fe_t rf, sf, rs; fe_from_u64x4(&rf, rk, R); fe_from_u64x4(&sf, sk, R); fe_mul(&rs, &rf, &sf, R); fe_neg(&rs, &rs, R);
    uint64_t nrs[4]; fe_to_u64x4(nrs, &rs, R); g1j_mul(&t, &d1, nrs); g1j_add(&piC, &piC, &t);
    g1a_t a; g2a_t b2;
    g1j_to_affine(&a, &piA); wr_g1(proof, &a); g2j_to_affine(&b2, &piB2); wr_g2(proof + 64, &b2); g1j_to_affine(&a, &piC); wr_g1(proof + 192, &a);
    memcpy(pub, wtns + 4, 32ull * np);
    free(w); free(P); free(Ps); free(g1); free(g2);
    return 0;
}
int zko_zkey_vk(const uint8_t *zkey, size_t len, uint8_t *vk) {
    zko_init(); zko_zkey_t z; int rc = zko_zkey_parse(zkey, len, &z); if (rc) return rc;
    g1a_t a; g2a_t b;
    rdm_g1(&a, z.alpha1); wr_g1(vk, &a); rdm_g2(&b, z.beta2); wr_g2(vk + 64, &b); rdm_g2(&b, z.gamma2); wr_g2(vk + 192, &b); rdm_g2(&b, z.delta2); wr_g2(vk + 320, &b);
    for (uint32_t i = 0; i <= z.nPublic; i++) { rdm_g1(&a, z.ic + 64ull * i); wr_g1(vk + 448 + 64 * i, &a); }
    return 0;
}
