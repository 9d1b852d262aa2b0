/* oracle/bn254_field.c -- TEST INFRASTRUCTURE ONLY.  BN254 (alt_bn128) base field Fq and scalar
 * field Fr, 4x64 Montgomery (CIOS), as used by the reference's provers
 * (ffjavascript F1Field / rapidsnark Fr,Fq; reached from ts_inputs/src/example.ts:358 and
 * zk_census_test.go:89).  Decimal codecs follow the JSON artifact encodings
 * (artifacts/zkCensus/dev/160/{proof,signals,inputs_example}.json). */
#include "bn254.h"
#include <string.h>
typedef unsigned __int128 u128;

field_t FQ, FR;
static const uint64_t Q_LIMBS[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t R_LIMBS[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};

int u256_cmp(const uint64_t a[4], const uint64_t b[4]) {
    for (int i = 3; i >= 0; i--) { if (a[i] < b[i]) return -1; if (a[i] > b[i]) return 1; }
    return 0;
}
static uint64_t add4(uint64_t o[4], const uint64_t a[4], const uint64_t b[4]) {
    u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)a[i] + b[i]; o[i] = (uint64_t)c; c >>= 64; } return (uint64_t)c;
}
static uint64_t sub4(uint64_t o[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - b[i] - br; o[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } return br;
}
static void field_setup(field_t *F, const uint64_t p[4]) {
    memcpy(F->p, p, 32);
    uint64_t x = 1; for (int i = 0; i < 6; i++) x *= 2 - p[0] * x;     /* p^-1 mod 2^64 */
    F->inv = (uint64_t)0 - x;
    /* r1 = 2^256 mod p, r2 = 2^512 mod p by 512 modular doublings of 1 */
    uint64_t t[4] = {1, 0, 0, 0};
    for (int i = 0; i < 512; i++) {
        uint64_t c = add4(t, t, t);
        if (c || u256_cmp(t, p) >= 0) sub4(t, t, p);
        if (i == 255) memcpy(F->r1, t, 32);
    }
    memcpy(F->r2, t, 32);
}
void fe_add(fe_t *o, const fe_t *a, const fe_t *b, const field_t *F) {
    uint64_t c = add4(o->v, a->v, b->v);
    if (c || u256_cmp(o->v, F->p) >= 0) sub4(o->v, o->v, F->p);
}
void fe_sub(fe_t *o, const fe_t *a, const fe_t *b, const field_t *F) {
    if (sub4(o->v, a->v, b->v)) add4(o->v, o->v, F->p);
}
void fe_neg(fe_t *o, const fe_t *a, const field_t *F) {
    if (fe_is_zero(a)) { memset(o, 0, sizeof *o); return; }
    sub4(o->v, F->p, a->v);
}
void fe_mul(fe_t *o, const fe_t *a, const fe_t *b, const field_t *F) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a->v[j] * b->v[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * F->inv;
        c = (u128)m * F->p[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * F->p[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    if (t[4] || u256_cmp(t, F->p) >= 0) sub4(t, t, F->p);
    memcpy(o->v, t, 32);
}
void fe_sqr(fe_t *o, const fe_t *a, const field_t *F) { fe_mul(o, a, a, F); }
void fe_from_u64x4(fe_t *o, const uint64_t s[4], const field_t *F) {
    fe_t a, r2; memcpy(a.v, s, 32); memcpy(r2.v, F->r2, 32); fe_mul(o, &a, &r2, F);
}
void fe_to_u64x4(uint64_t s[4], const fe_t *a, const field_t *F) {
    fe_t one = {{1, 0, 0, 0}}, t; fe_mul(&t, a, &one, F); memcpy(s, t.v, 32);
}
void fe_set_u64(fe_t *o, uint64_t x, const field_t *F) { uint64_t s[4] = {x, 0, 0, 0}; fe_from_u64x4(o, s, F); }
int fe_is_zero(const fe_t *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
int fe_eq(const fe_t *a, const fe_t *b) { return memcmp(a->v, b->v, 32) == 0; }
void fe_pow(fe_t *o, const fe_t *a, const uint64_t e[4], const field_t *F) {
    fe_t r; memcpy(r.v, F->r1, 32); fe_t b = *a;
    for (int i = 255; i >= 0; i--) {
        fe_sqr(&r, &r, F);
        if ((e[i >> 6] >> (i & 63)) & 1) fe_mul(&r, &r, &b, F);
    }
    *o = r;
}
void fe_inv(fe_t *o, const fe_t *a, const field_t *F) {
    uint64_t e[4], two[4] = {2, 0, 0, 0}; sub4(e, F->p, two); fe_pow(o, a, e, F);
}
void fe_from_bytes_reduce(fe_t *o, const uint8_t *be, size_t n, const field_t *F) {
    /* Horner over bytes in Montgomery domain: acc = acc*256 + byte */
    fe_t acc, c256, b; memset(&acc, 0, sizeof acc); fe_set_u64(&c256, 256, F);
    for (size_t i = 0; i < n; i++) { fe_mul(&acc, &acc, &c256, F); fe_set_u64(&b, be[i], F); fe_add(&acc, &acc, &b, F); }
    *o = acc;
}
int dec_to_u256(const char *s, size_t n, uint64_t out[4]) {
    uint64_t t[4] = {0, 0, 0, 0};
    if (n == 0) return -1;
    for (size_t i = 0; i < n; i++) {
        if (s[i] < '0' || s[i] > '9') return -1;
        u128 c = (uint64_t)(s[i] - '0');
        for (int j = 0; j < 4; j++) { c += (u128)t[j] * 10; t[j] = (uint64_t)c; c >>= 64; }
        if (c) return -1;
    }
    memcpy(out, t, 32); return 0;
}
int u256_to_dec(const uint64_t in[4], char *out) {
    uint64_t t[4]; memcpy(t, in, 32); char buf[80]; int n = 0;
    do {
        u128 rem = 0;
        for (int j = 3; j >= 0; j--) { u128 cur = (rem << 64) | t[j]; t[j] = (uint64_t)(cur / 10); rem = cur % 10; }
        buf[n++] = (char)('0' + (int)rem);
    } while (t[0] | t[1] | t[2] | t[3]);
    for (int i = 0; i < n; i++) out[i] = buf[n - 1 - i];
    out[n] = 0; return n;
}
int dec_mod_to_fe(const char *s, size_t n, fe_t *o, const field_t *F) {
    fe_t acc, ten, d; memset(&acc, 0, sizeof acc); fe_set_u64(&ten, 10, F);
    if (n == 0) return -1;
    for (size_t i = 0; i < n; i++) {
        if (s[i] < '0' || s[i] > '9') return -1;
        fe_mul(&acc, &acc, &ten, F); fe_set_u64(&d, (uint64_t)(s[i] - '0'), F); fe_add(&acc, &acc, &d, F);
    }
    *o = acc; return 0;
}
void zko_curve_init(void);
void zko_init(void) {
    static int done = 0; if (done) return;
    field_setup(&FQ, Q_LIMBS); field_setup(&FR, R_LIMBS);
    zko_curve_init();
    done = 1;
}
