/* oracle/bn254.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Plain-C restatement of the arithmetic the reference's hot path runs inside its un-vendored
 * dependencies (snarkjs 0.7.0 / ffjavascript 0.2.59 / wasmcurves 0.2.1 on the TS side,
 * go-rapidsnark on the Go side; call sites ts_inputs/src/example.ts:358-362 and
 * zk_census_test.go:89,122).  None of that source is under /root/reference, so the published
 * algorithms are restated here and pinned by the reference's own committed fixtures
 * (tests/golden/: witness sha256 + sampled wires from circuit.wasm, signals.json, the
 * proof.json/verification_key.json triple).  Prover OUTPUT parity vs snarkjs/rapidsnark is
 * unpinned (random r,s; proving_key.zkey is a missing blob) -- see DESIGN.md.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (zk-franchise-proof-circuit_amd/) never links or calls it.
 *
 * Representation: 4 x u64 little-endian limbs, Montgomery form (R = 2^256) inside Fp/Fr values.
 */
#ifndef ZKC_ORACLE_BN254_H
#define ZKC_ORACLE_BN254_H
#include <stdint.h>
#include <stddef.h>

typedef struct { uint64_t v[4]; } fe_t;            /* element of Fq or Fr, Montgomery form */
typedef struct { uint64_t p[4], r1[4], r2[4], inv; } field_t;
extern field_t FQ, FR;                              /* initialised by zko_init() */
void zko_init(void);

/* generic prime-field ops (F = &FQ or &FR) */
void fe_from_u64x4(fe_t *o, const uint64_t s[4], const field_t *F);   /* standard -> Montgomery (s < p) */
void fe_to_u64x4(uint64_t s[4], const fe_t *a, const field_t *F);     /* Montgomery -> standard */
void fe_from_bytes_reduce(fe_t *o, const uint8_t *be, size_t n, const field_t *F); /* big-endian bytes, any length, mod p */
void fe_set_u64(fe_t *o, uint64_t x, const field_t *F);
void fe_add(fe_t *o, const fe_t *a, const fe_t *b, const field_t *F);
void fe_sub(fe_t *o, const fe_t *a, const fe_t *b, const field_t *F);
void fe_neg(fe_t *o, const fe_t *a, const field_t *F);
void fe_mul(fe_t *o, const fe_t *a, const fe_t *b, const field_t *F);
void fe_sqr(fe_t *o, const fe_t *a, const field_t *F);
void fe_inv(fe_t *o, const fe_t *a, const field_t *F);                /* 0 -> 0 */
void fe_pow(fe_t *o, const fe_t *a, const uint64_t e[4], const field_t *F);
int  fe_is_zero(const fe_t *a);
int  fe_eq(const fe_t *a, const fe_t *b);
int  u256_cmp(const uint64_t a[4], const uint64_t b[4]);
int  dec_to_u256(const char *s, size_t n, uint64_t out[4]);           /* decimal string -> u256 (0 ok / -1 overflow, bad char) */
int  u256_to_dec(const uint64_t in[4], char *out);                    /* returns strlen; out needs 80 bytes */
int  dec_mod_to_fe(const char *s, size_t n, fe_t *o, const field_t *F); /* decimal of any size reduced mod p */

/* extension tower + curve */
typedef struct { fe_t c0, c1; } fq2_t;              /* c0 + c1 u, u^2 = -1 */
typedef struct { fe_t x, y; int inf; } g1a_t;       /* affine */
typedef struct { fe_t x, y, z; } g1j_t;             /* Jacobian, z=0 is infinity */
typedef struct { fq2_t x, y; int inf; } g2a_t;
typedef struct { fq2_t x, y, z; } g2j_t;

void g1j_set_inf(g1j_t *p);
void g1j_from_affine(g1j_t *o, const g1a_t *a);
void g1j_to_affine(g1a_t *o, const g1j_t *p);
void g1j_dbl(g1j_t *o, const g1j_t *p);
void g1j_add(g1j_t *o, const g1j_t *p, const g1j_t *q);
void g1j_add_affine(g1j_t *o, const g1j_t *p, const g1a_t *q);
void g1j_mul(g1j_t *o, const g1j_t *p, const uint64_t k[4]);          /* k standard-form scalar */
void g1a_neg(g1a_t *o, const g1a_t *a);
int  g1a_on_curve(const g1a_t *a);
void g2j_set_inf(g2j_t *p);
void g2j_from_affine(g2j_t *o, const g2a_t *a);
void g2j_to_affine(g2a_t *o, const g2j_t *p);
void g2j_dbl(g2j_t *o, const g2j_t *p);
void g2j_add(g2j_t *o, const g2j_t *p, const g2j_t *q);
void g2j_add_affine(g2j_t *o, const g2j_t *p, const g2a_t *q);
void g2j_mul(g2j_t *o, const g2j_t *p, const uint64_t k[4]);
int  g2a_on_curve(const g2a_t *a);
extern g1a_t G1_GEN; extern g2a_t G2_GEN;

/* product of pairings check: prod e(P_i, Q_i) == 1 */
int pairing_product_is_one(const g1a_t *P, const g2a_t *Q, int n);

#endif
