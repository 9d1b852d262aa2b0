/* oracle/witness.c -- TEST INFRASTRUCTURE ONLY.
 * Witness calculation for ZkFranchiseProofCircuit(nLevels) (reference: circuit/census.circom:49-115) with the
 * circomlib 2.0.5 templates it instantiates (poseidon.circom optimised schedule, smt/smtverifier.circom and its
 * SMTLevIns/SMTVerifierSM/SMTVerifierLevel/Switcher, bitify Num2Bits/Num2Bits_strict, aliascheck, compconstant,
 * comparators LessEqThan/IsZero/IsEqual/ForceEqualIfEnabled) -- restated from their published definitions
 *  because circomlib is an un-vendored dependency (circuit/package-lock.json:141-142).
 * Output is in the exact wire order circom 2.1.5 -O2 gave the reference's committed
 * artifacts/zkCensus/dev/160/circuit.wasm, recovered with tools/derive_wire_map.py:
 *   [1 | 8 public | 4 private scalars | censusSiblings[0..nL) | sikSiblings[0..nL)] then the component tree
 *   depth-first with components in name order and, inside a component, the signals that survived linear
 *   elimination (outputs, inputs, intermediates in declaration order).
 * Pinned by tests/golden/witness_*.json (sha256 + sampled wires produced by the reference wasm). */
#include "zkc_oracle.h"
#include <string.h>
#include <stdlib.h>
#include "../include/zkc_poseidon_constants.inc"

#define R (&FR)
typedef struct { int t, rp; fe_t *C, *S, *M, *P; } pparams_t;
static pparams_t PP[6];
static int pp_ready = 0;
static fe_t *conv(const unsigned long long (*src)[4], int n) {
    fe_t *o = malloc(sizeof(fe_t) * (size_t)n);
    for (int i = 0; i < n; i++) { uint64_t s[4] = {src[i][0], src[i][1], src[i][2], src[i][3]}; fe_from_u64x4(&o[i], s, R); }
    return o;
}
static void pp_init(void) {
    if (pp_ready) return;
    zko_init();
    PP[3] = (pparams_t){3, 57, conv(ZKC_POSEIDON_C3, 81), conv(ZKC_POSEIDON_S3, 285), conv(ZKC_POSEIDON_M3, 9), conv(ZKC_POSEIDON_P3, 9)};
    PP[4] = (pparams_t){4, 56, conv(ZKC_POSEIDON_C4, 88), conv(ZKC_POSEIDON_S4, 392), conv(ZKC_POSEIDON_M4, 16), conv(ZKC_POSEIDON_P4, 16)};
    PP[5] = (pparams_t){5, 60, conv(ZKC_POSEIDON_C5, 100), conv(ZKC_POSEIDON_S5, 540), conv(ZKC_POSEIDON_M5, 25), conv(ZKC_POSEIDON_P5, 25)};
    pp_ready = 1;
}

/* full trace of one optimised-Poseidon permutation (every signal circom could have kept) */
typedef struct {
    int t, rp;
    fe_t ark[8][5];            /* ark[r].out[j] */
    fe_t mix3[5];              /* mix[3].out[j] (after the P matrix) */
    fe_t sF2[8][5], sF4[8][5]; /* sigmaF[r][j].in2 / in4 */
    fe_t sP2[60], sP4[60];
    fe_t mSout[60][5], mSin0[60];
    fe_t last_in[5];           /* mixLast[0].in[j] = sigmaF[7][j].out */
    fe_t out;
} ptrace_t;

static void sbox(fe_t *o, fe_t *in2, fe_t *in4, const fe_t *x) { fe_mul(in2, x, x, R); fe_mul(in4, in2, in2, R); fe_mul(o, in4, x, R); }
static void matmul(fe_t *o, const fe_t *st, const fe_t *M, int t) {   /* o_i = sum_j M[j][i] st_j */
    fe_t acc, p, tmp[5];
    for (int i = 0; i < t; i++) { memset(&acc, 0, sizeof acc); for (int j = 0; j < t; j++) { fe_mul(&p, &M[j * t + i], &st[j], R); fe_add(&acc, &acc, &p, R); } tmp[i] = acc; }
    memcpy(o, tmp, sizeof(fe_t) * (size_t)t);
}
static void poseidon_trace(ptrace_t *T, const fe_t *in, int nin) {
    pp_init();
    const pparams_t *pp = &PP[nin + 1]; int t = pp->t, rp = pp->rp; T->t = t; T->rp = rp;
    fe_t st[5], o; memset(&st[0], 0, sizeof(fe_t)); for (int j = 1; j < t; j++) st[j] = in[j - 1];
    for (int j = 0; j < t; j++) { fe_add(&st[j], &st[j], &pp->C[j], R); T->ark[0][j] = st[j]; }
    for (int r = 0; r < 3; r++) {
        for (int j = 0; j < t; j++) { sbox(&o, &T->sF2[r][j], &T->sF4[r][j], &st[j]); fe_add(&st[j], &o, &pp->C[(r + 1) * t + j], R); T->ark[r + 1][j] = st[j]; }
        matmul(st, st, pp->M, t);
    }
    for (int j = 0; j < t; j++) { sbox(&o, &T->sF2[3][j], &T->sF4[3][j], &st[j]); fe_add(&st[j], &o, &pp->C[4 * t + j], R); T->ark[4][j] = st[j]; }
    matmul(st, st, pp->P, t); memcpy(T->mix3, st, sizeof(fe_t) * (size_t)t);
    for (int r = 0; r < rp; r++) {
        const fe_t *S = &pp->S[(2 * t - 1) * r];
        sbox(&o, &T->sP2[r], &T->sP4[r], &st[0]); fe_add(&st[0], &o, &pp->C[5 * t + r], R); T->mSin0[r] = st[0];
        fe_t n0, p; memset(&n0, 0, sizeof n0);
        for (int i = 0; i < t; i++) { fe_mul(&p, &S[i], &st[i], R); fe_add(&n0, &n0, &p, R); }
        for (int i = 1; i < t; i++) { fe_mul(&p, &st[0], &S[t + i - 1], R); fe_add(&st[i], &st[i], &p, R); }
        st[0] = n0; memcpy(T->mSout[r], st, sizeof(fe_t) * (size_t)t);
    }
    for (int r = 0; r < 3; r++) {
        for (int j = 0; j < t; j++) { sbox(&o, &T->sF2[4 + r][j], &T->sF4[4 + r][j], &st[j]); fe_add(&st[j], &o, &pp->C[5 * t + rp + r * t + j], R); T->ark[5 + r][j] = st[j]; }
        matmul(st, st, pp->M, t);
    }
    for (int j = 0; j < t; j++) { sbox(&st[j], &T->sF2[7][j], &T->sF4[7][j], &st[j]); T->last_in[j] = st[j]; }
    fe_t acc, p; memset(&acc, 0, sizeof acc);
    for (int j = 0; j < t; j++) { fe_mul(&p, &pp->M[j * t + 0], &st[j], R); fe_add(&acc, &acc, &p, R); }
    T->out = acc;
}
void zko_poseidon(uint64_t out[4], const uint64_t *in, int n) {
    pp_init(); fe_t x[4]; ptrace_t T;
    for (int i = 0; i < n; i++) fe_from_u64x4(&x[i], in + 4 * i, R);
    poseidon_trace(&T, x, n); fe_to_u64x4(out, &T.out, R);
}

/* ---- wire emission ---- */
typedef struct { fe_t *w; int k; } wout_t;
static void put(wout_t *W, const fe_t *v) { W->w[W->k++] = *v; }
static void put_u(wout_t *W, uint64_t x) { fe_t v; fe_set_u64(&v, x, R); put(W, &v); }
/* internals of a t=3/t=4 Poseidon block; cmask bit j set = state position j is a compile-time constant at round 0 */
static void emit_poseidon_std(wout_t *W, const ptrace_t *T, unsigned cmask) {
    int t = T->t;
    for (int r = 1; r <= 7; r++) for (int j = 0; j < t; j++) if (r > 1 || !((cmask >> j) & 1)) put(W, &T->ark[r][j]);
    for (int j = 0; j < t - 1; j++) put(W, &T->last_in[j]);
    for (int r = 0; r < T->rp; r++) put(W, &T->mSout[r][0]);
    for (int r = 0; r < 8; r++) for (int j = 0; j < t; j++) if (r > 0 || !((cmask >> j) & 1)) { put(W, &T->sF2[r][j]); put(W, &T->sF4[r][j]); }
    for (int r = 0; r < T->rp; r++) { put(W, &T->sP2[r]); put(W, &T->sP4[r]); }
}
/* the single t=5 instance (computedNullifier) kept a different survivor set */
static void emit_poseidon_t5(wout_t *W, const ptrace_t *T) {
    for (int j = 1; j < 5; j++) put(W, &T->ark[1][j]);
    for (int r = 2; r <= 3; r++) for (int j = 0; j < 5; j++) put(W, &T->ark[r][j]);
    put(W, &T->ark[4][0]);
    for (int r = 5; r <= 7; r++) for (int j = 0; j < 5; j++) put(W, &T->ark[r][j]);
    put(W, &T->mix3[4]);
    for (int j = 0; j < 4; j++) put(W, &T->last_in[j]);
    for (int r = 0; r <= 56; r++) put(W, &T->mSout[r][4]);
    for (int j = 1; j < 5; j++) put(W, &T->mSout[57][j]);
    put(W, &T->mSout[58][4]); put(W, &T->mSin0[59]);
    for (int r = 0; r < 8; r++) for (int j = 0; j < 5; j++) if (r > 0 || j > 0) { put(W, &T->sF2[r][j]); put(W, &T->sF4[r][j]); }
    for (int r = 0; r < 60; r++) { put(W, &T->sP2[r]); put(W, &T->sP4[r]); }
}
static int bit_of(const uint64_t s[4], int i) { return (int)((s[i >> 6] >> (i & 63)) & 1); }
/* CompConstant(ct = r-1) over the 254 bits of `s` (circomlib compconstant.circom): emits parts[0..126] then the
 * surviving bits of Num2Bits(135)(sum(parts)): out[0..126], out[128..133] */
static void emit_alias_check(wout_t *W, const uint64_t s[4]) {
    uint64_t ct[4]; memcpy(ct, FR.p, 32); ct[0] -= 1;     /* r-1 (r is odd, no borrow) */
    fe_t sum, b, a, e, part, zero; memset(&sum, 0, sizeof sum); memset(&zero, 0, sizeof zero);
    uint64_t bs[4] = {~0ULL, ~0ULL, 0, 0}; fe_from_u64x4(&b, bs, R); fe_set_u64(&a, 1, R); fe_set_u64(&e, 1, R);
    for (int i = 0; i < 127; i++) {
        int c = bit_of(ct, 2 * i) | (bit_of(ct, 2 * i + 1) << 1), v = bit_of(s, 2 * i) | (bit_of(s, 2 * i + 1) << 1);
        part = v > c ? b : v < c ? a : zero;
        put(W, &part); fe_add(&sum, &sum, &part, R);
        fe_sub(&b, &b, &e, R); fe_add(&a, &a, &e, R); fe_add(&e, &e, &e, R);
    }
    uint64_t so[4]; fe_to_u64x4(so, &sum, R);
    for (int i = 0; i < 134; i++) if (i != 127) put_u(W, (uint64_t)bit_of(so, i));
}
/* SMTVerifier(n) with enabled=1, fnc=0, oldKey=oldValue=isOld0=0 (census.circom:79-103). Returns computed root. */
static void emit_smt_verifier(wout_t *W, int n, const fe_t *key, const fe_t *value, const fe_t *sib, fe_t *root_out, int *last_sibling_bad) {
    fe_t one, zero, t; fe_set_u64(&one, 1, R); memset(&zero, 0, sizeof zero);
    uint64_t ks[4]; fe_to_u64x4(ks, key, R);
    fe_t kinv; fe_inv(&kinv, key, R);
    put_u(W, fe_is_zero(key) ? 1 : 0);           /* areKeyEquals.out */
    put(W, &kinv);                                /* areKeyEquals.isz.inv */
    put(W, &zero);                                /* checkRoot.isz.inv (in == 0 in every valid witness) */
    ptrace_t *T = malloc(sizeof *T);
    fe_t hin[3] = {*key, *value, one};
    poseidon_trace(T, hin, 3); fe_t h1new = T->out;
    put(W, &h1new); emit_poseidon_std(W, T, 1u | 8u);
    /* SMTLevIns */
    int *isz = malloc(sizeof(int) * (size_t)n), *lev = calloc((size_t)n, sizeof(int)), *done = calloc((size_t)n, sizeof(int));
    for (int i = 0; i < n; i++) isz[i] = fe_is_zero(&sib[i]);
    *last_sibling_bad = !isz[n - 1];
    lev[n - 1] = 1 - isz[n - 2]; done[n - 2] = lev[n - 1];
    for (int i = n - 2; i > 0; i--) { lev[i] = (1 - done[i]) * (1 - isz[i - 1]); done[i - 1] = lev[i] + done[i]; }
    lev[0] = 1 - done[0];
    /* SMTVerifierSM chain: st_top[i] = prev_top - prev_top*levIns[i]; st_inew[i] = prev_top*levIns[i] */
    int *sttop = malloc(sizeof(int) * (size_t)n), *stnew = malloc(sizeof(int) * (size_t)n), prev = 1;
    for (int i = 0; i < n; i++) { stnew[i] = prev * lev[i]; sttop[i] = prev - stnew[i]; prev = sttop[i]; }
    /* levels bottom-up; remember traces to emit top-down */
    ptrace_t *LT = malloc(sizeof(ptrace_t) * (size_t)n);
    fe_t *childv = malloc(sizeof(fe_t) * (size_t)n), *aux0 = malloc(sizeof(fe_t) * (size_t)n), *Lv = malloc(sizeof(fe_t) * (size_t)n);
    fe_t child = zero;
    for (int i = n - 1; i >= 0; i--) {
        fe_t aux, L, Rr, in2[2];
        childv[i] = child;
        if (bit_of(ks, i)) fe_sub(&aux, &sib[i], &child, R); else aux = zero;
        fe_add(&L, &child, &aux, R); fe_sub(&Rr, &sib[i], &aux, R); Lv[i] = L;
        in2[0] = L; in2[1] = Rr; poseidon_trace(&LT[i], in2, 2);
        if (sttop[i]) aux0[i] = LT[i].out; else aux0[i] = zero;
        child = aux0[i]; if (stnew[i]) fe_add(&child, &child, &h1new, R);
    }
    *root_out = child;
    /* circom kept st_inew[i] for levels 1..n-3; at the tail of the chain the linear constraints
     * st_top[i] = st_top[i-1] - st_inew[i] and (st_na + st_inew)[n-1] === 1 left st_top[n-3] and
     * st_inew[n-1] (== st_top[n-2]) as the free wires instead of st_inew[n-2]. */
    for (int i = 0; i < n - 1; i++) {
        if (i == n - 3) put_u(W, (uint64_t)sttop[i]);
        if (i > 0 && i < n - 2) put_u(W, (uint64_t)stnew[i]);
        put_u(W, (uint64_t)bit_of(ks, i)); put(W, &childv[i]); put(W, &aux0[i]); put(W, &LT[i].out); put(W, &Lv[i]);
        emit_poseidon_std(W, &LT[i], 1u);
    }
    put_u(W, (uint64_t)sttop[n - 2]);
    if (n <= 253) put_u(W, (uint64_t)bit_of(ks, n - 1));                    /* n = 254 (nLevels = 253): key bit 253 is the bit Num2Bits' linear constraint solves for, not a wire */
    for (int i = n; i <= 252; i++) put_u(W, (uint64_t)bit_of(ks, i));      /* n2bNew.out[n..252] */
    emit_alias_check(W, ks);
    uint64_t z4[4] = {0, 0, 0, 0};
    for (int i = 0; i < 253; i++) put(W, &zero);                            /* n2bOld.out[0..252], oldKey = 0 */
    emit_alias_check(W, z4);
    for (int i = 1; i <= n - 2; i++) put_u(W, (uint64_t)lev[i]);
    for (int i = 0; i <= n - 2; i++) {
        if (i < n - 2) put_u(W, (uint64_t)isz[i]);
        fe_inv(&t, &sib[i], R); put(W, &t);
    }
    put(W, &zero);                                                          /* isZero[n-1].inv */
    free(T); free(isz); free(lev); free(done); free(sttop); free(stnew); free(LT); free(childv); free(aux0); free(Lv);
}
int zko_n_wires(int nL) {
    int n = nL + 1;
    int pos3 = 20 + 2 + 57 + 46 + 114;                 /* level hash internals */
    int ver = 3 + 1 + (26 + 3 + 56 + 60 + 112) + (5 + pos3) + (n - 2) * (6 + pos3) + 2 + (253 - n) + 2 * (127 + 133) + 253 + (n - 2) + 2 * (n - 2) + 1 + 1;
    return 1 + 12 + 2 * nL + 2 * ver + 1 + 251 + 296 + 261;
}
int zko_witness(int nL, const uint64_t *inputs, uint64_t *wires) {
    pp_init();
    int n = nL + 1, nin = zko_n_inputs(nL), nw = zko_n_wires(nL), rc = ZKO_OK;
    for (int i = 0; i < nin; i++) if (u256_cmp(inputs + 4 * i, FR.p) >= 0) return ZKO_ERR_INPUT_RANGE;
    fe_t *in = malloc(sizeof(fe_t) * (size_t)nin);
    for (int i = 0; i < nin; i++) fe_from_u64x4(&in[i], inputs + 4 * i, R);
    const fe_t *eid = &in[0], *nullifier = &in[2], *avail = &in[3], *vh = &in[4], *sikRoot = &in[6], *censusRoot = &in[7],
               *address = &in[8], *password = &in[9], *signature = &in[10], *voteW = &in[11], *cs = &in[12], *ss = &in[12 + n];
    wout_t W = {malloc(sizeof(fe_t) * (size_t)nw), 0};
    put_u(&W, 1);
    put(&W, &eid[0]); put(&W, &eid[1]); put(&W, nullifier); put(&W, &vh[0]); put(&W, &vh[1]); put(&W, sikRoot); put(&W, censusRoot); put(&W, voteW);
    put(&W, avail); put(&W, address); put(&W, password); put(&W, signature);
    for (int i = 0; i < nL; i++) put(&W, &cs[i]);
    for (int i = 0; i < nL; i++) put(&W, &ss[i]);
    /* sik = Poseidon(address, password, signature) is needed by sikVerifier but emitted later (component name order) */
    ptrace_t *Tsik = malloc(sizeof *Tsik), *Tnul = malloc(sizeof *Tnul);
    fe_t sin[3] = {*address, *password, *signature}; poseidon_trace(Tsik, sin, 3);
    fe_t nin4[4] = {*signature, *password, eid[0], eid[1]}; poseidon_trace(Tnul, nin4, 4);
    /* The witness is emitted in wire order, but the status is the assert the reference's calculator reaches FIRST; it runs the main template top to
     * bottom (weight :72, sikVerifier :79-90, censusVerifier :92-103, nullifier :114) and, inside an SMTVerifier, the SMTLevIns assert (smtverifier.circom:70)
     * before the root comparison (:134).  tests/golden/witness_vectors.json holds the wasm's message for every pair and triple of violations. */
    fe_t root; int bad, e_weight = 0, e_sik_last = 0, e_sik_root = 0, e_census_last = 0, e_census_root = 0, e_nullifier = 0;
    emit_smt_verifier(&W, n, address, avail, cs, &root, &bad);
    e_census_last = bad; e_census_root = !fe_eq(&root, censusRoot);
    { fe_t z; memset(&z, 0, sizeof z); put(&W, &z); }                       /* checkNullifier.isz.inv */
    /* checkWeight = LessEqThan(252)(voteWeight, availableWeight): bits 0..250 of voteWeight + 2^252 - (availableWeight+1) */
    { fe_t x, p252, one; uint64_t s[4] = {0, 0, 0, 1ULL << 60}; fe_from_u64x4(&p252, s, R); fe_set_u64(&one, 1, R);
      fe_add(&x, voteW, &p252, R); fe_sub(&x, &x, avail, R); fe_sub(&x, &x, &one, R);
      uint64_t xs[4]; fe_to_u64x4(xs, &x, R);
      e_weight = bit_of(xs, 252) || bit_of(xs, 253);
      for (int i = 0; i <= 250; i++) put_u(&W, (uint64_t)bit_of(xs, i)); }
    e_nullifier = !fe_eq(&Tnul->out, nullifier);
    emit_poseidon_t5(&W, Tnul);
    put(&W, &Tsik->out); emit_poseidon_std(&W, Tsik, 1u);
    emit_smt_verifier(&W, n, address, &Tsik->out, ss, &root, &bad);
    e_sik_last = bad; e_sik_root = !fe_eq(&root, sikRoot);
    rc = e_weight ? ZKO_ERR_WEIGHT : e_sik_last ? ZKO_ERR_SIK_LAST_SIBLING : e_sik_root ? ZKO_ERR_SIK_ROOT : e_census_last ? ZKO_ERR_LAST_SIBLING
       : e_census_root ? ZKO_ERR_CENSUS_ROOT : e_nullifier ? ZKO_ERR_NULLIFIER : ZKO_OK;
    if (W.k != nw) { rc = -100 - (W.k > nw); }
    else for (int i = 0; i < nw; i++) fe_to_u64x4(wires + 4 * i, &W.w[i], R);
    free(in); free(W.w); free(Tsik); free(Tnul);
    return rc;
}
