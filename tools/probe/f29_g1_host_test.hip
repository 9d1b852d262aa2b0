// host check of zkc_f29_g1.h: full addition / doubling / mixed addition chains in radix 2^29 against the generic XYZZ formulas
// (arbitrary field elements: the formulas are polynomial identities)
#include <cstdio>
#include <cstring>
#include <random>
#include "zkc_curve.h"
#include "zkc_f29_g1.h"
using namespace zkc;
static std::mt19937_64 rng(4242);
static Fq rnd() { Fq r; for (int i = 0; i < 8; i++) r.v[i] = (uint32_t)rng(); r.v[7] &= 0x0fffffff; return r; }
static G1XYZZ rndpt() { return {rnd(), rnd(), rnd(), rnd()}; }
static bool same(const Acc29& a, const G1XYZZ& r) {
    if (r.is_inf()) return f29_pt_is_inf(a);
    const G1XYZZ g = f29_pt_to_xyzz(a);
    return g.X == r.X && g.Y == r.Y && g.ZZ == r.ZZ && g.ZZZ == r.ZZZ;
}
int main() {
    int bad = 0; uint32_t worst = 0;
    for (int chain = 0; chain < 400; chain++) {
        G1XYZZ ref = rndpt(); Acc29 acc = f29_pt_from_xyzz(ref);
        for (int it = 0; it < 60; it++) {
            const int op = (int)(rng() % 8);
            if (op == 0) { ref = xyzz_dbl(ref); f29_pt_dbl(acc, acc); }
            else if (op == 1) { G1XYZZ q = ref; ref = xyzz_add(ref, q); Acc29 b = acc; f29_pt_add(acc, acc, b); }                       // P + P -> doubling branch
            else if (op == 2 && it > 50) { G1XYZZ q = xyzz_neg(ref); ref = xyzz_add(ref, q); Acc29 b = f29_pt_from_xyzz(q); f29_pt_add(acc, acc, b); }   // P + (-P) -> infinity
            else if (op == 3) { G1XYZZ q = G1XYZZ::inf(); ref = xyzz_add(ref, q); Acc29 b; f29_pt_set_inf(b); f29_pt_add(acc, acc, b); }
            else { G1XYZZ q = rndpt(); ref = xyzz_add(ref, q); Acc29 b = f29_pt_from_xyzz(q); Acc29 sw; f29_pt_add(sw, b, acc); f29_pt_add(acc, acc, b);
                   // the swapped order gives another representative of the same point: X, ZZ equal, Y and ZZZ negated
                   const G1XYZZ g1 = f29_pt_to_xyzz(acc), g2 = f29_pt_to_xyzz(sw);
                   if (!(g1.X == g2.X && g1.ZZ == g2.ZZ && g1.Y == fp_neg(g2.Y) && g1.ZZZ == fp_neg(g2.ZZZ))) { printf("swapped operands disagree\n"); bad++; } }
            if (!same(acc, ref)) { if (bad < 5) printf("mismatch chain %d it %d op %d\n", chain, it, op); bad++; break; }
            const uint32_t* all[4] = {acc.X, acc.Y, acc.ZZ, acc.ZZZ};
            for (int q = 0; q < 4; q++) { for (int k = 0; k < 8; k++) if (all[q][k] >= (1u << 29)) { printf("limb bound\n"); bad++; } if (all[q][8] > worst) worst = all[q][8]; }
            if (ref.is_inf()) { ref = rndpt(); acc = f29_pt_from_xyzz(ref); }
        }
    }
    // mixed addition chain through the same header
    for (int chain = 0; chain < 200; chain++) {
        G1Affine a0 = {rnd(), rnd()}; G1XYZZ ref = G1XYZZ::from_affine(a0);
        Acc29 acc; f29_enter_fq(acc.X, a0.x.v); f29_enter_fq(acc.Y, a0.y.v); memcpy(acc.ZZ, F29K<FqParams>::one.l, 36); memcpy(acc.ZZZ, F29K<FqParams>::one.l, 36);
        for (int it = 0; it < 40; it++) {
            G1Affine a = {rnd(), rnd()}; uint32_t x2[9], y2[9]; f29_from_fp_shl5(x2, a.x.v); f29_from_fp_shl5(y2, a.y.v);
            bool sy; if (!f29_madd(acc, x2, y2, sy)) { printf("false special\n"); bad++; continue; }
            ref = xyzz_add_affine(ref, a);
            if (!same(acc, ref)) { if (bad < 5) printf("madd mismatch\n"); bad++; break; }
        }
    }
    printf("G1 radix-2^29 group operations: %d mismatches; largest top limb %08x (32 p = %08x)\n", bad, worst, 32u * (FqParams::p[7] >> 8));
    return bad != 0;
}
