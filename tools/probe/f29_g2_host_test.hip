// host check of zkc_f29_g2.h: chains of mixed additions in radix 2^29 against the generic Fq2 formulas (arbitrary field elements:
// the addition formulas are polynomial identities, the points need not lie on the curve)
#include <cstdio>
#include <cstring>
#include <random>
#include "zkc_curve.h"
#include "zkc_f29_g2.h"
using namespace zkc;
static std::mt19937_64 rng(777);
static Fq rnd() { Fq r; for (int i = 0; i < 8; i++) r.v[i] = (uint32_t)rng(); r.v[7] &= 0x0fffffff; return r; }
static Fq2 rnd2() { return {rnd(), rnd()}; }
static void enter2(F2x29& r, const Fq2& a) { f29_enter_fq(r.c0, a.c0.v); f29_enter_fq(r.c1, a.c1.v); }
static Fq2 leave2(const F2x29& a) { return {f29_to_fp<FqParams>(a.c0), f29_to_fp<FqParams>(a.c1)}; }
static uint32_t maxlimb(const uint32_t a[9]) { uint32_t m = 0; for (int i = 0; i < 8; i++) m = a[i] > m ? a[i] : m; return m; }
int main() {
    int bad = 0; uint32_t worst_top = 0;
    for (int chain = 0; chain < 300; chain++) {
        G2Affine a0 = {rnd2(), rnd2()};
        XYZZ<Fq2> ref = XYZZ<Fq2>::from_affine(a0);
        Acc29G2 acc; enter2(acc.X, a0.x); enter2(acc.Y, a0.y);
        memcpy(acc.ZZ.c0, F29K<FqParams>::one.l, 36); memset(acc.ZZ.c1, 0, 36); acc.ZZZ = acc.ZZ;
        for (int it = 0; it < 40; it++) {
            G2Affine a = {rnd2(), rnd2()};
            if (it == 7) { a = xyzz_to_affine(ref); }                       // equal point: must report same x, same y
            if (it == 9) { a = xyzz_to_affine(ref); a.y = fp_neg(a.y); }    // opposite point
            F2x29 x2, y2; enter2(x2, a.x); enter2(y2, a.y);
            bool same_y = false;
            const bool ok = f29g2_madd(acc, x2, y2, same_y);
            if (it == 7 || it == 9) { if (ok || same_y != (it == 7)) { printf("special case missed it=%d ok=%d same_y=%d\n", it, ok, same_y); bad++; } continue; }
            if (!ok) { printf("false special case\n"); bad++; continue; }
            ref = xyzz_add_affine(ref, a);
            Fq2 X = leave2(acc.X), Y = leave2(acc.Y), ZZ = leave2(acc.ZZ), ZZZ = leave2(acc.ZZZ);
            if (!(X == ref.X && Y == ref.Y && ZZ == ref.ZZ && ZZZ == ref.ZZZ)) { if (bad < 5) printf("mismatch chain %d it %d\n", chain, it); bad++; }
            const uint32_t* all[8] = {acc.X.c0, acc.X.c1, acc.Y.c0, acc.Y.c1, acc.ZZ.c0, acc.ZZ.c1, acc.ZZZ.c0, acc.ZZZ.c1};
            for (int q = 0; q < 8; q++) { if (maxlimb(all[q]) >= (1u << 29)) { printf("limb bound\n"); bad++; } if (all[q][8] > worst_top) worst_top = all[q][8]; }
        }
    }
    // full addition / doubling chains (bucket reduction)
    auto rndpt = [&]() { return XYZZ<Fq2>{rnd2(), rnd2(), rnd2(), rnd2()}; };
    auto same = [&](const Acc29G2& a, const XYZZ<Fq2>& r) {
        if (r.is_inf()) return f29g2_pt_is_inf(a);
        const XYZZ<Fq2> g = f29g2_pt_to_xyzz(a);
        return g.X == r.X && g.Y == r.Y && g.ZZ == r.ZZ && g.ZZZ == r.ZZZ;
    };
    for (int chain = 0; chain < 150; chain++) {
        XYZZ<Fq2> ref = rndpt(); Acc29G2 acc = f29g2_pt_from_xyzz(ref);
        for (int it = 0; it < 40; it++) {
            const int op = (int)(rng() % 8);
            if (op == 0) { ref = xyzz_dbl(ref); f29g2_pt_dbl(acc, acc); }
            else if (op == 1) { XYZZ<Fq2> q = ref; ref = xyzz_add(ref, q); Acc29G2 b = acc; f29g2_pt_add(acc, acc, b); }
            else if (op == 2 && it > 30) { XYZZ<Fq2> q = xyzz_neg(ref); ref = xyzz_add(ref, q); Acc29G2 b = f29g2_pt_from_xyzz(q); f29g2_pt_add(acc, acc, b); }
            else if (op == 3) { Acc29G2 b; f29g2_pt_set_inf(b); f29g2_pt_add(acc, acc, b); }
            else { XYZZ<Fq2> q = rndpt(); ref = xyzz_add(ref, q); Acc29G2 b = f29g2_pt_from_xyzz(q); f29g2_pt_add(acc, acc, b); }
            if (!same(acc, ref)) { if (bad < 5) printf("G2 reduction mismatch chain %d it %d op %d\n", chain, it, op); bad++; break; }
            const uint32_t* all[8] = {acc.X.c0, acc.X.c1, acc.Y.c0, acc.Y.c1, acc.ZZ.c0, acc.ZZ.c1, acc.ZZZ.c0, acc.ZZZ.c1};
            for (int q = 0; q < 8; q++) { if (maxlimb(all[q]) >= (1u << 29)) { printf("limb bound\n"); bad++; } if (all[q][8] > worst_top) worst_top = all[q][8]; }
            if (ref.is_inf()) { ref = rndpt(); acc = f29g2_pt_from_xyzz(ref); }
        }
    }
    printf("G2 radix-2^29 mixed addition: %d mismatches; largest top limb %08x (10 p = %08x)\n", bad, worst_top, (uint32_t)(10.0 * (FqParams::p[7] >> 8)));
    return bad != 0;
}
