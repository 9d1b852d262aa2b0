// Host check of the two field inversions of libzkcensus (csrc/zkc_field.h): fp_inv (square and multiply, what the device uses lane-parallel) against fp_inv_gcd (binary Euclid,
// what prove_batch_finish uses to make the points of a small pass affine), in Fq, Fr and Fq2, on edge values and random ones; and xyzz_to_affine_gcd against xyzz_to_affine.
//   hipcc --offload-arch=gfx950 -std=c++17 -O2 -I zk-franchise-proof-circuit_amd/csrc -I include tests/host/field_inv.hip -o field_inv && ./field_inv
#include "zkc_curve.h"
#include <cstdio>
#include <random>
using namespace zkc;
template <class P> static int check(const char* name) {
    std::mt19937_64 g(7); int bad = 0;
    for (int it = 0; it < 2000; it++) {
        uint32_t s[8]; for (int i = 0; i < 8; i++) s[i] = (uint32_t)g();
        s[7] &= 0x1fffffff;
        if (it == 0) { for (int i = 0; i < 8; i++) s[i] = 0; s[0] = 1; }
        if (it == 1) { for (int i = 0; i < 8; i++) s[i] = P::p[i]; s[0] -= 1; }          // p - 1
        if (it == 2) { for (int i = 0; i < 8; i++) s[i] = 0; s[0] = 2; }
        if (it == 3) { for (int i = 0; i < 8; i++) s[i] = 0; s[7] = 0x10000000; }        // a power of two
        const Fp<P> a = fp_from_std<P>(s), x = fp_inv(a), y = fp_inv_gcd(a);
        if (!(x == y) || !((a * y) == Fp<P>::one())) bad++;
    }
    if (!fp_inv_gcd(Fp<P>::zero()).is_zero()) bad++;
    printf("%s: %d mismatches\n", name, bad);
    return bad;
}
int main() {
    int bad = check<FqParams>("Fq") + check<FrParams>("Fr");
    std::mt19937_64 g(11);
    for (int it = 0; it < 300; it++) {
        uint32_t s[4][8]; for (auto& r : s) { for (int i = 0; i < 8; i++) r[i] = (uint32_t)g(); r[7] &= 0x1fffffff; }
        const Fq2 a{fp_from_std<FqParams>(s[0]), fp_from_std<FqParams>(s[1])};
        const Fq2 x = fp_inv(a), y = fp_inv_gcd(a);
        if (!(x.c0 == y.c0) || !(x.c1 == y.c1)) bad++;
        // any (X, Y, ZZ, ZZZ) with ZZ^3 = ZZZ^2: take z random, ZZ = z^2, ZZZ = z^3 (the conversions do not need the point to be on the curve)
        const Fq z = fp_from_std<FqParams>(s[2]); const G1XYZZ p{fp_from_std<FqParams>(s[0]), fp_from_std<FqParams>(s[1]), z * z, z * z * z};
        const G1Affine u = xyzz_to_affine(p), v = xyzz_to_affine_gcd(p);
        if (!(u.x == v.x) || !(u.y == v.y)) bad++;
    }
    if (!xyzz_to_affine_gcd(G1XYZZ::inf()).is_inf() || !xyzz_to_affine_gcd(G2XYZZ::inf()).is_inf()) bad++;
    printf("field inversions: %s\n", bad ? "MISMATCH" : "ok");
    return bad != 0;
}
