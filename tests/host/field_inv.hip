// Host check of the field products and the two field inversions of libzkcensus (csrc/zkc_field.h): fp_inv (square and multiply, what the device uses lane-parallel) against fp_inv_gcd (binary Euclid,
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
// the host product over 4 x 64-bit limbs (fp_mul_host64, what operator* is on the host) against the 8 x 32-bit CIOS the header started with and the radix-2^29 form the GPU runs
template <class P> static int check_mul(const char* name) {
    std::mt19937_64 g(19); int bad = 0;
    for (int it = 0; it < 200000; it++) {
        uint32_t a[8], b[8], r0[8], r1[8], r2[8];
        for (int i = 0; i < 8; i++) { a[i] = (uint32_t)g(); b[i] = (uint32_t)g(); }
        a[7] &= 0x3fffffff; b[7] &= 0x3fffffff;                                         // any residues below 2^254 > p: the routines take operands below 2p
        if (it < 64) for (int i = 0; i < 8; i++) { a[i] = (it & 1) ? P::p[i] : (it & 2) ? 0xffffffffu >> (i == 7 ? 2 : 0) : 0; b[i] = (it & 4) ? P::p[i] : (it & 8) ? (i == 0) : b[i]; }
        if (it < 64 && (it & 16)) a[0] -= 1;
        fp_mul_limbs<P>(r0, a, b); fp_mul_host64<P>(r1, a, b); fp_mul_r29<P>(r2, a, b);
        for (int i = 0; i < 8; i++) if (r0[i] != r1[i] || r0[i] != r2[i]) { bad++; break; }
    }
    printf("%s products: %d mismatches\n", name, bad);
    return bad;
}
int main() {
    int bad = check<FqParams>("Fq") + check<FrParams>("Fr") + check_mul<FqParams>("Fq") + check_mul<FrParams>("Fr");
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
