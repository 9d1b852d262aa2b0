#include <cstdio>
#include <cstring>
#include <random>
#include "zkc_curve.h"
#include "zkc_f29.h"
using namespace zkc;
static std::mt19937_64 rng(12345);
template <class P> Fp<P> rnd() { Fp<P> r; for (int i = 0; i < 8; i++) r.v[i] = (uint32_t)rng(); r.v[7] &= 0x0fffffff; return r; }   // < 2^252 < p : canonical
template <class P> int run(const char* name) {
    int bad = 0;
    constexpr L9 d1 = f29_dominator<P>(1u << 29, 1u << 25), d3 = f29_dominator<P>(3u << 29, 1u << 24);
    // dominators are multiples of p
    { uint32_t t[9]; memcpy(t, d1.l, 36); f29_carry(t); /* not < 64p necessarily: check via to_fp == 0 */
      Fp<P> z = f29_to_fp<P>(t); if (!z.is_zero()) { printf("%s dom1 not multiple of p\n", name); bad++; }
      memcpy(t, d3.l, 36); f29_carry(t); z = f29_to_fp<P>(t); if (!z.is_zero()) { printf("%s dom3 not multiple of p\n", name); bad++; }
      for (int i = 0; i < 8; i++) if (d1.l[i] < (1u << 29) || d1.l[i] >= (1u << 30) || d3.l[i] < (3u << 29)) { printf("%s dom limb range\n", name); bad++; }
      printf("%s dom1 top %08x dom3 top %08x one[0] %08x\n", name, d1.l[8], d3.l[8], F29K<P>::one.l[0]); }
    for (int it = 0; it < 200000; it++) {
        Fp<P> a = rnd<P>(), b = rnd<P>(), c = rnd<P>();
        if (it == 0) { a = Fp<P>::zero(); } if (it == 1) { for (int i = 0; i < 8; i++) a.v[i] = P::p[i]; a.v[0] -= 1; b = a; }
        uint32_t A[9], B[9], C[9], M[9], S[9];
        f29_from_fp_shl5(A, a.v); f29_from_fp_shl5(B, b.v); f29_from_fp_shl5(C, c.v);
        f29_mul<P>(M, A, B);
        Fp<P> got = f29_to_fp<P>(M), want; fp_mul_limbs<P>(want.v, a.v, b.v);
        if (!(got == want)) { if (bad < 5) printf("%s mul mismatch it=%d\n", name, it); bad++; }
        f29_sqr<P>(S, A); got = f29_to_fp<P>(S); fp_mul_limbs<P>(want.v, a.v, a.v);
        if (!(got == want)) { if (bad < 5) printf("%s sqr mismatch it=%d\n", name, it); bad++; }
        // lazy: (a*b - c + D) * (a + b)  vs reference
        uint32_t T[9], U[9], V[9];
        f29_sub(T, M, C, d1);           // limbs < 1.5 * 2^30
        f29_add(U, A, B);               // limbs < 2^30
        { uint32_t Tc[9]; memcpy(Tc, T, 36); f29_carry(Tc); f29_mul<P>(V, Tc, U); }
        Fp<P> ab; fp_mul_limbs<P>(ab.v, a.v, b.v); Fp<P> ref; Fp<P> l = ab - c, r = a + b; fp_mul_limbs<P>(ref.v, l.v, r.v);
        got = f29_to_fp<P>(V);
        if (!(got == ref)) { if (bad < 5) printf("%s lazy mismatch it=%d\n", name, it); bad++; }
        f29_mul<P>(V, U, T);            // uncarried T (1.5 * 2^30) times U (2^30): 2^60.58 -- at the documented limit
        got = f29_to_fp<P>(V); if (!(got == ref)) { if (bad < 5) printf("%s lazy2 mismatch it=%d\n", name, it); bad++; }
        // zero test: a - a + D
        f29_mul<P>(M, A, F29K<P>::one.l); uint32_t M2[9]; f29_mul<P>(M2, A, F29K<P>::one.l);
        f29_sub(T, M, M2, d1); f29_carry(T); if (!f29_is_zero_mod_p<P>(T)) { if (bad < 5) printf("%s zero test miss it=%d\n", name, it); bad++; }
        f29_sub(T, M, C, d1); f29_carry(T); bool z = f29_is_zero_mod_p<P>(T); Fp<P> mm = f29_to_fp<P>(M); bool zr = (f29_to_fp<P>(T).is_zero());
        if (z != zr) { if (bad < 5) printf("%s zero test false it=%d\n", name, it); bad++; }
    }
    printf("%s: %d mismatches\n", name, bad);
    return bad;
}
// full mixed addition in F29 vs xyzz_add_affine
int main() { int bad = run<FqParams>("Fq") + run<FrParams>("Fr"); return bad != 0; }
