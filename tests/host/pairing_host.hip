// Host checks of the verifier's pairing (csrc/zkc_pairing.h), beyond the pin of its value against the reference key's vk_alphabeta_12 (tests/test_oracle_pinning.py):
//   * bilinearity: e([a]P, [b]Q) = e(P, Q)^(ab), and the shared-accumulator loop over several pairs = the product of the single loops;
//   * cyclotomic squaring = plain squaring on elements past the easy part of the final exponentiation;
//   * membership in G2 by the endomorphism (psi(Q) = [6x^2]Q) = membership by definition ([r]Q = infinity), on multiples of the generator AND on twist points
//     outside the subgroup (found by taking square roots in Fq2);
//   * the signed-digit loop constant recomposes to 6x + 2.
//   hipcc --offload-arch=gfx950 -std=c++17 -O2 -I zk-franchise-proof-circuit_amd/csrc -I include tests/host/pairing_host.hip -o pairing_host && ./pairing_host
#include "zkc_pairing.h"
#include <cstdio>
#include <random>
using namespace zkc; using namespace zkc::pairing;

static Fq dec(const char* s) { uint32_t v[8] = {0}; for (const char* p = s; *p; p++) { uint64_t c = (uint64_t)(*p - '0'); for (int i = 0; i < 8; i++) { c += (uint64_t)v[i] * 10; v[i] = (uint32_t)c; c >>= 32; } } return fp_from_std<FqParams>(v); }
static bool eq12(const Fq12& a, const Fq12& b) { return memcmp(&a, &b, sizeof a) == 0; }
static Fq12 pow12(const Fq12& a, uint64_t e) { Fq12 r = one12(); for (int i = 63; i >= 0; i--) { r = sqr12(r); if ((e >> i) & 1) r = r * a; } return r; }
template <class F> static Affine<F> mul_small(const Affine<F>& p, uint64_t k) { uint32_t s[8] = {(uint32_t)k, (uint32_t)(k >> 32), 0, 0, 0, 0, 0, 0}; return xyzz_to_affine_gcd(xyzz_mul(XYZZ<F>::from_affine(p), s)); }
// square root in Fq2 (q = 3 mod 4, the complex method); false when a is no square
static bool sqrt2(const Fq2& a, Fq2& out) {
    if (a.is_zero()) { out = a; return true; }
    uint32_t e1[8], e2[8], q[8]; for (int i = 0; i < 8; i++) q[i] = FqParams::p[i];
    { uint32_t t[8]; for (int i = 0; i < 8; i++) t[i] = q[i]; t[0] -= 3; for (int i = 0; i < 8; i++) e1[i] = (t[i] >> 2) | (i < 7 ? t[i + 1] << 30 : 0); }       // (q - 3) / 4
    { uint32_t t[8]; for (int i = 0; i < 8; i++) t[i] = q[i]; t[0] -= 1; for (int i = 0; i < 8; i++) e2[i] = (t[i] >> 1) | (i < 7 ? t[i + 1] << 31 : 0); }       // (q - 1) / 2
    const Fq2 a1 = fq2_pow(a, e1, 254), alpha = fp_sqr(a1) * a, x0 = a1 * a;
    const Fq2 minus_one{fp_neg(Fq::one()), Fq::zero()};
    if (alpha == minus_one) out = Fq2{Fq::zero(), Fq::one()} * x0;
    else out = fq2_pow(Fq2{Fq::one() + alpha.c0, alpha.c1}, e2, 254) * x0;
    return fp_sqr(out) == a;
}
int main() {
    int bad = 0;
    const G1Affine P{fp_from_u32<FqParams>(1), fp_from_u32<FqParams>(2)};
    const G2Affine Q{{dec("10857046999023057135944570762232829481370756359578518086990519993285655852781"), dec("11559732032986387107991004021392285783925812861821192530917403151452391805634")},
                     {dec("8495653923123431417604973247489272438418190587263600148770280649306958101930"), dec("4082367875863433681332203403145435568316851327593401208105741076214120093531")}};
    if (!g1_on_curve(P) || !g2_on_curve(Q)) { printf("generators are off their curves\n"); return 1; }
    // bilinearity
    const Fq12 e = final_exp(miller(P, Q));
    if (is_one12(e)) { printf("degenerate pairing\n"); bad++; }
    const uint64_t as[3] = {2, 77, 123456789}, bs[3] = {3, 1000003, 987654321};
    for (int i = 0; i < 3; i++) {
        const Fq12 lhs = final_exp(miller(mul_small(P, as[i]), mul_small(Q, bs[i])));
        if (!eq12(lhs, pow12(pow12(e, as[i]), bs[i]))) { printf("bilinearity fails for a = %llu, b = %llu\n", (unsigned long long)as[i], (unsigned long long)bs[i]); bad++; }
    }
    // e(P, Q) e(-P, Q) = 1; several pairs on one accumulator
    {
        const G2Prepared q1 = prepare_g2(Q), q2 = prepare_g2(mul_small(Q, 5));
        const Pair two[2] = {{P, &q1}, {affine_neg(P), &q1}};
        if (!is_one12(final_exp(multi_miller(two, 2)))) { printf("e(P, Q) e(-P, Q) != 1\n"); bad++; }
        const Pair three[3] = {{mul_small(P, 9), &q1}, {mul_small(P, 4), &q2}, {G1Affine::inf(), &q2}};
        const Fq12 together = final_exp(multi_miller(three, 3)), apart = final_exp(multi_miller(three, 1)) * final_exp(multi_miller(three + 1, 1));
        if (!eq12(together, apart) || !eq12(together, pow12(e, 29))) { printf("shared accumulator differs from separate loops\n"); bad++; }
    }
    // cyclotomic squaring
    {
        Fq12 f = miller(mul_small(P, 31), Q); f = conj12(f) * inv12(f); f = frobenius(f, 2) * f;
        for (int i = 0; i < 20; i++) { const Fq12 a = sqr12(f), b = cyclotomic_sqr(f); if (!eq12(a, b)) { printf("cyclotomic squaring differs at step %d\n", i); bad++; break; } f = a * e; }
    }
    // Frobenius: f^(q^k) composed = identity after 12 steps of k = 1; k = 2 is k = 1 twice; k = 3 is thrice
    {
        const Fq12 f = miller(mul_small(P, 3), Q);
        Fq12 g = f; for (int i = 0; i < 12; i++) g = frobenius(g, 1);
        if (!eq12(g, f) || !eq12(frobenius(frobenius(f, 1), 1), frobenius(f, 2)) || !eq12(frobenius(frobenius(f, 2), 1), frobenius(f, 3))) { printf("Frobenius maps inconsistent\n"); bad++; }
    }
    // subgroup membership: both tests on subgroup points and on twist points outside it
    {
        std::mt19937_64 g(5); int in = 0, out = 0;
        for (int i = 0; i < 12; i++) { const G2Affine R = mul_small(Q, g() | 1); if (!g2_in_subgroup(R) || !g2_in_subgroup_by_order(R)) { printf("a multiple of the generator is refused\n"); bad++; } else in++; }
        for (uint32_t x0 = 1; x0 < 400 && out < 40; x0++) {
            const Fq2 x{fp_from_u32<FqParams>(x0), fp_from_u32<FqParams>(1 + x0 % 7)}; Fq2 y;
            if (!sqrt2(fp_sqr(x) * x + consts().twist_b, y)) continue;
            const G2Affine R{x, y};
            if (!g2_on_curve(R)) { printf("square root is wrong\n"); bad++; break; }
            const bool a = g2_in_subgroup(R), b = g2_in_subgroup_by_order(R);
            if (a != b) { printf("the two membership tests disagree at x0 = %u (%d vs %d)\n", x0, (int)a, (int)b); bad++; }
            if (!b) out++;
        }
        if (out < 40) { printf("found only %d twist points outside the subgroup\n", out); bad++; }
        printf("subgroup tests: %d inside, %d outside\n", in, out);
    }
    // loop constant
    { const AteLoop& L = ate_loop(); __int128 v = 0; int nz = 0; for (int i = 64; i >= 0; i--) { v = 2 * v + L.digit[i]; nz += L.digit[i] != 0; }
      if (v != (((__int128)1 << 64) | 0x9d797039be763ba8ull) || L.digit[64] != 1) { printf("loop digits do not recompose\n"); bad++; } printf("ate loop: %d nonzero digits\n", nz); }
    printf("pairing host checks: %s\n", bad ? "MISMATCH" : "ok");
    return bad != 0;
}
