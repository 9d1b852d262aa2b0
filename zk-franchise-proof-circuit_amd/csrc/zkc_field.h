// zkc_field.h -- BN254 prime-field arithmetic for the MI355X prover (product code, host + device).
//
// 8 x 32-bit little-endian limbs, Montgomery form with R = 2^256.  32-bit limbs are the native shape for
// CDNA4: the integer multiplier is 32x32 and `v_mad_u64_u32` folds the accumulate, so every inner step below is
// one mad plus a 64-bit carry add.  Fq is the curve base field, Fr the scalar field (witness, NTT, Poseidon).
// Replaces what the reference reaches inside ffjavascript/wasmcurves (ts_inputs/src/example.ts:358) and
// rapidsnark's Fr/Fq asm (zk_census_test.go:89).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define ZKC_HD __host__ __device__ __forceinline__

namespace zkc {

struct FqParams {
    static constexpr uint32_t p[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t r1[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t r2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
    static constexpr uint32_t inv = 0xe4866389u;   // -p^-1 mod 2^32
};
struct FrParams {
    static constexpr uint32_t p[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t r1[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t r2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
    static constexpr uint32_t inv = 0xefffffffu;
};

template <class P>
struct Fp {
    uint32_t v[8];

    ZKC_HD static Fp zero() { Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = 0; return r; }
    ZKC_HD static Fp one() { Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = P::r1[i]; return r; }
    ZKC_HD bool is_zero() const { uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= v[i]; return o == 0; }
    ZKC_HD bool operator==(const Fp& b) const { uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= v[i] ^ b.v[i]; return o == 0; }
    ZKC_HD bool operator!=(const Fp& b) const { return !(*this == b); }
};

// r = a - p if a >= p (a < 2p)
template <class P>
ZKC_HD void fp_reduce_once(uint32_t a[8]) {
    uint32_t t[8]; uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)a[i] - P::p[i] - br; t[i] = (uint32_t)d; br = (d >> 63) & 1; }
    if (!br) {
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = t[i];
    }
}
template <class P>
ZKC_HD Fp<P> operator+(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r; uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (uint64_t)a.v[i] + b.v[i]; r.v[i] = (uint32_t)c; c >>= 32; }
    fp_reduce_once<P>(r.v);      // p < 2^254 so a+b < 2^255: no carry out of limb 7
    return r;
}
template <class P>
ZKC_HD Fp<P> operator-(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r; uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)a.v[i] - b.v[i] - br; r.v[i] = (uint32_t)d; br = (d >> 63) & 1; }
    if (br) { uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { c += (uint64_t)r.v[i] + P::p[i]; r.v[i] = (uint32_t)c; c >>= 32; }
    }
    return r;
}
template <class P>
ZKC_HD Fp<P> fp_neg(const Fp<P>& a) { return a.is_zero() ? a : Fp<P>::zero() - a; }
template <class P>
ZKC_HD Fp<P> fp_dbl(const Fp<P>& a) { return a + a; }

// Montgomery product a*b/R mod p (CIOS over 32-bit limbs; p < 2^254 keeps every partial sum below 2^(256+32))
template <class P>
ZKC_HD void fp_mul_limbs(uint32_t r[8], const uint32_t a[8], const uint32_t b[8]) {
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
        const uint32_t bi = b[i];
#pragma unroll
        for (int j = 0; j < 8; j++) { c += (uint64_t)a[j] * bi + t[j]; t[j] = (uint32_t)c; c >>= 32; }
        c += t[8];                       // < 2^33
        const uint32_t m = t[0] * P::inv;
        uint64_t d = (uint64_t)m * P::p[0] + t[0]; d >>= 32;
#pragma unroll
        for (int j = 1; j < 8; j++) { d += (uint64_t)m * P::p[j] + t[j]; t[j - 1] = (uint32_t)d; d >>= 32; }
        d += c; t[7] = (uint32_t)d; t[8] = (uint32_t)(d >> 32);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = t[i];
    fp_reduce_once<P>(r);                // result < 2p and t[8] == 0 because 4p < R
}
// bits [lo, lo+29) of the 256-bit little-endian integer x, where lo may be negative (treated as zero bits) -- all constants after unrolling
template <int LO>
ZKC_HD uint32_t slice29(const uint32_t x[8]) {
    constexpr uint32_t MASK = (1u << 29) - 1;
    if constexpr (LO < 0) return (x[0] << (-LO)) & MASK;
    else {
        constexpr int w = LO >> 5, off = LO & 31;
        if constexpr (w >= 8) return 0;
        else if constexpr (off == 0) return x[w] & MASK;
        else if constexpr (w == 7) return (x[7] >> off) & MASK;
        else if constexpr (off + 29 <= 32) return (x[w] >> off) & MASK;
        else return ((x[w] >> off) | (x[w + 1] << (32 - off))) & MASK;
    }
}
template <class P> struct P29 {     // p in 9 x 29-bit limbs
    static constexpr uint32_t limb(int k) {
        uint64_t v = 0; int lo = 29 * k, w = lo >> 5, off = lo & 31;
        v = (uint64_t)P::p[w] >> off; if (w + 1 < 8) v |= (uint64_t)P::p[w + 1] << (32 - off);
        return (uint32_t)(v & ((1u << 29) - 1));
    }
};
// Montgomery product a*b/2^256 mod p computed in radix 2^29: 9 x 9 limb products accumulate in 64-bit columns with no carry
// handling at all (18 * 2^58 < 2^63); one operand is pre-multiplied by 2^5 so that reducing by R' = 2^261 lands on R = 2^256.
template <class P>
ZKC_HD void fp_mul_r29(uint32_t r[8], const uint32_t a[8], const uint32_t b[8]) {
    constexpr uint32_t MASK = (1u << 29) - 1;
    constexpr uint32_t INV29 = P::inv & MASK;
    const uint32_t A[9] = {slice29<-5>(a), slice29<24>(a), slice29<53>(a), slice29<82>(a), slice29<111>(a), slice29<140>(a), slice29<169>(a), slice29<198>(a), slice29<227>(a)};
    const uint32_t B[9] = {slice29<0>(b), slice29<29>(b), slice29<58>(b), slice29<87>(b), slice29<116>(b), slice29<145>(b), slice29<174>(b), slice29<203>(b), slice29<232>(b)};
    constexpr uint32_t Pl[9] = {P29<P>::limb(0), P29<P>::limb(1), P29<P>::limb(2), P29<P>::limb(3), P29<P>::limb(4), P29<P>::limb(5), P29<P>::limb(6), P29<P>::limb(7), P29<P>::limb(8)};
    uint64_t c[18];
#pragma unroll
    for (int k = 0; k < 18; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)A[i] * B[j];
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        c[i] += carry;
        const uint32_t m = ((uint32_t)c[i] * INV29) & MASK;
#pragma unroll
        for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)m * Pl[j];
        carry = c[i] >> 29;
    }
    uint32_t R[9];
#pragma unroll
    for (int k = 9; k < 18; k++) { c[k] += carry; R[k - 9] = (uint32_t)c[k] & MASK; carry = c[k] >> 29; }
    // repack 9 x 29 -> 8 x 32
    r[0] = R[0] | (R[1] << 29);
    r[1] = (R[1] >> 3) | (R[2] << 26);
    r[2] = (R[2] >> 6) | (R[3] << 23);
    r[3] = (R[3] >> 9) | (R[4] << 20);
    r[4] = (R[4] >> 12) | (R[5] << 17);
    r[5] = (R[5] >> 15) | (R[6] << 14);
    r[6] = (R[6] >> 18) | (R[7] << 11);
    r[7] = (R[7] >> 21) | (R[8] << 8);
    fp_reduce_once<P>(r);
}

// On the GPU the product is ONE out-of-line routine per field (about a thousand instructions), called with both operands
// and the result in VGPRs (native <8 x i32> vectors).  Inlining it into every group operation produced 200-300 KB
// straight-line kernels: far beyond the instruction cache, minutes of compile time, and on gfx950/ROCm 7.2 an
// out-of-line G2 addition built that way never terminated (tools/probe/, DESIGN.md "field multiplication").
typedef uint32_t zkc_u32x8 __attribute__((ext_vector_type(8)));
template <class P>
__device__ __noinline__ zkc_u32x8 fp_mul_dev(zkc_u32x8 a, zkc_u32x8 b) {
    uint32_t x[8], y[8], r[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { x[i] = a[i]; y[i] = b[i]; }
    fp_mul_r29<P>(r, x, y);          // radix-2^29 column accumulation: 162 mads + ~170 other instructions (CIOS over 32-bit limbs: 128 + ~530)
    zkc_u32x8 o;
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = r[i];
    return o;
}
template <class P>
__device__ __forceinline__ Fp<P> operator*(const Fp<P>& a, const Fp<P>& b) {
    zkc_u32x8 x, y;
#pragma unroll
    for (int i = 0; i < 8; i++) { x[i] = a.v[i]; y[i] = b.v[i]; }
    zkc_u32x8 o = fp_mul_dev<P>(x, y);
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = o[i];
    return r;
}
// host overload (setup, finalize, table generation, the CPU verifier): the same Montgomery product a*b/2^256 over 4 x 64-bit limbs with 128-bit partial products (CIOS).
// The eight 32-bit limbs of a little-endian host ARE the four 64-bit ones, so values and results are bit-identical to fp_mul_limbs -- at a third of its time
// (tests/test_host_field_cpu.py compares the two on random and extreme operands).
template <class P> struct P64 {
    static constexpr uint64_t limb(int k) { return (uint64_t)P::p[2 * k] | ((uint64_t)P::p[2 * k + 1] << 32); }
    static constexpr uint64_t inv() {                                // -p^-1 mod 2^64 from its low half: one Newton step on p^-1
        const uint64_t p0 = limb(0); uint64_t y = (uint64_t)(0u - P::inv);      // p^-1 mod 2^32
        y = y * (2 - p0 * y);                                        // mod 2^64
        return 0 - y;
    }
};
template <class P>
__host__ inline void fp_mul_host64(uint32_t r[8], const uint32_t a[8], const uint32_t b[8]) {
    typedef unsigned __int128 u128;
    constexpr uint64_t p0 = P64<P>::limb(0), p1 = P64<P>::limb(1), p2 = P64<P>::limb(2), p3 = P64<P>::limb(3), ninv = P64<P>::inv();
    uint64_t A[4], B[4]; __builtin_memcpy(A, a, 32); __builtin_memcpy(B, b, 32);
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    for (int i = 0; i < 4; i++) {
        const uint64_t bi = B[i];
        u128 c = (u128)A[0] * bi + t0; t0 = (uint64_t)c; c >>= 64;
        c += (u128)A[1] * bi + t1; t1 = (uint64_t)c; c >>= 64;
        c += (u128)A[2] * bi + t2; t2 = (uint64_t)c; c >>= 64;
        c += (u128)A[3] * bi + t3; t3 = (uint64_t)c; c >>= 64;
        c += t4; t4 = (uint64_t)c; const uint64_t t5 = (uint64_t)(c >> 64);
        const uint64_t m = t0 * ninv;
        c = (u128)m * p0 + t0; c >>= 64;
        c += (u128)m * p1 + t1; t0 = (uint64_t)c; c >>= 64;
        c += (u128)m * p2 + t2; t1 = (uint64_t)c; c >>= 64;
        c += (u128)m * p3 + t3; t2 = (uint64_t)c; c >>= 64;
        c += t4; t3 = (uint64_t)c; t4 = t5 + (uint64_t)(c >> 64);
    }
    // result < 2p (4p < 2^256 keeps t4 == 0): one conditional subtraction
    u128 d = (u128)t0 - p0; const uint64_t s0 = (uint64_t)d; uint64_t br = (uint64_t)(d >> 64) & 1;
    d = (u128)t1 - p1 - br; const uint64_t s1 = (uint64_t)d; br = (uint64_t)(d >> 64) & 1;
    d = (u128)t2 - p2 - br; const uint64_t s2 = (uint64_t)d; br = (uint64_t)(d >> 64) & 1;
    d = (u128)t3 - p3 - br; const uint64_t s3 = (uint64_t)d; br = (uint64_t)(d >> 64) & 1;
    const uint64_t o[4] = {br ? t0 : s0, br ? t1 : s1, br ? t2 : s2, br ? t3 : s3};
    __builtin_memcpy(r, o, 32);
}
template <class P>
__host__ inline Fp<P> operator*(const Fp<P>& a, const Fp<P>& b) { Fp<P> r; fp_mul_host64<P>(r.v, a.v, b.v); return r; }
template <class P>
ZKC_HD Fp<P> fp_sqr(const Fp<P>& a) { return a * a; }

template <class P>
ZKC_HD Fp<P> fp_from_std(const uint32_t s[8]) {        // standard -> Montgomery
    Fp<P> a, r2;
#pragma unroll
    for (int i = 0; i < 8; i++) { a.v[i] = s[i]; r2.v[i] = P::r2[i]; }
    return a * r2;
}
template <class P>
ZKC_HD void fp_to_std(uint32_t s[8], const Fp<P>& a) { // Montgomery -> standard
    Fp<P> one;
#pragma unroll
    for (int i = 0; i < 8; i++) one.v[i] = (i == 0);
    Fp<P> r = a * one;
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = r.v[i];
}
template <class P>
ZKC_HD Fp<P> fp_from_u32(uint32_t x) { uint32_t s[8] = {x, 0, 0, 0, 0, 0, 0, 0}; return fp_from_std<P>(s); }

// a^(p-2) by square-and-multiply (0 -> 0)
template <class P>
ZKC_HD Fp<P> fp_inv(const Fp<P>& a) {
    uint32_t e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = P::p[i] - (i == 0 ? 2u : 0u);   // p[0] >= 2 for both moduli
    Fp<P> r = Fp<P>::one();
    for (int k = 255; k >= 0; k--) {
        r = r * r;
        uint32_t w = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) w = (i == (k >> 5)) ? e[i] : w;
        if ((w >> (k & 31)) & 1) r = r * a;
    }
    return r;
}
// 1 / a by the binary extended Euclid algorithm (0 -> 0): about 760 rounds of 8-limb shifts, additions and subtractions -- a quarter of the instructions of
// the 254 squarings + ~127 products of fp_inv.  Data-dependent control flow: for ONE lane (or a few) at the end of a latency chain (the blinding of a lone
// proof, zkc_finalize.hip), not for a full wave of different values.  Works on the stored residue A = a R: A^-1 = a^-1 R^-1, and a^-1 R = A^-1 R^3 / R.
template <class P>
ZKC_HD Fp<P> fp_inv_gcd(const Fp<P>& a) {
    if (a.is_zero()) return a;
    uint32_t u[8], v[8], x1[8], x2[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { u[i] = a.v[i]; v[i] = P::p[i]; x1[i] = i == 0 ? 1u : 0u; x2[i] = 0; }
    auto is_one = [](const uint32_t* t) { uint32_t o = t[0] ^ 1u;
#pragma unroll
        for (int i = 1; i < 8; i++) o |= t[i]; return o == 0; };
    auto halve = [](uint32_t* t, uint32_t* x) {                      // t even: t /= 2, x /= 2 mod p
        for (int i = 0; i < 7; i++) t[i] = (t[i] >> 1) | (t[i + 1] << 31);
        t[7] >>= 1;
        uint64_t c = 0; const uint32_t odd = 0u - (x[0] & 1u);       // x odd: x + p is even and below 2^255
#pragma unroll
        for (int i = 0; i < 8; i++) { c += (uint64_t)x[i] + (P::p[i] & odd); x[i] = (uint32_t)c; c >>= 32; }
#pragma unroll
        for (int i = 0; i < 7; i++) x[i] = (x[i] >> 1) | (x[i + 1] << 31);
        x[7] >>= 1;
    };
    auto sub_from = [](uint32_t* t, const uint32_t* w, uint32_t* x, const uint32_t* y) {      // t -= w (t >= w), x = x - y mod p
        uint64_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { const uint64_t d = (uint64_t)t[i] - w[i] - br; t[i] = (uint32_t)d; br = (d >> 63) & 1; }
        br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { const uint64_t d = (uint64_t)x[i] - y[i] - br; x[i] = (uint32_t)d; br = (d >> 63) & 1; }
        const uint32_t neg = 0u - (uint32_t)br; uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { c += (uint64_t)x[i] + (P::p[i] & neg); x[i] = (uint32_t)c; c >>= 32; }
    };
    while (!is_one(u) && !is_one(v)) {
        while (!(u[0] & 1u)) halve(u, x1);
        while (!(v[0] & 1u)) halve(v, x2);
        bool ge = true;                                              // u >= v ?
        for (int i = 7; i >= 0; i--) if (u[i] != v[i]) { ge = u[i] > v[i]; break; }
        if (ge) sub_from(u, v, x1, x2); else sub_from(v, u, x2, x1);
    }
    Fp<P> w, r2;
#pragma unroll
    for (int i = 0; i < 8; i++) { w.v[i] = is_one(u) ? x1[i] : x2[i]; r2.v[i] = P::r2[i]; }
    return w * (r2 * r2);
}
template <class P>
ZKC_HD bool fp_std_lt_p(const uint32_t s[8]) {       // s < p ?
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)s[i] - P::p[i] - br; br = (d >> 63) & 1; }
    return br != 0;
}

using Fq = Fp<FqParams>;
using Fr = Fp<FrParams>;

}  // namespace zkc
