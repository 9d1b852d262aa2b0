// zkc_f29.h -- BN254 base/scalar field in nine unsaturated 29-bit limbs, for the inner loops that chain many products.
//
// zkc_field.h keeps field elements as 8 x u32 and converts to radix 2^29 inside every product: about a third of the
// 330 instructions of a product are that slicing, the repacking and the final conditional subtraction.  The bucket
// accumulation (zkc_msm.hip K5) performs ten products per point addition on values that never leave registers, so it
// keeps them in radix 2^29 throughout:
//
//   * value = sum l[i] 2^(29 i), nine u32 limbs, Montgomery form with R' = 2^261 (x~ = x 2^261 mod p), NOT reduced
//     below p: any representative below 2^261 = 169 p is legal.
//   * product: 81 + 81 v_mad_u64_u32 into 64-bit columns without carry handling, as fp_mul_r29.  The output has limbs
//     below 2^29 and is below a b / 2^261 + p, i.e. the product contracts magnitudes by 169: no conditional subtraction.
//   * addition is nine v_add_u32, subtraction adds a multiple of p whose limbs dominate the subtrahend's ("dominator").
//     Limbs may grow to 2^31 before the next product as long as max_limb(a) max_limb(b) < 2^60.6 (column sums stay
//     below 2^64); f29_carry() brings limbs back below 2^29 without touching the value.
//
// An 8 x u32 element in the usual R = 2^256 Montgomery form enters by slicing 32 x (its value), which IS its R' form, and
// leaves through an exact division by 32 (f29_to_fp).  Bounds for every formula are stated where it is used.
#pragma once
#include "zkc_field.h"

namespace zkc {

constexpr uint32_t F29_MASK = (1u << 29) - 1;
struct L9 { uint32_t l[9]; };

namespace f29_detail {
struct Big { uint64_t d[10]; };      // base-2^29 digits, normalised
constexpr Big big_norm(Big a) { uint64_t c = 0; for (int i = 0; i < 10; i++) { uint64_t v = a.d[i] + c; a.d[i] = v & F29_MASK; c = v >> 29; } return a; }
constexpr bool big_ge(const Big& a, const Big& b) { for (int i = 9; i >= 0; i--) { if (a.d[i] != b.d[i]) return a.d[i] > b.d[i]; } return true; }
constexpr Big big_sub(Big a, const Big& b) {       // a >= b
    int64_t br = 0;
    for (int i = 0; i < 10; i++) { int64_t v = (int64_t)a.d[i] - (int64_t)b.d[i] - br; br = v < 0; a.d[i] = (uint64_t)(v + (br ? (int64_t)1 << 29 : 0)); }
    return a;
}
constexpr Big big_shl1(Big a) { for (int i = 0; i < 10; i++) a.d[i] <<= 1; return big_norm(a); }
template <class P> constexpr Big big_p() { Big r{}; for (int k = 0; k < 9; k++) r.d[k] = P29<P>::limb(k); return r; }
template <class P> constexpr Big big_mod(Big a) {
    Big sh[40] = {}; sh[0] = big_p<P>(); int top = 0;
    for (int s = 1; s < 36; s++) { sh[s] = big_shl1(sh[s - 1]); top = s; }
    for (int s = top; s >= 0; s--) if (big_ge(a, sh[s])) a = big_sub(a, sh[s]);
    return a;
}
}  // namespace f29_detail

template <class P> constexpr L9 f29_p() { L9 r{}; for (int k = 0; k < 9; k++) r.l[k] = P29<P>::limb(k); return r; }
// a multiple of p with limbs 0..7 in [minlimb, minlimb + 2^29) and limb 8 in [toplimb, toplimb + 2^22]: a - b + D has no negative
// limb for every b whose limbs 0..7 are <= minlimb and whose limb 8 is <= toplimb
template <class P> constexpr L9 f29_dominator(uint32_t minlimb, uint32_t toplimb) {
    using namespace f29_detail;
    Big t{}; for (int i = 0; i < 8; i++) t.d[i] = minlimb; t.d[8] = toplimb;
    Big r = big_mod<P>(big_norm(t));
    bool zero = true; for (int i = 0; i < 10; i++) zero = zero && r.d[i] == 0;
    Big delta = zero ? r : big_sub(big_p<P>(), r);
    L9 o{}; for (int i = 0; i < 8; i++) o.l[i] = minlimb + (uint32_t)delta.d[i]; o.l[8] = toplimb + (uint32_t)delta.d[8];
    return o;
}
// 2^k mod p as normalised limbs (k < 290)
template <class P> constexpr L9 f29_pow2(int k) {
    using namespace f29_detail;
    Big t{}; t.d[k / 29] = 1ull << (k % 29);
    Big r = big_mod<P>(t);
    L9 o{}; for (int i = 0; i < 9; i++) o.l[i] = (uint32_t)r.d[i];
    return o;
}

template <class P> struct F29K {
    static constexpr L9 p = f29_p<P>();
    static constexpr L9 one = f29_pow2<P>(261);                                    // 1 in R' form
    static constexpr L9 dom1 = f29_dominator<P>(1u << 29, 1u << 27);               // >= any carried value below 2^259 (32 p)
    static constexpr L9 dom3 = f29_dominator<P>(3u << 29, 1u << 27);               // >= a + 2b for carried a, b below 2^258
    static constexpr uint32_t ninv = P::inv & F29_MASK;                            // -p^-1 mod 2^29
    static constexpr uint32_t pinv = (0u - P::inv) & F29_MASK;                     //  p^-1 mod 2^29
};

// limbs of 32 x (8 x u32 value): the R' = 2^261 Montgomery form of an element held in R = 2^256 form.  Limbs < 2^29, value < 32 p.
ZKC_HD void f29_from_fp_shl5(uint32_t r[9], const uint32_t a[8]) {
    r[0] = slice29<-5>(a); r[1] = slice29<24>(a); r[2] = slice29<53>(a); r[3] = slice29<82>(a); r[4] = slice29<111>(a);
    r[5] = slice29<140>(a); r[6] = slice29<169>(a); r[7] = slice29<198>(a); r[8] = slice29<227>(a);
}
// limbs < 2^29 (limb 8 takes what is left) without changing the value; input limbs < 2^32 - 2^3
ZKC_HD void f29_carry(uint32_t a[9]) {
#pragma unroll
    for (int k = 0; k < 8; k++) { a[k + 1] += a[k] >> 29; a[k] &= F29_MASK; }
}
ZKC_HD void f29_add(uint32_t r[9], const uint32_t a[9], const uint32_t b[9]) {
#pragma unroll
    for (int k = 0; k < 9; k++) r[k] = a[k] + b[k];
}
// r = a - b + D (D a dominator of b)
ZKC_HD void f29_sub(uint32_t r[9], const uint32_t a[9], const uint32_t b[9], const L9& D) {
#pragma unroll
    for (int k = 0; k < 9; k++) r[k] = a[k] + D.l[k] - b[k];
}

// Montgomery product a b / 2^261 mod p.  Requires max_limb(a) * max_limb(b) < 2^60.6.  Output limbs 0..7 < 2^29, value < a b / 2^261 + p.
template <class P>
ZKC_HD void f29_mul(uint32_t r[9], const uint32_t a[9], const uint32_t b[9]) {
    constexpr L9 Pl = F29K<P>::p;
    uint64_t c[18];
#pragma unroll
    for (int k = 0; k < 18; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)a[i] * b[j];
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        c[i] += carry;
        const uint32_t m = ((uint32_t)c[i] * F29K<P>::ninv) & F29_MASK;
#pragma unroll
        for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)m * Pl.l[j];
        carry = c[i] >> 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) { c[k] += carry; r[k - 9] = (uint32_t)c[k] & F29_MASK; carry = c[k] >> 29; }
    r[8] = (uint32_t)(c[17] + carry);
}
// a^2: 45 limb products instead of 81.  Requires max_limb(a) < 2^29.8 (the doubled operand stays below 2^31).
template <class P>
ZKC_HD void f29_sqr(uint32_t r[9], const uint32_t a[9]) {
    constexpr L9 Pl = F29K<P>::p;
    uint64_t c[18];
    uint32_t a2[9];
#pragma unroll
    for (int k = 0; k < 9; k++) a2[k] = a[k] << 1;
#pragma unroll
    for (int k = 0; k < 18; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        c[2 * i] += (uint64_t)a[i] * a[i];
#pragma unroll
        for (int j = i + 1; j < 9; j++) c[i + j] += (uint64_t)a2[i] * a[j];
    }
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        c[i] += carry;
        const uint32_t m = ((uint32_t)c[i] * F29K<P>::ninv) & F29_MASK;
#pragma unroll
        for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)m * Pl.l[j];
        carry = c[i] >> 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) { c[k] += carry; r[k - 9] = (uint32_t)c[k] & F29_MASK; carry = c[k] >> 29; }
    r[8] = (uint32_t)(c[17] + carry);
}

// ---- fused forms: the Montgomery reduction maps T -> T / 2^261, so a term h 2^261 added to the column sums before the reduction comes
// out as "+ h", and two products can share one reduction.  Each saves the separate add/subtract, its carry pass and (for the second
// product) 81 of the 162 mads. ----
template <class P>
ZKC_HD void f29_reduce_cols(uint32_t r[9], uint64_t c[18]) {
    constexpr L9 Pl = F29K<P>::p;
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        c[i] += carry;
        const uint32_t m = ((uint32_t)c[i] * F29K<P>::ninv) & F29_MASK;
#pragma unroll
        for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)m * Pl.l[j];
        carry = c[i] >> 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) { c[k] += carry; r[k - 9] = (uint32_t)c[k] & F29_MASK; carry = c[k] >> 29; }
    r[8] = (uint32_t)(c[17] + carry);
}
// a b / 2^261 + h  (h: any limbs below 2^32; same operand limits as f29_mul).  Output carried, below a b / 2^261 + h + p.
template <class P>
ZKC_HD void f29_mul_addhi(uint32_t r[9], const uint32_t a[9], const uint32_t b[9], const uint32_t h[9]) {
    uint64_t c[18];
#pragma unroll
    for (int k = 0; k < 9; k++) { c[k] = 0; c[9 + k] = h[k]; }
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)a[i] * b[j];
    f29_reduce_cols<P>(r, c);
}
template <class P>
ZKC_HD void f29_sqr_addhi(uint32_t r[9], const uint32_t a[9], const uint32_t h[9]) {
    uint64_t c[18]; uint32_t a2[9];
#pragma unroll
    for (int k = 0; k < 9; k++) { a2[k] = a[k] << 1; c[k] = 0; c[9 + k] = h[k]; }
#pragma unroll
    for (int i = 0; i < 9; i++) {
        c[2 * i] += (uint64_t)a[i] * a[i];
#pragma unroll
        for (int j = i + 1; j < 9; j++) c[i + j] += (uint64_t)a2[i] * a[j];
    }
    f29_reduce_cols<P>(r, c);
}
// (a1 b1 + a2 b2) / 2^261 [+ h].  Requires max_limb(a1) max_limb(b1) + max_limb(a2) max_limb(b2) < 2^60.5.
template <class P>
ZKC_HD void f29_mul2sum_addhi(uint32_t r[9], const uint32_t a1[9], const uint32_t b1[9], const uint32_t a2[9], const uint32_t b2[9], const uint32_t h[9]) {
    uint64_t c[18];
#pragma unroll
    for (int k = 0; k < 9; k++) { c[k] = 0; c[9 + k] = h[k]; }
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)a1[i] * b1[j] + (uint64_t)a2[i] * b2[j];
    f29_reduce_cols<P>(r, c);
}
template <class P>
ZKC_HD void f29_mul2sum(uint32_t r[9], const uint32_t a1[9], const uint32_t b1[9], const uint32_t a2[9], const uint32_t b2[9]) {
    uint64_t c[18];
#pragma unroll
    for (int k = 0; k < 18; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)a1[i] * b1[j] + (uint64_t)a2[i] * b2[j];
    f29_reduce_cols<P>(r, c);
}

// is the CARRIED value a (limbs 0..7 < 2^29, value < 64 p) a multiple of p?  If a = k p then k = a[0] p^-1 mod 2^29, so everything
// but one multiply, mask and compare runs with probability 2^-23.
template <class P>
ZKC_HD bool f29_is_zero_mod_p(const uint32_t a[9]) {
    const uint32_t k = (a[0] * F29K<P>::pinv) & F29_MASK;
    if (k >= 64) return false;
    constexpr L9 Pl = F29K<P>::p;
    uint64_t carry = 0; uint32_t diff = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint64_t v = (uint64_t)k * Pl.l[i] + carry;
        const uint32_t digit = i < 8 ? (uint32_t)v & F29_MASK : (uint32_t)v;
        carry = v >> 29; diff |= digit ^ a[i];
    }
    return diff == 0;
}

// carried value below 2^261  ->  carried value below 3 p, same residue: subtract q p with q = floor(top limb / (floor(p / 2^232) + 1)),
// which never exceeds floor(value / p) and falls short of it by at most 2
template <class P>
ZKC_HD void f29_reduce_small(uint32_t a[9]) {
    constexpr L9 Pl = F29K<P>::p;
    constexpr uint32_t ptop1 = (P::p[7] >> 8) + 1;
    constexpr uint32_t qmagic = (uint32_t)((1ull << 50) / ptop1);
    const uint32_t q = (uint32_t)(((uint64_t)a[8] * qmagic) >> 50);
    int64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int64_t v = (int64_t)a[k] - (int64_t)((uint64_t)q * Pl.l[k]) + carry;
        a[k] = k < 8 ? (uint32_t)v & F29_MASK : (uint32_t)v;
        carry = v >> 29;
    }
}

// bits [LO, LO+32) of the integer with normalised base-2^29 digits d[0..8]
template <int LO>
ZKC_HD uint32_t f29_word(const uint32_t d[9]) {
    constexpr int k = LO / 29, off = LO % 29;
    uint32_t w = d[k] >> off;
    if constexpr (k + 1 < 9) w |= d[k + 1] << (29 - off);
    if constexpr (k + 2 < 9 && 58 - off < 32) w |= d[k + 2] << (58 - off);
    return w;
}
// leave the R' domain: the canonical (< p) 8 x u32 element in R = 2^256 form, i.e. value / 32 mod p.  Input limbs < 2^31, value < 32 p.
template <class P>
ZKC_HD Fp<P> f29_to_fp(const uint32_t a[9]) {
    constexpr L9 Pl = F29K<P>::p;
    uint32_t t[9];
#pragma unroll
    for (int k = 0; k < 9; k++) t[k] = a[k];
    f29_carry(t);
    const uint32_t m = (t[0] * F29K<P>::ninv) & 31u;            // t + m p = 0 mod 32
    uint32_t d[9]; uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) { const uint64_t v = (uint64_t)m * Pl.l[k] + t[k] + carry; d[k] = k < 8 ? (uint32_t)v & F29_MASK : (uint32_t)v; carry = v >> 29; }
    Fp<P> r;                                                     // (t + m p) / 32 < p + p
    r.v[0] = f29_word<5>(d); r.v[1] = f29_word<37>(d); r.v[2] = f29_word<69>(d); r.v[3] = f29_word<101>(d);
    r.v[4] = f29_word<133>(d); r.v[5] = f29_word<165>(d); r.v[6] = f29_word<197>(d); r.v[7] = f29_word<229>(d);
    fp_reduce_once<P>(r.v);
    return r;
}

}  // namespace zkc
