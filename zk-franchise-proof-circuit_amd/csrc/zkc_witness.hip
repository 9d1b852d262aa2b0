// zkc_witness.hip -- K2: batched zkCensus witness generation on MI355X (product code).
//
// Replaces the reference's witness stage: circom_runtime + circuit.wasm inside snarkjs `groth16.fullProve`
// (ts_inputs/src/example.ts:358-362) and go-rapidsnark/witness + wasmer (zk_census_test.go:89).  The spec is
// circuit/census.circom:49-115 with the circomlib 2.0.5 templates; the output is the circom 2.1.5 wire order of
// artifacts/zkCensus/dev/160/circuit.wasm (see DESIGN.md "witness layout").
//
// Two kernels:
//   zkc_witness_fill     wide, coalesced: every voter's witness := the voter-independent template witness
//                        (empty-subtree Poseidon(0,0) level traces, oldKey=0 bit decomposition, ...)
//   zkc_witness_chains   one lane per (voter, chain) with chain in {census tree, sik tree, misc}: walks the
//                        Merkle path leaf->root (Poseidon is sequential along a path), and stores every surviving
//                        signal of the non-empty levels straight into its wire slot (Montgomery form, marked).
//   zkc_witness_chains_wave  [r2] the same with one WAVE per chain (S-boxes and mix rows of a round dealt over the lanes): half the
//                        latency for 64 times the issue slots -- used for small batches and for the first pass of a large one
//   zkc_witness_tostd    wide: converts the marked wires to standard form
// HBM layout: inputs  [B][nInputs][8 x u32]  standard form, census.circom declaration order
//             witness [B][nWires ][8 x u32]  standard form (what .wtns section 2 holds and what MSM digits read)
#include "zkc_field.h"
#include "zkc_device.h"
#include "zkc_f29.h"

namespace zkc {

struct Emit {                       // writes Montgomery values as standard-form wires
    uint32_t* base;                 // witness of this voter
    bool lead = true;               // [r2] a chain is walked by a whole wave: the chain-level stores below are made by its first lane only
    // The conversion out of Montgomery form is a product that nothing downstream in the chain waits for: the chain kernel is one long
    // dependency chain per lane, so it stores the Montgomery limbs with bit 255 set (values are below r < 2^254) and zkc_witness_tostd
    // converts all marked wires afterwards, fully parallel.  (239 of the ~840 products of a Poseidon(2) level were these conversions.)
    __device__ __forceinline__ void put(int wire, const Fr& v) const { if (lead) put_any(wire, v); }
    __device__ __forceinline__ void put_any(int wire, const Fr& v) const {      // from whichever lane holds the value (wave-wide Poseidon)
        uint4* d = reinterpret_cast<uint4*>(base + 8 * (size_t)wire);
        d[0] = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]); d[1] = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7] | 0x80000000u);
    }
    __device__ __forceinline__ void put_std(int wire, const uint32_t s[8]) const {
        if (!lead) return;
        uint4* d = reinterpret_cast<uint4*>(base + 8 * (size_t)wire);
        d[0] = make_uint4(s[0], s[1], s[2], s[3]); d[1] = make_uint4(s[4], s[5], s[6], s[7]);
    }
    __device__ __forceinline__ void put_small(int wire, uint32_t x) const {
        if (!lead) return;
        uint4* d = reinterpret_cast<uint4*>(base + 8 * (size_t)wire);
        d[0] = make_uint4(x, 0, 0, 0); d[1] = make_uint4(0, 0, 0, 0);
    }
    __device__ __forceinline__ void put_raw(int wire, const Fr& mont) const {      // scratch use of a wire slot
        if (!lead) return;
        uint4* d = reinterpret_cast<uint4*>(base + 8 * (size_t)wire);
        d[0] = make_uint4(mont.v[0], mont.v[1], mont.v[2], mont.v[3]); d[1] = make_uint4(mont.v[4], mont.v[5], mont.v[6], mont.v[7]);
    }
    __device__ __forceinline__ Fr get_raw(int wire) const {
        const uint4* d = reinterpret_cast<const uint4*>(base + 8 * (size_t)wire);
        uint4 a = d[0], b = d[1]; Fr r; r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w; return r;
    }
};

// Optimised Poseidon (circomlib 2.0.5 schedule) over t = T state words, storing the surviving trace signals.
// LAYOUT 0: the survivor set of every t=3 / t=4 instance; LAYOUT 1: the single t=5 instance (computedNullifier).
// cmask bit j = state word j is a compile-time constant at round 0 (bit 0 always: initialState = 0).
// `blk` = wire index of the first internal signal of this Poseidon block; emit.base == nullptr -> hash only.
// Arithmetic in radix 2^29 (zkc_f29.h): the chain kernel is one dependency chain per lane, so what counts is the latency of a
// product (inline 29-bit-limb code exposes the independent mads of a product to the scheduler; the out-of-line 8 x u32 product does not)
// and the number of reductions: a mix row is T products with ONE reduction, "state += in0 * S" is a product with the addend folded in.
// Magnitudes (multiples of r): constants < 1.2 (table base29); S-box outputs < 3; the partial rounds let state[1..] grow by about one
// per round (< 64 after 60 rounds, capacity 169), everything else passes through a product again and shrinks.
struct W29 { uint32_t l[9]; };
__device__ __forceinline__ W29 ld_c29(const PoseidonTable& tab, const Fr* c) {
    const uint4* q = reinterpret_cast<const uint4*>(tab.base29 + 12 * (size_t)(c - tab.base)); const uint4 a = q[0], b = q[1], d = q[2];
    return W29{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x}};
}
__device__ __forceinline__ void emit29(const Emit& e, int wire, const W29& v) {
    uint32_t t[9];
#pragma unroll
    for (int k = 0; k < 9; k++) t[k] = v.l[k];
    f29_reduce_small<FrParams>(t);
    e.put(wire, f29_to_fp<FrParams>(t));
}
__device__ __forceinline__ W29 sbox29(const W29& x, W29& in2, W29& in4) {
    W29 o; f29_sqr<FrParams>(in2.l, x.l); f29_sqr<FrParams>(in4.l, in2.l); f29_mul<FrParams>(o.l, in4.l, x.l); return o;
}
// sum_j a[j] * c_j / 2^261 with one reduction (T <= 5 carried operands: 5 * 9 * 2^58 + 9 * 2^58 < 2^64)
template <int T, class GetC>
__device__ __forceinline__ W29 dot29(const W29* a, GetC getc) {
    uint64_t col[18];
#pragma unroll
    for (int k = 0; k < 18; k++) col[k] = 0;
#pragma unroll
    for (int j = 0; j < T; j++) {
        const W29 c = getc(j);
#pragma unroll
        for (int x = 0; x < 9; x++)
#pragma unroll
            for (int y = 0; y < 9; y++) col[x + y] += (uint64_t)a[j].l[x] * c.l[y];
    }
    W29 r; f29_reduce_cols<FrParams>(r.l, col); return r;
}
__device__ __forceinline__ W29 add29(const W29& a, const W29& b) { W29 r; f29_add(r.l, a.l, b.l); f29_carry(r.l); return r; }

template <int T, int LAYOUT>
__device__ Fr poseidon_trace29(const Fr* in, unsigned cmask, const PoseidonTable& tab, const Emit& e, int blk) {
    constexpr int RP = (T == 3) ? 57 : (T == 4) ? 56 : 60;
    const Fr* __restrict__ C = tab.C[T]; const Fr* __restrict__ S = tab.S[T];
    const Fr* __restrict__ M = tab.M[T]; const Fr* __restrict__ Pm = tab.P[T];
    const bool on = e.base != nullptr;
    int rank[T]; int nc1 = 0;
#pragma unroll
    for (int j = 0; j < T; j++) { rank[j] = nc1; nc1 += !((cmask >> j) & 1); }
    const int nA = nc1 + 6 * T, oLast = nA, oMS = nA + T - 1, oF = oMS + RP, oP = oF + 2 * (nc1 + 7 * T);
    auto ark_idx = [&](int r, int j) -> int {
        if (LAYOUT == 0) return r == 1 ? (((cmask >> j) & 1) ? -1 : rank[j]) : nc1 + (r - 2) * T + j;
        if (r == 1) return j == 0 ? -1 : j - 1;
        if (r <= 3) return 4 + (r - 2) * 5 + j;
        if (r == 4) return j == 0 ? 14 : -1;
        return 15 + (r - 5) * 5 + j;
    };
    auto sF_idx = [&](int r, int j) -> int {
        if (LAYOUT == 0) return oF + 2 * (r == 0 ? (((cmask >> j) & 1) ? -1000 : rank[j]) : nc1 + (r - 1) * T + j);
        return 98 + 2 * (r == 0 ? (j == 0 ? -1000 : j - 1) : 4 + (r - 1) * 5 + j);
    };
    const int oPp = (LAYOUT == 0) ? oP : 176;

    W29 st[T];
#pragma unroll
    for (int k = 0; k < 9; k++) st[0].l[k] = 0;
#pragma unroll
    for (int j = 1; j < T; j++) f29_from_fp_shl5(st[j].l, in[j - 1].v);          // 32 x value: below 32
#pragma unroll
    for (int j = 0; j < T; j++) st[j] = add29(st[j], ld_c29(tab, C + j));
    for (int r = 0; r < 4; r++) {                                                  // first half of the full rounds
        W29 ns[T];
#pragma unroll
        for (int j = 0; j < T; j++) {
            W29 i2, i4; const W29 o = sbox29(st[j], i2, i4);
            const int fi = sF_idx(r, j);
            if (on && fi >= 0) { emit29(e, blk + fi, i2); emit29(e, blk + fi + 1, i4); }
            ns[j] = add29(o, ld_c29(tab, C + (r + 1) * T + j));
            const int ai = ark_idx(r + 1, j);
            if (on && ai >= 0) emit29(e, blk + ai, ns[j]);
        }
        const Fr* __restrict__ MM = (r < 3) ? M : Pm;
#pragma unroll
        for (int i = 0; i < T; i++) st[i] = dot29<T>(ns, [&](int j) { return ld_c29(tab, MM + j * T + i); });
    }
    if constexpr (LAYOUT == 1) { if (on) emit29(e, blk + 30, st[4]); }                    // mix[3].out[4]
    for (int r = 0; r < RP; r++) {                                                 // partial rounds
        const Fr* __restrict__ Sr = S + (2 * T - 1) * r;
        W29 i2, i4; const W29 o = sbox29(st[0], i2, i4);
        if (on) { emit29(e, blk + oPp + 2 * r, i2); emit29(e, blk + oPp + 2 * r + 1, i4); }
        st[0] = add29(o, ld_c29(tab, C + 5 * T + r));                              // in0
        if (LAYOUT == 1 && on && r == 59) emit29(e, blk + 97, st[0]);             // mixS[59].in[0]
        const W29 n0 = dot29<T>(st, [&](int j) { return ld_c29(tab, Sr + j); });
#pragma unroll
        for (int i = 1; i < T; i++) { const W29 c = ld_c29(tab, Sr + T + i - 1); W29 t; f29_mul_addhi<FrParams>(t.l, st[0].l, c.l, st[i].l); st[i] = t; }
        st[0] = n0;
        if (on) {
            if constexpr (LAYOUT == 0) emit29(e, blk + oMS + r, st[0]);
            else {
                if (r <= 56) emit29(e, blk + 35 + r, st[4]);
                else if (r == 57) { emit29(e, blk + 92, st[1]); emit29(e, blk + 93, st[2]); emit29(e, blk + 94, st[3]); emit29(e, blk + 95, st[4]); }
                else if (r == 58) emit29(e, blk + 96, st[4]);
            }
        }
    }
    for (int r = 0; r < 3; r++) {                                                  // second half of the full rounds
        W29 ns[T];
#pragma unroll
        for (int j = 0; j < T; j++) {
            W29 i2, i4; const W29 o = sbox29(st[j], i2, i4);
            const int fi = sF_idx(4 + r, j);
            if (on) { emit29(e, blk + fi, i2); emit29(e, blk + fi + 1, i4); }
            ns[j] = add29(o, ld_c29(tab, C + 5 * T + RP + r * T + j));
            if (on) emit29(e, blk + ark_idx(5 + r, j), ns[j]);
        }
#pragma unroll
        for (int i = 0; i < T; i++) st[i] = dot29<T>(ns, [&](int j) { return ld_c29(tab, M + j * T + i); });
    }
    W29 os[T];                                                                     // sigmaF[7] and mixLast
    const int oL = (LAYOUT == 0) ? oLast : 31;
#pragma unroll
    for (int j = 0; j < T; j++) {
        W29 i2, i4; os[j] = sbox29(st[j], i2, i4);
        const int fi = sF_idx(7, j);
        if (on) { emit29(e, blk + fi, i2); emit29(e, blk + fi + 1, i4); if (j < T - 1) emit29(e, blk + oL + j, os[j]); }
    }
    W29 out = dot29<T>(os, [&](int j) { return ld_c29(tab, M + j * T); });
    f29_reduce_small<FrParams>(out.l);
    return f29_to_fp<FrParams>(out.l);
}

// ---- [r2] the same permutation (survivor layout 0, t = 3 or 4) walked by a whole WAVE: every lane calls it with the same arguments ----
// The state is kept replicated in all lanes; what is expensive is dealt out and gathered back with v_readlane:
//   full round    lane j < T : S-box of word j, its trace stores, + round constant; gather; lane i < T : row i of the mix; gather
//   partial round every lane : the one S-box (3 products in a row -- lanes of a wave share their instruction stream, so this chain cannot be
//                 split); then lane L < 2T-1 ONE product: L < T a term of the new word 0, L >= T the update of word L-T+1; gather
// The three trace values of partial round r are parked in lane r and stored once, 57 lanes side by side, after the last partial round.
// A partial round is about 1100 instructions instead of 1900 (S-box 650, one product 250, selects and 45 v_readlane), a full round 1450
// instead of 4100.  Values are congruent to, not identical with, the single-lane routine's (separate reductions); the wires are canonical.
// PIN = true: an empty asm pins every gathered limb in a VGPR and hides from the compiler that it is wave-uniform.  Round 2 needed it: word 0 of the state was uniform
// through its whole S-box and hipcc moved that S-box to the SCALAR unit -- 81 limb products as s_mul_i32 / s_mul_hi_u32 / s_add_u32 / s_addc_u32 quadruples, 1850 scalar
// instructions per partial round where the vector unit needs 650.  [r3] With three products per round every product has a per-lane operand (lanes play different roles), nothing
// of that size is uniform any more, and the gathered limbs may stay in SGPRs as operands: 54 v_mov and 37 s_nop less per partial round (949 -> 870 instructions), the carry of the
// uniform word 0 on the scalar unit.  The ISA holds 11 s_mul in all; tools/single_proof_trace.py: witness chain 2.00 -> 1.90 ms.
template <bool PIN = false>
__device__ __forceinline__ W29 bcast29(const W29& v, int src) {
    W29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) { uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)v.l[k], src); if (PIN) asm volatile("" : "+v"(x)); r.l[k] = x; }
    return r;
}
__device__ __forceinline__ void emit29_any(const Emit& e, int wire, const W29& v) {
    uint32_t t[9];
#pragma unroll
    for (int k = 0; k < 9; k++) t[k] = v.l[k];
    f29_reduce_small<FrParams>(t);
    e.put_any(wire, f29_to_fp<FrParams>(t));
}
template <int T>
__device__ Fr poseidon_wave29(const Fr* in, unsigned cmask, const PoseidonTable& tab, const Emit& e, int blk) {
    static_assert(T == 3 || T == 4, "survivor layout 0 only");
    constexpr int RP = (T == 3) ? 57 : 56;
    const int lane = threadIdx.x & 63;
    const Fr* __restrict__ C = tab.C[T]; const Fr* __restrict__ S = tab.S[T];
    const Fr* __restrict__ M = tab.M[T]; const Fr* __restrict__ Pm = tab.P[T];
    const bool on = e.base != nullptr;
    int nc1 = 0;
#pragma unroll
    for (int j = 0; j < T; j++) nc1 += !((cmask >> j) & 1);
    const int nA = nc1 + 6 * T, oLast = nA, oMS = nA + T - 1, oF = oMS + RP, oP = oF + 2 * (nc1 + 7 * T);
    auto rank_of = [&](int j) { return (int)__popc(~cmask & ((1u << j) - 1u)); };
    auto ark_idx = [&](int r, int j) -> int { return r == 1 ? (((cmask >> j) & 1) ? -1 : rank_of(j)) : nc1 + (r - 2) * T + j; };
    auto sF_idx = [&](int r, int j) -> int { return oF + 2 * (r == 0 ? (((cmask >> j) & 1) ? -1000 : rank_of(j)) : nc1 + (r - 1) * T + j); };
    const int jl = lane < T ? lane : T - 1;                 // the word this lane serves in a full round (lanes >= T shadow the last one; never stored)
    const bool serve = lane < T;
    auto sel = [&](const W29* st, int idx) { W29 r = st[0];
#pragma unroll
        for (int j = 1; j < T; j++) if (idx == j) r = st[j];
        return r; };
    W29 st[T];
#pragma unroll
    for (int k = 0; k < 9; k++) st[0].l[k] = 0;
#pragma unroll
    for (int j = 1; j < T; j++) f29_from_fp_shl5(st[j].l, in[j - 1].v);
#pragma unroll
    for (int j = 0; j < T; j++) st[j] = add29(st[j], ld_c29(tab, C + j));
    auto full_round = [&](int r_sf, int r_ark, const Fr* cst, const Fr* MM) {       // S-boxes dealt over the lanes, mix rows dealt over the lanes
        W29 i2, i4; const W29 o = sbox29(sel(st, jl), i2, i4);
        const int fi = sF_idx(r_sf, jl);
        if (on && serve && fi >= 0) { emit29_any(e, blk + fi, i2); emit29_any(e, blk + fi + 1, i4); }
        const W29 nsl = add29(o, ld_c29(tab, cst + jl));
        const int ai = ark_idx(r_ark, jl);
        if (on && serve && ai >= 0) emit29_any(e, blk + ai, nsl);
        W29 ns[T];
#pragma unroll
        for (int j = 0; j < T; j++) ns[j] = bcast29(nsl, j);
        const W29 row = dot29<T>(ns, [&](int j) { return ld_c29(tab, MM + j * T + jl); });
#pragma unroll
        for (int i = 0; i < T; i++) st[i] = bcast29(row, i);
    };
    for (int r = 0; r < 4; r++) full_round(r, r + 1, C + (r + 1) * T, r < 3 ? M : Pm);
    W29 E1, E2, E3;
#pragma unroll
    for (int k = 0; k < 9; k++) E1.l[k] = E2.l[k] = E3.l[k] = 0;
    // [r3] partial rounds, three products per round instead of four.  With x = word 0 and c its round constant the round needs
    //   new word 0 = S[0] (x^5 + c) + sum_j S[j] word_j ,   new word i = word_i + S[T + i - 1] (x^5 + c)
    // and every lane runs the same three products on its own operands (a wave has ONE instruction stream, so what counts is the number of products in a row):
    //   A: lanes >= 2T-1: x x (= in2) | lane 0: S[0] x | lanes 1..T-1: S[j] word_j | lanes T..2T-2: S[T+i-1] x
    //   B: the square of A: in4 in the lanes >= 2T-1, gathered from lane 63
    //   C: A in4 + h with h = S[.] c (table K29) in lane 0 and S[.] c + word_i in lanes T..2T-2: the new word-0 term and the new words 1..T-1
    // instead of x^2, x^4, x^5 and then one product per lane: 162 + 126 + 162 multiply-adds instead of 126 + 126 + 162 + 162.  The three trace values of
    // round r are parked in lane 2T-1+r (in2 only exists in those lanes) and stored once after the last round.
    constexpr int CH0 = 2 * T - 1;                          // first lane that walks the S-box chain
    const bool chain = lane >= CH0;
    const uint32_t* __restrict__ K29 = tab.K29[T];
    for (int r = 0; r < RP; r++) {
        const Fr* __restrict__ Sr = S + (2 * T - 1) * r;
        const W29 c = ld_c29(tab, Sr + (chain ? 0 : lane));
        W29 Xa = st[0], Ya = chain ? st[0] : c;
#pragma unroll
        for (int j = 1; j < T; j++) if (lane == j) Xa = st[j];
        W29 A; f29_mul<FrParams>(A.l, Xa.l, Ya.l);
        W29 B; f29_sqr<FrParams>(B.l, A.l);
        const W29 x4 = bcast29<false>(B, 63);
        W29 H;
        { const int ki = lane == 0 ? 0 : (lane >= T && lane < CH0 ? lane - T + 1 : 0);
          const uint4* q = reinterpret_cast<const uint4*>(K29 + 12 * (size_t)(r * T + ki)); const uint4 a = q[0], b = q[1], d = q[2];
          H = W29{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x}}; }
#pragma unroll
        for (int i2 = 1; i2 < T; i2++) if (lane == T + i2 - 1) {
#pragma unroll
            for (int k = 0; k < 9; k++) H.l[k] += st[i2].l[k];
        }
        W29 Pd; f29_mul_addhi<FrParams>(Pd.l, A.l, x4.l, H.l);
        W29 n0 = bcast29<false>(Pd, 0);
#pragma unroll
        for (int j = 1; j < T; j++) { const W29 t = bcast29<false>(A, j);
#pragma unroll
            for (int k = 0; k < 9; k++) n0.l[k] += t.l[k]; }
        f29_carry(n0.l);
#pragma unroll
        for (int i2 = 1; i2 < T; i2++) st[i2] = bcast29<false>(Pd, T + i2 - 1);
        st[0] = n0;
        if (lane == CH0 + r) { E1 = A; E2 = x4; E3 = n0; }
    }
    if (on && lane >= CH0 && lane < CH0 + RP) { const int r = lane - CH0; emit29_any(e, blk + oP + 2 * r, E1); emit29_any(e, blk + oP + 2 * r + 1, E2); emit29_any(e, blk + oMS + r, E3); }
    for (int r = 0; r < 3; r++) full_round(4 + r, 5 + r, C + 5 * T + RP + r * T, M);
    W29 i2, i4; const W29 osl = sbox29(sel(st, jl), i2, i4);                       // sigmaF[7] and mixLast
    {
        const int fi = sF_idx(7, jl);
        if (on && serve) { emit29_any(e, blk + fi, i2); emit29_any(e, blk + fi + 1, i4); if (jl < T - 1) emit29_any(e, blk + oLast + jl, osl); }
    }
    W29 os[T];
#pragma unroll
    for (int j = 0; j < T; j++) os[j] = bcast29(osl, j);
    W29 out = dot29<T>(os, [&](int j) { return ld_c29(tab, M + j * T); });
    f29_reduce_small<FrParams>(out.l);
    return f29_to_fp<FrParams>(out.l);
}

__device__ __forceinline__ Fr load_std(const uint32_t* p) {
    const uint4* d = reinterpret_cast<const uint4*>(p); uint4 a = d[0], b = d[1];
    uint32_t s[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return fp_from_std<FrParams>(s);
}
__device__ __forceinline__ void load_raw(uint32_t s[8], const uint32_t* p) {
    const uint4* d = reinterpret_cast<const uint4*>(p); uint4 a = d[0], b = d[1];
    s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
}
__device__ __forceinline__ bool raw_is_zero(const uint32_t* p) {
    const uint4* d = reinterpret_cast<const uint4*>(p); uint4 a = d[0], b = d[1];
    return (a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w) == 0;
}
__device__ __forceinline__ int bit_of(const uint32_t s[8], int i) {
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) w = (k == (i >> 5)) ? s[k] : w;
    return (w >> (i & 31)) & 1;
}

// CompConstant(r-1) over the bits of key `ks` (circomlib compconstant.circom): parts[0..126] and the surviving bits
// of Num2Bits(135)(sum parts): out[0..126], out[128..133].  Everything is a small integer (< 2^135): plain limbs.
__device__ void emit_alias_check(const Emit& e, int wire, const uint32_t ks[8]) {
    uint32_t ct[8];
#pragma unroll
    for (int i = 0; i < 8; i++) ct[i] = FrParams::p[i];
    ct[0] -= 1;
    uint32_t sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 127; i++) {
        int c = bit_of(ct, 2 * i) | (bit_of(ct, 2 * i + 1) << 1), v = bit_of(ks, 2 * i) | (bit_of(ks, 2 * i + 1) << 1);
        uint32_t part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (v < c) {                                   // a = 2^i
#pragma unroll
            for (int k = 0; k < 8; k++) part[k] = (k == (i >> 5)) ? (1u << (i & 31)) : 0u;
        } else if (v > c) {                            // b = 2^128 - 2^i : bits i..127 set
#pragma unroll
            for (int k = 0; k < 4; k++) part[k] = (k > (i >> 5)) ? 0xffffffffu : (k == (i >> 5)) ? (0xffffffffu << (i & 31)) : 0u;
        }
        e.put_std(wire + i, part);
        uint64_t cy = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { cy += (uint64_t)sum[k] + part[k]; sum[k] = (uint32_t)cy; cy >>= 32; }
    }
    int w = wire + 127;
    for (int i = 0; i < 134; i++) if (i != 127) e.put_small(w++, (uint32_t)bit_of(sum, i));
}

// SMTVerifier(n) with enabled=1, fnc=0, old*=0 (census.circom:79-103): all non-template wires of one verifier block.
// Returns the recomputed root; *bad_last = siblings[n-1] != 0.
template <bool WAVE>
__device__ Fr smt_verifier_chain(const WitnessLayout& L, const PoseidonTable& tab, const Emit& e, int blk,
                                 const uint32_t key_s[8], const Fr& key, const Fr& value, const uint32_t* sib /* n x 8, std */,
                                 bool* bad_last, bool tmpl_mode) {
    const int n = L.n;
    // depth d = 1 + last non-zero sibling among 0..n-2
    int d = 0;
    for (int i = 0; i <= n - 2; i++) if (!raw_is_zero(sib + 8 * i)) d = i + 1;
    *bad_last = !raw_is_zero(sib + 8 * (n - 1));
    const bool key0 = key.is_zero();
    e.put_small(blk + 0, key0 ? 1u : 0u);                               // areKeyEquals.out
    e.put_small(blk + 2, 0u);                                           // checkRoot.isz.inv
    Fr hin[3] = {key, value, Fr::one()};
    Fr h1new = WAVE ? poseidon_wave29<4>(hin, 1u | 8u, tab, e, blk + 4) : poseidon_trace29<4, 0>(hin, 1u | 8u, tab, e, blk + 4);
    e.put(blk + 3, h1new);
    // levels d-1 .. 0 hold real hashes; levels >= d keep the template's Poseidon(0,0) trace
    Fr child = h1new;
    for (int i = d - 1; i >= 0; i--) {
        const int lb = blk + L.lvl_off(i);
        int o = 0;
        if (i == n - 3) e.put_small(lb + o++, 1u);                      // st_top[n-3] = 1 (i < d)
        if (i > 0 && i < n - 2) e.put_small(lb + o++, 0u);              // st_inew[i] = 0
        const int bit = bit_of(key_s, i);
        e.put_small(lb + o++, (uint32_t)bit);                           // lrbit
        e.put(lb + o++, child);                                         // child = root of level i+1
        Fr s = load_std(sib + 8 * i);
        Fr in2[2];
        if (bit) { in2[0] = s; in2[1] = child; } else { in2[0] = child; in2[1] = s; }   // Switcher
        Fr h = WAVE ? poseidon_wave29<3>(in2, 1u, tab, e, lb + o + 3) : poseidon_trace29<3, 0>(in2, 1u, tab, e, lb + o + 3);
        e.put(lb + o, h);                                               // aux[0] = h * st_top (st_top = 1)
        e.put(lb + o + 1, h);                                           // proofHash.out
        e.put(lb + o + 2, in2[0]);                                      // proofHash.L
        child = h;
    }
    for (int i = d; i <= n - 2; i++) {                                  // empty levels: control wires only
        const int lb = blk + L.lvl_off(i);
        int o = 0;
        if (i == n - 3) e.put_small(lb + o++, 0u);                      // st_top[n-3] = 0 (i >= d)
        if (i > 0 && i < n - 2) e.put_small(lb + o++, i == d ? 1u : 0u);// st_inew[i]
        e.put_small(lb + o++, (uint32_t)bit_of(key_s, i));
        if (tmpl_mode) {                                                // template: the Poseidon(0,0) trace of an empty level
            Fr z2[2] = {Fr::zero(), Fr::zero()};
            e.put_small(lb + o, 0u); e.put_small(lb + o + 1, 0u);       // child, aux[0]
            Fr h = WAVE ? poseidon_wave29<3>(z2, 1u, tab, e, lb + o + 4) : poseidon_trace29<3, 0>(z2, 1u, tab, e, lb + o + 4);
            e.put(lb + o + 2, h); e.put_small(lb + o + 3, 0u);          // proofHash.out, proofHash.L
        }
    }
    {   // level n-1: st_top[n-2] (== st_inew[n-1]) and lrbit
        const int lb = blk + L.lvl_off(n - 1);
        e.put_small(lb, (d == n - 1) ? 1u : 0u);
        if (n <= 253) e.put_small(lb + 1, (uint32_t)bit_of(key_s, n - 1));      // n = 254 (nLevels = 253): key bit 253 is the solved bit of Num2Bits, not a wire; the alias check starts here
    }
    int w = blk + L.off_n2bnew + (n > 253 ? 253 - n : 0);
    for (int i = n; i <= 252; i++) e.put_small(w++, (uint32_t)bit_of(key_s, i));
    emit_alias_check(e, w, key_s);
    if (tmpl_mode) {                                                    // n2bOld: oldKey = 0 (voter independent)
        uint32_t z8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 253; i++) e.put_small(blk + L.off_n2bold + i, 0u);
        emit_alias_check(e, blk + L.off_n2bold + 253, z8);
    }
    w = blk + L.off_levins;
    for (int i = 1; i <= n - 2; i++) e.put_small(w++, i == d ? 1u : 0u);
    // IsZero(sibling[i]).{out,inv}, i = 0..n-2 (out of i = n-2 is not a wire) and 1/key: one shared inversion.
    // The inv slots double as scratch for the running prefix products (Montgomery form).
    const int oz = blk + L.off_iszero;
    auto inv_slot = [&](int i) { return i < n - 2 ? oz + 2 * i + 1 : oz + 2 * (n - 2); };
    Fr acc = key0 ? Fr::one() : key;
    for (int i = 0; i <= n - 2; i++) {
        const bool z = raw_is_zero(sib + 8 * i);
        if (i < n - 2) e.put_small(oz + 2 * i, z ? 1u : 0u);
        if (z) e.put_small(inv_slot(i), 0u);
        else { e.put_raw(inv_slot(i), acc); acc = acc * load_std(sib + 8 * i); }
    }
    Fr ai = fp_inv<FrParams>(acc);
    if (WAVE) __threadfence();                                          // the prefix products parked by the first lane are read back by every lane
    for (int i = n - 2; i >= 0; i--) {
        if (raw_is_zero(sib + 8 * i)) continue;
        Fr pre = e.get_raw(inv_slot(i));
        e.put(inv_slot(i), ai * pre);
        ai = ai * load_std(sib + 8 * i);
    }
    if (key0) e.put_small(blk + 1, 0u); else e.put(blk + 1, ai);        // areKeyEquals.isz.inv = 1/key
    return child;
}

// one LANE per (voter, chain): 64 chains per wave -- the throughput form (a wave per chain issues 64 times the wave-instructions for the same
// hashes, 5 % of a pass' VALU work at batch 1024); zkc_witness_chains_wave below is the latency form (2.3 ms instead of 6.5 for one voter)
extern "C" __global__ void __launch_bounds__(64)
zkc_witness_chains(WitnessLayout L, PoseidonTable tab, const uint32_t* __restrict__ inputs, uint32_t* __restrict__ wtns,
                   int32_t* __restrict__ status, int B, int tmpl_mode) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= 3 * B) return;
    const int kind = gid / B, b = gid - kind * B;          // lanes of a wave share `kind`
    const uint32_t* in = inputs + (size_t)b * L.nInputs * 8;
    Emit e{wtns + (size_t)b * L.nWires * 8};
    const int n = L.n;
    // input slots (census.circom:51-67 declaration order)
    const uint32_t *eid = in, *nullifier = in + 16, *avail = in + 24, *vh = in + 32, *sikRoot = in + 48, *censusRoot = in + 56,
                   *address = in + 64, *password = in + 72, *signature = in + 80, *voteW = in + 88, *cs = in + 96, *ss = in + 96 + 8 * n;
    int32_t st = ZKC_W_OK;
    uint32_t key_s[8]; load_raw(key_s, address);
    if (kind == 2) {
        // ---- misc chain: header copy, range check, checkWeight, computedNullifier, sik ----
        bool in_range = true;
        for (int i = 0; i < L.nInputs; i++) { uint32_t s[8]; load_raw(s, in + 8 * i); in_range &= fp_std_lt_p<FrParams>(s); }
        if (!in_range) st = ZKC_W_ERR_INPUT_RANGE;
        e.put_small(0, 1u);
        const uint32_t* hdr[12] = {eid, eid + 8, nullifier, vh, vh + 8, sikRoot, censusRoot, voteW, avail, address, password, signature};
        for (int i = 0; i < 12; i++) { uint32_t s[8]; load_raw(s, hdr[i]); e.put_std(1 + i, s); }
        for (int i = 0; i < L.nL; i++) { uint32_t s[8]; load_raw(s, cs + 8 * i); e.put_std(13 + i, s); }
        for (int i = 0; i < L.nL; i++) { uint32_t s[8]; load_raw(s, ss + 8 * i); e.put_std(13 + L.nL + i, s); }
        e.put_small(L.off_checknull, 0u);
        // LessEqThan(252): bits 0..250 of voteWeight + 2^252 - (availableWeight + 1); bit 252 must be clear
        uint32_t p252[8] = {0, 0, 0, 0, 0, 0, 0, 1u << 28};
        Fr x = load_std(voteW) + fp_from_std<FrParams>(p252) - load_std(avail) - Fr::one();
        uint32_t xs[8]; fp_to_std<FrParams>(xs, x);
        if ((bit_of(xs, 252) | bit_of(xs, 253)) && st == ZKC_W_OK) st = ZKC_W_ERR_WEIGHT;
        for (int i = 0; i <= 250; i++) e.put_small(L.off_checkweight + i, (uint32_t)bit_of(xs, i));
        Fr sig = load_std(signature), pw = load_std(password);
        Fr nin[4] = {sig, pw, load_std(eid), load_std(eid + 8)};
        Fr nul = poseidon_trace29<5, 1>(nin, 1u, tab, e, L.off_nullifier);
        if (nul != load_std(nullifier) && st == ZKC_W_OK) st = ZKC_W_ERR_NULLIFIER;
        Fr sin[3] = {load_std(address), pw, sig};
        Fr sik = poseidon_trace29<4, 0>(sin, 1u, tab, e, L.off_sik + 1);
        e.put(L.off_sik, sik);
    } else {
        Fr key = load_std(address), value; const uint32_t *sib, *root; int blk;
        if (kind == 0) { value = load_std(avail); sib = cs; root = censusRoot; blk = L.off_census; }
        else {
            Fr sin[3] = {key, load_std(password), load_std(signature)};
            Emit none{nullptr};
            value = poseidon_trace29<4, 0>(sin, 1u, tab, none, 0);
            sib = ss; root = sikRoot; blk = L.off_sikver;
        }
        bool bad_last;
        Fr r = smt_verifier_chain<false>(L, tab, e, blk, key_s, key, value, sib, &bad_last, tmpl_mode != 0);
        if (bad_last) st = kind == 0 ? ZKC_W_ERR_LAST_SIBLING : ZKC_W_ERR_SIK_LAST_SIBLING;
        else if (r != load_std(root)) st = kind == 0 ? ZKC_W_ERR_CENSUS_ROOT : ZKC_W_ERR_SIK_ROOT;
    }
    status[(size_t)b * 3 + kind] = st;
}

// [r2] one WAVE per (voter, chain): grid = 3 B workgroups of 64 lanes, workgroup g walks chain g / B of voter g % B.  The two tree chains spread
// every Poseidon over the lanes (poseidon_wave29); everything else of a chain is small next to its 18 hashes and is done by all lanes in
// lockstep with only the first lane storing.  The misc chain (two hashes: nullifier t = 5, sik t = 4) keeps the single-lane routine.
extern "C" __global__ void __launch_bounds__(64)
zkc_witness_chains_wave(WitnessLayout L, PoseidonTable tab, const uint32_t* __restrict__ inputs, uint32_t* __restrict__ wtns,
                   int32_t* __restrict__ status, int B, int tmpl_mode) {
    const int gid = blockIdx.x;
    if (gid >= 3 * B) return;
    const int kind = gid / B, b = gid - kind * B, lane = threadIdx.x;
    const uint32_t* in = inputs + (size_t)b * L.nInputs * 8;
    Emit e{wtns + (size_t)b * L.nWires * 8, lane == 0};
    const int n = L.n;
    // input slots (census.circom:51-67 declaration order)
    const uint32_t *eid = in, *nullifier = in + 16, *avail = in + 24, *vh = in + 32, *sikRoot = in + 48, *censusRoot = in + 56,
                   *address = in + 64, *password = in + 72, *signature = in + 80, *voteW = in + 88, *cs = in + 96, *ss = in + 96 + 8 * n;
    int32_t st = ZKC_W_OK;
    uint32_t key_s[8]; load_raw(key_s, address);
    if (kind == 2) {
        if (lane != 0) return;
        // ---- misc chain: header copy, range check, checkWeight, computedNullifier, sik ----
        bool in_range = true;
        for (int i = 0; i < L.nInputs; i++) { uint32_t s[8]; load_raw(s, in + 8 * i); in_range &= fp_std_lt_p<FrParams>(s); }
        if (!in_range) st = ZKC_W_ERR_INPUT_RANGE;
        e.put_small(0, 1u);
        const uint32_t* hdr[12] = {eid, eid + 8, nullifier, vh, vh + 8, sikRoot, censusRoot, voteW, avail, address, password, signature};
        for (int i = 0; i < 12; i++) { uint32_t s[8]; load_raw(s, hdr[i]); e.put_std(1 + i, s); }
        for (int i = 0; i < L.nL; i++) { uint32_t s[8]; load_raw(s, cs + 8 * i); e.put_std(13 + i, s); }
        for (int i = 0; i < L.nL; i++) { uint32_t s[8]; load_raw(s, ss + 8 * i); e.put_std(13 + L.nL + i, s); }
        e.put_small(L.off_checknull, 0u);
        // LessEqThan(252): bits 0..250 of voteWeight + 2^252 - (availableWeight + 1); bit 252 must be clear
        uint32_t p252[8] = {0, 0, 0, 0, 0, 0, 0, 1u << 28};
        Fr x = load_std(voteW) + fp_from_std<FrParams>(p252) - load_std(avail) - Fr::one();
        uint32_t xs[8]; fp_to_std<FrParams>(xs, x);
        if ((bit_of(xs, 252) | bit_of(xs, 253)) && st == ZKC_W_OK) st = ZKC_W_ERR_WEIGHT;
        for (int i = 0; i <= 250; i++) e.put_small(L.off_checkweight + i, (uint32_t)bit_of(xs, i));
        Fr sig = load_std(signature), pw = load_std(password);
        Fr nin[4] = {sig, pw, load_std(eid), load_std(eid + 8)};
        Fr nul = poseidon_trace29<5, 1>(nin, 1u, tab, e, L.off_nullifier);
        if (nul != load_std(nullifier) && st == ZKC_W_OK) st = ZKC_W_ERR_NULLIFIER;
        Fr sin[3] = {load_std(address), pw, sig};
        Fr sik = poseidon_trace29<4, 0>(sin, 1u, tab, e, L.off_sik + 1);
        e.put(L.off_sik, sik);
        status[(size_t)b * 3 + kind] = st;
        return;
    }
    Fr key = load_std(address), value; const uint32_t *sib, *root; int blk;
    if (kind == 0) { value = load_std(avail); sib = cs; root = censusRoot; blk = L.off_census; }
    else {
        Fr sin[3] = {key, load_std(password), load_std(signature)};
        Emit none{nullptr, false};
        value = poseidon_wave29<4>(sin, 1u, tab, none, 0);
        sib = ss; root = sikRoot; blk = L.off_sikver;
    }
    bool bad_last;
    Fr r = smt_verifier_chain<true>(L, tab, e, blk, key_s, key, value, sib, &bad_last, tmpl_mode != 0);
    if (bad_last) st = kind == 0 ? ZKC_W_ERR_LAST_SIBLING : ZKC_W_ERR_SIK_LAST_SIBLING;
    else if (r != load_std(root)) st = kind == 0 ? ZKC_W_ERR_CENSUS_ROOT : ZKC_W_ERR_SIK_ROOT;
    if (lane == 0) status[(size_t)b * 3 + kind] = st;
}

// Batched Poseidon (t = nin+1 in {3,4,5}), one lane per hash; standard-form in/out.  Used by the census builder (f1).
extern "C" __global__ void __launch_bounds__(64)
zkc_poseidon_batch_kernel(PoseidonTable tab, const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int nin, size_t B) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    Emit none{nullptr};
    Fr x[4];
    for (int k = 0; k < nin; k++) x[k] = load_std(in + 8 * (i * nin + k));
    Fr h = nin == 2 ? poseidon_trace29<3, 0>(x, 1u, tab, none, 0) : nin == 3 ? poseidon_trace29<4, 0>(x, 1u, tab, none, 0) : poseidon_trace29<5, 1>(x, 1u, tab, none, 0);
    uint32_t s[8]; fp_to_std<FrParams>(s, h);
    uint4* d = reinterpret_cast<uint4*>(out + 8 * i);
    d[0] = make_uint4(s[0], s[1], s[2], s[3]); d[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// ---- f1: the hashing of the census builder (csrc/zkc_census.hip: arbo-style Poseidon sparse Merkle trees, internal/helpers.go:36-85), one lane per hash, standard form in and
// out.  The tree's values live in ONE array: val[0] = 0 (an empty subtree), val[1 + i] = leaf i, val[1 + n + j] = inner node j. ----
__device__ __forceinline__ void store_std(uint32_t* p, const Fr& h) {
    uint32_t s[8]; fp_to_std<FrParams>(s, h);
    uint4* d = reinterpret_cast<uint4*>(p); d[0] = make_uint4(s[0], s[1], s[2], s[3]); d[1] = make_uint4(s[4], s[5], s[6], s[7]);
}
// kind 0: out[i] = H(a[i], b[i], 1)            a leaf: key, value, 1 (arbo's leaf hash; smtverifier.circom's hash1New)
// kind 1: out[i] = H(a[i], b[i], c[i])         the SIK: address, password, signature (census.circom:74-77)
// kind 2: out[i] = H(a[i], b[i], c[0], c[1])   the nullifier: signature, password, electionId[0], electionId[1] (census.circom:105-109)
extern "C" __global__ void __launch_bounds__(64)
zkc_census_hash(PoseidonTable tab, int kind, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, const uint32_t* __restrict__ c, uint32_t* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Emit none{nullptr};
    Fr x[4]; x[0] = load_std(a + 8 * i); x[1] = load_std(b + 8 * i);
    Fr h;
    if (kind == 0) { x[2] = Fr::one(); h = poseidon_trace29<4, 0>(x, 1u, tab, none, 0); }
    else if (kind == 1) { x[2] = load_std(c + 8 * i); h = poseidon_trace29<4, 0>(x, 1u, tab, none, 0); }
    else { x[2] = load_std(c); x[3] = load_std(c + 8); h = poseidon_trace29<5, 1>(x, 1u, tab, none, 0); }
    store_std(out + 8 * i, h);
}
// the inner nodes of one depth: node j = order[first + t] gets val[node0 + j] = H(val[left[j]], val[right[j]]) (children one level down are finished: launches go bottom-up)
extern "C" __global__ void __launch_bounds__(64)
zkc_census_level(PoseidonTable tab, const uint32_t* __restrict__ left, const uint32_t* __restrict__ right, const uint32_t* __restrict__ order, uint32_t first, uint32_t count,
                 uint32_t* val, uint32_t node0) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint32_t j = order[first + t];
    Emit none{nullptr};
    Fr x[2]; x[0] = load_std(val + 8 * (size_t)left[j]); x[1] = load_std(val + 8 * (size_t)right[j]);
    store_std(val + 8 * ((size_t)node0 + j), poseidon_trace29<3, 0>(x, 1u, tab, none, 0));
}
// 32-byte copies val[ref] -> out[dst] for a list of (dst, ref) pairs: the sibling lists of every leaf, straight into the voters' input blocks
extern "C" __global__ void __launch_bounds__(256)
zkc_census_scatter(const uint32_t* __restrict__ val, const uint2* __restrict__ pairs, size_t count, uint32_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint2 pr = pairs[i];
    const uint4* s = reinterpret_cast<const uint4*>(val + 8 * (size_t)pr.y); uint4* d = reinterpret_cast<uint4*>(out + 8 * (size_t)pr.x);
    d[0] = s[0]; d[1] = s[1];
}
// the twelve scalar inputs of every voter's block (census.circom:51-67 order; internal/inputs.go:33-98 MockInputs): block = nIn x 32 B, zeroed beforehand
extern "C" __global__ void __launch_bounds__(256)
zkc_census_scalars(const uint32_t* __restrict__ eid, const uint32_t* __restrict__ nullifier, const uint32_t* __restrict__ avail, const uint32_t* __restrict__ vhash,
                   const uint32_t* __restrict__ sik_root, const uint32_t* __restrict__ census_root, const uint32_t* __restrict__ address, const uint32_t* __restrict__ password,
                   const uint32_t* __restrict__ signature, const uint32_t* __restrict__ vweight, size_t n, int nIn, uint32_t* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * 12) return;
    const size_t v = t / 12; const int k = (int)(t % 12);
    const uint32_t* src = k < 2 ? eid + 8 * k : k == 2 ? nullifier + 8 * v : k == 3 ? avail + 8 * v : k < 6 ? vhash + 8 * (2 * v + (k - 4)) : k == 6 ? sik_root : k == 7 ? census_root
                        : k == 8 ? address + 8 * v : k == 9 ? password + 8 * v : k == 10 ? signature + 8 * v : vweight + 8 * v;
    const uint4* s = reinterpret_cast<const uint4*>(src); uint4* d = reinterpret_cast<uint4*>(out + 8 * (v * (size_t)nIn + k));
    d[0] = s[0]; d[1] = s[1];
}

// coalesced broadcast of the template witness: one uint4 (half a wire) per lane
extern "C" __global__ void __launch_bounds__(256)
zkc_witness_tostd(uint32_t* __restrict__ wtns, size_t nwires_total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwires_total; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t* w = wtns + 8 * i;
        if (!(w[7] & 0x80000000u)) continue;
        uint4* d = reinterpret_cast<uint4*>(w); const uint4 a = d[0], b = d[1];
        Fr v; v.v[0] = a.x; v.v[1] = a.y; v.v[2] = a.z; v.v[3] = a.w; v.v[4] = b.x; v.v[5] = b.y; v.v[6] = b.z; v.v[7] = b.w & 0x7fffffffu;
        uint32_t s[8]; fp_to_std<FrParams>(s, v);
        d[0] = make_uint4(s[0], s[1], s[2], s[3]); d[1] = make_uint4(s[4], s[5], s[6], s[7]);
    }
}
extern "C" __global__ void __launch_bounds__(256)
zkc_witness_fill(const uint4* __restrict__ tmpl, uint4* __restrict__ wtns, int nWires, int B) {
    const size_t per = (size_t)nWires * 2;
    const size_t total = per * (size_t)B;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        wtns[i] = tmpl[i % per];
}

}  // namespace zkc
