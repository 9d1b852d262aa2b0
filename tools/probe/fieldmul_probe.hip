// fieldmul_probe.hip -- three ways to form a 254-bit Montgomery product on gfx950, timed in one harness ("next" item 1):
//
//   A  f29_mul            the product the prover uses (csrc/zkc_f29.h): nine 29-bit limbs, 81 + 81 v_mad_u64_u32 into carry-free 64-bit columns
//   B  fp52_mul           five 52-bit limbs held in doubles; every limb product is split exactly into its high and low 52 bits by two
//                         round-toward-zero FMAs (Emmart, Zheng, Weems, ARITH 2018: hi = fma(a, b, 2^104), lo = fma(a, b, 2^104 + 2^52 - hi)) and the
//                         halves are summed as 64-bit integers on their raw bit patterns; word-serial Montgomery reduction, radix 2^52
//   C  m x p on the matrix pipe: the constant-operand half of the reduction (m p, p the modulus) as v_mfma_i32_32x32x32_i8 with a Toeplitz matrix of
//      p's bytes as the B operand.  64 lanes hold 64 different m, so the A operands need a half-wave swap (v_permlane32_swap), and the 32x32
//      results come back one byte-COLUMN per lane: a transposition through LDS and a recombination of 64 byte columns into words follow.
//      (The i8 operands are signed: the probe restricts both operands to 7-bit digits, which a real implementation could not; it would need a
//      balanced-digit recoding of m on top of what is timed here.)
//   D  the 81 v_mad_u64_u32 that C would replace (m_i p_j into 18 columns), same harness.
//
// Every kernel runs dependent chains (the output of a product is an operand of the next), W waves per SIMD on all 1024 SIMDs, like tools/probe/rate_probe.hip;
// A and B are verified against a host integer implementation before they are timed, C against the integer convolution.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zk-franchise-proof-circuit_amd/csrc tools/probe/fieldmul_probe.hip -o /tmp/fieldmul_probe && /tmp/fieldmul_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "zkc_f29.h"

using namespace zkc;
typedef unsigned __int128 u128;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// ------------------------------------------------------------------ B: 5 x 52-bit limbs in doubles ------------------------------------------------------------------
__device__ __constant__ double kP52[5] = {(double)0x8c16d87cfd47ull, (double)0x916871ca8d3c2ull, (double)0x181585d97816aull, (double)0xa029b85045b68ull, (double)0x30644e72e131ull};
static const uint64_t hP52[5] = {0x8c16d87cfd47ull, 0x916871ca8d3c2ull, 0x181585d97816aull, 0xa029b85045b68ull, 0x30644e72e131ull};
constexpr uint64_t NINV52 = 0x20782e4866389ull;              // -p^-1 mod 2^52
constexpr uint64_t MASK52 = (1ull << 52) - 1;
constexpr uint64_t EXP52 = 0x433ull << 52, EXP104 = 0x467ull << 52;     // bit patterns of 2^52 and 2^104

__device__ __forceinline__ uint64_t dbits(double d) { return (uint64_t)__double_as_longlong(d); }
__device__ __forceinline__ double bitsd(uint64_t u) { return __longlong_as_double((long long)u); }
// exact split of a b (a, b integers below 2^52 held in doubles) under round-toward-zero: hi = 2^104 + 2^52 floor(a b / 2^52), lo = 2^52 + (a b mod 2^52)
__device__ __forceinline__ void split52(double a, double b, uint64_t& hi_sum, uint64_t& lo_sum) {
    const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
    const double h = __builtin_fma(a, b, C1);
    const double l = __builtin_fma(a, b, C2 - h);
    hi_sum += dbits(h); lo_sum += dbits(l);
}
// r = a b / 2^260 mod p (limbs below 2^52, value below a b / 2^260 + p).  Must run with the FP64 rounding mode set to toward-zero.
__device__ __forceinline__ void fp52_mul(double r[5], const double a[5], const double b[5]) {
    uint64_t hi[10], lo[10];
#pragma unroll
    for (int k = 0; k < 10; k++) hi[k] = lo[k] = 0;
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j < 5; j++) split52(a[i], b[j], hi[i + j], lo[i + j]);
    // column k = lo[k] + hi[k - 1], each a sum of n raw patterns: the exponent fields are removed with wrap-around arithmetic (the true sums are below 2^57)
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        // products already in column i: min(i, 4 - ... ) -- counted at compile time
        const int n_lo = (i + 1) + i;              // a b terms with i' + j' = i (i + 1 of them, i <= 4) and m_t p_j terms with t + j = i, t < i (i of them)
        const int n_hi = 2 * i;                     // column i - 1: i products a b and i products m_t p_j (t + j = i - 1, the j = 0 term of round i - 1 included)
        uint64_t q = lo[i] - (uint64_t)n_lo * EXP52 + carry;
        if (i > 0) q += hi[i - 1] - (uint64_t)n_hi * EXP104;
        const double qd = bitsd((q & MASK52) | EXP52) - 0x1p52;
        uint64_t mh = 0, ml = 0; split52(qd, (double)NINV52, mh, ml);
        const double md = bitsd(ml) - 0x1p52;                               // (q ninv) mod 2^52
        uint64_t l0 = 0;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            if (j == 0) { uint64_t h0 = 0; split52(md, kP52[0], h0, l0); hi[i] += h0; }
            else split52(md, kP52[j], hi[i + j], lo[i + j]);
        }
        carry = (q + (l0 - EXP52)) >> 52;                                   // the low 52 bits cancel
    }
    // columns 5..9 -> limbs
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int c = 5 + k;
        const int n_lo = c <= 8 ? (9 - c) + (9 - c) : 0;                     // a b terms: i + j = c -> 9 - c of them; m p terms: t + j = c -> 9 - c
        const int n_hi = (9 - (c - 1)) * 2;                                  // column c - 1 >= 4: 9 - (c - 1) of each kind (c - 1 = 4: 5 and 5)
        uint64_t t = hi[c - 1] - (uint64_t)n_hi * EXP104 + carry;
        if (c <= 8) t += lo[c] - (uint64_t)n_lo * EXP52;
        if (k < 4) { r[k] = bitsd((t & MASK52) | EXP52) - 0x1p52; carry = t >> 52; }
        else r[k] = (double)(long long)t;                                    // top limb: whatever is left (below 2^52 for operands below 2^256)
    }
}
static void host_fp52_mul(uint64_t r[5], const uint64_t a[5], const uint64_t b[5]) {
    u128 c[11]; for (auto& v : c) v = 0;
    for (int i = 0; i < 5; i++) for (int j = 0; j < 5; j++) c[i + j] += (u128)a[i] * b[j];
    for (int i = 0; i < 5; i++) {
        const uint64_t m = ((uint64_t)c[i] * NINV52) & MASK52;
        for (int j = 0; j < 5; j++) c[i + j] += (u128)m * hP52[j];
        c[i + 1] += c[i] >> 52;
    }
    for (int k = 0; k < 5; k++) { if (k < 4) { r[k] = (uint64_t)c[5 + k] & MASK52; c[6 + k] += c[5 + k] >> 52; } else r[k] = (uint64_t)c[9]; }
}
static void host_f29_mul(uint32_t r[9], const uint32_t a[9], const uint32_t b[9]) {          // the same column arithmetic as zkc_f29.h, restated with 128-bit columns
    constexpr L9 Pl = F29K<FqParams>::p;
    u128 c[19]; for (auto& v : c) v = 0;
    for (int i = 0; i < 9; i++) for (int j = 0; j < 9; j++) c[i + j] += (u128)a[i] * b[j];
    for (int i = 0; i < 9; i++) {
        const uint32_t m = ((uint32_t)c[i] * F29K<FqParams>::ninv) & F29_MASK;
        for (int j = 0; j < 9; j++) c[i + j] += (u128)m * Pl.l[j];
        c[i + 1] += c[i] >> 29;
    }
    for (int k = 0; k < 9; k++) { if (k < 8) { r[k] = (uint32_t)c[9 + k] & F29_MASK; c[10 + k] += c[9 + k] >> 29; } else r[k] = (uint32_t)c[17]; }
}

template <int CHAINS>
__global__ void __launch_bounds__(256) k_f29(uint32_t* out, const uint32_t* in, int iters) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x[CHAINS][9], y[9];
#pragma unroll
    for (int k = 0; k < 9; k++) { y[k] = in[k]; for (int c = 0; c < CHAINS; c++) x[c][k] = in[9 + k] ^ (k < 8 ? ((uint32_t)(gid * 2654435761u + c * 97u) & 0xffffu) : 0u); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) { uint32_t t[9]; f29_mul<FqParams>(t, x[c], y);
#pragma unroll
            for (int k = 0; k < 9; k++) x[c][k] = t[k]; }
    }
#pragma unroll
    for (int c = 0; c < CHAINS; c++)
#pragma unroll
        for (int k = 0; k < 9; k++) out[((size_t)gid * CHAINS + c) * 9 + k] = x[c][k];
}
template <int CHAINS>
__global__ void __launch_bounds__(256) k_fp52(double* out, const double* in, int iters) {
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");         // FP64 rounding: toward zero
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    double x[CHAINS][5], y[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { y[k] = in[k]; for (int c = 0; c < CHAINS; c++) x[c][k] = in[5 + k] + (k < 4 ? (double)((uint32_t)(gid * 2654435761u + c * 97u) & 0xffffu) : 0.0); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) { double t[5]; fp52_mul(t, x[c], y);
#pragma unroll
            for (int k = 0; k < 5; k++) x[c][k] = t[k]; }
    }
#pragma unroll
    for (int c = 0; c < CHAINS; c++)
#pragma unroll
        for (int k = 0; k < 5; k++) out[((size_t)gid * CHAINS + c) * 5 + k] = x[c][k];
}

// ------------------------------------------------------------------ C / D: m x p, p constant ------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int ROWW = 68;                                                    // LDS row of 64 columns, padded (16-byte aligned rows)
__device__ __constant__ uint8_t kPB[32] = {71, 125, 124, 88, 22, 12, 32, 60, 13, 74, 113, 104, 17, 106, 1, 23, 93, 88, 1, 1, 54, 69, 80, 56, 41, 32, 49, 97, 114, 78, 100, 48};   // p's bytes & 0x7f
static const uint8_t hPB[32] = {71, 125, 124, 88, 22, 12, 32, 60, 13, 74, 113, 104, 17, 106, 1, 23, 93, 88, 1, 1, 54, 69, 80, 56, 41, 32, 49, 97, 114, 78, 100, 48};
// one wave: 64 lanes x (m of 8 words = 32 digits of 7 bits, one per byte) -> the 64 byte columns of m p, summed into 16 words of 64 bits per lane
__global__ void __launch_bounds__(128) k_mfma(uint64_t* out, const uint32_t* in, int iters) {
    __shared__ int tile[2][64 * ROWW];                                      // 34 KB: two waves per workgroup
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l32 = lane & 31;
    int* T = tile[wv];
    uint32_t m[8];
#pragma unroll
    for (int k = 0; k < 8; k++) m[k] = (in[k] + (uint32_t)(blockIdx.x * 128 + threadIdx.x) * 0x01010101u * (k + 1)) & 0x7f7f7f7fu;
    // B operands: column n = 32 nt + l32, K slice (half, byte i) stands for digit kd = 16 half + i: B[kd][n] = p_byte[n - kd]
    v4i B[2];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int w = 0; w < 4; w++) { uint32_t v = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) { const int kd = 16 * half + 4 * w + i, d = 32 * nt + l32 - kd; v |= (uint32_t)((d >= 0 && d < 32) ? kPB[d] : 0) << (8 * i); }
            B[nt][w] = (int)v; }
    uint64_t c[16];
    for (int it = 0; it < iters; it++) {
        // A operands: tile 0 = elements of lanes 0..31 (low lanes give digits 0..15 = words 0..3, high lanes must supply their words 4..7), tile 1 = lanes 32..63
        v4i A0, A1;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            auto r = __builtin_amdgcn_permlane32_swap(m[w], m[w + 4], false, false);      // r[0]: low lanes own m[w], high lanes m[w+4] of lane - 32; r[1]: low lanes m[w] of lane + 32, high lanes own m[w+4]
            A0[w] = (int)r[0]; A1[w] = (int)r[1];
        }
        const v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        v16i D00 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A0, B[0], z, 0, 0, 0), D01 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A0, B[1], z, 0, 0, 0);
        v16i D10 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1, B[0], z, 0, 0, 0), D11 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1, B[1], z, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 16; v++) {
            const int row = 8 * (v / 4) + 4 * half + (v % 4);
            T[row * ROWW + l32] = D00[v]; T[row * ROWW + 32 + l32] = D01[v];
            T[(32 + row) * ROWW + l32] = D10[v]; T[(32 + row) * ROWW + 32 + l32] = D11[v];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        const int4* rowp = reinterpret_cast<const int4*>(T + lane * ROWW);
#pragma unroll
        for (int k = 0; k < 16; k++) {                                      // word k of m p = sum of byte columns 4k..4k+3 at their weights
            const int4 q = rowp[k];
            uint64_t s = (uint32_t)q.x; s += (uint64_t)(uint32_t)q.y << 8; s += (uint64_t)(uint32_t)q.z << 16; s += (uint64_t)(uint32_t)q.w << 24;
            c[k] = s;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 8; k++) m[k] = ((uint32_t)c[k] ^ (uint32_t)c[k + 8]) & 0x7f7f7f7fu;          // dependency: the next m comes out of this product
    }
#pragma unroll
    for (int k = 0; k < 16; k++) out[(size_t)(blockIdx.x * 128 + threadIdx.x) * 16 + k] = c[k];
}
// D: the 81 mads of one reduction (nine 29-bit m_i times nine limbs of p into 18 columns), the work C would take off the vector unit
__global__ void __launch_bounds__(256) k_mads(uint64_t* out, const uint32_t* in, int iters) {
    constexpr L9 Pl = F29K<FqParams>::p;
    uint32_t m[9];
#pragma unroll
    for (int k = 0; k < 9; k++) m[k] = (in[k] + (uint32_t)(blockIdx.x * 256 + threadIdx.x) * (k + 3)) & F29_MASK;
    uint64_t c[18];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 18; k++) c[k] = 0;
#pragma unroll
        for (int i = 0; i < 9; i++)
#pragma unroll
            for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)m[i] * Pl.l[j];
#pragma unroll
        for (int k = 0; k < 9; k++) m[k] = ((uint32_t)c[k] ^ (uint32_t)c[k + 9]) & F29_MASK;
    }
#pragma unroll
    for (int k = 0; k < 18; k++) out[(size_t)(blockIdx.x * 256 + threadIdx.x) * 18 + k] = c[k];
}

template <class F> static float time_it(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
static void report(const char* name, int w, int chains, int iters, float ms, const char* unit) {
    const double per_lane = (double)chains * iters, total = per_lane * 256.0 * 256 * w;
    // cycles per product per wave on one SIMD at the 2.15 GHz these loops sustain (profiles/r02_power_clock_trace.json): SIMD time / products issued on it
    const double cyc = ms * 1e-3 * 2.15e9 / (per_lane * w);
    printf("%-44s W=%d  %8.3f ms  %8.2f G %s/s  %7.1f SIMD cycles per wave-level %s at 2.15 GHz\n", name, w, ms, total / ms / 1e6, unit, cyc, unit);
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("# %s, %d CUs; dependent chains, W = waves per SIMD on every SIMD\n", prop.gcnArchName, prop.multiProcessorCount);
    const int iters = 2000;
    // ---- operands: y = a fixed value below p, x = another, both as 29-bit and as 52-bit limbs ----
    uint64_t y52[5] = {0x123456789abcdull, 0xfedcba9876543ull, 0x0f0f0f0f0f0f0ull, 0x7777777777777ull, 0x2abcdef01234ull};
    uint64_t x52[5] = {0x3141592653589ull, 0x2718281828459ull, 0x1618033988749ull, 0x1414213562373ull, 0x1732050807568ull};
    auto to29 = [](uint32_t o[9], const uint64_t a[5]) {
        for (int i = 0; i < 9; i++) { const int li = 29 * i / 52, off = 29 * i % 52; u128 v = a[li] >> off; if (li + 1 < 5) v |= (u128)a[li + 1] << (52 - off); o[i] = (uint32_t)v & F29_MASK; } };
    uint32_t y29[9], x29[9]; to29(y29, y52); to29(x29, x52);

    // ---- correctness: 64 lanes x 3 products against the host integer code ----
    {
        const int n = 256;
        uint32_t h_in[18]; memcpy(h_in, y29, 36); memcpy(h_in + 9, x29, 36);
        double h_ind[10]; for (int k = 0; k < 5; k++) { h_ind[k] = (double)y52[k]; h_ind[5 + k] = (double)x52[k]; }
        uint32_t* d_in; uint32_t* d_out; double* d_ind; double* d_outd;
        CHECK(hipMalloc(&d_in, sizeof h_in)); CHECK(hipMalloc(&d_out, n * 9 * 4)); CHECK(hipMalloc(&d_ind, sizeof h_ind)); CHECK(hipMalloc(&d_outd, n * 5 * 8));
        CHECK(hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_ind, h_ind, sizeof h_ind, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_f29<1>, dim3(1), dim3(n), 0, 0, d_out, d_in, 3); hipLaunchKernelGGL(k_fp52<1>, dim3(1), dim3(n), 0, 0, d_outd, d_ind, 3);
        std::vector<uint32_t> o29(n * 9); std::vector<double> o52(n * 5);
        CHECK(hipMemcpy(o29.data(), d_out, n * 9 * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(o52.data(), d_outd, n * 5 * 8, hipMemcpyDeviceToHost));
        int bad29 = 0, bad52 = 0;
        for (int g = 0; g < n; g++) {
            const uint32_t tw = (uint32_t)(g * 2654435761u) & 0xffffu;
            uint32_t a[9], t[9]; for (int k = 0; k < 9; k++) a[k] = x29[k] ^ (k < 8 ? tw : 0u);
            for (int it = 0; it < 3; it++) { host_f29_mul(t, a, y29); memcpy(a, t, 36); }
            for (int k = 0; k < 9; k++) bad29 += a[k] != o29[g * 9 + k];
            uint64_t b[5], u[5]; for (int k = 0; k < 5; k++) b[k] = x52[k] + (k < 4 ? tw : 0);
            for (int it = 0; it < 3; it++) { host_fp52_mul(u, b, y52); memcpy(b, u, 40); }
            for (int k = 0; k < 5; k++) bad52 += (double)b[k] != o52[g * 5 + k];
        }
        printf("# check: f29_mul %s, fp52_mul %s (256 lanes x chains of 3 products against host integer code)\n", bad29 ? "MISMATCH" : "ok", bad52 ? "MISMATCH" : "ok");
        // m x p on the matrix pipe against the integer convolution
        uint32_t h_m[8] = {0x01020304u, 0x11121314u, 0x21222324u, 0x31323334u, 0x41424344u, 0x51525354u, 0x61626364u, 0x71727374u};
        uint32_t* d_m; uint64_t* d_c; CHECK(hipMalloc(&d_m, 32)); CHECK(hipMalloc(&d_c, 256 * 16 * 8)); CHECK(hipMemcpy(d_m, h_m, 32, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_mfma, dim3(2), dim3(128), 0, 0, d_c, d_m, 1);
        std::vector<uint64_t> oc(256 * 16); CHECK(hipMemcpy(oc.data(), d_c, 256 * 16 * 8, hipMemcpyDeviceToHost));
        int badm = 0;
        for (int t = 0; t < 256; t++) {
            uint8_t mb[32]; for (int k = 0; k < 8; k++) { const uint32_t w = (h_m[k] + (uint32_t)t * 0x01010101u * (k + 1)) & 0x7f7f7f7fu; memcpy(mb + 4 * k, &w, 4); }
            uint64_t col[64] = {0}; for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) col[i + j] += (uint64_t)mb[i] * hPB[j];
            for (int k = 0; k < 16; k++) { const uint64_t s = col[4 * k] + (col[4 * k + 1] << 8) + (col[4 * k + 2] << 16) + (col[4 * k + 3] << 24); badm += s != oc[t * 16 + k]; }
        }
        printf("# check: m x p through v_mfma_i32_32x32x32_i8 + LDS transposition %s (7-bit digits)\n", badm ? "MISMATCH" : "ok");
        if (bad29 || bad52 || badm) return 2;
        hipFree(d_in); hipFree(d_out); hipFree(d_ind); hipFree(d_outd); hipFree(d_m); hipFree(d_c);
    }
    // ---- timing ----
    uint32_t h_in[18]; memcpy(h_in, y29, 36); memcpy(h_in + 9, x29, 36);
    double h_ind[10]; for (int k = 0; k < 5; k++) { h_ind[k] = (double)y52[k]; h_ind[5 + k] = (double)x52[k]; }
    uint32_t* d_in; double* d_ind; void* d_out;
    CHECK(hipMalloc(&d_in, sizeof h_in)); CHECK(hipMalloc(&d_ind, sizeof h_ind)); CHECK(hipMalloc(&d_out, (size_t)256 * 8 * 256 * 18 * 8));
    CHECK(hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_ind, h_ind, sizeof h_ind, hipMemcpyHostToDevice));
    for (int w : {1, 2, 3, 4}) {
        const dim3 grid(256 * w), blk(256);
        float ms = time_it([&] { hipLaunchKernelGGL(k_f29<2>, grid, blk, 0, 0, (uint32_t*)d_out, d_in, iters); });
        report("A  f29_mul (9 x 29 bit, v_mad_u64_u32)", w, 2, iters, ms, "products");
        ms = time_it([&] { hipLaunchKernelGGL(k_fp52<2>, grid, blk, 0, 0, (double*)d_out, d_ind, iters); });
        report("B  fp52_mul (5 x 52 bit, v_fma_f64 pairs)", w, 2, iters, ms, "products");
    }
    for (int w : {1, 2}) {                                                   // 34 KB of LDS per two-wave workgroup: four workgroups (two waves per SIMD) per CU at most
        const dim3 grid(256 * w * 2), blk(128);
        float ms = time_it([&] { hipLaunchKernelGGL(k_mfma, grid, blk, 0, 0, (uint64_t*)d_out, d_in, iters); });
        report("C  m x p: MFMA i8 + LDS transposition", w, 1, iters, ms, "reductions");
    }
    for (int w : {1, 2, 3}) {
        const dim3 grid(256 * w), blk(256);
        float ms = time_it([&] { hipLaunchKernelGGL(k_mads, grid, blk, 0, 0, (uint64_t*)d_out, d_in, iters); });
        report("D  m x p: 81 v_mad_u64_u32", w, 1, iters, ms, "reductions");
    }
    return 0;
}
