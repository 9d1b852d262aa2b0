// zkc_prover.h -- device-resident proving key and the stage launchers shared by the prover translation units.
#pragma once
#include "zkc_internal.h"
#include "zkc_curve.h"

namespace zkc {
constexpr int MSM_C = 13;                          // Pippenger window bits (signed digits -> 2^(c-1) buckets per window)
constexpr int MSM_NW = (254 + MSM_C) / MSM_C;      // 20 windows cover 260 bits
constexpr int MSM_HALF = 1 << (MSM_C - 1);
constexpr int MSM_NB = MSM_NW * MSM_HALF;          // buckets per MSM
constexpr int MSM_HEAVY = 256;                     // buckets with more points go to the block-per-bucket kernel
constexpr int MSM_MAX_HEAVY = 4096;
constexpr int MSM_GROUP = 32;                      // buckets per thread in the running-sum reduction
}

struct zkc_zkey {
    zkc_ctx* ctx = nullptr;
    uint32_t nVars = 0, nPub = 0, n = 0, logn = 0, nCoeffs = 0;
    zkc::G1Affine alpha1, beta1, delta1;           // host copies, Montgomery
    zkc::G2Affine beta2, gamma2, delta2;
    std::vector<zkc::G1Affine> ic;
    // device: CSR of section 4 (rows [0,n) = A, [n,2n) = B), values as stored (val * R^2)
    uint32_t *d_rowptr = nullptr, *d_col = nullptr; zkc::Fr* d_val = nullptr;
    zkc::Fr *d_tw_fwd = nullptr, *d_tw_inv = nullptr, *d_coset = nullptr;   // w^j, w^-j (j < n/2), g^i / n
    // window-shifted base tables: T[w][i] = 2^(c*w) * P_i, affine Montgomery
    zkc::G1Affine *d_A = nullptr, *d_B1 = nullptr, *d_C = nullptr, *d_H = nullptr; zkc::G2Affine* d_B2 = nullptr;
    // per-proof work buffers (one proof in flight per zkey handle)
    zkc::Fr *d_a = nullptr, *d_b = nullptr, *d_c = nullptr, *d_t = nullptr; uint32_t* d_p = nullptr;   // n each
    uint32_t *d_keys = nullptr, *d_vals = nullptr, *d_keys2 = nullptr, *d_vals2 = nullptr, *d_off = nullptr, *d_heavy = nullptr;
    void* d_sort_tmp = nullptr; size_t sort_tmp_sz = 0;
    void *d_buckets = nullptr, *d_partial = nullptr, *d_results = nullptr;   // XYZZ arrays
    void* h_results = nullptr;                      // pinned host mirror of d_results
};

namespace zkc {
int ntt_run(zkc_ctx* ctx, const Fr* src, Fr* dst, const Fr* tw, const Fr* scale, int logn);
// result slot r of zk->d_results receives sum_i scalars[i] * P_i for the table `table` (nw x count points)
int msm_g1_run(zkc_zkey* zk, const G1Affine* table, const uint32_t* d_scalars_std, uint32_t count, int slot);
int msm_g2_run(zkc_zkey* zk, const G2Affine* table, const uint32_t* d_scalars_std, uint32_t count, int slot);
int msm_precompute_g1(zkc_ctx* ctx, const G1Affine* d_base, uint32_t count, G1Affine* d_table);   // d_table[0..count) = base on entry
int msm_precompute_g2(zkc_ctx* ctx, const G2Affine* d_base, uint32_t count, G2Affine* d_table);
}
