// zkc_pairing_dev.hip -- f4 on the GPU: the Miller loops of the batch verifier (zkc_verify_batch, csrc/zkc_verify.hip).
//
// A batch of N proofs needs prod_i f_{6x+2, B_i}(-rho_i A_i).  On one shared accumulator that product is
//     F <- F^2 * prod_i line_{i, step}          for each of the 87 steps of the loop,
// and the lines of a pair depend on its points only, never on F.  So the work splits into three data-parallel parts and a short tail:
//   zkc_miller_lines   one lane per pair walks R <- 2R / R + Q on the twist and writes the 87 line coefficients -- functions of Q alone, so it runs (third stream) while
//                      the fold kernels are still computing the G1 points;
//   zkc_g2_membership  one lane per B_i: on the twist and psi(B) = [6x^2]B, on the context's second stream, beside the lines and the tree;
//   zkc_line_pairs     the first level of a product tree per step: two lines, each evaluated at its G1 point (left in the XYZZ form of the fold kernels: the ZZ ZZZ
//                      scaling costs nothing after the final exponentiation) -> one dense Fq12 (9 products in Fq2);
//   zkc_fq12_tree      the remaining levels, one Fq12 product per lane, 87 steps side by side;
//   host               87 values come back; F = (..(L_0)^2 L_1..) is 63 squarings and 87 products, then the three pairs of the key and ONE final exponentiation.
// Field elements are the 8 x 32-bit Montgomery residues of zkc_field.h on both sides, so the host continues where the device stopped.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <vector>
#include "zkc_prover.h"
#include "zkc_pairing.h"

namespace zkc {
using namespace zkc::pairing;

struct MillerConsts { Fq2 twist_b, psi_x, psi_y, psi2_x, psi2_y; Fq half; uint64_t pos, neg; uint64_t t_lo, t_hi; };

__device__ __forceinline__ void store_line(Fq2* __restrict__ lines, uint32_t N, uint32_t step, uint32_t i, const Fq2 l[3]) {
    Fq2* o = lines + ((size_t)step * N + i) * 3; o[0] = l[0]; o[1] = l[1]; o[2] = l[2];
}
// B_i on the twist and in G2 (psi(B) = [6x^2]B, zkc_pairing.h); *bad is set when one is not.  Its own launch on the context's second stream: a 126-step double-and-add
// per lane, longer than the walk of the lines, and nothing but the verdict waits for it -- the product tree starts as soon as the lines are written.
__global__ void __launch_bounds__(64)
zkc_g2_membership(const G2Affine* __restrict__ Q, uint32_t N, MillerConsts C, int* __restrict__ bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const G2Affine q = Q[i];
    if (q.is_inf()) return;
    bool ok = fp_sqr(q.y) == fp_sqr(q.x) * q.x + C.twist_b;
    if (ok) {
        G2XYZZ acc = G2XYZZ::from_affine(q);
        for (int b = 125; b >= 0; b--) { acc = xyzz_dbl(acc); if (((b < 64 ? C.t_lo >> b : C.t_hi >> (b - 64)) & 1)) acc = xyzz_add_affine(acc, q); }
        ok = !acc.is_inf() && acc.X == conj2(q.x) * C.psi_x * acc.ZZ && acc.Y == conj2(q.y) * C.psi_y * acc.ZZZ;
    }
    if (!ok) atomicOr(bad, 1);
}
// the lines of the Miller loop of Q_i, as coefficients (c, d0, d1) NOT yet evaluated at a G1 point: they depend on Q alone, so this kernel runs beside the fold kernels that
// are still computing the G1 side.  A Q at infinity writes the line 1.
__global__ void __launch_bounds__(64)
zkc_miller_lines(const G2Affine* __restrict__ Q, uint32_t N, MillerConsts C, Fq2* __restrict__ lines) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const G2Affine q = Q[i];
    const uint32_t nlines = 66 + (uint32_t)__popcll(C.pos | C.neg);
    if (q.is_inf()) {
        const Fq2 one[3] = {Fq2::one(), Fq2::zero(), Fq2::zero()};
        for (uint32_t s = 0; s < nlines; s++) store_line(lines, N, s, i, one);
        return;
    }
    LinePoint R{q.x, q.y, Fq2::one()};
    const Fq2 nqy = fp_neg(q.y);
    uint32_t step = 0; Fq2 l[3];
    for (int b = 63; b >= 0; b--) {
        line_dbl(R, C.twist_b, C.half, l); store_line(lines, N, step++, i, l);
        if ((C.pos >> b) & 1) { line_add(R, q.x, q.y, l); store_line(lines, N, step++, i, l); }
        else if ((C.neg >> b) & 1) { line_add(R, q.x, nqy, l); store_line(lines, N, step++, i, l); }
    }
    line_add(R, conj2(q.x) * C.psi_x, conj2(q.y) * C.psi_y, l); store_line(lines, N, step++, i, l);
    line_add(R, q.x * C.psi2_x, fp_neg(q.y * C.psi2_y), l); store_line(lines, N, step++, i, l);
}
// a line at (-P): P in XYZZ (x = X / ZZ, y = Y / ZZZ), so c y, d0 x, d1 become c Y ZZ, d0 X ZZZ, d1 ZZ ZZZ -- the whole line scaled by ZZ ZZZ, a factor in Fq that the
// final exponentiation removes (as it removes the factor -Y ZZ a line 1 of an infinite Q picks up here).  P at infinity: the pair contributes 1.
__device__ __forceinline__ void line_at(const Fq2* __restrict__ raw, const G1XYZZ& p, Fq2 out[3]) {
    if (p.is_inf()) { out[0] = Fq2::one(); out[1] = out[2] = Fq2::zero(); return; }
    out[0] = scale2(raw[0], fp_neg(p.Y * p.ZZ)); out[1] = scale2(raw[1], p.X * p.ZZZ); out[2] = scale2(raw[2], p.ZZ * p.ZZZ);
}
// out[s][t] = line[s][2t](-P_2t) * line[s][2t + 1](-P_2t+1) (the last one alone when N is odd), t < ceil(N / 2): the lines meet their G1 points here
__global__ void __launch_bounds__(64)
zkc_line_pairs(const Fq2* __restrict__ lines, const G1XYZZ* __restrict__ P, uint32_t N, uint32_t nlines, Fq12* __restrict__ out) {
    const uint32_t half = (N + 1) / 2, gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nlines * half) return;
    const uint32_t s = gid / half, t = gid - s * half;
    const Fq2* a = lines + ((size_t)s * N + 2 * t) * 3;
    Fq2 l0[3], l1[3];
    line_at(a, P[2 * t], l0);
    if (2 * t + 1 < N) { line_at(a + 3, P[2 * t + 1], l1); out[(size_t)s * half + t] = mul_034_by_034(l0, l1); }
    else out[(size_t)s * half + t] = dense_of_034(l0);
}
// out[s][t] = in[s][2t] * in[s][2t + 1], t < ceil(n / 2)
__global__ void __launch_bounds__(64)
zkc_fq12_tree(const Fq12* __restrict__ in, uint32_t n, uint32_t nlines, Fq12* __restrict__ out) {
    const uint32_t half = (n + 1) / 2, gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nlines * half) return;
    const uint32_t s = gid / half, t = gid - s * half;
    const Fq12* a = in + (size_t)s * n + 2 * t;
    out[(size_t)s * half + t] = (2 * t + 1 < n) ? a[0] * a[1] : a[0];
}

static MillerConsts miller_consts() {
    const Consts& K = consts(); const AteLoop& L = ate_loop();
    MillerConsts C; C.twist_b = K.twist_b; C.psi_x = K.psi_x; C.psi_y = K.psi_y; C.psi2_x = K.psi2_x; C.psi2_y = K.psi2_y; C.half = K.half; C.pos = C.neg = 0;
    for (int b = 0; b < 64; b++) { if (L.digit[b] > 0) C.pos |= 1ull << b; else if (L.digit[b] < 0) C.neg |= 1ull << b; }
    C.t_lo = 0xf83e9682e87cfd46ull; C.t_hi = 0x6f4d8248eeb859fbull;           // 6 x^2, bit 126 on top (zkc_pairing.h g2_in_subgroup)
    return C;
}
static int dev_fail(zkc_ctx* ctx, hipError_t e, const char* what) { ctx->err = std::string(what) + ": " + hipGetErrorString(e); (void)hipGetLastError(); return ZKC_ERR_HIP; }

static uint32_t verify_chunk() {                                              // pairs per round of kernels (tests shrink it to walk several rounds with a few hundred proofs)
    const char* ce = getenv("ZKC_VERIFY_CHUNK");
    return ce ? (uint32_t)std::min(16384, std::max(2, atoi(ce))) : 16384u;
}
static uint32_t n_lines(const MillerConsts& C) { return 66 + (uint32_t)__builtin_popcountll(C.pos | C.neg); }
void miller_join(zkc_ctx* ctx) { (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamSynchronize(ctx->fin_stream); }

// First half, before anything else of the batch touches the GPU: all N points B_i go up (second stream); their membership tests start there -- the longest kernel of a
// batch, needed only for the verdict -- and the lines of the first round of pairs start on the third stream: neither needs the G1 side, so both run beside the fold
// kernels.  miller_product_dev joins them (miller_join after any failure in between).  The caller holds the context's lock.
int miller_membership_begin(zkc_ctx* ctx, const G2Affine* h_Q, uint32_t N) {
    const MillerConsts C = miller_consts();
    const uint32_t cap = std::min(N, verify_chunk());
    void* q; int rc;
    if ((rc = zkc_vws(ctx, zkc_ctx::VWS_Q, (size_t)N * sizeof(G2Affine), &q))) return rc;
    G2Affine* d_Q = (G2Affine*)q;
    if ((rc = zkc_vws(ctx, zkc_ctx::VWS_BAD, sizeof(int), &q))) return rc;
    int* d_bad = (int*)q;
    if ((rc = zkc_vws(ctx, zkc_ctx::VWS_LINES, (size_t)n_lines(C) * cap * 3 * sizeof(Fq2), &q))) return rc;
    Fq2* d_lines = (Fq2*)q;
    hipError_t e;
    if (!ctx->ev_vws_up && (e = hipEventCreateWithFlags(&ctx->ev_vws_up, hipEventDisableTiming)) != hipSuccess) return dev_fail(ctx, e, "miller_membership_begin: hipEventCreate");
    if (!ctx->ev_vws_lines && (e = hipEventCreateWithFlags(&ctx->ev_vws_lines, hipEventDisableTiming)) != hipSuccess) return dev_fail(ctx, e, "miller_membership_begin: hipEventCreate");
    if ((e = hipMemcpyAsync(d_Q, h_Q, (size_t)N * sizeof(G2Affine), hipMemcpyHostToDevice, ctx->stream2)) != hipSuccess ||
        (e = hipMemsetAsync(d_bad, 0, sizeof(int), ctx->stream2)) != hipSuccess ||
        (e = hipEventRecord(ctx->ev_vws_up, ctx->stream2)) != hipSuccess ||                   // the line kernels read these points: they wait for the copy, not for the membership kernel
        (e = hipStreamWaitEvent(ctx->fin_stream, ctx->ev_vws_up, 0)) != hipSuccess ||
        (e = hipStreamWaitEvent(ctx->stream, ctx->ev_vws_up, 0)) != hipSuccess) { miller_join(ctx); return dev_fail(ctx, e, "miller_membership_begin: upload"); }
    hipLaunchKernelGGL(zkc_g2_membership, dim3((N + 63) / 64), dim3(64), 0, ctx->stream2, d_Q, N, C, d_bad);
    hipLaunchKernelGGL(zkc_miller_lines, dim3((cap + 63) / 64), dim3(64), 0, ctx->fin_stream, d_Q, cap, C, d_lines);
    if ((e = hipGetLastError()) != hipSuccess || (e = hipEventRecord(ctx->ev_vws_lines, ctx->fin_stream)) != hipSuccess) { miller_join(ctx); return dev_fail(ctx, e, "miller_membership_begin: launch"); }
    return ZKC_OK;
}
// Second half: prod_i f_{6x+2, Q_i}(-P_i) over the N pairs, P on the device (XYZZ, as the fold kernels write them), Q where miller_membership_begin put them; *bad != 0: some
// Q_i is not in G2 (the product is meaningless then).  Pairs are taken 16 384 at a time (300 MB of line coefficients); the lines of the first round are already on their way.
// Always waits for the second and third streams, error or not.
int miller_product_dev(zkc_ctx* ctx, const G1XYZZ* d_P, uint32_t N, Fq12* product, int* bad) {
    const AteLoop& L = ate_loop(); const MillerConsts C = miller_consts();
    const uint32_t nlines = n_lines(C), CHUNK = verify_chunk();
    const uint32_t cap = std::min(N, CHUNK), hcap = (cap + 1) / 2;
    const bool vtrace = getenv("ZKC_VERIFY_TRACE") != nullptr;
    auto vnow = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double mt0 = vnow(); double mt_chunks = 0;
    std::vector<Fq12> step(nlines), acc(nlines, one12());
    const int rc = [&]() -> int {
        const G2Affine* d_Q = (const G2Affine*)ctx->vws[zkc_ctx::VWS_Q];
        Fq2* d_lines = (Fq2*)ctx->vws[zkc_ctx::VWS_LINES];
        Fq12 *d_a, *d_b; void* q; int r;
        if ((r = zkc_vws(ctx, zkc_ctx::VWS_TREE_A, (size_t)nlines * hcap * sizeof(Fq12), &q))) return r; d_a = (Fq12*)q;
        if ((r = zkc_vws(ctx, zkc_ctx::VWS_TREE_B, (size_t)nlines * ((hcap + 1) / 2) * sizeof(Fq12), &q))) return r; d_b = (Fq12*)q;
        hipError_t e;
        for (uint32_t lo = 0; lo < N; lo += CHUNK) {
            const uint32_t n = std::min(CHUNK, N - lo);
            // this round's lines were started on the third stream: by miller_membership_begin (round 0) or by the round before, as soon as ITS pairs kernel had read the buffer
            if ((e = hipStreamWaitEvent(ctx->stream, ctx->ev_vws_lines, 0)) != hipSuccess) return dev_fail(ctx, e, "miller_product_dev: event");
            uint32_t m = (n + 1) / 2;
            hipLaunchKernelGGL(zkc_line_pairs, dim3((nlines * m + 63) / 64), dim3(64), 0, ctx->stream, d_lines, d_P + lo, n, nlines, d_a);
            if (lo + CHUNK < N) {                                   // the next round's lines run beside this round's product tree (the pairs kernel is the only reader of d_lines)
                const uint32_t lo2 = lo + CHUNK, n2 = std::min(CHUNK, N - lo2);
                if ((e = hipEventRecord(ctx->ev_vws_up, ctx->stream)) != hipSuccess || (e = hipStreamWaitEvent(ctx->fin_stream, ctx->ev_vws_up, 0)) != hipSuccess) return dev_fail(ctx, e, "miller_product_dev: event");
                hipLaunchKernelGGL(zkc_miller_lines, dim3((n2 + 63) / 64), dim3(64), 0, ctx->fin_stream, d_Q + lo2, n2, C, d_lines);
                if ((e = hipEventRecord(ctx->ev_vws_lines, ctx->fin_stream)) != hipSuccess) return dev_fail(ctx, e, "miller_product_dev: event");
            }
            Fq12 *src = d_a, *dst = d_b;
            while (m > 1) {
                const uint32_t h = (m + 1) / 2;
                hipLaunchKernelGGL(zkc_fq12_tree, dim3((nlines * h + 63) / 64), dim3(64), 0, ctx->stream, src, m, nlines, dst);
                std::swap(src, dst); m = h;
            }
            if ((e = hipGetLastError()) != hipSuccess) return dev_fail(ctx, e, "miller_product_dev: launch");
            if ((e = hipMemcpyAsync(step.data(), src, (size_t)nlines * sizeof(Fq12), hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess ||
                (e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return dev_fail(ctx, e, "miller_product_dev: download");
            for (uint32_t s = 0; s < nlines; s++) acc[s] = lo ? acc[s] * step[s] : step[s];
        }
        return ZKC_OK;
    }();
    const hipError_t ej = hipStreamSynchronize(ctx->stream2), ej2 = hipStreamSynchronize(ctx->fin_stream);      // the membership tests (and, after a failure above, the first lines)
    if (rc) return rc;
    if (ej != hipSuccess || ej2 != hipSuccess) return dev_fail(ctx, ej != hipSuccess ? ej : ej2, "miller_product_dev: membership kernel");
    mt_chunks = vnow();
    int hb = 0; hipError_t e;
    if ((e = hipMemcpy(&hb, ctx->vws[zkc_ctx::VWS_BAD], sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess) return dev_fail(ctx, e, "miller_product_dev: download");
    *bad = hb;
    // the accumulator, step by step (the same walk as pairing::multi_miller)
    Fq12 f = one12(); uint32_t idx = 0;
    for (int b = 63; b >= 0; b--) { if (b != 63) f = sqr12(f); f = f * acc[idx++]; if (L.digit[b]) f = f * acc[idx++]; }
    f = f * acc[idx++]; f = f * acc[idx++];
    *product = f;
    if (vtrace) fprintf(stderr, "miller_product_dev N=%u: kernels + transfers %.2f ms, accumulator on the host %.2f ms\n", N, mt_chunks - mt0, vnow() - mt_chunks);
    return ZKC_OK;
}
}  // namespace zkc
