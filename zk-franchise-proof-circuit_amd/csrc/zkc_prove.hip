// zkc_prove.hip -- proving-key residency and the Groth16 prove pipeline behind the C ABI (product code).
//
// zkc_zkey_load  parses a snarkjs-format Groth16 .zkey (the reference's proving_key.zkey format),
//                turns section 4 into CSR, uploads the bases and pre-shifts them for the MSM windows.
// zkc_prove_dev  witness (device, standard form) -> proof: buildABC -> 3 x (iNTT, coset shift, NTT) -> joinABC ->
//                5 MSMs -> blinding (a7) on the host with injectable (r, s).
// Mirrors snarkjs groth16.prove (ts_inputs/src/example.ts:358-362 via fullProve) / rapidsnark groth16_prover
// (zk_census_test.go:89).
#include "zkc_prover.h"
#include "zkc_fixedbase.h"
#include "zkc_hostparse.h"
#include <cstring>
#include <ctime>
#include <cstdio>
#include <algorithm>
#include <atomic>

using namespace zkc;

extern "C" __global__ void zkc_matvec_jds(const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, const Fr*, const Fr*, size_t, Fr*, int, uint32_t, const Fr*, size_t);
extern "C" __global__ void zkc_wtns_mont(const Fr*, size_t, Fr*, size_t, uint32_t);
extern "C" __global__ void zkc_pointwise_mul(Fr*, int);
extern "C" __global__ void zkc_join_abc(const Fr*, uint32_t*, int);
// buildABC for `nb` proofs on stream `mv` (zkc_ntt.hip): unit coefficients add the wire's Montgomery form, made once per wire in the lane's transform scratch (d_t is idle here:
// the pair runs in place, the two-transform form writes it afterwards) when it is large enough for nVars elements per proof
static void matvec_launch(zkc_zkey* zk, zkc_lane& L, const uint32_t* d_wtns0, int nb, hipStream_t mv) {
    const uint32_t n = zk->n, nv = zk->nVars;
    static const bool unit_off = [] { const char* e = getenv("ZKC_MATVEC_UNITS"); return e && atoi(e) == 0; }();
    const bool units = zk->n_unit_coeffs > 0 && nv <= 3 * (size_t)n && !unit_off;
    if (units) hipLaunchKernelGGL(zkc_wtns_mont, dim3((nv + 255) / 256, nb), dim3(256), 0, mv, (const Fr*)d_wtns0, (size_t)nv, L.d_t, 3 * (size_t)n, nv);
    hipLaunchKernelGGL(zkc_matvec_jds, dim3((2 * n + 63 * zk->nlong + 255) / 256, nb), dim3(256), 0, mv, zk->d_perm, zk->d_rowlen, zk->d_jdptr, zk->d_col, zk->d_val,
                       (const Fr*)d_wtns0, (size_t)nv, L.d_abc, (int)n, zk->nlong, units ? (const Fr*)L.d_t : (const Fr*)nullptr, 3 * (size_t)n);
    hipLaunchKernelGGL(zkc_pointwise_mul, dim3((n + 255) / 256, nb), dim3(256), 0, mv, L.d_abc, (int)n);
}
static constexpr uint32_t MATVEC_LONG = 16;      // rows with more coefficients are summed by a whole wave

namespace {
uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
G1Affine rd_g1(const uint8_t* p) { G1Affine a; memcpy(a.x.v, p, 32); memcpy(a.y.v, p + 32, 32); return a; }
G2Affine rd_g2(const uint8_t* p) { G2Affine a; memcpy(a.x.c0.v, p, 32); memcpy(a.x.c1.v, p + 32, 32); memcpy(a.y.c0.v, p + 64, 32); memcpy(a.y.c1.v, p + 96, 32); return a; }
Fr fr_root_of_unity(int logn) {
    uint32_t e[8]; for (int i = 0; i < 8; i++) e[i] = FrParams::p[i]; e[0] -= 1;
    for (int i = 0; i < 8; i++) e[i] = (e[i] >> 28) | (i < 7 ? e[i + 1] << 4 : 0);
    Fr g = fp_from_u32<FrParams>(5), w = Fr::one();
    for (int i = 255; i >= 0; i--) { w = w * w; if ((e[i >> 5] >> (i & 31)) & 1) w = w * g; }
    for (int i = 28; i > logn; i--) w = w * w;
    return w;
}
template <class T> int dmalloc(zkc_ctx* ctx, T** p, size_t count) { ZKC_HIP_CHECK(ctx, hipMalloc((void**)p, count * sizeof(T))); return ZKC_OK; }
void g1_to_std(uint8_t* out, const G1Affine& a) { uint32_t s[8]; fp_to_std<FqParams>(s, a.x); memcpy(out, s, 32); fp_to_std<FqParams>(s, a.y); memcpy(out + 32, s, 32); }
void g2_to_std(uint8_t* out, const G2Affine& a) {
    uint32_t s[8]; fp_to_std<FqParams>(s, a.x.c0); memcpy(out, s, 32); fp_to_std<FqParams>(s, a.x.c1); memcpy(out + 32, s, 32);
    fp_to_std<FqParams>(s, a.y.c0); memcpy(out + 64, s, 32); fp_to_std<FqParams>(s, a.y.c1); memcpy(out + 96, s, 32);
}
}  // namespace

uint32_t* zkc_get_template(zkc_ctx* ctx, int nLevels);     // zkc_api.hip: device template witness (nullptr on error)
int zkc_lane_streams(zkc_ctx* ctx, int l, bool with_red, zkc_ctx::LaneStreams* out);      // zkc_api.hip

// ---- fold check: for every proof and every foldable group, does the witness equal the template there? ----
// group g of a tree block = level g's non-control wires (g < n-1) ; group n-1 = the n2bOld block
extern "C" __global__ void __launch_bounds__(64)
zkc_fold_check(WitnessLayout L, const uint32_t* __restrict__ wtns, const uint32_t* __restrict__ tmpl, uint32_t* __restrict__ flags, int B) {
    const int n = L.n, g = blockIdx.x, tree = blockIdx.y, b = blockIdx.z;
    const int blk = tree == 0 ? L.off_census : L.off_sikver;
    int start, end;
    if (g < n - 1) {
        const int nctrl = (g == n - 3) + (g > 0 && g < n - 2) + 1;
        start = blk + L.lvl_off(g) + nctrl; end = blk + (g + 1 < n - 1 ? L.lvl_off(g + 1) : L.lvl_off(n - 1));
    } else { start = blk + L.off_n2bold; end = start + 253 + 127 + 133; }
    const uint4* w = reinterpret_cast<const uint4*>(wtns + (size_t)b * L.nWires * 8) + 2 * (size_t)start;
    const uint4* t = reinterpret_cast<const uint4*>(tmpl) + 2 * (size_t)start;
    uint32_t diff = 0;
    for (int i = threadIdx.x; i < 2 * (end - start); i += 64) { uint4 x = w[i], y = t[i]; diff |= (x.x ^ y.x) | (x.y ^ y.y) | (x.z ^ y.z) | (x.w ^ y.w); }
    const unsigned long long any = __ballot(diff != 0);
    if (threadIdx.x == 0) flags[((size_t)b * 2 + tree) * n + g] = any ? 1u : 0u;
}

// ---- [r3] the folding depths of a voter straight from its inputs: 1 + the index of its last non-zero sibling, per tree.  That is where SMTLevIns puts the leaf, and every level
// from there down carries the voter-independent trace.  A call of one or two voters reads these BEFORE its witness kernel has run, so that everything behind the witness can be
// enqueued while it runs; zkc_fold_check still compares the finished witness with the template, and prove_batch_finish refuses the call if the two disagree for an accepted voter. ----
extern "C" __global__ void __launch_bounds__(64)
zkc_input_depths(const uint32_t* __restrict__ inputs, int nInputs, int n, uint32_t* __restrict__ out) {
    const int b = blockIdx.x, tree = blockIdx.y;
    const uint32_t* sib = inputs + ((size_t)b * nInputs + 12 + (size_t)tree * n) * 8;
    int d = 0;
    for (int i = threadIdx.x; i < n; i += 64) {
        const uint4* q = reinterpret_cast<const uint4*>(sib + 8 * (size_t)i); const uint4 x = q[0], y = q[1];
        if (x.x | x.y | x.z | x.w | y.x | y.y | y.z | y.w) d = i + 1;
    }
    for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(d, m, 64); d = o > d ? o : d; }
    if (threadIdx.x == 0) out[2 * b + tree] = (uint32_t)d;
}

extern "C" void zkc_zkey_free(zkc_zkey* zk) {
    if (!zk) return;
    ZKC_LOCK(zk->ctx);
    (void)hipSetDevice(zk->ctx->device);
    (void)hipStreamSynchronize(zk->ctx->stream);
    void* ptrs[] = {zk->d_perm, zk->d_rowlen, zk->d_jdptr, zk->d_col, zk->d_val, zk->d_tw_fwd, zk->d_tw_inv, zk->d_tw_fwd29, zk->d_tw_inv29, zk->d_coset, zk->d_coset_br, zk->d_g1, zk->d_g2, zk->d_g2_29, zk->d_g2_29_lone, zk->d_g2_29_deep,
                    zk->d_tblDelta1, zk->d_tblAlpha1, zk->d_tblBeta1, zk->d_tblDelta2, zk->d_fb4, zk->d_fb4g2, zk->d_depths};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& kv : zk->fold.vmaps) if (kv.second.d) (void)hipFree(kv.second.d);
    for (void* q : {(void*)zk->fold.d_foldA, (void*)zk->fold.d_foldB1, (void*)zk->fold.d_foldC, (void*)zk->fold.d_foldB2}) if (q) (void)hipFree(q);
    if (zk->h_depths) (void)hipHostFree(zk->h_depths);
    for (auto& c : zk->call) { for (void* q : {(void*)c.d_rs, (void*)c.d_proofs, (void*)c.d_flags, (void*)c.d_status3}) if (q) (void)hipFree(q); if (c.h_flags) (void)hipHostFree(c.h_flags); if (c.h_out) (void)hipHostFree(c.h_out); if (c.h_rs) (void)hipHostFree(c.h_rs); if (c.h_xyzz) (void)hipHostFree(c.h_xyzz); if (c.h_early) (void)hipHostFree(c.h_early); if (c.d_xyzz) (void)hipFree(c.d_xyzz); for (hipEvent_t e : c.ev_done) if (e) (void)hipEventDestroy(e); for (hipEvent_t e : c.ev_chunk) (void)hipEventDestroy(e); }
    if (zk->ctx->lanes) for (int l = 0; l < zk->nlanes; l++) for (hipStream_t q : {zk->ctx->lanes[l].st, zk->ctx->lanes[l].st2, zk->ctx->lanes[l].fin, zk->ctx->lanes[l].red}) if (q) (void)hipStreamSynchronize(q);      // (lanes, streams and work space are the context's and stay)
    delete zk;
}

static size_t dev_free_bytes() { size_t f = 0, t = 0; if (hipMemGetInfo(&f, &t) != hipSuccess) { (void)hipGetLastError(); return 0; } return f; }
static int fold_prepare(zkc_zkey* zk);
static int fb4_prepare(zkc_zkey* zk);
// ---- the pipeline lanes: the CONTEXT's (zkc_ctx::lanes), shared by every key of the context ----
// what one lane must hold for `inflight` proofs of this key per pass
struct LaneNeed { size_t abc, p, fin, bs, e1, b1, e2, b2; int j1, j2; };
static LaneNeed lane_need(const zkc_zkey* zk, int inflight) {
    const uint32_t n = zk->n, nv = zk->nVars; const int NWS = msm_nw(zk->c_sec), NWB = msm_nw(zk->c_h);
    const size_t per_proof_entries = (size_t)NWS * 3 * nv + (size_t)NWB * n;
    const size_t per_proof_buckets = 3 * (size_t)msm_half(std::max(zk->c_sec, zk->c_deep)) + msm_half(zk->c_h);      // (a deep pass has fewer entries and more buckets)
    LaneNeed nd; nd.abc = 3 * (size_t)n * inflight; nd.p = 8 * (size_t)n * inflight; nd.fin = finalize_scratch_bytes(inflight); nd.bs = 2 * 2 * 8 * (size_t)nv;
    nd.e1 = per_proof_entries * inflight; nd.b1 = per_proof_buckets * inflight; nd.j1 = 4 * inflight;
    nd.e2 = (size_t)NWS * nv * inflight; nd.b2 = (size_t)msm_half(std::max(zk->c_sec, zk->c_deep)) * inflight; nd.j2 = inflight;
    return nd;
}
static bool lane_holds(const zkc_lane& L, const LaneNeed& nd) {
    return L.cap_abc >= nd.abc && L.cap_p >= nd.p && L.cap_fin >= nd.fin && L.cap_bs >= nd.bs && L.w1.max_entries >= nd.e1 && L.w1.max_buckets >= nd.b1 && L.w1.max_jobs >= nd.j1 &&
           L.w2.max_entries >= nd.e2 && L.w2.max_buckets >= nd.b2 && L.w2.max_jobs >= nd.j2;
}
static void lane_release(zkc_lane& L) {      // the buffers only: streams, events and the pass counter stay
    for (void** q : {(void**)&L.d_abc, (void**)&L.d_t, (void**)&L.d_p, &L.d_fin, (void**)&L.d_bs}) if (*q) { (void)hipFree(*q); *q = nullptr; }
    msm_work_free(L.w1); msm_work_free(L.w2);
    L.cap_abc = L.cap_p = L.cap_fin = L.cap_bs = 0; L.bytes = 0;
}
static zkc_lane* ctx_lanes(zkc_ctx* ctx) { if (!ctx->lanes) ctx->lanes = new zkc_lane[MAX_LANES]; return ctx->lanes; }
static size_t ctx_work_bytes(const zkc_ctx* ctx) { size_t t = 0; if (ctx->lanes) for (int l = 0; l < MAX_LANES; l++) t += ctx->lanes[l].bytes; return t; }
void zkc_ctx_lanes_destroy(zkc_ctx* ctx) {
    if (!ctx->lanes) return;
    for (int l = 0; l < MAX_LANES; l++) {
        zkc_lane& L = ctx->lanes[l];
        lane_release(L);
        for (hipEvent_t e : {L.ev_msm, L.ev_msm2, L.ev_sorted, L.ev_ntt, L.ev_mv, L.ev_acc, L.ev_fin[0], L.ev_fin[1], L.ev_red}) if (e) (void)hipEventDestroy(e);
    }
    delete[] ctx->lanes; ctx->lanes = nullptr;
}
// streams and events of lane l, once per context
static int lane_init(zkc_ctx* ctx, int l) {
    zkc_lane& L = ctx_lanes(ctx)[l];
    if (L.made) return ZKC_OK;
    static const bool red_wanted = [] { const char* e = getenv("ZKC_REDUCE_STREAM"); return e && atoi(e) == 1; }();
    zkc_ctx::LaneStreams ls; int rc = zkc_lane_streams(ctx, l, red_wanted, &ls); if (rc) return rc;
    L.st = ls.st; L.st2 = ls.st2; L.fin = ls.fin; L.red = ls.red;
    for (hipEvent_t* e : {&L.ev_msm, &L.ev_msm2, &L.ev_sorted, &L.ev_ntt, &L.ev_mv, &L.ev_acc, &L.ev_fin[0], &L.ev_fin[1], &L.ev_red}) if (!*e) ZKC_HIP_CHECK(ctx, hipEventCreateWithFlags(e, hipEventDisableTiming));
    L.made = true; return ZKC_OK;
}
// the streams a pass of THIS key runs on in lane L: the lane's own, or -- a key loaded with ZKC_SERIAL_STREAMS=1 (measurement only: every stage of a pass on the context's one
// stream, so that the per-category HIP-event brackets of zkc_profile_* are ISOLATED kernel times; the pipeline's overlap is gone, the proofs are the same bytes) -- ctx->stream
struct LaneSt { hipStream_t st, st2, fin, red; };
static LaneSt lane_st(const zkc_zkey* zk, const zkc_lane& L) { hipStream_t c = zk->ctx->stream; return zk->serial_streams ? LaneSt{c, c, c, c} : LaneSt{L.st, L.st2, L.fin, L.red}; }
// Grows lane l of the key's context until it holds `inflight` proofs of this key per pass (grow only, never below what another key of the context asked for).  The caller holds
// the context lock.  Footprint per proof in flight at nLevels = 160: abc + NTT scratch 2 x 12 MiB, p 4 MiB, MSM entries 59 MB, bucket / segment arrays and partial sums ~60 MB.
static int lane_ensure(zkc_zkey* zk, int l, int inflight) {
    zkc_ctx* ctx = zk->ctx; int rc;
    if ((rc = lane_init(ctx, l))) return rc;
    zkc_lane& L = ctx->lanes[l];
    inflight = std::max(1, std::min(inflight, zk->max_inflight));
    if (lane_holds(L, lane_need(zk, inflight))) return ZKC_OK;
    // grow in two steps only: a caller of one to four proofs at a time reserves four proofs' work space (0.6 GB at nLevels = 160), anything larger the full pass (14 GB for 96
    // proofs): every growth frees and re-allocates the lane, which a burst of growing batches would otherwise pay five or six times, hundreds of milliseconds each
    inflight = std::min(inflight <= 4 ? 4 : zk->max_inflight, zk->max_inflight);
    LaneNeed nd = lane_need(zk, inflight);
    nd.abc = std::max(nd.abc, L.cap_abc); nd.p = std::max(nd.p, L.cap_p); nd.fin = std::max(nd.fin, L.cap_fin); nd.bs = std::max(nd.bs, L.cap_bs);      // what other keys of the context needed stays
    nd.e1 = std::max(nd.e1, L.w1.max_entries); nd.b1 = std::max(nd.b1, L.w1.max_buckets); nd.j1 = std::max(nd.j1, L.w1.max_jobs);
    nd.e2 = std::max(nd.e2, L.w2.max_entries); nd.b2 = std::max(nd.b2, L.w2.max_buckets); nd.j2 = std::max(nd.j2, L.w2.max_jobs);
    // the old work space goes first: from here until every allocation has succeeded the lane has NO work space and says so (capacities zero), so a failure leaves a lane that
    // re-allocates on its next call instead of launching on freed buffers
    for (hipStream_t q : {L.st, L.st2, L.fin, L.red}) if (q) ZKC_HIP_CHECK(ctx, hipStreamSynchronize(q));
    ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    lane_release(L);
    const size_t free_before = dev_free_bytes();
    const char* fail_at = getenv("ZKC_TEST_FAIL_ALLOC");          // test hook: pretend the allocation for this many proofs in flight (or more) fails
    rc = (fail_at && inflight >= atoi(fail_at)) ? zkc_fail(ctx, ZKC_ERR_HIP, "lanes_ensure: allocation failure injected by ZKC_TEST_FAIL_ALLOC") : ZKC_OK;
    if (!rc) rc = dmalloc(ctx, &L.d_abc, nd.abc);
    if (!rc) rc = dmalloc(ctx, &L.d_t, nd.abc);
    if (!rc) rc = dmalloc(ctx, &L.d_p, nd.p);
    if (!rc && hipMalloc(&L.d_fin, nd.fin) != hipSuccess) rc = zkc_fail(ctx, ZKC_ERR_HIP, "lanes_ensure: hipMalloc failed (blinding scratch)");
    if (!rc) rc = dmalloc(ctx, &L.d_bs, nd.bs);
    if (!rc) rc = msm_work_alloc(ctx, L.w1, nd.e1, nd.b1, nd.j1, false);
    if (!rc) rc = msm_work_alloc(ctx, L.w2, nd.e2, nd.b2, nd.j2, true);
    if (rc) { (void)hipGetLastError(); lane_release(L); return rc; }
    L.cap_abc = nd.abc; L.cap_p = nd.p; L.cap_fin = nd.fin; L.cap_bs = nd.bs;
    { const size_t f = dev_free_bytes(); L.bytes = free_before > f ? free_before - f : 0; }
    return ZKC_OK;
}
// lanes [first, first + count) of the key's context, each for `inflight` proofs per pass
static int lanes_ensure(zkc_zkey* zk, int inflight, int first = 0, int count = -1) {
    if (count < 0) count = zk->nlanes - first;
    for (int l = first; l < first + count && l < zk->nlanes; l++) { const int rc = lane_ensure(zk, l, inflight); if (rc) return rc; }
    return ZKC_OK;
}
extern "C" int zkc_zkey_load(zkc_ctx* ctx, const void* zkey_bytes, size_t len, zkc_zkey** out) { return zkc::zkey_load_opts(ctx, zkey_bytes, len, 0, 0, out); }
size_t zkc::zkey_device_bytes(const zkc_zkey* zk, size_t* tables, size_t* work) {
    if (!zk) return 0;
    const size_t w = ctx_work_bytes(zk->ctx);
    if (tables) *tables = zk->bytes_tables; if (work) *work = w;
    return zk->bytes_tables + w;
}
int zkc::zkey_load_opts(zkc_ctx* ctx, const void* zkey_bytes, size_t len, int opt_lanes, int opt_inflight, zkc_zkey** out) {
    if (!ctx || !zkey_bytes || !out) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "zkc_zkey_load: bad argument");
    ZKC_LOCK(ctx);
    ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t free_at_entry = dev_free_bytes(), work_at_entry = ctx_work_bytes(ctx);
    const uint8_t* buf = (const uint8_t*)zkey_bytes;
    // every length and index of the file is validated by the host-only parser (zkc_hostparse.h, also built under ASan/UBSan by the tests)
    parse::BinSections bs; parse::ZkeyHeader zh; std::string perr;
    if (!parse::binfile_sections(buf, len, "zkey", 1, bs, perr) || !parse::zkey_check(bs, zh, perr)) return zkc_fail(ctx, ZKC_ERR_FORMAT, perr);
    const uint8_t* const* sec = bs.sec;
    const uint8_t* h = sec[2];
    zkc_zkey* zk = new zkc_zkey(); zk->ctx = ctx;
    zk->nVars = zh.nVars; zk->nPub = zh.nPub; zk->n = zh.n; zk->logn = zh.logn; zk->nCoeffs = zh.nCoeffs;
    parse::sha256(buf, len, zk->sha256); parse::zkey_fingerprint(buf, len, bs, zk->fingerprint);
    const uint32_t n = zk->n, nv = zk->nVars, np = zk->nPub, nc = nv - np - 1;
    zk->alpha1 = rd_g1(h + 84); zk->beta1 = rd_g1(h + 148); zk->beta2 = rd_g2(h + 212); zk->gamma2 = rd_g2(h + 340);
    zk->delta1 = rd_g1(h + 468); zk->delta2 = rd_g2(h + 532);
    for (uint32_t i = 0; i <= np; i++) zk->ic.push_back(rd_g1(sec[3] + 64ull * i));
    {   // wires whose A / B / C polynomial is zero have the point at infinity as base: they never enter an MSM
        auto zero64 = [](const uint8_t* q) { for (int i = 0; i < 64; i++) if (q[i]) return false; return true; };
        zk->fold.infA.resize(nv); zk->fold.infB.resize(nv); zk->fold.infC.assign(nv, 0);
        for (uint32_t i = 0; i < nv; i++) { zk->fold.infA[i] = zero64(sec[5] + 64ull * i); zk->fold.infB[i] = zero64(sec[6] + 64ull * i); if (i > np) zk->fold.infC[i] = zero64(sec[8] + 64ull * (i - np - 1)); }
    }
    // a key whose shape is ZkFranchiseProofCircuit(nLevels) can use the voter-independent witness template
    if (np == 8 && !getenv("ZKC_NO_FOLD")) for (int nl = 3; nl <= 253; nl++) if ((uint32_t)WitnessLayout::make(nl).nWires == nv) { zk->nLevels = nl; break; }
    int rc = ZKC_OK;
    auto bail = [&](int code) { zkc_zkey_free(zk); return code; };
    // from here on every failure releases the half-built key: no early return without bail()
#define ZKC_HIP_BAIL(call)                                                                                      \
    do { hipError_t _e = (call); if (_e != hipSuccess) return bail(zkc_fail(ctx, ZKC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e))); } while (0)
#define ZKC_UP(dst, src, bytes) ZKC_HIP_BAIL(hipMemcpy((dst), (src), (bytes), hipMemcpyHostToDevice))
    // ---- section 4 -> CSR (row = matrix * n + constraint) ----
    {
        std::vector<uint32_t> rowptr(2 * (size_t)n + 1, 0), col(zk->nCoeffs); std::vector<Fr> val(zk->nCoeffs);
        const uint8_t* c = sec[4] + 4;
        for (uint32_t i = 0; i < zk->nCoeffs; i++) {
            uint32_t m = rd32(c + 44ull * i), cc = rd32(c + 44ull * i + 4), s = rd32(c + 44ull * i + 8);
            rowptr[(size_t)m * n + cc + 1]++;
        }
        for (size_t r = 0; r < 2 * (size_t)n; r++) rowptr[r + 1] += rowptr[r];
        std::vector<uint32_t> fill(rowptr.begin(), rowptr.end() - 1);
        for (uint32_t i = 0; i < zk->nCoeffs; i++) {
            uint32_t m = rd32(c + 44ull * i), cc = rd32(c + 44ull * i + 4), s = rd32(c + 44ull * i + 8);
            uint32_t k = fill[(size_t)m * n + cc]++;
            col[k] = s; memcpy(val[k].v, c + 44ull * i + 12, 32);
        }
        // jagged-diagonal order: rows by decreasing length; slot jdptr[k] + r holds the k-th coefficient of the r-th longest row
        const size_t nrows = 2 * (size_t)n;
        std::vector<uint32_t> perm(nrows), rowlen(nrows);
        for (size_t r = 0; r < nrows; r++) perm[r] = (uint32_t)r;
        std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return rowptr[a + 1] - rowptr[a] > rowptr[b + 1] - rowptr[b]; });
        for (size_t r = 0; r < nrows; r++) rowlen[r] = rowptr[perm[r] + 1] - rowptr[perm[r]];
        const uint32_t maxlen = nrows ? rowlen[0] : 0;
        zk->nlong = 0; while (zk->nlong < nrows && rowlen[zk->nlong] > MATVEC_LONG) zk->nlong++;
        std::vector<uint32_t> jdptr(maxlen + 1, 0);
        { size_t live = nrows; for (uint32_t k = 0; k < maxlen; k++) { while (live > 0 && rowlen[live - 1] <= k) live--; jdptr[k + 1] = jdptr[k] + (uint32_t)live; } }
        std::vector<uint32_t> jcol(zk->nCoeffs); std::vector<Fr> jval(zk->nCoeffs);
        // [r4] +1 and -1 (stored, like every coefficient, times R^2) are marked in the top bits of the column word: zkc_matvec_jds adds or subtracts the wire's Montgomery form
        // for them instead of multiplying (276 k of the census circuit's 463 k coefficients)
        if (nv >= (1u << 30)) return bail(zkc_fail(ctx, ZKC_ERR_FORMAT, "zkey too large for 30-bit wire indices"));
        Fr one_r2, neg_r2; for (int i = 0; i < 8; i++) one_r2.v[i] = FrParams::r2[i];
        neg_r2 = Fr::zero() - one_r2;
        zk->n_unit_coeffs = 0;
        for (size_t r = 0; r < nrows; r++) for (uint32_t k = 0; k < rowlen[r]; k++) {
            const size_t dst = (size_t)jdptr[k] + r, src = (size_t)rowptr[perm[r]] + k;
            uint32_t c = col[src];
            if (val[src] == one_r2) { c |= 0x80000000u; zk->n_unit_coeffs++; } else if (val[src] == neg_r2) { c |= 0xc0000000u; zk->n_unit_coeffs++; }
            jcol[dst] = c; jval[dst] = val[src];
        }
        if ((rc = dmalloc(ctx, &zk->d_perm, nrows)) || (rc = dmalloc(ctx, &zk->d_rowlen, nrows)) || (rc = dmalloc(ctx, &zk->d_jdptr, jdptr.size())) ||
            (rc = dmalloc(ctx, &zk->d_col, jcol.size() + 1)) || (rc = dmalloc(ctx, &zk->d_val, jval.size() + 1))) return bail(rc);
        ZKC_UP(zk->d_perm, perm.data(), nrows * 4); ZKC_UP(zk->d_rowlen, rowlen.data(), nrows * 4); ZKC_UP(zk->d_jdptr, jdptr.data(), jdptr.size() * 4);
        ZKC_UP(zk->d_col, jcol.data(), jcol.size() * 4);
        ZKC_UP(zk->d_val, jval.data(), jval.size() * sizeof(Fr));
    }
    // ---- twiddles and the coset/1-over-n scale ----
    {
        const Fr w = fr_root_of_unity((int)zk->logn), g = fr_root_of_unity((int)zk->logn + 1);
        const Fr wi = fp_inv<FrParams>(w), ninv = fp_inv<FrParams>(fp_from_u32<FrParams>(n));
        std::vector<Fr> f(n / 2), b(n / 2), cs(n);
        f[0] = b[0] = Fr::one(); for (uint32_t i = 1; i < n / 2; i++) { f[i] = f[i - 1] * w; b[i] = b[i - 1] * wi; }
        cs[0] = ninv; for (uint32_t i = 1; i < n; i++) cs[i] = cs[i - 1] * g;
        if ((rc = dmalloc(ctx, &zk->d_tw_fwd, n / 2)) || (rc = dmalloc(ctx, &zk->d_tw_inv, n / 2)) || (rc = dmalloc(ctx, &zk->d_coset, n))) return bail(rc);
        ZKC_UP(zk->d_tw_fwd, f.data(), f.size() * sizeof(Fr));
        ZKC_UP(zk->d_tw_inv, b.data(), b.size() * sizeof(Fr));
        ZKC_UP(zk->d_coset, cs.data(), cs.size() * sizeof(Fr));
        if ((rc = ntt_make_tw29(ctx, zk->d_tw_fwd, n / 2, &zk->d_tw_fwd29)) || (rc = ntt_make_tw29(ctx, zk->d_tw_inv, n / 2, &zk->d_tw_inv29))) return bail(rc);
        if ((rc = ntt_bitrev_table(ctx, zk->d_coset, &zk->d_coset_br, (int)zk->logn))) return bail(rc);
    }
    // ---- bases: one G1 array [A | B1 | C | H] and one G2 array [B2]; window 0 = the zkey points as stored (affine,
    //      Montgomery), windows 1..19 pre-shifted on the device ----
    // [r4] windows per key (msm_c_for, zkc_prover.h: the best window grows with the number of scalars):
    auto c_for = [](size_t W) { return msm_c_for(W); };
    //   sections of a census key : 12 -- what folding leaves in a voter's MSMs is 8-11 k wires per section (13 measured equal) -- plus the deep tables below
    //   sections of any other key: by its wire count;   H: 17 for a census key (16 measured 2.7 % behind in the census pass: 3097 / 3106 against 3190 / 3174 proofs/s), else by the domain size, never below the sections'
    //   (msm_pass wants the jobs with the larger window first, and the H jobs are first)
    { const char* e_c = getenv("ZKC_C_SECTIONS"); zk->c_sec = e_c ? std::max(8, std::min(atoi(e_c), MSM_C_BIG)) : zk->nLevels >= 0 ? MSM_C_SMALL : c_for(nv); }
    { const char* e_h = getenv("ZKC_C_H"); zk->c_h = e_h ? std::max(8, std::min(atoi(e_h), MSM_C_BIG)) : zk->nLevels >= 0 ? MSM_C_BIG : std::max(c_for(n), n >= 12000 ? 15 : 12); }      // (H's scalars are all full width: at 2^14 points 15 bits measured 3 % ahead of 13, 17 5 % behind)
    const int NWS = msm_nw(zk->c_sec);
    // (ZKC_DEEP_TABLES=0: not built; =2: built for a census key of any size -- with ZKC_DEEP_WIRES=1 the switch test drives the deep path at nLevels = 10)
    { const char* e_d = getenv("ZKC_DEEP_TABLES"); const int dt = e_d ? atoi(e_d) : 1; const char* e_c = getenv("ZKC_C_DEEP");
      const int cd = e_c ? std::max(13, std::min(atoi(e_c), MSM_C_BIG)) : c_for(dt == 2 ? (size_t)40000 : (size_t)nv * 4 / 5);        // a deep voter keeps most of the key's wires
      zk->c_deep = (zk->nLevels >= 0 && (nv >= (1u << 16) || dt == 2) && zk->c_sec < cd && dt != 0) ? cd : 0; }
    zk->c_h = std::max(zk->c_h, std::max(zk->c_sec, zk->c_deep));
    const int NWB = msm_nw(zk->c_h);
    zk->offA = 0; zk->offB1 = NWS * nv; zk->offC = 2 * NWS * nv; zk->offH = 2 * NWS * nv + NWS * nc;
    const int NWD = zk->c_deep ? msm_nw(zk->c_deep) : 0;
    zk->offA_deep = zk->offH + NWB * n; zk->offB1_deep = zk->offA_deep + NWD * nv; zk->offC_deep = zk->offB1_deep + NWD * nv;
    const size_t g1_points = (size_t)NWS * (2 * (size_t)nv + nc) + (size_t)NWB * n + (size_t)NWD * (2 * (size_t)nv + nc);
    if (g1_points >= (1ull << 31)) return bail(zkc_fail(ctx, ZKC_ERR_FORMAT, "zkey too large for 31-bit point indices"));
    if ((rc = dmalloc(ctx, &zk->d_g1, g1_points)) || (rc = dmalloc(ctx, &zk->d_g2, (size_t)NWS * nv))) return bail(rc);
    ZKC_UP(zk->d_g1 + zk->offA, sec[5], 64ull * nv); ZKC_UP(zk->d_g1 + zk->offB1, sec[6], 64ull * nv);
    ZKC_UP(zk->d_g1 + zk->offC, sec[8], 64ull * nc); ZKC_UP(zk->d_g1 + zk->offH, sec[9], 64ull * n);
    ZKC_UP(zk->d_g2, sec[7], 128ull * nv);
    if ((rc = msm_precompute_g1(ctx, nv, zk->d_g1 + zk->offA, zk->c_sec)) || (rc = msm_precompute_g1(ctx, nv, zk->d_g1 + zk->offB1, zk->c_sec)) ||
        (rc = msm_precompute_g1(ctx, nc, zk->d_g1 + zk->offC, zk->c_sec)) || (rc = msm_precompute_g1(ctx, n, zk->d_g1 + zk->offH, zk->c_h)) ||
        (rc = msm_precompute_g2(ctx, nv, zk->d_g2, zk->c_sec))) return bail(rc);
    if ((rc = dmalloc(ctx, &zk->d_g2_29, 60 * (size_t)NWS * nv)) || (rc = msm_g2_table29(ctx, zk->d_g2, zk->d_g2_29, (size_t)NWS * nv))) return bail(rc);
    if (zk->c_deep) {   // the sections again for c_deep-bit windows (at nLevels = 160: 15 bits, 17 x 64 B per wire and G1 section, 17 x 240 B for G2: 0.6 GB beside the 0.8 GB of the 12-bit tables)
        ZKC_UP(zk->d_g1 + zk->offA_deep, sec[5], 64ull * nv); ZKC_UP(zk->d_g1 + zk->offB1_deep, sec[6], 64ull * nv); ZKC_UP(zk->d_g1 + zk->offC_deep, sec[8], 64ull * nc);
        if ((rc = msm_precompute_g1(ctx, nv, zk->d_g1 + zk->offA_deep, zk->c_deep)) || (rc = msm_precompute_g1(ctx, nv, zk->d_g1 + zk->offB1_deep, zk->c_deep)) ||
            (rc = msm_precompute_g1(ctx, nc, zk->d_g1 + zk->offC_deep, zk->c_deep))) return bail(rc);
        G2Affine* tmp = nullptr;
        if ((rc = dmalloc(ctx, &tmp, (size_t)NWD * nv))) return bail(rc);
        hipError_t e2 = hipMemcpyAsync(tmp, zk->d_g2, 128ull * nv, hipMemcpyDeviceToDevice, ctx->stream);
        if (e2 == hipSuccess) rc = msm_precompute_g2(ctx, nv, tmp, zk->c_deep);
        if (e2 == hipSuccess && !rc) rc = dmalloc(ctx, &zk->d_g2_29_deep, 60 * (size_t)NWD * nv);
        if (e2 == hipSuccess && !rc) rc = msm_g2_table29(ctx, tmp, zk->d_g2_29_deep, (size_t)NWD * nv);
        if (e2 == hipSuccess && !rc) e2 = hipStreamSynchronize(ctx->stream);
        (void)hipFree(tmp);
        if (rc) return bail(rc);
        if (e2 != hipSuccess) return bail(zkc_fail(ctx, ZKC_ERR_HIP, std::string("second section tables: ") + hipGetErrorString(e2)));
    }
    {   // [r3] the 8-bit-window G2 table of the lone-proof path (MSM_C_G2_LONE): shifted in a temporary affine table, kept in radix 2^29 only
        const char* e_lone = getenv("ZKC_G2_LONE_TABLE");
        if (!(e_lone && atoi(e_lone) == 0) && (zk->nLevels >= 0 || nv < (1u << 16))) {      // (a key of 2^16 wires and more that is not the census circuit: 32 x 240 B per wire for a latency path its proofs are too large to notice)
            constexpr int NWL = msm_nw(MSM_C_G2_LONE);
            G2Affine* tmp = nullptr;
            if ((rc = dmalloc(ctx, &tmp, (size_t)NWL * nv))) return bail(rc);
            hipError_t e2 = hipMemcpyAsync(tmp, zk->d_g2, 128ull * nv, hipMemcpyDeviceToDevice, ctx->stream);
            if (e2 == hipSuccess) rc = msm_precompute_g2(ctx, nv, tmp, MSM_C_G2_LONE);
            if (e2 == hipSuccess && !rc) rc = dmalloc(ctx, &zk->d_g2_29_lone, 60 * (size_t)NWL * nv);
            if (e2 == hipSuccess && !rc) rc = msm_g2_table29(ctx, tmp, zk->d_g2_29_lone, (size_t)NWL * nv);
            if (e2 == hipSuccess && !rc) e2 = hipStreamSynchronize(ctx->stream);
            (void)hipFree(tmp);
            if (rc) return bail(rc);
            if (e2 != hipSuccess) return bail(zkc_fail(ctx, ZKC_ERR_HIP, std::string("lone-proof G2 table: ") + hipGetErrorString(e2)));
        }
    }
    // ---- work buffers: up to `max_inflight` proofs share one MSM pipeline pass; the buffers themselves are sized by lanes_ensure() for
    //      the number of proofs a call actually puts in flight (a single-proof caller does not reserve the work space of 96) ----
    const char* e_inf = getenv("ZKC_INFLIGHT");
    // [r5] a census key: passes of 64 proofs over four lanes (below); any other key: 96 (cut down by its size, next block) on one lane as before
    zk->max_inflight = opt_inflight > 0 ? std::min(opt_inflight, MSM_MAX_JOBS / 4) : e_inf ? std::max(1, std::min(atoi(e_inf), MSM_MAX_JOBS / 4)) : zk->nLevels >= 0 ? 64 : 96;
    {   // [r4] the pass is sized for the census key (96 proofs of 7.4 M (scalar, window) entries unfolded): a larger circuit puts fewer proofs in flight -- the same ~0.7 G entries per
        // pass, which also keeps every entry index of a pass inside 32 bits (MsmJob::ent_off) -- 11 at a 2^20 domain, where ONE proof is 60 M additions and fills the chip
        const size_t per_proof_entries = (size_t)NWS * 3 * nv + (size_t)NWB * n, census_pass = 96ull * (22ull * 3 * 82754 + 15ull * 131072);
        const int fit = (int)std::max<size_t>(1, census_pass / std::max<size_t>(per_proof_entries, 1));
        zk->max_inflight = std::min(zk->max_inflight, fit);
    }      // same box, batch 1024: 64 -> 1955, 80 -> 1985, 96 -> 2015, 112 -> 2000, 128 -> 2022 proofs/s; round 2: 96 -> 3010, 114 -> 3004, 128 -> 3031
    // ZKC_LANES=2 lets two lanes take alternate passes; measured no gain in round 1 (the GPU is already saturated) and -4 % at the end of round 2 (3005
    // against 3142 proofs/s on one box: the second lane has no buildABC prefetch), so one lane is the default
    // [r5] up to MAX_LANES; the proving service asks for one lane per worker (opt_lanes) and keeps every call on its worker's lane
    // [r5''] ... and, for a census key, the default of the batch entry points too: four lanes of 64-proof passes, the passes of a call rotating over them.  With every stream on
    // a hardware queue of its own (GPU_MAX_HW_QUEUES, zkc_api.hip) and the lanes made only when a call reaches them, the overlap that rounds 1-4 could not find is there:
    // alternating on one box, (1 lane, 96) 3243 / 3235 proofs/s, (4, 96) 3291, (4, 64) 3314 / 3308 (+2.2 %), (4, 48) 3251, (3, 64) 3179, (2, 96) 3146 / 3155.  A lone proof
    // still touches lane 0 only (0.6 GB); a 1 024-voter call all four (37 GB; ZKC_LANES=1 ZKC_INFLIGHT=96: 14 GB and the old shape).
    { const char* e_l = getenv("ZKC_LANES"); zk->nlanes = opt_lanes > 0 ? std::min(opt_lanes, (int)MAX_LANES) : e_l ? std::max(1, std::min(atoi(e_l), (int)MAX_LANES)) : zk->nLevels >= 0 ? (int)MAX_LANES : 1; }
    zk->serial_streams = getenv("ZKC_SERIAL_STREAMS") != nullptr;          // (lane_st)
    // the lanes are the context's: lane 0 is made (or found) now with room for a lone caller, the others when a call first lands on them
    if ((rc = lanes_ensure(zk, 1, 0, 1))) return bail(rc);
    {   // fixed-base tables for the blinding step (delta1, alpha1, beta1 in G1; delta2 in G2)
        FixedBase<Fq> td(zk->delta1), ta(zk->alpha1), tb(zk->beta1); FixedBase<Fq2> t2(zk->delta2);
        if ((rc = dmalloc(ctx, &zk->d_tblDelta1, td.tab.size())) || (rc = dmalloc(ctx, &zk->d_tblAlpha1, ta.tab.size())) ||
            (rc = dmalloc(ctx, &zk->d_tblBeta1, tb.tab.size())) || (rc = dmalloc(ctx, &zk->d_tblDelta2, t2.tab.size()))) return bail(rc);
        ZKC_UP(zk->d_tblDelta1, td.tab.data(), td.tab.size() * sizeof(G1Affine)); ZKC_UP(zk->d_tblAlpha1, ta.tab.data(), ta.tab.size() * sizeof(G1Affine));
        ZKC_UP(zk->d_tblBeta1, tb.tab.data(), tb.tab.size() * sizeof(G1Affine)); ZKC_UP(zk->d_tblDelta2, t2.tab.data(), t2.tab.size() * sizeof(G2Affine));
    }
    ZKC_HIP_BAIL(hipStreamSynchronize(ctx->stream));
    // the folding tables of the voter-independent witness part are part of the key's one-time cost, not of the first proof
    if (zk->nLevels >= 0 && (rc = fold_prepare(zk))) return bail(rc);
    if ((rc = fb4_prepare(zk))) return bail(rc);
    { const size_t f = dev_free_bytes(), grown = ctx_work_bytes(ctx) > work_at_entry ? ctx_work_bytes(ctx) - work_at_entry : 0; zk->bytes_tables = free_at_entry > f + grown ? free_at_entry - f - grown : 0; }
    *out = zk;
    return ZKC_OK;
#undef ZKC_UP
#undef ZKC_HIP_BAIL
}

// the work space of a full pass now, instead of at the first call that brings more than four proofs (the proving service: a key that serves concurrent callers will see
// such a call, and growing then stalls the device for the free + re-allocation of ~6 GB in the middle of the first burst)
int zkc::prove_reserve(zkc_zkey* zk, int inflight) {
    if (!zk) return ZKC_ERR_BAD_ARG;
    ZKC_LOCK(zk->ctx);
    ZKC_HIP_CHECK(zk->ctx, hipSetDevice(zk->ctx->device));
    return lanes_ensure(zk, inflight);      // every lane of the key
}
extern "C" int zkc_zkey_pass_info(const zkc_zkey* zk, int* pass_size, int* lanes) {
    if (!zk) return ZKC_ERR_BAD_ARG;
    if (pass_size) *pass_size = zk->max_inflight; if (lanes) *lanes = zk->nlanes;
    return ZKC_OK;
}
extern "C" int zkc_zkey_info(const zkc_zkey* zk, uint32_t* nVars, uint32_t* nPublic, uint32_t* domainSize) {
    if (!zk) return ZKC_ERR_BAD_ARG;
    if (nVars) *nVars = zk->nVars; if (nPublic) *nPublic = zk->nPub; if (domainSize) *domainSize = zk->n;
    return ZKC_OK;
}

// ---- constant folding tables: per-level sums of template_value * base for every section, then suffix sums ----
static std::vector<std::pair<int, int>> fold_group_ranges(const WitnessLayout& L, int tree) {   // [start, end) wire ranges, groups 0..n-1
    std::vector<std::pair<int, int>> r; const int n = L.n, blk = tree == 0 ? L.off_census : L.off_sikver;
    for (int g = 0; g < n - 1; g++) {
        const int nctrl = (g == n - 3) + (g > 0 && g < n - 2) + 1;
        r.push_back({blk + L.lvl_off(g) + nctrl, blk + (g + 1 < n - 1 ? L.lvl_off(g + 1) : L.lvl_off(n - 1))});
    }
    r.push_back({blk + L.off_n2bold, blk + L.off_n2bold + 253 + 127 + 133});
    return r;
}
template <class X> static void suffix_sums(std::vector<X>& suf, const X* g, int n) {   // suf[D] = sum_{k >= D} g[k], k < n-1 ; suf[n-1] = inf
    suf.assign(n, X::inf());
    for (int D = n - 2; D >= 0; D--) suf[D] = xyzz_add(suf[D + 1], g[D]);
}
template <class PT> static int fold_upload(zkc_ctx* ctx, PT** d, const std::vector<PT>& b, const std::vector<PT>* suf) {
    std::vector<PT> t; t.push_back(b[0]); t.insert(t.end(), suf[0].begin(), suf[0].end()); t.insert(t.end(), suf[1].begin(), suf[1].end());
    ZKC_HIP_CHECK(ctx, hipMalloc((void**)d, t.size() * sizeof(PT)));
    ZKC_HIP_CHECK(ctx, hipMemcpy(*d, t.data(), t.size() * sizeof(PT), hipMemcpyHostToDevice));
    return ZKC_OK;
}
static int fold_prepare(zkc_zkey* zk) {
    if (zk->fold.ready) return ZKC_OK;
    zkc_ctx* ctx = zk->ctx; const WitnessLayout L = WitnessLayout::make(zk->nLevels); const int n = L.n;
    uint32_t* tmpl = zkc_get_template(ctx, zk->nLevels); if (!tmpl) return ZKC_ERR_HIP;
    std::vector<uint32_t> wires, gstart;
    for (int tree = 0; tree < 2; tree++) for (auto& rg : fold_group_ranges(L, tree)) { gstart.push_back((uint32_t)wires.size()); for (int w = rg.first; w < rg.second; w++) wires.push_back((uint32_t)w); }
    gstart.push_back((uint32_t)wires.size());
    const uint32_t ng = 2 * n;
    uint32_t *d_w = nullptr, *d_g = nullptr;
    ZKC_HIP_CHECK(ctx, hipMalloc((void**)&d_w, wires.size() * 4)); ZKC_HIP_CHECK(ctx, hipMalloc((void**)&d_g, gstart.size() * 4));
    ZKC_HIP_CHECK(ctx, hipMemcpy(d_w, wires.data(), wires.size() * 4, hipMemcpyHostToDevice));
    ZKC_HIP_CHECK(ctx, hipMemcpy(d_g, gstart.data(), gstart.size() * 4, hipMemcpyHostToDevice));
    std::vector<G1XYZZ> gA(ng), gB1(ng), gC(ng); std::vector<G2XYZZ> gB2(ng);
    int rc;
    if ((rc = fold_group_sums_g1(ctx, zk->d_g1 + zk->offA, tmpl, d_w, (uint32_t)wires.size(), 0, d_g, ng, gA.data())) ||
        (rc = fold_group_sums_g1(ctx, zk->d_g1 + zk->offB1, tmpl, d_w, (uint32_t)wires.size(), 0, d_g, ng, gB1.data())) ||
        (rc = fold_group_sums_g1(ctx, zk->d_g1 + zk->offC, tmpl, d_w, (uint32_t)wires.size(), (int32_t)zk->nPub + 1, d_g, ng, gC.data())) ||
        (rc = fold_group_sums_g2(ctx, zk->d_g2, tmpl, d_w, (uint32_t)wires.size(), 0, d_g, ng, gB2.data()))) return rc;
    ZKC_HIP_CHECK(ctx, hipFree(d_w)); ZKC_HIP_CHECK(ctx, hipFree(d_g));
    auto& f = zk->fold;
    for (int t = 0; t < 2; t++) {
        suffix_sums(f.sufA[t], gA.data() + t * n, n); suffix_sums(f.sufB1[t], gB1.data() + t * n, n);
        suffix_sums(f.sufC[t], gC.data() + t * n, n); suffix_sums(f.sufB2[t], gB2.data() + t * n, n);
    }
    f.baseA = {xyzz_add(gA[n - 1], gA[2 * n - 1])}; f.baseB1 = {xyzz_add(gB1[n - 1], gB1[2 * n - 1])};
    f.baseC = {xyzz_add(gC[n - 1], gC[2 * n - 1])}; f.baseB2 = {xyzz_add(gB2[n - 1], gB2[2 * n - 1])};
    // device tables for the blinding kernel: [base | suf[0][0..n) | suf[1][0..n)]
    if ((rc = fold_upload(ctx, &f.d_foldA, f.baseA, f.sufA)) || (rc = fold_upload(ctx, &f.d_foldB1, f.baseB1, f.sufB1)) || (rc = fold_upload(ctx, &f.d_foldC, f.baseC, f.sufC)) ||
        (rc = fold_upload(ctx, &f.d_foldB2, f.baseB2, f.sufB2))) return rc;
    f.ready = true;
    return ZKC_OK;
}
// 4-bit fixed-base tables of everything the blinding of a small pass multiplies by r or s (zkc_finalize.hip): delta1, alpha1, beta1, and -- with folding -- alpha1 and beta1
// plus the base constants of A and B1 and the per-depth suffix constants of both trees.  960 points per base: 80 MB at nLevels = 160, built on the device in about a millisecond.
static int fb4_prepare(zkc_zkey* zk) {
    const char* e_off = getenv("ZKC_BLIND_TREE");                      // = 0 at key load: this key's small passes take the general blinding kernels (A/B, tests)
    if ((e_off && atoi(e_off) == 0) || zk->d_fb4) return ZKC_OK;
    zkc_ctx* ctx = zk->ctx;
    std::vector<G1XYZZ> bases = {G1XYZZ::from_affine(zk->delta1), G1XYZZ::from_affine(zk->alpha1), G1XYZZ::from_affine(zk->beta1)};
    const auto& f = zk->fold;
    if (f.ready) {
        bases.push_back(xyzz_add_affine(f.baseA[0], zk->alpha1)); bases.push_back(xyzz_add_affine(f.baseB1[0], zk->beta1));
        for (int t = 0; t < 2; t++) bases.insert(bases.end(), f.sufA[t].begin(), f.sufA[t].end());
        for (int t = 0; t < 2; t++) bases.insert(bases.end(), f.sufB1[t].begin(), f.sufB1[t].end());
    }
    G1XYZZ* d_bases = nullptr; int rc;
    if ((rc = dmalloc(ctx, &d_bases, bases.size()))) return rc;
    if ((rc = dmalloc(ctx, &zk->d_fb4, bases.size() * FB4_WIN * FB4_ROW)) || (rc = dmalloc(ctx, &zk->d_fb4g2, (size_t)FB4_WIN * FB4_ROW + 2))) { (void)hipFree(d_bases); return rc; }
    hipError_t e = hipMemcpyAsync(d_bases, bases.data(), bases.size() * sizeof(G1XYZZ), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) { rc = fb4_build(ctx, ctx->stream, d_bases, (int)bases.size(), zk->d_fb4, G2XYZZ::from_affine(zk->delta2), G2XYZZ::from_affine(zk->beta2),
                                             f.ready ? xyzz_add_affine(f.baseB2[0], zk->beta2) : G2XYZZ::from_affine(zk->beta2), zk->d_fb4g2); if (!rc) e = hipStreamSynchronize(ctx->stream); }
    (void)hipFree(d_bases);
    if (rc || e != hipSuccess) {
        (void)hipFree(zk->d_fb4); (void)hipFree(zk->d_fb4g2); zk->d_fb4 = nullptr; zk->d_fb4g2 = nullptr;
        return rc ? rc : zkc_fail(ctx, ZKC_ERR_HIP, std::string("fb4_prepare: ") + hipGetErrorString(e));
    }
    zk->fb4_bases = (int)bases.size();
    return ZKC_OK;
}
// device lists of the wires that stay in the MSMs of each section when levels >= Dc (census) / >= Ds (sik) and the n2bOld
// blocks are folded into constants; wires whose base is the point at infinity are dropped per section
static int fold_vmap(zkc_zkey* zk, int Dc, int Ds, zkc_zkey::Fold::VMap* out) {
    auto it = zk->fold.vmaps.find({Dc, Ds});
    if (it != zk->fold.vmaps.end()) { *out = it->second; return ZKC_OK; }
    zkc_ctx* ctx = zk->ctx; const WitnessLayout L = WitnessLayout::make(zk->nLevels); const int n = L.n;
    std::vector<uint8_t> folded(L.nWires, 0);
    for (int tree = 0; tree < 2; tree++) {
        auto rg = fold_group_ranges(L, tree); const int D = tree == 0 ? Dc : Ds;
        for (int g = D; g < n - 1; g++) for (int w = rg[g].first; w < rg[g].second; w++) folded[w] = 1;
        for (int w = rg[n - 1].first; w < rg[n - 1].second; w++) folded[w] = 1;
    }
    std::vector<uint32_t> v; zkc_zkey::Fold::VMap m;
    m.offA = 0; for (int w = 0; w < L.nWires; w++) if (!folded[w] && !zk->fold.infA[w]) v.push_back((uint32_t)w); m.nA = (uint32_t)v.size();
    m.offB = (uint32_t)v.size(); for (int w = 0; w < L.nWires; w++) if (!folded[w] && !zk->fold.infB[w]) v.push_back((uint32_t)w); m.nB = (uint32_t)v.size() - m.offB;
    m.offC = (uint32_t)v.size(); for (int w = (int)zk->nPub + 1; w < L.nWires; w++) if (!folded[w] && !zk->fold.infC[w]) v.push_back((uint32_t)w); m.nC = (uint32_t)v.size() - m.offC;
    ZKC_HIP_CHECK(ctx, hipMalloc((void**)&m.d, v.size() * 4 + 4));
    ZKC_HIP_CHECK(ctx, hipMemcpyAsync(m.d, v.data(), v.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    zk->fold.vmaps[{Dc, Ds}] = m; *out = m;
    return ZKC_OK;
}

// [r4] the same three lists with NOTHING folded: every wire whose base is not the point at infinity.  A third of the census circuit's wires have a zero B polynomial (27 263 of
// 82 754: B1 and B2 bases at infinity) and real R1CS are like that in general; an unfolded pass (a foreign circuit, a foreign witness, ZKC_NO_FOLD) used to carry them through
// digit extraction, bucketing and the gathers only to skip them inside the accumulation, where a skipped entry still costs its wave an addition's time.  out->d == nullptr:
// the key has no such wires, the pass indexes the sections directly.
static int nofold_vmap(zkc_zkey* zk, zkc_zkey::Fold::VMap* out) {
    auto it = zk->fold.vmaps.find({-1, -1});
    if (it != zk->fold.vmaps.end()) { *out = it->second; return ZKC_OK; }
    zkc_ctx* ctx = zk->ctx; const uint32_t nv = zk->nVars, np = zk->nPub;
    std::vector<uint32_t> v; zkc_zkey::Fold::VMap m;
    m.offA = 0; for (uint32_t w = 0; w < nv; w++) if (!zk->fold.infA[w]) v.push_back(w); m.nA = (uint32_t)v.size();
    m.offB = (uint32_t)v.size(); for (uint32_t w = 0; w < nv; w++) if (!zk->fold.infB[w]) v.push_back(w); m.nB = (uint32_t)v.size() - m.offB;
    m.offC = (uint32_t)v.size(); for (uint32_t w = np + 1; w < nv; w++) if (!zk->fold.infC[w]) v.push_back(w); m.nC = (uint32_t)v.size() - m.offC;
    if ((size_t)m.nA + m.nB + m.nC + (size_t)nv / 50 < 3 * (size_t)nv - np - 1) {          // worth an indirection only when at least ~2 % of a section's entries drop out
        ZKC_HIP_CHECK(ctx, hipMalloc((void**)&m.d, v.size() * 4 + 4));
        ZKC_HIP_CHECK(ctx, hipMemcpyAsync(m.d, v.data(), v.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    zk->fold.vmaps[{-1, -1}] = m; *out = m;
    return ZKC_OK;
}

// stages a2..a4 for `nb` proofs: leaves (A'B' - C') on the odd coset in d_p[q] (standard form), q < nb
// buildABC (a2) for `nb` proofs on stream `mv`: A_w, B_w by the jagged-diagonal mat-vec, C_w = A_w o B_w
static int h_matvec_dev(zkc_zkey* zk, zkc_lane& L, const uint32_t* d_wtns0, int nb, hipStream_t mv) {
    zkc_ctx* ctx = zk->ctx; const uint32_t n = zk->n, nv = zk->nVars;
    zkc_prof_scope _ps(ctx, ZKC_PROF_MATVEC, (uint64_t)nb * ((uint64_t)zk->nCoeffs * 68 + 3ull * n * 32), mv);
    matvec_launch(zk, L, d_wtns0, nb, mv);
    ZKC_HIP_CHECK(ctx, hipGetLastError());
    return ZKC_OK;
}
// stages a2..a4 for `nb` proofs: leaves (A'B' - C') on the odd coset in d_p[q] (standard form), q < nb.  with_matvec = false: buildABC has been
// run already (on another stream; L.st has been made to wait for it)
static int h_evals_dev(zkc_zkey* zk, zkc_lane& L, hipStream_t st, const uint32_t* d_wtns0, int nb, bool with_matvec = true) {
    zkc_ctx* ctx = zk->ctx; const uint32_t n = zk->n;
    if (with_matvec) { int rc0 = h_matvec_dev(zk, L, d_wtns0, nb, st); if (rc0) return rc0; }
    zkc_prof_scope _pn(ctx, ZKC_PROF_NTT, (uint64_t)nb * (6ull * 2 * n * 32 + 4ull * n * 32), st);   // 6 transforms r+w, joinABC
    int rc;
    static const bool ntt_two_transforms = getenv("ZKC_NTT_SEPARATE") != nullptr;      // diagnostics: the round-1 path (two full transforms, four HBM round trips)
    if (!ntt_two_transforms && zk->logn >= 12 && zk->logn <= 27) {
        if ((rc = ntt_pair_run(ctx, st, L.d_abc, zk->d_tw_inv29, zk->d_tw_fwd29, zk->d_coset_br, (int)zk->logn, 3 * nb))) return rc;
    } else {
        if ((rc = ntt_run(ctx, st, L.d_abc, L.d_t, zk->d_tw_inv29, zk->d_coset, (int)zk->logn, 3 * nb))) return rc;
        if ((rc = ntt_run(ctx, st, L.d_t, L.d_abc, zk->d_tw_fwd29, nullptr, (int)zk->logn, 3 * nb))) return rc;
    }
    hipLaunchKernelGGL(zkc_join_abc, dim3((n + 255) / 256, nb), dim3(256), 0, st, L.d_abc, L.d_p, (int)n);
    ZKC_HIP_CHECK(ctx, hipGetLastError());
    return ZKC_OK;
}

extern "C" int zkc_debug_stage(zkc_zkey* zk, const void* d_wtns, int stage, void* host_out) {
    if (!zk || !d_wtns || !host_out) return ZKC_ERR_BAD_ARG;
    zkc_ctx* ctx = zk->ctx; const uint32_t n = zk->n;
    ZKC_LOCK(ctx);
    ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device)); ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    { int rc0 = lanes_ensure(zk, 1, 0, 1); if (rc0) return rc0; }
    zkc_lane& L0 = ctx->lanes[0]; const hipStream_t st0 = lane_st(zk, L0).st;
    if (stage == 0) {
        matvec_launch(zk, L0, (const uint32_t*)d_wtns, 1, st0);
        ZKC_HIP_CHECK(ctx, hipGetLastError());
        ZKC_HIP_CHECK(ctx, hipMemcpyAsync(host_out, L0.d_abc, 96ull * n, hipMemcpyDeviceToHost, st0));
    } else {
        int rc = h_evals_dev(zk, L0, st0, (const uint32_t*)d_wtns, 1); if (rc) return rc;
        ZKC_HIP_CHECK(ctx, hipMemcpyAsync(host_out, L0.d_p, 32ull * n, hipMemcpyDeviceToHost, st0));
    }
    ZKC_HIP_CHECK(ctx, hipStreamSynchronize(st0));
    return ZKC_OK;
}

extern "C" int zkc_msm_debug(zkc_zkey* zk, int which, const void* d_scalars, uint32_t count, void* host_out) {
    if (!zk || !d_scalars || !host_out || which < 0 || which > 4) return ZKC_ERR_BAD_ARG;
    zkc_ctx* ctx = zk->ctx;
    ZKC_LOCK(ctx);
    ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device)); ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    { int rc0 = lanes_ensure(zk, 1, 0, 1); if (rc0) return rc0; }
    zkc_lane& L0 = ctx->lanes[0]; const hipStream_t st_l0 = lane_st(zk, L0).st;
    const uint32_t full = which == 3 ? zk->nVars - zk->nPub - 1 : which == 4 ? zk->n : zk->nVars;
    if (count != full) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "zkc_msm_debug: count must equal the section size");
    static thread_local MsmJobList jl; jl.clear(256, 1024, which == 2 ? (uint32_t)MSM_MAX_VW_PER_JOB : (uint32_t)MSM_MAX_VW_G1);
    const uint32_t offs[5] = {zk->offA, zk->offB1, 0, zk->offC, zk->offH};
    jl.add((const uint32_t*)d_scalars, nullptr, count, offs[which], full, 0, which == 4 ? zk->c_h : zk->c_sec);
    int rc = which == 2 ? msm_pass_g2(zk, L0.w2, jl, 0, true, st_l0) : msm_pass_g1(zk, L0.w1, jl, 0, true, st_l0); if (rc) return rc;
    ZKC_HIP_CHECK(ctx, hipStreamSynchronize(st_l0));
    if (which == 2) g2_to_std((uint8_t*)host_out, xyzz_to_affine(*(G2XYZZ*)L0.w2.h_results));
    else g1_to_std((uint8_t*)host_out, xyzz_to_affine(*(G1XYZZ*)L0.w1.h_results));
    return ZKC_OK;
}

// d_inputs != nullptr: the witnesses are computed here as well, a chunk per pass on ctx->stream, so that the (latency-bound, few-wave)
// witness kernels of pass p+1 run underneath the MSMs of pass p
int zkc_witness_chunk_async(zkc_ctx* ctx, int nLevels, const void* d_inputs, int B, void* d_wtns, int32_t* d_status3, int32_t* d_status, hipStream_t st);
int zkc::prove_batch_begin(zkc_zkey* zk, int cs, const void* d_wtns, uint32_t nWitness, int B, const uint8_t* rs, bool want_publics, const void* d_inputs, int32_t* d_status,
                           int lane0, hipEvent_t wait_first, const uint8_t* host_depths, bool no_early) {
    if (!zk || !d_wtns || !rs || B <= 0 || cs < 0 || cs >= CALL_SLOTS || lane0 >= zk->nlanes) return zkc_fail(zk ? zk->ctx : nullptr, ZKC_ERR_BAD_ARG, "zkc_prove_batch_dev: bad argument");
    zkc_ctx* ctx = zk->ctx;
    ZKC_LOCK(ctx);
    if (nWitness != zk->nVars) return zkc_fail(ctx, ZKC_ERR_INVALID_WITNESS_LENGTH, "Invalid witness length. Circuit: " + std::to_string(zk->nVars) + ", witness: " + std::to_string(nWitness));
    for (int b = 0; b < 2 * B; b++) { uint32_t t[8]; memcpy(t, rs + 32 * (size_t)b, 32); if (!fp_std_lt_p<FrParams>(t)) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "r or s >= field order"); }
    ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t nv = zk->nVars, np = zk->nPub, nc = nv - np - 1, n = zk->n;
    const bool can_fold = zk->nLevels >= 0;
    // the stream of the call's witness kernels, fold check and small uploads: the context's for a call that may use every lane, the lane's own for a call that stays on one
    // (the proving service: the Poseidon chains of concurrent calls run side by side instead of queueing on one stream)
    // [r5'] ... which, for a call of ONE pass, is the lane's G1 stream itself: witness -> fold check -> buildABC are a chain anyway, and every further stream is a further
    // hardware queue to share (a stream that shares its queue with another lane's pending barrier packet runs behind that lane's call: profiles/r05_service_hw_queues.txt).
    // A call of several passes on one lane keeps the context's stream for its witness groups, so that pass p + 1's Poseidon chains run under pass p's MSMs.
    WitnessLayout L{}; int rc;
    // the lanes this call lands on -- its one lane, or as many of the key's lanes as it has passes -- hold a pass of this key (they are the context's: grown here if another,
    // smaller key shaped them, or if this is the first call that needs them)
    {
        const int npasses0 = (B + zk->max_inflight - 1) / zk->max_inflight, per0 = (B + npasses0 - 1) / npasses0;
        if ((rc = lanes_ensure(zk, per0, lane0 >= 0 ? lane0 : 0, lane0 >= 0 ? 1 : std::min(zk->nlanes, npasses0)))) return rc;
    }
    hipStream_t st0 = (lane0 >= 0 && !zk->serial_streams && B <= zk->max_inflight) ? ctx->lanes[lane0].st : ctx->stream;
    zkc_zkey::CallSlot& CS = zk->call[cs];
    if (CS.pending) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "prove_batch_begin: this call slot has a call in flight (finish it first)");
    if (CS.cap < (size_t)B) {                                    // the slot is idle (finished), so its buffers are nobody's
        if (CS.d_rs) { ZKC_HIP_CHECK(ctx, hipFree(CS.d_rs)); ZKC_HIP_CHECK(ctx, hipFree(CS.d_proofs)); ZKC_HIP_CHECK(ctx, hipHostFree(CS.h_out)); ZKC_HIP_CHECK(ctx, hipHostFree(CS.h_rs)); ZKC_HIP_CHECK(ctx, hipFree(CS.d_xyzz)); ZKC_HIP_CHECK(ctx, hipHostFree(CS.h_xyzz));
                       CS.d_rs = CS.d_proofs = CS.h_out = CS.h_rs = CS.d_xyzz = CS.h_xyzz = nullptr; CS.cap = 0; }
        const size_t want = std::max<size_t>((size_t)B, std::max<size_t>(256, std::min<size_t>(2 * CS.cap, 4096)));      // never less than 256 proofs (0.5 MB): a service whose batches grow from 1 to 60 does not come through this (device-synchronising) path six times
        ZKC_HIP_CHECK(ctx, hipMalloc((void**)&CS.d_rs, 64 * want)); ZKC_HIP_CHECK(ctx, hipMalloc((void**)&CS.d_proofs, 256 * want));
        ZKC_HIP_CHECK(ctx, hipHostMalloc((void**)&CS.h_out, (256 + 32 * (size_t)zk->nPub) * want)); ZKC_HIP_CHECK(ctx, hipHostMalloc((void**)&CS.h_rs, 64 * want));
        ZKC_HIP_CHECK(ctx, hipMalloc((void**)&CS.d_xyzz, 512 * want)); ZKC_HIP_CHECK(ctx, hipHostMalloc((void**)&CS.h_xyzz, 512 * want)); CS.cap = want;
    }
    // [r5] the events a host thread waits on (zkc_wait_event: polled with naps, not hipEventSynchronize'd)
    static const unsigned ev_host_flags = hipEventDisableTiming;           // (hipEventBlockingSync was tried: the waits spin all the same on this stack, see zkc_wait_event)
    for (int l = 0; l < zk->nlanes; l++) if (!CS.ev_done[l]) ZKC_HIP_CHECK(ctx, hipEventCreateWithFlags(&CS.ev_done[l], ev_host_flags));
    CS.as_xyzz.assign((size_t)B, 0); CS.lanes_used = 0;
    if (wait_first) ZKC_HIP_CHECK(ctx, hipStreamWaitEvent(st0, wait_first, 0));
    uint8_t* const h_pub = CS.h_out + 256ull * CS.cap;      // results land in pinned memory so that no copy blocks the enqueueing thread
    if (rs != CS.h_rs_copy.data()) CS.h_rs_copy.assign(rs, rs + 64 * (size_t)B);      // (kept for a second begin of the same call: prove_batch_finish, early layout refused)
    memcpy(CS.h_rs, rs, 64 * (size_t)B);                                     // the caller's rs is its own again when begin returns
    ZKC_HIP_CHECK(ctx, hipMemcpyAsync(CS.d_rs, CS.h_rs, 64 * (size_t)B, hipMemcpyHostToDevice, st0));
    // per pass (chunk of max_inflight proofs), all enqueued now on st0: [witness kernels] -> fold check (which levels of the witness differ
    // from the voter-independent template?) -> flags to the host -> event.  The pass loop below waits for a chunk's event only.
    uint32_t* tmpl = nullptr;
    if (can_fold) {
        L = WitnessLayout::make(zk->nLevels);
        if ((rc = fold_prepare(zk))) return rc;
        tmpl = zkc_get_template(ctx, zk->nLevels); if (!tmpl) return ZKC_ERR_HIP;
        const size_t nflags = (size_t)B * 2 * L.n;
        if (CS.flags_cap < nflags) {                                   // (the slot is idle: nothing reads its old buffers)
            if (CS.d_flags) { ZKC_HIP_CHECK(ctx, hipFree(CS.d_flags)); ZKC_HIP_CHECK(ctx, hipHostFree(CS.h_flags)); CS.d_flags = CS.h_flags = nullptr; CS.flags_cap = 0; }
            const size_t want = std::max(nflags, (size_t)256 * 2 * L.n);
            ZKC_HIP_CHECK(ctx, hipMalloc((void**)&CS.d_flags, want * 4)); ZKC_HIP_CHECK(ctx, hipHostMalloc((void**)&CS.h_flags, want * 4)); CS.flags_cap = want;
        }
    }
    // [r3] a call of one or two voters that brings its inputs: the depths its folding needs are read from the inputs now (zkc_input_depths, on a stream of its own, ~30 us),
    // so the pass below is enqueued while the witness kernel runs instead of after a host round trip behind it (fold flags to the host, wake-up, ~70 launches: 0.1-0.2 ms of a 3.5 ms proof)
    // [r5] ... and so does any call that is ONE pass (the proving service's batches of concurrent callers): begin then never waits for a witness kernel, it returns after ~0.5 ms
    // of enqueueing and the next worker's begin can follow at once (ZKC_EARLY_LAYOUT=0: calls of more than two voters wait for their fold flags as before)
    // ... and with the depths handed in by the caller (host_depths: the proving service reads them off the inputs -- or, for a witness that was computed elsewhere, off the
    // sibling wires of the witness itself -- on the host) begin has no GPU round trip at all: the depth kernel below is microseconds of work, but its stream waits for a
    // hardware queue behind whatever long kernels the other lanes have in flight, and begin holds the context's lock meanwhile.
    CS.early_n = 0; CS.arg_wtns = d_wtns; CS.arg_nw = nWitness; CS.arg_publics = want_publics; CS.arg_lane0 = lane0; CS.arg_inputs = d_inputs != nullptr; CS.arg_wait = nullptr;
    static const bool early_any = [] { const char* e = getenv("ZKC_EARLY_LAYOUT"); return !(e && atoi(e) == 0); }();
    const size_t ecap = MSM_MAX_JOBS / 4;                                          // voters of a pass at most
    const bool early_wanted = !no_early && can_fold && B <= zk->max_inflight && (B <= 2 ? zk->d_fb4 != nullptr : early_any);      // one pass
    if (early_wanted && !CS.h_early) { ZKC_HIP_CHECK(ctx, hipHostMalloc((void**)&CS.h_early, (ecap + ecap * 2 * 256) * 4)); CS.early_cap = ecap; }      // status [ecap], then fold flags [B][2][n <= 254]
    if (early_wanted && host_depths) {
        bool ok = true; for (int i = 0; i < 2 * B; i++) ok = ok && host_depths[i] < (uint8_t)L.n && L.n <= 255;
        if (ok) { CS.early_n = B; CS.early_depth.assign(host_depths, host_depths + 2 * (size_t)B); }
    } else if (early_wanted && d_inputs) {
        if (!zk->d_depths) { ZKC_HIP_CHECK(ctx, hipMalloc((void**)&zk->d_depths, 2 * ecap * 4)); ZKC_HIP_CHECK(ctx, hipHostMalloc((void**)&zk->h_depths, 2 * ecap * 4)); }
        uint32_t* hd = zk->h_depths;                                               // read back below, under the context lock: one buffer per key is enough
        if (wait_first) ZKC_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream2, wait_first, 0));
        hipLaunchKernelGGL(zkc_input_depths, dim3(B, 2), dim3(64), 0, ctx->stream2, (const uint32_t*)d_inputs, L.nInputs, L.n, zk->d_depths);
        ZKC_HIP_CHECK(ctx, hipGetLastError());
        ZKC_HIP_CHECK(ctx, hipMemcpyAsync(hd, zk->d_depths, 8 * (size_t)B, hipMemcpyDeviceToHost, ctx->stream2));
        ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream2));
        bool ok = true; for (int i = 0; i < 2 * B; i++) ok = ok && hd[i] < (uint32_t)L.n;          // a non-zero LAST sibling fails SMTLevIns: no early path for that call
        if (ok) { CS.early_n = B; CS.early_depth.resize(2 * (size_t)B); for (int i = 0; i < 2 * B; i++) CS.early_depth[i] = (uint8_t)hd[i]; }
    }
    const int npasses = (B + zk->max_inflight - 1) / zk->max_inflight;
    const int per_pass = (B + npasses - 1) / npasses;       // passes of equal size (1024 -> 10 x 94 + 84, not 10 x 96 + 64): the last pass' fixed-latency tail is not spent on a stub
    while ((int)CS.ev_chunk.size() < npasses) { hipEvent_t e; ZKC_HIP_CHECK(ctx, hipEventCreateWithFlags(&e, ev_host_flags)); CS.ev_chunk.push_back(e); }      // per call slot: a call laid out early returns from begin before its event has fired, and the next call must not re-record it
    int32_t* d_status3 = nullptr;
    if (d_inputs) {
        if ((rc = zkc_ensure(ctx, (void**)&CS.d_status3, &CS.status3_cap, std::max<size_t>((size_t)B, 256) * 3 * sizeof(int32_t)))) return rc;      // per call slot: the witness kernels of two calls may run side by side
        d_status3 = CS.d_status3;
    }
    static const int wgroup = [] { const char* e = getenv("ZKC_WITNESS_GROUP"); return e ? std::max(1, atoi(e)) : 8; }();      // same box: 2 -> 2452, 4 -> 2472, 8 -> 2482 proofs/s at batch 1024
    for (int c = 0; c < npasses; c++) {
        const int p0 = c * per_pass, nb = std::min(per_pass, B - p0);
        uint32_t* wc = (uint32_t*)d_wtns + (size_t)p0 * nv * 8;
        // the chain kernel is a latency chain (one lane per Merkle path, ~14 ms alone whatever the batch, several times that while the MSMs
        // own the SIMDs): one launch covers the voters of ZKC_WITNESS_GROUP passes so that it never becomes the pipeline's pace
        // [r2] the very first pass of a call gets a launch of its own: with at most max_inflight voters it takes the wave-per-chain kernel and is
        // through in a third of the time, so the pipeline starts ~6 ms earlier; the rest of the first group follows behind its fold check
        if (d_inputs && (c % wgroup == 0 || c == 1)) {
            const int glast = c == 0 ? 1 : std::min(((c / wgroup) + 1) * wgroup, npasses);          // passes [c, glast)
            const int g0 = p0, gn = std::min((glast - c) * per_pass, B - g0);
            if (gn > 0 && (rc = zkc_witness_chunk_async(ctx, zk->nLevels, (const uint8_t*)d_inputs + (size_t)g0 * L.nInputs * 32, gn, wc, d_status3 + 3 * (size_t)g0, d_status + g0, st0))) return rc;
        }
        if (can_fold) {
            uint32_t* fl = CS.d_flags + (size_t)p0 * 2 * L.n;
            hipLaunchKernelGGL(zkc_fold_check, dim3(L.n, 2, nb), dim3(64), 0, st0, L, (const uint32_t*)wc, tmpl, fl, nb);
            ZKC_HIP_CHECK(ctx, hipGetLastError());
            ZKC_HIP_CHECK(ctx, hipMemcpyAsync(CS.h_flags + (size_t)p0 * 2 * L.n, fl, (size_t)nb * 2 * L.n * 4, hipMemcpyDeviceToHost, st0));
        }
        if (CS.early_n) {            // the early layout's evidence for finish, in this call's own pinned memory
            if (d_status) ZKC_HIP_CHECK(ctx, hipMemcpyAsync(CS.h_early, d_status, 4 * (size_t)B, hipMemcpyDeviceToHost, st0));
            else memset(CS.h_early, 0, 4 * (size_t)B);                             // witnesses given: nobody was rejected here
            ZKC_HIP_CHECK(ctx, hipMemcpyAsync(CS.h_early + CS.early_cap, CS.d_flags + (size_t)p0 * 2 * L.n, (size_t)nb * 2 * L.n * 4, hipMemcpyDeviceToHost, st0));
        }
        ZKC_HIP_CHECK(ctx, hipEventRecord(CS.ev_chunk[c], st0));                // wtns of this chunk (and rs) ready, flags on the host
    }
    // (npass is NOT reset per call: the result slots and their ev_fin guards carry over, because the previous call's last blinding may still be reading them)
    // ZKC_TRACE_HOST=1: where the enqueueing thread spends its time, per pass (diagnostics: a blocking call here idles a stream)
    static const bool trace_host = getenv("ZKC_TRACE_HOST") != nullptr;
    auto now_ms = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; };
    const double t_begin = now_ms(); double tr[6] = {0};
    int pass = 0, done_on_st_lane = -1;
    const bool one_lane = lane0 >= 0 || zk->nlanes == 1;
    for (int p0 = 0; p0 < B; p0 += per_pass, pass++) {
        const int nb = std::min(per_pass, B - p0);
        const int li = lane0 >= 0 ? lane0 : pass % zk->nlanes; CS.lanes_used |= 1u << li;
        zkc_lane& LN = ctx->lanes[li]; const LaneSt LS = lane_st(zk, LN); const int slot = LN.npass & 1;       // MSM results are double-buffered: the blinding of pass k overlaps pass k+1
        tr[0] = now_ms();
        const bool early = CS.early_n > 0;
        if (!early) ZKC_HIP_CHECK(ctx, zkc_wait_event(CS.ev_chunk[pass]));              // host: this chunk's fold flags have arrived
        tr[1] = now_ms();
        ZKC_HIP_CHECK(ctx, hipStreamWaitEvent(LS.st, CS.ev_chunk[pass], 0)); ZKC_HIP_CHECK(ctx, hipStreamWaitEvent(LS.st2, CS.ev_chunk[pass], 0));
        hipStream_t st = LS.st, st2 = LS.st2, fin = LS.fin;
        const uint32_t* w0 = (const uint32_t*)d_wtns + (size_t)p0 * nv * 8;
        // constant folding is per proof: voter q keeps the census levels below its own leaf depth dcq[q] (sik: dsq[q]) in its MSMs; the levels
        // above are the template's and come back as a constant in the blinding kernel.  One foreign witness (n2bOld block differs) unfolds the pass.
        uint8_t dcq[MSM_MAX_JOBS / 4], dsq[MSM_MAX_JOBS / 4]; bool fold = can_fold;
        if (early) for (int q = 0; q < nb; q++) { dcq[q] = CS.early_depth[2 * q]; dsq[q] = CS.early_depth[2 * q + 1]; }
        else for (int q = 0; q < nb && fold; q++) for (int t = 0; t < 2; t++) {
            const uint32_t* f = CS.h_flags + ((size_t)(p0 + q) * 2 + t) * L.n;
            if (f[L.n - 1]) { fold = false; break; }                              // n2bOld block differs: not one of our witnesses
            int D = 0; for (int g = 0; g < L.n - 1; g++) if (f[g]) D = g + 1;
            (t == 0 ? dcq : dsq)[q] = (uint8_t)D;
        }
        static thread_local zkc_zkey::Fold::VMap vms[MSM_MAX_JOBS / 4];
        if (fold) for (int q = 0; q < nb; q++) if ((rc = fold_vmap(zk, dcq[q], dsq[q], &vms[q]))) return rc;
        static const bool nofold_lists = [] { const char* e = getenv("ZKC_NOFOLD_LISTS"); return !(e && atoi(e) == 0); }();
        if (!fold && nofold_lists) {                        // an unfolded pass still leaves out the wires whose bases are at infinity (one list set per key)
            zkc_zkey::Fold::VMap nf; if ((rc = nofold_vmap(zk, &nf))) return rc;
            for (int q = 0; q < nb; q++) vms[q] = nf;
        }
        const bool listed = fold || (nofold_lists && vms[0].d != nullptr);                      // the pass' jobs run over wire lists (vms) instead of whole sections
        if (LN.npass >= 2) { ZKC_HIP_CHECK(ctx, hipStreamWaitEvent(st, LN.ev_fin[slot], 0)); ZKC_HIP_CHECK(ctx, hipStreamWaitEvent(st2, LN.ev_fin[slot], 0)); }   // slot still read by the blinding two passes back?
        LN.npass++;
        // [r2] buildABC is gather-bound and needs only the witness: the one of pass p+1 is issued on the blinding stream as soon as the accumulation
        // of pass p starts (VALU-bound, 15 ms) and is through long before the G1 stream gets there; only the first pass runs it in line
        static const bool mv_prefetch_env = getenv("ZKC_MATVEC_INLINE") == nullptr;
        const bool mv_prefetch = mv_prefetch_env && one_lane;
        const bool mv_done = mv_prefetch && pass > 0;
        if (mv_done) ZKC_HIP_CHECK(ctx, hipStreamWaitEvent(st, LN.ev_mv, 0));
        // (tried for passes of a few proofs while their G2 side was the longer chain: G2 enqueued first and its accumulation not held for the transforms -- its 1024 fat waves
        // then slowed buildABC and the first transform kernel threefold; with the 8-bit-window G2 table the G1 side is the longer one and goes first again)
        if ((rc = h_evals_dev(zk, LN, st, w0, nb, !mv_done))) return rc;
        ZKC_HIP_CHECK(ctx, hipEventRecord(LN.ev_ntt, st));
        tr[2] = now_ms();
        static thread_local MsmJobList j1, j2;     // 6 KB each: kept off the stack frame of a C-ABI entry point
        // buckets per wave of the bucket reduction ("virtual window"): a lane walks vw / 64 buckets with two full additions each, then the 64 lanes pay ~17 more for the scan and
        // the tree -- fixed cost per wave, so a full pass wants large windows (fewer additions in all), a small one small windows (fewer in a row).  [r4] re-measured on the
        // round-4 pipeline, one box, (H, sections): (1024, 256) 3195 / 3205 / 3212 / 3214 proofs/s -- round 1's optimum, the default until now -- (2048, 512) 3235 / 3238,
        // (4096, 1024) 3229, (4096, 2048) 3248 / 3261 / 3264 / 3268 (+1.8 %), (8192, 1024) 3100.  Passes of fewer than 32 proofs keep the smaller windows (latency).
        static const uint32_t vwb_env = [] { const char* e = getenv("ZKC_VW_BIG"); return e ? (uint32_t)atoi(e) : 0u; }();
        static const uint32_t vws_env = [] { const char* e = getenv("ZKC_VW_SMALL"); return e ? (uint32_t)atoi(e) : 0u; }();
        const uint32_t vws = nb <= 4 ? 64u : vws_env ? vws_env : nb < 32 ? 256u : 2048u, vwb = nb <= 4 ? 256u : vwb_env ? vwb_env : nb < 32 ? 1024u : 4096u;
        // the G2 section of one or two proofs takes the 8-bit-window table: 128 buckets per job, reduced by one wave (vw = 128), if the G2 work space holds 32 entries per scalar
        size_t lone_entries = 0; for (int q = 0; q < nb; q++) lone_entries += (size_t)msm_nw(MSM_C_G2_LONE) * (listed ? vms[q].nB : nv);
        // [r4] deep pass: the voters of this pass keep, on average, more than ZKC_DEEP_WIRES (16 000) wires per section in their MSMs (leaves far down the trees, or witnesses that
        // do not fold): the sections take the key's second tables (c_deep bits: msm_c_for).  Per section of W full-width scalars: 22 W + 2048 x 4.2 additions at 12 bits,
        // 17 W + 16384 x 4.2 at 15: break-even at W = 12 k by that count, at 14 k measured (profiles/r04_deep_pass_threshold.json: voters 30 levels down; +1.4 % at 40 levels,
        // +16 % at 160); a census of 2^13 .. 2^20 voters keeps 6-9 k and stays at 12 bits (-3 % if forced over the second tables).
        static const size_t deep_wires = [] { const char* e = getenv("ZKC_DEEP_WIRES"); return e ? (size_t)atol(e) : (size_t)16000; }();
        size_t live_wires = 0; for (int q = 0; q < nb; q++) live_wires += listed ? (size_t)vms[q].nA + vms[q].nB + vms[q].nC : 2 * (size_t)nv + nc;
        const bool deep = zk->c_deep != 0 && nb > 2 && live_wires >= 3 * deep_wires * (size_t)nb;
        const int cw = deep ? zk->c_deep : zk->c_sec;                                                     // window bits of this pass' witness sections
        const uint32_t oA = deep ? zk->offA_deep : zk->offA, oB1 = deep ? zk->offB1_deep : zk->offB1, oC = deep ? zk->offC_deep : zk->offC;
        const int c2 = (nb <= 2 && zk->d_g2_29_lone && lone_entries <= LN.w2.max_entries) ? MSM_C_G2_LONE : cw;
        static const uint32_t vwg2_env = [] { const char* e = getenv("ZKC_VW_G2"); return e ? (uint32_t)atoi(e) : 0u; }();      // A/B: the G2 jobs' window apart from the G1 sections'
        j1.clear(vws, vwb, MSM_MAX_VW_G1); j2.clear(c2 == MSM_C_G2_LONE ? 128u : (vwg2_env && nb >= 32) ? vwg2_env : vws, 1024, MSM_MAX_VW_PER_JOB);
        // job order of the G1 pass: the nb H jobs first (the 16-bit bucket sort wants the jobs with the larger bucket count in front), then
        // A, B1, C per proof.  zkc_finalize reads results[q] = H_q and results[nb + 3 q + {0, 1, 2}] = A_q, B1_q, C_q.
        for (int q = 0; q < nb; q++) j1.add(LN.d_p + 8 * (size_t)n * q, nullptr, n, zk->offH, n, 0, zk->c_h);
        // [r3] a pass of one or two proofs carries its blinding's two variable-base products as two more MSM jobs each -- sum (s w_i) A_i and sum (r w_i) B1_i over the
        // wires that stay in the proof's MSMs -- so that the blinding kernel is left with fixed-base products only (zkc_finalize.hip)
        // (the lanes' work space is sized for max(4, passes of this key) proofs of 3 + 1 jobs each: two proofs of 5 + 1 fit unless ZKC_INFLIGHT made the passes smaller than that)
        size_t tree_entries = 0; for (int q = 0; q < nb; q++) tree_entries += (size_t)msm_nw(zk->c_h) * n + (size_t)msm_nw(zk->c_sec) * (listed ? 2 * (size_t)vms[q].nA + 2 * (size_t)vms[q].nB + vms[q].nC : 4 * (size_t)nv + nc);
        const bool tree = zk->d_fb4 != nullptr && nb <= 2 && 6 * nb <= LN.w1.max_jobs && tree_entries <= LN.w1.max_entries && (size_t)nb * (5 * msm_half(zk->c_sec) + msm_half(zk->c_h)) <= LN.w1.max_buckets;
        BlindArgs ba{}; ba.rs = CS.d_rs + 64 * (size_t)p0; ba.nv = nv;
        for (int q = 0; q < nb; q++) {
            const uint32_t* w = w0 + (size_t)q * nv * 8;
            uint32_t* bs = tree ? LN.d_bs + (size_t)q * 2 * nv * 8 : nullptr;
            if (tree) { ba.w[q] = w; ba.out[q] = bs; }
            if (listed) {
                const zkc_zkey::Fold::VMap& vm = vms[q];
                j1.add(w, vm.d + vm.offA, vm.nA, oA, nv, 0, cw);
                j1.add(w, vm.d + vm.offB, vm.nB, oB1, nv, 0, cw);
                j1.add(w, vm.d + vm.offC, vm.nC, oC, nc, (int32_t)np + 1, cw);
                if (tree) { j1.add(bs, vm.d + vm.offA, vm.nA, oA, nv, 0, cw); j1.add(bs + 8ull * nv, vm.d + vm.offB, vm.nB, oB1, nv, 0, cw); }
                j2.add(w, vm.d + vm.offB, vm.nB, 0, nv, 0, c2);
                if (tree) { ba.mapA[q] = vm.d + vm.offA; ba.nA[q] = vm.nA; ba.mapB[q] = vm.d + vm.offB; ba.nB[q] = vm.nB; }
            } else {
                j1.add(w, nullptr, nv, oA, nv, 0, cw);
                j1.add(w, nullptr, nv, oB1, nv, 0, cw);
                j1.add(w + 8ull * (np + 1), nullptr, nc, oC, nc, 0, cw);
                if (tree) { j1.add(bs, nullptr, nv, oA, nv, 0, cw); j1.add(bs + 8ull * nv, nullptr, nv, oB1, nv, 0, cw); }
                j2.add(w, nullptr, nv, 0, nv, 0, c2);
                if (tree) { ba.mapA[q] = ba.mapB[q] = nullptr; ba.nA[q] = ba.nB[q] = nv; }
            }
        }
        if (tree && (rc = blind_scalars_launch(ctx, st, ba, nb))) return rc;                  // on the G1 stream, ahead of its bucketing (the witness and (r, s) are there: ev_chunk)
        // the blinding's arguments: piB of a small pass is written on the G2 stream right behind the G2 MSM, everything else on the blinding stream below
        FinalizeArgs fa{};
        fa.r1 = (const G1XYZZ*)LN.w1.results + (size_t)slot * LN.w1.max_jobs; fa.r2 = (const G2XYZZ*)LN.w2.results + (size_t)slot * LN.w2.max_jobs;
        fa.foldA = zk->fold.d_foldA; fa.foldB1 = zk->fold.d_foldB1; fa.foldC = zk->fold.d_foldC; fa.foldB2 = zk->fold.d_foldB2; fa.fold_n = can_fold ? L.n : 0;
        for (int q = 0; q < nb; q++) { fa.dc[q] = fold ? dcq[q] : (uint8_t)255; fa.ds[q] = fold ? dsq[q] : (uint8_t)255; }
        fa.tblDelta1 = zk->d_tblDelta1; fa.tblAlpha1 = zk->d_tblAlpha1; fa.tblBeta1 = zk->d_tblBeta1; fa.tblDelta2 = zk->d_tblDelta2;
        fa.alpha1 = zk->alpha1; fa.beta2 = zk->beta2; fa.rs = CS.d_rs + 64 * (size_t)p0; fa.out = CS.d_proofs + 256 * (size_t)p0; fa.scratch = LN.d_fin;
        fa.per = tree ? 5 : 3; fa.fb4 = zk->d_fb4; fa.fb4g2 = zk->d_fb4g2; fa.out_xyzz = CS.d_xyzz + 512 * (size_t)p0;
        // The G2 MSM needs only the witness: by default it starts with the pass and runs beside buildABC/NTT/G1 sort.  ZKC_G2_LATE holds it back
        // until the G1 stream has finished its short kernels (they were seen to stall next to the G2 chain's low-occupancy kernels); the G2
        // kernels then starve behind the 13 ms G1 accumulation instead and spill into the next pass -- measured equal within noise.
        static const bool g2_early = getenv("ZKC_G2_LATE") == nullptr;
        // [r2] its bucketing starts with the pass, but its accumulation (VALU-bound, 3.3 ms alone) is held until the G1 stream leaves the NTT: it then
        // runs beside the G1 bucketing and segment kernels, which wait on memory and LDS atomics, instead of beside the NTT, which is VALU-bound too
        // [r4] ... which was right for round 2's pipeline and is not for this one: with the larger reduction windows the G2 side's tail is longer (one wave walks a job's 2048
        // buckets: ~64 G2 additions in a row), and holding its accumulation back makes the blinding -- and with it the result slot the pass after next needs -- wait for it.
        // Alternating on one box: 3160 / 3143 / 3160 / 3150 / 3151 proofs/s held back, 3205 / 3198 / 3206 / 3169 / 3206 started with the pass: +1.4 %.  ZKC_G2_ACC_HOLD=1: the old order.
        // (Two pipeline lanes, -4 % in round 2, are +1.4 % now too -- 3202 / 3206 / 3197 / 3202 / 3201 -- but not on top of this (3107 / 3115 with both) and for twice the work space.)
        // Passes of fewer than 32 proofs keep the hold: a lone proof's G2 kernels (1024 waves of 400 registers) slowed buildABC and the first transform kernel threefold when they were
        // not held (round 3), and the twelve-proof passes of a 2^20-domain key run 140 proofs/s held against 136 early (their bucketing phase is 9 ms long: room for the G2 accumulation).
        static const bool g2_hold_env = getenv("ZKC_G2_ACC_HOLD") != nullptr, g2_early_env = getenv("ZKC_G2_ACC_EARLY") != nullptr;
        // [r5] ... and with the passes of a call rotating over four lanes the hold is right again for full passes: the G2 accumulation of lane k then lands beside lane k's own
        // bucketing instead of beside another lane's transforms (alternating on one box: 3380 / 3386 proofs/s held, 3363 / 3363 started with the pass: +0.6 %)
        const bool g2_acc_with_sort = g2_hold_env || ((nb < 32 || !one_lane) && !g2_early_env);
        if (g2_early) {
            if ((rc = msm_pass_g2(zk, LN.w2, j2, slot, false, st2, g2_acc_with_sort ? LN.ev_ntt : nullptr))) return rc;
            if (tree && (rc = finalize_tree_g2_launch(ctx, st2, fa, nb))) return rc;
            ZKC_HIP_CHECK(ctx, hipEventRecord(LN.ev_msm2, st2));
        }
        tr[3] = now_ms();
        // [r4] the work space of the G1 pass (segment lists, partial sums, job and window lists) is the previous pass' until its bucket reduction is through -- which, for a
        // full pass, runs on the lane's reduction stream beside THIS pass' buildABC and transforms (a no-op wait when the reduction ran on this stream)
        ZKC_HIP_CHECK(ctx, hipStreamWaitEvent(st, LN.ev_red, 0));
        // (measured, alternating on one box: 3102 / 3089 / 3083 proofs/s with the reduction on its own stream, 3074 / 3087 / 3097 without -- nothing: the pass is bound by the
        // sum of its kernels' VALU work, and where the reduction's waves run does not change that sum.  Off unless ZKC_REDUCE_STREAM=1.)
        static const bool red_on = [] { const char* e = getenv("ZKC_REDUCE_STREAM"); return e && atoi(e) == 1; }();
        const bool red_split = red_on && !zk->serial_streams && nb >= 32;
        if ((rc = msm_pass_g1(zk, LN.w1, j1, slot, false, st, LN.ev_sorted, LN.ev_acc, (red_split && LS.red) ? LS.red : nullptr, LN.ev_red))) return rc;
        tr[4] = now_ms();
        if (!g2_early) {
            ZKC_HIP_CHECK(ctx, hipStreamWaitEvent(st2, LN.ev_sorted, 0));
            if ((rc = msm_pass_g2(zk, LN.w2, j2, slot, false, st2))) return rc;
            if (tree && (rc = finalize_tree_g2_launch(ctx, st2, fa, nb))) return rc;
            ZKC_HIP_CHECK(ctx, hipEventRecord(LN.ev_msm2, st2));
        }
        if (want_publics)       // wires 1..nPublic of every witness, one strided copy
            ZKC_HIP_CHECK(ctx, hipMemcpy2DAsync(h_pub + 32ull * np * p0, 32ull * np, w0 + 8, 32ull * nv, 32ull * np, nb, hipMemcpyDeviceToHost, st));
        ZKC_HIP_CHECK(ctx, hipEventRecord(LN.ev_msm, st));
        if (mv_prefetch && p0 + per_pass < B) {                                   // buildABC of the next pass, beside this pass' accumulation
            const int p1 = p0 + per_pass, nb1 = std::min(per_pass, B - p1);
            static const bool mv_at_ntt = [] { const char* e = getenv("ZKC_MV_PREFETCH_AT_NTT"); return e && atoi(e) == 1; }();      // A/B: start it beside this pass' bucketing instead
            ZKC_HIP_CHECK(ctx, hipStreamWaitEvent(fin, mv_at_ntt ? LN.ev_ntt : LN.ev_sorted, 0));         // this pass' NTT and joinABC are through (d_abc is free), its accumulation is next (waiting on ev_ntt instead, i.e. starting beside the bucketing, is 1 % slower: 3113 / 3093 against 3143 / 3137 proofs/s on one box)
            ZKC_HIP_CHECK(ctx, hipStreamWaitEvent(fin, CS.ev_chunk[pass + 1], 0));
            if ((rc = h_matvec_dev(zk, LN, (const uint32_t*)d_wtns + (size_t)p1 * nv * 8, nb1, fin))) return rc;
            ZKC_HIP_CHECK(ctx, hipEventRecord(LN.ev_mv, fin));
        }
        // a7 on the second stream: overlaps the next pass
        // a call that is one small pass: its blinding and its copy go on the G1 stream itself -- nothing follows that they could overlap with, and a hop to the blinding stream is
        // ~25 us of a 3.2 ms proof
        // [r5] ... and so does every one-pass call that stays on its lane (the proving service): its blinding can only start when both MSM sides are through and nothing of the
        // call follows it, so a stream of its own buys nothing and costs a hardware queue (a service of four lanes keeps 8 streams busy instead of 12)
        hipStream_t bl = ((tree || lane0 >= 0) && npasses == 1) ? st : fin;
        if (bl == fin) { ZKC_HIP_CHECK(ctx, hipStreamWaitEvent(fin, LN.ev_msm, 0)); ZKC_HIP_CHECK(ctx, hipStreamWaitEvent(fin, LN.ev_red, 0)); }      // ev_msm: the G1 stream up to the copy of the public signals; ev_red: the G1 results
        // [r4] ... but the blinding scratch (LN.d_fin) is the lane's, and the pass before this one -- the last pass of the PREVIOUS call, begun on the other call slot -- blinds on
        // `fin`: without this wait its lane-per-product kernels and this call's tree kernels could write the scratch at the same time (a wrong proof for one of the two callers,
        // seen once in tests/test_gpu_service.py::test_queue_spills_over_further_device_entries under a load of mixed batch sizes).  That pass' blinding is long over when
        // this call's MSMs are through, so the wait costs nothing.
        else { if (LN.npass >= 2) ZKC_HIP_CHECK(ctx, hipStreamWaitEvent(st, LN.ev_fin[slot ^ 1], 0)); if (red_split) ZKC_HIP_CHECK(ctx, hipStreamWaitEvent(st, LN.ev_red, 0)); }
        if (!tree) ZKC_HIP_CHECK(ctx, hipStreamWaitEvent(bl, LN.ev_msm2, 0));      // a small pass' piB is written on the G2 stream: piA and piC need not wait for it ...
        if ((rc = finalize_launch(ctx, bl, fa, nb))) return rc;
        if (tree) ZKC_HIP_CHECK(ctx, hipStreamWaitEvent(bl, LN.ev_msm2, 0));        // ... only the copy of the finished proof does
        if (tree) { ZKC_HIP_CHECK(ctx, hipMemcpyAsync(CS.h_xyzz + 512ull * p0, CS.d_xyzz + 512 * (size_t)p0, 512ull * nb, hipMemcpyDeviceToHost, bl)); for (int q = 0; q < nb; q++) CS.as_xyzz[p0 + q] = 1; }
        else ZKC_HIP_CHECK(ctx, hipMemcpyAsync(CS.h_out + 256ull * p0, CS.d_proofs + 256 * (size_t)p0, 256ull * nb, hipMemcpyDeviceToHost, bl));
        ZKC_HIP_CHECK(ctx, hipEventRecord(LN.ev_fin[slot], bl));
        if (bl == st) done_on_st_lane = li;
        tr[5] = now_ms();
        if (trace_host) fprintf(stderr, "[zkc host] pass %2d: start %8.2f | chunk wait %6.2f | h_evals %6.2f | g2 pass %6.2f | g1 pass %6.2f | blinding %6.2f ms\n", pass, tr[0] - t_begin,
                                tr[1] - tr[0], tr[2] - tr[1], tr[3] - tr[2], tr[4] - tr[3], tr[5] - tr[4]);
    }
    // every lane's blinding stream already waits for its G1 and G2 streams (ev_msm, ev_msm2) and carries the last copies: one event per lane closes the call
    for (int l = 0; l < zk->nlanes; l++) if (CS.lanes_used >> l & 1) { const LaneSt ls = lane_st(zk, ctx->lanes[l]); ZKC_HIP_CHECK(ctx, hipEventRecord(CS.ev_done[l], done_on_st_lane == l ? ls.st : ls.fin)); }      // only the lanes this call ran on: another lane's streams carry another call
    CS.B = B; CS.pending = true;
    return ZKC_OK;
}
static std::atomic<unsigned long long> g_early_retries{0};
extern "C" unsigned long long zkc_debug_early_retries(void) { return g_early_retries.load(); }
// second half of a batch call: wait for call slot cs, copy proofs (B x 256 B) and public signals (B x nPublic x 32 B, may be NULL) out of the pinned staging.
// Takes no context lock while it waits, so that the next call's begin (the other slot) can run meanwhile.
int zkc::prove_batch_finish(zkc_zkey* zk, int cs, uint8_t* proofs, uint8_t* publics) {
    if (!zk || !proofs || cs < 0 || cs >= CALL_SLOTS) return zkc_fail(zk ? zk->ctx : nullptr, ZKC_ERR_BAD_ARG, "prove_batch_finish: bad argument");
    zkc_zkey::CallSlot& CS = zk->call[cs];
    if (!CS.pending) return zkc_fail(zk->ctx, ZKC_ERR_BAD_ARG, "prove_batch_finish: no call in flight on this slot");
    hipError_t e = hipSuccess;
    for (int l = 0; l < zk->nlanes && e == hipSuccess; l++) if (CS.lanes_used >> l & 1) e = zkc_wait_event(CS.ev_done[l], CS.B <= 2 ? 6000u : 0u);      // a call of one or two proofs is somebody's latency: poll through it
    CS.pending = false;
    if (e != hipSuccess) { ZKC_LOCK(zk->ctx); return zkc_fail(zk->ctx, ZKC_ERR_HIP, std::string("prove_batch_finish: ") + hipGetErrorString(e)); }
    if (CS.early_n) {                // the pass was laid out from the inputs' depths before its witness existed: the fold check of the finished witness has to agree
        const WitnessLayout L = WitnessLayout::make(zk->nLevels);
        for (int q = 0; q < CS.early_n; q++) {
            if ((int32_t)CS.h_early[q] != ZKC_W_OK) continue;                   // a rejected voter: its proof is discarded whatever it is
            for (int t = 0; t < 2; t++) {
                const uint32_t* f = CS.h_early + CS.early_cap + ((size_t)q * 2 + t) * L.n;
                int D = 0; for (int g = 0; g < L.n - 1; g++) if (f[g]) D = g + 1;
                if (f[L.n - 1] || D > (int)CS.early_depth[2 * q + t]) {
                    // a witness that was computed elsewhere and does not carry the template below its own sibling depth (a valid witness of this circuit always does): nothing
                    // is wrong with the call, only with the shortcut -- the same call once more, laid out from its fold flags
                    if (!CS.arg_inputs) { g_early_retries++; int rc = prove_batch_begin(zk, cs, CS.arg_wtns, CS.arg_nw, CS.B, CS.h_rs_copy.data(), CS.arg_publics, nullptr, nullptr, CS.arg_lane0, nullptr, nullptr, true); if (rc) return rc; return prove_batch_finish(zk, cs, proofs, publics); }
                    ZKC_LOCK(zk->ctx); return zkc_fail(zk->ctx, ZKC_ERR_GENERIC, "prove_batch_finish: the witness does not fold at the depth its inputs gave (internal error)");
                }
            }
        }
    }
    memcpy(proofs, CS.h_out, 256ull * CS.B);
    // the proofs of a small pass (one or two proofs: a call of that size, or the stub at the end of a larger one) arrive as XYZZ and are divided here: three inversions,
    // microseconds on a core, ~0.2 ms at the end of the device's chain
    for (int q = 0; q < CS.B; q++) if (CS.as_xyzz[q]) {
        const uint8_t* x = CS.h_xyzz + 512 * (size_t)q; uint8_t* o = proofs + 256 * (size_t)q;
        G1XYZZ a, c; G2XYZZ b; memcpy(&a, x, 128); memcpy(&b, x + 128, 256); memcpy(&c, x + 384, 128);
        g1_to_std(o, xyzz_to_affine_gcd(a)); g2_to_std(o + 64, xyzz_to_affine_gcd(b)); g1_to_std(o + 192, xyzz_to_affine_gcd(c));
    }
    if (publics) memcpy(publics, CS.h_out + 256ull * CS.cap, 32ull * zk->nPub * CS.B);
    return ZKC_OK;
}
// the synchronous form: begin + finish on slot 0 under the context lock
static int prove_batch_impl(zkc_zkey* zk, const void* d_wtns, uint32_t nWitness, int B, const uint8_t* rs, uint8_t* proofs, uint8_t* publics,
                            const void* d_inputs, int32_t* d_status) {
    if (!zk || !proofs) return zkc_fail(zk ? zk->ctx : nullptr, ZKC_ERR_BAD_ARG, "zkc_prove_batch_dev: bad argument");
    ZKC_LOCK(zk->ctx);
    for (int c = 1; c < CALL_SLOTS; c++) if (zk->call[c].pending) return zkc_fail(zk->ctx, ZKC_ERR_BAD_ARG, "zkc_prove_batch_dev: the key has a split call in flight (proving service)");
    int rc = prove_batch_begin(zk, 0, d_wtns, nWitness, B, rs, publics != nullptr, d_inputs, d_status);
    if (rc) return rc;
    return prove_batch_finish(zk, 0, proofs, publics);
}

// the two halves as C entry points (include/zkcensus.h): what a caller that pipelines batch after batch uses
extern "C" int zkc_batch_begin(zkc_zkey* zk, int slot, const void* d_inputs, int B, void* d_wtns, int32_t* d_status, const uint8_t* rs) {
    if (!zk || !d_wtns || (d_inputs && !d_status) || slot < 0 || slot >= CALL_SLOTS) return zkc_fail(zk ? zk->ctx : nullptr, ZKC_ERR_BAD_ARG, "zkc_batch_begin: bad argument");
    if (d_inputs && zk->nLevels < 0) return zkc_fail(zk->ctx, ZKC_ERR_BAD_ARG, "zkc_batch_begin: the key is not a ZkFranchiseProofCircuit(nLevels) key; compute the witness elsewhere and pass d_inputs = NULL");
    return prove_batch_begin(zk, slot, d_wtns, zk->nVars, B, rs, true, d_inputs, d_status);
}
extern "C" int zkc_batch_finish(zkc_zkey* zk, int slot, uint8_t* proofs, uint8_t* publics) { return prove_batch_finish(zk, slot, proofs, publics); }

// B witnesses resident in HBM -> B proofs.  rs: B x 64 B (r || s).  proofs: B x 256 B, publics: B x nPublic x 32 B (host).
extern "C" int zkc_prove_batch_dev(zkc_zkey* zk, const void* d_wtns, uint32_t nWitness, int B, const uint8_t* rs, uint8_t* proofs, uint8_t* publics) {
    return prove_batch_impl(zk, d_wtns, nWitness, B, rs, proofs, publics, nullptr, nullptr);
}
// groth16.fullProve for a batch: B input blocks (334 x 32 B each, census.circom:51-67 order) resident in HBM -> witnesses (left in d_wtns,
// B x nWires x 32 B), per-voter circuit status (d_status, ZKC_W_*) and proofs.  Witness generation of pass p+1 overlaps the MSMs of pass p.
extern "C" int zkc_fullprove_batch_dev(zkc_zkey* zk, const void* d_inputs, int B, void* d_wtns, int32_t* d_status, const uint8_t* rs, uint8_t* proofs, uint8_t* publics) {
    if (!zk || !d_inputs || !d_status) return zkc_fail(zk ? zk->ctx : nullptr, ZKC_ERR_BAD_ARG, "zkc_fullprove_batch_dev: bad argument");
    if (zk->nLevels < 0) return zkc_fail(zk->ctx, ZKC_ERR_BAD_ARG, "zkc_fullprove_batch_dev: the key is not a ZkFranchiseProofCircuit(nLevels) key; compute the witness elsewhere and call zkc_prove_batch_dev");
    return prove_batch_impl(zk, d_wtns, zk->nVars, B, rs, proofs, publics, d_inputs, d_status);
}

extern "C" int zkc_prove_dev(zkc_zkey* zk, const void* d_wtns, uint32_t nWitness, const uint8_t r32[32], const uint8_t s32[32],
                             uint8_t proof[256], uint8_t* public_out) {
    if (!r32 || !s32) return zkc_fail(zk ? zk->ctx : nullptr, ZKC_ERR_BAD_ARG, "zkc_prove_dev: bad argument");
    uint8_t rs[64]; memcpy(rs, r32, 32); memcpy(rs + 32, s32, 32);
    return zkc_prove_batch_dev(zk, d_wtns, nWitness, 1, rs, proof, public_out);
}

extern "C" int zkc_prove(zkc_zkey* zk, const void* wtns, uint32_t nWitness, const uint8_t r32[32], const uint8_t s32[32],
                         uint8_t proof[256], uint8_t* public_out) {
    if (!zk || !wtns) return zkc_fail(zk ? zk->ctx : nullptr, ZKC_ERR_BAD_ARG, "zkc_prove: bad argument");
    zkc_ctx* ctx = zk->ctx;
    ZKC_LOCK(ctx);
    if (nWitness != zk->nVars) return zkc_fail(ctx, ZKC_ERR_INVALID_WITNESS_LENGTH, "Invalid witness length. Circuit: " + std::to_string(zk->nVars) + ", witness: " + std::to_string(nWitness));
    ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int rc = zkc_ensure(ctx, &ctx->d_scratch_out, &ctx->scratch_out_sz, 32ull * nWitness); if (rc) return rc;
    ZKC_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_scratch_out, wtns, 32ull * nWitness, hipMemcpyHostToDevice, ctx->stream));
    return zkc_prove_dev(zk, ctx->d_scratch_out, nWitness, r32, s32, proof, public_out);
}
