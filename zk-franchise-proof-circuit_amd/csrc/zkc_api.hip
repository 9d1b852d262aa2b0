// zkc_api.hip -- context + witness entry points of the C ABI (include/zkcensus.h).  Product code: there is no
// CPU fallback here -- without a working HIP device every call fails with ZKC_ERR_HIP.
#include "zkc_internal.h"
#include "zkc_f29.h"
#include <array>
#include <cstring>
#include <ctime>
#include <map>
#include <mutex>
#include "../../include/zkc_poseidon_constants.inc"

using namespace zkc;

extern "C" __global__ void zkc_witness_chains(WitnessLayout L, PoseidonTable tab, const uint32_t* inputs, uint32_t* wtns, int32_t* status, int B, int tmpl_mode);
extern "C" __global__ void zkc_witness_chains_wave(WitnessLayout L, PoseidonTable tab, const uint32_t* inputs, uint32_t* wtns, int32_t* status, int B, int tmpl_mode);
// a wave per chain halves the latency of a chain and costs 64 times its issue slots: worth it while the chains alone cannot fill the part
// (5 % of a pass' VALU work for 1024 voters).  Inside the batch pipeline, where the chains run underneath the MSMs, only small launches take it; a
// stand-alone zkc_witness[_dev] call has nothing to hide behind and takes it up to 1024 voters (3072 waves on 1024 SIMDs).
static constexpr int ZKC_WITNESS_WAVE_MAX_B = 128, ZKC_WITNESS_WAVE_MAX_B_ALONE = 1024;
extern "C" __global__ void zkc_witness_fill(const uint4* tmpl, uint4* wtns, int nWires, int B);
extern "C" __global__ void zkc_witness_tostd(uint32_t* wtns, size_t nwires_total);

static thread_local std::string g_create_err;

int zkc_fail(zkc_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_create_err = msg;
    return code;
}
hipError_t zkc_wait_event(hipEvent_t ev, unsigned spin_us) {
    static const bool spin = getenv("ZKC_SPIN_WAIT") != nullptr;
    if (spin) return hipEventSynchronize(ev);
    timespec t0; if (spin_us) clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0;; i++) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        if (i < 48) continue;
        if (spin_us) { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); if ((t.tv_sec - t0.tv_sec) * 1000000ll + (t.tv_nsec - t0.tv_nsec) / 1000 < (long long)spin_us) continue; spin_us = 0; }
        timespec ts{0, 50000}; nanosleep(&ts, nullptr);
    }
}
hipError_t zkc_wait_stream(hipStream_t st, hipEvent_t scratch_ev) {
    const hipError_t e = hipEventRecord(scratch_ev, st);
    return e != hipSuccess ? e : zkc_wait_event(scratch_ev);
}
void zkc_ctx_lanes_destroy(zkc_ctx* ctx);      // zkc_prove.hip
int zkc_ensure(zkc_ctx* ctx, void** p, size_t* cur, size_t need) {
    if (*cur >= need) return ZKC_OK;
    if (*p) { ZKC_HIP_CHECK(ctx, hipFree(*p)); *p = nullptr; *cur = 0; }
    ZKC_HIP_CHECK(ctx, hipMalloc(p, need));
    *cur = need;
    return ZKC_OK;
}

void zkc_verify_ws_trim(zkc_ctx* ctx, size_t keep_bytes) {
    size_t total = 0; for (size_t b : ctx->vws_sz) total += b;
    if (total <= keep_bytes && keep_bytes) return;
    for (int i = 0; i < zkc_ctx::VWS_N; i++) { if (ctx->vws[i]) (void)hipFree(ctx->vws[i]); ctx->vws[i] = nullptr; ctx->vws_sz[i] = 0; }
}

extern "C" __global__ void zkc_poseidon_batch_kernel(PoseidonTable tab, const uint32_t* in, uint32_t* out, int nin, size_t B);

static hipEvent_t prof_event(zkc_ctx* ctx) {
    if (!ctx->prof.free_events.empty()) { hipEvent_t e = ctx->prof.free_events.back(); ctx->prof.free_events.pop_back(); return e; }
    hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
}
zkc_prof_scope::zkc_prof_scope(zkc_ctx* c, int category, uint64_t alg_bytes, hipStream_t stream) : ctx(c), cat(category), st(stream ? stream : (c ? c->stream : nullptr)) {
    on = c && ((c->prof.mask >> category) & 1u);
    if (!on) return;
    a = prof_event(c); b = prof_event(c);
    c->prof.bytes[cat] += alg_bytes; c->prof.launches[cat] += 1;
    (void)hipEventRecord(a, st);
}
zkc_prof_scope::~zkc_prof_scope() {
    if (!on) return;
    (void)hipEventRecord(b, st);
    ctx->prof.pending.push_back({a, b, cat});
}
extern "C" int zkc_profile_enable(zkc_ctx* ctx, uint32_t mask) {
    if (!ctx) return ZKC_ERR_BAD_ARG;
    ZKC_LOCK(ctx);
    ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device));          // the counter below must live on THIS context's device (a pool leaves the thread on its last device)
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& r : ctx->prof.pending) { ctx->prof.free_events.push_back(r.a); ctx->prof.free_events.push_back(r.b); }
    ctx->prof.pending.clear();
    for (int i = 0; i < ZKC_PROF_NCAT; i++) { ctx->prof.ms[i] = 0; ctx->prof.launches[i] = 0; ctx->prof.bytes[i] = 0; }
    if (!ctx->d_prof_entries && hipMalloc((void**)&ctx->d_prof_entries, 8) != hipSuccess) ctx->d_prof_entries = nullptr;
    if (ctx->d_prof_entries) (void)hipMemset(ctx->d_prof_entries, 0, 8);
    ctx->prof.mask = mask;
    return ZKC_OK;
}
extern "C" int zkc_profile_read(zkc_ctx* ctx, int cat, double* total_ms, uint64_t* launches, uint64_t* alg_bytes) {
    if (!ctx || cat < 0 || cat >= ZKC_PROF_NCAT) return ZKC_ERR_BAD_ARG;
    ZKC_LOCK(ctx);
    ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream2)); ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->fin_stream));
    for (auto& r : ctx->prof.pending) {
        float ms = 0; (void)hipEventElapsedTime(&ms, r.a, r.b); ctx->prof.ms[r.cat] += ms;
        ctx->prof.free_events.push_back(r.a); ctx->prof.free_events.push_back(r.b);
    }
    ctx->prof.pending.clear();
    if (cat == ZKC_PROF_MSM_G1_STREAMED && ctx->d_prof_entries) {      // launches of this bytes-only category = mixed additions the G1 accumulation really performed
        unsigned long long v = 0; (void)hipMemcpyAsync(&v, ctx->d_prof_entries, 8, hipMemcpyDeviceToHost, ctx->stream); (void)hipStreamSynchronize(ctx->stream); ctx->prof.launches[cat] = v;
    }
    if (total_ms) *total_ms = ctx->prof.ms[cat]; if (launches) *launches = ctx->prof.launches[cat]; if (alg_bytes) *alg_bytes = ctx->prof.bytes[cat];
    return ZKC_OK;
}

extern "C" int zkc_poseidon_batch(zkc_ctx* ctx, int n_inputs, const void* inputs, size_t B, void* out) {
    if (!ctx || !inputs || !out || n_inputs < 2 || n_inputs > 4 || B == 0) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "zkc_poseidon_batch: bad argument");
    ZKC_LOCK(ctx);
    ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = zkc_ensure(ctx, &ctx->d_scratch_in, &ctx->scratch_in_sz, B * n_inputs * 32))) return rc;
    if ((rc = zkc_ensure(ctx, &ctx->d_scratch_out, &ctx->scratch_out_sz, B * 32))) return rc;
    ZKC_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_scratch_in, inputs, B * n_inputs * 32, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(zkc_poseidon_batch_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, ctx->stream, ctx->ptab, (const uint32_t*)ctx->d_scratch_in,
                       (uint32_t*)ctx->d_scratch_out, n_inputs, B);
    ZKC_HIP_CHECK(ctx, hipGetLastError());
    ZKC_HIP_CHECK(ctx, hipMemcpyAsync(out, ctx->d_scratch_out, B * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ZKC_OK;
}

__global__ void zkc_status_combine(const int32_t* __restrict__ s3, int32_t* __restrict__ s, int B) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    // report in the order the circuit states its asserts: weight (misc), sik tree, census tree, nullifier (misc)
    int m = s3[3 * b + 2], c = s3[3 * b + 0], k = s3[3 * b + 1], r = 0;
    if (m == ZKC_W_ERR_INPUT_RANGE) r = m;
    else if (m == ZKC_W_ERR_WEIGHT) r = m;
    else if (k) r = k; else if (c) r = c; else r = m;
    s[b] = r;
}

template <size_t N>
static void conv_consts(std::vector<Fr>& out, const unsigned long long (&src)[N][4]) {
    for (size_t i = 0; i < N; i++) {
        uint32_t s[8];
        for (int k = 0; k < 4; k++) { s[2 * k] = (uint32_t)src[i][k]; s[2 * k + 1] = (uint32_t)(src[i][k] >> 32); }
        out.push_back(fp_from_std<FrParams>(s));
    }
}

// [r5] Hardware queues.  ROCm maps a process' HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and a queue executes in order: a stream whose queue also
// carries another stream's pending barrier packet (a hipStreamWaitEvent that has not fired) runs behind it.  This pipeline keeps seven to sixteen streams busy with cross-
// stream waits; on four queues the batch path lost 5 % (3104 -> 3267 proofs/s with 8, same box) and the proving service's lanes ran one after another instead of side by
// side (profiles/r05_service_hw_queues.txt).  The variable is read when the HIP runtime initialises, so it is set here, at library load, unless the host has set it; a host
// that initialises HIP before loading this library (or wants another value) sets GPU_MAX_HW_QUEUES itself -- bench.py and the N-API addon do.
__attribute__((constructor)) static void zkc_runtime_defaults() { setenv("GPU_MAX_HW_QUEUES", "24", 0); }

extern "C" int zkc_ctx_create(int device, zkc_ctx** out) {
    if (!out) return zkc_fail(nullptr, ZKC_ERR_BAD_ARG, "out == NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return zkc_fail(nullptr, ZKC_ERR_HIP, "no HIP device: libzkcensus has no CPU path (the CPU restatement lives in oracle/ and is test-only)");
    if (device < 0 || device >= ndev) return zkc_fail(nullptr, ZKC_ERR_BAD_ARG, "bad device ordinal");
    zkc_ctx* ctx = new zkc_ctx();
    ctx->device = device;
    auto fail = [&](hipError_t e, const char* what) { g_create_err = std::string(what) + ": " + hipGetErrorString(e); delete ctx; return (int)ZKC_ERR_HIP; };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return fail(e, "hipSetDevice");
    // [r5] host threads that wait for this device SLEEP (hipStreamSynchronize and friends yield to the driver's interrupt instead of spinning on a flag): a rank of an
    // 8-GPU run has two cores' worth of CPU time on these boxes (cpu.max = 16 cores for the container, profiles/r04_cpu_baseline_scaling.json).  Best effort -- a host that
    // has already initialised the device with other flags (torch) keeps them; the events this library waits on carry hipEventBlockingSync themselves.  ZKC_SPIN_WAIT=1: leave the default.
    { const char* eb = getenv("ZKC_DEVICE_BLOCKING_SYNC"); if (eb && atoi(eb) == 1 && hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess) (void)hipGetLastError(); }
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
    if ((e = hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate(2)");
    if ((e = hipStreamCreateWithFlags(&ctx->fin_stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate(fin)");
    // Poseidon parameter tables -> Montgomery form -> HBM (about 1.6k Fr = 51 KB; L2/scalar-cache resident)
    std::vector<Fr> all; size_t off[12]; int k = 0;
    off[k++] = all.size(); conv_consts(all, ZKC_POSEIDON_C3); off[k++] = all.size(); conv_consts(all, ZKC_POSEIDON_S3);
    off[k++] = all.size(); conv_consts(all, ZKC_POSEIDON_M3); off[k++] = all.size(); conv_consts(all, ZKC_POSEIDON_P3);
    off[k++] = all.size(); conv_consts(all, ZKC_POSEIDON_C4); off[k++] = all.size(); conv_consts(all, ZKC_POSEIDON_S4);
    off[k++] = all.size(); conv_consts(all, ZKC_POSEIDON_M4); off[k++] = all.size(); conv_consts(all, ZKC_POSEIDON_P4);
    off[k++] = all.size(); conv_consts(all, ZKC_POSEIDON_C5); off[k++] = all.size(); conv_consts(all, ZKC_POSEIDON_S5);
    off[k++] = all.size(); conv_consts(all, ZKC_POSEIDON_M5); off[k++] = all.size(); conv_consts(all, ZKC_POSEIDON_P5);
    if ((e = hipMalloc(&ctx->d_ptab_mem, all.size() * sizeof(Fr))) != hipSuccess) return fail(e, "hipMalloc(poseidon)");
    if ((e = hipMemcpy(ctx->d_ptab_mem, all.data(), all.size() * sizeof(Fr), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy(poseidon)");
    const Fr* base = (const Fr*)ctx->d_ptab_mem;
    {   // radix-2^29 copies for the witness chains (host arithmetic: zkc_f29.h is host + device)
        std::vector<uint32_t> l29(all.size() * 12, 0);
        for (size_t i = 0; i < all.size(); i++) { uint32_t t9[9]; f29_from_fp_shl5(t9, all[i].v); f29_mul<FrParams>(&l29[12 * i], t9, F29K<FrParams>::one.l); }
        if ((e = hipMalloc(&ctx->d_ptab29_mem, l29.size() * 4)) != hipSuccess) return fail(e, "hipMalloc(poseidon29)");
        if ((e = hipMemcpy(ctx->d_ptab29_mem, l29.data(), l29.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy(poseidon29)");
        ctx->ptab.base = base; ctx->ptab.base29 = (const uint32_t*)ctx->d_ptab29_mem;
        // K29 (PoseidonTable): per partial round of t = 3, 4 the products S_r[.] * C[5 t + r], canonical
        std::vector<uint32_t> k29; size_t koff[2];
        for (int t = 3; t <= 4; t++) {
            koff[t - 3] = k29.size();
            const int RP = t == 3 ? 57 : 56; const size_t oC = off[(t - 3) * 4 + 0], oS = off[(t - 3) * 4 + 1];
            for (int r = 0; r < RP; r++) for (int k = 0; k < t; k++) {
                const size_t si = oS + (size_t)(2 * t - 1) * r + (k ? t + k - 1 : 0), ci = oC + 5 * t + r;
                const Fr prod = all[si] * all[ci];
                uint32_t t9[9], o9[12] = {0}; f29_from_fp_shl5(t9, prod.v); f29_mul<FrParams>(o9, t9, F29K<FrParams>::one.l); f29_reduce_small<FrParams>(o9);
                k29.insert(k29.end(), o9, o9 + 12);
            }
        }
        if ((e = hipMalloc(&ctx->d_pk29_mem, k29.size() * 4)) != hipSuccess) return fail(e, "hipMalloc(poseidon K29)");
        if ((e = hipMemcpy(ctx->d_pk29_mem, k29.data(), k29.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy(poseidon K29)");
        for (int t = 3; t <= 4; t++) ctx->ptab.K29[t] = (const uint32_t*)ctx->d_pk29_mem + koff[t - 3];
    }
    for (int t = 3; t <= 5; t++) {
        ctx->ptab.C[t] = base + off[(t - 3) * 4 + 0]; ctx->ptab.S[t] = base + off[(t - 3) * 4 + 1];
        ctx->ptab.M[t] = base + off[(t - 3) * 4 + 2]; ctx->ptab.P[t] = base + off[(t - 3) * 4 + 3];
    }
    *out = ctx;
    return ZKC_OK;
}
extern "C" void zkc_ctx_destroy(zkc_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto& r : ctx->prof.pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : ctx->prof.free_events) (void)hipEventDestroy(e);
    for (auto& kv : ctx->tmpl) (void)hipFree(kv.second);
    for (auto& kv : ctx->ntt_tw) { if (kv.second.fwd) (void)hipFree(kv.second.fwd); if (kv.second.inv) (void)hipFree(kv.second.inv); if (kv.second.ninv) (void)hipFree(kv.second.ninv); }
    if (ctx->d_ptab_mem) (void)hipFree(ctx->d_ptab_mem);
    if (ctx->d_ptab29_mem) (void)hipFree(ctx->d_ptab29_mem);
    if (ctx->d_pk29_mem) (void)hipFree(ctx->d_pk29_mem);
    if (ctx->d_scratch_in) (void)hipFree(ctx->d_scratch_in);
    if (ctx->d_scratch_out) (void)hipFree(ctx->d_scratch_out);
    if (ctx->d_status3) (void)hipFree(ctx->d_status3);
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->d_prof_entries) (void)hipFree(ctx->d_prof_entries);
    zkc_verify_ws_trim(ctx, 0);
    if (ctx->ev_vws_up) (void)hipEventDestroy(ctx->ev_vws_up);
    if (ctx->ev_vws_lines) (void)hipEventDestroy(ctx->ev_vws_lines);
    for (auto& ls : ctx->lane_streams) for (hipStream_t q : {ls.st, ls.st2, ls.fin, ls.red}) if (q) (void)hipStreamSynchronize(q);      // (the streams are the DEVICE's and stay: zkc_lane_streams)
    zkc_ctx_lanes_destroy(ctx);
    if (ctx->ev_acc_chain) (void)hipEventDestroy(ctx->ev_acc_chain);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->fin_stream) (void)hipStreamDestroy(ctx->fin_stream);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}
// the three streams of lane l of this context (buildABC / NTT / G1 MSM ; G2 MSM ; blinding + D2H), created on first use; with_red: the optional bucket-reduction stream too
// [r5'] ... of this DEVICE: every context of the process on one device gets the same twelve streams (made once, kept for the life of the process).  A host that holds a
// context of its own beside the proving service's -- bench.py does, so does any process that both batch-proves and serves single calls -- would otherwise run 2 x 12 lane
// streams, and past ~24 busy hardware queues the service's lanes queue behind one another again whatever GPU_MAX_HW_QUEUES says (measured with 32 and 48: the service legs
// of bench.py at 1 250-1 600 proofs/s instead of 2 800-3 100; profiles/r05_service_layout_sweeps.json).  Sharing a stream between contexts only adds ordering: each context
// still has its own lanes' work space, events and lock.
int zkc_lane_streams(zkc_ctx* ctx, int l, bool with_red, zkc_ctx::LaneStreams* out) {
    if (!ctx || l < 0 || l >= 4 || !out) return ZKC_ERR_BAD_ARG;
    static std::mutex mu; static std::map<int, std::array<zkc_ctx::LaneStreams, 4>> by_device;
    std::lock_guard<std::mutex> g(mu);
    zkc_ctx::LaneStreams& ls = by_device[ctx->device][l];
    for (hipStream_t* q : {&ls.st, &ls.st2, &ls.fin}) if (!*q) ZKC_HIP_CHECK(ctx, hipStreamCreateWithFlags(q, hipStreamNonBlocking));
    if (with_red && !ls.red) ZKC_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ls.red, hipStreamNonBlocking));
    ctx->lane_streams[l] = ls; *out = ls; return ZKC_OK;
}
extern "C" const char* zkc_last_error(const zkc_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }
extern "C" void* zkc_ctx_stream(zkc_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
extern "C" int zkc_circuit_n_inputs(int nLevels) { return nLevels < 3 ? 0 : WitnessLayout::make(nLevels).nInputs; }
extern "C" int zkc_circuit_n_wires(int nLevels) { return nLevels < 3 ? 0 : WitnessLayout::make(nLevels).nWires; }

// The voter-independent part of the witness (empty-level Poseidon(0,0) traces, oldKey = 0 decomposition) is computed
// once per (ctx, nLevels) ON THE DEVICE by running the chain kernel in template mode over an all-zero voter.
static int get_template(zkc_ctx* ctx, const WitnessLayout& L, uint32_t** out) {
    auto it = ctx->tmpl.find(L.nL);
    if (it != ctx->tmpl.end()) { *out = it->second; return ZKC_OK; }
    uint32_t *d_t = nullptr, *d_in = nullptr; int32_t* d_st = nullptr;
    const int rc = [&]() -> int {
        ZKC_HIP_CHECK(ctx, hipMalloc(&d_t, (size_t)L.nWires * 32));
        ZKC_HIP_CHECK(ctx, hipMalloc(&d_in, (size_t)L.nInputs * 32));
        ZKC_HIP_CHECK(ctx, hipMalloc(&d_st, 3 * sizeof(int32_t)));
        ZKC_HIP_CHECK(ctx, hipMemsetAsync(d_t, 0, (size_t)L.nWires * 32, ctx->stream));
        ZKC_HIP_CHECK(ctx, hipMemsetAsync(d_in, 0, (size_t)L.nInputs * 32, ctx->stream));
        hipLaunchKernelGGL(zkc_witness_chains_wave, dim3(3), dim3(64), 0, ctx->stream, L, ctx->ptab, d_in, d_t, d_st, 1, 1);
        hipLaunchKernelGGL(zkc_witness_tostd, dim3((L.nWires + 255) / 256), dim3(256), 0, ctx->stream, d_t, (size_t)L.nWires);
        ZKC_HIP_CHECK(ctx, hipGetLastError());
        ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        return ZKC_OK;
    }();
    if (d_in) (void)hipFree(d_in);
    if (d_st) (void)hipFree(d_st);
    if (rc) { if (d_t) (void)hipFree(d_t); return rc; }                  // a failed build leaves nothing behind
    ctx->tmpl[L.nL] = d_t; *out = d_t;
    return ZKC_OK;
}

uint32_t* zkc_get_template(zkc_ctx* ctx, int nLevels) {
    uint32_t* t = nullptr; WitnessLayout L = WitnessLayout::make(nLevels);
    return get_template(ctx, L, &t) == ZKC_OK ? t : nullptr;
}

static int witness_dev3(zkc_ctx* ctx, const WitnessLayout& L, const void* d_inputs, int B, void* d_wtns, int32_t* d_status3, bool alone, hipStream_t st = nullptr) {
    if (!st) st = ctx->stream;
    uint32_t* tmpl; int rc = get_template(ctx, L, &tmpl); if (rc) return rc;
    const size_t total = (size_t)L.nWires * 2 * (size_t)B;
    int fill_blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 16);
    zkc_prof_scope _ps(ctx, ZKC_PROF_WITNESS, (uint64_t)B * ((uint64_t)L.nWires + L.nInputs) * 32, st);
    hipLaunchKernelGGL(zkc_witness_fill, dim3(fill_blocks), dim3(256), 0, st, (const uint4*)tmpl, (uint4*)d_wtns, L.nWires, B);
    ZKC_HIP_CHECK(ctx, hipGetLastError());
    if (B <= (alone ? ZKC_WITNESS_WAVE_MAX_B_ALONE : ZKC_WITNESS_WAVE_MAX_B))             // one wave per (voter, chain): latency form
        hipLaunchKernelGGL(zkc_witness_chains_wave, dim3(3 * B), dim3(64), 0, st, L, ctx->ptab, (const uint32_t*)d_inputs, (uint32_t*)d_wtns, d_status3, B, 0);
    else                                         // one lane per (voter, chain), lanes of a wave share the chain kind when B % 64 == 0 (harmless otherwise): throughput form
        hipLaunchKernelGGL(zkc_witness_chains, dim3((3 * B + 63) / 64), dim3(64), 0, st, L, ctx->ptab, (const uint32_t*)d_inputs, (uint32_t*)d_wtns, d_status3, B, 0);
    {   // the chains leave their wires in Montgomery form, marked: convert them (every lane busy, unlike the chains)
        const size_t nw = (size_t)L.nWires * (size_t)B;
        hipLaunchKernelGGL(zkc_witness_tostd, dim3((unsigned)std::min<size_t>((nw + 255) / 256, 256 * 64)), dim3(256), 0, st, (uint32_t*)d_wtns, nw);
    }
    ZKC_HIP_CHECK(ctx, hipGetLastError());
    return ZKC_OK;
}

// one chunk of voters, enqueued on ctx->stream without any synchronisation (the full-prove pipeline of zkc_prove.hip issues a chunk per pass)
int zkc_witness_chunk_async(zkc_ctx* ctx, int nLevels, const void* d_inputs, int B, void* d_wtns, int32_t* d_status3, int32_t* d_status, hipStream_t st) {
    if (!st) st = ctx->stream;
    WitnessLayout L = WitnessLayout::make(nLevels);
    int rc = witness_dev3(ctx, L, d_inputs, B, d_wtns, d_status3, false, st); if (rc) return rc;
    hipLaunchKernelGGL(zkc_status_combine, dim3((B + 255) / 256), dim3(256), 0, st, d_status3, d_status, B);
    ZKC_HIP_CHECK(ctx, hipGetLastError());
    return ZKC_OK;
}

extern "C" int zkc_witness_dev(zkc_ctx* ctx, int nLevels, const void* d_inputs, int B, void* d_wtns, int32_t* d_status) {
    if (!ctx || !d_inputs || !d_wtns || !d_status || B <= 0 || nLevels < 3 || nLevels > 253) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "zkc_witness_dev: bad argument");
    ZKC_LOCK(ctx);
    ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    WitnessLayout L = WitnessLayout::make(nLevels);
    int rc = zkc_ensure(ctx, (void**)&ctx->d_status3, &ctx->status3_n, (size_t)B * 3 * sizeof(int32_t)); if (rc) return rc;
    rc = witness_dev3(ctx, L, d_inputs, B, d_wtns, ctx->d_status3, true); if (rc) return rc;
    hipLaunchKernelGGL(zkc_status_combine, dim3((B + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_status3, d_status, B);
    ZKC_HIP_CHECK(ctx, hipGetLastError());
    return ZKC_OK;
}

extern "C" int zkc_witness(zkc_ctx* ctx, int nLevels, const void* inputs, int B, void* wtns, int32_t* status) {
    if (!ctx || !inputs || !wtns || !status || B <= 0 || nLevels < 3 || nLevels > 253) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "zkc_witness: bad argument");
    ZKC_LOCK(ctx);
    ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    WitnessLayout L = WitnessLayout::make(nLevels);
    const size_t in_sz = (size_t)B * L.nInputs * 32, out_sz = (size_t)B * L.nWires * 32;
    int rc;
    if ((rc = zkc_ensure(ctx, &ctx->d_scratch_in, &ctx->scratch_in_sz, in_sz))) return rc;
    if ((rc = zkc_ensure(ctx, &ctx->d_scratch_out, &ctx->scratch_out_sz, out_sz))) return rc;
    if ((rc = zkc_ensure(ctx, (void**)&ctx->d_status, &ctx->status_n, (size_t)B * sizeof(int32_t)))) return rc;
    ZKC_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_scratch_in, inputs, in_sz, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = zkc_witness_dev(ctx, nLevels, ctx->d_scratch_in, B, ctx->d_scratch_out, ctx->d_status))) return rc;
    ZKC_HIP_CHECK(ctx, hipMemcpyAsync(wtns, ctx->d_scratch_out, out_sz, hipMemcpyDeviceToHost, ctx->stream));
    ZKC_HIP_CHECK(ctx, hipMemcpyAsync(status, ctx->d_status, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < B; b++) if (status[b] != ZKC_W_OK) return zkc_fail(ctx, ZKC_ERR_WITNESS, "witness: voter " + std::to_string(b) + " failed circuit assert " + std::to_string(status[b]));
    return ZKC_OK;
}
