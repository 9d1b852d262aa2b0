// zkc_internal.h -- host-side internals of libzkcensus (product code).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <map>
#include <mutex>
#include "../../include/zkcensus.h"
#include "zkc_device.h"

#define ZKC_LOCK(ctx) std::lock_guard<std::recursive_mutex> _zkc_guard((ctx)->mu)
#define ZKC_HIP_CHECK(ctx, call)                                                                            \
    do { hipError_t _e = (call); if (_e != hipSuccess) {                                                     \
        return zkc_fail((ctx), ZKC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); } } while (0)

// Per-kernel-category HIP-event timing on the context's own stream (bench.py's roofline figure reads it).
enum { ZKC_PROF_WITNESS = 0, ZKC_PROF_MATVEC = 1, ZKC_PROF_NTT = 2, ZKC_PROF_MSM_SORT = 3, ZKC_PROF_MSM_ACC_G1 = 4, ZKC_PROF_MSM_ACC_G2 = 5,
       ZKC_PROF_MSM_REDUCE = 6, ZKC_PROF_MSM_G1_STREAMED = 7 /* bytes only: pairs left after constant folding */, ZKC_PROF_NCAT = 8 };
struct zkc_prof {
    uint32_t mask = 0;
    struct Rec { hipEvent_t a, b; int cat; };
    std::vector<Rec> pending; std::vector<hipEvent_t> free_events;
    double ms[ZKC_PROF_NCAT] = {0}; uint64_t launches[ZKC_PROF_NCAT] = {0}; uint64_t bytes[ZKC_PROF_NCAT] = {0};
};
// Thread-safety contract (include/zkcensus.h): every entry point that takes a zkc_ctx* or a zkc_zkey* locks the context's mutex for its whole
// duration, so calls on one context from several threads are serialised (one GPU pipeline per context); different contexts run concurrently.
struct zkc_ctx {
    std::recursive_mutex mu;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;        // G2 MSM pipeline (independent of buildABC/NTT): overlaps the G1 pipeline
    hipStream_t fin_stream = nullptr;     // blinding kernel + proof D2H, overlapping the next pipeline pass
    // [r5] the stream sets of the pipeline lanes belong to the CONTEXT, not to a key: every key of the context runs its lane l on the same three streams.  Streams are a
    // device resource -- each takes a hardware queue (or a share of one: GPU_MAX_HW_QUEUES), and a stream that shares its queue with another stream's barrier packet waits
    // behind it -- so four resident keys must not mean four times the streams.  Created on first use (zkc_lane_streams); [r5'] they are in fact the DEVICE's, shared by every
    // context of the process on that device and kept for the life of the process: these are this context's copies of the handles.
    struct LaneStreams { hipStream_t st = nullptr, st2 = nullptr, fin = nullptr, red = nullptr; } lane_streams[4];
    // ... and so do the lanes themselves -- the per-pass work space (transform vectors, MSM entry lists, bucket and segment arrays, partial sums: ~0.15 GB per proof in flight at
    // nLevels 160) and the events that order a lane's streams: zkc_lane[MAX_LANES] (zkc_prover.h), made on first use and grown to the largest need any key of the context has
    // shown (zkc_prove.hip lane_ensure).  Calls on one lane are ordered by its streams whatever key they prove with, so four resident keys cost four sets of TABLES, not four
    // work spaces (round 4 and the first half of round 5: 37 GB of work space per service key).
    struct zkc_lane* lanes = nullptr;
    // [r5] the G1 accumulations of a context run ONE AT A TIME, whatever lanes they are launched on: each waits for the one enqueued before it (ev_acc_chain; begins are
    // serialised by the context lock, so the chain follows the enqueue order and has no cycle).  The kernel saturates the chip's vector issue by itself (VALU-busy 1.00):
    // two of them side by side finish no earlier than one after the other, keep 2 x 3 waves per SIMD of registers away from the latency-bound kernels the lanes exist to
    // overlap, and double every launch's duration -- which is also what the roofline line divides by.  ZKC_ACC_CHAIN=0: let them overlap (A/B).
    hipEvent_t ev_acc_chain = nullptr; bool acc_chain_armed = false;
    std::string err;
    zkc::PoseidonTable ptab{};            // device pointers
    void* d_ptab_mem = nullptr;
    void* d_ptab29_mem = nullptr;         // the Poseidon constants again as radix-2^29 limbs (witness chains)
    void* d_pk29_mem = nullptr;           // PoseidonTable::K29
    std::map<int, uint32_t*> tmpl;        // nLevels -> device template witness (nWires x 8 u32)
    struct TwiddleSet { uint32_t *fwd = nullptr, *inv = nullptr; void* ninv = nullptr; };
    std::map<int, TwiddleSet> ntt_tw;     // log n -> twiddles of the stand-alone NTT entry point (zkc_ntt_dev); freed with the context
    // scratch reused by the host-buffer entry points
    void* d_scratch_in = nullptr; size_t scratch_in_sz = 0;
    void* d_scratch_out = nullptr; size_t scratch_out_sz = 0;
    int32_t* d_status3 = nullptr; size_t status3_n = 0;
    int32_t* d_status = nullptr; size_t status_n = 0;
    // [r5] the batch verifier's device buffers (zkc_verify_batch, zkc_pairing_dev.hip): points, weights, group table, fold sums, line coefficients, product-tree levels.
    // Kept between calls (a node verifies batch after batch; hipFree waits for the whole device, including any proving lanes) while they add up to at most 256 MB, released
    // at the end of a call that needed more (zkc_verify_ws_trim).
    enum { VWS_PTS = 0, VWS_RHO, VWS_IDX, VWS_GS, VWS_FOLD_TMP, VWS_FOLD_OUT, VWS_Q, VWS_LINES, VWS_TREE_A, VWS_TREE_B, VWS_BAD, VWS_N };
    void* vws[VWS_N] = {nullptr}; size_t vws_sz[VWS_N] = {0};
    hipEvent_t ev_vws_up = nullptr, ev_vws_lines = nullptr;       // the batch verifier's upload (second stream) -> its line kernel (third stream) -> the product tree (first): zkc_pairing_dev.hip
    zkc_prof prof;
    unsigned long long* d_prof_entries = nullptr;      // device counter: (digit, point) entries = group additions of the G1 passes while profiling is on
};

int zkc_fail(zkc_ctx* ctx, int code, const std::string& msg);
// [r5] Host waits that SLEEP.  On this stack a host thread inside hipEventSynchronize / hipStreamSynchronize burns its core: the completion signals are ROCr BusyWaitSignals,
// which spin whatever wait state is asked for -- hipEventBlockingSync events and hipDeviceScheduleBlockingSync included (rocgdb backtrace in profiles/r05_host_cpu_threads.txt:
// BusyWaitSignal::WaitRelaxed under hsa_signal_wait_scacquire).  A rank of an 8-GPU run has two of the container's 16 cores' worth of CPU time, so the
// waits on the proving path poll the event instead: a few dozen hipEventQuery calls back to back (a wait that is nearly over costs nothing extra), then one query per 50 us
// nap.  At most ~0.1 ms later than a spinning wait, per wait: 12 waits per 1024-proof step of 320 ms.  ZKC_SPIN_WAIT=1: hipEventSynchronize as before (A/B).
hipError_t zkc_wait_event(hipEvent_t ev, unsigned spin_us = 0);        // spin_us: keep polling back to back for that long before the naps start (a lone caller's 3 ms proof: the nap would add 2 % to its latency)
hipError_t zkc_wait_stream(hipStream_t st, hipEvent_t scratch_ev);      // record scratch_ev on st, then zkc_wait_event (scratch_ev: any event of the caller's that is not otherwise in flight)
int zkc_ensure(zkc_ctx* ctx, void** p, size_t* cur, size_t need);
inline int zkc_vws(zkc_ctx* ctx, int which, size_t need, void** out) { const int rc = zkc_ensure(ctx, &ctx->vws[which], &ctx->vws_sz[which], need); *out = ctx->vws[which]; return rc; }
void zkc_verify_ws_trim(zkc_ctx* ctx, size_t keep_bytes);      // free the verifier's buffers when they hold more than keep_bytes (0: always)
// RAII bracket: records two events around the launches made while it is alive when category `cat` is enabled
struct zkc_prof_scope {
    zkc_ctx* ctx; int cat; hipEvent_t a = nullptr, b = nullptr; bool on; hipStream_t st;
    zkc_prof_scope(zkc_ctx* c, int category, uint64_t alg_bytes, hipStream_t stream = nullptr);
    ~zkc_prof_scope();
};
