// zkc_prover.h -- device-resident proving key and the stage launchers shared by the prover translation units.
#pragma once
#include "zkc_internal.h"
#include "zkc_curve.h"

namespace zkc {
// Pippenger window bits per section (signed digits).  Because the bases are pre-shifted per window (T[w][i] = 2^(c w) P_i), a
// digit d of ANY window lands in the same bucket d: a job has 2^(c-1) buckets in total, not per window, so large windows are cheap.
// H (2^17 random scalars): c = 17 -> 15 additions per scalar into 65536 buckets (about 30 entries each); the witness sections A, B1,
// C, B2 (8-11 k wires after constant folding): c = 12 -> 22 additions per scalar into 2048 buckets (same box: c = 14 -> 1798, 13 -> 1919,
// 12 -> 1958, 11 -> ~1856 proofs/s; round 2, per-job bucketing: 12 -> 3030, 13 -> 2965).
constexpr int MSM_C_BIG = 17, MSM_C_SMALL = 12;
// [r3] the G2 section of a pass of one or two proofs: 8-bit windows -> 32 additions per scalar into 128 buckets.  A lone proof's B2 MSM is ~5 000 scalars and its latency is the
// bucket REDUCTION (2048 buckets: ~30 of the ~55 G2 additions in a row); with 128 buckets one wave reduces the job.  Costs a second pre-shifted G2 table (0.6 GB at nLevels 160).
constexpr int MSM_C_G2_LONE = 8;
// [r4] the window for an MSM of W full-width scalars: W x ceil(254 / c) mixed additions plus ~4 per bucket for the reduction (2^(c-1) buckets, two full additions each,
// latency-shaped), so the best c grows with W.  Measured on circuit-shaped random R1CS (tools/gpu/csec_sweep.sh: proofs/s for c = 12 .. 17 at 15 k / 31 k / 62 k / 123 k / 246 k
// wires per section): best 13 / 15 / 15 / 16 / 17; the 12 of rounds 1-3 is 15 % behind at 62 k wires, 17 is 3 % behind at 123 k.
inline int msm_c_for(size_t W) { return W < 12000 ? 12 : W < 22000 ? 13 : W < 90000 ? 15 : W < 180000 ? 16 : 17; }
constexpr int msm_nw(int c) { return (254 + c) / c; }          // 17 -> 15 windows (255 bits), 12 -> 22 windows (264 bits)
constexpr int msm_half(int c) { return 1 << (c - 1); }         // buckets per job
// buckets per workgroup of the reduction ("virtual window"): 1024 for H (64 waves per job; 512: 1496, 1024: 1548, 2048: 1505 proofs/s),
// 256 for the witness sections (16 waves per job instead of 4: their reduction is a latency chain, not a throughput problem); passes of a few proofs take 256 and 64
// (MsmJobList::vw_big / vw_small)
constexpr int MSM_VW_MIN = 64, MSM_MAX_VW_PER_JOB = 64;
constexpr int MSM_MAX_VW_G1 = 256;                 // virtual windows per G1 job the per-job sum takes (zkc_msm_final29): H with windows of 256 buckets in a small pass
constexpr int MSM_SEG_MIN = 16;                    // ... down to this for small passes (latency of a single proof)
#ifndef ZKC_MSM_SEG
#define ZKC_MSM_SEG 128
#endif
constexpr int MSM_SEG = ZKC_MSM_SEG;               // [r4] re-measured on the round-4 pipeline (variant builds, alternating on one box): 64 -> 3181 / 3181, 128 -> 3215 / 3212, 256 -> 3217 / 3191 proofs/s                       // sorted entries per accumulation lane; with length-sorted waves (same box): 32 -> 2200, 40 -> 2222, 64 -> 2270, 128 -> 2297, 256 -> 2274, 512 -> 2220 proofs/s
constexpr int MSM_MERGE_T = 8;                     // buckets with more segments get a wave of their own before the window pass
constexpr int MSM_MAX_HEAVY = 1 << 20;
constexpr int MSM_MAX_JOBS = 512;                  // jobs per pipeline pass (proofs in flight x sections)
// [r5] pipeline lanes (stream sets + per-pass work space) and call slots of a key.  A direct caller uses one lane and slots 0 / 1 (bench.py); the proving service loads its
// keys with one lane per worker so that the calls of concurrent single-proof callers are independent stream sets and really overlap on the GPU (zkc_service.hip).
constexpr int MAX_LANES = 4, CALL_SLOTS = 4;

// One multi-scalar multiplication inside a pipeline pass: sum_j scalar[j] * P[point(j)]
struct MsmJob {
    const uint32_t* scalars;   // standard form, 8 x u32 each
    const uint32_t* vmap;      // nullptr: scalar j / point j.  else: scalar at scalars + 8*vmap[j], point vmap[j] - pt_shift
    uint32_t count;            // number of (scalar, point) pairs
    uint32_t tbl_off;          // first point of this section's pre-shifted table inside the unified point array
    uint32_t tbl_count;        // points per window in that table
    int32_t pt_shift;
    uint32_t c, nw, vw;        // window bits / windows of this job's table / buckets per virtual window
    uint32_t ent_off, win_off; // first (scalar, window) entry / first virtual window of this job inside the pass
    // bucketing (zkc_msm_sort.hip): the job's 2^(c-1) buckets are ids [bucket0, bucket0 + 2^(c-1)) (job-major); its entries are first split into
    // 2^hbits level-1 bins by the high bits of the bucket index (bins [bin0, bin0 + 2^hbits) of the pass), then each bin into 2^lbits buckets
    uint32_t bucket0, bin0, hbits, lbits, tile0, cnt0;   // tile0: first 1024-scalar tile of this job in the flattened tile list of the pass; cnt0: its (bin, tile) counters
};
// arguments of the blinding kernel (zkc_finalize.hip); everything except r1/r2/rs/out is constant per proving key
struct FinalizeArgs {
    const G1XYZZ* r1; const G2XYZZ* r2;                 // MSM results of this pass: r1[q] = H_q, r1[n + 3q + {0,1,2}] = A_q, B1_q, C_q ; r2[q]
    // constant folding, per proof: proof q folds census levels >= dc[q] and sik levels >= ds[q] (255: this pass is not folded); its constant for
    // a section is tab[0] + tab[1 + dc] + tab[1 + fold_n + ds] with tab = [base | suffix sums census tree | suffix sums sik tree] (device)
    const G1XYZZ *foldA, *foldB1, *foldC; const G2XYZZ* foldB2; int fold_n;
    uint8_t dc[MSM_MAX_JOBS / 4], ds[MSM_MAX_JOBS / 4];
    const G1Affine *tblDelta1, *tblAlpha1, *tblBeta1; const G2Affine* tblDelta2;   // 32 x 255 fixed-base tables
    G1Affine alpha1; G2Affine beta2;
    const uint8_t* rs; uint8_t* out;                    // device: nproofs x 64 (r || s) -> nproofs x 256 proof bytes
    void* scratch;                                      // finalize_scratch_bytes(nproofs) of device memory: products and window tables of the lane-per-product kernels (nullptr: one wave per task)
    // [r3] passes of one or two proofs (zkc_finalize.hip, "no variable-base product at all"): per = 5 MSM results per proof instead of 3 (A, B1, C, sum (s w) A, sum (r w) B1)
    // and 4-bit fixed-base tables of delta1, alpha1, beta1 and every folding constant of A and B1 (fb4), of delta2 (fb4g2)
    int per; const G1XYZZ* fb4; const G2XYZZ* fb4g2;
    uint8_t* out_xyzz;                                  // device: nproofs x 512 bytes, piA (G1XYZZ) | piB (G2XYZZ) | piC (G1XYZZ): the host makes them affine (prove_batch_finish)
};
constexpr int FB4_WIN = 64, FB4_ROW = 15;              // windows per base, multiples per window: T[base][w][d - 1] = d 16^w base
// zkc_blind_scalars: per proof q of a small pass, out[q][0][idx] = s w[idx] over the A list and out[q][1][idx] = r w[idx] over the B list (nullptr list: every wire)
struct BlindArgs { const uint32_t* w[2]; const uint32_t* mapA[2]; const uint32_t* mapB[2]; uint32_t nA[2], nB[2]; uint32_t* out[2]; const uint8_t* rs; uint32_t nv; };
// The (digit, point) entries of a pass are bucketed JOB BY JOB (an entry never leaves its job's region [ent_off, ent_off + count nw) of the
// value arrays), in two counting passes instead of a device-wide key sort: the job is implicit in the position, the key is never stored.
//   entry word after level 1 :  sign(1) | low bucket bits (lbits) | table row relative to the job's table (31 - lbits bits)
//   entry word after level 2 :  sign(1) | table row (31 bits)          -- what the accumulation reads; zero digits are never emitted
// Bucket ids are job-major; a pass may hold jobs of at most two window sizes, the larger ones first (decode() is then closed-form).
constexpr int MSM_TILE = 256, MSM_TILE_SCALARS = 1024;  // threads / scalars per workgroup of the counting and level-1 kernels
constexpr uint32_t MSM_MAX_HBITS = 10;                 // level-1 bins per job <= 1024 (LDS histogram)
struct MsmJobList {
    MsmJob job[MSM_MAX_JOBS]; int njobs; uint32_t total_buckets, total_entries, total_windows, total_bins, total_tiles, total_tilecnt;
    uint32_t hs, hb, nbig;                              // set by finish(): bucket counts of the small / big jobs, number of big jobs (they come first)
    uint32_t hbits_big, hbits_small;                    // level-1 bin bits of the two job classes when uniform inside each class (else 0xff: search)
    uint32_t vw_big = 1024;                             // virtual window of the c >= 16 jobs (H): 1024 for throughput, 256 for a pass of a few proofs -- a lane then walks 4 buckets in a row instead of 16 (zkc_msm_window29), for four times the windows in the per-job sum
    uint32_t vw_small = 256;                            // virtual window of the c < 16 jobs: 256 for throughput, 64 for the latency of a small pass (same box: 256 -> 2431 proofs/s, 4.9 ms ; 128 -> 2375, 4.0 ms ; 64 -> 2304, 3.9 ms single prove)
    void add(const uint32_t* scalars, const uint32_t* vmap, uint32_t count, uint32_t tbl_off, uint32_t tbl_count, int32_t pt_shift, int c) {
        MsmJob& j = job[njobs++];
        uint32_t vw = c >= 16 ? vw_big : vw_small;
        if (vw > (uint32_t)msm_half(c)) vw = (uint32_t)msm_half(c);          // a job never has less than one virtual window (ZKC_C_SECTIONS below 12 with the 2048-bucket windows of a full pass)
        while ((uint32_t)msm_half(c) / vw > max_vw_per_job) vw <<= 1;        // ... and never more than the per-job sum takes (a lone G2 job of a 14- or 15-bit key without the 8-bit table)
        j = MsmJob{scalars, vmap, count, tbl_off, tbl_count, pt_shift, (uint32_t)c, (uint32_t)msm_nw(c), vw, total_entries, total_windows, 0, 0, 0, 0, 0, 0};
        total_entries += count * (uint32_t)msm_nw(c); total_windows += (uint32_t)msm_half(c) / vw;
    }
    uint32_t max_vw_per_job = 256;                      // MSM_MAX_VW_G1 for a G1 list, MSM_MAX_VW_PER_JOB for a G2 list (clear)
    void clear(uint32_t vw_small_jobs = 256, uint32_t vw_big_jobs = 1024, uint32_t max_per_job = 256) { vw_small = vw_small_jobs; vw_big = vw_big_jobs; max_vw_per_job = max_per_job; njobs = 0; total_buckets = total_entries = total_windows = total_bins = total_tiles = total_tilecnt = 0; hs = hb = nbig = 0; }
    // false: more than two distinct window sizes, big jobs not in front, or a table too large for the row field of the level-1 entry word
    bool finish() {
        hs = 0xffffffffu; hb = 0;
        for (int j = 0; j < njobs; j++) { const uint32_t h = 1u << (job[j].c - 1); hs = h < hs ? h : hs; hb = h > hb ? h : hb; }
        nbig = 0; total_buckets = total_bins = total_tiles = total_tilecnt = 0; hbits_big = hbits_small = 0;
        for (int j = 0; j < njobs; j++) {
            MsmJob& q = job[j]; const uint32_t h = 1u << (q.c - 1);
            if (h != hs && h != hb) return false;
            if (h == hb) { if ((uint32_t)j != nbig) return false; nbig++; }
            uint32_t hbits = q.c - 1 < 8 ? q.c - 1 : 8, lbits = q.c - 1 - hbits;
            const uint64_t rows = (uint64_t)q.nw * q.tbl_count;
            while (lbits > 8 || (lbits > 0 && rows >= (1ull << (31 - lbits)))) { lbits--; hbits++; }      // level 2 handles <= 256 buckets per bin
            if (hbits > MSM_MAX_HBITS || rows >= (1ull << (31 - lbits))) return false;
            q.hbits = hbits; q.lbits = lbits;
            { uint32_t& cls = h == hb ? hbits_big : hbits_small; cls = (cls == 0 || cls == hbits) ? hbits : 0xffu; }
            q.bucket0 = total_buckets; total_buckets += h;
            q.bin0 = total_bins; total_bins += 1u << hbits;
            const uint32_t nt = (q.count + MSM_TILE_SCALARS - 1) / MSM_TILE_SCALARS;
            q.tile0 = total_tiles; total_tiles += nt; q.cnt0 = total_tilecnt; total_tilecnt += nt << hbits;
        }
        return true;
    }
    ZKC_HD uint32_t id_of(uint32_t d, uint32_t j) const { return job[j].bucket0 + d; }
    // job owning level-1 bin `bin` of the pass: closed form when the two job classes have uniform bin counts (the prover's passes), else a search
    ZKC_HD uint32_t job_of_bin(uint32_t bin) const {
        if (hbits_big != 0xffu && hbits_small != 0xffu) {
            const uint32_t lim = nbig << hbits_big;
            return bin < lim ? bin >> hbits_big : nbig + ((bin - lim) >> hbits_small);
        }
        uint32_t lo = 0, hi = (uint32_t)njobs - 1;
        while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (job[mid].bin0 <= bin) lo = mid; else hi = mid - 1; }
        return lo;
    }
    ZKC_HD void decode(uint32_t id, uint32_t& d, uint32_t& j) const {
        const uint32_t lim = hb * nbig;
        if (id < lim) { j = id / hb; d = id - j * hb; }
        else { const uint32_t t = id - lim, q = t / hs; j = nbig + q; d = t - q * hs; }
    }
};
struct MsmWindow { uint32_t bucket0, out, per; };   // one wave of zkc_msm_window: the 64 per consecutive buckets from bucket0 -> wres[2*out] (weighted), wres[2*out+1] (plain sum)

// Work space of one pipeline pass (sized for max_jobs jobs and max_entries (scalar, window) pairs)
struct MsmWork {
    uint32_t *vals = nullptr, *vals2 = nullptr;       // max_entries each: level-1 output / bucketed entries
    uint32_t *hist = nullptr, *bin_start = nullptr;      // per level-1 bin of the pass (max_bins): entries, first position
    uint32_t *tilecnt = nullptr; size_t max_tilecnt = 0; // per (job, bin, tile): entries, then (scanned) the start of the tile's run inside the bin
    uint32_t *off = nullptr, *bcnt = nullptr;          // per bucket: first entry in vals2, number of entries
    uint32_t *segcnt = nullptr, *segoff = nullptr, *seg2bucket = nullptr, *heavy = nullptr;
    uint32_t *seglen = nullptr, *perm = nullptr;       // per segment: entries; segment ids by decreasing length
    uint32_t *scan_blk = nullptr, *lencnt = nullptr; size_t max_lencnt = 0;  // block sums of the device-wide scan; per (length key, workgroup) segment counts
    void *partial = nullptr;        // XYZZ per segment
    void *wres = nullptr;           // XYZZ per (job, window)
    void *results = nullptr;        // XYZZ per job, two slots of max_jobs (device) ; h_results pinned host mirror of slot 0
    void *h_results = nullptr;
    MsmJobList* d_jobs = nullptr;   // device copy of the pass' job list (too large for kernel arguments)
    MsmWindow* d_windows = nullptr; size_t max_buckets = 0, max_bins = 0, max_windows = 0;
    // pinned host staging of (job list, window list), two slots: a pageable source made hipMemcpyAsync hold the enqueueing thread until
    // the stream had drained, which kept the G2 stream idle for 8 ms of every pass.  h_ev[s] = the copy out of slot s has executed.
    uint16_t* d_tilejob = nullptr; uint16_t* h_tilejob[2] = {nullptr, nullptr}; size_t max_tiles = 0;      // tile of the pass -> job
    MsmJobList* h_jobs[2] = {nullptr, nullptr}; MsmWindow* h_windows[2] = {nullptr, nullptr}; hipEvent_t h_ev[2] = {nullptr, nullptr}; int h_next = 0;
    size_t max_entries = 0, max_segments = 0; int max_jobs = 0; size_t xyzz_size = 0;
};
}  // namespace zkc

// One pipeline lane: its own three streams and every per-pass buffer.  Two lanes take alternate passes so that the
// latency-bound tail of a pass (window reduce, final sums, blinding) overlaps the throughput-bound head of the next one.
struct zkc_lane {
    hipStream_t st = nullptr, st2 = nullptr, fin = nullptr;               // buildABC/NTT/G1 MSM ; G2 MSM ; blinding + D2H.  [r5] Borrowed from the context (zkc_lane_streams): every key of a context shares them
    hipStream_t red = nullptr; hipEvent_t ev_red = nullptr;               // [r4] the G1 bucket reduction of a full pass runs here, beside the next pass' transforms; ev_red: the reduction of the lane's latest pass is through (wherever it ran)
    zkc::Fr *d_abc = nullptr, *d_t = nullptr; uint32_t* d_p = nullptr;   // [inflight][3n] x2 (d_t = NTT scratch), [inflight][n x 8]
    void* d_fin = nullptr;                                                // blinding scratch, finalize_scratch_bytes(inflight)
    uint32_t* d_bs = nullptr;                                             // [2 proofs][2 sections][nVars x 8]: the blinded scalars of a small pass (zkc_blind_scalars)
    zkc::MsmWork w1, w2;                                                  // G1 and G2 pipelines
    size_t cap_abc = 0, cap_p = 0, cap_fin = 0, cap_bs = 0, bytes = 0; bool made = false;      // capacities (elements of d_abc / d_t, words of d_p, bytes of d_fin, words of d_bs), HBM held (hipMemGetInfo delta), events and streams created
    hipEvent_t ev_msm = nullptr, ev_msm2 = nullptr, ev_sorted = nullptr, ev_ntt = nullptr, ev_mv = nullptr, ev_acc = nullptr, ev_fin[2] = {nullptr, nullptr}; int npass = 0;      // ev_acc: the G1 accumulation of the lane's latest pass is through      // ev_ntt: buildABC/NTT/joinABC of the pass are through      // ev_fin[slot]: blinding of the pass that used result slot `slot`
};

struct zkc_zkey {
    zkc_ctx* ctx = nullptr;
    uint32_t nVars = 0, nPub = 0, n = 0, logn = 0, nCoeffs = 0;
    int nLevels = -1;                              // >= 0 when the key's shape matches ZkFranchiseProofCircuit(nLevels): enables folding
    zkc::G1Affine alpha1, beta1, delta1;           // host copies, Montgomery
    zkc::G2Affine beta2, gamma2, delta2;
    std::vector<zkc::G1Affine> ic;
    // device: section 4 in jagged-diagonal order (rows [0,n) = A, [n,2n) = B, sorted by length), values as stored (val * R^2)
    uint32_t *d_perm = nullptr, *d_rowlen = nullptr, *d_jdptr = nullptr, *d_col = nullptr; zkc::Fr* d_val = nullptr; uint32_t nlong = 0;
    uint32_t n_unit_coeffs = 0;                    // [r4] coefficients that are +1 / -1: marked in the two top bits of d_col (bit 31 unit, bit 30 negative)
    zkc::Fr *d_tw_fwd = nullptr, *d_tw_inv = nullptr, *d_coset = nullptr;   // w^j, w^-j (j < n/2), g^i / n
    zkc::Fr *d_coset_br = nullptr;                                          // g^i / n at bit-reversed positions (ntt_pair_run)
    uint32_t *d_tw_fwd29 = nullptr, *d_tw_inv29 = nullptr;                  // the twiddles in radix 2^29 (what the NTT passes read)
    // pre-shifted base tables, one allocation per group: G1 = [A | B1 | C | H], G2 = [B2]; T[w][i] = 2^(c*w) * P_i
    zkc::G1Affine* d_g1 = nullptr; zkc::G2Affine* d_g2 = nullptr;
    uint32_t* d_g2_29 = nullptr;                                            // the G2 table again in radix 2^29 (60 words per point: x, y, -y), read by the accumulation
    uint32_t* d_g2_29_lone = nullptr;                                       // the same bases pre-shifted for MSM_C_G2_LONE (32 windows), radix 2^29 only; nullptr: not built (ZKC_G2_LONE_TABLE=0)
    uint32_t offA = 0, offB1 = 0, offC = 0, offH = 0;                       // table offsets inside d_g1 (points)
    // [r4] a census key of 2^16 wires and more keeps its witness sections a second time, pre-shifted for c_deep = 17-bit windows: a pass whose voters sit deep in the trees (or
    // whose witnesses do not fold at all) has 50-80 k wires per section instead of 8-11 k, and 15 additions per scalar into 65536 buckets then beat 22 into 2048.  0: not built
    int c_deep = 0; uint32_t offA_deep = 0, offB1_deep = 0, offC_deep = 0; uint32_t* d_g2_29_deep = nullptr;
    // per-proof work buffers
    int c_h = zkc::MSM_C_BIG;                                                    // [r4] window bits of the H section of THIS key (17 for a census key, else by the domain size)
    int c_sec = zkc::MSM_C_SMALL;                                                // [r4] window bits of the witness sections A, B1, C, B2 of THIS key (zkc_zkey_load: 12, or 17 for sections of 2^16 wires and more)
    int max_inflight = 0;                                                   // proofs per pipeline pass (upper limit)
    uint8_t sha256[32] = {0};                                               // of the whole .zkey image (taken once, at load)
    size_t bytes_tables = 0;                                                // HBM held by the key's constant tables (hipMemGetInfo delta around the load; the lanes' work space is the context's)
    uint8_t fingerprint[32] = {0};                                          // parse::zkey_fingerprint of the image: the per-call identity of the resident-key caches
    int nlanes = 1; bool serial_streams = false;          // lanes of the context this key's passes may use (zkc_ctx::lanes); serial_streams: every stage on ctx->stream instead of the lane's streams (ZKC_SERIAL_STREAMS, measurement only)
    zkc::G1Affine *d_tblDelta1 = nullptr, *d_tblAlpha1 = nullptr, *d_tblBeta1 = nullptr; zkc::G2Affine* d_tblDelta2 = nullptr;
    uint32_t *d_depths = nullptr, *h_depths = nullptr; size_t depths_cap = 0;                     // zkc_input_depths: device [2 B], pinned [2 B] (read back inside begin, under the context lock)
    zkc::G1XYZZ* d_fb4 = nullptr; zkc::G2XYZZ* d_fb4g2 = nullptr; int fb4_bases = 0;   // 4-bit fixed-base tables of the small-pass blinding (FinalizeArgs::fb4)
    // per CALL state, two slots: a call is begin (everything enqueued, returns) + finish (wait, copy out), and the proving service lets the begin of the next
    // call run while the previous one drains (its witness kernels beside the other call's MSMs, its transforms beside the other's bucket reduction and blinding)
    struct CallSlot {
        uint8_t *d_rs = nullptr, *d_proofs = nullptr; size_t cap = 0;       // [B][64], [B][256]
        uint8_t* h_out = nullptr;                                           // pinned: [B][256] proofs then [B][nPub][32] public signals (async D2H target)
        uint8_t* h_rs = nullptr;                                            // pinned copy of the caller's (r, s): begin returns before the upload has executed
        uint8_t *d_xyzz = nullptr, *h_xyzz = nullptr; std::vector<uint8_t> as_xyzz;     // [B][512]: a small pass (one or two proofs) hands its three points over as XYZZ; as_xyzz[q]: proof q is to be made affine by finish
        hipEvent_t ev_done[zkc::MAX_LANES] = {};                            // per lane: recorded on its blinding stream behind the call's last copy
        std::vector<hipEvent_t> ev_chunk;                                   // ev_chunk[p]: witness (if made here) and fold flags of the call's pass p are ready
        uint32_t lanes_used = 0;                                            // bit l: this call enqueued a pass on lane l (finish waits for those lanes only)
        int B = 0; bool pending = false;
        uint32_t *d_flags = nullptr, *h_flags = nullptr; size_t flags_cap = 0;  // fold check of this call's witnesses: [B][2][n] (device, pinned)
        int32_t* d_status3 = nullptr; size_t status3_cap = 0;                   // the witness kernels' three status words per voter
        // [r3] a call of <= 2 voters laid out from zkc_input_depths (prove_batch_begin): voters, (census, sik) depths; pinned: their status [B], then the fold flags of the
        // finished witness [B][2][n].  [r5] any one-pass call that brings its inputs (the proving service's batches): begin no longer waits for the witness kernel
        int early_n = 0; std::vector<uint8_t> early_depth; uint32_t* h_early = nullptr; size_t early_cap = 0;
        const void* arg_wtns = nullptr; uint32_t arg_nw = 0; bool arg_publics = false, arg_inputs = false; int arg_lane0 = -1; hipEvent_t arg_wait = nullptr; std::vector<uint8_t> h_rs_copy;      // what begin was called with (finish begins a witness-given call again when its early layout is refused)
    } call[zkc::CALL_SLOTS];
    // constant folding of the voter-independent witness part (SURVEY.md hard part 4)
    struct Fold {
        std::vector<zkc::G1XYZZ> baseA, baseB1, baseC; std::vector<zkc::G2XYZZ> baseB2;        // [1]
        std::vector<zkc::G1XYZZ> sufA[2], sufB1[2], sufC[2]; std::vector<zkc::G2XYZZ> sufB2[2];   // [tree][D] = sum over levels >= D
        struct VMap { uint32_t* d = nullptr; uint32_t offA = 0, nA = 0, offB = 0, nB = 0, offC = 0, nC = 0; };   // three lists in one allocation
        std::map<std::pair<int, int>, VMap> vmaps;                                              // (Dc, Ds) -> surviving wires per section
        std::vector<uint8_t> infA, infB, infC;                                                  // base point is the point at infinity (zero polynomial)
        zkc::G1XYZZ *d_foldA = nullptr, *d_foldB1 = nullptr, *d_foldC = nullptr; zkc::G2XYZZ* d_foldB2 = nullptr;    // [1 + 2 n] each: base, suf[0][], suf[1][]
        bool ready = false;
    } fold;
};

namespace zkc {
int ntt_run(zkc_ctx* ctx, hipStream_t st, const Fr* src, Fr* dst, const uint32_t* tw29, const Fr* scale, int logn, int nvec);
int ntt_make_tw29(zkc_ctx* ctx, const Fr* d_tw, uint32_t count, uint32_t** out);
// [r2] the prover's pair iNTT -> x scale -> NTT in three HBM round trips without bit reversal (zkc_ntt.hip); scale_br = scale table in bit-reversed order
int ntt_pair_run(zkc_ctx* ctx, hipStream_t st, Fr* data, const uint32_t* tw_inv29, const uint32_t* tw_fwd29, const Fr* scale_br, int logn, int nvec);
int ntt_bitrev_table(zkc_ctx* ctx, const Fr* d_src, Fr** out, int logn);      // twiddles in the 12-word radix-2^29 form ntt_run reads
int msm_work_alloc(zkc_ctx* ctx, MsmWork& w, size_t max_entries, size_t max_buckets, int max_jobs, bool g2);
void msm_work_free(MsmWork& w);
// runs all jobs of `jl` through one pipeline pass; results (XYZZ per job) go to device slot `slot` (0/1) of w.results and, when
// to_host is set, to w.h_results (valid after the caller syncs the stream)
// ev_sorted (optional): recorded on st once the digit/sort/segment kernels are through, i.e. right before the long accumulation kernel
// st_red (optional, needs ev_acc): the bucket reduction (merge, windows, per-job sums) runs on that stream behind ev_acc instead of on st; ev_red (optional) is recorded behind it
int msm_pass_g1(zkc_zkey* zk, MsmWork& w, const MsmJobList& jl, int slot, bool to_host, hipStream_t st, hipEvent_t ev_sorted = nullptr, hipEvent_t ev_acc = nullptr,
                hipStream_t st_red = nullptr, hipEvent_t ev_red = nullptr);
int msm_pass_g2(zkc_zkey* zk, MsmWork& w, const MsmJobList& jl, int slot, bool to_host, hipStream_t st, hipEvent_t wait_before_acc = nullptr);
// zkc_msm_sort.hip -- K4: scalars -> signed digits -> entries bucketed per job (vals2, off, bcnt), then the segment lists of the accumulation
// (segcnt, segoff, seg2bucket, seglen, perm, heavy).  `jl` is the finished host copy of what w.d_jobs already holds on the device.
int msm_bucket_entries(zkc_ctx* ctx, MsmWork& w, const MsmJobList& jl, hipStream_t st, unsigned long long* d_entry_counter = nullptr);
int msm_build_segments(zkc_ctx* ctx, MsmWork& w, const MsmJobList& jl, uint32_t seg, size_t seg_bound, hipStream_t st);
int finalize_launch(zkc_ctx* ctx, hipStream_t st, const FinalizeArgs& a, int nproofs);
int finalize_tree_g2_launch(zkc_ctx* ctx, hipStream_t st, const FinalizeArgs& a, int nproofs);           // piB of a small pass, on the G2 stream
int fb4_build(zkc_ctx* ctx, hipStream_t st, const G1XYZZ* d_bases, int nbases, G1XYZZ* d_out, const G2XYZZ& delta2, const G2XYZZ& beta2, const G2XYZZ& beta2_base, G2XYZZ* d_out2);   // d_out2: 64 x 15 + 2 points
int blind_scalars_launch(zkc_ctx* ctx, hipStream_t st, const BlindArgs& a, int nproofs);
// a batch call in two halves (zkc_prove.hip; the public zkc_[full]prove_batch_dev are begin + finish on slot 0): begin validates, enqueues every pass and returns
// without waiting for the GPU; finish waits for that call and copies proofs / public signals out.  cs = call slot 0 / 1; a slot must be finished before it is
// begun again; d_wtns (and d_inputs, d_status) stay the caller's until finish returns.  d_inputs == nullptr: the witnesses are given.
// lane0 >= 0: every pass of the call runs on that lane (the service: one lane per worker); -1: the passes rotate over the key's lanes.
// wait_first (optional): an event the call's first kernels wait for (the uploads of its inputs / witnesses on the caller's stream)
// host_depths (optional, 2 per voter: census, sik): 1 + index of the voter's last non-zero sibling per tree, read by the caller on the host; a one-pass call is then laid out
// from them without any GPU round trip inside begin (255 in any entry: unknown, the call takes the usual path); no_early: never lay the pass out ahead of the fold flags
int prove_batch_begin(zkc_zkey* zk, int cs, const void* d_wtns, uint32_t nWitness, int B, const uint8_t* rs, bool want_publics, const void* d_inputs, int32_t* d_status,
                      int lane0 = -1, hipEvent_t wait_first = nullptr, const uint8_t* host_depths = nullptr, bool no_early = false);
int prove_batch_finish(zkc_zkey* zk, int cs, uint8_t* proofs, uint8_t* publics);
int prove_reserve(zkc_zkey* zk, int inflight);              // grow the key's work space to `inflight` proofs per pass now (clamped to the key's own limit)
// zkc_zkey_load with the lane count and pass size given instead of read from $ZKC_LANES / $ZKC_INFLIGHT (0: take the environment / defaults)
int zkey_load_opts(zkc_ctx* ctx, const void* zkey_bytes, size_t len, int nlanes, int max_inflight, zkc_zkey** out);
size_t zkey_device_bytes(const zkc_zkey* zk, size_t* tables, size_t* work);      // HBM now: the key's constant tables / the per-pass work space of its context's lanes (shared by every key of the context)
size_t finalize_scratch_bytes(int nproofs);
int msm_precompute_g1(zkc_ctx* ctx, uint32_t count, G1Affine* d_table, int c);   // d_table[0..count) = base on entry
int msm_precompute_g2(zkc_ctx* ctx, uint32_t count, G2Affine* d_table, int c);
int msm_g2_table29(zkc_ctx* ctx, const G2Affine* d_table, uint32_t* d_out, size_t count);       // d_out: 60 words per point
// out[i] = scalar[wires[i]] * P[wires[i] - pt_shift] (window-0 table), then per-group sums: gsum[g] = sum out[gstart[g]..gstart[g+1])
int fold_group_sums_g1(zkc_ctx* ctx, const G1Affine* tbl, const uint32_t* d_scalars, const uint32_t* d_wires, uint32_t nw, int32_t pt_shift,
                       const uint32_t* d_gstart, uint32_t ngroups, G1XYZZ* h_out);
int fold_group_sums_g1_ws(zkc_ctx* ctx, const G1Affine* tbl, const uint32_t* d_scalars, const uint32_t* d_wires, uint32_t nw, const uint32_t* d_gstart, uint32_t ngroups,
                          G1XYZZ* d_tmp /* nw */, G1XYZZ* d_out /* ngroups */, G1XYZZ* h_out);
int fold_group_sums_g2(zkc_ctx* ctx, const G2Affine* tbl, const uint32_t* d_scalars, const uint32_t* d_wires, uint32_t nw, int32_t pt_shift,
                       const uint32_t* d_gstart, uint32_t ngroups, G2XYZZ* h_out);
}
