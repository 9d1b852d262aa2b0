// zkc_msm_sort.hip -- K4 of the MSM pipeline: scalars -> signed c-bit digits -> (digit, point) entries grouped by bucket, per job.
//
// Replaces what round 1 did with a device-wide rocPRIM radix sort of 250 M (16-bit key, 32-bit value) pairs per pass.  The entries of a
// job never leave the job's own region of the value array, the job and the key are implicit in the position, and only the 4-byte
// value word is ever written:
//   zkc_msm_count    one workgroup per tile of 1024 scalars of a job: digits -> LDS histogram over the job's 2^hbits level-1 bins (high bits of
//                    the bucket index) -> one counter per (bin, tile)
//   zkc_msm_binscan  one workgroup per job: exclusive scan of those counters in (bin, tile) order -> where every tile's run starts in every
//                    bin (absolute positions inside the job's region), bin boundaries
//   zkc_msm_split    same tiling as the count: ranks its entries with LDS atomics, window by window, and writes
//                    sign | low bucket bits | table row  straight into its runs (no global atomics: the placement is deterministic per tile)
//   zkc_msm_bucket   one workgroup per level-1 bin (7.7 k entries for an H job, L2 resident): LDS histogram over the 2^lbits buckets of the
//                    bin, scan, bucket boundaries (off, bcnt) and the final scatter of  sign | table row
// Zero digits are never emitted.  Inside a bucket the entries are ordered by (tile, window), arbitrarily below that: group addition is
// commutative and exact, so the proof bytes do not depend on the order (tests/test_gpu_prover.py compares them with the CPU oracle).
//   zkc_msm_segcount / zkc_scan_* / zkc_msm_seg2bucket / zkc_msm_len*   buckets -> segments of <= seg entries, an exclusive scan over the
//                    buckets of the pass, and the permutation that lists the segments longest first (counting sort over <= 129 lengths).
#include "zkc_prover.h"
#include <algorithm>

namespace zkc {

// The tile of a workgroup: 1024 consecutive scalars of one job, four per thread, walked WINDOW BY WINDOW with a barrier in between, so that the
// entries a tile contributes to a bin are ordered by window (then arbitrarily inside the 1024 scalars): together with the tile-ordered
// placement below, the entries of a bucket end up ordered by (tile, window) -- the accumulation then streams through compact regions of the
// pre-shifted tables (64 KB per (tile, window)) and the jobs of different proofs, which run side by side, touch the same table rows at about
// the same time.  (An order left to atomics alone made the G2 accumulation 4x slower: every gather a random row of a 437 MB table.)
struct TileState { uint32_t s[MSM_TILE_SCALARS / MSM_TILE][8]; uint32_t pt[MSM_TILE_SCALARS / MSM_TILE]; uint32_t carry[MSM_TILE_SCALARS / MSM_TILE]; bool live[MSM_TILE_SCALARS / MSM_TILE]; };
__device__ __forceinline__ void msm_tile_load(TileState& t, const MsmJob& job, uint32_t i0) {
#pragma unroll
    for (int k = 0; k < MSM_TILE_SCALARS / MSM_TILE; k++) {
        const uint32_t i = i0 + k * MSM_TILE;
        t.live[k] = i < job.count; t.pt[k] = 0; t.carry[k] = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) t.s[k][q] = 0;
        if (t.live[k]) {
            const uint32_t wire = job.vmap ? job.vmap[i] : i;
            const uint4* sp = reinterpret_cast<const uint4*>(job.scalars + 8 * (size_t)wire); const uint4 a = sp[0], b = sp[1];
            t.s[k][0] = a.x; t.s[k][1] = a.y; t.s[k][2] = a.z; t.s[k][3] = a.w; t.s[k][4] = b.x; t.s[k][5] = b.y; t.s[k][6] = b.z; t.s[k][7] = b.w;
            t.pt[k] = (uint32_t)((int32_t)wire - job.pt_shift);
        }
    }
}
// signed digit of window w of scalar k of the tile (call with w = 0, 1, ... in order: the carry travels in t): returns |d| (0: nothing to add)
__device__ __forceinline__ uint32_t msm_tile_digit(TileState& t, int k, uint32_t w, uint32_t c, uint32_t& neg) {
    const uint32_t half = 1u << (c - 1), mask = (1u << c) - 1;
    const uint32_t bit = w * c, li = bit >> 5, sh = bit & 31;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) { lo = ((uint32_t)q == li) ? t.s[k][q] : lo; hi = ((uint32_t)q == li + 1) ? t.s[k][q] : hi; }
    const uint64_t two = (uint64_t)lo | ((uint64_t)hi << 32);
    uint32_t d = (uint32_t)((two >> sh) & mask) + t.carry[k];
    neg = 0;
    if (d > half) { d = (1u << c) - d; neg = 1; t.carry[k] = 1; } else t.carry[k] = 0;
    return d;
}

// tilecnt[job.cnt0 + bin * ntiles + tile] = entries the tile has for the bin (no atomics on global memory: placement is deterministic)
__global__ void __launch_bounds__(MSM_TILE)
zkc_msm_count(const MsmJobList* __restrict__ jlp, const uint16_t* __restrict__ tilejob, uint32_t* __restrict__ tilecnt) {
    __shared__ uint32_t cnt[1u << MSM_MAX_HBITS];
    const MsmJobList& jl = *jlp;
    const uint32_t j = tilejob[blockIdx.x];
    const MsmJob job = jl.job[j];
    const uint32_t nbins = 1u << job.hbits, tile = blockIdx.x - job.tile0, ntiles = (job.count + MSM_TILE_SCALARS - 1) / MSM_TILE_SCALARS;
    for (uint32_t b = threadIdx.x; b < nbins; b += MSM_TILE) cnt[b] = 0;
    __syncthreads();
    TileState t; msm_tile_load(t, job, tile * MSM_TILE_SCALARS + threadIdx.x);
    for (uint32_t w = 0; w < job.nw; w++) {
#pragma unroll
        for (int k = 0; k < MSM_TILE_SCALARS / MSM_TILE; k++) { uint32_t neg; const uint32_t d = msm_tile_digit(t, k, w, job.c, neg); if (t.live[k] && d) atomicAdd(&cnt[(d - 1) >> job.lbits], 1u); }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nbins; b += MSM_TILE) tilecnt[job.cnt0 + b * ntiles + tile] = cnt[b];
}
// one workgroup per job: exclusive scan (in place) of its nbins x ntiles tile counts in (bin, tile) order, offset by ent_off -> where each tile's
// run inside each bin starts; hist[bin] = entries of the bin, bin_start[bin] = its first position
__global__ void __launch_bounds__(256)
zkc_msm_binscan(const MsmJobList* __restrict__ jlp, uint32_t* __restrict__ tilecnt, uint32_t* __restrict__ hist, uint32_t* __restrict__ bin_start,
                unsigned long long* __restrict__ entry_counter) {
    __shared__ uint32_t part[256];
    __shared__ uint32_t carry_sh;
    const MsmJob& job = jlp->job[blockIdx.x];
    const uint32_t nbins = 1u << job.hbits, ntiles = (job.count + MSM_TILE_SCALARS - 1) / MSM_TILE_SCALARS, total = nbins * ntiles;
    uint32_t* c = tilecnt + job.cnt0;
    if (threadIdx.x == 0) carry_sh = job.ent_off;
    __syncthreads();
    for (uint32_t base = 0; base < total; base += 1024) {
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint32_t i = base + threadIdx.x * 4 + k; v[k] = i < total ? c[i] : 0; sum += v[k]; }
        part[threadIdx.x] = sum; __syncthreads();
        for (int o = 1; o < 256; o <<= 1) { uint32_t a = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0; __syncthreads(); part[threadIdx.x] += a; __syncthreads(); }
        uint32_t run = carry_sh + part[threadIdx.x] - sum;
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint32_t i = base + threadIdx.x * 4 + k; if (i < total) c[i] = run; run += v[k]; }
        __syncthreads();
        if (threadIdx.x == 255) carry_sh = run;
        __syncthreads();
    }
    // bin boundaries from the scanned counts: bin b starts where its tile 0 starts
    const uint32_t end = carry_sh;
    if (entry_counter && threadIdx.x == 0) atomicAdd(entry_counter, (unsigned long long)(end - job.ent_off));      // non-zero digits = group additions this job needs (measurement only)
    for (uint32_t b = threadIdx.x; b < nbins; b += 256) {
        const uint32_t st = c[b * ntiles], en = b + 1 < nbins ? c[(b + 1) * ntiles] : end;
        bin_start[job.bin0 + b] = st; hist[job.bin0 + b] = en - st;
    }
}
// (Tried in round 2: grouping the tile's entries by bin in 45 KB of LDS and storing every (tile, bin) run with one contiguous store per wave.
// The stores became cheap, but two workgroups per CU and a serial copy-out loop per wave made the kernel slower: 2.4 ms against 1.6 beside the G2
// accumulation.  The direct form stays.)
__global__ void __launch_bounds__(MSM_TILE)
zkc_msm_split(const MsmJobList* __restrict__ jlp, const uint16_t* __restrict__ tilejob, const uint32_t* __restrict__ tilecnt, uint32_t* __restrict__ vals) {
    __shared__ uint32_t cur[1u << MSM_MAX_HBITS];      // next free position of this tile's run inside each bin
    const MsmJobList& jl = *jlp;
    const uint32_t j = tilejob[blockIdx.x];
    const MsmJob job = jl.job[j];
    const uint32_t nbins = 1u << job.hbits, tile = blockIdx.x - job.tile0, ntiles = (job.count + MSM_TILE_SCALARS - 1) / MSM_TILE_SCALARS;
    for (uint32_t b = threadIdx.x; b < nbins; b += MSM_TILE) cur[b] = tilecnt[job.cnt0 + b * ntiles + tile];
    TileState t; msm_tile_load(t, job, tile * MSM_TILE_SCALARS + threadIdx.x);         // second read of the tile: L2 hit
    const uint32_t lowmask = (1u << job.lbits) - 1, lowshift = 31 - job.lbits;
    for (uint32_t w = 0; w < job.nw; w++) {
        __syncthreads();                                // window by window: see the comment above TileState
#pragma unroll
        for (int k = 0; k < MSM_TILE_SCALARS / MSM_TILE; k++) {
            uint32_t neg; const uint32_t d = msm_tile_digit(t, k, w, job.c, neg);
            if (t.live[k] && d) {
                const uint32_t idx = d - 1, pos = atomicAdd(&cur[idx >> job.lbits], 1u);
                vals[pos] = (neg << 31) | ((idx & lowmask) << lowshift) | (t.pt[k] + w * job.tbl_count);
            }
        }
    }
}
// one workgroup per level-1 bin: its entries (contiguous in `vals`) go to `vals2` grouped by bucket; off / bcnt for the bin's 2^lbits buckets.
// A bin of up to 8192 entries (every bin of an H job: 7.7 k) is held in registers, 32 loads in flight per lane, and read once; larger bins
// (repeated witness values; every bin of a job of 2^19 scalars and more) are counted first and then staged through LDS in chunks of 4096.
// CHUNKED = false: the kernel of rounds 2-3 (52 VGPRs: two of its waves fit beside a wave of the G2 accumulation, which it is scheduled beside in a census pass; bins above 8192
// entries -- a few, from repeated witness values -- take the slow two-pass path).  CHUNKED = true: bins above 8192 entries are the rule (jobs of 2^19 scalars and more) and go
// through LDS in chunks (108 VGPRs).  msm_bucket_entries picks by the expected bin size of the pass' largest job.
template <bool CHUNKED>
__global__ void __launch_bounds__(256)
zkc_msm_bucket(const MsmJobList* __restrict__ jlp, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ bin_start, const uint32_t* __restrict__ vals,
               uint32_t* __restrict__ vals2, uint32_t* __restrict__ off, uint32_t* __restrict__ bcnt) {
    __shared__ uint32_t cnt[256], pos[256];
    __shared__ uint32_t stage[256 * 32];              // the bin, grouped by bucket, before it leaves in whole cache lines (register path)
    const MsmJobList& jl = *jlp;
    const uint32_t bin = blockIdx.x, j = jl.job_of_bin(bin);
    const MsmJob& job = jl.job[j];
    const uint32_t lbits = job.lbits, nb2 = 1u << lbits, lowshift = 31 - lbits, lowmask = nb2 - 1, rowmask = (1u << lowshift) - 1;
    const uint32_t start = bin_start[bin], n = hist[bin];
    const uint32_t bucket_first = job.bucket0 + ((bin - job.bin0) << lbits);
    constexpr int BIG = 32, SMALL = 8;
    const bool in_regs = n <= 256u * BIG;
    uint32_t e[BIG];
    cnt[threadIdx.x] = 0; __syncthreads();
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < BIG; k++) { const uint32_t i = k * 256 + threadIdx.x; e[k] = i < n ? vals[start + i] : 0u; }
#pragma unroll
        for (int k = 0; k < BIG; k++) if ((uint32_t)k * 256 + threadIdx.x < n) atomicAdd(&cnt[(e[k] >> lowshift) & lowmask], 1u);
    } else {
        for (uint32_t i0 = 0; i0 < n; i0 += 256 * SMALL) {
            uint32_t v[SMALL];
#pragma unroll
            for (int k = 0; k < SMALL; k++) { const uint32_t i = i0 + k * 256 + threadIdx.x; v[k] = i < n ? vals[start + i] : 0u; }
#pragma unroll
            for (int k = 0; k < SMALL; k++) if (i0 + k * 256 + threadIdx.x < n) atomicAdd(&cnt[(v[k] >> lowshift) & lowmask], 1u);
        }
    }
    __syncthreads();
    const uint32_t mine = cnt[threadIdx.x];
    pos[threadIdx.x] = mine; __syncthreads();
    for (int o = 1; o < 256; o <<= 1) { uint32_t a = (int)threadIdx.x >= o ? pos[threadIdx.x - o] : 0; __syncthreads(); pos[threadIdx.x] += a; __syncthreads(); }
    const uint32_t excl = pos[threadIdx.x] - mine;
    __syncthreads();
    if (threadIdx.x < nb2) { off[bucket_first + threadIdx.x] = start + excl; bcnt[bucket_first + threadIdx.x] = mine; }
    pos[threadIdx.x] = excl; __syncthreads();
    constexpr uint32_t excl_base = 0;
    // scatter, 256 entries at a time with a barrier in between, so that a bucket keeps the (tile, window) order of the bin
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < BIG; k++) {
            if ((uint32_t)k * 256 < n) {                     // uniform
                if ((uint32_t)k * 256 + threadIdx.x < n) { const uint32_t p = atomicAdd(&pos[(e[k] >> lowshift) & lowmask], 1u) - excl_base; stage[p] = (e[k] & 0x80000000u) | (e[k] & rowmask); }
                __syncthreads();
            }
        }
        // scattered 4-byte stores into a 31 KB window left the L2 as partial lines (2.4 ms per pass); from LDS the bin goes out coalesced
        for (uint32_t i = threadIdx.x; i < n; i += 256) vals2[start + i] = stage[i];
    } else if constexpr (!CHUNKED) {
        for (uint32_t i0 = 0; i0 < n; i0 += 256 * SMALL) {
            uint32_t v[SMALL];
#pragma unroll
            for (int k = 0; k < SMALL; k++) { const uint32_t i = i0 + k * 256 + threadIdx.x; v[k] = i < n ? vals[start + i] : 0u; }
#pragma unroll
            for (int k = 0; k < SMALL; k++) {
                if (i0 + k * 256 + threadIdx.x < n) { const uint32_t p = atomicAdd(&pos[(v[k] >> lowshift) & lowmask], 1u); vals2[start + p] = (v[k] & 0x80000000u) | (v[k] & rowmask); }
                __syncthreads();
            }
        }
    } else {
        // [r4] a bin of more than 8192 entries (repeated witness values; every bin of a job of 2^19 scalars and more: 30 k entries at 2^20) goes through the same LDS staging in
        // CHUNKS of 4096: a chunk is ranked by bucket inside LDS and leaves as one contiguous run per bucket (~32 entries = 128 B each at 128 buckets per bin) behind that bucket's
        // running position, instead of one scattered 4-byte store per entry (which left the L2 as partial lines: 3.7 ms per launch at a 2^20 domain, the largest bucketing
        // kernel there).  Order inside a bucket: chunk, then 256-entry group -- the (tile, window) order of the bin, as before.  pos[] = running position of every bucket.
        uint32_t* ccnt = cnt;                              // per-chunk bucket counts (the bin-wide counts have been consumed: off / bcnt are written)
        __shared__ uint32_t cbase[256];                    // exclusive scan of ccnt: where a bucket's run starts inside the staged chunk
        __shared__ uint32_t cpos[256];
        constexpr int CH = BIG / 2;                        // 4096 entries per chunk: the staged words take half of stage[], their bucket numbers (a byte each) sit in the other half -- no more LDS than the register path (occupancy)
        uint8_t* sbkt = reinterpret_cast<uint8_t*>(stage + 256 * CH);
        for (uint32_t c0 = 0; c0 < n; c0 += 256u * CH) {
            const uint32_t m = n - c0 < 256u * CH ? n - c0 : 256u * CH;
            ccnt[threadIdx.x] = 0; __syncthreads();
#pragma unroll
            for (int k = 0; k < CH; k++) { const uint32_t i = k * 256 + threadIdx.x; e[k] = i < m ? vals[start + c0 + i] : 0u; }
#pragma unroll
            for (int k = 0; k < CH; k++) if ((uint32_t)k * 256 + threadIdx.x < m) atomicAdd(&ccnt[(e[k] >> lowshift) & lowmask], 1u);
            __syncthreads();
            const uint32_t cm = ccnt[threadIdx.x];
            cpos[threadIdx.x] = cm; __syncthreads();
            for (int o = 1; o < 256; o <<= 1) { uint32_t a2 = (int)threadIdx.x >= o ? cpos[threadIdx.x - o] : 0; __syncthreads(); cpos[threadIdx.x] += a2; __syncthreads(); }
            const uint32_t cex = cpos[threadIdx.x] - cm;
            __syncthreads();
            cbase[threadIdx.x] = cex; cpos[threadIdx.x] = cex; __syncthreads();
#pragma unroll
            for (int k = 0; k < CH; k++) {
                if ((uint32_t)k * 256 < m) {                 // uniform
                    if ((uint32_t)k * 256 + threadIdx.x < m) { const uint32_t bk = (e[k] >> lowshift) & lowmask, q = atomicAdd(&cpos[bk], 1u); stage[q] = (e[k] & 0x80000000u) | (e[k] & rowmask); sbkt[q] = (uint8_t)bk; }
                    __syncthreads();
                }
            }
            for (uint32_t i = threadIdx.x; i < m; i += 256) { const uint32_t bk = sbkt[i]; vals2[start + pos[bk] + (i - cbase[bk])] = stage[i]; }
            __syncthreads();
            pos[threadIdx.x] += cm;                          // (buckets >= nb2 have cm = 0)
            __syncthreads();
        }
    }
}

// ---- segments ----
// segcnt[b] = ceil(size_b / seg); buckets cut into more than MSM_MERGE_T segments are listed for the wave-per-bucket merge
__global__ void __launch_bounds__(256)
zkc_msm_segcount(const uint32_t* __restrict__ bcnt, uint32_t nbuckets, uint32_t* __restrict__ segcnt, uint32_t* __restrict__ heavy,
                 uint32_t* __restrict__ heavy_count, uint32_t seg) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nbuckets) return;
    const uint32_t c = b < nbuckets ? (bcnt[b] + seg - 1) / seg : 0;
    segcnt[b] = c;
    if (c > (uint32_t)MSM_MERGE_T) { uint32_t k = atomicAdd(heavy_count, 1u); if (k < (uint32_t)MSM_MAX_HEAVY) heavy[k] = b; }
}
// device-wide exclusive scan of n u32 values in three launches: per-block sums (1024 values per block), scan of the sums (one block), fix-up
constexpr uint32_t SCAN_PER_BLOCK = 1024;
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* sh, uint32_t& total) {      // 256 threads
    sh[threadIdx.x] = v; __syncthreads();
    for (int o = 1; o < 256; o <<= 1) { uint32_t a = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0; __syncthreads(); sh[threadIdx.x] += a; __syncthreads(); }
    const uint32_t incl = sh[threadIdx.x]; total = sh[255]; __syncthreads();
    return incl - v;
}
__global__ void __launch_bounds__(256)
zkc_scan_local(const uint32_t* in, uint32_t* out, uint32_t* __restrict__ blk, uint32_t n) {        // in == out allowed
    __shared__ uint32_t sh[256];
    const uint32_t i0 = blockIdx.x * SCAN_PER_BLOCK + threadIdx.x * 4;
    uint32_t v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = i0 + k < n ? in[i0 + k] : 0; s += v[k]; }
    uint32_t total; uint32_t run = block_excl_scan_256(s, sh, total);
#pragma unroll
    for (int k = 0; k < 4; k++) { if (i0 + k < n) out[i0 + k] = run; run += v[k]; }
    if (threadIdx.x == 0) blk[blockIdx.x] = total;
}
__global__ void __launch_bounds__(256)
zkc_scan_blocks(uint32_t* __restrict__ blk, uint32_t nblk) {                 // in place, exclusive; one workgroup
    __shared__ uint32_t sh[256];
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nblk; b0 += 256) {
        const uint32_t i = b0 + threadIdx.x; const uint32_t v = i < nblk ? blk[i] : 0;
        uint32_t total; const uint32_t e = block_excl_scan_256(v, sh, total);
        if (i < nblk) blk[i] = carry + e;
        carry += total;
    }
}
__global__ void __launch_bounds__(256)
zkc_scan_fix(uint32_t* __restrict__ out, const uint32_t* __restrict__ blk, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] += blk[i / SCAN_PER_BLOCK];
}
// A bucket of L entries cut into k = ceil(L / seg) segments is split EVENLY: segment i covers [floor(i L / k), floor((i + 1) L / k)).
__device__ __forceinline__ void msm_seg_range_s(uint32_t L, uint32_t k, uint32_t i, uint32_t& lo, uint32_t& hi) {
    lo = (uint32_t)(((uint64_t)i * L) / k); hi = (uint32_t)(((uint64_t)(i + 1) * L) / k);
}
// seg2bucket[s] = bucket of segment s; seglen[s] = its number of entries
__global__ void __launch_bounds__(256)
zkc_msm_seg2bucket(const uint32_t* __restrict__ bcnt, const uint32_t* __restrict__ segoff, uint32_t nbuckets, uint32_t* __restrict__ seg2bucket,
                   uint32_t* __restrict__ seglen, uint32_t max_segments, uint32_t* __restrict__ perm_identity) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbuckets) return;
    const uint32_t s0 = segoff[b], k = segoff[b + 1] - s0, L = bcnt[b];
    for (uint32_t i = 0; i < k && s0 + i < max_segments; i++) {
        uint32_t lo, hi; msm_seg_range_s(L, k, i, lo, hi);
        seg2bucket[s0 + i] = b; seglen[s0 + i] = hi - lo;
        if (perm_identity) perm_identity[s0 + i] = s0 + i;          // a small pass: the segments keep their order (no length sort: five launches less on a latency chain)
    }
}
// ---- segments by decreasing length, STABLE (equal lengths keep their index order: neighbouring lanes of the accumulation then hold neighbouring
// segments, i.e. neighbouring table rows) ----  A counting sort over the keys MSM_SEG - len in [0, MSM_SEG]: per-workgroup histograms laid out
// [key][workgroup], one exclusive scan over that array, then a scatter with stable ranks inside the workgroup.
constexpr uint32_t LEN_KEYS = MSM_SEG + 1;
// stable rank of every thread among the threads of its 256-thread workgroup holding the same key (key < LEN_KEYS; invalid threads: key = ~0u);
// wcnt = LDS scratch of 4 x LEN_KEYS words.  Returns the rank; total[key] of the workgroup is left in wcnt[3 * LEN_KEYS + key] as INCLUSIVE wave prefix.
__device__ __forceinline__ uint32_t stable_rank_256(uint32_t key, bool valid, uint32_t* wcnt) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t k = threadIdx.x; k < 4 * LEN_KEYS; k += 256) wcnt[k] = 0;
    __syncthreads();
    uint32_t rank = 0;
    unsigned long long remaining = __ballot(valid);
    while (remaining) {
        const int leader = __ffsll((long long)remaining) - 1;
        const uint32_t v = (uint32_t)__shfl((int)key, leader);
        const unsigned long long m = __ballot(valid && key == v);
        if (valid && key == v) { rank = (uint32_t)__popcll(m & ((1ull << lane) - 1)); if ((int)lane == leader) wcnt[wave * LEN_KEYS + v] = (uint32_t)__popcll(m); }
        remaining &= ~m;
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < LEN_KEYS; k += 256) { uint32_t run = 0; for (int wv = 0; wv < 4; wv++) { run += wcnt[wv * LEN_KEYS + k]; wcnt[wv * LEN_KEYS + k] = run; } }   // inclusive over waves
    __syncthreads();
    if (valid && wave > 0) rank += wcnt[(wave - 1) * LEN_KEYS + key];
    return rank;
}
__global__ void __launch_bounds__(256)
zkc_msm_lenhist(const uint32_t* __restrict__ seglen, const uint32_t* __restrict__ segoff, uint32_t nbuckets, uint32_t* __restrict__ lencnt, uint32_t max_segments) {
    __shared__ uint32_t wcnt[4 * LEN_KEYS];
    uint32_t nseg = segoff[nbuckets]; if (nseg > max_segments) nseg = max_segments;
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    const bool valid = s < nseg;
    const uint32_t key = valid ? (uint32_t)MSM_SEG - seglen[s] : ~0u;
    (void)stable_rank_256(key, valid, wcnt);
    for (uint32_t k = threadIdx.x; k < LEN_KEYS; k += 256) lencnt[(size_t)k * gridDim.x + blockIdx.x] = wcnt[3 * LEN_KEYS + k];
}
__global__ void __launch_bounds__(256)
zkc_msm_lenscatter(const uint32_t* __restrict__ seglen, const uint32_t* __restrict__ segoff, uint32_t nbuckets, const uint32_t* __restrict__ lencnt,
                   uint32_t* __restrict__ perm, uint32_t max_segments) {
    __shared__ uint32_t wcnt[4 * LEN_KEYS];
    uint32_t nseg = segoff[nbuckets]; if (nseg > max_segments) nseg = max_segments;
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    const bool valid = s < nseg;
    const uint32_t key = valid ? (uint32_t)MSM_SEG - seglen[s] : ~0u;
    const uint32_t rank = stable_rank_256(key, valid, wcnt);
    if (valid) perm[lencnt[(size_t)key * gridDim.x + blockIdx.x] + rank] = s;
}

#define ZKC_SORT_LAUNCH_CHECK(name) do { hipError_t _e = hipGetLastError(); if (_e != hipSuccess) return zkc_fail(ctx, ZKC_ERR_HIP, std::string(name ": ") + hipGetErrorString(_e)); } while (0)

int msm_bucket_entries(zkc_ctx* ctx, MsmWork& w, const MsmJobList& jl, hipStream_t st, unsigned long long* d_entry_counter) {
    if (jl.total_bins > w.max_bins) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "msm_bucket_entries: too many level-1 bins for the work space");
    const MsmJobList* dj = (const MsmJobList*)w.d_jobs;
    if (jl.total_tilecnt > w.max_tilecnt) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "msm_bucket_entries: too many (bin, tile) counters for the work space");
    hipLaunchKernelGGL(zkc_msm_count, dim3(jl.total_tiles), dim3(MSM_TILE), 0, st, dj, (const uint16_t*)w.d_tilejob, w.tilecnt);
    ZKC_SORT_LAUNCH_CHECK("zkc_msm_count");
    hipLaunchKernelGGL(zkc_msm_binscan, dim3(jl.njobs), dim3(256), 0, st, dj, w.tilecnt, w.hist, w.bin_start, d_entry_counter);
    ZKC_SORT_LAUNCH_CHECK("zkc_msm_binscan");
    hipLaunchKernelGGL(zkc_msm_split, dim3(jl.total_tiles), dim3(MSM_TILE), 0, st, dj, (const uint16_t*)w.d_tilejob, w.tilecnt, w.vals);
    ZKC_SORT_LAUNCH_CHECK("zkc_msm_split");
    uint64_t biggest_bin = 0;                               // expected entries per level-1 bin of the pass' largest job (uniform digits)
    for (int j = 0; j < jl.njobs; j++) biggest_bin = std::max<uint64_t>(biggest_bin, ((uint64_t)jl.job[j].count * jl.job[j].nw) >> jl.job[j].hbits);
    if (biggest_bin > 8192) hipLaunchKernelGGL(HIP_KERNEL_NAME(zkc_msm_bucket<true>), dim3(jl.total_bins), dim3(256), 0, st, dj, w.hist, w.bin_start, w.vals, w.vals2, w.off, w.bcnt);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(zkc_msm_bucket<false>), dim3(jl.total_bins), dim3(256), 0, st, dj, w.hist, w.bin_start, w.vals, w.vals2, w.off, w.bcnt);
    ZKC_SORT_LAUNCH_CHECK("zkc_msm_bucket");
    return ZKC_OK;
}

static int device_scan(zkc_ctx* ctx, const uint32_t* in, uint32_t* out, uint32_t* blk, uint32_t n, hipStream_t st) {
    const uint32_t nblk = (n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;
    hipLaunchKernelGGL(zkc_scan_local, dim3(nblk), dim3(256), 0, st, in, out, blk, n);
    hipLaunchKernelGGL(zkc_scan_blocks, dim3(1), dim3(256), 0, st, blk, nblk);
    hipLaunchKernelGGL(zkc_scan_fix, dim3((n + 255) / 256), dim3(256), 0, st, out, blk, n);
    ZKC_SORT_LAUNCH_CHECK("zkc_scan");
    return ZKC_OK;
}
int msm_build_segments(zkc_ctx* ctx, MsmWork& w, const MsmJobList& jl, uint32_t seg, size_t seg_bound, hipStream_t st) {
    const uint32_t nb = jl.total_buckets;
    if (seg > (uint32_t)MSM_SEG) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "msm_build_segments: segment length");
    ZKC_HIP_CHECK(ctx, hipMemsetAsync(w.heavy + MSM_MAX_HEAVY, 0, 4, st));
    hipLaunchKernelGGL(zkc_msm_segcount, dim3((nb + 1 + 255) / 256), dim3(256), 0, st, w.bcnt, nb, w.segcnt, w.heavy, w.heavy + MSM_MAX_HEAVY, seg);
    ZKC_SORT_LAUNCH_CHECK("zkc_msm_segcount");
    int rc = device_scan(ctx, w.segcnt, w.segoff, w.scan_blk, nb + 1, st); if (rc) return rc;
    // sorting the segments by length keeps the lanes of a wave busy for the same time: worth five launches in a full pass (seg = 128), not when every segment is ~16 entries
    const bool sort_by_len = seg > (uint32_t)MSM_SEG_MIN * 2;
    hipLaunchKernelGGL(zkc_msm_seg2bucket, dim3((nb + 255) / 256), dim3(256), 0, st, w.bcnt, w.segoff, nb, w.seg2bucket, w.seglen, (uint32_t)w.max_segments, sort_by_len ? nullptr : w.perm);
    ZKC_SORT_LAUNCH_CHECK("zkc_msm_seg2bucket");
    if (!sort_by_len) return ZKC_OK;
    const uint32_t nwg = (uint32_t)((seg_bound + 255) / 256), nlen = LEN_KEYS * nwg;
    if ((size_t)nlen > w.max_lencnt) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "msm_build_segments: too many segments for the work space");
    hipLaunchKernelGGL(zkc_msm_lenhist, dim3(nwg), dim3(256), 0, st, w.seglen, w.segoff, nb, w.lencnt, (uint32_t)w.max_segments);
    ZKC_SORT_LAUNCH_CHECK("zkc_msm_lenhist");
    rc = device_scan(ctx, w.lencnt, w.lencnt, w.scan_blk, nlen, st); if (rc) return rc;
    hipLaunchKernelGGL(zkc_msm_lenscatter, dim3(nwg), dim3(256), 0, st, w.seglen, w.segoff, nb, w.lencnt, w.perm, (uint32_t)w.max_segments);
    ZKC_SORT_LAUNCH_CHECK("zkc_msm_lenscatter");
    return ZKC_OK;
}

}  // namespace zkc
