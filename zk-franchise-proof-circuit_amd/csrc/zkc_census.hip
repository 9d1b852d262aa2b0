// zkc_census.hip -- f1: the census / voter generator in native code.
//
// The reference builds its test census with arbo (internal/helpers.go:36-85 GenTree: arbo.NewTree{HashFunctionPoseidon}, Add per voter, GenProof, UnpackSiblings, zero padding
// to nLevels + 1) and fills the circuit inputs from it (internal/inputs.go:33-98 MockInputs; ts_inputs/src/inputs.ts:38-88).  Rounds 1-4 did that in Python over a batched GPU
// Poseidon: 10 s for the 8 192-voter census of BASELINE configs[2..3], most of it Python lists and one host round trip per tree level.  Here: the trie over the keys is split
// on the host in C++ (sort by path, binary-search the split of every node: microseconds per thousand leaves), every hash runs on the GPU -- leaves in one launch, inner nodes
// one launch per depth, bottom-up, values never leaving HBM -- and every voter's sibling list is scattered straight into its 334 x 32-byte input block on the device.
//
// arbo tree semantics (pinned by the reference's one arbo-built path, tests/test_gpu_census.py): leaf = H(key, value, 1); node = H(left, right); path bit i =
// bit i (LSB first) of the key; an empty subtree is 0; a subtree holding a single leaf is that leaf's hash (a leaf sits at the first depth where its path is unique).
// A static build (all leaves at once), not arbo's incremental Add: the tree is the same, the order of insertion does not matter to a Merkle radix tree.
#include "zkc_internal.h"
#include "zkc_field.h"
#include <algorithm>
#include <array>
#include <cstring>
#include <vector>

using namespace zkc;

extern "C" __global__ void zkc_census_hash(PoseidonTable, int, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*, size_t);
extern "C" __global__ void zkc_census_level(PoseidonTable, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t);
extern "C" __global__ void zkc_census_scatter(const uint32_t*, const uint2*, size_t, uint32_t*);
extern "C" __global__ void zkc_census_scalars(const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*,
                                              const uint32_t*, const uint32_t*, size_t, int, uint32_t*);

namespace {
// the radix trie over n keys: inner nodes with their two child references and depth, grouped by depth; for every leaf the (depth, sibling reference) pairs of its path.
// References index the tree's value array: 0 = empty, 1 + i = leaf i, 1 + n + j = inner node j.
struct Trie {
    size_t n = 0; std::vector<uint32_t> left, right, depth, order, first_of_depth;      // per node / nodes grouped by depth (order), first_of_depth[d] .. first_of_depth[d + 1]
    uint32_t root = 0, max_depth = 0;
    struct Sib { uint32_t leaf, level, ref; }; std::vector<Sib> sibs;                    // non-zero siblings only
    std::vector<int32_t> leaf_depth;                                                       // 1 + the deepest level at which leaf i has an inner node above it (0: alone in the tree)
};
inline uint64_t bitrev64(uint64_t x) {
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1); x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0f0f0f0f0f0f0f0full) | ((x & 0x0f0f0f0f0f0f0f0full) << 4); return __builtin_bswap64(x);
}
inline int key_bit(const uint8_t* keys, size_t i, int d) { return (keys[32 * i + (d >> 3)] >> (d & 7)) & 1; }
// false: two keys agree on their first max_levels path bits (arbo: the tree cannot hold both)
bool trie_build(const uint8_t* keys, size_t n, int max_levels, Trie& t, std::string& err) {
    t = Trie(); t.n = n; t.leaf_depth.assign(n, 0);
    if (n == 0) return true;
    // leaves in path order: sort by the key read bit by bit from the LSB, i.e. by the bit-reversed 256-bit key
    std::vector<std::array<uint64_t, 4>> rk(n); std::vector<uint32_t> idx(n);
    for (size_t i = 0; i < n; i++) { uint64_t w[4]; memcpy(w, keys + 32 * i, 32); for (int k = 0; k < 4; k++) rk[i][k] = bitrev64(w[k]); idx[i] = (uint32_t)i; }
    std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return rk[a] < rk[b]; });
    for (size_t i = 0; i + 1 < n; i++) if (rk[idx[i]] == rk[idx[i + 1]]) { err = "zkc census: duplicate key"; return false; }
    // work list: the range [lo, hi) of idx whose subtree reference becomes child `side` of `node` (child references go in through indices, not pointers: the vectors grow)
    struct Pend { uint32_t lo, hi, depth, node, side; };
    std::vector<Pend> pend; pend.push_back({0, (uint32_t)n, 0, 0xffffffffu, 0});
    auto ref_of = [&](uint32_t lo, uint32_t hi, uint32_t depth, bool& ok) -> uint32_t {
        if (hi == lo) return 0;
        if (hi - lo == 1) return 1 + idx[lo];
        if ((int)depth >= max_levels) { ok = false; return 0; }
        const uint32_t nid = (uint32_t)t.left.size();
        t.left.push_back(0); t.right.push_back(0); t.depth.push_back(depth);
        // the split: first position whose key has bit `depth` set (within the range every key shares the bits below `depth`, so the range is partitioned)
        uint32_t a = lo, b = hi; while (a < b) { const uint32_t m = (a + b) / 2; if (key_bit(keys, idx[m], (int)depth)) b = m; else a = m + 1; }
        pend.push_back({lo, a, depth + 1, nid, 0}); pend.push_back({a, hi, depth + 1, nid, 1});
        return 1 + (uint32_t)n + nid;
    };
    std::vector<std::array<uint32_t, 3>> range;                        // per node: lo, mid, hi
    bool ok = true;
    for (size_t q = 0; q < pend.size() && ok; q++) {
        const Pend p = pend[q];
        const size_t before = t.left.size();
        const uint32_t r = ref_of(p.lo, p.hi, p.depth, ok);
        if (t.left.size() > before) { const Pend& l = pend[pend.size() - 2]; range.push_back({l.lo, l.hi, p.hi}); }
        if (p.node == 0xffffffffu) t.root = r; else (p.side ? t.right : t.left)[p.node] = r;
    }
    if (!ok) { err = "zkc census: two keys collide on the first " + std::to_string(max_levels) + " bits of their paths"; return false; }
    const size_t nn = t.left.size();
    for (size_t j = 0; j < nn; j++) t.max_depth = std::max(t.max_depth, t.depth[j] + 1);
    t.first_of_depth.assign(t.max_depth + 2, 0);
    for (size_t j = 0; j < nn; j++) t.first_of_depth[t.depth[j] + 1]++;
    for (size_t d = 0; d + 1 < t.first_of_depth.size(); d++) t.first_of_depth[d + 1] += t.first_of_depth[d];
    t.order.resize(nn); { std::vector<uint32_t> fill(t.first_of_depth.begin(), t.first_of_depth.end() - 1); for (size_t j = 0; j < nn; j++) t.order[fill[t.depth[j]]++] = (uint32_t)j; }
    // sibling lists: the members of node j's left range see its right child, and the other way round (zero siblings are what the zeroed block already holds)
    for (size_t j = 0; j < nn; j++) {
        const uint32_t lo = range[j][0], mid = range[j][1], hi = range[j][2], d = t.depth[j];
        for (uint32_t k = lo; k < hi; k++) {
            const uint32_t leaf = idx[k], sib = k < mid ? t.right[j] : t.left[j];
            if (sib) t.sibs.push_back({leaf, d, sib});
            if ((int32_t)d + 1 > t.leaf_depth[leaf]) t.leaf_depth[leaf] = (int32_t)d + 1;
        }
    }
    return true;
}
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(zkc_ctx* ctx, size_t bytes) { ZKC_HIP_CHECK(ctx, hipMalloc(&p, bytes ? bytes : 4)); return ZKC_OK; }
    template <class T> T* as() { return (T*)p; }
};
struct DevTrie { DevBuf left, right, order; };
int trie_upload(zkc_ctx* ctx, const Trie& t, DevTrie& d) {
    int rc; const size_t nn = t.left.size();
    if ((rc = d.left.alloc(ctx, nn * 4)) || (rc = d.right.alloc(ctx, nn * 4)) || (rc = d.order.alloc(ctx, nn * 4))) return rc;
    if (nn) {
        ZKC_HIP_CHECK(ctx, hipMemcpyAsync(d.left.p, t.left.data(), nn * 4, hipMemcpyHostToDevice, ctx->stream)); ZKC_HIP_CHECK(ctx, hipMemcpyAsync(d.right.p, t.right.data(), nn * 4, hipMemcpyHostToDevice, ctx->stream));
        ZKC_HIP_CHECK(ctx, hipMemcpyAsync(d.order.p, t.order.data(), nn * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    return ZKC_OK;
}
// val (device, (1 + n + nodes) x 32 B): leaf hashes from (d_keys, d_values), then the inner nodes depth by depth, bottom-up; all on ctx->stream
int tree_hash(zkc_ctx* ctx, const Trie& t, const DevTrie& dt, const uint32_t* d_keys, const uint32_t* d_values, uint32_t* d_val) {
    const size_t n = t.n;
    ZKC_HIP_CHECK(ctx, hipMemsetAsync(d_val, 0, 32, ctx->stream));
    if (n) hipLaunchKernelGGL(zkc_census_hash, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, ctx->ptab, 0, d_keys, d_values, (const uint32_t*)nullptr, d_val + 8, n);
    for (int d = (int)t.max_depth - 1; d >= 0; d--) {
        const uint32_t first = t.first_of_depth[d], count = t.first_of_depth[d + 1] - first;
        if (count) hipLaunchKernelGGL(zkc_census_level, dim3((count + 63) / 64), dim3(64), 0, ctx->stream, ctx->ptab, (const uint32_t*)dt.left.p, (const uint32_t*)dt.right.p,
                                      (const uint32_t*)dt.order.p, first, count, d_val, (uint32_t)(1 + n));
    }
    ZKC_HIP_CHECK(ctx, hipGetLastError());
    return ZKC_OK;
}
// the sibling lists as (destination element, value reference) pairs: element = 32-byte slot of the output array; leaf i's level-l sibling goes to i * stride + base + l
int scatter_siblings(zkc_ctx* ctx, const Trie& t, const uint32_t* d_val, size_t stride, size_t base, uint32_t* d_out, DevBuf& pairs) {
    std::vector<uint2> h(t.sibs.size());
    for (size_t k = 0; k < t.sibs.size(); k++) { const size_t dst = (size_t)t.sibs[k].leaf * stride + base + t.sibs[k].level; if (dst >> 32) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "zkc census: too many voters for 32-bit slots"); h[k] = make_uint2((uint32_t)dst, t.sibs[k].ref); }
    int rc; if ((rc = pairs.alloc(ctx, h.size() * sizeof(uint2)))) return rc;
    if (!h.empty()) {
        ZKC_HIP_CHECK(ctx, hipMemcpyAsync(pairs.p, h.data(), h.size() * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(zkc_census_scatter, dim3((unsigned)((h.size() + 255) / 256)), dim3(256), 0, ctx->stream, d_val, (const uint2*)pairs.p, h.size(), d_out);
        ZKC_HIP_CHECK(ctx, hipGetLastError());
        ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));          // `h` is pageable: the copy must have left it before it goes out of scope
    }
    return ZKC_OK;
}
bool all_below_r(const void* v, size_t count) {
    for (size_t i = 0; i < count; i++) { uint32_t t[8]; memcpy(t, (const uint8_t*)v + 32 * i, 32); if (!fp_std_lt_p<FrParams>(t)) return false; }
    return true;
}
}  // namespace

// One tree.  keys, values: n x 32 B, host, standard form, < r, keys distinct.  root: 32 B.  siblings (may be NULL): n x (nLevels + 1) x 32 B, leaf i's sibling at level l in
// slot i (nLevels + 1) + l, zero-padded the way internal/helpers.go:72-79 pads arbo's packed siblings.  depths (may be NULL): per leaf the number of levels above it.
extern "C" int zkc_smt_build(zkc_ctx* ctx, const void* keys, const void* values, size_t n, int nLevels, uint8_t root[32], void* siblings, int32_t* depths) {
    if (!ctx || !keys || !values || !root || n == 0 || n > (1u << 28) || nLevels < 1 || nLevels > 253) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "zkc_smt_build: bad argument");
    if (!all_below_r(keys, n) || !all_below_r(values, n)) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "zkc_smt_build: a key or value is not below the field order");
    Trie t; std::string err;
    if (!trie_build((const uint8_t*)keys, n, nLevels, t, err)) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, err);
    ZKC_LOCK(ctx);
    ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    DevTrie dt; DevBuf dk, dv, dval, dout, pairs; int rc;
    const size_t nn = t.left.size(), stride = (size_t)nLevels + 1;
    if ((rc = trie_upload(ctx, t, dt)) || (rc = dk.alloc(ctx, 32 * n)) || (rc = dv.alloc(ctx, 32 * n)) || (rc = dval.alloc(ctx, 32 * (1 + n + nn)))) return rc;
    ZKC_HIP_CHECK(ctx, hipMemcpyAsync(dk.p, keys, 32 * n, hipMemcpyHostToDevice, ctx->stream)); ZKC_HIP_CHECK(ctx, hipMemcpyAsync(dv.p, values, 32 * n, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = tree_hash(ctx, t, dt, dk.as<uint32_t>(), dv.as<uint32_t>(), dval.as<uint32_t>()))) return rc;
    ZKC_HIP_CHECK(ctx, hipMemcpyAsync(root, dval.as<uint8_t>() + 32 * (size_t)t.root, 32, hipMemcpyDeviceToHost, ctx->stream));
    if (siblings) {
        if ((rc = dout.alloc(ctx, 32 * n * stride))) return rc;
        ZKC_HIP_CHECK(ctx, hipMemsetAsync(dout.p, 0, 32 * n * stride, ctx->stream));
        if ((rc = scatter_siblings(ctx, t, dval.as<uint32_t>(), stride, 0, dout.as<uint32_t>(), pairs))) return rc;
        ZKC_HIP_CHECK(ctx, hipMemcpyAsync(siblings, dout.p, 32 * n * stride, hipMemcpyDeviceToHost, ctx->stream));
    }
    ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (depths) memcpy(depths, t.leaf_depth.data(), 4 * n);
    return ZKC_OK;
}

// The circuit inputs of a whole census (internal/inputs.go:33-98 MockInputs for every voter of one election): from the voters' own data -- address, password, signature,
// available and cast weight, vote hash -- the SIK H(address, password, signature) and the nullifier H(signature, password, electionId) of every voter, the census tree
// (address -> available weight) and the SIK tree (address -> SIK), both roots, and every voter's two sibling lists, assembled into n blocks of zkc_circuit_n_inputs(nLevels)
// x 32 B in census.circom:51-67 order.  All arrays n x 32 B (vote_hash n x 2 x 32 B, election_id 2 x 32 B), host, standard form, < r.  The blocks go to inputs_out (host)
// and / or d_inputs_out (device: what zkc_fullprove_batch_dev / zkc_batch_begin take), either may be NULL.  roots_out (may be NULL): census root, SIK root, 2 x 32 B.
extern "C" int zkc_census_inputs(zkc_ctx* ctx, size_t n, int nLevels, const uint8_t election_id[64], const void* address, const void* password, const void* signature,
                                 const void* available_weight, const void* vote_weight, const void* vote_hash, void* inputs_out, void* d_inputs_out, uint8_t* roots_out) {
    if (!ctx || !election_id || !address || !password || !signature || !available_weight || !vote_weight || !vote_hash || n == 0 || n > (1u << 24) || nLevels < 3 || nLevels > 253 || (!inputs_out && !d_inputs_out))
        return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "zkc_census_inputs: bad argument");
    if (!all_below_r(election_id, 2) || !all_below_r(address, n) || !all_below_r(password, n) || !all_below_r(signature, n) || !all_below_r(available_weight, n) || !all_below_r(vote_weight, n) || !all_below_r(vote_hash, 2 * n))
        return zkc_fail(ctx, ZKC_ERR_BAD_ARG, "zkc_census_inputs: a value is not below the field order");
    Trie t; std::string err;
    if (!trie_build((const uint8_t*)address, n, nLevels, t, err)) return zkc_fail(ctx, ZKC_ERR_BAD_ARG, err);      // both trees are keyed by the address: one trie
    ZKC_LOCK(ctx);
    ZKC_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t nn = t.left.size(), nIn = 12 + 2 * ((size_t)nLevels + 1);
    DevTrie dt; DevBuf d_eid, d_addr, d_pw, d_sig, d_av, d_vw, d_vh, d_sik, d_null, val_c, val_s, own_out, pairs_c, pairs_s; int rc;
    if ((rc = trie_upload(ctx, t, dt))) return rc;
    struct Up { DevBuf* b; const void* src; size_t bytes; } ups[] = {{&d_eid, election_id, 64}, {&d_addr, address, 32 * n}, {&d_pw, password, 32 * n}, {&d_sig, signature, 32 * n},
                                                                    {&d_av, available_weight, 32 * n}, {&d_vw, vote_weight, 32 * n}, {&d_vh, vote_hash, 64 * n}};
    for (auto& u : ups) { if ((rc = u.b->alloc(ctx, u.bytes))) return rc; ZKC_HIP_CHECK(ctx, hipMemcpyAsync(u.b->p, u.src, u.bytes, hipMemcpyHostToDevice, ctx->stream)); }
    if ((rc = d_sik.alloc(ctx, 32 * n)) || (rc = d_null.alloc(ctx, 32 * n)) || (rc = val_c.alloc(ctx, 32 * (1 + n + nn))) || (rc = val_s.alloc(ctx, 32 * (1 + n + nn)))) return rc;
    uint32_t* d_out = (uint32_t*)d_inputs_out;
    if (!d_out) { if ((rc = own_out.alloc(ctx, 32 * n * nIn))) return rc; d_out = own_out.as<uint32_t>(); }
    const unsigned g64 = (unsigned)((n + 63) / 64);
    hipLaunchKernelGGL(zkc_census_hash, dim3(g64), dim3(64), 0, ctx->stream, ctx->ptab, 1, d_addr.as<uint32_t>(), d_pw.as<uint32_t>(), d_sig.as<uint32_t>(), d_sik.as<uint32_t>(), n);       // census.circom:74-77
    hipLaunchKernelGGL(zkc_census_hash, dim3(g64), dim3(64), 0, ctx->stream, ctx->ptab, 2, d_sig.as<uint32_t>(), d_pw.as<uint32_t>(), d_eid.as<uint32_t>(), d_null.as<uint32_t>(), n);      // :105-109
    ZKC_HIP_CHECK(ctx, hipGetLastError());
    if ((rc = tree_hash(ctx, t, dt, d_addr.as<uint32_t>(), d_av.as<uint32_t>(), val_c.as<uint32_t>()))) return rc;       // census tree: address -> available weight
    if ((rc = tree_hash(ctx, t, dt, d_addr.as<uint32_t>(), d_sik.as<uint32_t>(), val_s.as<uint32_t>()))) return rc;      // SIK tree: address -> SIK
    ZKC_HIP_CHECK(ctx, hipMemsetAsync(d_out, 0, 32 * n * nIn, ctx->stream));
    hipLaunchKernelGGL(zkc_census_scalars, dim3((unsigned)((12 * n + 255) / 256)), dim3(256), 0, ctx->stream, d_eid.as<uint32_t>(), d_null.as<uint32_t>(), d_av.as<uint32_t>(), d_vh.as<uint32_t>(),
                       val_s.as<uint32_t>() + 8 * (size_t)t.root, val_c.as<uint32_t>() + 8 * (size_t)t.root, d_addr.as<uint32_t>(), d_pw.as<uint32_t>(), d_sig.as<uint32_t>(), d_vw.as<uint32_t>(), n, (int)nIn, d_out);
    ZKC_HIP_CHECK(ctx, hipGetLastError());
    if ((rc = scatter_siblings(ctx, t, val_c.as<uint32_t>(), nIn, 12, d_out, pairs_c))) return rc;
    if ((rc = scatter_siblings(ctx, t, val_s.as<uint32_t>(), nIn, 12 + (size_t)nLevels + 1, d_out, pairs_s))) return rc;
    if (inputs_out) ZKC_HIP_CHECK(ctx, hipMemcpyAsync(inputs_out, d_out, 32 * n * nIn, hipMemcpyDeviceToHost, ctx->stream));
    if (roots_out) { ZKC_HIP_CHECK(ctx, hipMemcpyAsync(roots_out, val_c.as<uint8_t>() + 32 * (size_t)t.root, 32, hipMemcpyDeviceToHost, ctx->stream));
                     ZKC_HIP_CHECK(ctx, hipMemcpyAsync(roots_out + 32, val_s.as<uint8_t>() + 32 * (size_t)t.root, 32, hipMemcpyDeviceToHost, ctx->stream)); }
    ZKC_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ZKC_OK;
}
