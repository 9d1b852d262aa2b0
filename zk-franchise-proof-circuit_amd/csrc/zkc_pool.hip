// zkc_pool.hip -- several GPUs from ONE host process ("GPU g of G proves voters [g B/G, (g+1) B/G); one host thread + one HIP
// stream set per device; zkey uploaded once per device").  The reference's hosts are single processes -- a Go service looping prover.Prove over
// voters (zk_census_test.go:89) or a Node process calling groth16.fullProve (ts_inputs/src/example.ts:358-362) -- so the C ABI offers the same block
// split bench.py performs with one process per GPU: a pool owns one context and one resident key per device and proves a batch with one host
// thread per device.  Independent proofs: no data-path exchange between the devices, the "gather" is each thread writing its own slice of the
// caller's output arrays.  Host code over the public entry points of include/zkcensus.h; nothing here launches a kernel itself.
#include "zkc_prover.h"
#include <mutex>
#include <string>
#include <thread>
#include <vector>

struct zkc_pool {
    struct Dev {
        zkc_ctx* ctx = nullptr; zkc_zkey* key = nullptr; int device = 0;
        void *d_in = nullptr, *d_wtns = nullptr; int32_t* d_status = nullptr; size_t in_sz = 0, wtns_sz = 0, status_n = 0;
        int rc = 0; std::string err;
    };
    std::vector<Dev> dev;
    std::mutex mu;                 // one batch (or key load) at a time per pool; the contexts keep their own locks
    std::string err;
};
static thread_local std::string g_pool_create_err;

namespace {
int pool_fail(zkc_pool* p, int code, const std::string& msg) { if (p) p->err = msg; else g_pool_create_err = msg; return code; }
// voters [lo, hi) of device g: contiguous blocks whose sizes differ by at most one (parallel.py::shard_range)
void shard_range(int g, int G, int total, int& lo, int& hi) {
    const int base = total / G, rem = total % G;
    lo = g * base + (g < rem ? g : rem); hi = lo + base + (g < rem ? 1 : 0);
}
// grow-only device buffer on the calling thread's current device
int ensure(void** p, size_t* have, size_t want) {
    if (*have >= want) return 0;
    if (*p) { (void)hipFree(*p); *p = nullptr; *have = 0; }
    if (hipMalloc(p, want) != hipSuccess) { (void)hipGetLastError(); return 1; }
    *have = want; return 0;
}
void dev_release(zkc_pool::Dev& d) {
    if (d.ctx) (void)hipSetDevice(d.device);
    if (d.key) zkc_zkey_free(d.key);
    if (d.d_in) (void)hipFree(d.d_in);
    if (d.d_wtns) (void)hipFree(d.d_wtns);
    if (d.d_status) (void)hipFree(d.d_status);
    if (d.ctx) zkc_ctx_destroy(d.ctx);
    d = zkc_pool::Dev();
}
}  // namespace

extern "C" int zkc_pool_create(const int* hip_devices, int n, zkc_pool** out) {
    if (!hip_devices || !out || n <= 0 || n > 64) return pool_fail(nullptr, ZKC_ERR_BAD_ARG, "zkc_pool_create: bad argument");
    zkc_pool* p = new zkc_pool(); p->dev.resize((size_t)n);
    for (int g = 0; g < n; g++) {
        p->dev[g].device = hip_devices[g];
        const int rc = zkc_ctx_create(hip_devices[g], &p->dev[g].ctx);
        if (rc) {
            const std::string why = "zkc_pool_create: device " + std::to_string(hip_devices[g]) + ": " + zkc_last_error(nullptr);
            for (auto& d : p->dev) dev_release(d);
            delete p; return pool_fail(nullptr, rc, why);
        }
    }
    *out = p; return ZKC_OK;
}

extern "C" void zkc_pool_destroy(zkc_pool* p) {
    if (!p) return;
    { std::lock_guard<std::mutex> g(p->mu); for (auto& d : p->dev) dev_release(d); }
    delete p;
}

extern "C" int zkc_pool_size(const zkc_pool* p) { return p ? (int)p->dev.size() : 0; }
extern "C" zkc_ctx* zkc_pool_ctx(zkc_pool* p, int i) { return (p && i >= 0 && i < (int)p->dev.size()) ? p->dev[i].ctx : nullptr; }
extern "C" zkc_zkey* zkc_pool_zkey(zkc_pool* p, int i) { return (p && i >= 0 && i < (int)p->dev.size()) ? p->dev[i].key : nullptr; }
extern "C" const char* zkc_pool_last_error(const zkc_pool* p) { return p ? p->err.c_str() : g_pool_create_err.c_str(); }

// one resident copy of the key per device, the devices loading side by side (parsing is per device too: the loader's host work is
// 0.2 s of a 0.55 s load, not worth sharing across contexts that otherwise share nothing)
extern "C" int zkc_pool_zkey_load(zkc_pool* p, const void* zkey_bytes, size_t len) {
    if (!p || !zkey_bytes || !len) return pool_fail(p, ZKC_ERR_BAD_ARG, "zkc_pool_zkey_load: bad argument");
    std::lock_guard<std::mutex> g(p->mu);
    std::vector<std::thread> th;
    for (auto& d : p->dev)
        th.emplace_back([&d, zkey_bytes, len] {
            (void)hipSetDevice(d.device);
            if (d.key) { zkc_zkey_free(d.key); d.key = nullptr; }
            d.rc = zkc_zkey_load(d.ctx, zkey_bytes, len, &d.key);
            if (d.rc) d.err = zkc_last_error(d.ctx);
        });
    for (auto& t : th) t.join();
    for (auto& d : p->dev)
        if (d.rc) {
            const int rc = d.rc; const std::string why = "zkc_pool_zkey_load: device " + std::to_string(d.device) + ": " + d.err;
            for (auto& e : p->dev) if (e.key) { (void)hipSetDevice(e.device); zkc_zkey_free(e.key); e.key = nullptr; }      // all devices or none
            return pool_fail(p, rc, why);
        }
    return ZKC_OK;
}

// groth16.fullProve for B voters over all devices of the pool.  Host buffers throughout (what a cgo / N-API caller holds):
//   inputs  B x zkc_circuit_n_inputs x 32 B ; rs  B x 64 B (NULL: drawn here, uniform in Fr) ; proofs  B x 256 B ; publics  B x nPublic x 32 B or NULL ;
//   status  B x int32 (ZKC_W_*) or NULL.
// Returns ZKC_OK, or ZKC_ERR_WITNESS when every device finished and at least one voter failed a circuit assert (its status says which; the other
// proofs are valid), or the first device's error otherwise.
extern "C" int zkc_pool_fullprove_batch(zkc_pool* p, const void* inputs, int B, const uint8_t* rs, uint8_t* proofs, uint8_t* publics, int32_t* status) {
    if (!p || !inputs || !proofs || B <= 0) return pool_fail(p, ZKC_ERR_BAD_ARG, "zkc_pool_fullprove_batch: bad argument");
    std::lock_guard<std::mutex> g(p->mu);
    const int G = (int)p->dev.size();
    for (auto& d : p->dev) if (!d.key) return pool_fail(p, ZKC_ERR_BAD_ARG, "zkc_pool_fullprove_batch: no key loaded (zkc_pool_zkey_load)");
    const zkc_zkey* k0 = p->dev[0].key;
    if (k0->nLevels < 0) return pool_fail(p, ZKC_ERR_BAD_ARG, "zkc_pool_fullprove_batch: the key is not a ZkFranchiseProofCircuit(nLevels) key");
    const size_t nIn = (size_t)zkc_circuit_n_inputs(k0->nLevels), nW = k0->nVars, nPub = k0->nPub;
    std::vector<uint8_t> own_rs;
    if (!rs) { own_rs.resize((size_t)B * 64); zkc_random_scalars(own_rs.data(), (size_t)B * 2); rs = own_rs.data(); }
    std::vector<std::thread> th;
    for (int gi = 0; gi < G; gi++) {
        zkc_pool::Dev& d = p->dev[gi]; d.rc = 0; d.err.clear();
        int lo, hi; shard_range(gi, G, B, lo, hi);
        if (hi == lo) continue;
        th.emplace_back([&d, lo, hi, nIn, nW, nPub, inputs, rs, proofs, publics, status] {
            const size_t n = (size_t)(hi - lo);
            auto hip_fail = [&d](const char* what) { d.rc = ZKC_ERR_HIP; d.err = what; (void)hipGetLastError(); };
            if (hipSetDevice(d.device) != hipSuccess) return hip_fail("hipSetDevice");
            size_t st_bytes = d.status_n * sizeof(int32_t);
            if (ensure(&d.d_in, &d.in_sz, n * nIn * 32) || ensure(&d.d_wtns, &d.wtns_sz, n * nW * 32) || ensure((void**)&d.d_status, &st_bytes, n * sizeof(int32_t)))
                return hip_fail("hipMalloc of the per-device input / witness buffers");
            d.status_n = st_bytes / sizeof(int32_t);
            if (hipMemcpy(d.d_in, (const uint8_t*)inputs + (size_t)lo * nIn * 32, n * nIn * 32, hipMemcpyHostToDevice) != hipSuccess) return hip_fail("hipMemcpy of the inputs");
            d.rc = zkc_fullprove_batch_dev(d.key, d.d_in, (int)n, d.d_wtns, d.d_status, rs + (size_t)lo * 64, proofs + (size_t)lo * 256,
                                           publics ? publics + (size_t)lo * nPub * 32 : nullptr);
            if (d.rc) { d.err = zkc_last_error(d.ctx); return; }
            std::vector<int32_t> st(n);
            if (hipMemcpy(st.data(), d.d_status, n * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return hip_fail("hipMemcpy of the status words");
            bool bad = false;
            for (size_t i = 0; i < n; i++) { if (status) status[lo + i] = st[i]; bad = bad || st[i] != ZKC_W_OK; }
            if (bad) { d.rc = ZKC_ERR_WITNESS; d.err = "at least one voter failed a circuit assert"; }
        });
    }
    for (auto& t : th) t.join();
    int soft = 0;
    for (auto& d : p->dev) {
        if (d.rc == ZKC_ERR_WITNESS) { soft = ZKC_ERR_WITNESS; continue; }
        if (d.rc) return pool_fail(p, d.rc, "zkc_pool_fullprove_batch: device " + std::to_string(d.device) + ": " + d.err);
    }
    if (soft) return pool_fail(p, soft, "zkc_pool_fullprove_batch: at least one voter failed a circuit assert (see status)");
    return ZKC_OK;
}
