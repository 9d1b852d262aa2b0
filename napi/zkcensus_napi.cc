// zkcensus_napi.cc -- Node N-API shim over libzkcensus.so (the thin C ABI in include/zkcensus.h).
//
// Keeps the snarkjs surface the reference imports at ts_inputs/src/example.ts:1 and calls at :358-362:
//     groth16.fullProve(input, wasmFile, zkeyFile) -> Promise<{proof, publicSignals}>
//     groth16.prove(zkeyFile, wtnsFile), groth16.verify(vk, publicSignals, proof), wtns.calculate(input, wasmFile, wtnsFile)
// The JS wrapper (index.js) flattens the input object, reads artifact files and formats decimal strings; this file only moves buffers
// across the ABI.  The event loop is never blocked: fullProve / prove hand their request to the library's proving service (zkc_service_*,
// csrc/zkc_service.hip) from the main thread and return a promise at once; the service coalesces whatever is pending -- Promise.all over 64
// fullProve calls is a handful of pipeline passes, not 64 serialised proofs -- and its completion callback comes back to JavaScript through a
// thread-safe function.  (libuv's pool has four threads: blocking one per proof, as round 2 did, capped a burst at four requests in flight.)
// wtns.calculate and fullProveBatch run in napi_create_async_work behind one mutex.  libzkcensus.so is dlopen'ed at first use
// (path: $ZKCENSUS_LIB or next to the package), so the addon itself builds with plain g++ and N-API >= 4 headers:
//     g++ -std=c++17 -shared -fPIC -I/usr/include/node napi/zkcensus_napi.cc -o napi/zkcensus.node -ldl
#include <node_api.h>
#include <dlfcn.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct Api {
    void* h = nullptr;
    int (*ctx_create)(int, void**) = nullptr; const char* (*last_error)(const void*) = nullptr;
    int (*n_wires)(int) = nullptr; int (*n_inputs)(int) = nullptr;
    int (*witness)(void*, int, const void*, int, void*, int32_t*) = nullptr;
    int (*zkey_load)(void*, const void*, size_t, void**) = nullptr; void (*zkey_free)(void*) = nullptr;
    int (*zkey_info)(const void*, uint32_t*, uint32_t*, uint32_t*) = nullptr;
    int (*prove)(void*, const void*, uint32_t, const uint8_t*, const uint8_t*, uint8_t*, uint8_t*) = nullptr;
    int (*verify)(const char*, const char*, const char*) = nullptr; const char* (*verify_err)() = nullptr;
    int (*wtns_parse)(const void*, unsigned long, const uint8_t**, uint32_t*) = nullptr;
    unsigned long (*wtns_write)(const void*, uint32_t, void*, unsigned long) = nullptr;
    int (*from_wasm)(const void*, size_t, char*) = nullptr; int (*fingerprint)(const void*, size_t, uint8_t*) = nullptr;
    void (*random_scalars)(uint8_t*, size_t) = nullptr;
    int (*pool_create)(const int*, int, void**) = nullptr; void (*pool_destroy)(void*) = nullptr; const char* (*pool_err)(const void*) = nullptr;
    int (*pool_zkey_load)(void*, const void*, size_t) = nullptr; void* (*pool_zkey)(void*, int) = nullptr;
    int (*pool_fullprove)(void*, const void*, int, const uint8_t*, uint8_t*, uint8_t*, int32_t*) = nullptr;
    typedef void (*done_fn)(void*, int, int32_t, const char*);
    void* (*service_default)() = nullptr; const char* (*service_err)() = nullptr;
    int (*submit_fullprove)(void*, const void*, size_t, int, const void*, const uint8_t*, uint8_t*, uint8_t*, done_fn, void*) = nullptr;
    int (*submit_prove)(void*, const void*, size_t, const void*, uint32_t, const uint8_t*, uint8_t*, uint8_t*, done_fn, void*) = nullptr;
    int (*header_info)(const void*, size_t, uint32_t*, uint32_t*, uint32_t*) = nullptr;
    const char* (*status_text)(int, int32_t) = nullptr;
    int (*inputs_from_json)(const char*, size_t, int, void*, char*, size_t) = nullptr;
    std::string err;
} g;
std::mutex g_mu;                       // guards everything below and every use of the shared context / key
void* g_ctx = nullptr;                 // wtns.calculate only: the proving paths go through the library's service
void* g_pool = nullptr; std::vector<int> g_pool_devs; uint8_t g_pool_sha[32]; bool g_pool_has_key = false;      // fullProveBatch: one context + key per listed device

bool load_api(const std::string& hint) {               // caller holds g_mu
    if (g.h) return true;
    const char* env = getenv("ZKCENSUS_LIB");
    std::string path = env ? env : hint;
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!h) { g.err = std::string("cannot load libzkcensus.so: ") + dlerror(); return false; }
#define SYM(field, name) *(void**)(&g.field) = dlsym(h, name); if (!g.field) { g.err = "missing symbol " name; dlclose(h); return false; }
    SYM(ctx_create, "zkc_ctx_create") SYM(last_error, "zkc_last_error") SYM(n_wires, "zkc_circuit_n_wires") SYM(n_inputs, "zkc_circuit_n_inputs")
    SYM(witness, "zkc_witness") SYM(zkey_load, "zkc_zkey_load") SYM(zkey_free, "zkc_zkey_free") SYM(zkey_info, "zkc_zkey_info") SYM(prove, "zkc_prove")
    SYM(verify, "zkc_verify") SYM(verify_err, "zkc_verify_last_error") SYM(wtns_parse, "zkc_wtns_parse") SYM(wtns_write, "zkc_wtns_write")
    SYM(from_wasm, "zkc_circuit_nlevels_from_wasm") SYM(fingerprint, "zkc_zkey_fingerprint") SYM(random_scalars, "zkc_random_scalars")
    SYM(pool_create, "zkc_pool_create") SYM(pool_destroy, "zkc_pool_destroy") SYM(pool_err, "zkc_pool_last_error") SYM(pool_zkey_load, "zkc_pool_zkey_load")
    SYM(pool_zkey, "zkc_pool_zkey") SYM(pool_fullprove, "zkc_pool_fullprove_batch")
    SYM(service_default, "zkc_service_default") SYM(service_err, "zkc_service_last_error") SYM(submit_fullprove, "zkc_service_submit_fullprove")
    SYM(submit_prove, "zkc_service_submit_prove") SYM(header_info, "zkc_zkey_header_info") SYM(status_text, "zkc_witness_status_text") SYM(inputs_from_json, "zkc_inputs_from_json")
#undef SYM
    g.h = h;
    return true;
}
bool ensure_ctx(std::string& err) {                     // caller holds g_mu
    if (g_ctx) return true;
    const char* d = getenv("ZKC_DEVICE");
    if (g.ctx_create(d ? atoi(d) : 0, &g_ctx)) { g_ctx = nullptr; err = g.last_error(nullptr); return false; }
    return true;
}
// the message snarkjs's Error carries for a failed assert ("Assert Failed.\nError in template ... line: N\n"), from the library (zkc_witness_status_text)
std::string assert_text(int nLevels, int32_t status) {
    const char* t = g.status_text(nLevels, status);
    return t ? std::string(t) : "Assert Failed.\n(witness status " + std::to_string(status) + ")";
}

enum Kind { FULLPROVE, PROVE, WITNESS, BATCH };
struct Work {
    napi_async_work work = nullptr; napi_deferred deferred = nullptr; Kind kind = FULLPROVE;
    std::vector<uint8_t> inputs, zkey, wtns_file, r, s, proof, pub, out; int nLevels = 160; std::string err; std::string libhint;
    std::vector<int> devices; std::vector<uint8_t> rs; std::vector<int32_t> status;      // BATCH
    uint8_t rs64[64]; bool has_rs = false;                                                // FULLPROVE / PROVE through the service
    // the .zkey image is NOT copied (tens of MB per call): the JS Buffer is pinned by a reference until complete() and read in place by the worker
    napi_ref zkey_ref = nullptr; const uint8_t* zkey_p = nullptr; size_t zkey_n = 0;
};
// inputs -> witness payload (nw x 32 B); caller holds g_mu
bool run_witness(Work* w, std::vector<uint8_t>& wtns) {
    const int nw = g.n_wires(w->nLevels), ni = g.n_inputs(w->nLevels);
    if (nw <= 0 || (int)w->inputs.size() != ni * 32) { w->err = "Not all inputs have been set"; return false; }
    wtns.resize((size_t)nw * 32); int32_t status = 0;
    const int rc = g.witness(g_ctx, w->nLevels, w->inputs.data(), 1, wtns.data(), &status);
    if (rc) { w->err = status ? assert_text(w->nLevels, status) : std::string(g.last_error(g_ctx)); return false; }
    return true;
}
// B voters over the listed devices (zkc_pool_*): one context, resident key and host thread per device; caller holds g_mu
bool run_batch(Work* w) {
    const int ni = g.n_inputs(w->nLevels);
    if (ni <= 0 || w->inputs.empty() || w->inputs.size() % ((size_t)ni * 32)) { w->err = "Not all inputs have been set"; return false; }
    const int B = (int)(w->inputs.size() / ((size_t)ni * 32));
    if (!w->rs.empty() && w->rs.size() != (size_t)B * 64) { w->err = "rs must hold 64 bytes (r || s) per voter"; return false; }
    if (w->devices.empty()) w->devices.push_back(0);
    uint8_t d[32]; if (!w->zkey_p || g.fingerprint(w->zkey_p, w->zkey_n, d)) { w->err = "not a zkey file"; return false; }
    if (g_pool && g_pool_devs != w->devices) { g.pool_destroy(g_pool); g_pool = nullptr; g_pool_has_key = false; }
    if (!g_pool) {
        if (g.pool_create(w->devices.data(), (int)w->devices.size(), &g_pool)) { g_pool = nullptr; w->err = g.pool_err(nullptr); return false; }
        g_pool_devs = w->devices;
    }
    if (!g_pool_has_key || memcmp(d, g_pool_sha, 32)) {
        g_pool_has_key = false;
        if (g.pool_zkey_load(g_pool, w->zkey_p, w->zkey_n)) { w->err = g.pool_err(g_pool); return false; }
        memcpy(g_pool_sha, d, 32); g_pool_has_key = true;
    }
    uint32_t nv, np, dn; g.zkey_info(g.pool_zkey(g_pool, 0), &nv, &np, &dn);
    w->proof.resize(256 * (size_t)B); w->pub.resize(32 * (size_t)np * B); w->status.assign((size_t)B, 0);
    const int rc = g.pool_fullprove(g_pool, w->inputs.data(), B, w->rs.empty() ? nullptr : w->rs.data(), w->proof.data(), w->pub.data(), w->status.data());
    if (rc && rc != 7 /* ZKC_ERR_WITNESS: per-voter status says which */) { w->err = g.pool_err(g_pool); return false; }
    return true;
}
void execute(napi_env, void* data) {
    Work* w = (Work*)data;
    std::lock_guard<std::mutex> guard(g_mu);
    if (!load_api(w->libhint)) { w->err = g.err; return; }
    if (w->kind == BATCH) { run_batch(w); return; }
    if (!ensure_ctx(w->err)) return;
    std::vector<uint8_t> wtns;                                        // WITNESS: inputs -> .wtns file image
    if (!run_witness(w, wtns)) return;
    const uint32_t nw = (uint32_t)(wtns.size() / 32);
    w->out.resize(g.wtns_write(wtns.data(), nw, nullptr, 0));
    g.wtns_write(wtns.data(), nw, w->out.data(), w->out.size());
}
void settle(napi_env env, Work* w) {                                  // main thread: resolve or reject the promise of w, release what it pinned
    if (!w->err.empty()) {
        napi_value msg, e; napi_create_string_utf8(env, w->err.c_str(), NAPI_AUTO_LENGTH, &msg); napi_create_error(env, nullptr, msg, &e);
        napi_reject_deferred(env, w->deferred, e);
    } else if (w->kind == WITNESS) {
        napi_value b; void* dst; napi_create_buffer_copy(env, w->out.size(), w->out.data(), &dst, &b);
        napi_resolve_deferred(env, w->deferred, b);
    } else if (w->kind == BATCH) {
        napi_value obj, p, q, st; void* dst;
        napi_create_object(env, &obj);
        napi_create_buffer_copy(env, w->proof.size(), w->proof.data(), &dst, &p); napi_create_buffer_copy(env, w->pub.size(), w->pub.data(), &dst, &q);
        napi_create_buffer_copy(env, w->status.size() * 4, w->status.data(), &dst, &st);
        napi_set_named_property(env, obj, "proofs", p); napi_set_named_property(env, obj, "publicSignals", q); napi_set_named_property(env, obj, "status", st);
        napi_resolve_deferred(env, w->deferred, obj);
    } else {
        napi_value obj, p, q; void* dst;
        napi_create_object(env, &obj);
        napi_create_buffer_copy(env, w->proof.size(), w->proof.data(), &dst, &p); napi_create_buffer_copy(env, w->pub.size(), w->pub.data(), &dst, &q);
        napi_set_named_property(env, obj, "proof", p); napi_set_named_property(env, obj, "publicSignals", q);
        napi_resolve_deferred(env, w->deferred, obj);
    }
    if (w->zkey_ref) napi_delete_reference(env, w->zkey_ref);
    if (w->work) napi_delete_async_work(env, w->work);
    delete w;
}
void complete(napi_env env, napi_status, void* data) { settle(env, (Work*)data); }

// ---- fullProve / prove through the proving service: submitted from the main thread, completed on a service thread, settled back on the main thread ----
napi_threadsafe_function g_tsfn = nullptr; int g_inflight = 0;        // g_inflight: main thread only
void on_done(void* user, int rc, int32_t status, const char* text) {  // service thread
    Work* w = (Work*)user;
    if (rc == 7 /* ZKC_ERR_WITNESS */) w->err = assert_text(w->nLevels, status);
    else if (rc) w->err = text && *text ? text : "proving failed";
    napi_call_threadsafe_function(g_tsfn, w, napi_tsfn_blocking);
}
void settle_js(napi_env env, napi_value, void*, void* data) {         // main thread
    if (env) settle(env, (Work*)data);
    if (--g_inflight == 0 && env) napi_unref_threadsafe_function(env, g_tsfn);      // nothing pending: do not keep the event loop alive
}
napi_value submit(napi_env env, Work* w, const uint8_t* payload, uint32_t nw) {
    napi_value promise; napi_create_promise(env, &w->deferred, &promise);
    { std::lock_guard<std::mutex> guard(g_mu); if (!load_api(w->libhint)) w->err = g.err; }
    uint32_t nv = 0, np = 0, dn = 0; void* svc = nullptr;
    if (w->err.empty() && (!w->zkey_p || g.header_info(w->zkey_p, w->zkey_n, &nv, &np, &dn))) w->err = "not a zkey file";
    if (w->err.empty() && w->kind == FULLPROVE && (int)w->inputs.size() != g.n_inputs(w->nLevels) * 32) w->err = "Not all inputs have been set";
    if (w->err.empty() && !(svc = g.service_default())) w->err = g.service_err();
    if (w->err.empty()) {
        w->proof.resize(256); w->pub.resize(32 * (size_t)np);
        if (w->r.size() == 32 && w->s.size() == 32) { memcpy(w->rs64, w->r.data(), 32); memcpy(w->rs64 + 32, w->s.data(), 32); w->has_rs = true; }
        if (g_inflight++ == 0) napi_ref_threadsafe_function(env, g_tsfn);
        const int rc = w->kind == FULLPROVE
            ? g.submit_fullprove(svc, w->zkey_p, w->zkey_n, w->nLevels, w->inputs.data(), w->has_rs ? w->rs64 : nullptr, w->proof.data(), w->pub.data(), on_done, w)
            : g.submit_prove(svc, w->zkey_p, w->zkey_n, payload, nw, w->has_rs ? w->rs64 : nullptr, w->proof.data(), w->pub.data(), on_done, w);
        if (rc == 0) return promise;
        w->err = g.service_err();
        if (--g_inflight == 0) napi_unref_threadsafe_function(env, g_tsfn);
    }
    settle(env, w);                                                   // rejected before anything was queued
    return promise;
}
std::vector<uint8_t> buf_arg(napi_env env, napi_value v) {
    bool isb = false; napi_is_buffer(env, v, &isb); if (!isb) return {};
    void* p; size_t n; napi_get_buffer_info(env, v, &p, &n); return std::vector<uint8_t>((uint8_t*)p, (uint8_t*)p + n);
}
void pin_zkey(napi_env env, napi_value v, Work* w) {
    bool isb = false; napi_is_buffer(env, v, &isb); if (!isb) return;
    void* p; size_t n; napi_get_buffer_info(env, v, &p, &n);
    if (napi_create_reference(env, v, 1, &w->zkey_ref) != napi_ok) { w->zkey_ref = nullptr; w->zkey.assign((uint8_t*)p, (uint8_t*)p + n); p = w->zkey.data(); }
    w->zkey_p = (const uint8_t*)p; w->zkey_n = n;
}
std::string str_arg(napi_env env, napi_value v) { size_t n = 0; napi_get_value_string_utf8(env, v, nullptr, 0, &n); std::string s(n, 0); napi_get_value_string_utf8(env, v, &s[0], n + 1, &n); return s; }
napi_value queue(napi_env env, Work* w, const char* what) {
    napi_value promise, name; napi_create_promise(env, &w->deferred, &promise);
    napi_create_string_utf8(env, what, NAPI_AUTO_LENGTH, &name);
    napi_create_async_work(env, nullptr, name, execute, complete, w, &w->work); napi_queue_async_work(env, w->work);
    return promise;
}
// fullProveRaw(flatInputs: Buffer, nLevels, zkey: Buffer, r: Buffer(32)|null, s: Buffer(32)|null, libPath) -> Promise<{proof, publicSignals}>
napi_value FullProveRaw(napi_env env, napi_callback_info info) {
    size_t argc = 6; napi_value a[6]; napi_get_cb_info(env, info, &argc, a, nullptr, nullptr);
    Work* w = new Work(); w->kind = FULLPROVE;
    w->inputs = buf_arg(env, a[0]); napi_get_value_int32(env, a[1], &w->nLevels); pin_zkey(env, a[2], w); w->r = buf_arg(env, a[3]); w->s = buf_arg(env, a[4]);
    w->libhint = str_arg(env, a[5]);
    return submit(env, w, nullptr, 0);
}
// fullProveBatchRaw(flatInputs: Buffer (B x nInputs x 32), nLevels, zkey: Buffer, devices: Buffer (int32 LE each), rs: Buffer (B x 64)|null, libPath)
//   -> Promise<{proofs: Buffer (B x 256), publicSignals: Buffer (B x nPublic x 32), status: Buffer (B x int32 LE, ZKC_W_*)}>
napi_value FullProveBatchRaw(napi_env env, napi_callback_info info) {
    size_t argc = 6; napi_value a[6]; napi_get_cb_info(env, info, &argc, a, nullptr, nullptr);
    Work* w = new Work(); w->kind = BATCH;
    w->inputs = buf_arg(env, a[0]); napi_get_value_int32(env, a[1], &w->nLevels); pin_zkey(env, a[2], w);
    const std::vector<uint8_t> dv = buf_arg(env, a[3]);
    for (size_t i = 0; i + 4 <= dv.size(); i += 4) { int32_t x; memcpy(&x, dv.data() + i, 4); w->devices.push_back(x); }
    w->rs = buf_arg(env, a[4]); w->libhint = str_arg(env, a[5]);
    return queue(env, w, "zkcensus.fullProveBatch");
}
// proveRaw(zkey: Buffer, wtnsFileImage: Buffer, r|null, s|null, libPath) -> Promise<{proof, publicSignals}>          (snarkjs groth16.prove)
napi_value ProveRaw(napi_env env, napi_callback_info info) {
    size_t argc = 5; napi_value a[5]; napi_get_cb_info(env, info, &argc, a, nullptr, nullptr);
    Work* w = new Work(); w->kind = PROVE;
    pin_zkey(env, a[0], w); w->wtns_file = buf_arg(env, a[1]); w->r = buf_arg(env, a[2]); w->s = buf_arg(env, a[3]); w->libhint = str_arg(env, a[4]);
    const uint8_t* payload = nullptr; uint32_t nw = 0;
    { std::lock_guard<std::mutex> guard(g_mu); if (!load_api(w->libhint)) w->err = g.err; }
    if (w->err.empty() && g.wtns_parse(w->wtns_file.data(), w->wtns_file.size(), &payload, &nw)) w->err = "Invalid witness file";
    return submit(env, w, payload, nw);
}
// witnessRaw(flatInputs: Buffer, nLevels, libPath) -> Promise<Buffer>  (.wtns file image; snarkjs wtns.calculate)
napi_value WitnessRaw(napi_env env, napi_callback_info info) {
    size_t argc = 3; napi_value a[3]; napi_get_cb_info(env, info, &argc, a, nullptr, nullptr);
    Work* w = new Work(); w->kind = WITNESS;
    w->inputs = buf_arg(env, a[0]); napi_get_value_int32(env, a[1], &w->nLevels); w->libhint = str_arg(env, a[2]);
    return queue(env, w, "zkcensus.wtns.calculate");
}
// circuitFromWasm(wasm: Buffer, libPath) -> {nLevels: number (-1 = unknown circuit), sha256: string}     (host only)
napi_value CircuitFromWasm(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value a[2]; napi_get_cb_info(env, info, &argc, a, nullptr, nullptr);
    std::lock_guard<std::mutex> guard(g_mu);
    if (!load_api(str_arg(env, a[1]))) { napi_throw_error(env, nullptr, g.err.c_str()); return nullptr; }
    void* p; size_t n; napi_get_buffer_info(env, a[0], &p, &n);
    char hex[65] = {0}; const int nl = g.from_wasm(p, n, hex);
    napi_value obj, v, s; napi_create_object(env, &obj); napi_create_int32(env, nl, &v); napi_create_string_utf8(env, hex, 64, &s);
    napi_set_named_property(env, obj, "nLevels", v); napi_set_named_property(env, obj, "sha256", s);
    return obj;
}
// zkeyInfo(zkey: Buffer, libPath) -> {nVars, nPublic, domainSize, nLevels}: from the file header alone (host only); nLevels = the depth n for which the key has the shape of
// ZkFranchiseProofCircuit(n) (8 public signals, zkc_circuit_n_wires(n) wires), -1 for any other circuit
napi_value ZkeyInfo(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value a[2]; napi_get_cb_info(env, info, &argc, a, nullptr, nullptr);
    std::lock_guard<std::mutex> guard(g_mu);
    if (!load_api(str_arg(env, a[1]))) { napi_throw_error(env, nullptr, g.err.c_str()); return nullptr; }
    void* p; size_t n; napi_get_buffer_info(env, a[0], &p, &n);
    uint32_t nv = 0, np = 0, dom = 0;
    if (g.header_info(p, n, &nv, &np, &dom)) { napi_throw_error(env, nullptr, "not a Groth16 .zkey file"); return nullptr; }
    int nl = -1;
    if (np == 8) for (int k = 3; k <= 253; k++) if ((uint32_t)g.n_wires(k) == nv) { nl = k; break; }
    napi_value obj, v; napi_create_object(env, &obj);
    napi_create_uint32(env, nv, &v); napi_set_named_property(env, obj, "nVars", v); napi_create_uint32(env, np, &v); napi_set_named_property(env, obj, "nPublic", v);
    napi_create_uint32(env, dom, &v); napi_set_named_property(env, obj, "domainSize", v); napi_create_int32(env, nl, &v); napi_set_named_property(env, obj, "nLevels", v);
    return obj;
}
// statusText(nLevels, status, libPath) -> string: the Error.message of the reference for a per-voter witness status (fullProveBatch builds its Error objects from it)
napi_value StatusText(napi_env env, napi_callback_info info) {
    size_t argc = 3; napi_value a[3]; napi_get_cb_info(env, info, &argc, a, nullptr, nullptr);
    std::lock_guard<std::mutex> guard(g_mu);
    if (!load_api(str_arg(env, a[2]))) { napi_throw_error(env, nullptr, g.err.c_str()); return nullptr; }
    int32_t nl = 0, st = 0; napi_get_value_int32(env, a[0], &nl); napi_get_value_int32(env, a[1], &st);
    const std::string t = assert_text(nl, st);
    napi_value out; napi_create_string_utf8(env, t.c_str(), t.size(), &out); return out;
}
// flattenJson(inputsJson: string, nLevels, libPath) -> Buffer: the 12-key input object, as JSON text, -> the flat block of the C ABI (zkc_inputs_from_json: the ONE reading of the
// reference's inputs schema -- key order, zero padding of sibling lists, reduction mod r, circom_runtime's messages -- shared with the cgo and Python hosts)
napi_value FlattenJson(napi_env env, napi_callback_info info) {
    size_t argc = 3; napi_value a[3]; napi_get_cb_info(env, info, &argc, a, nullptr, nullptr);
    { std::lock_guard<std::mutex> guard(g_mu); if (!load_api(str_arg(env, a[2]))) { napi_throw_error(env, nullptr, g.err.c_str()); return nullptr; } }
    int32_t nl = 0; napi_get_value_int32(env, a[1], &nl);
    const int ni = g.n_inputs(nl);
    if (ni <= 0) { napi_throw_error(env, nullptr, "bad nLevels"); return nullptr; }
    const std::string js = str_arg(env, a[0]);
    void* data = nullptr; napi_value buf; napi_create_buffer(env, (size_t)ni * 32, &data, &buf);
    char err[256] = {0};
    if (g.inputs_from_json(js.data(), js.size(), nl, data, err, sizeof err)) { napi_throw_error(env, nullptr, err); return nullptr; }
    return buf;
}
// decimals(buf: Buffer) -> string[]: every 32-byte little-endian word of buf as a decimal string (what snarkjs' proof.json / public.json hold).  In C++ because
// BigInt("0x..").toString() costs Node 4 us per value -- 16 values per proof, on the main thread, after the GPU is done: a tenth of a 256-voter burst
napi_value Decimals(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value a[1]; napi_get_cb_info(env, info, &argc, a, nullptr, nullptr);
    void* p = nullptr; size_t n = 0; napi_get_buffer_info(env, a[0], &p, &n);
    const size_t words = n / 32;
    napi_value arr; napi_create_array_with_length(env, words, &arr);
    for (size_t w = 0; w < words; w++) {
        uint64_t v[4]; memcpy(v, (const uint8_t*)p + 32 * w, 32);
        char digits[80]; int len = 0;
        uint64_t chunk[5]; int nchunk = 0;                       // base 10^19 digits, least significant first
        while (v[0] | v[1] | v[2] | v[3]) {
            unsigned __int128 rem = 0;
            for (int i = 3; i >= 0; i--) { const unsigned __int128 cur = (rem << 64) | v[i]; v[i] = (uint64_t)(cur / 10000000000000000000ull); rem = cur % 10000000000000000000ull; }
            chunk[nchunk++] = (uint64_t)rem;
        }
        if (nchunk == 0) { digits[len++] = '0'; }
        else {
            len += snprintf(digits + len, sizeof digits - len, "%llu", (unsigned long long)chunk[nchunk - 1]);
            for (int i = nchunk - 2; i >= 0; i--) len += snprintf(digits + len, sizeof digits - len, "%019llu", (unsigned long long)chunk[i]);
        }
        napi_value str; napi_create_string_utf8(env, digits, (size_t)len, &str); napi_set_element(env, arr, (uint32_t)w, str);
    }
    return arr;
}
// verifyJson(vkeyJson, publicJson, proofJson, libPath) -> boolean   (CPU pairing check, milliseconds)
napi_value VerifyJson(napi_env env, napi_callback_info info) {
    size_t argc = 4; napi_value a[4]; napi_get_cb_info(env, info, &argc, a, nullptr, nullptr);
    { std::lock_guard<std::mutex> guard(g_mu); if (!load_api(str_arg(env, a[3]))) { napi_throw_error(env, nullptr, g.err.c_str()); return nullptr; } }
    int rc = g.verify(str_arg(env, a[0]).c_str(), str_arg(env, a[1]).c_str(), str_arg(env, a[2]).c_str());
    if (rc < 0) { napi_throw_error(env, nullptr, g.verify_err()); return nullptr; }
    napi_value out; napi_get_boolean(env, rc == 1, &out); return out;
}
napi_value Init(napi_env env, napi_value exports) {
    napi_value f, name;
    napi_create_string_utf8(env, "zkcensus.settle", NAPI_AUTO_LENGTH, &name);
    napi_create_threadsafe_function(env, nullptr, nullptr, name, 0, 1, nullptr, nullptr, nullptr, settle_js, &g_tsfn);
    napi_unref_threadsafe_function(env, g_tsfn);
#define EXPORT(name, fn) napi_create_function(env, name, NAPI_AUTO_LENGTH, fn, nullptr, &f); napi_set_named_property(env, exports, name, f);
    EXPORT("flattenJson", FlattenJson) EXPORT("fullProveRaw", FullProveRaw) EXPORT("fullProveBatchRaw", FullProveBatchRaw) EXPORT("proveRaw", ProveRaw) EXPORT("witnessRaw", WitnessRaw) EXPORT("circuitFromWasm", CircuitFromWasm) EXPORT("verifyJson", VerifyJson) EXPORT("statusText", StatusText) EXPORT("zkeyInfo", ZkeyInfo) EXPORT("decimals", Decimals)
#undef EXPORT
    return exports;
}
}  // namespace
NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
