// zkcensus_napi.cc -- Node N-API shim over libzkcensus.so (the thin C ABI in include/zkcensus.h).
//
// Keeps the snarkjs surface the reference calls at ts_inputs/src/example.ts:358-362:
//     groth16.fullProve(input, wasmFile, zkeyFile) -> Promise<{proof, publicSignals}>
// The JS wrapper (index.js) flattens the input object and formats decimal strings; this file only moves buffers across the
// ABI.  All GPU work runs in napi_create_async_work so the event loop is never blocked.  libzkcensus.so is dlopen'ed at
// first use (path: $ZKCENSUS_LIB or next to the package), so the addon itself builds with plain g++ and N-API >= 4 headers:
//     g++ -std=c++17 -shared -fPIC -I/usr/include/node napi/zkcensus_napi.cc -o napi/zkcensus.node -ldl
#include <node_api.h>
#include <dlfcn.h>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace {
struct Api {
    void* h = nullptr;
    int (*ctx_create)(int, void**) = nullptr; const char* (*last_error)(const void*) = nullptr;
    int (*n_wires)(int) = nullptr; int (*n_inputs)(int) = nullptr;
    int (*witness)(void*, int, const void*, int, void*, int32_t*) = nullptr;
    int (*zkey_load)(void*, const void*, size_t, void**) = nullptr;
    int (*zkey_info)(const void*, uint32_t*, uint32_t*, uint32_t*) = nullptr;
    int (*prove)(void*, const void*, uint32_t, const uint8_t*, const uint8_t*, uint8_t*, uint8_t*) = nullptr;
    int (*verify)(const char*, const char*, const char*) = nullptr; const char* (*verify_err)() = nullptr;
    void* ctx = nullptr; std::string err;
} g;

bool load_api(const std::string& hint) {
    if (g.h) return true;
    const char* env = getenv("ZKCENSUS_LIB");
    std::string path = env ? env : hint;
    g.h = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!g.h) { g.err = std::string("cannot load libzkcensus.so: ") + dlerror(); return false; }
#define SYM(field, name) *(void**)(&g.field) = dlsym(g.h, name); if (!g.field) { g.err = "missing symbol " name; return false; }
    SYM(ctx_create, "zkc_ctx_create") SYM(last_error, "zkc_last_error") SYM(n_wires, "zkc_circuit_n_wires") SYM(n_inputs, "zkc_circuit_n_inputs")
    SYM(witness, "zkc_witness") SYM(zkey_load, "zkc_zkey_load") SYM(zkey_info, "zkc_zkey_info") SYM(prove, "zkc_prove")
    SYM(verify, "zkc_verify") SYM(verify_err, "zkc_verify_last_error")
#undef SYM
    return true;
}

struct Work {
    napi_async_work work = nullptr; napi_deferred deferred = nullptr;
    std::vector<uint8_t> inputs, zkey, r, s, proof, pub; int nLevels = 160; int32_t status = 0; std::string err; std::string libhint;
    void* key = nullptr;
};
void* g_key = nullptr; std::vector<uint8_t> g_key_bytes;       // the last key stays resident, like the Python surface

void execute(napi_env, void* data) {
    Work* w = (Work*)data;
    if (!load_api(w->libhint)) { w->err = g.err; return; }
    if (!g.ctx) { const char* d = getenv("ZKC_DEVICE"); if (g.ctx_create(d ? atoi(d) : 0, &g.ctx)) { g.ctx = nullptr; w->err = g.last_error(nullptr); return; } }
    if (!g_key || g_key_bytes != w->zkey) {
        if (g.zkey_load(g.ctx, w->zkey.data(), w->zkey.size(), &g_key)) { g_key = nullptr; w->err = g.last_error(g.ctx); return; }
        g_key_bytes = w->zkey;
    }
    const int nw = g.n_wires(w->nLevels), ni = g.n_inputs(w->nLevels);
    if ((int)w->inputs.size() != ni * 32) { w->err = "Not all inputs have been set"; return; }
    std::vector<uint8_t> wtns((size_t)nw * 32);
    int rc = g.witness(g.ctx, w->nLevels, w->inputs.data(), 1, wtns.data(), &w->status);
    if (rc) { w->err = w->status ? "Error: Assert Failed. circuit assert " + std::to_string(w->status) : std::string(g.last_error(g.ctx)); return; }
    uint32_t nv, np, dn; g.zkey_info(g_key, &nv, &np, &dn);
    w->proof.resize(256); w->pub.resize(32 * (size_t)np);
    rc = g.prove(g_key, wtns.data(), (uint32_t)nw, w->r.data(), w->s.data(), w->proof.data(), w->pub.data());
    if (rc) w->err = g.last_error(g.ctx);
}
void complete(napi_env env, napi_status, void* data) {
    Work* w = (Work*)data;
    if (!w->err.empty()) {
        napi_value msg, e; napi_create_string_utf8(env, w->err.c_str(), NAPI_AUTO_LENGTH, &msg); napi_create_error(env, nullptr, msg, &e);
        napi_reject_deferred(env, w->deferred, e);
    } else {
        napi_value obj, p, q; void* dst;
        napi_create_object(env, &obj);
        napi_create_buffer_copy(env, w->proof.size(), w->proof.data(), &dst, &p); napi_create_buffer_copy(env, w->pub.size(), w->pub.data(), &dst, &q);
        napi_set_named_property(env, obj, "proof", p); napi_set_named_property(env, obj, "publicSignals", q);
        napi_resolve_deferred(env, w->deferred, obj);
    }
    napi_delete_async_work(env, w->work); delete w;
}
std::vector<uint8_t> buf_arg(napi_env env, napi_value v) { void* p; size_t n; napi_get_buffer_info(env, v, &p, &n); return std::vector<uint8_t>((uint8_t*)p, (uint8_t*)p + n); }
std::string str_arg(napi_env env, napi_value v) { size_t n; napi_get_value_string_utf8(env, v, nullptr, 0, &n); std::string s(n, 0); napi_get_value_string_utf8(env, v, &s[0], n + 1, &n); return s; }

// fullProveRaw(flatInputs: Buffer, nLevels: number, zkey: Buffer, r: Buffer(32), s: Buffer(32), libPath: string) -> Promise<{proof, publicSignals}>
napi_value FullProveRaw(napi_env env, napi_callback_info info) {
    size_t argc = 6; napi_value a[6]; napi_get_cb_info(env, info, &argc, a, nullptr, nullptr);
    Work* w = new Work();
    w->inputs = buf_arg(env, a[0]); napi_get_value_int32(env, a[1], &w->nLevels); w->zkey = buf_arg(env, a[2]); w->r = buf_arg(env, a[3]); w->s = buf_arg(env, a[4]);
    w->libhint = str_arg(env, a[5]);
    napi_value promise, name; napi_create_promise(env, &w->deferred, &promise);
    napi_create_string_utf8(env, "zkcensus.fullProve", NAPI_AUTO_LENGTH, &name);
    napi_create_async_work(env, nullptr, name, execute, complete, w, &w->work); napi_queue_async_work(env, w->work);
    return promise;
}
// verifyJson(vkeyJson, publicJson, proofJson, libPath) -> boolean   (CPU pairing check, milliseconds)
napi_value VerifyJson(napi_env env, napi_callback_info info) {
    size_t argc = 4; napi_value a[4]; napi_get_cb_info(env, info, &argc, a, nullptr, nullptr);
    if (!load_api(str_arg(env, a[3]))) { napi_throw_error(env, nullptr, g.err.c_str()); return nullptr; }
    int rc = g.verify(str_arg(env, a[0]).c_str(), str_arg(env, a[1]).c_str(), str_arg(env, a[2]).c_str());
    if (rc < 0) { napi_throw_error(env, nullptr, g.verify_err()); return nullptr; }
    napi_value out; napi_get_boolean(env, rc == 1, &out); return out;
}
napi_value Init(napi_env env, napi_value exports) {
    napi_value f; napi_create_function(env, "fullProveRaw", NAPI_AUTO_LENGTH, FullProveRaw, nullptr, &f); napi_set_named_property(env, exports, "fullProveRaw", f);
    napi_create_function(env, "verifyJson", NAPI_AUTO_LENGTH, VerifyJson, nullptr, &f); napi_set_named_property(env, exports, "verifyJson", f);
    return exports;
}
}  // namespace
NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
