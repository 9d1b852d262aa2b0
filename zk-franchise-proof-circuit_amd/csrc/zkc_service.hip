// zkc_service.hip -- the reference's own call shape made fast: concurrent SINGLE-proof callers coalesced into pipeline passes.
//
// The reference proves one voter per call: prover.Prove(zkey, wasm, inputs) in a loop or from goroutines (zk_census_test.go:89, reaching
// rapidsnark's groth16_prover through cgo) and groth16.fullProve(inputs, wasm, zkey) per ballot (ts_inputs/src/example.ts:358-362).  A GPU pipeline
// pass proves up to 96 voters in the time two single proofs take, so behind those entry points sits a submission queue: callers enqueue
// (inputs | witness, r, s) and one pair of worker threads per GPU drains whatever has accumulated -- group commit, no timer: a lone caller is
// served at once with a batch of one, callers that arrive while the GPU is busy share its next pass -- into one zkc_fullprove_batch_dev /
// zkc_prove_batch_dev call, and every caller gets its own proof, status and error back.  Two workers per device: while one holds the GPU the
// other collects and uploads the next batch and tops it up until the GPU is free.  Devices: an explicit list, $ZKC_DEVICE ("2", "0,1,2,3", "all")
// or, unset, every visible device -- but a device is only brought up (context + 1 GB of key tables, 0.6 s) when the queue is long enough to pay for
// it, so a sequential caller stays on the first one; and a device that is brought up loads its key from the service's OWN copy of the image BEFORE it takes
// any request, so nobody waits behind a key load that a warm device could have served meanwhile.
// [r4] A device keeps SEVERAL keys resident (least recently used out first, $ZKC_SERVICE_KEYS of them, default 4: the reference has a key per environment and per depth,
// circuit/circuit-compiler.sh:15,82, and 288 GB hold dozens): requests for different keys alternate on one GPU without a reload, each batch still of one key.  A key is freed
// only by a worker that holds the device's GPU lock, after it has left the routing table and its last call in flight has finished.
// Key identity: the service keeps a copy of every .zkey image it has seen (at most four), found per call through the sampled fingerprint and confirmed, the
// first time a given caller buffer (pointer, length) shows up, by the SHA-256 of the whole image: two images that differ only in unsampled bytes are two keys.
// Host code over the public batch entry points; launches no kernel of its own.
#include "zkc_prover.h"
#include "zkc_hostparse.h"
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

namespace {
enum { KIND_FULLPROVE = 0, KIND_PROVE = 1 };
struct KeyImage {                                                  // the service's own copy of a .zkey image: outlives the caller's buffer, shared by all devices
    uint8_t fp[32], sha[32]; std::vector<uint8_t> bytes;
    std::set<std::pair<const void*, size_t>> confirmed;            // caller buffers whose full SHA-256 equalled sha (under zkc_service::img_mu)
    uint64_t last_use = 0;
};
struct Req {
    int kind; std::shared_ptr<KeyImage> img; int nLevels; const uint8_t* data; uint32_t nW;
    uint8_t rs[64]; uint8_t* proof; uint8_t* pub;
    zkc_done_fn done; void* user;
    bool same_class(const Req& o) const { return kind == o.kind && nLevels == o.nLevels && img.get() == o.img.get(); }
};
struct Waiter { std::mutex m; std::condition_variable cv; bool finished = false; int rc = 0; int32_t status = 0; std::string err; };
void waiter_done(void* u, int rc, int32_t status, const char* err) {
    Waiter* w = (Waiter*)u; std::lock_guard<std::mutex> g(w->m);
    w->rc = rc; w->status = status; w->err = err ? err : ""; w->finished = true; w->cv.notify_one();
}
struct HipBuf {     // grow-only buffer on the calling thread's current device (or pinned host memory)
    void* p = nullptr; size_t sz = 0; bool host = false;
    bool ensure(size_t want) {
        if (sz >= want) return true;
        release();
        const hipError_t e = host ? hipHostMalloc(&p, want) : hipMalloc(&p, want);
        if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; return false; }
        sz = want; return true;
    }
    void release() { if (p) { if (host) (void)hipHostFree(p); else (void)hipFree(p); } p = nullptr; sz = 0; }
};
}  // namespace

struct zkc_service {
    struct KeySlot { std::shared_ptr<KeyImage> img; zkc_zkey* key = nullptr; int in_flight = 0; uint64_t last_use = 0; };      // in_flight: split calls begun on this key and not finished yet
    struct Dev {
        int device = 0; std::mutex gpu_mu;              // held while a worker owns the GPU pipeline of this device (key load / eviction + the begin half of a batch call)
        zkc_ctx* ctx = nullptr;
        // under zkc_service::mu -- the routing table: the images whose keys are resident here (entered after a successful load, removed BEFORE the key is freed) and the one a
        // worker of this device is loading right now
        std::vector<std::shared_ptr<KeyImage>> resident; std::shared_ptr<KeyImage> loading;
        bool has(const KeyImage* img) const { for (auto& r : resident) if (r.get() == img) return true; return false; }
        uint64_t batches = 0, proofs = 0;
        // under fl_mu -- the keys themselves.  The vector changes only under gpu_mu AND fl_mu; a slot is erased (and its key freed) only when its in_flight is zero, and
        // in_flight rises only under gpu_mu: whoever reads a slot's key under fl_mu with in_flight > 0, or under gpu_mu, reads a live key.
        std::mutex fl_mu; std::condition_variable fl_cv; std::vector<KeySlot> keys; uint64_t use_clock = 0;
        int calls_in_flight() const { int n = 0; for (auto& k : keys) n += k.in_flight; return n; }
    };
    struct Worker {
        Dev* dev = nullptr; int index = 0; std::thread th; std::condition_variable cv; bool wake = false, idle = false;
        std::shared_ptr<KeyImage> warm;                  // set by dispatch: bring this device up for that key before taking requests
        HipBuf h_in{nullptr, 0, true}, h_wtns{nullptr, 0, true}, d_in, d_wtns, d_status, h_proofs{nullptr, 0, true}, h_pubs{nullptr, 0, true}, h_status{nullptr, 0, true};
        hipStream_t st = nullptr; size_t cap = 0;        // requests the staging buffers hold (grows geometrically with the batches this worker has seen)
    };
    std::mutex mu; std::deque<Req*> q; bool stop = false;
    std::mutex img_mu; std::vector<std::shared_ptr<KeyImage>> images; uint64_t img_clock = 0;      // key images by (fingerprint, SHA-256); at most four kept
    std::vector<std::unique_ptr<Dev>> devs; std::vector<std::unique_ptr<Worker>> workers;
    int max_batch = 256, spill = 32, keys_per_dev = 4;
    uint64_t n_requests = 0, n_batches = 0, largest_batch = 0, key_loads = 0, key_evictions = 0, n_failed = 0;
    uint64_t us_stage = 0, us_gpu_wait = 0, us_key = 0, us_prove = 0, us_finish = 0, n_proved = 0;      // where the workers' time went (microseconds, summed over batches)
};
static thread_local std::string g_service_err;

namespace {
void finish(Req* r, int rc, int32_t status, const std::string& err) { r->done(r->user, rc, status, err.c_str()); delete r; }

// ---- dispatch (all under svc->mu) ----
bool any_dev_has(zkc_service* s, const KeyImage* img) { for (auto& d : s->devs) if (d->has(img) || d->loading.get() == img) return true; return false; }
// the requests worker w takes now, up to `cap` of one class in arrival order:
//   its device holds keys             -> the class of the first queued request for any of them (arrival order: two resident keys take turns);
//   else, no device holds or is loading the head's key (the very first request, or a new key)
//                                     -> the head's class: this worker will load the key with those requests in hand (somebody has to);
//   else                              -> nothing: other devices serve that key (if the queue grows past `spill`, dispatch brings this device up FIRST, without requests)
std::vector<Req*> grab(zkc_service* s, zkc_service::Worker* w, size_t cap, const Req* like = nullptr) {
    std::vector<Req*> out;
    if (s->q.empty() || cap == 0) return out;
    zkc_service::Dev* d = w->dev;
    const Req* cls = like;
    if (!cls && !d->resident.empty() && !d->loading) for (Req* r : s->q) if (d->has(r->img.get())) { cls = r; break; }
    if (!cls && !d->loading && !any_dev_has(s, s->q.front()->img.get())) { cls = s->q.front(); d->loading = cls->img; }
    if (!cls) return out;
    const Req key = *cls;
    for (auto it = s->q.begin(); it != s->q.end() && out.size() < cap;) { if ((*it)->same_class(key)) { out.push_back(*it); it = s->q.erase(it); } else ++it; }
    return out;
}
// wake at most one idle worker for the head of the queue: a worker on a device that holds its key; if no device holds or loads it, any idle worker (it will load
// it); if the queue has grown past `spill` requests, an idle worker on a device without that key -- told to load the key first (warm) and only then to take requests
void dispatch(zkc_service* s) {
    if (s->q.empty()) return;
    const Req* head = s->q.front();
    zkc_service::Worker* pick = nullptr;
    for (auto& w : s->workers) if (w->idle && !w->wake && !w->dev->loading && w->dev->has(head->img.get())) { pick = w.get(); break; }
    if (!pick && !any_dev_has(s, head->img.get())) {
        // the device with the fewest resident keys first (a cold one, then one with room, then one that has to evict)
        for (auto& w : s->workers) if (w->idle && !w->wake && !w->dev->loading && (!pick || w->dev->resident.size() < pick->dev->resident.size())) pick = w.get();
    }
    if (!pick && s->q.size() >= (size_t)s->spill) {
        for (auto& w : s->workers)
            if (w->idle && !w->wake && !w->dev->loading && !w->dev->has(head->img.get())) {
                bool sibling_busy = false;                                  // a FULL device whose other worker is in a call keeps its keys: do not pull one out from under it
                for (auto& o : s->workers) if (o.get() != w.get() && o->dev == w->dev && !o->idle) sibling_busy = true;
                if (sibling_busy && (int)w->dev->resident.size() >= s->keys_per_dev) continue;
                pick = w.get(); pick->warm = head->img; pick->dev->loading = head->img; break;
            }
    }
    if (pick) { pick->wake = true; pick->cv.notify_one(); }
}

// ---- one batch on one device ----
struct Batch { std::vector<Req*> reqs; std::vector<uint8_t> rs; };
// uploads requests [from, reqs.size()) of the batch into the worker's device buffers; false = HIP failure (text in err)
bool stage(zkc_service::Worker* w, Batch& b, size_t from, size_t nIn, size_t nW, std::string& err) {
    const size_t B = b.reqs.size();
    if (from >= B) return true;
    const bool full = b.reqs[0]->kind == KIND_FULLPROVE;
    if (full) {
        uint8_t* h = (uint8_t*)w->h_in.p;
        for (size_t i = from; i < B; i++) memcpy(h + i * nIn * 32, b.reqs[i]->data, nIn * 32);
        if (hipMemcpyAsync((uint8_t*)w->d_in.p + from * nIn * 32, h + from * nIn * 32, (B - from) * nIn * 32, hipMemcpyHostToDevice, w->st) != hipSuccess) { err = "hipMemcpyAsync of the inputs"; return false; }
    } else {
        // witnesses arrive in the callers' pageable memory (2.6 MB each at nLevels = 160).  hipMemcpyAsync from there moved 1.8 GB/s; copied first into this worker's
        // pinned staging (beside the other worker's GPU batch) they go up in one transfer at PCIe speed
        uint8_t* h = (uint8_t*)w->h_wtns.p;
        for (size_t i = from; i < B; i++) memcpy(h + i * nW * 32, b.reqs[i]->data, nW * 32);
        if (hipMemcpyAsync((uint8_t*)w->d_wtns.p + from * nW * 32, h + from * nW * 32, (B - from) * nW * 32, hipMemcpyHostToDevice, w->st) != hipSuccess) { err = "hipMemcpyAsync of the witnesses"; return false; }
    }
    if (hipStreamSynchronize(w->st) != hipSuccess) { err = "hipStreamSynchronize after the uploads"; return false; }
    return true;
}
void fail_all(zkc_service* s, Batch& b, int rc, const std::string& err) {
    { std::lock_guard<std::mutex> g(s->mu); s->n_failed += b.reqs.size(); }
    for (Req* r : b.reqs) finish(r, rc, 0, err);
    b.reqs.clear();
}
// makes `img` a resident key of device d and returns it in *out.  Caller holds d->gpu_mu.  A full device gives up its least recently used key: the key leaves the routing
// table first (no new batch is formed for it), then its calls in flight are waited for, then it is freed.
int ensure_key(zkc_service* s, zkc_service::Dev* d, const std::shared_ptr<KeyImage>& img, zkc_zkey** out, std::string& why) {
    int rc = ZKC_OK; *out = nullptr;
    if (!d->ctx && (rc = zkc_ctx_create(d->device, &d->ctx))) { d->ctx = nullptr; why = std::string("device ") + std::to_string(d->device) + ": " + zkc_last_error(nullptr); }
    if (!rc) {
        std::lock_guard<std::mutex> fl(d->fl_mu);
        for (auto& k : d->keys) if (k.img.get() == img.get()) { k.last_use = ++d->use_clock; *out = k.key; }
    }
    if (!rc && !*out) {
        for (;;) {                                                             // make room
            std::shared_ptr<KeyImage> victim;
            {
                std::lock_guard<std::mutex> fl(d->fl_mu);
                if ((int)d->keys.size() < s->keys_per_dev) break;
                size_t v = 0; for (size_t i = 1; i < d->keys.size(); i++) if (d->keys[i].last_use < d->keys[v].last_use) v = i;
                victim = d->keys[v].img;
            }
            { std::lock_guard<std::mutex> g(s->mu); for (size_t i = 0; i < d->resident.size(); i++) if (d->resident[i].get() == victim.get()) { d->resident.erase(d->resident.begin() + (long)i); break; } }
            zkc_zkey* dead = nullptr;
            {
                std::unique_lock<std::mutex> fl(d->fl_mu);
                auto slot = [&]() -> zkc_service::KeySlot* { for (auto& k : d->keys) if (k.img.get() == victim.get()) return &k; return nullptr; };
                d->fl_cv.wait(fl, [&] { zkc_service::KeySlot* k = slot(); return !k || k->in_flight == 0; });      // the other worker's call still reads that key
                for (size_t i = 0; i < d->keys.size(); i++) if (d->keys[i].img.get() == victim.get()) { dead = d->keys[i].key; d->keys.erase(d->keys.begin() + (long)i); break; }
            }
            if (dead) { zkc_zkey_free(dead); std::lock_guard<std::mutex> g(s->mu); s->key_evictions++; }
        }
        zkc_zkey* key = nullptr;
        rc = zkc_zkey_load(d->ctx, img->bytes.data(), img->bytes.size(), &key);
        if (rc) { key = nullptr; why = zkc_last_error(d->ctx); }
        // [r4] the work space of a full pass at once: a key behind the service is there for concurrent callers, and growing from four proofs to a pass in the middle of the
        // first burst stalled the device for ~0.5 s (free + re-allocation of ~6 GB at nLevels 160).  ZKC_SERVICE_RESERVE=0: grow on demand as the direct entry points do.
        static const bool reserve = [] { const char* e = getenv("ZKC_SERVICE_RESERVE"); return !(e && atoi(e) == 0); }();
        if (!rc && reserve && zkc::prove_reserve(key, 1 << 20) != ZKC_OK) (void)zkc_last_error(d->ctx);      // not fatal: the first large call will try again and report
        if (!rc) { std::lock_guard<std::mutex> fl(d->fl_mu); zkc_service::KeySlot k; k.img = img; k.key = key; k.last_use = ++d->use_clock; d->keys.push_back(k); }
        std::lock_guard<std::mutex> g(s->mu); s->key_loads++;
        if (!rc) { d->resident.push_back(img); *out = key; }
    }
    { std::lock_guard<std::mutex> g(s->mu); if (d->loading.get() == img.get()) d->loading.reset(); }
    return rc;
}
void process(zkc_service* s, zkc_service::Worker* w, std::vector<Req*>&& first) {
    zkc_service::Dev* d = w->dev;
    Batch b; b.reqs = std::move(first);
    const Req cls = *b.reqs[0];
    struct LoadingGuard { zkc_service* s; zkc_service::Dev* d; const KeyImage* img; ~LoadingGuard() { std::lock_guard<std::mutex> g(s->mu); if (d->loading.get() == img) d->loading.reset(); } } lguard{s, d, cls.img.get()};
    const bool full = cls.kind == KIND_FULLPROVE;
    std::string err;
    if (hipSetDevice(d->device) != hipSuccess) { (void)hipGetLastError(); return fail_all(s, b, ZKC_ERR_HIP, "hipSetDevice failed"); }
    // shapes from the file header alone (the key may not be resident yet)
    zkc::parse::BinSections bs; zkc::parse::ZkeyHeader zh;
    if (!zkc::parse::binfile_sections(cls.img->bytes.data(), cls.img->bytes.size(), "zkey", 1, bs, err) || !zkc::parse::zkey_check(bs, zh, err, false)) return fail_all(s, b, ZKC_ERR_FORMAT, err);
    const size_t nW = zh.nVars, nPub = zh.nPub, nIn = full ? (size_t)zkc_circuit_n_inputs(cls.nLevels) : 0;
    if (full && (nIn == 0 || (size_t)zkc_circuit_n_wires(cls.nLevels) != nW)) return fail_all(s, b, ZKC_ERR_BAD_ARG, "the key is not a ZkFranchiseProofCircuit(" + std::to_string(cls.nLevels) + ") key");
    if (!full) {                                            // a witness of the wrong length fails alone, like rapidsnark's INVALID_WITNESS_LENGTH
        std::vector<Req*> keep;
        for (Req* r : b.reqs) { if (r->nW == nW) keep.push_back(r); else finish(r, ZKC_ERR_INVALID_WITNESS_LENGTH, 0, "Invalid witness length. Circuit: " + std::to_string(nW) + ", witness: " + std::to_string(r->nW)); }
        b.reqs.swap(keep);
        if (b.reqs.empty()) return;
    }
    // staging: a lone sequential caller reserves room for 8 voters; the first batch of more than one caller sizes it for good (max_batch voters for the
    // inputs path -- 11 KB each, plus 2.6 MB of device memory per witness -- and at most 128 for the witness path, whose pinned staging is 2.6 MB per request too)
    const size_t kind_max = full ? (size_t)s->max_batch : std::min<size_t>((size_t)s->max_batch, 128);
    w->cap = std::min(kind_max, std::max(w->cap, b.reqs.size() > 1 ? kind_max : (size_t)8));
    if (b.reqs.size() > w->cap) {                            // the other kind sized this worker before: give the tail back to the queue
        std::lock_guard<std::mutex> g(s->mu);
        while (b.reqs.size() > w->cap) { s->q.push_front(b.reqs.back()); b.reqs.pop_back(); }
        dispatch(s);
    }
    const size_t cap = w->cap;
    if (!w->st && hipStreamCreateWithFlags(&w->st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); w->st = nullptr; return fail_all(s, b, ZKC_ERR_HIP, "hipStreamCreate failed"); }
    if ((full && (!w->h_in.ensure(cap * nIn * 32) || !w->d_in.ensure(cap * nIn * 32))) || !w->d_wtns.ensure(cap * nW * 32) || (!full && !w->h_wtns.ensure(cap * nW * 32)) || !w->d_status.ensure(cap * 4) ||
        !w->h_proofs.ensure(cap * 256) || !w->h_pubs.ensure(cap * nPub * 32 + 32) || !w->h_status.ensure(cap * 4))
        return fail_all(s, b, ZKC_ERR_HIP, "out of memory for the service's staging buffers on device " + std::to_string(d->device));
    auto now_us = [] { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const uint64_t t_a = now_us();
    if (!stage(w, b, 0, nIn, nW, err)) { (void)hipGetLastError(); return fail_all(s, b, ZKC_ERR_HIP, err); }      // beside the other worker's batch, which owns the GPU now
    int rc = ZKC_OK; zkc_zkey* zk = nullptr; int slot = 0;
    uint64_t t_b = now_us(), t_c = 0, t_d = 0, t_e = 0;
    // while the other worker's call is in its body (accumulations still running) there is nothing to gain from beginning: this call's kernels would only queue
    // behind it.  Keep collecting requests instead and begin when that call reaches its tail (bucket reduction, blinding, copies: 4-5 ms of latency chains that
    // this call's witness kernels and transforms run beside).  [r4] The calls in flight may be on any of the device's keys; each key is asked UNDER fl_mu while its
    // in_flight is positive -- an eviction waits, under the same mutex, for that count to reach zero before it frees the key, so the query never sees a freed key
    // (round 3 read d->key after dropping the lock: a use-after-free window when the other worker switched keys in between).
    for (;;) {
        bool body = false;
        { std::lock_guard<std::mutex> fl(d->fl_mu); for (auto& k : d->keys) if (k.in_flight > 0 && !zkc::prove_tail_reached(k.key)) body = true; }
        if (!body) break;
        const size_t from = b.reqs.size();
        if (from < cap) {
            std::lock_guard<std::mutex> g(s->mu);
            std::vector<Req*> more = grab(s, w, cap - from, &cls);
            for (Req* r : more) if (full || r->nW == nW) b.reqs.push_back(r); else finish(r, ZKC_ERR_INVALID_WITNESS_LENGTH, 0, "Invalid witness length");
        }
        if (b.reqs.size() > from) { if (!stage(w, b, from, nIn, nW, err)) { (void)hipGetLastError(); return fail_all(s, b, ZKC_ERR_HIP, err); } }
        else std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    // an idle device and a burst still arriving (Promise.all over a census: one request every ~25 us from one thread): beginning with the first handful would spend a
    // latency-bound call on them while the rest queue up behind it.  While requests keep coming, keep taking them: stop after three polls of 50 us without a new one, at `cap`,
    // or after 3 ms.  A lone request with nothing behind it does not wait at all.
    {
        bool idle_dev; { std::lock_guard<std::mutex> fl(d->fl_mu); idle_dev = d->calls_in_flight() == 0; }
        const uint64_t t_l0 = now_us(); int quiet = 0; bool first = true;
        while (idle_dev && b.reqs.size() < cap && quiet < 3 && now_us() - t_l0 < 3000) {
            const size_t from = b.reqs.size();
            { std::lock_guard<std::mutex> g(s->mu); std::vector<Req*> more = grab(s, w, cap - from, &cls); for (Req* r : more) if (full || r->nW == nW) b.reqs.push_back(r); else finish(r, ZKC_ERR_INVALID_WITNESS_LENGTH, 0, "Invalid witness length"); }
            if (b.reqs.size() > from) { if (!stage(w, b, from, nIn, nW, err)) { (void)hipGetLastError(); return fail_all(s, b, ZKC_ERR_HIP, err); } quiet = 0; }
            else { if (first && from == 1) break; std::this_thread::sleep_for(std::chrono::microseconds(50)); quiet++; }
            first = false;
        }
    }
    {
        std::lock_guard<std::mutex> gpu(d->gpu_mu);
        t_c = now_us();
        // whoever queued up meanwhile for the same key rides along
        size_t from = b.reqs.size();
        { std::lock_guard<std::mutex> g(s->mu); std::vector<Req*> more = grab(s, w, cap - b.reqs.size(), &cls); for (Req* r : more) if (full || r->nW == nW) b.reqs.push_back(r); else finish(r, ZKC_ERR_INVALID_WITNESS_LENGTH, 0, "Invalid witness length"); }
        if (!stage(w, b, from, nIn, nW, err)) { (void)hipGetLastError(); return fail_all(s, b, ZKC_ERR_HIP, err); }
        const int B = (int)b.reqs.size();
        // resident key: the service's own image object is the identity (fingerprint + full SHA-256 were settled when the request was accepted)
        { std::string why; if ((rc = ensure_key(s, d, cls.img, &zk, why))) return fail_all(s, b, rc, why); }
        if (full && zk->nLevels != cls.nLevels) return fail_all(s, b, ZKC_ERR_BAD_ARG, "the key is not a ZkFranchiseProofCircuit(" + std::to_string(cls.nLevels) + ") key");
        b.rs.resize((size_t)B * 64);
        for (int i = 0; i < B; i++) memcpy(b.rs.data() + 64 * (size_t)i, b.reqs[i]->rs, 64);
        t_d = now_us();
        // the call in two halves: begin enqueues every pass and returns (its witness kernels run beside the other worker's MSMs); the GPU lock is given up
        // before finish waits, so that the other worker's next begin overlaps this call's bucket reduction, blinding and copies
        slot = w->index & 1;
        rc = zkc::prove_batch_begin(zk, slot, w->d_wtns.p, (uint32_t)nW, B, b.rs.data(), true, full ? w->d_in.p : nullptr, full ? (int32_t*)w->d_status.p : nullptr);
        if (rc) return fail_all(s, b, rc, zkc_last_error(d->ctx));
        { std::lock_guard<std::mutex> fl(d->fl_mu); for (auto& k : d->keys) if (k.key == zk) k.in_flight++; }      // still under gpu_mu: the slot is there
    }
    {
        const int B = (int)b.reqs.size();
        rc = zkc::prove_batch_finish(zk, slot, (uint8_t*)w->h_proofs.p, (uint8_t*)w->h_pubs.p);
        if (!rc && full && (hipMemcpyAsync(w->h_status.p, w->d_status.p, (size_t)B * 4, hipMemcpyDeviceToHost, w->st) != hipSuccess || hipStreamSynchronize(w->st) != hipSuccess)) { (void)hipGetLastError(); rc = ZKC_ERR_HIP; }
        const std::string why = rc ? (rc == ZKC_ERR_HIP ? std::string("HIP failure while the batch finished: ") : std::string()) + zkc_last_error(d->ctx) : std::string();
        { std::lock_guard<std::mutex> fl(d->fl_mu); for (auto& k : d->keys) if (k.key == zk) k.in_flight--; } d->fl_cv.notify_all();
        if (rc) return fail_all(s, b, rc, why);
    }
    const size_t B = b.reqs.size();
    t_e = now_us();
    { std::lock_guard<std::mutex> g(s->mu); s->n_batches++; s->largest_batch = std::max<uint64_t>(s->largest_batch, B); d->batches++; d->proofs += B; s->n_proved += B;
      s->us_stage += t_b - t_a; s->us_gpu_wait += t_c - t_b; s->us_key += t_d - t_c; s->us_prove += t_e - t_d; }
    for (size_t i = 0; i < B; i++) {
        Req* r = b.reqs[i];
        const int32_t st = full ? ((const int32_t*)w->h_status.p)[i] : 0;
        if (st == ZKC_W_OK) { memcpy(r->proof, (const uint8_t*)w->h_proofs.p + 256 * i, 256); if (r->pub) memcpy(r->pub, (const uint8_t*)w->h_pubs.p + nPub * 32 * i, nPub * 32); }
        const char* why = st == ZKC_W_OK ? "" : zkc_witness_status_text(r->nLevels, st);
        finish(r, st == ZKC_W_OK ? ZKC_OK : ZKC_ERR_WITNESS, st, why ? why : "a circuit assert failed (see status)");
    }
    { std::lock_guard<std::mutex> g(s->mu); s->us_finish += now_us() - t_e; }
}
void worker_main(zkc_service* s, zkc_service::Worker* w) {
    std::unique_lock<std::mutex> lk(s->mu);
    for (;;) {
        std::vector<Req*> batch;
        while (!s->stop && !w->warm && (batch = grab(s, w, (size_t)s->max_batch)).empty()) {
            w->idle = true; w->cv.wait(lk, [&] { return w->wake || s->stop; }); w->wake = false; w->idle = false;
        }
        if (!s->stop && w->warm) {                           // bring this device up for that key first; the queue is being served by the devices that have it
            std::shared_ptr<KeyImage> img = std::move(w->warm); w->warm.reset();
            lk.unlock();
            if (hipSetDevice(w->dev->device) == hipSuccess) { std::lock_guard<std::mutex> gpu(w->dev->gpu_mu); std::string why; zkc_zkey* k = nullptr; (void)ensure_key(s, w->dev, img, &k, why); }
            else { (void)hipGetLastError(); std::lock_guard<std::mutex> g(s->mu); if (w->dev->loading.get() == img.get()) w->dev->loading.reset(); }
            lk.lock();
            continue;
        }
        if (s->stop) { for (Req* r : batch) s->q.push_front(r); break; }
        dispatch(s);                                        // what is left in the queue may be another worker's
        lk.unlock();
        process(s, w, std::move(batch));
        lk.lock();
    }
    lk.unlock();
    (void)hipSetDevice(w->dev->device);
    for (HipBuf* hb : {&w->h_in, &w->h_wtns, &w->d_in, &w->d_wtns, &w->d_status, &w->h_proofs, &w->h_pubs, &w->h_status}) hb->release();
    if (w->st) (void)hipStreamDestroy(w->st);
}
int service_fail(int code, const std::string& msg) { g_service_err = msg; return code; }
bool parse_device_list(const char* e, std::vector<int>& out) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); count = 0; }
    if (!e || !*e || !strcmp(e, "all")) { for (int i = 0; i < count; i++) out.push_back(i); return !out.empty(); }
    for (const char* p = e; *p;) { char* end; const long v = strtol(p, &end, 10); if (end == p || v < 0) return false; out.push_back((int)v); p = *end == ',' ? end + 1 : end; if (*end && *end != ',') return false; }
    return !out.empty();
}
}  // namespace

extern "C" const char* zkc_service_last_error(void) { return g_service_err.c_str(); }
extern "C" int zkc_service_create(const int* hip_devices, int n, zkc_service** out) {
    if (!out || n < 0 || n > 64 || (n > 0 && !hip_devices)) return service_fail(ZKC_ERR_BAD_ARG, "zkc_service_create: bad argument");
    std::vector<int> devs(hip_devices, hip_devices + n);
    if (n == 0 && !parse_device_list(getenv("ZKC_DEVICE"), devs)) return service_fail(ZKC_ERR_HIP, "zkc_service_create: no GPU visible (or $ZKC_DEVICE is not a list of device numbers)");
    zkc_service* s = new zkc_service();
    if (const char* e = getenv("ZKC_SERVICE_MAX_BATCH")) s->max_batch = std::max(1, std::min(atoi(e), 4096));
    if (const char* e = getenv("ZKC_SERVICE_SPILL")) s->spill = std::max(1, atoi(e));
    if (const char* e = getenv("ZKC_SERVICE_KEYS")) s->keys_per_dev = std::max(1, std::min(atoi(e), 64));
    for (int dv : devs) { s->devs.emplace_back(new zkc_service::Dev()); s->devs.back()->device = dv; }
    int idx = 0;
    for (auto& d : s->devs) for (int k = 0; k < 2; k++) { s->workers.emplace_back(new zkc_service::Worker()); s->workers.back()->dev = d.get(); s->workers.back()->index = idx++; }
    for (auto& w : s->workers) w->th = std::thread(worker_main, s, w.get());
    *out = s; return ZKC_OK;
}
extern "C" void zkc_service_destroy(zkc_service* s) {
    if (!s) return;
    { std::lock_guard<std::mutex> g(s->mu); s->stop = true; for (auto& w : s->workers) w->cv.notify_all(); }
    for (auto& w : s->workers) if (w->th.joinable()) w->th.join();
    for (Req* r : s->q) finish(r, ZKC_ERR_GENERIC, 0, "the proving service was shut down");
    s->q.clear();
    for (auto& d : s->devs) { (void)hipSetDevice(d->device); for (auto& k : d->keys) if (k.key) zkc_zkey_free(k.key); if (d->ctx) zkc_ctx_destroy(d->ctx); }
    delete s;
}
extern "C" zkc_service* zkc_service_default(void) {
    static std::mutex mu; static zkc_service* svc = nullptr;          // lives as long as the process: never destroyed, its idle workers sleep on a condition variable
    static std::string why;
    std::lock_guard<std::mutex> g(mu);
    if (!svc && zkc_service_create(nullptr, 0, &svc)) { svc = nullptr; why = g_service_err; }
    if (!svc) g_service_err = why;                                    // every thread that asks gets the reason, not only the first
    return svc;
}
// the service's image object for a caller's .zkey buffer: found by the sampled fingerprint, confirmed -- once per (buffer pointer, length) -- by the SHA-256 of
// the whole image, copied the first time it is seen.  At most four images are kept (least recently used out first; requests in flight keep theirs alive).
static std::shared_ptr<KeyImage> image_of(zkc_service* s, const void* zkey, size_t zkey_len, std::string& why) {
    uint8_t fp[32];
    if (zkc_zkey_fingerprint(zkey, zkey_len, fp)) { why = "not a zkey file"; return nullptr; }
    std::lock_guard<std::mutex> g(s->img_mu);
    const std::pair<const void*, size_t> ident(zkey, zkey_len);
    bool sha_known = false; uint8_t sha[32];
    for (auto& im : s->images) {
        if (memcmp(im->fp, fp, 32) || im->bytes.size() != zkey_len) continue;
        if (!im->confirmed.count(ident)) {
            if (!sha_known) { zkc::parse::sha256(zkey, zkey_len, sha); sha_known = true; }
            if (memcmp(sha, im->sha, 32)) continue;                          // same sampled bytes, different image: another key
            if (im->confirmed.size() > 256) im->confirmed.clear();
            im->confirmed.insert(ident);
        }
        im->last_use = ++s->img_clock;
        return im;
    }
    auto im = std::make_shared<KeyImage>();
    memcpy(im->fp, fp, 32);
    if (!sha_known) zkc::parse::sha256(zkey, zkey_len, sha);
    memcpy(im->sha, sha, 32);
    im->bytes.assign((const uint8_t*)zkey, (const uint8_t*)zkey + zkey_len);
    im->confirmed.insert(ident); im->last_use = ++s->img_clock;
    if (s->images.size() >= 4) {
        size_t lru = 0; for (size_t i = 1; i < s->images.size(); i++) if (s->images[i]->last_use < s->images[lru]->last_use) lru = i;
        s->images.erase(s->images.begin() + (long)lru);
    }
    s->images.push_back(im);
    return im;
}
static int submit(zkc_service* s, int kind, const void* zkey, size_t zkey_len, int nLevels, const void* data, uint32_t nW, const uint8_t* rs, uint8_t* proof, uint8_t* pub,
                  zkc_done_fn done, void* user) {
    if (!s || !zkey || !data || !proof || !done) return service_fail(ZKC_ERR_BAD_ARG, "zkc_service_submit: bad argument");
    std::string why;
    std::shared_ptr<KeyImage> img = image_of(s, zkey, zkey_len, why);
    if (!img) return service_fail(ZKC_ERR_FORMAT, why);
    Req* r = new Req();
    r->kind = kind; r->img = std::move(img); r->nLevels = kind == KIND_FULLPROVE ? nLevels : 0; r->data = (const uint8_t*)data; r->nW = nW;
    r->proof = proof; r->pub = pub; r->done = done; r->user = user;
    if (rs) {
        for (int k = 0; k < 2; k++) { uint32_t t[8]; memcpy(t, rs + 32 * k, 32); if (!zkc::fp_std_lt_p<zkc::FrParams>(t)) { delete r; return service_fail(ZKC_ERR_BAD_ARG, "r or s >= field order"); } }
        memcpy(r->rs, rs, 64);
    } else zkc_random_scalars(r->rs, 2);
    std::lock_guard<std::mutex> g(s->mu);
    if (s->stop) { delete r; return service_fail(ZKC_ERR_GENERIC, "the proving service was shut down"); }
    s->q.push_back(r); s->n_requests++;
    dispatch(s);
    return ZKC_OK;
}
extern "C" int zkc_service_submit_fullprove(zkc_service* s, const void* zkey, size_t zkey_len, int nLevels, const void* inputs, const uint8_t* rs,
                                            uint8_t proof[256], uint8_t* publics, zkc_done_fn done, void* user) {
    return submit(s, KIND_FULLPROVE, zkey, zkey_len, nLevels, inputs, 0, rs, proof, publics, done, user);
}
extern "C" int zkc_service_submit_prove(zkc_service* s, const void* zkey, size_t zkey_len, const void* wtns, uint32_t nWitness, const uint8_t* rs,
                                        uint8_t proof[256], uint8_t* publics, zkc_done_fn done, void* user) {
    return submit(s, KIND_PROVE, zkey, zkey_len, 0, wtns, nWitness, rs, proof, publics, done, user);
}
static int wait_for(Waiter& w, int32_t* status, char* err, size_t errlen) {
    std::unique_lock<std::mutex> lk(w.m); w.cv.wait(lk, [&] { return w.finished; });
    if (status) *status = w.status;
    if (err && errlen) snprintf(err, errlen, "%s", w.err.c_str());
    if (w.rc) g_service_err = w.err;
    return w.rc;
}
extern "C" int zkc_service_fullprove(zkc_service* s, const void* zkey, size_t zkey_len, int nLevels, const void* inputs, const uint8_t* rs,
                                     uint8_t proof[256], uint8_t* publics, int32_t* status, char* err, size_t errlen) {
    Waiter w; if (status) *status = 0;
    const int rc = zkc_service_submit_fullprove(s, zkey, zkey_len, nLevels, inputs, rs, proof, publics, waiter_done, &w);
    if (rc) { if (err && errlen) snprintf(err, errlen, "%s", g_service_err.c_str()); return rc; }
    return wait_for(w, status, err, errlen);
}
extern "C" int zkc_service_prove(zkc_service* s, const void* zkey, size_t zkey_len, const void* wtns, uint32_t nWitness, const uint8_t* rs,
                                 uint8_t proof[256], uint8_t* publics, char* err, size_t errlen) {
    Waiter w;
    const int rc = zkc_service_submit_prove(s, zkey, zkey_len, wtns, nWitness, rs, proof, publics, waiter_done, &w);
    if (rc) { if (err && errlen) snprintf(err, errlen, "%s", g_service_err.c_str()); return rc; }
    return wait_for(w, nullptr, err, errlen);
}
extern "C" int zkc_service_timing(zkc_service* s, uint64_t out[8]) {
    if (!s || !out) return ZKC_ERR_BAD_ARG;
    std::lock_guard<std::mutex> g(s->mu);
    out[0] = s->us_stage; out[1] = s->us_gpu_wait; out[2] = s->us_key; out[3] = s->us_prove; out[4] = s->us_finish; out[5] = s->n_proved; out[6] = s->n_batches; out[7] = s->key_evictions;
    return ZKC_OK;
}
extern "C" int zkc_service_stats(zkc_service* s, uint64_t out[8]) {
    if (!s || !out) return ZKC_ERR_BAD_ARG;
    std::lock_guard<std::mutex> g(s->mu);
    int used = 0; for (auto& d : s->devs) used += d->batches > 0;
    out[0] = s->n_requests; out[1] = s->n_batches; out[2] = s->largest_batch; out[3] = s->key_loads; out[4] = (uint64_t)s->devs.size(); out[5] = (uint64_t)used;
    out[6] = s->n_failed; out[7] = (uint64_t)s->q.size();
    return ZKC_OK;
}
