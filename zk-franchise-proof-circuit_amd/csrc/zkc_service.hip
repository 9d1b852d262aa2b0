// zkc_service.hip -- the reference's own call shape made fast: concurrent SINGLE-proof callers coalesced into pipeline passes.
//
// The reference proves one voter per call: prover.Prove(zkey, wasm, inputs) in a loop or from goroutines (zk_census_test.go:89, reaching
// rapidsnark's groth16_prover through cgo) and groth16.fullProve(inputs, wasm, zkey) per ballot (ts_inputs/src/example.ts:358-362).  A GPU pipeline
// pass proves 64 voters in the time four single proofs take, so behind those entry points sits a submission queue: callers enqueue (inputs | witness, r, s), the workers of
// a GPU form pipeline passes out of whoever is waiting -- group commit, no timer: a lone caller is served at once with a batch of one, callers that arrive while the GPU is busy
// share a coming pass -- and every caller gets its own proof, status and error back.
// Workers: one per pipeline LANE of the device (default four per GPU; $ZKC_SERVICE_WORKERS, passes of up to $ZKC_SERVICE_PASS = 64 proofs).  Worker k owns call slot k of the key
// and lane k of the device -- streams and work space of its own -- so the calls of concurrent callers are independent on the GPU and overlap in full.  (Round 4: two workers on
// ONE lane, whose calls could only overlap tail on head: 58-71 % of the batch rate from 64 callers.  And lanes only overlap once every busy stream has a hardware queue of its
// own: GPU_MAX_HW_QUEUES, zkc_api.hip.)  One worker per device at a time COLLECTS: it takes its share of what is queued (everything the device's callers have outstanding divided
// by the workers, at least $ZKC_SERVICE_MIN_BATCH = 16), waits -- on a condition variable, woken by every arrival of its class -- while requests keep coming (three quiet 50 us
// waits on an idle device, two quiet 100 us waits or $ZKC_SERVICE_BUSY_WAIT_US = 300 on a busy one), uploads on its lane's stream as it goes, hands the collector role on and
// begins its call.  A one-pass call is laid out from sibling depths read on the HOST (off the inputs, or off the sibling wires of a given witness), so begin is ~0.4 ms of
// enqueueing and never waits for the GPU.  Witnesses (2.6 MB each, the groth16_prover shape) are copied into pinned slots by the CALLERS' threads, side by side.
// Devices: an explicit list, $ZKC_DEVICE ("2", "0,1,2,3", "all") or, unset, every visible device -- but a device is only brought up (context, ~2.5 GB of key tables, 0.6 s,
// the lanes' work space) when the queue is long enough to pay for it, so a sequential caller stays on the first one; and a device that is brought up loads its key from the
// service's OWN copy of the image BEFORE it takes any request, so nobody waits behind a key load that a warm device could have served meanwhile.
// Keys: a device keeps SEVERAL resident (least recently used out first, $ZKC_SERVICE_KEYS of them, default 4: the reference has a key per environment and per depth,
// circuit/circuit-compiler.sh:15,82): requests for different keys alternate on one GPU without a reload, each batch still of one key; the lanes' work space is the device
// context's and shared by them (zkc_prove.hip lane_ensure).  A key is freed only by a worker that holds the device's GPU lock, after it has left the routing table and its
// last call in flight has finished; a key load that fails for want of memory evicts idle keys and tries again.
// Key identity: the service keeps a copy of every .zkey image its devices may hold (never dropping one that is resident or in a request), found per call through the sampled
// fingerprint and confirmed, the first time a given caller buffer (pointer, length) shows up, by the SHA-256 of the whole image: two images that differ only in unsampled
// bytes are two keys.
// Host code over the public batch entry points; launches no kernel of its own.
#include "zkc_prover.h"
#include "zkc_hostparse.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

int zkc_lane_streams(zkc_ctx* ctx, int l, bool with_red, zkc_ctx::LaneStreams* out);      // zkc_api.hip
namespace {
enum { KIND_FULLPROVE = 0, KIND_PROVE = 1 };
struct KeyImage {                                                  // the service's own copy of a .zkey image: outlives the caller's buffer, shared by all devices
    uint8_t fp[32], sha[32]; std::vector<uint8_t> bytes;
    std::set<std::pair<const void*, size_t>> confirmed;            // caller buffers whose full SHA-256 equalled sha (under zkc_service::img_mu)
    uint64_t last_use = 0;
    int shape_nl = -2;                                             // nLevels if the key has the shape of ZkFranchiseProofCircuit(nLevels) (8 public signals, that wire count), -1 if not, -2 not looked at yet
};
struct Req {
    int kind; std::shared_ptr<KeyImage> img; int nLevels; const uint8_t* data; uint32_t nW; void* pin = nullptr;      // pin: the witness already copied into a pinned slot by the caller's thread (PinPool)
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
// pinned host slots for witnesses on their way up, by size class (the witness size rounded up to 256 KB: a process that proves for several circuits -- the census circuit beside
// whatever a caller's own wasm computes -- keeps a free list per class; 2 GB in all, beyond that a witness goes through the worker's own staging).  The copy from the caller's
// pageable buffer -- 2.6 MB at nLevels 160, ~0.3 ms -- is made by the CALLER's thread inside submit, so sixty-four callers copy side by side instead of one worker copying for all
struct PinPool {
    std::mutex m; size_t total = 0, max_total = (size_t)2 << 30; std::map<size_t, std::vector<void*>> free_; std::map<void*, size_t> cls_of; std::vector<void*> chunks;
    static size_t cls(size_t bytes) { return (bytes + 262143) & ~(size_t)262143; }
    void* get(size_t bytes) {
        const size_t sl = cls(bytes);
        {
            std::lock_guard<std::mutex> g(m);
            auto& fl = free_[sl];
            if (!fl.empty()) { void* r = fl.back(); fl.pop_back(); return r; }
            if (total + 2 * sl > max_total) return nullptr;
            total += 2 * sl;                                         // reserved; pinned below, OUTSIDE the lock: a burst of first-time callers pins its slots side by side
        }
        void* p = nullptr;                                           // two slots at a time: this caller's and one for whoever comes next
        if (hipHostMalloc(&p, 2 * sl, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); std::lock_guard<std::mutex> g(m); total -= 2 * sl; return nullptr; }
        std::lock_guard<std::mutex> g(m);
        chunks.push_back(p); cls_of[p] = sl; cls_of[(uint8_t*)p + sl] = sl; free_[sl].push_back((uint8_t*)p + sl);
        return p;
    }
    void put(void* p) { if (!p) return; std::lock_guard<std::mutex> g(m); free_[cls_of[p]].push_back(p); }
    size_t bytes() { std::lock_guard<std::mutex> g(m); return total; }
    void release() { for (void* p : chunks) (void)hipHostFree(p); chunks.clear(); cls_of.clear(); free_.clear(); total = 0; }
};
}  // namespace

struct zkc_service {
    struct KeySlot { std::shared_ptr<KeyImage> img; zkc_zkey* key = nullptr; int in_flight = 0; uint64_t last_use = 0; };      // in_flight: split calls begun on this key and not finished yet
    struct Worker;
    struct Dev {
        int device = 0; std::mutex gpu_mu;              // held while a worker owns the GPU pipeline of this device (key load / eviction + the begin half of a batch call)
        zkc_ctx* ctx = nullptr;
        // under zkc_service::mu -- the routing table: the images whose keys are resident here (entered after a successful load, removed BEFORE the key is freed) and the one a
        // worker of this device is loading right now
        std::vector<std::shared_ptr<KeyImage>> resident; std::shared_ptr<KeyImage> loading;
        bool has(const KeyImage* img) const { for (auto& r : resident) if (r.get() == img) return true; return false; }
        uint64_t batches = 0, proofs = 0;
        std::atomic<int> proofs_in_flight{0};           // voters of the calls begun and not finished on this device
        Worker* collector = nullptr;             // the worker of this device that is gathering a batch right now (at most one; the others sleep or are in their calls)
        // under fl_mu -- the keys themselves.  The vector changes only under gpu_mu AND fl_mu; a slot is erased (and its key freed) only when its in_flight is zero, and
        // in_flight rises only under gpu_mu: whoever reads a slot's key under fl_mu with in_flight > 0, or under gpu_mu, reads a live key.
        std::mutex fl_mu; std::condition_variable fl_cv; std::vector<KeySlot> keys; uint64_t use_clock = 0;
        int calls_in_flight() const { int n = 0; for (auto& k : keys) n += k.in_flight; return n; }
    };
    struct Worker {
        Dev* dev = nullptr; int index = 0, slot = 0; std::thread th; std::condition_variable cv; bool wake = false, idle = false;      // slot: index among the device's workers = its call slot = its lane
        const Req* collecting = nullptr; size_t room = 0;   // set while this worker collects: the class it gathers and how many more requests it takes (dispatch wakes it for those)
        std::shared_ptr<KeyImage> warm;                  // set by dispatch: bring this device up for that key before taking requests
        HipBuf h_in{nullptr, 0, true}, h_wtns{nullptr, 0, true}, d_in, d_wtns, d_status, h_proofs{nullptr, 0, true}, h_pubs{nullptr, 0, true}, h_status{nullptr, 0, true};
        hipStream_t st = nullptr; hipEvent_t ev_up = nullptr; size_t cap = 0;        // ev_up: this batch's uploads are through (the call's first kernels wait for it, not the host); cap: requests the staging buffers hold
    };
    std::mutex mu; std::deque<Req*> q; bool stop = false;
    std::mutex img_mu; std::vector<std::shared_ptr<KeyImage>> images; uint64_t img_clock = 0;      // key images by (fingerprint, SHA-256): as many as the devices may hold keys, none dropped while resident
    PinPool pin;
    std::vector<std::unique_ptr<Dev>> devs; std::vector<std::unique_ptr<Worker>> workers;
    int max_batch = 256, spill = 32, keys_per_dev = 4, workers_per_dev = 4, pass = 64, min_batch = 16; uint64_t busy_wait_us = 300;
    uint64_t n_requests = 0, n_batches = 0, largest_batch = 0, key_loads = 0, key_evictions = 0, n_failed = 0, reserve_failures = 0, oom_evictions = 0;
    uint64_t us_stage = 0, us_gpu_wait = 0, us_key = 0, us_prove = 0, us_finish = 0, n_proved = 0;      // where the workers' time went (microseconds, summed over batches)
};
static thread_local std::string g_service_err;

namespace {
void finish(zkc_service* s, Req* r, int rc, int32_t status, const std::string& err) { void* pin = r->pin; r->done(r->user, rc, status, err.c_str()); delete r; s->pin.put(pin); }

// ---- dispatch (all under svc->mu) ----
bool any_dev_has(zkc_service* s, const KeyImage* img) { for (auto& d : s->devs) if (d->has(img) || d->loading.get() == img) return true; return false; }
// how many requests a batch on device d should take when `have` are in hand already: everything the device's callers have outstanding -- in hand, queued, in the calls in
// flight -- shared out over the device's workers, and at least min_batch.  Sixty-four callers that come back from one call together would otherwise go into ONE call again,
// and a single call in flight leaves its latency-bound phases (witness chains, bucket reductions, blinding) uncovered; four calls of sixteen drift apart and cover one another.
size_t batch_target(zkc_service* s, zkc_service::Dev* d, size_t have) {
    const size_t all = have + s->q.size() + (size_t)std::max(0, d->proofs_in_flight.load());
    return std::max<size_t>((size_t)s->min_batch, (all + (size_t)s->workers_per_dev - 1) / (size_t)s->workers_per_dev);
}
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
// somebody for the head of the queue: the COLLECTOR of a device that holds its key, if it gathers that class and has room (it wakes and takes it); else at most one idle worker on
// a device that holds the key and has no collector; if no device holds or loads the key, any idle worker of a collector-less device (it will load it); if the queue has grown past
// `spill` requests, an idle worker on a device without that key -- told to load the key first (warm) and only then to take requests
void dispatch(zkc_service* s) {
    if (s->q.empty()) return;
    const Req* head = s->q.front();
    for (auto& d : s->devs) if (d->collector && d->collector->collecting && d->collector->room > 0 && d->collector->collecting->same_class(*head)) { d->collector->wake = true; d->collector->cv.notify_one(); return; }
    zkc_service::Worker* pick = nullptr;
    for (auto& w : s->workers) if (w->idle && !w->wake && !w->dev->collector && !w->dev->loading && w->dev->has(head->img.get())) { pick = w.get(); break; }
    if (!pick && !any_dev_has(s, head->img.get())) {
        // the device with the fewest resident keys first (a cold one, then one with room, then one that has to evict)
        for (auto& w : s->workers) if (w->idle && !w->wake && !w->dev->collector && !w->dev->loading && (!pick || w->dev->resident.size() < pick->dev->resident.size())) pick = w.get();
    }
    if (!pick && s->q.size() >= (size_t)s->spill) {
        for (auto& w : s->workers)
            if (w->idle && !w->wake && !w->dev->collector && !w->dev->loading && !w->dev->has(head->img.get())) {
                bool sibling_busy = false;                                  // a FULL device whose other workers are in a call keeps its keys: do not pull one out from under it
                for (auto& o : s->workers) if (o.get() != w.get() && o->dev == w->dev && !o->idle) sibling_busy = true;
                if (sibling_busy && (int)w->dev->resident.size() >= s->keys_per_dev) continue;
                pick = w.get(); pick->warm = head->img; pick->dev->loading = head->img; break;
            }
    }
    if (pick) { pick->wake = true; pick->cv.notify_one(); }
}

// ---- one batch on one device ----
struct Batch { std::vector<Req*> reqs; std::vector<uint8_t> rs, depths; };      // depths: (census, sik) per request, read on the host (prove_batch_begin's host_depths); 255 = unknown
// 1 + index of the last non-zero 32-byte value among `count` (0: all zero)
int last_nonzero(const uint8_t* v, int count) {
    for (int i = count - 1; i >= 0; i--) { const uint64_t* q = (const uint64_t*)(v + 32 * (size_t)i); if (q[0] | q[1] | q[2] | q[3]) return i + 1; }
    return 0;
}
// enqueues the uploads of requests [from, reqs.size()) into the worker's device buffers on its stream (no host wait: the call's first kernels wait for ev_up); false = HIP failure
bool stage(zkc_service::Worker* w, Batch& b, size_t from, size_t nIn, size_t nW, std::string& err) {
    const size_t B = b.reqs.size();
    if (from >= B) return true;
    const bool full = b.reqs[0]->kind == KIND_FULLPROVE;
    if (full) {
        uint8_t* h = (uint8_t*)w->h_in.p;
        for (size_t i = from; i < B; i++) memcpy(h + i * nIn * 32, b.reqs[i]->data, nIn * 32);
        if (hipMemcpyAsync((uint8_t*)w->d_in.p + from * nIn * 32, h + from * nIn * 32, (B - from) * nIn * 32, hipMemcpyHostToDevice, w->st) != hipSuccess) { err = "hipMemcpyAsync of the inputs"; return false; }
    } else {
        // witnesses arrive in the callers' pageable memory (2.6 MB each at nLevels = 160).  hipMemcpyAsync from there moved 1.8 GB/s; they go up from pinned memory at PCIe
        // speed: from the slot the caller's own thread copied it into (submit), or -- pool exhausted, odd size -- through this worker's staging
        for (size_t i = from; i < B; i++) {
            const void* src = b.reqs[i]->pin;
            if (!src) { if (!w->h_wtns.ensure(w->cap * nW * 32)) { err = "out of pinned memory for the witnesses"; return false; } src = (uint8_t*)w->h_wtns.p + i * nW * 32; memcpy((void*)src, b.reqs[i]->data, nW * 32); }
            if (hipMemcpyAsync((uint8_t*)w->d_wtns.p + i * nW * 32, src, nW * 32, hipMemcpyHostToDevice, w->st) != hipSuccess) { err = "hipMemcpyAsync of the witnesses"; return false; }
        }
    }
    return true;
}
void fail_all(zkc_service* s, Batch& b, int rc, const std::string& err) {
    { std::lock_guard<std::mutex> g(s->mu); s->n_failed += b.reqs.size(); }
    for (Req* r : b.reqs) finish(s, r, rc, 0, err);
    b.reqs.clear();
}
// evicts the least recently used key of device d (caller holds d->gpu_mu): out of the routing table first (no new batch is formed for it), then its calls in flight are
// waited for, then it is freed.  false: the device holds no key
bool evict_lru(zkc_service* s, zkc_service::Dev* d) {
    std::shared_ptr<KeyImage> victim;
    {
        std::lock_guard<std::mutex> fl(d->fl_mu);
        if (d->keys.empty()) return false;
        size_t v = 0; for (size_t i = 1; i < d->keys.size(); i++) if (d->keys[i].last_use < d->keys[v].last_use) v = i;
        victim = d->keys[v].img;
    }
    { std::lock_guard<std::mutex> g(s->mu); for (size_t i = 0; i < d->resident.size(); i++) if (d->resident[i].get() == victim.get()) { d->resident.erase(d->resident.begin() + (long)i); break; } }
    zkc_zkey* dead = nullptr;
    {
        std::unique_lock<std::mutex> fl(d->fl_mu);
        auto slot = [&]() -> zkc_service::KeySlot* { for (auto& k : d->keys) if (k.img.get() == victim.get()) return &k; return nullptr; };
        d->fl_cv.wait(fl, [&] { zkc_service::KeySlot* k = slot(); return !k || k->in_flight == 0; });      // another worker's call still reads that key
        for (size_t i = 0; i < d->keys.size(); i++) if (d->keys[i].img.get() == victim.get()) { dead = d->keys[i].key; d->keys.erase(d->keys.begin() + (long)i); break; }
    }
    if (dead) { zkc_zkey_free(dead); std::lock_guard<std::mutex> g(s->mu); s->key_evictions++; }
    return true;
}
// makes `img` a resident key of device d and returns it in *out.  Caller holds d->gpu_mu.  A full device gives up its least recently used key; so does one whose memory does
// not hold another key (a load that fails for want of memory while idle keys are resident evicts and tries again instead of failing the batch).
int ensure_key(zkc_service* s, zkc_service::Dev* d, const std::shared_ptr<KeyImage>& img, zkc_zkey** out, std::string& why) {
    int rc = ZKC_OK; *out = nullptr;
    if (!d->ctx && (rc = zkc_ctx_create(d->device, &d->ctx))) { d->ctx = nullptr; why = std::string("device ") + std::to_string(d->device) + ": " + zkc_last_error(nullptr); }
    if (!rc) {
        std::lock_guard<std::mutex> fl(d->fl_mu);
        for (auto& k : d->keys) if (k.img.get() == img.get()) { k.last_use = ++d->use_clock; *out = k.key; }
    }
    if (!rc && !*out) {
        for (;;) {                                                             // make room
            { std::lock_guard<std::mutex> fl(d->fl_mu); if ((int)d->keys.size() < s->keys_per_dev) break; }
            if (!evict_lru(s, d)) break;
        }
        zkc_zkey* key = nullptr;
        const char* fail_loads = getenv("ZKC_TEST_FAIL_KEY_LOADS");           // test hook: pretend the device is out of memory while it holds this many keys or more
        for (;;) {
            size_t held; { std::lock_guard<std::mutex> fl(d->fl_mu); held = d->keys.size(); }
            if (fail_loads && held >= (size_t)atoi(fail_loads)) { rc = ZKC_ERR_HIP; why = "zkc_zkey_load: out of memory (injected by ZKC_TEST_FAIL_KEY_LOADS)"; }
            else {
                rc = zkc::zkey_load_opts(d->ctx, img->bytes.data(), img->bytes.size(), s->workers_per_dev, s->pass, &key);
                if (rc) { key = nullptr; why = zkc_last_error(d->ctx); }
            }
            if (rc != ZKC_ERR_HIP || held == 0) break;
            (void)hipGetLastError();
            if (!evict_lru(s, d)) break;                                       // an idle key's tables and work space for this one, then once more
            { std::lock_guard<std::mutex> g(s->mu); s->oom_evictions++; }
        }
        // the work space of a full pass on every lane at once: a key behind the service is there for concurrent callers, and growing a lane from four proofs to a pass in the
        // middle of the first burst stalled the device for ~0.5 s (free + re-allocation of GBs).  [r5] The lanes are the CONTEXT's (zkc_prove.hip lane_ensure): the first key of
        // a device pays for them (~37 GB at nLevels 160 with four lanes of 64 proofs), later keys of the same shape find them there; a failed reserve is counted, not fatal
        // (the lanes then grow on demand).  ZKC_SERVICE_RESERVE=0: always on demand, as the direct entry points do.
        static const bool reserve = [] { const char* e = getenv("ZKC_SERVICE_RESERVE"); return !(e && atoi(e) == 0); }();
        if (!rc && reserve && zkc::prove_reserve(key, 1 << 20) != ZKC_OK) { (void)zkc_last_error(d->ctx); std::lock_guard<std::mutex> g(s->mu); s->reserve_failures++; }
        if (!rc) { std::lock_guard<std::mutex> fl(d->fl_mu); zkc_service::KeySlot k; k.img = img; k.key = key; k.last_use = ++d->use_clock; d->keys.push_back(k); }
        std::lock_guard<std::mutex> g(s->mu); s->key_loads++;
        if (!rc) { d->resident.push_back(img); *out = key; }
    }
    { std::lock_guard<std::mutex> g(s->mu); if (d->loading.get() == img.get()) { d->loading.reset(); dispatch(s); } }      // whoever queued up behind the load finds a worker now
    return rc;
}
uint64_t now_us() { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// called with s->mu held by lk, this worker being its device's collector: gives the role up and lets dispatch find somebody for what is queued
void stop_collecting(zkc_service* s, zkc_service::Worker* w) { w->collecting = nullptr; w->room = 0; if (w->dev->collector == w) w->dev->collector = nullptr; dispatch(s); }
struct CollectorGuard { zkc_service* s; zkc_service::Worker* w; ~CollectorGuard() { std::lock_guard<std::mutex> g(s->mu); if (w->dev->collector == w) stop_collecting(s, w); } };

void process(zkc_service* s, zkc_service::Worker* w, std::vector<Req*>&& first) {
    zkc_service::Dev* d = w->dev;
    CollectorGuard cguard{s, w};                               // whatever way this function is left, the device gets its collector role back
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
    // the folding depths of a voter from the host's copy of its inputs (sibling lists at entries 12 .. of the input block) or, for a witness computed elsewhere, from the sibling
    // wires of the witness (wires 13 .. 13 + 2 nLevels: zkc_device.h WitnessLayout) -- what zkc_input_depths reads on the device, without the round trip
    int shape_nl = cls.nLevels;
    if (!full) {
        std::lock_guard<std::mutex> g(s->img_mu);
        if (cls.img->shape_nl == -2) { cls.img->shape_nl = -1; if (nPub == 8) for (int nl = 3; nl <= 253; nl++) if ((size_t)zkc_circuit_n_wires(nl) == nW) { cls.img->shape_nl = nl; break; } }
        shape_nl = cls.img->shape_nl;
    }
    auto depths_of = [&](const Req* r, uint8_t out[2]) {
        out[0] = out[1] = 255;
        if (shape_nl < 3 || shape_nl > 253) return;
        const uint8_t* v = (const uint8_t*)(r->pin ? r->pin : r->data);
        for (int t = 0; t < 2; t++) {
            const int d = full ? last_nonzero(v + 32 * (12 + (size_t)t * (shape_nl + 1)), shape_nl + 1) : last_nonzero(v + 32 * (13 + (size_t)t * shape_nl), shape_nl);
            out[t] = d <= shape_nl ? (uint8_t)d : (uint8_t)255;                // a non-zero LAST sibling fails SMTLevIns: that voter is rejected, the call takes the usual path
        }
    };
    auto keep_right_length = [&](std::vector<Req*>& in) {       // a witness of the wrong length fails alone, like rapidsnark's INVALID_WITNESS_LENGTH
        for (Req* r : in) { if (full || r->nW == nW) { b.reqs.push_back(r); uint8_t dd[2]; depths_of(r, dd); b.depths.push_back(dd[0]); b.depths.push_back(dd[1]); } else finish(s, r, ZKC_ERR_INVALID_WITNESS_LENGTH, 0, "Invalid witness length. Circuit: " + std::to_string(nW) + ", witness: " + std::to_string(r->nW)); }
    };
    { std::vector<Req*> in; in.swap(b.reqs); keep_right_length(in); if (b.reqs.empty()) return; }
    // staging: a lone sequential caller reserves room for 8 voters; the first batch of more than one caller sizes it for good (max_batch voters for the
    // inputs path -- 11 KB each, plus 2.6 MB of device memory per witness -- and at most 128 for the witness path)
    const size_t kind_max = full ? (size_t)s->max_batch : std::min<size_t>((size_t)s->max_batch, 128);
    w->cap = std::min(kind_max, std::max(w->cap, b.reqs.size() > 1 ? kind_max : (size_t)8));
    if (b.reqs.size() > w->cap) {                            // the other kind sized this worker before: give the tail back to the queue
        std::lock_guard<std::mutex> g(s->mu);
        while (b.reqs.size() > w->cap) { s->q.push_front(b.reqs.back()); b.reqs.pop_back(); b.depths.pop_back(); b.depths.pop_back(); }
    }
    const size_t cap = w->cap;
    // the worker's uploads go on ITS LANE's G1 stream (the context's: zkc_lane_streams) -- the stream its one-pass calls start on, so upload -> witness -> buildABC is one
    // chain and the worker adds no stream (no hardware queue) of its own
    if (!w->st) {
        std::lock_guard<std::mutex> gpu(d->gpu_mu);
        if (!d->ctx && zkc_ctx_create(d->device, &d->ctx)) { d->ctx = nullptr; return fail_all(s, b, ZKC_ERR_HIP, std::string("device ") + std::to_string(d->device) + ": " + zkc_last_error(nullptr)); }
        zkc_ctx::LaneStreams ls; if (zkc_lane_streams(d->ctx, w->slot % zkc::MAX_LANES, false, &ls)) return fail_all(s, b, ZKC_ERR_HIP, zkc_last_error(d->ctx));
        w->st = ls.st;
    }
    if (!w->ev_up && hipEventCreateWithFlags(&w->ev_up, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); w->ev_up = nullptr; return fail_all(s, b, ZKC_ERR_HIP, "hipEventCreate failed"); }
    if ((full && (!w->h_in.ensure(cap * nIn * 32) || !w->d_in.ensure(cap * nIn * 32))) || !w->d_wtns.ensure(cap * nW * 32) || !w->d_status.ensure(cap * 4) ||
        !w->h_proofs.ensure(cap * 256) || !w->h_pubs.ensure(cap * nPub * 32 + 32) || !w->h_status.ensure(cap * 4))
        return fail_all(s, b, ZKC_ERR_HIP, "out of memory for the service's staging buffers on device " + std::to_string(d->device));
    const uint64_t t_a = now_us();
    if (!stage(w, b, 0, nIn, nW, err)) { (void)hipGetLastError(); return fail_all(s, b, ZKC_ERR_HIP, err); }
    int rc = ZKC_OK; zkc_zkey* zk = nullptr;
    uint64_t t_b = now_us(), t_c = 0, t_d = 0, t_e = 0;
    // ---- collect: this worker is its device's collector (worker_main made it so).  It keeps taking requests of its class while they keep coming --
    //   device idle (no call in flight): a lone request with nothing behind it does not wait at all; a burst still arriving (Promise.all over a census: one request every
    //     ~25 us from one thread; sixty-four threads coming back from their previous call) is taken until three waits of 50 us pass without a new one, at most 3 ms;
    //   device busy (other lanes keep the GPU fed): there is no hurry -- until two waits of 100 us pass without an arrival, at most $ZKC_SERVICE_BUSY_WAIT_US (1 ms).
    // The waits are on this worker's condition variable; dispatch wakes it for every request of its class (no polling).
    {
        const uint64_t t0 = now_us(); int quiet = 0; bool first = true;
        for (;;) {
            size_t want; { std::lock_guard<std::mutex> g(s->mu); want = std::min(cap, batch_target(s, d, b.reqs.size())); }
            if (b.reqs.size() >= want) break;
            bool busy; { std::lock_guard<std::mutex> fl(d->fl_mu); busy = d->calls_in_flight() > 0; }
            const uint64_t waited = now_us() - t0;
            if (busy ? (quiet >= 2 || waited >= s->busy_wait_us) : (quiet >= 3 || waited >= 3000)) break;
            std::vector<Req*> more;
            {
                std::unique_lock<std::mutex> lk(s->mu);
                more = grab(s, w, want - b.reqs.size(), &cls);
                if (more.empty()) {
                    if (!busy && first && b.reqs.size() == 1) break;              // a lone caller on an idle device: go
                    w->collecting = &cls; w->room = want - b.reqs.size(); w->wake = false;
                    w->cv.wait_for(lk, std::chrono::microseconds(busy ? 100 : 50), [&] { return w->wake || s->stop; });
                    w->wake = false; w->collecting = nullptr; w->room = 0;
                    if (s->stop) break;
                    want = std::min(cap, batch_target(s, d, b.reqs.size()));
                    if (want > b.reqs.size()) more = grab(s, w, want - b.reqs.size(), &cls);
                }
            }
            first = false;
            if (more.empty()) { quiet++; continue; }
            quiet = 0;
            const size_t from = b.reqs.size();
            keep_right_length(more);
            if (!stage(w, b, from, nIn, nW, err)) { (void)hipGetLastError(); return fail_all(s, b, ZKC_ERR_HIP, err); }
        }
        std::lock_guard<std::mutex> g(s->mu); stop_collecting(s, w);              // the next arrivals are the next collector's
    }
    if (hipEventRecord(w->ev_up, w->st) != hipSuccess) { (void)hipGetLastError(); return fail_all(s, b, ZKC_ERR_HIP, "hipEventRecord failed"); }
    {
        std::lock_guard<std::mutex> gpu(d->gpu_mu);
        t_c = now_us();
        const int B = (int)b.reqs.size();
        // resident key: the service's own image object is the identity (fingerprint + full SHA-256 were settled when the request was accepted)
        { std::string why; if ((rc = ensure_key(s, d, cls.img, &zk, why))) return fail_all(s, b, rc, why); }
        if (full && zk->nLevels != cls.nLevels) return fail_all(s, b, ZKC_ERR_BAD_ARG, "the key is not a ZkFranchiseProofCircuit(" + std::to_string(cls.nLevels) + ") key");
        b.rs.resize((size_t)B * 64);
        for (int i = 0; i < B; i++) memcpy(b.rs.data() + 64 * (size_t)i, b.reqs[i]->rs, 64);
        t_d = now_us();
        // the call in two halves: begin enqueues every pass on THIS worker's lane and returns; the GPU lock is given up before finish waits, so the other workers' calls
        // begin -- and run, on their own lanes -- meanwhile
        rc = zkc::prove_batch_begin(zk, w->slot, w->d_wtns.p, (uint32_t)nW, B, b.rs.data(), true, full ? w->d_in.p : nullptr, full ? (int32_t*)w->d_status.p : nullptr, w->slot % zk->nlanes, w->ev_up,
                                     b.depths.size() == 2 * (size_t)B ? b.depths.data() : nullptr);
        if (rc) return fail_all(s, b, rc, zkc_last_error(d->ctx));
        { std::lock_guard<std::mutex> fl(d->fl_mu); for (auto& k : d->keys) if (k.key == zk) k.in_flight++; }      // still under gpu_mu: the slot is there
        d->proofs_in_flight += B;
    }
    {
        const int B = (int)b.reqs.size();
        rc = zkc::prove_batch_finish(zk, w->slot, (uint8_t*)w->h_proofs.p, (uint8_t*)w->h_pubs.p);
        if (!rc && full && (hipMemcpyAsync(w->h_status.p, w->d_status.p, (size_t)B * 4, hipMemcpyDeviceToHost, w->st) != hipSuccess || zkc_wait_stream(w->st, w->ev_up) != hipSuccess)) { (void)hipGetLastError(); rc = ZKC_ERR_HIP; }
        const std::string why = rc ? (rc == ZKC_ERR_HIP ? std::string("HIP failure while the batch finished: ") : std::string()) + zkc_last_error(d->ctx) : std::string();
        { std::lock_guard<std::mutex> fl(d->fl_mu); for (auto& k : d->keys) if (k.key == zk) k.in_flight--; } d->fl_cv.notify_all();
        d->proofs_in_flight -= B;
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
        finish(s, r, st == ZKC_W_OK ? ZKC_OK : ZKC_ERR_WITNESS, st, why ? why : "a circuit assert failed (see status)");
    }
    { std::lock_guard<std::mutex> g(s->mu); s->us_finish += now_us() - t_e; }
}
void worker_main(zkc_service* s, zkc_service::Worker* w) {
    std::unique_lock<std::mutex> lk(s->mu);
    for (;;) {
        std::vector<Req*> batch;
        // one collector per device: a worker takes requests only while no sibling is gathering a batch (the sibling takes the arrivals; when it stops, dispatch wakes somebody)
        while (!s->stop && !w->warm && (w->dev->collector || (batch = grab(s, w, std::min<size_t>((size_t)s->max_batch, batch_target(s, w->dev, 0)))).empty())) {
            w->idle = true; w->cv.wait(lk, [&] { return w->wake || s->stop; }); w->wake = false; w->idle = false;
        }
        if (!s->stop && w->warm) {                           // bring this device up for that key first; the queue is being served by the devices that have it
            std::shared_ptr<KeyImage> img = std::move(w->warm); w->warm.reset();
            lk.unlock();
            if (hipSetDevice(w->dev->device) == hipSuccess) { std::lock_guard<std::mutex> gpu(w->dev->gpu_mu); std::string why; zkc_zkey* k = nullptr; (void)ensure_key(s, w->dev, img, &k, why); }
            else { (void)hipGetLastError(); std::lock_guard<std::mutex> g(s->mu); if (w->dev->loading.get() == img.get()) w->dev->loading.reset(); }
            lk.lock();
            dispatch(s);
            continue;
        }
        if (s->stop) { for (Req* r : batch) s->q.push_front(r); break; }
        w->dev->collector = w;
        dispatch(s);                                        // what is left in the queue may be another device's
        lk.unlock();
        process(s, w, std::move(batch));
        lk.lock();
    }
    lk.unlock();
    (void)hipSetDevice(w->dev->device);
    for (HipBuf* hb : {&w->h_in, &w->h_wtns, &w->d_in, &w->d_wtns, &w->d_status, &w->h_proofs, &w->h_pubs, &w->h_status}) hb->release();
    if (w->ev_up) (void)hipEventDestroy(w->ev_up);
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
    // [r5] workers per device = lanes of a service key = calls in flight per key; passes of up to `pass` proofs on each lane (work space: workers x pass x ~62 MB at nLevels 160)
    if (const char* e = getenv("ZKC_SERVICE_WORKERS")) s->workers_per_dev = std::max(1, std::min(atoi(e), (int)zkc::MAX_LANES));
    if (const char* e = getenv("ZKC_SERVICE_PASS")) s->pass = std::max(1, std::min(atoi(e), zkc::MSM_MAX_JOBS / 4));
    if (const char* e = getenv("ZKC_SERVICE_BUSY_WAIT_US")) s->busy_wait_us = (uint64_t)std::max(0, atoi(e));
    if (const char* e = getenv("ZKC_SERVICE_MIN_BATCH")) s->min_batch = std::max(1, atoi(e));
    for (int dv : devs) { s->devs.emplace_back(new zkc_service::Dev()); s->devs.back()->device = dv; }
    int idx = 0;
    for (auto& d : s->devs) for (int k = 0; k < s->workers_per_dev; k++) { s->workers.emplace_back(new zkc_service::Worker()); s->workers.back()->dev = d.get(); s->workers.back()->index = idx++; s->workers.back()->slot = k; }
    for (auto& w : s->workers) w->th = std::thread(worker_main, s, w.get());
    *out = s; return ZKC_OK;
}
extern "C" void zkc_service_destroy(zkc_service* s) {
    if (!s) return;
    { std::lock_guard<std::mutex> g(s->mu); s->stop = true; for (auto& w : s->workers) w->cv.notify_all(); }
    for (auto& w : s->workers) if (w->th.joinable()) w->th.join();
    for (Req* r : s->q) finish(s, r, ZKC_ERR_GENERIC, 0, "the proving service was shut down");
    s->q.clear();
    for (auto& d : s->devs) { (void)hipSetDevice(d->device); for (auto& k : d->keys) if (k.key) zkc_zkey_free(k.key); if (d->ctx) zkc_ctx_destroy(d->ctx); }
    s->pin.release();
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
// the whole image, copied the first time it is seen.  [r5] As many images are kept as the devices may hold keys (at least eight), least recently used out first -- but never one
// that a device holds or a request carries: residency is decided by image OBJECT, so dropping the image of a resident key would make its next caller load it a second time
// beside an unreachable copy.
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
    const size_t keep = std::max<size_t>(8, (size_t)s->keys_per_dev * s->devs.size());
    while (s->images.size() >= keep) {
        long lru = -1;                                                       // use_count 1 = only this list holds it: not resident anywhere, no request in flight
        for (size_t i = 0; i < s->images.size(); i++) if (s->images[i].use_count() == 1 && (lru < 0 || s->images[i]->last_use < s->images[(size_t)lru]->last_use)) lru = (long)i;
        if (lru < 0) break;                                                  // every image is in use: the list grows (bounded by keys per device x devices + requests in flight)
        s->images.erase(s->images.begin() + lru);
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
    if (kind == KIND_PROVE && nW) { r->pin = s->pin.get((size_t)nW * 32); if (r->pin) memcpy(r->pin, data, (size_t)nW * 32); }      // by the caller's thread: callers copy side by side
    std::lock_guard<std::mutex> g(s->mu);
    if (s->stop) { s->pin.put(r->pin); delete r; return service_fail(ZKC_ERR_GENERIC, "the proving service was shut down"); }
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
// HBM and pinned host memory the service holds now: out[0] resident keys (all devices), [1] their constant tables, [2] the lanes' work space of the devices' contexts (ONE per
// device, shared by its keys), [3] the largest key's tables, [4] the largest device's work space, [5] the workers' device staging, [6] pinned host memory (workers' staging +
// witness slots), [7] work-space reservations that failed (the lanes then grow on demand)
extern "C" int zkc_service_memory(zkc_service* s, uint64_t out[8]) {
    if (!s || !out) return ZKC_ERR_BAD_ARG;
    for (int i = 0; i < 8; i++) out[i] = 0;
    for (auto& d : s->devs) {
        std::lock_guard<std::mutex> fl(d->fl_mu);
        size_t dev_work = 0;
        for (auto& k : d->keys) {
            size_t t = 0, wk = 0; zkc::zkey_device_bytes(k.key, &t, &wk);
            out[0]++; out[1] += t; dev_work = std::max(dev_work, wk);
            if (t > out[3]) out[3] = t;
        }
        out[2] += dev_work; if (dev_work > out[4]) out[4] = dev_work;
    }
    for (auto& w : s->workers) { for (HipBuf* hb : {&w->d_in, &w->d_wtns, &w->d_status}) out[5] += hb->sz; for (HipBuf* hb : {&w->h_in, &w->h_wtns, &w->h_proofs, &w->h_pubs, &w->h_status}) out[6] += hb->sz; }
    out[6] += s->pin.bytes();
    std::lock_guard<std::mutex> g(s->mu); out[7] = s->reserve_failures;
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
