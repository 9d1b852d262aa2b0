/* loadgen.c -- N native caller threads on the single-proof entry points of libzkcensus.so, the way goroutines reach rapidsnark's groth16_prover through cgo
 * (zk_census_test.go:89: prover.Prove per voter).  Measurement tooling for bench.py / tools/service_bench.py: Python threads would put the interpreter lock between
 * the callers.  Built by __graft_entry__.build() with gcc; the entry points arrive as function pointers (ctypes hands them over), so nothing is linked here.
 * Every caller makes `calls` calls in a row for its own voter (thread t: voter t mod nvoters); all start behind one barrier; wall = barrier -> last return. */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int (*prover_fn)(const void*, unsigned long, const void*, unsigned long, char*, unsigned long*, char*, unsigned long*, char*, unsigned long);
typedef int (*fullprove_fn)(void*, const void*, size_t, int, const void*, const uint8_t*, uint8_t*, uint8_t*, int32_t*, char*, size_t);
typedef int (*service_prove_fn)(void*, const void*, size_t, const void*, uint32_t, const uint8_t*, uint8_t*, uint8_t*, char*, size_t);
typedef int (*g16_fullprove_fn)(const void*, unsigned long, const void*, unsigned long, const char*, unsigned long, char*, unsigned long*, char*, unsigned long*, char*, unsigned long);
typedef int (*fullprove_json_fn)(void*, const void*, size_t, const void*, size_t, const char*, size_t, const uint8_t*, uint8_t*, uint8_t*, int32_t*, char*, size_t);

struct gate { pthread_mutex_t m; pthread_cond_t cv; int go; };      /* go: 0 wait, 1 run, -1 leave (a thread could not be started) */
typedef struct {
    int mode, t, calls, nvoters, nLevels, nPub; void* fn; void* svc;
    const void* zkey; size_t zkey_len; const void* wasm; size_t wasm_len;
    const void* const* items; const size_t* item_len;            /* per voter: .wtns image (mode 0), flat inputs (mode 1), inputs JSON text (mode 2), witness payload (mode 3: zkc_service_prove), inputs JSON text again (mode 4: groth16_fullprove, JSON out) */
    char* proof_json; char* public_json;                          /* mode 0: [threads][calls][2048] each */
    uint8_t* proofs; uint8_t* publics; int32_t* statuses;         /* modes 1, 2: [threads][calls][256], [..][nPub * 32], [..] */
    double* lat_ms; struct gate* gate; int failed; double t_end;
} caller_t;

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }

static void* caller_main(void* p) {
    caller_t* c = (caller_t*)p;
    const int v = c->t % c->nvoters;
    pthread_mutex_lock(&c->gate->m); while (c->gate->go == 0) pthread_cond_wait(&c->gate->cv, &c->gate->m); const int go = c->gate->go; pthread_mutex_unlock(&c->gate->m);
    if (go < 0) return NULL;
    for (int k = 0; k < c->calls; k++) {
        const size_t i = (size_t)c->t * c->calls + k;
        const double t0 = now_s(); int rc;
        if (c->mode == 0) {
            unsigned long ps = 2048, us = 2048; char err[256];
            rc = ((prover_fn)c->fn)(c->zkey, c->zkey_len, c->items[v], c->item_len[v], c->proof_json + 2048 * i, &ps, c->public_json + 2048 * i, &us, err, sizeof err);
        } else if (c->mode == 1) {
            char err[256];
            rc = ((fullprove_fn)c->fn)(c->svc, c->zkey, c->zkey_len, c->nLevels, c->items[v], NULL, c->proofs + 256 * i, c->publics + (size_t)c->nPub * 32 * i, c->statuses + i, err, sizeof err);
        } else if (c->mode == 4) {
            unsigned long ps = 2048, us = 2048; char err[256];
            rc = ((g16_fullprove_fn)c->fn)(c->zkey, c->zkey_len, c->wasm, c->wasm_len, (const char*)c->items[v], c->item_len[v], c->proof_json + 2048 * i, &ps, c->public_json + 2048 * i, &us, err, sizeof err);
        } else if (c->mode == 3) {
            char err[256];
            rc = ((service_prove_fn)c->fn)(c->svc, c->zkey, c->zkey_len, c->items[v], (uint32_t)(c->item_len[v] / 32), NULL, c->proofs + 256 * i, c->publics + (size_t)c->nPub * 32 * i, err, sizeof err);
            if (c->statuses) c->statuses[i] = 0;
        } else {
            char err[256];
            rc = ((fullprove_json_fn)c->fn)(c->svc, c->zkey, c->zkey_len, c->wasm, c->wasm_len, (const char*)c->items[v], c->item_len[v], NULL, c->proofs + 256 * i,
                                            c->publics + (size_t)c->nPub * 32 * i, c->statuses + i, err, sizeof err);
        }
        if (rc) c->failed++;
        if (c->lat_ms) c->lat_ms[i] = (now_s() - t0) * 1e3;
    }
    c->t_end = now_s();
    return NULL;
}

/* returns the number of failed calls (-1: could not start the threads); *wall_s = barrier -> last caller done */
int zkc_loadgen_run(int mode, void* fn, void* svc, int threads, int calls, const void* zkey, size_t zkey_len, const void* wasm, size_t wasm_len, int nLevels, int nPub,
                    const void* const* items, const size_t* item_len, int nvoters, char* proof_json, char* public_json, uint8_t* proofs, uint8_t* publics, int32_t* statuses,
                    double* wall_s, double* lat_ms) {
    if (threads <= 0 || calls <= 0 || nvoters <= 0 || !fn) return -1;
    struct gate g; pthread_mutex_init(&g.m, NULL); pthread_cond_init(&g.cv, NULL); g.go = 0;
    caller_t* cs = (caller_t*)calloc((size_t)threads, sizeof(caller_t)); pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    int started = 0;
    for (int t = 0; t < threads; t++) {
        caller_t* c = &cs[t];
        c->mode = mode; c->t = t; c->calls = calls; c->nvoters = nvoters; c->nLevels = nLevels; c->nPub = nPub; c->fn = fn; c->svc = svc; c->zkey = zkey; c->zkey_len = zkey_len;
        c->wasm = wasm; c->wasm_len = wasm_len; c->items = items; c->item_len = item_len; c->proof_json = proof_json; c->public_json = public_json; c->proofs = proofs; c->publics = publics;
        c->statuses = statuses; c->lat_ms = lat_ms; c->gate = &g;
        pthread_attr_t at; pthread_attr_init(&at); pthread_attr_setstacksize(&at, 1 << 20);
        const int e = pthread_create(&th[t], &at, caller_main, c); pthread_attr_destroy(&at);
        if (e != 0) break;
        started++;
    }
    pthread_mutex_lock(&g.m); g.go = started == threads ? 1 : -1; pthread_cond_broadcast(&g.cv); pthread_mutex_unlock(&g.m);
    const double t0 = now_s(); double t1 = t0; int failed = 0;
    for (int t = 0; t < started; t++) { pthread_join(th[t], NULL); failed += cs[t].failed; if (cs[t].t_end > t1) t1 = cs[t].t_end; }
    if (wall_s) *wall_s = t1 - t0;
    pthread_mutex_destroy(&g.m); pthread_cond_destroy(&g.cv); free(cs); free(th);
    return started == threads ? failed : -1;
}
