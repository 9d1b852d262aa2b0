/* abi_client.c -- a plain C99 client of include/zkcensus.h, the way a cgo preamble or any C host sees the library: no C++, no torch, no Python.
 * Build (tests/test_c_client.py does):  gcc -std=c99 -Wall -Wextra -Werror -pedantic tests/host/abi_client.c -Iinclude -ldl -o abi_client
 * With no argument it only checks that the header compiles as C and that every entry point it uses resolves in the shared library (CPU boxes).
 * With <lib> <zkey> <inputs.bin> <n> it proves n voters through a device pool (device 0 listed twice) and verifies each proof with
 * zkc_verify_bin-compatible data written for the caller: prints one line of JSON.
 * With <lib> json <zkey> <inputs.json> <proof.json> <public.json> it does what a cgo prover.Prove(zkey, wasm, inputs) does (zk_census_test.go:81-93):
 * the FILE IMAGES of the key and of inputs_example.json go to groth16_fullprove -- size query first (SHORT_BUFFER), then the call -- and the two JSON texts it returns are
 * written out for the caller to verify; a damaged inputs text must come back as 1 with circom_runtime's message. */
#include "zkcensus.h"
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int (*pool_create_t)(const int*, int, zkc_pool**);
typedef void (*pool_destroy_t)(zkc_pool*);
typedef int (*pool_load_t)(zkc_pool*, const void*, size_t);
typedef int (*pool_prove_t)(zkc_pool*, const void*, int, const uint8_t*, uint8_t*, uint8_t*, int32_t*);
typedef const char* (*pool_err_t)(const zkc_pool*);
typedef int (*n_inputs_t)(int);
typedef int (*g16_fullprove_t)(const void*, unsigned long, const void*, unsigned long, const char*, unsigned long, char*, unsigned long*, char*, unsigned long*, char*, unsigned long);

static void* must(void* h, const char* name) {
    void* p = dlsym(h, name);
    if (!p) { fprintf(stderr, "missing symbol %s\n", name); exit(2); }
    return p;
}
static unsigned char* slurp(const char* path, size_t* len) {
    FILE* f = fopen(path, "rb"); unsigned char* b; long n;
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
    b = (unsigned char*)malloc((size_t)n);
    if (!b || fread(b, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
    fclose(f); *len = (size_t)n; return b;
}

int main(int argc, char** argv) {
    const char* lib = argc > 1 ? argv[1] : "zk-franchise-proof-circuit_amd/libzkcensus.so";
    void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
    pool_create_t pool_create; pool_destroy_t pool_destroy; pool_load_t pool_load; pool_prove_t pool_prove; pool_err_t pool_err; n_inputs_t n_inputs;
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    *(void**)(&pool_create) = must(h, "zkc_pool_create"); *(void**)(&pool_destroy) = must(h, "zkc_pool_destroy");
    *(void**)(&pool_load) = must(h, "zkc_pool_zkey_load"); *(void**)(&pool_prove) = must(h, "zkc_pool_fullprove_batch");
    *(void**)(&pool_err) = must(h, "zkc_pool_last_error"); *(void**)(&n_inputs) = must(h, "zkc_circuit_n_inputs");
    if (n_inputs(160) != 334) { fprintf(stderr, "zkc_circuit_n_inputs(160) = %d\n", n_inputs(160)); return 1; }
    if (argc < 5) { (void)must(h, "groth16_fullprove"); (void)must(h, "zkc_inputs_from_json"); printf("{\"header_compiles_as_c\": true, \"symbols_resolve\": true}\n"); return 0; }
    if (!strcmp(argv[2], "json") && argc >= 7) {
        g16_fullprove_t fullprove; size_t zlen, jlen; unsigned char* zkey = slurp(argv[3], &zlen); unsigned char* js = slurp(argv[4], &jlen);
        unsigned long ps = 0, us = 0, need_p, need_u; char err[512]; char *pb, *ub; int rc, rc_short, rc_bad; FILE* f;
        *(void**)(&fullprove) = must(h, "groth16_fullprove");
        rc_short = fullprove(zkey, zlen, NULL, 0, (const char*)js, jlen, NULL, &ps, NULL, &us, err, sizeof err);      /* size query: nothing is proved */
        if (rc_short != ZKC_ERR_SHORT_BUFFER || ps == 0 || us == 0) { fprintf(stderr, "size query: rc %d\n", rc_short); return 1; }
        pb = (char*)malloc(ps); ub = (char*)malloc(us); need_p = ps; need_u = us;
        rc = fullprove(zkey, zlen, NULL, 0, (const char*)js, jlen, pb, &ps, ub, &us, err, sizeof err);
        if (rc != ZKC_OK) { fprintf(stderr, "groth16_fullprove: %d %s\n", rc, err); return 1; }
        f = fopen(argv[5], "wb"); if (!f || fwrite(pb, 1, strlen(pb), f) != strlen(pb)) { perror("proof"); return 2; } fclose(f);
        f = fopen(argv[6], "wb"); if (!f || fwrite(ub, 1, strlen(ub), f) != strlen(ub)) { perror("public"); return 2; } fclose(f);
        js[jlen / 2] = '}';                                /* a damaged document: refused with a message, nothing proved */
        { unsigned long p2 = need_p, u2 = need_u; err[0] = 0; rc_bad = fullprove(zkey, zlen, NULL, 0, (const char*)js, jlen, pb, &p2, ub, &u2, err, sizeof err); }
        printf("{\"rc\": %d, \"size_query_rc\": %d, \"damaged_inputs_rc\": %d, \"damaged_inputs_message_is_set\": %s}\n", rc, rc_short, rc_bad, err[0] ? "true" : "false");
        free(pb); free(ub); free(zkey); free(js);
        return 0;
    }
    {
        size_t zlen, ilen; unsigned char* zkey = slurp(argv[2], &zlen); unsigned char* in = slurp(argv[3], &ilen);
        const int n = atoi(argv[4]), nlevels = argc > 5 ? atoi(argv[5]) : 160;
        const int devs[2] = {0, 0};
        zkc_pool* pool = NULL; int rc, i, bad = 0;
        uint8_t* proofs = (uint8_t*)calloc((size_t)n, 256); uint8_t* pubs = (uint8_t*)calloc((size_t)n, 8 * 32); int32_t* status = (int32_t*)calloc((size_t)n, sizeof(int32_t));
        FILE* out;
        if ((size_t)n * (size_t)n_inputs(nlevels) * 32 != ilen) { fprintf(stderr, "inputs file has %lu bytes, expected %d voters\n", (unsigned long)ilen, n); return 2; }
        if ((rc = pool_create(devs, 2, &pool)) != ZKC_OK) { fprintf(stderr, "pool_create: %s\n", pool_err(NULL)); return 1; }
        if ((rc = pool_load(pool, zkey, zlen)) != ZKC_OK) { fprintf(stderr, "zkey_load: %s\n", pool_err(pool)); return 1; }
        rc = pool_prove(pool, in, n, NULL, proofs, pubs, status);
        if (rc != ZKC_OK && rc != ZKC_ERR_WITNESS) { fprintf(stderr, "fullprove: %s\n", pool_err(pool)); return 1; }
        for (i = 0; i < n; i++) bad += status[i] != ZKC_W_OK;
        out = fopen(argv[argc - 1], "wb");                 /* last argument: where proofs || publics go for the caller to verify */
        if (!out || fwrite(proofs, 256, (size_t)n, out) != (size_t)n || fwrite(pubs, 256, (size_t)n, out) != (size_t)n) { perror("output"); return 2; }
        fclose(out);
        pool_destroy(pool);
        printf("{\"voters\": %d, \"rc\": %d, \"failed_asserts\": %d}\n", n, rc, bad);
        free(proofs); free(pubs); free(status); free(zkey); free(in);
    }
    return 0;
}
