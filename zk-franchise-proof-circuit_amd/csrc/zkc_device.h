// zkc_device.h -- structs shared between the host side of libzkcensus and its HIP kernels (product code).
#pragma once
#include "zkc_field.h"
#include "../../include/zkcensus.h"   // ZKC_W_* status codes

namespace zkc {

struct PoseidonTable {           // device pointers, Montgomery form; index = t (3,4,5)
    const Fr* C[6]; const Fr* S[6]; const Fr* M[6]; const Fr* P[6];
    const Fr* base; const uint32_t* base29;      // the same constants as 12-word entries (nine 29-bit limbs, R' form, below 1.2 p): entry k <-> base[k]
    const uint32_t* K29[6];                      // t = 3, 4: K29[t][r t + k] = S_r[k ? t + k - 1 : 0] * C[5 t + r], 12-word entries below p: the round constant of word 0 already multiplied into the sparse-mix row and column of partial round r (zkc_witness.hip, poseidon_wave29)
};

// Wire layout of ZkFranchiseProofCircuit(nL) as circom 2.1.5 -O2 numbered it (DESIGN.md "witness layout").
struct WitnessLayout {
    int nL, n, nInputs, nWires;
    int off_census, off_checknull, off_checkweight, off_nullifier, off_sik, off_sikver;   // absolute wire indices
    int off_n2bnew, off_n2bold, off_levins, off_iszero, ver_size;                          // relative to a verifier block
    static constexpr int kHash3 = 20 + 2 + 57 + 46 + 114;     // internals of a Poseidon(2) level hash
    static constexpr int kHash1New = 26 + 3 + 56 + 60 + 112;  // internals of Poseidon(key,value,1)
    static constexpr int kSik = 27 + 3 + 56 + 62 + 112;       // internals of Poseidon(address,password,signature)
    static constexpr int kNullifier = 296;                    // internals of the t=5 instance (out is a public wire)

    // first wire of level i's sub-block, relative to the verifier block
    __host__ __device__ int lvl_off(int i) const {
        const int base = 4 + kHash1New;                       // areKeyEquals.out, .inv, checkRoot.inv, hash1New.out + internals
        if (i == 0) return base;
        int o = base + (5 + kHash3) + (i - 1) * (6 + kHash3);
        if (i > n - 3) o += 1;                                // level n-3 carries st_top as an extra wire
        if (i > n - 2) o -= 1;                                // level n-2 has no st_inew wire
        return o;
    }
    static WitnessLayout make(int nLevels) {
        WitnessLayout L{};
        L.nL = nLevels; L.n = nLevels + 1; L.nInputs = 12 + 2 * L.n;
        const int n = L.n;
        int o = L.lvl_off(n - 1) + 2;                         // level n-1: st_top[n-2], lrbit
        L.off_n2bnew = o; o += (253 - n) + 127 + 133;
        L.off_n2bold = o; o += 253 + 127 + 133;
        L.off_levins = o; o += n - 2;
        L.off_iszero = o; o += 2 * (n - 2) + 1;               // (out,inv) x (n-2), inv of i = n-2
        o += 1;                                               // isZero[n-1].inv
        L.ver_size = o;
        L.off_census = 13 + 2 * nLevels;
        L.off_checknull = L.off_census + L.ver_size;
        L.off_checkweight = L.off_checknull + 1;
        L.off_nullifier = L.off_checkweight + 251;
        L.off_sik = L.off_nullifier + kNullifier;
        L.off_sikver = L.off_sik + 1 + kSik;
        L.nWires = L.off_sikver + L.ver_size;
        return L;
    }
};

}  // namespace zkc
