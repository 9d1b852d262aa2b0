// zkc_fixedbase.h -- host-side fixed-base scalar multiplication tables (8-bit windows) shared by the test setup and the
// prover's blinding step.  table[w][d-1] = d * 2^(8w) * G, affine, Montgomery.
#pragma once
#include <functional>
#include <thread>
#include <vector>
#include "zkc_curve.h"

namespace zkc {

inline void parallel_for(size_t n, const std::function<void(size_t, size_t)>& f) {
    unsigned nt = std::thread::hardware_concurrency(); if (nt == 0) nt = 4; if (nt > 32) nt = 32;
    if (n < 1024) { f(0, n); return; }
    std::vector<std::thread> th; size_t chunk = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) { size_t a = t * chunk, b = std::min(n, a + chunk); if (a < b) th.emplace_back(f, a, b); }
    for (auto& t : th) t.join();
}


// fixed-base scalar multiplication k*G with 8-bit windows (table[w][d-1] = d * 2^(8w) * G, affine)
template <class F>
struct FixedBase {
    std::vector<Affine<F>> tab;   // 32 x 255
    explicit FixedBase(const Affine<F>& g) {
        std::vector<XYZZ<F>> t(32 * 255);
        XYZZ<F> base = XYZZ<F>::from_affine(g);
        for (int w = 0; w < 32; w++) {
            XYZZ<F> acc = base;
            for (int d = 1; d <= 255; d++) { t[w * 255 + d - 1] = acc; acc = xyzz_add(acc, base); }
            base = acc;           // 256 * previous base
        }
        tab.resize(t.size());
        parallel_for(t.size(), [&](size_t a, size_t b) { for (size_t i = a; i < b; i++) tab[i] = xyzz_to_affine(t[i]); });
    }
    Affine<F> mul(const Fr& k) const {
        uint32_t s[8]; fp_to_std<FrParams>(s, k);
        XYZZ<F> acc = XYZZ<F>::inf();
        for (int w = 0; w < 32; w++) { uint32_t d = (s[w >> 2] >> (8 * (w & 3))) & 0xff; if (d) acc = xyzz_add_affine(acc, tab[w * 255 + d - 1]); }
        return xyzz_to_affine_gcd(acc);      // host only (test setup, key load): the binary-Euclid inversion is a quarter of Fermat's work on a CPU core
    }
};


}  // namespace zkc
