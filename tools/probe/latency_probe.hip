// Latency probe for gfx950 (MI355X): what ONE wave pays per instruction when it runs a dependency chain alone on its SIMD -- the situation of the
// single-proof path (witness hash chain, bucket-reduction scans, blinding products), where instruction COUNT per wave, not chip throughput, sets the time.
// Questions: (1) issue interval of dependent v_mad_u64_u32 / v_add_u32 / v_readlane_b32 for a lone wave; (2) does it fall when independent chains are
// interleaved in the same wave (is the interval dependency latency or issue cadence?); (3) does a wave whose EXEC mask covers 16 lanes or 1 lane issue faster
// than one with all 64 (does the SIMD skip empty quarter passes?); (4) cycles per dependent f29_mul / f29_sqr.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zk-franchise-proof-circuit_amd/csrc tools/probe/latency_probe.hip -o /tmp/latency_probe && /tmp/latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "zkc_f29.h"
using namespace zkc;
#define ITER 20000

template <int OP, int CHAINS>
__global__ void __launch_bounds__(64) chain(uint64_t* out, uint32_t seed, int live_lanes) {
    if ((int)threadIdx.x >= live_lanes) return;                 // EXEC = lanes [0, live_lanes) from here on
    uint64_t a[4]; uint32_t x = seed + threadIdx.x, y = seed * 3u + 1u;
    for (int i = 0; i < 4; i++) a[i] = ((uint64_t)(seed + i) << 33) + i + threadIdx.x;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < CHAINS; i++) {
                uint32_t lo = (uint32_t)a[i];
                if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (OP == 1) { asm volatile("v_add_u32 %0, %1, %0" : "+v"(lo) : "v"(x)); a[i] = lo; }
                if (OP == 2) { uint32_t s; asm volatile("v_readlane_b32 %0, %1, 0\n\ts_nop 3\n\tv_add_u32 %1, %0, %1" : "=s"(s), "+v"(lo)); a[i] = lo; }     // lane -> scalar -> lane round trip
                if (OP == 3) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(lo), "v"(y) : "vcc"); }                       // low word of the result feeds the multiplier
            }
        }
    }
    uint64_t s = x + y; for (int i = 0; i < 4; i++) s += a[i];
    out[threadIdx.x] = s;
}
template <int OP>
__global__ void __launch_bounds__(64) fchain(uint32_t* out, uint32_t seed, int live_lanes, int iters) {
    if ((int)threadIdx.x >= live_lanes) return;
    uint32_t a[9], b[9];
    for (int k = 0; k < 9; k++) { a[k] = (seed * (k + 3) + threadIdx.x) & F29_MASK; b[k] = (seed * (k + 7) + 11u) & F29_MASK; }
    a[8] &= 0xfffff; b[8] &= 0xfffff;
    for (int it = 0; it < iters; it++) {
        uint32_t r[9];
        if (OP == 0) f29_mul<FrParams>(r, a, b); else f29_sqr<FrParams>(r, a);
#pragma unroll
        for (int k = 0; k < 9; k++) a[k] = r[k];
    }
    for (int k = 0; k < 9; k++) out[threadIdx.x * 9 + k] = a[k];
}
template <class F> static double time_ms(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    uint64_t* d; hipMalloc(&d, 1 << 16);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("# %s, clockRate %d kHz.  ONE wave on an idle chip; ns per wave-instruction (x 2.4 = cycles at the nominal clock); empty-launch time subtracted\n", p.gcnArchName, p.clockRate);
    const double t_empty = time_ms([&] { hipLaunchKernelGGL((chain<0, 1>), dim3(1), dim3(64), 0, 0, d, 7u, 0); });
    printf("empty launch %.4f ms\n", t_empty);
#define RUN(OP, CH, NAME, PER) for (int live : {64, 32, 16, 1}) { const double ms = time_ms([&] { hipLaunchKernelGGL((chain<OP, CH>), dim3(1), dim3(64), 0, 0, d, 7u, live); }) - t_empty; \
        printf("%-44s chains=%d live lanes=%2d  %8.3f ms  %6.2f ns per instruction\n", NAME, CH, live, ms, ms * 1e6 / ((double)ITER * 8 * CH * PER)); }
    RUN(0, 1, "v_mad_u64_u32, dependent", 1) RUN(0, 2, "v_mad_u64_u32, 2 interleaved chains", 1) RUN(0, 4, "v_mad_u64_u32, 4 interleaved chains", 1)
    RUN(3, 1, "v_mad_u64_u32, result feeds the multiplier", 1)
    RUN(1, 1, "v_add_u32, dependent", 1) RUN(1, 4, "v_add_u32, 4 interleaved chains", 1)
    RUN(2, 1, "v_readlane_b32 + s_nop 3 + v_add_u32 round trip", 1)
    uint32_t* d32 = (uint32_t*)d; const int iters = 4000;
    for (int live : {64, 16, 1}) {
        const double m = time_ms([&] { hipLaunchKernelGGL(fchain<0>, dim3(1), dim3(64), 0, 0, d32, 7u, live, iters); }) - t_empty;
        const double s = time_ms([&] { hipLaunchKernelGGL(fchain<1>, dim3(1), dim3(64), 0, 0, d32, 7u, live, iters); }) - t_empty;
        printf("f29_mul dependent chain, live lanes=%2d: %7.1f ns per product   f29_sqr: %7.1f ns\n", live, m * 1e6 / iters, s * 1e6 / iters);
    }
    return 0;
}
