// Instruction-rate probe for gfx950: v_mad_u64_u32, v_fma_f64, v_lshl_add_u64, v_mul_lo_u32, v_mul_hi_u32, v_add_co/addc, v_mad_u32_u24.
// Each lane runs 8 independent dependency chains of ITER x 8 instructions; the grid oversubscribes every SIMD with 8 waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 65536
template <int OP> __global__ void __launch_bounds__(256) probe(uint64_t* out, uint32_t seed) {
    uint64_t a[8]; double d[8]; uint32_t x = seed + threadIdx.x, y = seed * 3 + blockIdx.x;
    for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; d[i] = (double)(seed + i) + threadIdx.x; }
    double m = 1.0000001, c = 0.5;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
            if (OP == 1) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(m), "v"(c));
            if (OP == 2) asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if (OP == 3) { uint32_t lo = (uint32_t)a[i]; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(x)); a[i] = lo; }
            if (OP == 4) { uint32_t lo = (uint32_t)a[i]; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(x)); a[i] = lo; }
            if (OP == 5) { uint32_t lo = (uint32_t)a[i]; asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(lo) : "v"(x), "v"(y)); a[i] = lo; }
            if (OP == 6) { uint32_t lo = (uint32_t)a[i]; asm volatile("v_add_u32 %0, %1, %0" : "+v"(lo) : "v"(x)); a[i] = lo; }
            if (OP == 7) { uint32_t lo = (uint32_t)a[i]; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(lo) : "vcc"); }   // dependent operand too
            if (OP == 8) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(m));
            if (OP == 9) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c));
        }
    }
    uint64_t s = 0; for (int i = 0; i < 8; i++) s += a[i] + (uint64_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, uint64_t* d_out) {
    const int blocks = 256 * 8, threads = 256;      // 8 workgroups of 4 waves per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 7u);
    hipEventRecord(e0); hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 9u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * threads * ITER * 8;
    printf("%-28s %8.3f ms  %8.2f T lane-ops/s  (%.1f lanes/clk/CU at 2.4 GHz)\n", name, ms, ops / ms / 1e9, ops / (ms * 1e-3) / 256 / 2.4e9);
}
int main() {
    uint64_t* d; hipMalloc(&d, 256 * 8 * 256 * 8);
    run<6>("v_add_u32", d); run<0>("v_mad_u64_u32", d); run<7>("v_mad_u64_u32 (dep operand)", d); run<1>("v_fma_f64", d); run<8>("v_mul_f64", d); run<9>("v_add_f64", d);
    run<2>("v_lshl_add_u64", d); run<3>("v_mul_lo_u32", d); run<4>("v_mul_hi_u32", d); run<5>("v_mad_u32_u24", d);
    return 0;
}
