// Instruction-rate probe for gfx950 (MI355X): how many lane-operations per second the VALU sustains for the instructions the 254-bit
// field arithmetic of libzkcensus is made of (v_mad_u64_u32 and the 32/64-bit carry-handling ops around it).
//
// Each lane runs 8 independent dependency chains of ITER x 8 instructions.  The grid is sized so that every wave is resident from
// the first cycle to the last (W waves per SIMD on all 1024 SIMDs), so besides the wall-clock rate the probe reports the shader clock
// estimate (s_memtime ticks of a wave / kernel time) and the issue cost in those ticks per wave-instruction per SIMD.  NOTE: on gfx950 s_memtime does not
// follow sclk (rocm-smi and GRBM_GUI_ACTIVE give 2.1-2.2 GHz under these loops, profiles/r02_power_clock_trace.json): only the lane-ops/s column is used.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/rate_probe.hip -o /tmp/rate_probe && /tmp/rate_probe > profiles/r02_rate_probe.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define ITER 32768
template <int OP> __global__ void __launch_bounds__(256) probe(uint64_t* out, uint64_t* ticks, uint32_t seed) {
    uint64_t a[8]; double d[8]; uint32_t x = seed + threadIdx.x, y = seed * 3 + blockIdx.x, z = seed ^ 0x55aa55aau;
    for (int i = 0; i < 8; i++) { a[i] = ((uint64_t)(seed + i) << 33) + i + threadIdx.x; d[i] = (double)(seed + i) + threadIdx.x; }
    double m = 1.0000001, c = 0.5;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < (OP == 17 ? 0 : ITER); it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t lo = (uint32_t)a[i];
            if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
            if (OP == 1) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(m), "v"(c));
            if (OP == 2) asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if (OP == 3) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(x)); a[i] = lo; }
            if (OP == 4) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(x)); a[i] = lo; }
            if (OP == 5) { asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(lo) : "v"(x), "v"(y)); a[i] = lo; }
            if (OP == 6) { asm volatile("v_add_u32 %0, %1, %0" : "+v"(lo) : "v"(x)); a[i] = lo; }
            if (OP == 7) { asm volatile("v_and_b32 %0, %1, %0" : "+v"(lo) : "v"(x)); a[i] = lo; }
            if (OP == 8) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(a[i]));
            if (OP == 9) { asm volatile("v_alignbit_b32 %0, %1, %0, 29" : "+v"(lo) : "v"(x)); a[i] = lo; }
            if (OP == 10) { asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(lo) : "v"(x), "v"(y)); a[i] = lo; }
            if (OP == 11) { asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(lo) : "v"(x)); a[i] = lo; }
            if (OP == 12) { asm volatile("v_mov_b32 %0, %1" : "+v"(lo) : "v"(x)); a[i] = lo; }
            if (OP == 13) { uint32_t hi = (uint32_t)(a[i] >> 32);          // 64-bit add as a carry pair
                            asm volatile("v_add_co_u32 %0, vcc, %2, %0\n\tv_addc_co_u32 %1, vcc, %3, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(x), "v"(y) : "vcc");
                            a[i] = ((uint64_t)hi << 32) | lo; }
            if (OP == 14) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(x) : "vcc"); a[i] = lo; }
            if (OP == 15) { asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(lo) : "v"(x), "v"(z)); a[i] = lo; }
            if (OP == 16) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_and_b32 %1, %2, %1" : "+v"(a[i]), "+v"(z) : "v"(x), "v"(y) : "vcc");   // 1 mad : 1 full-rate op, interleaved
        }
    }
    if (OP == 17) {     // the accumulation kernel's mix: 10 v_mad_u64_u32, 3 half-rate 64-bit / 3-operand ops, 3 plain 32-bit ops per 16 instructions, no two
                        // neighbours on the same register (profiles/r02_accumulate_isa_histogram.txt: 63.5 % / 19 % / 17.5 % of one mixed addition)
        uint64_t b0 = a[0] ^ 5, b1 = a[1] ^ 9; uint32_t z0 = x ^ 1, z1 = y ^ 2, z2 = z ^ 3, z3 = x + y;
        for (int it = 0; it < ITER * 4; it++) {
            asm volatile(
                "v_mad_u64_u32 %0, vcc, %14, %15, %0\n\tv_mad_u64_u32 %1, vcc, %14, %15, %1\n\tv_and_b32 %10, %14, %10\n\tv_mad_u64_u32 %2, vcc, %14, %15, %2\n\t"
                "v_lshrrev_b64 %8, 1, %8\n\tv_mad_u64_u32 %3, vcc, %14, %15, %3\n\tv_mad_u64_u32 %4, vcc, %14, %15, %4\n\tv_add_u32 %11, %15, %11\n\t"
                "v_mad_u64_u32 %5, vcc, %14, %15, %5\n\tv_lshl_add_u64 %9, %8, 0, %9\n\tv_mad_u64_u32 %6, vcc, %14, %15, %6\n\tv_mad_u64_u32 %7, vcc, %14, %15, %7\n\t"
                "v_and_b32 %12, %15, %12\n\tv_mad_u64_u32 %0, vcc, %15, %14, %0\n\tv_alignbit_b32 %13, %14, %13, 29\n\tv_mad_u64_u32 %1, vcc, %15, %14, %1"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b0), "+v"(b1), "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3)
                : "v"(x), "v"(y) : "vcc");
        }
        a[2] += b0 + b1 + z0 + z1 + z2 + z3;
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t s = x + z + y; for (int i = 0; i < 8; i++) s += a[i] + (uint64_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
template <int OP> void run(const char* name, int waves_per_simd, int ops_per_slot, uint64_t* d_out, uint64_t* d_ticks) {
    const int cus = 256, threads = 256;                       // one workgroup = 4 waves = one wave per SIMD of a CU
    const int blocks = cus * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, d_ticks, 7u);
    hipEventRecord(e0); hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, d_ticks, 9u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> t(blocks * 4); hipMemcpy(t.data(), d_ticks, t.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : t) avg += (double)v; avg /= t.size();
    const double ops = (double)blocks * threads * ITER * 8 * ops_per_slot;
    const double wave_instr = (double)ITER * 8 * ops_per_slot;
    printf("%-34s W=%d  %8.3f ms  %7.2f T lane-ops/s  s_memtime %.2f GHz  %5.2f ticks per wave-instruction per SIMD\n", name, waves_per_simd, ms, ops / ms / 1e9,
           avg / (ms * 1e-3) / 1e9, avg / (wave_instr * waves_per_simd));
}
int main() {
    uint64_t *d, *dt; hipMalloc(&d, 256 * 8 * 256 * 8); hipMalloc(&dt, 256 * 8 * 4 * 8);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("# %s, %d CUs, clockRate %d kHz; 8 chains per lane, %d iterations; W = waves per SIMD\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, ITER);
    for (int w : {1, 2, 3, 4, 8}) run<17>("accumulation mix 10 mad : 3 half : 3 full", w, 8, d, dt);     // 16 instructions per 8-instruction slot of the generic loop, x 4 iterations: see ops below
    for (int w : {2, 3, 8}) {
        run<6>("v_add_u32", w, 1, d, dt); run<7>("v_and_b32", w, 1, d, dt); run<0>("v_mad_u64_u32", w, 1, d, dt);
        run<16>("v_mad_u64_u32 + v_and_b32 (1:1)", w, 2, d, dt);
    }
    const int w = 3;                                          // the accumulation kernel runs at 2-3 waves per SIMD
    run<3>("v_mul_lo_u32", w, 1, d, dt); run<4>("v_mul_hi_u32", w, 1, d, dt); run<5>("v_mad_u32_u24", w, 1, d, dt);
    run<2>("v_lshl_add_u64", w, 1, d, dt); run<8>("v_lshrrev_b64", w, 1, d, dt); run<13>("v_add_co_u32 + v_addc_co_u32", w, 2, d, dt);
    run<9>("v_alignbit_b32", w, 1, d, dt); run<10>("v_add3_u32", w, 1, d, dt); run<11>("v_lshl_add_u32", w, 1, d, dt); run<12>("v_mov_b32", w, 1, d, dt);
    run<14>("v_cndmask_b32", w, 1, d, dt); run<15>("v_perm_b32", w, 1, d, dt); run<1>("v_fma_f64", w, 1, d, dt);
    return 0;
}
