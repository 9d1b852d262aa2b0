// Calibration of rocprofv3's FETCH_SIZE for the access pattern of the G1 bucket accumulation: every lane gathers 64-byte rows (four 16-byte loads)
// at random from a table far larger than the caches.  The microarch guide calibrates FETCH_SIZE only for wide coalesced streams (where gfx950
// reports half the bytes) and asks for a calibration in one's own pattern before an absolute figure is trusted.  This program runs
//   gather64  : ROWS_PER_LANE random 64-byte rows per lane out of a 512 MB table     (known bytes = lanes x rows x 64)
//   stream16  : the same number of bytes as one coalesced stream, 16 bytes per lane   (the guide's calibrated case: expect 1/2)
// and prints the known byte counts; tools/pmc_calibrate.py divides them by the counter values of
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- /tmp/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr size_t TABLE_ROWS = 8u << 20;          // x 64 B = 512 MB
constexpr int ROWS_PER_LANE = 256;
__global__ void __launch_bounds__(128) gather64(const uint4* __restrict__ table, uint32_t* __restrict__ out) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = gid * 2654435761u + 12345u, acc = 0;
    uint32_t row = (x >> 7) & (TABLE_ROWS - 1);
    uint4 a = table[4 * (size_t)row], b = table[4 * (size_t)row + 1], c = table[4 * (size_t)row + 2], d = table[4 * (size_t)row + 3];
    for (int i = 0; i < ROWS_PER_LANE; i++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t nrow = (x >> 7) & (TABLE_ROWS - 1);                       // next gather in flight while this one is consumed, as in the kernel
        const uint4 na = table[4 * (size_t)nrow], nb = table[4 * (size_t)nrow + 1], nc = table[4 * (size_t)nrow + 2], nd = table[4 * (size_t)nrow + 3];
        acc += a.x ^ b.y ^ c.z ^ d.w;
        a = na; b = nb; c = nc; d = nd;
    }
    out[gid] = acc + a.x;
}
__global__ void __launch_bounds__(256) stream16(const uint4* __restrict__ table, uint32_t* __restrict__ out, size_t n16) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (size_t i = gid; i < n16; i += stride) { const uint4 v = table[i]; acc += v.x ^ v.w; }
    out[gid] = acc;
}
int main() {
    uint4* table; uint32_t* out;
    hipMalloc(&table, TABLE_ROWS * 64); hipMalloc(&out, 4u << 20);
    hipMemset(table, 1, TABLE_ROWS * 64);
    const int lanes = 256 * 8 * 128;                                             // 8 workgroups of 128 per CU
    hipLaunchKernelGGL(gather64, dim3(lanes / 128), dim3(128), 0, 0, table, out);
    hipDeviceSynchronize();
    const size_t gather_bytes = (size_t)lanes * (ROWS_PER_LANE + 1) * 64;
    hipLaunchKernelGGL(stream16, dim3(256 * 8), dim3(256), 0, 0, table, out, gather_bytes / 16 < TABLE_ROWS * 4 ? gather_bytes / 16 : TABLE_ROWS * 4);
    hipDeviceSynchronize();
    const size_t stream_bytes = (gather_bytes / 16 < TABLE_ROWS * 4 ? gather_bytes / 16 : TABLE_ROWS * 4) * 16;
    printf("{\"gather64_known_bytes\": %zu, \"stream16_known_bytes\": %zu, \"table_bytes\": %zu}\n", gather_bytes, stream_bytes, TABLE_ROWS * 64);
    return 0;
}
