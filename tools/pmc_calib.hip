// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access widths the DP sweep uses (the guide
// calibrates them for 16 B/lane streams only).  Every kernel moves exactly 1 GiB of a fresh buffer, so
// counter / 2^20 KiB is the factor by which the counter mis-states that pattern.
//   hipcc --offload-arch=gfx950 -O2 tools/pmc_calib.hip -o bin/pmc_calib
//   rocprofv3 --pmc FETCH_SIZE -- bin/pmc_calib      (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr size_t GIB = (size_t)1 << 30;

__global__ void rd_b32(const uint32_t *p, size_t n, uint32_t *sink) {         // 4 B per lane, coalesced (value gathers)
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345679u) *sink = acc;
}
__global__ void rd_b16(const uint16_t *p, size_t n, uint32_t *sink) {         // 2 B per lane, coalesced (delta reads)
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x5679u) *sink = acc;                                          // (a 16-bit xor never has high bits)
}
__global__ void rd_b128(const uint4 *p, size_t n, uint32_t *sink) {           // 16 B per lane (the guide's calibrated case)
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345679u) *sink = acc;
}
// rows of `seg` dwords read by the first `seg` lanes of a wave (a column group narrower than the wave)
__global__ void rd_b32_seg(const uint32_t *p, size_t nrows, int seg, uint32_t *sink) {
    uint32_t acc = 0;
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t r = wave; r < nrows; r += nw) if (lane < seg) acc ^= p[r * seg + lane];
    if (acc == 0x12345679u) *sink = acc;
}
__global__ void wr_b32(uint32_t *p, size_t n) {                                // 4 B per lane, plain store (values)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
__global__ void wr_b32_nt(uint32_t *p, size_t n) {                             // 4 B per lane, non-temporal (back-pointers)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store((uint32_t)i, &p[i]);
}
// segment heads only: `seg` consecutive dwords per wave instruction, written by lanes 0, 3, 6, ... (3 in-edges per column)
__global__ void wr_b32_heads(uint32_t *p, size_t nrows, int seg, int nt) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t r = wave; r < nrows; r += nw)
        if (lane % 3 == 0 && lane / 3 < seg) { if (nt) __builtin_nontemporal_store((uint32_t)r, &p[r * seg + lane / 3]); else p[r * seg + lane / 3] = (uint32_t)r; }
}

// the 16-bit back-pointer stream: 2 B per lane plain / non-temporal, and segment heads only
__global__ void wr_b16(uint16_t *p, size_t n, int nt) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (nt) asm volatile("global_store_short %0, %1, off nt" ::"v"(p + i), "v"((uint32_t)i) : "memory"); else p[i] = (uint16_t)i;
    }
}
__global__ void wr_b16_heads(uint16_t *p, size_t nrows, int seg, int nt) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t r = wave; r < nrows; r += nw)
        if (lane % 3 == 0 && lane / 3 < seg) {
            uint16_t *q = &p[r * seg + lane / 3];
            if (nt) asm volatile("global_store_short %0, %1, off nt" ::"v"(q), "v"((uint32_t)r) : "memory"); else *q = (uint16_t)r;
        }
}

int main() {
    void *buf = nullptr; uint32_t *sink = nullptr;
    CK(hipMalloc(&buf, GIB)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, GIB)); CK(hipDeviceSynchronize());
    const dim3 grid(256 * 16), block(256);
    hipLaunchKernelGGL(rd_b32, grid, block, 0, 0, (const uint32_t *)buf, GIB / 4, sink); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(rd_b16, grid, block, 0, 0, (const uint16_t *)buf, GIB / 2, sink); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(rd_b128, grid, block, 0, 0, (const uint4 *)buf, GIB / 16, sink); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(rd_b32_seg, grid, block, 0, 0, (const uint32_t *)buf, GIB / 4 / 21, 21, sink); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(wr_b32, grid, block, 0, 0, (uint32_t *)buf, GIB / 4); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(wr_b32_nt, grid, block, 0, 0, (uint32_t *)buf, GIB / 4); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(wr_b32_heads, grid, block, 0, 0, (uint32_t *)buf, GIB / 4 / 21, 21, 0); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(wr_b32_heads, grid, block, 0, 0, (uint32_t *)buf, GIB / 4 / 21, 21, 1); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(wr_b16, grid, block, 0, 0, (uint16_t *)buf, GIB / 2, 0); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(wr_b16, grid, block, 0, 0, (uint16_t *)buf, GIB / 2, 1); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(wr_b16_heads, grid, block, 0, 0, (uint16_t *)buf, GIB / 2 / 21, 21, 0); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(wr_b16_heads, grid, block, 0, 0, (uint16_t *)buf, GIB / 2 / 21, 21, 1); CK(hipDeviceSynchronize());
    printf("done: every kernel moved 1 GiB (the *_seg / *_heads ones %zu bytes)\n", (GIB / 4 / 21) * 21 * 4);
    return 0;
}
