// Measurement (not product): does a value written by one kernel survive the kernel boundary in the WRITER XCD's L2, so that a
// consumer workgroup placed on the same XCD reads it faster than one on another XCD?  And is the block -> XCD dealing stable from
// launch to launch (block b of consecutive launches on the same XCD)?  The DP sweep is a chain of dependent launches whose tasks
// gather what the previous launch stored: if both hold, giving a destination row to the XCD that produced its source row would
// turn the state gathers (0.6-0.7 us, served by the Infinity Cache today) into L2 hits.
//   hipcc --offload-arch=gfx950 -O2 tools/xcd_affinity.hip -o bin/xcd_affinity && bin/xcd_affinity [G] [iters] [words_per_block]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ int xcc_id() { return (int)__builtin_amdgcn_s_getreg((3 << 11) | 20); }

// one launch of the chain: block b reads the region that block src(b) of the PREVIOUS launch wrote (one dependent load round,
// timed by wave 0 with the 100 MHz counter), checks the tag, and writes its own region of the other buffer
__global__ __launch_bounds__(256) void k_step(const uint32_t *__restrict__ cur, uint32_t *__restrict__ nxt, int words, int shift, uint32_t tag,
                                              int *__restrict__ xcc_of_block, unsigned long long *__restrict__ lat_same, unsigned long long *__restrict__ lat_cross,
                                              unsigned int *__restrict__ n_same, unsigned int *__restrict__ n_cross, unsigned int *__restrict__ bad,
                                              const int *__restrict__ xcc_prev) {
    const int b = (int)blockIdx.x, G = (int)gridDim.x;
    const int src = (b + shift) % G;
    const int me = xcc_id();
    const uint32_t *p = cur + (size_t)src * words;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
    for (int t = (int)threadIdx.x; t < words; t += 256) acc |= p[t] ^ (tag - 1);          // every word must carry the previous launch's tag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (acc != 0 && tag > 1) atomicAdd(bad, 1u);
    if (threadIdx.x == 0 && tag > 1) {
        if (xcc_prev[src] == me) { atomicAdd(lat_same, t1 - t0); atomicAdd(n_same, 1u); }
        else { atomicAdd(lat_cross, t1 - t0); atomicAdd(n_cross, 1u); }
    }
    for (int t = (int)threadIdx.x; t < words; t += 256) nxt[(size_t)b * words + t] = tag;
    if (threadIdx.x == 0) xcc_of_block[b] = me;
}

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 1024, iters = argc > 2 ? atoi(argv[2]) : 2000, words = argc > 3 ? atoi(argv[3]) : 1024;
    uint32_t *s[2];
    int *xcc[2];
    unsigned long long *lat;
    unsigned int *cnt;
    CK(hipMalloc(&s[0], (size_t)G * words * 4)); CK(hipMalloc(&s[1], (size_t)G * words * 4));
    CK(hipMalloc(&xcc[0], G * 4)); CK(hipMalloc(&xcc[1], G * 4));
    CK(hipMalloc(&lat, 16)); CK(hipMalloc(&cnt, 12));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int shift : {0, 1, 8, 9}) {
        CK(hipMemset(lat, 0, 16)); CK(hipMemset(cnt, 0, 12)); CK(hipMemset(xcc[0], 0xFF, G * 4)); CK(hipMemset(xcc[1], 0xFF, G * 4));
        std::vector<int> prev(G, -1), now(G);
        long long stable = 0, total = 0;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int it = 0; it < iters; ++it)
            hipLaunchKernelGGL(k_step, dim3(G), dim3(256), 0, st, s[it & 1], s[(it + 1) & 1], words, shift, (uint32_t)(it + 1), xcc[(it + 1) & 1], lat, lat + 1, cnt, cnt + 1, cnt + 2,
                               xcc[it & 1]);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        // stability of the dealing: a short second chain with a download after every launch
        for (int it = 0; it < 16; ++it) {
            hipLaunchKernelGGL(k_step, dim3(G), dim3(256), 0, st, s[it & 1], s[(it + 1) & 1], words, shift, 1u, xcc[(it + 1) & 1], lat, lat + 1, cnt, cnt + 1, cnt + 2, xcc[it & 1]);
            CK(hipMemcpyAsync(now.data(), xcc[(it + 1) & 1], G * 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
            if (it > 0) for (int b = 0; b < G; ++b) { stable += now[b] == prev[b]; ++total; }
            prev = now;
        }
        unsigned long long hl[2]; unsigned int hc[3];
        CK(hipMemcpy(hl, lat, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc, cnt, 12, hipMemcpyDeviceToHost));
        printf("G %d, %d words/block, reader of block b reads what block b+%d wrote: %.3f us per launch; same-XCD reads %u (%.0f ns), cross-XCD reads %u (%.0f ns); stale words seen by %u blocks; "
               "block b on the same XCD in consecutive launches: %.1f %%; block 0..15 XCDs:", G, words, shift, 1e3 * ms / iters, hc[0], hc[0] ? 10.0 * hl[0] / hc[0] : 0.0, hc[1],
               hc[1] ? 10.0 * hl[1] / hc[1] : 0.0, hc[2], total ? 100.0 * stable / total : 0.0);
        for (int b = 0; b < 16 && b < G; ++b) printf(" %d", now[b]);
        printf("\n");
    }
    return 0;
}
