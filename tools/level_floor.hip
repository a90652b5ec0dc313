// What a chain of dependent per-level launches costs on this device, piece by piece -- the floor under the DP sweep
// (one launch per graph level, every level reads what the previous launch wrote).  Each variant issues N launches
// back to back on one stream and reports microseconds per launch (HIP events around the whole chain):
//   empty1      1 workgroup, empty body, 80-byte by-value argument block (the sweep passes its LevelDesc that way)
//   emptyG      G workgroups of 256 threads, empty body                     (wave dispatch of a typical level)
//   round1      G workgroups; every wave: load a record, store a value      (one memory round + store)
//   round2      ... load a record, then the value of the previous launch at the index found there, store  (two dependent rounds:
//               the sweep's common task -- row/slot record, then state values)
//   round2cold  round2, the records of every launch come from a fresh region (HBM misses, like the sweep's tables)
//   round3..5   more dependent rounds (rows with 9-24 in-edges walk them in steps of eight)
//   hipcc --offload-arch=gfx950 -O2 -mllvm -amdgpu-kernarg-preload-count=16 tools/level_floor.hip -o bin/level_floor && bin/level_floor [N] [G]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Desc { int32_t w[20]; };                                     // 80 bytes, by value

__global__ __launch_bounds__(256) void k_empty(Desc d) { if (d.w[0] == 0x7fffffff) __builtin_trap(); }

// ROUNDS dependent loads: rec[] holds indices into the state; after the first record load every further round loads the
// state word at the index obtained in the previous round (the state of the previous launch), then one store.
template <int ROUNDS>
__global__ __launch_bounds__(256) void k_rounds(Desc d, const uint32_t *__restrict__ rec, const uint32_t *__restrict__ cur, uint32_t *__restrict__ nxt, uint32_t mask) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    uint32_t x = rec[(size_t)d.w[1] + t];                          // round 1: record (table read, never written)
#pragma unroll
    for (int r = 1; r < ROUNDS; ++r) x = cur[(x + (uint32_t)r * 64u) & mask];   // rounds 2..: values written by the previous launch (coalesced: a wave reads 64 consecutive words)
    nxt[t & mask] = (x + 4096u) & mask;                            // keeps lanes consecutive: x = wave base + lane throughout
}

// The same body with the pointers it needs first in the argument list and the file compiled with
// -mllvm -amdgpu-kernarg-preload-count=16: the command processor hands the leading 16 dwords of scalar arguments to every
// wave in SGPRs, so the first load does not wait for a load of the kernarg segment (struct arguments stop the preload).
template <int ROUNDS>
__global__ __launch_bounds__(256) void k_rounds_pre(const uint32_t *__restrict__ rec, const uint32_t *__restrict__ cur, uint32_t *__restrict__ nxt, uint32_t mask, Desc d) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    uint32_t x = rec[t];
#pragma unroll
    for (int r = 1; r < ROUNDS; ++r) x = cur[(x + (uint32_t)r * 64u) & mask];
    nxt[t & mask] = (x + 4096u + (uint32_t)d.w[3]) & mask;
}

// Persistent alternative: one resident grid loops over the levels; hand-off = agent-scope release of this workgroup's
// stores, one atomic counter, agent-scope (cache-bypassing) loads of the state.  Spins are bounded: a workgroup that waits
// too long raises `abort` and every workgroup leaves.
struct Ctl { uint32_t count, abort, pad[14]; };
template <int ROUNDS>
__global__ __launch_bounds__(256) void k_persist(int n_iter, const uint32_t *__restrict__ rec, uint32_t *s0, uint32_t *s1, uint32_t mask, Ctl *ctl, int cold_stride) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    for (int it = 0; it < n_iter; ++it) {
        const uint32_t *cur = (it & 1) ? s1 : s0;
        uint32_t *nxt = (it & 1) ? s0 : s1;
        uint32_t x = rec[(size_t)(it & 4095) * cold_stride + t];
#pragma unroll
        for (int r = 1; r < ROUNDS; ++r) x = __hip_atomic_load(&cur[(x + (uint32_t)r * 64u) & mask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&nxt[t & mask], (x + 4096u) & mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();                                            // (compiler waits for this wave's stores before the barrier)
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&ctl->count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t want = gridDim.x * (uint32_t)(it + 1);
            uint32_t spin = 0;
            while (__hip_atomic_load(&ctl->count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++spin > (1u << 20) || __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __syncthreads();
        if (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    }
}

// Chained levels: ONE dispatch covers M levels of G workgroups each; a workgroup of level q does its first (independent)
// load round, then waits until all G workgroups of level q - 1 have signalled, reads their values with agent-scope loads,
// stores with agent-scope stores and signals.  Workgroups are dispatched in id order (per XCD), so every workgroup a
// spinner waits for is already placed or ahead of it in its XCD's queue -- the forward-progress argument of decoupled
// look-back scans.  Spins are bounded (abort flag).  What it would buy the sweep: the drain of level q overlaps the
// dispatch ramp and first load round of level q + 1, and there is no release / acquire of whole L2s in between.
template <int ROUNDS>
__global__ __launch_bounds__(256) void k_chain(int G, int level0, const uint32_t *__restrict__ rec, uint32_t *s0, uint32_t *s1, uint32_t mask, uint32_t *done, Ctl *ctl, int cold_stride) {
    const int q = (int)blockIdx.x / G, w = (int)blockIdx.x % G, lvl = level0 + q;
    const uint32_t t = (uint32_t)w * 256u + threadIdx.x;
    const uint32_t *cur = (lvl & 1) ? s1 : s0;
    uint32_t *nxt = (lvl & 1) ? s0 : s1;
    uint32_t x = rec[(size_t)(lvl & 4095) * cold_stride + t];
    if (q > 0) {                                                      // (level0 follows a kernel boundary)
        if (threadIdx.x == 0) {
            uint32_t spin = 0;
            while (__hip_atomic_load(&done[lvl - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)G) {
                __builtin_amdgcn_s_sleep(1);
                if (++spin > (1u << 18) || __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 1; r < ROUNDS; ++r) x = __hip_atomic_load(&cur[(x + (uint32_t)r * 64u) & mask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&nxt[t & mask], (x + 4096u) & mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&done[lvl], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// Chained levels with PER-PRODUCER flags (round 3): the consumer wave polls only the flag of the ONE workgroup of the previous level
// whose 1 KB region it gathers from -- no counter shared by a whole level.  Producer: `sc1` (write-through) stores, every wave's
// s_waitcnt vmcnt(0), workgroup barrier, one `sc1` flag store on a 128-byte line of its own; consumer: `sc1` poll, then `sc1` gathers
// (MI355X_MICROARCH.md, "hand-offs measured with sc1 loads in place of the acquire", first row).  The state lives in a ring of RING
// buffers so that a line is rewritten only RING levels later (ping-pong would leave a two-level-old copy in the reader XCD's L2, which
// `sc1` loads are served from).  Spins are bounded (abort flag); workgroups of level q have higher block ids than those of level
// q - 1, so what a spinner waits for is resident or ahead of it in the dispatch order.
constexpr int RING = 8;
struct RingBufs { uint32_t *b[RING]; };
template <int ROUNDS>
__global__ __launch_bounds__(256) void k_chain_rows(int G, int level0, const uint32_t *__restrict__ rec, RingBufs st, uint32_t mask, uint32_t *flags /* [RING][G][32] */, Ctl *ctl, int cold_stride) {
    const int q = (int)blockIdx.x / G, w = (int)blockIdx.x % G, lvl = level0 + q;
    const uint32_t t = (uint32_t)w * 256u + threadIdx.x;
    const uint32_t *cur = st.b[lvl % RING];
    uint32_t *nxt = st.b[(lvl + 1) % RING];
    uint32_t x = rec[(size_t)(lvl & 4095) * cold_stride + t];          // round 1: independent of the previous level
#pragma unroll
    for (int r = 1; r < ROUNDS; ++r) {
        const uint32_t idx = (x + (uint32_t)r * 64u) & mask;
        if (q > 0) {                                                    // (level0 follows a kernel boundary: everything visible)
            const uint32_t prod = __builtin_amdgcn_readfirstlane(idx >> 8);   // the producer workgroup of this wave's 64 words
            const uint32_t *f = flags + ((size_t)(lvl % RING) * G + prod) * 32;
            uint32_t spin = 0;
            while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (uint32_t)lvl) {
                __builtin_amdgcn_s_sleep(1);
                if (++spin > (1u << 16) || __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        x = __hip_atomic_load(&cur[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __hip_atomic_store(&nxt[t & mask], (x + 4096u) & mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flags + ((size_t)((lvl + 1) % RING) * G + w) * 32, (uint32_t)(lvl + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the same levels as ONE launch each over the same ring (reference for time and for the final state)
template <int ROUNDS>
__global__ __launch_bounds__(256) void k_ring_level(int lvl, const uint32_t *__restrict__ rec, RingBufs st, uint32_t mask, int cold_stride) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t *cur = st.b[lvl % RING];
    uint32_t *nxt = st.b[(lvl + 1) % RING];
    uint32_t x = rec[(size_t)(lvl & 4095) * cold_stride + t];
#pragma unroll
    for (int r = 1; r < ROUNDS; ++r) x = cur[(x + (uint32_t)r * 64u) & mask];
    nxt[t & mask] = (x + 4096u) & mask;
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 20000;
    const int G = argc > 2 ? atoi(argv[2]) : 1024;                  // workgroups per launch (MHC-24 levels: 800-2200)
    const uint32_t words = 1u << 18, mask = words - 1;              // 1 MB of state, like an average MHC-24 level
    CK(hipSetDevice(0));
    hipStream_t s; CK(hipStreamCreate(&s));
    uint32_t *rec, *st[2];
    const size_t rec_words = (size_t)G * 256 * 4096;                // "cold": 4096 distinct record regions, cycled
    CK(hipMalloc(&rec, rec_words * 4)); CK(hipMalloc(&st[0], words * 4)); CK(hipMalloc(&st[1], words * 4));
    {
        std::vector<uint32_t> h((size_t)G * 256);
        for (size_t i = 0; i < h.size(); ++i) h[i] = ((((uint32_t)(i >> 6) * 2654435761u) & mask & ~63u) | (uint32_t)(i & 63));   // wave base + lane
        for (size_t o = 0; o < rec_words; o += h.size()) CK(hipMemcpy(rec + o, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        { std::vector<uint32_t> z(words); for (uint32_t i = 0; i < words; ++i) z[i] = i; CK(hipMemcpy(st[0], z.data(), words * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(st[1], z.data(), words * 4, hipMemcpyHostToDevice)); }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) -> int {
        for (int rep = 0; rep < 2; ++rep) {                         // first repetition warms clocks and code
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < N; ++i) launch(i);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%-11s %7.3f us/launch  (%d launches, %d workgroups)\n", name, 1e3 * ms / N, N, G);
        }
        return 0;
    };
    Desc d{};
    if (run("empty1", [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, s, d); })) return 1;
    if (run("emptyG", [&](int) { hipLaunchKernelGGL(k_empty, dim3(G), dim3(256), 0, s, d); })) return 1;
#define ROUND(NAME, R, COLD) if (run(NAME, [&](int i) { Desc q = d; q.w[1] = (COLD) ? (int)((size_t)(i & 4095) * G * 256) : 0; \
        hipLaunchKernelGGL((k_rounds<R>), dim3(G), dim3(256), 0, s, q, rec, st[i & 1], st[(i + 1) & 1], mask); })) return 1
    ROUND("round1", 1, 0);
    ROUND("round2", 2, 0);
    ROUND("round2cold", 2, 1);
    ROUND("round3", 3, 0);
    ROUND("round4", 4, 0);
    ROUND("round5", 5, 0);
    ROUND("round5cold", 5, 1);
    // the same chains replayed from a hipGraph (2,000 kernel nodes captured once, launched 10 times): no per-launch host work
    auto graph = [&](const char *name, auto launch) -> int {
        const int M = 2000;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < M; ++i) launch(i);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("graph %-11s %7.3f us/node   (10 x %d nodes)\n", name, 1e3 * ms / (10.0 * M), M);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        return 0;
    };
    if (graph("empty1", [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, s, d); })) return 1;
    if (graph("emptyG", [&](int) { hipLaunchKernelGGL(k_empty, dim3(G), dim3(256), 0, s, d); })) return 1;
    if (graph("round2", [&](int i) { hipLaunchKernelGGL((k_rounds<2>), dim3(G), dim3(256), 0, s, d, rec, st[i & 1], st[(i + 1) & 1], mask); })) return 1;
    if (graph("round2pre", [&](int i) { hipLaunchKernelGGL((k_rounds_pre<2>), dim3(G), dim3(256), 0, s, rec, st[i & 1], st[(i + 1) & 1], mask, d); })) return 1;
    if (graph("round5pre", [&](int i) { hipLaunchKernelGGL((k_rounds_pre<5>), dim3(G), dim3(256), 0, s, rec, st[i & 1], st[(i + 1) & 1], mask, d); })) return 1;
    if (graph("round2pre/64", [&](int i) { hipLaunchKernelGGL((k_rounds_pre<2>), dim3(64), dim3(256), 0, s, rec, st[i & 1], st[(i + 1) & 1], mask, d); })) return 1;
    if (run("round2pre", [&](int i) { hipLaunchKernelGGL((k_rounds_pre<2>), dim3(G), dim3(256), 0, s, rec, st[i & 1], st[(i + 1) & 1], mask, d); })) return 1;
    if (graph("round5", [&](int i) { hipLaunchKernelGGL((k_rounds<5>), dim3(G), dim3(256), 0, s, d, rec, st[i & 1], st[(i + 1) & 1], mask); })) return 1;
    for (int g2 : {16, 64, 256, 512, 2048, 4096}) {
        char nm[32]; snprintf(nm, sizeof nm, "round2/%dwg", g2);
        if (g2 > G && g2 * 256 > (int)(rec_words / 4096)) continue;
        if (graph(nm, [&](int i) { hipLaunchKernelGGL((k_rounds<2>), dim3(g2), dim3(256), 0, s, d, rec, st[i & 1], st[(i + 1) & 1], mask); })) return 1;
        if (run(nm, [&](int i) { hipLaunchKernelGGL((k_rounds<2>), dim3(g2), dim3(256), 0, s, d, rec, st[i & 1], st[(i + 1) & 1], mask); })) return 1;
    }
    // Two bands on two streams: band B's launch of level l waits (event) for band A's launch of level l, which it reads;
    // both bands run half the workgroups.  A skewed pipeline of the recombination-count ranges of the DP would look like this.
    {
        hipStream_t s2; CK(hipStreamCreate(&s2));
        const int NB = 4000;
        std::vector<hipEvent_t> evs(NB);
        for (auto &evq : evs) CK(hipEventCreateWithFlags(&evq, hipEventDisableTiming));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < NB; ++i) {
                hipLaunchKernelGGL((k_rounds<2>), dim3(G / 2), dim3(256), 0, s, d, rec, st[i & 1], st[(i + 1) & 1], mask);
                CK(hipEventRecord(evs[i], s));
                CK(hipStreamWaitEvent(s2, evs[i], 0));
                hipLaunchKernelGGL((k_rounds<2>), dim3(G / 2), dim3(256), 0, s2, d, rec + (size_t)G * 128, st[i & 1] + 131072, st[(i + 1) & 1] + 131072, mask >> 1);
            }
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("bands2      %7.3f us/level   (%d levels, 2 streams x %d workgroups, event per level)\n", 1e3 * ms / NB, NB, G / 2);
        }
        // the same pattern captured once into a hipGraph (fork/join through events) and replayed: no host work per level
        {
            const int M = 2000;
            hipGraph_t g; hipGraphExec_t ge;
            hipEvent_t fork, join; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            CK(hipEventRecord(fork, s)); CK(hipStreamWaitEvent(s2, fork, 0));
            for (int i = 0; i < M; ++i) {
                hipLaunchKernelGGL((k_rounds<2>), dim3(G / 2), dim3(256), 0, s, d, rec, st[i & 1], st[(i + 1) & 1], mask);
                CK(hipEventRecord(evs[i], s));
                CK(hipStreamWaitEvent(s2, evs[i], 0));
                hipLaunchKernelGGL((k_rounds<2>), dim3(G / 2), dim3(256), 0, s2, d, rec + (size_t)G * 128, st[i & 1] + 131072, st[(i + 1) & 1] + 131072, mask >> 1);
            }
            CK(hipEventRecord(join, s2)); CK(hipStreamWaitEvent(s, join, 0));
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("graph bands2 %6.3f us/level   (5 x %d levels, 2 branches x %d workgroups, edge per level)\n", 1e3 * ms / (5.0 * M), M, G / 2);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
        // reference: two fully independent chains on the two streams
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < NB; ++i) {
                hipLaunchKernelGGL((k_rounds<2>), dim3(G / 2), dim3(256), 0, s, d, rec, st[i & 1], st[(i + 1) & 1], mask >> 1);
                hipLaunchKernelGGL((k_rounds<2>), dim3(G / 2), dim3(256), 0, s2, d, rec + (size_t)G * 128, st[i & 1] + 131072, st[(i + 1) & 1] + 131072, mask >> 1);
            }
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("indep2      %7.3f us/level   (%d levels, 2 independent streams x %d workgroups)\n", 1e3 * ms / NB, NB, G / 2);
        }
        for (auto &evq : evs) CK(hipEventDestroy(evq));
        CK(hipStreamDestroy(s2));
    }
    if (argc > 4) return 0;                                         // a fourth argument: stop here
    Ctl *ctl; CK(hipMalloc(&ctl, sizeof(Ctl)));
    {   // chained levels in one dispatch (k_chain), M levels per dispatch
        uint32_t *done; CK(hipMalloc(&done, 4 * (size_t)(N + 64)));
        for (int M : {2, 4, 16, 64}) {
            for (int rounds : {2, 5}) {
                float best = 1e30f; uint32_t aborted = 0;
                const int NL = (N / M) * M;
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipMemsetAsync(ctl, 0, sizeof(Ctl), s));
                    CK(hipMemsetAsync(done, 0, 4 * (size_t)(N + 64), s));
                    CK(hipEventRecord(e0, s));
                    for (int l0 = 0; l0 < NL; l0 += M) {
                        if (rounds == 2) hipLaunchKernelGGL((k_chain<2>), dim3(G * M), dim3(256), 0, s, G, l0, rec, st[0], st[1], mask, done, ctl, G * 256);
                        else hipLaunchKernelGGL((k_chain<5>), dim3(G * M), dim3(256), 0, s, G, l0, rec, st[0], st[1], mask, done, ctl, G * 256);
                    }
                    CK(hipEventRecord(e1, s));
                    CK(hipStreamSynchronize(s));
                    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                    Ctl h; CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
                    aborted |= h.abort;
                    if (ms < best) best = ms;
                }
                printf("chain%d x%-3d %7.3f us/level   (%d levels, %d workgroups per level, %d levels per dispatch)%s\n", rounds, M, 1e3 * best / NL, NL, G, M,
                       aborted ? "  ABORTED (spin timeout)" : "");
            }
        }
        CK(hipFree(done));
    }
    if (G * 256 == (int)words) {   // chained levels with per-producer flags (k_chain_rows); needs one workgroup per 256 state words
        RingBufs rb;
        for (int q = 0; q < RING; ++q) CK(hipMalloc(&rb.b[q], words * 4));
        uint32_t *flags; CK(hipMalloc(&flags, (size_t)RING * G * 32 * 4));
        std::vector<uint32_t> init(words), ref(words), got(words);
        for (uint32_t i = 0; i < words; ++i) init[i] = i;
        auto reset = [&]() -> int { for (int q = 0; q < RING; ++q) CK(hipMemcpy(rb.b[q], init.data(), words * 4, hipMemcpyHostToDevice)); CK(hipMemset(flags, 0xFF, (size_t)RING * G * 32 * 4)); return 0; };
        const int NL = (std::min(N, 4096) / 64) * 64;
        for (int rounds : {2, 5}) {
            if (reset()) return 1;
            CK(hipEventRecord(e0, s));
            for (int l = 0; l < NL; ++l) {
                if (rounds == 2) hipLaunchKernelGGL((k_ring_level<2>), dim3(G), dim3(256), 0, s, l, rec, rb, mask, G * 256);
                else hipLaunchKernelGGL((k_ring_level<5>), dim3(G), dim3(256), 0, s, l, rec, rb, mask, G * 256);
            }
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(ref.data(), rb.b[NL % RING], words * 4, hipMemcpyDeviceToHost));
            printf("ring%d       %7.3f us/level   (%d levels, one launch per level, ring of %d state buffers)\n", rounds, 1e3 * ms / NL, NL, RING);
            for (int M : {4, 16, 64}) {
                float best = 1e30f; uint32_t aborted = 0; size_t wrong = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    if (reset()) return 1;
                    CK(hipMemsetAsync(ctl, 0, sizeof(Ctl), s));
                    CK(hipEventRecord(e0, s));
                    for (int l0 = 0; l0 < NL; l0 += M) {
                        if (rounds == 2) hipLaunchKernelGGL((k_chain_rows<2>), dim3(G * M), dim3(256), 0, s, G, l0, rec, rb, mask, flags, ctl, G * 256);
                        else hipLaunchKernelGGL((k_chain_rows<5>), dim3(G * M), dim3(256), 0, s, G, l0, rec, rb, mask, flags, ctl, G * 256);
                    }
                    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    Ctl h; CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
                    aborted |= h.abort;
                    CK(hipMemcpy(got.data(), rb.b[NL % RING], words * 4, hipMemcpyDeviceToHost));
                    wrong = 0; for (uint32_t i = 0; i < words; ++i) wrong += got[i] != ref[i];
                    if (ms < best) best = ms;
                }
                printf("rowflag%d x%-3d %6.3f us/level   (%d levels, %d workgroups per level, %d levels per dispatch; final state: %zu of %u words differ)%s\n", rounds, M, 1e3 * best / NL, NL, G, M,
                       wrong, words, aborted ? "  ABORTED (spin timeout)" : "");
            }
        }
        CK(hipFree(flags));
        for (int q = 0; q < RING; ++q) CK(hipFree(rb.b[q]));
    }
    if (argc > 3) return 0;                                         // a third argument: skip the persistent variants
    // persistent grid + device-wide barrier per level (all workgroups resident: at most 2 per CU)
    int dev_cus = 0; CK(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, 0));
    for (int PG : {dev_cus / 8, dev_cus, 2 * dev_cus}) {
        for (int rounds : {2, 5}) {
            for (int cold = 0; cold < 2; ++cold) {
                float best = 1e30f; uint32_t aborted = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipMemsetAsync(ctl, 0, sizeof(Ctl), s));
                    CK(hipEventRecord(e0, s));
                    if (rounds == 2) hipLaunchKernelGGL((k_persist<2>), dim3(PG), dim3(256), 0, s, N, rec, st[0], st[1], mask, ctl, cold ? PG * 256 : 0);
                    else hipLaunchKernelGGL((k_persist<5>), dim3(PG), dim3(256), 0, s, N, rec, st[0], st[1], mask, ctl, cold ? PG * 256 : 0);
                    CK(hipEventRecord(e1, s));
                    CK(hipStreamSynchronize(s));
                    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                    Ctl h; CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
                    aborted |= h.abort;
                    if (ms < best) best = ms;
                }
                printf("persist%d%s %7.3f us/level   (%d levels, %d resident workgroups)%s\n", rounds, cold ? "cold" : "    ", 1e3 * best / N, N, PG,
                       aborted ? "  ABORTED (barrier timeout)" : "");
            }
        }
    }
    return 0;
}
