// Device-side helpers shared by the sweep's translation units (dg_dp_sweep.hip; dg_dp_sweep_sym.hip in the measurement build).
#pragma once
#include "dg_dp.hpp"

namespace dgi {

// state slot of a level: the two slots are ping-pong buffers
#define DG_SLOT(ARGS, LEVEL) ((LEVEL) & 1)

constexpr unsigned long long DIGEST_PRED_MUL = 0x9E3779B97F4A7C15ULL;   // oracle_dp.cpp: weight of the predecessor term

// narrow form: ord = (255 - eu) << 8 | (255 - ev) is never 0 for a real candidate, and the stored back-pointer is
// simply ~ord (an untouched best keeps ord 0 -> 0xFFFF = unreachable)
// non-temporal 16-bit store as inline asm: with the builtin on one side of a branch and a plain store on the other the
// optimiser merges the two into ONE plain store (the !nontemporal hint is dropped)
__device__ __forceinline__ void store_bp_nt(uint16_t *p, uint32_t v) { asm volatile("global_store_short %0, %1, off nt" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ uint32_t ord_rank(int eu, int ev) { return ((uint32_t)(BP_MAX_RANK - eu) << 8) | (uint32_t)(BP_MAX_RANK - ev); }

// digest of one reachable cell: the oracle's definition (oracle_dp.cpp), o = its r-major cell index
__device__ __forceinline__ unsigned long long digest_term(int value, unsigned long long o, uint32_t pred_i, uint32_t pred_j) {
    return (unsigned long long)(uint32_t)(value + 1) * (o + 1) +
           DIGEST_PRED_MUL * ((((unsigned long long)pred_i << 15) | pred_j) + 1ULL) * (o + 1);
}

// neighbour exchange by one lane as DPP wave shifts (a few cycles) instead of ds_bpermute (an LDS crossbar round trip):
// most column groups need exactly one step of the segmented max (columns with at most two in-edges)
__device__ __forceinline__ int lane_down1(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x130 /* wave_shl:1: lane i <- lane i + 1 */, 0xF, 0xF, false); }
__device__ __forceinline__ int lane_up1(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x138 /* wave_shr:1: lane i <- lane i - 1 */, 0xF, 0xF, false); }

// What a task needs before its stores, as leading scalar kernel arguments: the command processor preloads the first 16
// dwords of scalar arguments into SGPRs (-mllvm -amdgpu-kernarg-preload-count; by-value structs stop the preload), so
// neither load round waits for a load of the kernel-argument segment (a cold miss on every CU at every launch: 0.3-0.4 us
// per level when the first round had to wait for it).  Plain launches preload everything below; cooperative launches
// spend five of the sixteen dwords on the heavy-row list and read {dm, pad_bytes, dT, buf_bytes} the ordinary way.
struct LevelHead {
    const uint4 *rowrec_l;              // rowrec + b0
    const uint2 *slots_l;               // slots + slot_first
    const uint32_t *rowx_l;             // rowx + rowx_off
    const int32_t *cur;                 // padded start of the source level's state buffer
    const uint16_t *dm;                 // delta matrix biased by -in_base * dT (entry of in-edge pair (e_u, e_v): dm[e_u * dT + dcol]); the zero slot if dT = 0
    int rowx_stride, RP, k, pad_bytes, dT;
    uint32_t buf_bytes;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t state_rsrc(const int32_t *padded_base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)padded_base, 0, bytes, 0x00020000);
}

// per-level launch of the fast form: grid = (ceil(nblocks/4), nchunk, k2 [+ 4 per heavy row]), one task per wave (a 3-D
// grid: splitting a combined index would cost a runtime integer division -- ~40 instructions of a 380-instruction task)
// workgroup 0 of a launch tells the L2 prefetcher which level is running
__device__ __forceinline__ void publish_level(int *progress, int lvl) {
    if ((blockIdx.x | blockIdx.y | blockIdx.z) == 0 && threadIdx.x == 0) __hip_atomic_store(progress, lvl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace dgi
