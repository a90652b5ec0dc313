// Internal declarations shared by the translation units of libdipgenie_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/dipgenie_hip.h"

namespace dgi {

void set_error(const char *fmt, ...);

#define DG_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            dgi::set_error("%s failed at %s:%d: %s", #call, __FILE__, __LINE__, hipGetErrorString(e_)); \
            return e_ == hipErrorOutOfMemory ? DG_ERR_OOM : DG_ERR_HIP;                       \
        }                                                                                     \
    } while (0)

struct DevBuf {                       // owning device allocation
    void *p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
    int ensure(size_t n) {            // grow-only
        if (n <= bytes && p) return DG_OK;
        release();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) { p = nullptr; set_error("hipMalloc(%zu bytes) failed: %s", n, hipGetErrorString(e)); return DG_ERR_OOM; }
        bytes = n;
        return DG_OK;
    }
    template <class T> T *as() const { return (T *)p; }
};

struct DpState;       // dg_dp_*.hip
struct SketchState;   // dg_sketch.hip
struct AnchorState;   // dg_anchor.hip

}  // namespace dgi

struct dg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    hipDeviceProp_t prop;
    dgi::DpState *dp = nullptr;
    dgi::SketchState *sk = nullptr;
    dgi::AnchorState *an = nullptr;
};

namespace dgi {
void dp_state_free(DpState *);
void sketch_state_free(SketchState *);
void anchor_state_free(AnchorState *);
// dg_sketch.hip: minimizer list of one haplotype left on the device (valid until the next sketch call on the ctx)
int sketch_haplotype_dev(dg_ctx *c, const char *seq, int64_t len, int k, int w, const uint64_t **hash_dev, const int64_t **pos_dev, int64_t *n);
inline int bind(dg_ctx *c) {
    if (!c) { set_error("null ctx"); return DG_ERR_ARG; }
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) { set_error("hipSetDevice(%d): %s", c->device, hipGetErrorString(e)); return DG_ERR_NO_DEVICE; }
    return DG_OK;
}
}  // namespace dgi
