// libdipgenie_hip.so -- context management and error plumbing of the C ABI (include/dipgenie_hip.h).
#include <cstdlib>
#include <cstring>

#include "dg_internal.hpp"

namespace dgi {
static thread_local std::string g_err;
void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}
}  // namespace dgi

extern "C" const char *dg_last_error(void) { return dgi::g_err.c_str(); }

extern "C" dg_ctx *dg_create(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        dgi::set_error("no HIP device available (%s); libdipgenie_hip has no CPU fallback",
                       e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return nullptr;
    }
    if (device < 0 || device >= n) { dgi::set_error("device %d out of range (0..%d)", device, n - 1); return nullptr; }
    dg_ctx *c = new dg_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&c->prop, device) != hipSuccess) {
        dgi::set_error("cannot query device %d", device);
        delete c;
        return nullptr;
    }
    if (strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
        dgi::set_error("device %d is %s; this library is built for gfx950 (MI355X) only", device, c->prop.gcnArchName);
        delete c;
        return nullptr;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        dgi::set_error("hipStreamCreate failed");
        delete c;
        return nullptr;
    }
    c->own_stream = true;
    int rt = 0;
    if (hipRuntimeGetVersion(&rt) == hipSuccess && rt / 100000 != HIP_VERSION / 100000 && getenv("DG_DEBUG"))
        fprintf(stderr, "[dipgenie_hip] built against HIP %d.%d, running on HIP runtime %d.%d\n", HIP_VERSION / 10000000, HIP_VERSION / 100000 % 100, rt / 10000000, rt / 100000 % 100);
    return c;
}

extern "C" void dg_destroy(dg_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    dgi::dp_state_free(c->dp);
    dgi::sketch_state_free(c->sk);
    dgi::anchor_state_free(c->an);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int dg_set_stream(dg_ctx *c, void *s) {
    if (int rc = dgi::bind(c)) return rc;
    DG_HIP(hipStreamSynchronize(c->stream));
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)s;
    c->own_stream = false;
    return DG_OK;
}

extern "C" int dg_synchronize(dg_ctx *c) {
    if (int rc = dgi::bind(c)) return rc;
    DG_HIP(hipStreamSynchronize(c->stream));
    return DG_OK;
}

extern "C" int dg_device_info(dg_ctx *c, char *name, int cap, int *n_cu, int64_t *hbm) {
    if (int rc = dgi::bind(c)) return rc;
    if (name && cap > 0) { strncpy(name, c->prop.name, cap - 1); name[cap - 1] = 0; }
    if (n_cu) *n_cu = c->prop.multiProcessorCount;
    if (hbm) *hbm = (int64_t)c->prop.totalGlobalMem;
    return DG_OK;
}

extern "C" void dg_free(void *p) { free(p); }

// HIP_VERSION this library was compiled against vs the runtime it is bound to in this process (a Python host that imports torch
// first runs it on torch's bundled runtime): recorded by bench.py, checked by the GPU tests
extern "C" int dg_hip_versions(int *compiled, int *runtime) {
    if (compiled) *compiled = HIP_VERSION;
    int rt = 0;
    if (hipRuntimeGetVersion(&rt) != hipSuccess) { (void)hipGetLastError(); dgi::set_error("hipRuntimeGetVersion failed"); return DG_ERR_HIP; }
    if (runtime) *runtime = rt;
    return DG_OK;
}
