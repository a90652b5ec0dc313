// Measurement (not product): how long does the OS take to tear a process down after _exit, as a function of the HBM it
// had mapped / touched and of its host memory?  usage: exit_probe <chunks of 8 GB> <touch 0|1> <host GB>
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <ctime>
static double now() { struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
int main(int argc, char **argv) {
    const int chunks = argc > 1 ? atoi(argv[1]) : 0, touch = argc > 2 ? atoi(argv[2]) : 0;
    const double host_gb = argc > 3 ? atof(argv[3]) : 0;
    int n = 0;
    const double t0 = now();
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return 2;
    hipStream_t s; (void)hipStreamCreate(&s);
    std::vector<void *> p;
    const double t1 = now();
    for (int i = 0; i < chunks; ++i) { void *q = nullptr; if (hipMalloc(&q, (size_t)8 << 30) != hipSuccess) return 3; p.push_back(q); if (touch) (void)hipMemsetAsync(q, 1, (size_t)8 << 30, s); }
    (void)hipStreamSynchronize(s);
    const double t2 = now();
    if (host_gb > 0) { size_t nb = (size_t)(host_gb * (1 << 30)); char *h = (char *)malloc(nb); memset(h, 1, nb); if (h[nb / 2] == 7) puts("x"); }
    const double t3 = now();
    printf("init %.3f alloc+touch %.3f host %.3f exit_at %.6f\n", t1 - t0, t2 - t1, t3 - t2, t3);
    fflush(nullptr);
    _exit(0);
}
