// Global operator new for the CLI: allocations of 4 MB and more are 2 MB-aligned and advised MADV_HUGEPAGE.
//
// Measured on the MI355X hosts (tools/exit_probe.hip, profiles/README.md): with 4 KB pages every GB of host memory a process
// touches costs ~0.17 s of page faults while it runs and ~0.1 s of teardown after _exit -- the MHC-24 run touches 2.5 GB
// (465 k minor faults, 1.5 s of system time over its threads) and its parent waited 0.2 s for the exit alone.  The hosts run
// transparent huge pages in "madvise" mode, so the big arrays (graph CSR, sequences, sketches) ask for them: 512 times
// fewer faults, and the teardown of a 2 MB page costs what a 4 KB page's does.  Linked into bin/DipGenie only; a program
// embedding the host pipeline keeps its own allocator.
#include <sys/mman.h>

#include <cstdlib>
#include <new>

namespace {
constexpr size_t HUGE = (size_t)2 << 20, BIG = (size_t)4 << 20;
inline void *dg_alloc(size_t n) {
    if (n >= BIG) {
        const size_t r = (n + HUGE - 1) & ~(HUGE - 1);
        void *p = aligned_alloc(HUGE, r);
        if (p) (void)madvise(p, r, MADV_HUGEPAGE);                  // a hint: failure leaves ordinary pages
        return p;
    }
    return malloc(n ? n : 1);
}
}  // namespace

void *operator new(size_t n) { if (void *p = dg_alloc(n)) return p; throw std::bad_alloc(); }
void *operator new[](size_t n) { if (void *p = dg_alloc(n)) return p; throw std::bad_alloc(); }
void *operator new(size_t n, const std::nothrow_t &) noexcept { return dg_alloc(n); }
void *operator new[](size_t n, const std::nothrow_t &) noexcept { return dg_alloc(n); }
void operator delete(void *p) noexcept { free(p); }
void operator delete[](void *p) noexcept { free(p); }
void operator delete(void *p, size_t) noexcept { free(p); }
void operator delete[](void *p, size_t) noexcept { free(p); }
void operator delete(void *p, const std::nothrow_t &) noexcept { free(p); }
void operator delete[](void *p, const std::nothrow_t &) noexcept { free(p); }
