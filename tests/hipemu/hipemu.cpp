// TEST INFRASTRUCTURE -- see hipemu.h.
#include "hipemu.h"
#include <ucontext.h>
#include <sys/mman.h>
#include <time.h>
#include <vector>

namespace hipemu {

ThreadCtx *cur = nullptr;

namespace {
const size_t kStack = 512 * 1024;
enum State { READY, AT_BARRIER, DONE };
struct Fiber {
    ucontext_t ctx;
    ThreadCtx tc;
    State st;
    char *stack;
};
std::vector<Fiber> fibers;
ucontext_t sched_ctx;
const std::function<void()> *body_fn = nullptr;
int cur_idx = 0;
unsigned char xchg[1024][16];
unsigned long long ballot_bits[16];
char *stack_pool = nullptr;
size_t stack_pool_n = 0;

void trampoline()
{
    (*body_fn)();
    fibers[cur_idx].st = DONE;
    swapcontext(&fibers[cur_idx].ctx, &sched_ctx);
}
} // namespace

void barrier()
{
    int me = cur_idx;
    fibers[me].st = AT_BARRIER;
    swapcontext(&fibers[me].ctx, &sched_ctx);
}

unsigned long long ballot(int pred)
{
    // two-phase: clear is done by the scheduler-independent protocol below
    int lane = (int) (cur->tid.x & 63), wave = (int) (cur->tid.x >> 6);
    if (lane == 0) ballot_bits[wave] = 0;
    barrier();
    if (pred) ballot_bits[wave] |= 1ull << lane;
    barrier();
    unsigned long long r = ballot_bits[wave];
    barrier();
    return r;
}

void exchange_put(const void *src, size_t n)
{
    if (n > 16) { fprintf(stderr, "hipemu: shuffle of %zu bytes unsupported\n", n); abort(); }
    memcpy(xchg[cur->tid.x], src, n);
}

void exchange_get(void *dst, size_t n, int src_lane)
{
    unsigned base = cur->tid.x & ~63u;
    memcpy(dst, xchg[base + (unsigned) src_lane], n);
}

void launch(dim3 grid, dim3 block, const std::function<void()> &body)
{
    size_t nthreads = (size_t) block.x * block.y * block.z;
    if (nthreads > 1024) { fprintf(stderr, "hipemu: block too large\n"); abort(); }
    if (stack_pool_n < nthreads) {
        if (stack_pool) munmap(stack_pool, stack_pool_n * kStack);
        stack_pool = (char *) mmap(nullptr, nthreads * kStack, PROT_READ | PROT_WRITE,
                                   MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (stack_pool == MAP_FAILED) { perror("mmap"); abort(); }
        stack_pool_n = nthreads;
    }
    body_fn = &body;
    fibers.resize(nthreads);
    ThreadCtx *saved = cur;
    for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++) {
        size_t t = 0;
        for (unsigned tz = 0; tz < block.z; tz++)
        for (unsigned ty = 0; ty < block.y; ty++)
        for (unsigned tx = 0; tx < block.x; tx++, t++) {
            Fiber &f = fibers[t];
            f.tc.tid = dim3(tx, ty, tz);
            f.tc.bid = dim3(bx, by, bz);
            f.tc.bdim = block;
            f.tc.gdim = grid;
            f.st = READY;
            f.stack = stack_pool + t * kStack;
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack;
            f.ctx.uc_stack.ss_size = kStack;
            f.ctx.uc_link = &sched_ctx;
            makecontext(&f.ctx, (void (*)()) trampoline, 0);
        }
        bool forward = true;
        for (;;) {
            size_t done = 0;
            for (size_t k = 0; k < nthreads; k++) {
                size_t i = forward ? k : nthreads - 1 - k;
                Fiber &f = fibers[i];
                if (f.st == DONE) { done++; continue; }
                f.st = READY;
                cur_idx = (int) i;
                cur = &f.tc;
                swapcontext(&sched_ctx, &f.ctx);
                if (f.st == DONE) done++;
            }
            if (done == nthreads) break;
            forward = !forward;
        }
    }
    cur = saved;
    body_fn = nullptr;
}

} // namespace hipemu

double hipemu_now()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}
