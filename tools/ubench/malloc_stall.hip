// hipMalloc / hipFree of large blocks: what a request costs by size (exact or rounded up to eighths of a power of two, as the
// pool of trc_kernels.hip rounds its small classes), fresh or right after a block of that size was freed, and whether touching the
// block matters.  Written to look into the 1.2 s calls of round 2 (gpurun_out/pair_on.log) that made requests above 128 MiB bypass
// the pool's rounding.  build: hipcc --offload-arch=gfx950 -O2 -o /tmp/malloc_stall tools/ubench/malloc_stall.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void touch(char *p, size_t n) {
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4096; i < n; i += (size_t)gridDim.x * blockDim.x * 4096) p[i] = 1;
}

static size_t round8(size_t b) { int k = 63 - __builtin_clzll((unsigned long long)b); size_t step = (size_t)1 << (k - 3); return (b + step - 1) & ~(step - 1); }

int main() {
    (void)hipFree(nullptr);
    const size_t sizes[] = {(size_t)900e6, (size_t)1.9e9, (size_t)3.7e9, (size_t)7.3e9, (size_t)14e9};
    printf("%14s %10s | %9s %9s %9s | %9s %9s\n", "bytes", "", "malloc", "touch", "free", "malloc#2", "free#2");
    for (int rounded = 0; rounded < 2; ++rounded)
        for (size_t s0 : sizes) {
            const size_t s = rounded ? round8(s0) : s0;
            void *p = nullptr;
            double t0 = now_ms();
            if (hipMalloc(&p, s) != hipSuccess) { printf("%14zu malloc failed\n", s); continue; }
            double t1 = now_ms();
            hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, 0, (char *)p, s);
            (void)hipDeviceSynchronize();
            double t2 = now_ms();
            (void)hipFree(p);
            double t3 = now_ms();
            (void)hipMalloc(&p, s);          // the same size again, right after the free
            double t4 = now_ms();
            (void)hipFree(p);
            double t5 = now_ms();
            printf("%14zu %10s | %9.2f %9.2f %9.2f | %9.2f %9.2f\n", s, rounded ? "rounded" : "as asked", t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4);
        }
    // many blocks alive (a workspace of the streaming engine: ~20 blocks of 0.3 .. 2 GB), then one more request of 2 GiB exactly
    std::vector<void *> held;
    for (int k = 0; k < 20; ++k) { void *q = nullptr; if (hipMalloc(&q, (size_t)(300e6 + 85e6 * k)) == hipSuccess) held.push_back(q); }
    for (int rep = 0; rep < 3; ++rep) {
        void *p = nullptr;
        double t0 = now_ms();
        (void)hipMalloc(&p, (size_t)2 << 30);
        double t1 = now_ms();
        (void)hipFree(p);
        double t2 = now_ms();
        printf("with %zu blocks (%.1f GB) alive: malloc of 2 GiB %.2f ms, free %.2f ms\n", held.size(), 0.3 * held.size() + 0.085 * 190, t1 - t0, t2 - t1);
    }
    for (void *q : held) (void)hipFree(q);
    return 0;
}
