// Throughput of the integer multiplies a Philox round can be built from, on one MI355X (gfx950).
// build: hipcc -O3 --offload-arch=gfx950 -o intmul intmul.hip ; run: ./intmul
// Each kernel runs ITER x 8 independent instructions of one kind per lane, 8 waves per SIMD on every CU; the result is
// printed as cycles per wave-instruction per SIMD at 2.4 GHz (4 = full rate for a wave64 on this SIMD arrangement when one
// wave issues alone; with several waves the float32 rate is 2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
#define DEF(NAME, ASM, CONSTR)                                                                      \
__global__ __launch_bounds__(512) void NAME(uint32_t *out, uint32_t seed) {                          \
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u + 1u, a2 = a0 ^ 0x55u, a3 = a0 + 7u,               \
             a4 = a0 * 5u, a5 = a0 + 11u, a6 = a0 ^ 0x77u, a7 = a0 + 13u;                             \
    uint64_t p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0, p6 = 0, p7 = 0;                          \
    const uint32_t M = 0xD2511F53u;                                                                   \
    for (int i = 0; i < ITER; ++i) {                                                                  \
        asm volatile(ASM : CONSTR(p0) : "v"(a0), "s"(M));                                            \
        asm volatile(ASM : CONSTR(p1) : "v"(a1), "s"(M));                                            \
        asm volatile(ASM : CONSTR(p2) : "v"(a2), "s"(M));                                            \
        asm volatile(ASM : CONSTR(p3) : "v"(a3), "s"(M));                                            \
        asm volatile(ASM : CONSTR(p4) : "v"(a4), "s"(M));                                            \
        asm volatile(ASM : CONSTR(p5) : "v"(a5), "s"(M));                                            \
        asm volatile(ASM : CONSTR(p6) : "v"(a6), "s"(M));                                            \
        asm volatile(ASM : CONSTR(p7) : "v"(a7), "s"(M));                                            \
        a0 ^= (uint32_t)p0; a1 ^= (uint32_t)p1; a2 ^= (uint32_t)p2; a3 ^= (uint32_t)p3;               \
        a4 ^= (uint32_t)p4; a5 ^= (uint32_t)p5; a6 ^= (uint32_t)p6; a7 ^= (uint32_t)p7;               \
    }                                                                                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;              \
}
#define C64(x) "=v"(x)
DEF(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, 0", C64)
#define DEF32(NAME, ASM)                                                                              \
__global__ __launch_bounds__(512) void NAME(uint32_t *out, uint32_t seed) {                          \
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u + 1u, a2 = a0 ^ 0x55u, a3 = a0 + 7u,               \
             a4 = a0 * 5u, a5 = a0 + 11u, a6 = a0 ^ 0x77u, a7 = a0 + 13u;                             \
    uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0, p6 = 0, p7 = 0;                          \
    const uint32_t M = 0xD2511F53u;                                                                   \
    for (int i = 0; i < ITER; ++i) {                                                                  \
        asm volatile(ASM : "=v"(p0) : "v"(a0), "s"(M));                                              \
        asm volatile(ASM : "=v"(p1) : "v"(a1), "s"(M));                                              \
        asm volatile(ASM : "=v"(p2) : "v"(a2), "s"(M));                                              \
        asm volatile(ASM : "=v"(p3) : "v"(a3), "s"(M));                                              \
        asm volatile(ASM : "=v"(p4) : "v"(a4), "s"(M));                                              \
        asm volatile(ASM : "=v"(p5) : "v"(a5), "s"(M));                                              \
        asm volatile(ASM : "=v"(p6) : "v"(a6), "s"(M));                                              \
        asm volatile(ASM : "=v"(p7) : "v"(a7), "s"(M));                                              \
        a0 ^= p0; a1 ^= p1; a2 ^= p2; a3 ^= p3; a4 ^= p4; a5 ^= p5; a6 ^= p6; a7 ^= p7;               \
    }                                                                                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;              \
}
DEF32(k_mul_hi_u32, "v_mul_hi_u32 %0, %1, %2")
DEF32(k_mul_lo_u32, "v_mul_lo_u32 %0, %1, %2")
DEF32(k_mul_u32_u24, "v_mul_u32_u24 %0, %1, %2")
DEF32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %1, %2")
DEF32(k_xor, "v_xor_b32 %0, %1, %2")
DEF32(k_mul_f32, "v_mul_f32 %0, %1, %2")

template <class K> static void run(const char *name, K kern, uint32_t *d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4;          // 4 x 512 threads per CU = 8 waves per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, d, (uint32_t)r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    // per SIMD: waves = blocks * 8 / (256 CUs * 4 SIMDs) = 8; instructions per wave = ITER * (8 measured + 8 xor)
    const double cyc = ms * 1e-3 * 2.4e9 / (8.0 * ITER * 8.0);
    printf("%-20s %.3f ms  -> %.1f cycles per wave-instruction per SIMD (incl. one v_xor each)\n", name, ms, cyc);
}
int main() {
    uint32_t *d; hipMalloc(&d, 256 * 4 * 512 * 4);
    run("v_xor_b32", k_xor, d);
    run("v_mul_f32", k_mul_f32, d);
    run("v_mul_u32_u24", k_mul_u32_u24, d);
    run("v_mul_hi_u32_u24", k_mul_hi_u32_u24, d);
    run("v_mul_lo_u32", k_mul_lo_u32, d);
    run("v_mul_hi_u32", k_mul_hi_u32, d);
    run("v_mad_u64_u32", k_mad_u64_u32, d);
    return 0;
}
