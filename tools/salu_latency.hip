// Micro-benchmark (profiles/r03_salu_issue_rate.txt): what a lone wave pays per scalar op -- dependent chains vs k independent xorshift128 chains interleaved.
// hipcc --offload-arch=gfx950 -O3 -o salu_latency tools/salu_latency.hip && ./salu_latency
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ uint32_t step(uint32_t &x, uint32_t &y, uint32_t &z, uint32_t &w) {
    uint32_t t = x ^ (x << 11); x = y; y = z; z = w; w = w ^ (w >> 19) ^ (t ^ (t >> 8)); return w;
}
template <int CH> __global__ void k(unsigned long long *out, uint32_t seed, int iters) {
    uint32_t s[CH][4];
    for (int c = 0; c < CH; c++) { s[c][0] = __builtin_amdgcn_readfirstlane(seed + c); s[c][1] = 2; s[c][2] = 3; s[c][3] = 4 + c; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < CH; c++) acc ^= step(s[c][0], s[c][1], s[c][2], s[c][3]);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = acc; }
}
__global__ void kmul(unsigned long long *out, uint32_t seed, int iters) {
    uint32_t x = __builtin_amdgcn_readfirstlane(seed);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 32; u++) { x = __umulhi(x, 0x9e3779b9u) + 12345u; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = x; }
}
int main() {
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    const int it = 2000;
#define RUN(name, K, ops) do { for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(K, dim3(1), dim3(64), 0, 0, d, 7u, it); hipDeviceSynchronize(); } hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); \
    printf("%-34s %8.2f ticks per op (%llu ticks, %d ops)\n", name, (double)h[0] / ((double)it * (ops)), h[0], it * (ops)); } while (0)
    RUN("dependent mul_hi+add chain (2 ops)", kmul, 64);
    RUN("xorshift128 x1 (7 ops/output)", k<1>, 8 * 7);
    RUN("xorshift128 x2 interleaved", k<2>, 16 * 7);
    RUN("xorshift128 x3 interleaved", k<3>, 24 * 7);
    RUN("xorshift128 x4 interleaved", k<4>, 32 * 7);
    return 0;
}
