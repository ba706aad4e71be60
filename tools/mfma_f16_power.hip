// What the f16 matrix pipe SUSTAINS (GPU box): the bare v_mfma_f32_32x32x16_f16 stream on every SIMD - 4 independent accumulator
// tiles per wave, operands in registers - with the in-kernel shader clock (s_memtime against the 100 MHz s_memrealtime), for
// operands of zeros, of small integers and of random halves, at duty cycles 1 (back to back), 3/4, 1/2 and 1/4 (idle gaps of
// s_sleep between bursts).  The nameplate 2516.6 TFLOP/s is 1024 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz; this prints the clock the
// chip actually holds under each load, i.e. the ceiling the split-precision convolutions are measured against.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f16_power tools/mfma_f16_power.hip && /tmp/mfma_f16_power
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// burst = 32 matrix instructions (4 tiles x 8); gap = s_sleep units (64 cycles each) after every burst
template <int GAP>
__global__ __launch_bounds__(256, 2) void stream(float *out, unsigned long long *clk, int iters, int pattern) {
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    h8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        unsigned h = (threadIdx.x * 8 + e) * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float r = (float)(int)h * (1.0f / 2147483648.0f);
        a[e] = pattern == 0 ? (_Float16)0.f : (pattern == 1 ? (_Float16)(float)((int)(h & 7) - 3) : (_Float16)(r * 20000.f));
        h *= 2654435761u;
        const float r2 = (float)(int)h * (1.0f / 2147483648.0f);
        b[e] = pattern == 0 ? (_Float16)0.f : (pattern == 1 ? (_Float16)(float)((int)(h >> 29) - 3) : (_Float16)(r2 * 20000.f));
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
        if (GAP > 0) __builtin_amdgcn_s_sleep(GAP);
    }
    __builtin_amdgcn_sched_barrier(0);
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[t][e];
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clk[0] = c1 - c0;
        clk[1] = r1 - r0;
    }
}

template <int GAP>
void run(const char *what, int pattern, float *out, unsigned long long *clk) {
    const int iters = 40000, blocks = 512;           // 2 workgroups of 4 waves per CU = 2 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((stream<GAP>), dim3(blocks), dim3(256), 0, 0, out, clk, 1000, pattern);
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream<GAP>), dim3(blocks), dim3(256), 0, 0, out, clk, iters, pattern);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 * iters * 32 * (32.0 * 32 * 16 * 2);
    const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
    const double busy = (double)iters * 32 * 32 * 2 / (double)h[0];       // 2 waves per SIMD x 32 cycles per instruction
    printf("%-14s gap %2d: %8.2f ms  %7.1f TFLOP/s (%.3f of 2516.6)  shader clock %.2f GHz  pipe busy %.2f  busy x clock %.2f GHz\n", what, GAP,
           ms, flops / ms / 1e9, flops / ms / 1e9 / 2516.6, ghz, busy, busy * ghz);
    fflush(stdout);
}

int main() {
    float *out;
    unsigned long long *clk;
    hipMalloc(&out, 4);
    hipMalloc(&clk, 16);
    const char *names[3] = {"zeros", "small ints", "random halves"};
    for (int p = 0; p < 3; ++p) {
        run<0>(names[p], p, out, clk);
        run<4>(names[p], p, out, clk);
        run<16>(names[p], p, out, clk);
        run<48>(names[p], p, out, clk);
    }
    return 0;
}
