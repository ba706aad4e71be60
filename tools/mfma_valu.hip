// Do vector instructions beside the fp32 matrix stream cost matrix-pipe time, and does the answer depend on the tile form?
// (GPU box.)  Per 384 cycles of matrix work - six v_mfma_f32_32x32x2_f32 or twelve v_mfma_f32_16x16x4_f32, register
// operands - NV dependent-free fused multiply-adds (the F(4,3) input transform is 14 per six 32x32x2 instructions).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu tools/mfma_valu.hip && /tmp/mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int W, bool SMALL, int NV>
__global__ __launch_bounds__(256, W) void stream(float *out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    float a = seed + lane, b = seed * 0.5f + lane;
    float v[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) v[i] = seed * (i + 1) + lane;
    f32x16 big[6];
    f32x4 small[24];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) big[t][e] = 0.f;
#pragma unroll
    for (int t = 0; t < 24; ++t) small[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i % 14] = __builtin_fmaf(v[i % 14], 1.0001f, v[(i + 5) % 14]);
            // the matrix instructions take their B operand from the vector results, as the transform's consumers do
            if (SMALL) {
#pragma unroll
                for (int t = 0; t < 12; ++t)
                    small[(g & 1) * 12 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v[t % 14], small[(g & 1) * 12 + t], 0, 0, 0);
            } else {
#pragma unroll
                for (int t = 0; t < 6; ++t) big[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v[t], big[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = b;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += big[t][e];
#pragma unroll
    for (int t = 0; t < 24; ++t) s += small[t][0] + small[t][1] + small[t][2] + small[t][3];
#pragma unroll
    for (int i = 0; i < 14; ++i) s += v[i];
    if (s == 12345.678f) out[0] = s;
}

template <int W, bool SMALL, int NV>
void run(float *out) {
    const int iters = 6000, blocks = 256 * W * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((stream<W, SMALL, NV>), dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream<W, SMALL, NV>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double flops = (double)blocks * 4 * iters * 4 * 6 * (32.0 * 32 * 2 * 2);
    printf("%-10s %2d vector instr / 384 matrix cycles, %d waves/SIMD: %8.3f ms  %7.1f TFLOP/s  (%.3f of 157.3)\n", SMALL ? "16x16x4" : "32x32x2", NV, W,
           ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
    fflush(stdout);
}

int main() {
    float *out;
    hipMalloc(&out, 4);
    run<3, false, 0>(out);
    run<3, false, 14>(out);
    run<3, false, 28>(out);
    run<3, true, 0>(out);
    run<3, true, 14>(out);
    run<3, true, 28>(out);
    run<3, true, 42>(out);
    run<2, false, 14>(out);
    run<2, true, 14>(out);
    run<2, true, 28>(out);
    run<1, false, 14>(out);
    run<1, true, 14>(out);
    return 0;
}
