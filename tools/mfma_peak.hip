// What the fp32 matrix pipes of this chip deliver when NOTHING else is in the way (GPU box): v_mfma_f32_32x32x2_f32 streams
// with 6 independent accumulator tiles per wave (the shape of the Winograd kernels' inner loop), operands in registers,
// no memory traffic, 1 / 2 / 3 waves per SIMD; then the same stream with the LDS operand reads of the real loop
// (one ds_read_b32 + half a ds_read_b128 per instruction).  Prints TFLOP/s against the 157.3 TFLOP/s nameplate - the
// gap is the clock the chip holds under a pure matrix load, i.e. the ceiling every convolution kernel is measured against.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WAVES_PER_SIMD, bool LDS, bool RANDOM>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void mfma_stream(float *out, int iters, float seed) {
    __shared__ float lds[4096];
    // RANDOM operand bits (an integer hash mapped to [-1, 1)): a matrix pipe multiplying constants draws less power than
    // one multiplying data, and the clock the chip holds depends on the power it draws
    for (int i = threadIdx.x; i < 4096; i += 256) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        lds[i] = RANDOM ? (float)(int)h * (1.0f / 2147483648.0f) * seed : seed * (float)(i & 15);
    }
    __syncthreads();
    f32x16 acc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float a = seed + threadIdx.x, b = seed * 0.5f + threadIdx.x;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            float b6[6];
            if (LDS) {
                const float4 q0 = *reinterpret_cast<const float4 *>(&lds[(4 * lane + 256 * g) & 4092]);
                const float4 q1 = *reinterpret_cast<const float4 *>(&lds[(4 * lane + 256 * g + 4) & 4092]);
                b6[0] = q0.x; b6[1] = q0.y; b6[2] = q0.z; b6[3] = q0.w; b6[4] = q1.x; b6[5] = q1.y;
            } else {
#pragma unroll
                for (int t = 0; t < 6; ++t) b6[t] = b + t;
            }
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const float av = LDS ? lds[(lane + 64 * t + 384 * g) & 4095] : a;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b6[t], acc[t], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[t][e];
    if (s == 12345.678f) out[0] = s;
}

template <int W, bool LDS, bool RANDOM = false>
void run(const char *what, float *out) {
    const int iters = 4000, blocks = 256 * W * 4;          // W workgroups per CU, several rounds
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((mfma_stream<W, LDS, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);      // warm the clocks
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_stream<W, LDS, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 4 * iters * 48 * (32.0 * 32 * 2 * 2);
        printf("%-44s %d waves/SIMD: %8.3f ms  %7.1f TFLOP/s  (%.3f of 157.3)\n", what, W, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
    }
}

int main() {
    float *out;
    hipMalloc(&out, 4);
    run<1, false>("register operands", out);
    run<2, false>("register operands", out);
    run<3, false>("register operands", out);
    run<1, true>("LDS operands (b32 + b128/2 per instruction)", out);
    run<2, true>("LDS operands (b32 + b128/2 per instruction)", out);
    run<3, true>("LDS operands (b32 + b128/2 per instruction)", out);
    run<1, true, true>("LDS operands, random data", out);
    run<2, true, true>("LDS operands, random data", out);
    run<3, true, true>("LDS operands, random data", out);
    return 0;
}
