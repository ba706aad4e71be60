// The bare v_mfma_f32_16x16x4_f32 stream (GPU box): NT independent 4-register accumulator tiles per wave, operands in
// registers (or one ds_read_b32 per instruction), 1 / 2 / 4 waves per SIMD - the ceiling a 16x16 tiling of the convolution
// kernels would be measured against (tools/mfma_peak.hip is the same for the 32x32x2 form: 0.95-0.99 of 157.3).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak16 tools/mfma_peak16.hip && /tmp/mfma_peak16
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WAVES_PER_SIMD, int NT, bool LDS>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void mfma_stream(float *out, int iters, float seed) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        lds[i] = (float)(int)h * (1.0f / 2147483648.0f) * seed;
    }
    __syncthreads();
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    float a = lds[lane], b = lds[lane + 64];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float av = LDS ? lds[(lane + 64 * t + 192 * (it & 7)) & 4095] : a;
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[t], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    if (s == 12345.678f) out[0] = s;
}

template <int W, int NT, bool LDS>
void run(const char *what, float *out) {
    const int iters = 8000 * 24 / NT, blocks = 256 * W * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((mfma_stream<W, NT, LDS>), dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_stream<W, NT, LDS>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 4 * iters * NT * (16.0 * 16 * 4 * 2);
        printf("%-28s %2d tiles, %d waves/SIMD: %8.3f ms  %7.1f TFLOP/s  (%.3f of 157.3)\n", what, NT, W, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
        fflush(stdout);
    }
}

int main() {
    float *out;
    hipMalloc(&out, 4);
    run<1, 24, false>("register operands", out);
    run<2, 24, false>("register operands", out);
    run<4, 24, false>("register operands", out);
    run<1, 48, false>("register operands", out);
    run<2, 48, false>("register operands", out);
    run<2, 8, false>("register operands", out);
    run<4, 8, false>("register operands", out);
    run<2, 48, true>("LDS A operand (b32)", out);
    run<4, 24, true>("LDS A operand (b32)", out);
    return 0;
}
