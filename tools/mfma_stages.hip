// Which ingredient of the Winograd convolution loop costs the matrix pipes their time?  (GPU box.)  The same
// v_mfma_f32_32x32x2_f32 stream as tools/mfma_peak.hip (6 accumulator tiles, LDS operands, 3 workgroups per CU), in
// "chunks" of 36 instructions per wave like conv_wino43_glds_kernel, with the real loop's other ingredients added one
// at a time:   B = one workgroup barrier per chunk      D = 6 LDS-DMA wave-instructions (16 B per lane) per wave and chunk
//              T = 14 vector instructions per 6 matrix instructions (the input transform)
//              E = a 64 KB store epilogue per workgroup every 16 chunks (one 64 -> 64 layer tile)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_stages tools/mfma_stages.hip && /tmp/mfma_stages
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lptr_t;

#ifndef DMA_MODE
#define DMA_MODE 0      // what a "piece" is made of (-DDMA_MODE=k): 0 the LDS-DMA of the kernels; 1 a 16-byte-per-lane load into registers
#endif                  // (no LDS write); 2 the scalar bookkeeping alone (M0 save / set / restore, no memory instruction); 3 the address arithmetic alone
__device__ __forceinline__ void lds_dma16(const float *src, float *dst_wave_base) {
    const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t *)dst_wave_base);
    unsigned keep_m0;
#if DMA_MODE == 0
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep_m0) : "v"(src), "s"(lds_dst) : "memory");
#elif DMA_MODE == 1
    // fixed high registers as the sink (declared clobbered, never read): a late-landing load cannot hit a live value
    asm volatile("global_load_dwordx4 v[160:163], %0, off" :: "v"(src) : "memory", "v160", "v161", "v162", "v163");
    (void)lds_dst; (void)keep_m0;
#elif DMA_MODE == 2
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\ts_mov_b32 m0, %0" : "=&s"(keep_m0) : "s"(lds_dst) : "memory");
    asm volatile("" :: "v"(src));
#else
    asm volatile("" :: "v"(src), "s"(lds_dst));
    (void)keep_m0;
#endif
}

// DMA_MODE 4: the same transfer addressed as a wave-uniform 64-bit base in scalar registers + a 32-bit per-lane offset (no
// 64-bit vector address arithmetic per piece)
__device__ __forceinline__ void lds_dma16_s(const float *uniform_base, unsigned lane_off_bytes, float *dst_wave_base) {
    const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t *)dst_wave_base);
    const unsigned long long ub = (unsigned long long)uniform_base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ub), hi = __builtin_amdgcn_readfirstlane((unsigned)(ub >> 32));
    const unsigned long long sb = ((unsigned long long)hi << 32) | lo;
    unsigned keep_m0;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep_m0) : "v"(lane_off_bytes), "s"(sb), "s"(lds_dst) : "memory");
}

template <bool BAR, bool DMA, int TRF, bool EPI, bool L2SRC = false, int NW = 4, int PIECES = 24, bool SPREAD = false>
__global__ __launch_bounds__(64 * NW, 12 / NW) void stages(const float *__restrict__ src, float *__restrict__ dst, float *out, int tiles,
                                                 float seed) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // 2 stages x 6144 floats (24 KB each)
    constexpr int STG = PIECES * 256;                 // floats per LDS stage
    for (int i = threadIdx.x; i < 2 * STG; i += 64 * NW) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        lds[i] = (float)(int)h * (1.0f / 2147483648.0f) * seed;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // shader clock during the run: cycles (s_memtime) per 100 MHz tick (s_memrealtime), stamped by one wave of every 97th block
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int tile = 0; tile < tiles; ++tile) {
        f32x16 acc[6];
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        const float *gsrc = src + ((size_t)(blockIdx.x * tiles + tile) % (L2SRC ? 4 : 4096)) * 16 * 6144;   // L2SRC: 1.5 MB, cache resident
        for (int ch = 0; ch < 16; ++ch) {
            const float *cur = lds + (ch & 1) * STG;
            if (BAR) __syncthreads();
            if (DMA && !SPREAD) {                        // the next chunk's tile + weight slab: 24 x 1 KB, 6 per wave
                float *nxt = lds + ((ch + 1) & 1) * STG;
#pragma unroll
                for (int j = 0; j < (PIECES + NW - 1) / NW; ++j) {
                    const int id = wave + NW * j;
#if DMA_MODE == 4
                    if (id < PIECES) lds_dma16_s(gsrc + (size_t)((ch + 1) & 15) * 6144 + (id % 24) * 256, lane * 16, nxt + id * 256);
#else
                    if (id < PIECES) lds_dma16(gsrc + (size_t)((ch + 1) & 15) * 6144 + (id % 24) * 256 + lane * 4, nxt + id * 256);
#endif
                }
            }
            if (TRF == 2) {
                typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int g = 0; g < 6; g += 2) {         // two groups per pass: the transform in packed fp32 (v_pk_fma_f32)
                    const float4 p0 = *reinterpret_cast<const float4 *>(&cur[(4 * lane + 256 * g) & 4092]);
                    const float4 p1 = *reinterpret_cast<const float4 *>(&cur[(4 * lane + 256 * g + 4) & 4092]);
                    const float4 r0 = *reinterpret_cast<const float4 *>(&cur[(4 * lane + 256 * g + 256) & 4092]);
                    const float4 r1 = *reinterpret_cast<const float4 *>(&cur[(4 * lane + 256 * g + 260) & 4092]);
                    const f2 d0 = {p0.x, r0.x}, d1 = {p0.y, r0.y}, d2 = {p0.z, r0.z}, d3 = {p0.w, r0.w}, d4 = {p1.x, r1.x}, d5 = {p1.y, r1.y};
                    const f2 c4 = {4.f, 4.f}, cm5 = {-5.f, -5.f}, cm4 = {-4.f, -4.f}, c2 = {2.f, 2.f}, cm2 = {-2.f, -2.f}, cm1 = {-1.f, -1.f};
                    f2 b[6];
                    b[0] = __builtin_elementwise_fma(c4, d0, __builtin_elementwise_fma(cm5, d2, d4));
                    const f2 s12 = d1 + d2, s34 = d3 + d4, m12 = d2 - d1, m34 = d4 - d3;
                    b[1] = __builtin_elementwise_fma(cm4, s12, s34);
                    b[2] = __builtin_elementwise_fma(c4, m12, -m34);
                    b[3] = __builtin_elementwise_fma(cm2, d1, __builtin_elementwise_fma(cm1, d2, __builtin_elementwise_fma(c2, d3, d4)));
                    b[4] = __builtin_elementwise_fma(c2, d1, __builtin_elementwise_fma(cm1, d2, __builtin_elementwise_fma(cm2, d3, d4)));
                    b[5] = __builtin_elementwise_fma(c4, d1, __builtin_elementwise_fma(cm5, d3, d5));
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int t = 0; t < 6; ++t) {
                            const float av = cur[(lane + 64 * t + 384 * (g + h)) & 4095];
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, h ? b[t].y : b[t].x, acc[t], 0, 0, 0);
                        }
                }
            } else
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                if (DMA && SPREAD) {                     // one piece per group, from inside the matrix-instruction stream
                    float *nxt = lds + ((ch + 1) & 1) * STG;
                    const int id = wave + NW * g;
                    if (id < PIECES) lds_dma16(gsrc + (size_t)((ch + 1) & 15) * 6144 + (id % 24) * 256 + lane * 4, nxt + id * 256);
                }
                const float4 q0 = *reinterpret_cast<const float4 *>(&cur[(4 * lane + 256 * g) & 4092]);
                const float4 q1 = *reinterpret_cast<const float4 *>(&cur[(4 * lane + 256 * g + 4) & 4092]);
                float b6[6] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y};
                if (TRF == 1) {                         // 14 fused multiply-adds on the six values (integer-coefficient transform)
                    const float d0 = b6[0], d1 = b6[1], d2 = b6[2], d3 = b6[3], d4 = b6[4], d5 = b6[5];
                    b6[0] = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4));
                    const float s12 = d1 + d2, s34 = d3 + d4, m12 = d2 - d1, m34 = d4 - d3;
                    b6[1] = __builtin_fmaf(-4.f, s12, s34);
                    b6[2] = __builtin_fmaf(4.f, m12 * 1.f, -m34) + d3 * 0.f;
                    b6[3] = __builtin_fmaf(-2.f, d1, __builtin_fmaf(-1.f, d2, __builtin_fmaf(2.f, d3, d4)));
                    b6[4] = __builtin_fmaf(2.f, d1, __builtin_fmaf(-1.f, d2, __builtin_fmaf(-2.f, d3, d4)));
                    b6[5] = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
                }
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    const float av = cur[(lane + 64 * t + 384 * g) & 4095];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b6[t], acc[t], 0, 0, 0);
                }
            }
            if (DMA) __builtin_amdgcn_s_waitcnt(0x0070);
        }
        if (EPI) {                                       // 4 rows x 128 pixels x 32 couts: 64 KB per workgroup, 16-byte stores
            float *o = dst + ((size_t)(blockIdx.x * tiles + tile) % 2048) * 65536 + wave * 4096;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 16; e += 4)
                    *reinterpret_cast<float4 *>(o + ((t * 4 + e / 4) * 64 + lane) * 4) =
                        make_float4(acc[t][e], acc[t][e + 1], acc[t][e + 2] + acc[4][e], acc[t][e + 3] + acc[5][e]);
        } else {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) s += acc[t][e];
            if (s == 12345.678f) out[0] = s;
        }
    }
    if (threadIdx.x == 0 && blockIdx.x % 97 == 0) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        out[1 + blockIdx.x / 97] = (float)((double)(c1 - c0) / (double)(r1 - r0) * 0.1);       // GHz
    }
}

template <bool BAR, bool DMA, int TRF, bool EPI, bool L2SRC = false, int NW = 4, int PIECES = 24, bool SPREAD = false>
void run(const char *what, const float *src, float *dst, float *out) {
    const int blocks = 256 * (12 / NW) * 2, tiles = 48;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipFuncSetAttribute(reinterpret_cast<const void *>(&stages<BAR, DMA, TRF, EPI, L2SRC, NW, PIECES, SPREAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PIECES * 1024);
        hipLaunchKernelGGL((stages<BAR, DMA, TRF, EPI, L2SRC, NW, PIECES, SPREAD>), dim3(blocks), dim3(64 * NW), 2 * PIECES * 1024, 0, src, dst, out, tiles, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double flops = (double)blocks * NW * tiles * 16 * 36 * (32.0 * 32 * 2 * 2);
    float ghz[64];
    hipMemcpy(ghz, out + 1, sizeof(float) * 16, hipMemcpyDeviceToHost);
    float med[16];
    int n = 0;
    for (int i = 0; i < 16 && i * 97 < blocks; ++i) med[n++] = ghz[i];
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) if (med[j] < med[i]) { float t = med[i]; med[i] = med[j]; med[j] = t; }
    const double clk = n ? med[n / 2] : 0.0, at_clk = 65536.0 * clk * 1e9 / 1e12;       // 1024 SIMDs x 64 FLOP per cycle
    printf("%-52s %8.3f ms  %7.1f TFLOP/s  (%.3f of 157.3)  clock %.2f GHz -> %.3f of the %.0f TFLOP/s at that clock\n", what, best,
           flops / best / 1e9, flops / best / 1e9 / 157.3, clk, flops / best / 1e9 / at_clk, at_clk);
}

int main() {
    float *src, *dst, *out;
    hipMalloc(&src, (size_t)4096 * 16 * 6144 * 4);      // 1.6 GB of tiles + slabs
    hipMalloc(&dst, (size_t)8192 * 16384 * 4);          // 512 MB of outputs
    hipMalloc(&out, 4 * 128);
    hipMemset(src, 0, (size_t)4096 * 16 * 6144 * 4);
    run<false, false, 0, false>("matrix stream + LDS operand reads", src, dst, out);
    run<true, false, 0, false>("+ barrier per chunk", src, dst, out);
    run<true, true, 0, false>("+ barrier + LDS-DMA of the next chunk (HBM source)", src, dst, out);
    run<true, true, 0, false, true>("+ barrier + LDS-DMA of the next chunk (L2 source)", src, dst, out);
    run<true, false, 1, false>("+ barrier + input transform (14 fma)", src, dst, out);
    run<true, false, 2, false>("+ barrier + input transform (7 pk_fma / group)", src, dst, out);
    run<true, true, 1, false>("+ barrier + LDS-DMA + input transform", src, dst, out);
    run<true, true, 2, false>("+ barrier + LDS-DMA + packed input transform", src, dst, out);
    run<true, true, 1, true>("+ barrier + LDS-DMA + transform + store epilogue", src, dst, out);
    run<true, true, 2, true>("+ barrier + LDS-DMA + packed transform + epilogue", src, dst, out);
    run<false, false, 0, true>("matrix stream + store epilogue", src, dst, out);
    run<true, true, 0, false, false, 4, 24, true>("+ barrier + LDS-DMA, one piece per group (spread)", src, dst, out);
    run<true, true, 1, false, false, 4, 24, true>("+ barrier + LDS-DMA spread + input transform", src, dst, out);
    run<true, true, 1, true, false, 4, 24, true>("+ barrier + LDS-DMA spread + transform + epilogue", src, dst, out);
    // one 12-wave workgroup per CU instead of three 4-wave ones: a 12-row tile and ONE weight slab = 40 pieces per chunk
    run<true, true, 1, true, false, 12, 40>("12-wave workgroup, 40 pieces / chunk, everything", src, dst, out);
    run<true, true, 1, false, false, 12, 40>("12-wave workgroup, 40 pieces / chunk, no epilogue", src, dst, out);
    run<true, true, 1, true, false, 8, 31>("8-wave workgroup (both cout blocks), 31 pieces, everything", src, dst, out);
    run<true, true, 1, true, false, 4, 12>("4-wave workgroup, 12 pieces / chunk (half the staging), everything", src, dst, out);
    return 0;
}
