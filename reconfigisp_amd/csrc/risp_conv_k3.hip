// 9x9 convolution over THREE input channels on the fp32 matrix cores: the first layer of SRCNNRes once its 9+P
// broadcast planes are folded out (srcnn_res_arch.py:18, 41-46; convnets.py::SrcnnResFold) - 3 -> 64, K = 243.
//
// The general kernel (risp_conv.hip) feeds v_mfma_f32_32x32x2_f32 with PAIRS OF CHANNELS of one filter tap (lanes 0-31
// channel ci, lanes 32-63 channel ci+1), so 3 channels cost as much as 4: 2 chunks x 81 taps = 162 matrix instructions
// per accumulator tile, a quarter of them multiplying the zero channel.  Here the reduction index is LINEAR,
// k = ci * 81 + ky * 9 + kx, and the instruction's two k-slots are consecutive k: 122 instructions (244 slots, one
// padded).  Lanes 32-63 read the activation of slot k+1 at the address of slot k plus one of three constants (next
// column: +1; next filter row: + row stride - 8; next channel), so the B operand stays ONE ds_read_b32 per lane.
// All three channels and the whole 244 x 64 weight matrix are staged ONCE per workgroup (74 KB of LDS, two workgroups
// per CU; the weights arrive by LDS-DMA): no chunk loop, one barrier.  Output tile 16 x 32 pixels x 64 couts as in the
// general kernel; epilogue bias / border-case table (RISP_EPI_CASEBIAS) / ReLU, 16-byte stores.
// Same exact-fp32 arithmetic, another summation order than the general kernel (agreement ~1e-7 of the magnitude).
#include "risp_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TH = 16, TW = 32, RW = TH / 4;

__device__ __forceinline__ int border_case(int v, int L, int P) {
    return v < P ? v : (v >= L - P ? 2 * P - (L - 1 - v) : P);
}

template <int KS, int CIN>
__global__ __launch_bounds__(256, 2) void conv_k3_kernel(const risp_conv_desc d_in) {
    constexpr int PAD = KS / 2, IH = TH + KS - 1, IWP = TW + 2 * PAD, TAPS = KS * KS, K = CIN * TAPS, K2 = (K + 1) / 2;
    constexpr int CB = 2, CP = 32 * CB;
    constexpr int XN = CIN * IH * IWP, WN = 2 * K2 * CP;
    static_assert(IWP % 4 == 0 && PAD % 4 == 0 && XN % 4 == 0 && (WN / 4) % 64 == 0, "staging layout");
    constexpr int NXV = (XN / 4 + 255) / 256, NWD = WN / 4 / 64;       // float4 per thread (tile); DMA wave-instructions (weights)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sx = smem;                                  // [CIN][IH][IWP], column c <-> image x0 - PAD + c
    float *sw = smem + XN;                             // [2 K2][CP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, n = blockIdx.z;
    const risp_conv_desc d = risp_conv_group_view(d_in, n);
    const int wrow = wave * RW;
    const size_t hw = (size_t)d.H * d.W;

    // ---- stage the weight matrix (LDS-DMA, 61 wave-instructions shared by the 4 waves) and the halo tile
    for (int i = wave; i < NWD; i += 4) lds_dma16(d.wpack + (size_t)i * 256 + lane * 4, sw + i * 256, ~0ull);
    {
        const float *xn = d.x + (size_t)n * CIN * hw;
        float4 xr[NXV];
#pragma unroll
        for (int i = 0; i < NXV; ++i) {
            const int v = tid + 256 * i;
            const int cl = v / (IH * (IWP / 4)), rem = v - cl * (IH * (IWP / 4));
            const int iy = rem / (IWP / 4), q = rem - iy * (IWP / 4);
            const int gy = y0 + iy - PAD, gx = x0 - PAD + 4 * q;
            const bool ok = v < XN / 4 && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;      // W % 4 == 0: all in or all out
            xr[i] = ok ? *reinterpret_cast<const float4 *>(xn + ((size_t)cl * d.H + gy) * d.W + gx) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NXV; ++i) {
            const int v = tid + 256 * i;
            if (v < XN / 4) reinterpret_cast<float4 *>(sx)[v] = xr[i];
        }
    }
    f32x16 acc[RW][CB];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;
    __builtin_amdgcn_s_waitcnt(0x0070);                // vmcnt(0): this wave's DMA transfers have landed
    __syncthreads();

    // ---- 122 groups of RW x CB matrix instructions; the LDS operands of group j+1 are read before group j is issued
    const float *bx = sx + wrow * IWP + l31;
    const float *aw = sw + half * CP + l31;
    auto koff = [](int k) constexpr {                  // LDS offset of reduction slot k inside the tile (row r = 0)
        const int ci = k / TAPS, t = k - ci * TAPS, ky = t / KS, kx = t - ky * KS;
        return (ci * IH + ky) * IWP + kx;
    };
    float opa[2][CB], opb[2][RW];
    auto load_group = [&](int j, int slot) {
        const int o0 = koff(2 * j), delta = (2 * j + 1 < K) ? koff(2 * j + 1) - o0 : 0;     // padded slot: weight 0, any valid address
#pragma unroll
        for (int c = 0; c < CB; ++c) opa[slot][c] = aw[2 * j * CP + c * 32];
#pragma unroll
        for (int r = 0; r < RW; ++r) opb[slot][r] = bx[o0 + half * delta + r * IWP];
    };
    load_group(0, 0);
#pragma unroll
    for (int j = 0; j < K2; ++j) {
        const int slot = j & 1;
        if (j + 1 < K2) load_group(j + 1, slot ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int c = 0; c < CB; ++c)
                acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[slot][c], opb[slot][r], acc[r][c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue (as risp_conv.hip's vector path, without residual / mask): each output row of the wave is transposed
    // through a private LDS tile so that a lane owns 4 consecutive pixels of one cout plane
    const int epi = d.epilogue;
    constexpr int NV = CP * 8 / 64;
    __syncthreads();                                   // every wave is done with the staged operands
    float *tile = smem + wave * (CP * 32);
    float bq[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int co = (lane >> 3) + 8 * i;
        bq[i] = 0.f;
        if (!(epi & RISP_EPI_NOBIAS)) bq[i] = d.bias[co < d.cout ? co : d.cout - 1];
    }
    const int q4 = 4 * (lane & 7);
    const bool caseb = (epi & RISP_EPI_CASEBIAS) != 0;
    const float *__restrict__ ctab = d.cvals + (size_t)n * d.cout * TAPS;
    const bool case_edge = caseb && (x0 < PAD || y0 < PAD || x0 + TW > d.W - PAD || y0 + TH > d.H - PAD);
    if (caseb && !case_edge) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int co = (lane >> 3) + 8 * i;
            if (co < d.cout) bq[i] += ctab[co * TAPS + PAD * KS + PAD];
        }
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int oy = y0 + wrow + r;
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) tile[(c * 32 + (e & 3) + 8 * (e >> 2) + 4 * half) * 32 + l31] = acc[r][c][e];
        __builtin_amdgcn_wave_barrier();
        const bool row_ok = oy < d.H && x0 + q4 < d.W;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int co = (lane >> 3) + 8 * i;
            const float4 v = *reinterpret_cast<const float4 *>(tile + co * 32 + q4);
            float4 o;
            o.x = v.x + bq[i]; o.y = v.y + bq[i]; o.z = v.z + bq[i]; o.w = v.w + bq[i];
            if (case_edge && row_ok && co < d.cout) {
                const float *ct = ctab + co * TAPS + border_case(oy, d.H, PAD) * KS;
                const int ox4 = x0 + q4;
                o.x += ct[border_case(ox4, d.W, PAD)];
                o.y += ct[border_case(ox4 + 1, d.W, PAD)];
                o.z += ct[border_case(ox4 + 2, d.W, PAD)];
                o.w += ct[border_case(ox4 + 3, d.W, PAD)];
            }
            if (epi & RISP_EPI_RELU) {
                o.x = o.x > 0.f ? o.x : 0.f;
                o.y = o.y > 0.f ? o.y : 0.f;
                o.z = o.z > 0.f ? o.z : 0.f;
                o.w = o.w > 0.f ? o.w : 0.f;
            }
            if (row_ok && co < d.cout)
                *reinterpret_cast<float4 *>(d.y + ((size_t)n * d.cout + co) * hw + (size_t)oy * d.W + x0 + q4) = o;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

extern "C" {

size_t risp_conv_k3_wpack_floats(int cin, int cout, int ksize) {
    return (size_t)2 * ((cin * ksize * ksize + 1) / 2) * 32 * ((cout + 31) / 32);
}

int risp_conv2d_k3(const risp_conv_desc *dp, void *stream) {
    RISP_CHECK_ARG(dp, "risp_conv2d_k3: null descriptor");
    const risp_conv_desc &d = *dp;
    RISP_CHECK_ARG(d.x && d.wpack && d.y, "risp_conv2d_k3: null tensor");
    RISP_CHECK_GROUP(d, "risp_conv2d_k3");
    RISP_CHECK_ARG(d.N > 0 && d.N <= 65535 && d.H > 0 && d.W > 0 && d.W % 4 == 0 && d.cin == 3 && d.cout > 32 && d.cout <= 64 &&
                       d.ksize == 9,
                   "risp_conv2d_k3: needs a 9x9 layer with 3 input and 33..64 output channels, W %% 4 == 0 (N=%d H=%d W=%d cin=%d "
                   "cout=%d k=%d)", d.N, d.H, d.W, d.cin, d.cout, d.ksize);
    RISP_CHECK_ARG(d.load_mode == RISP_LOAD_PLAIN, "risp_conv2d_k3: only plain loads");
    RISP_CHECK_ARG(!(d.epilogue & ~(RISP_EPI_RELU | RISP_EPI_NOBIAS | RISP_EPI_CASEBIAS)), "risp_conv2d_k3: epilogue %d not supported",
                   d.epilogue);
    RISP_CHECK_ARG((d.epilogue & RISP_EPI_NOBIAS) || d.bias, "risp_conv2d_k3: bias missing");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_CASEBIAS) || (d.cvals && d.H >= 8 && d.W >= 8),
                   "risp_conv2d_k3: border-case bias needs its table in cvals and H, W >= 8");
    RISP_CHECK_ARG(((reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.y) | reinterpret_cast<uintptr_t>(d.wpack)) & 15) == 0,
                   "risp_conv2d_k3: tensors must be 16-byte aligned");
    constexpr int IH = TH + 8, IWP = TW + 8, K2 = (3 * 81 + 1) / 2;
    const size_t lds = sizeof(float) * ((size_t)3 * IH * IWP + (size_t)2 * K2 * 64);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_k3_kernel<9, 3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
        risp_set_error("risp_conv2d_k3: cannot raise the dynamic LDS limit to %zu bytes", lds);
        return 2;
    }
    dim3 grid((d.W + TW - 1) / TW, (d.H + TH - 1) / TH, d.N);
    hipLaunchKernelGGL((conv_k3_kernel<9, 3>), grid, dim3(256), lds, (hipStream_t)stream, d);
    RISP_LAUNCH_CHECK("risp_conv2d_k3");
    return 0;
}

}  // extern "C"
