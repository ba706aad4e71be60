// First layers of the learned proxies on the fp32 matrix cores with a LINEAR reduction index: few input channels, the
// whole (cin x k x k) x cout weight matrix and every channel of the halo tile staged ONCE per workgroup.
//
//   9x9,  3 -> 64   SRCNNRes once its 9+P broadcast planes are folded out (srcnn_res_arch.py:18, 41-46; SrcnnResFold)
//   9x9,  4 -> 64   SRCNNDemosaic on the space-to-depth mosaic (srcnn_demosaic_arch.py:14-16, 39-43)
//   3x3,  4 -> 64   Path14lBayer's first convolution on the space-to-depth mosaic (path_14l_bayer_arch.py:37-40, 70-75)
//   3x3,  3 -> 64   Path14lBgr's first convolution (path_14l_bgr_arch.py:40-43)
//
// The general kernel (risp_conv.hip) feeds v_mfma_f32_32x32x2_f32 with PAIRS OF CHANNELS of one filter tap (lanes 0-31
// channel ci, lanes 32-63 channel ci+1), loops over channel chunks with a barrier each, and reads a space-to-depth input
// element by element.  Here the reduction index is k = ci * K*K + ky * K + kx and the instruction's two k-slots are
// consecutive k: 3 channels of a 9x9 layer cost 122 instructions per accumulator tile instead of 162 (a quarter of the
// general kernel's multiply the zero channel).  Lanes 32-63 read the activation of slot k+1 at the address of slot k plus
// one of three constants (next column: +1; next filter row; next channel), so the B operand stays ONE ds_read_b32 per
// lane.  The weights arrive by LDS-DMA; the mosaic is read in 16-byte vectors and split into its four colour planes on
// the way into LDS.  No chunk loop, one barrier.  Output tile 16 x 32 pixels x 32 CB couts; epilogue bias / border-case
// table (RISP_EPI_CASEBIAS) / ReLU, 16-byte stores.  Exact fp32 arithmetic in another summation order than the general
// kernel (agreement ~1e-7 of the magnitude).
#include "risp_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TH = 16, TW = 32, RW = TH / 4;
constexpr int HPADC = 4;                       // halo columns staged left and right of the tile (16-byte aligned rows)
constexpr int IWP = TW + 2 * HPADC;             // staged row: column c <-> image x0 - HPADC + c

__device__ __forceinline__ int border_case(int v, int L, int P) {
    return v < P ? v : (v >= L - P ? 2 * P - (L - 1 - v) : P);
}

// UNSHUF: x is the (N, CIN/4, 2H, 2W) mosaic, plane 2i+j = x[2y+i][2x+j] (RISP_LOAD_UNSHUFFLE2 with CIN == 4)
template <int KS, int CIN, bool UNSHUF, int CB>
__global__ __launch_bounds__(256, 2) void conv_lin_kernel(const risp_conv_desc d_in, int ncb) {
    constexpr int PAD = KS / 2, IH = TH + KS - 1, TAPS = KS * KS, K = CIN * TAPS, K2 = (K + 1) / 2;
    constexpr int CP = 32 * CB;
    constexpr int XN = CIN * IH * IWP, WN = 2 * K2 * CP;
    static_assert(PAD <= HPADC && XN % 4 == 0 && WN % 4 == 0, "staging layout");
    constexpr int NWD = (WN / 4 + 63) / 64;            // DMA wave-instructions (weights), the last one may be partial
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sx = smem;                                  // [CIN][IH][IWP]
    float *sw = smem + XN;                             // [2 K2][CP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, n = blockIdx.z / ncb, cb = blockIdx.z - n * ncb;
    const risp_conv_desc d = risp_conv_group_view(d_in, n);
    const int wrow = wave * RW;
    const size_t hw = (size_t)d.H * d.W;

    // ---- stage the weight matrix of this cout block (LDS-DMA, shared by the 4 waves) and the halo tile
    const float *wsrc = d.wpack + (size_t)cb * WN;
    for (int i = wave; i < NWD; i += 4) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(i * 64 + lane < WN / 4);
        lds_dma16(wsrc + (size_t)i * 256 + lane * 4, sw + i * 256, m);
    }
    if constexpr (!UNSHUF) {
        constexpr int NXV = (XN / 4 + 255) / 256;
        const float *xn = d.x + (size_t)n * CIN * hw;
        float4 xr[NXV];
#pragma unroll
        for (int i = 0; i < NXV; ++i) {
            const int v = tid + 256 * i;
            const int cl = v / (IH * (IWP / 4)), rem = v - cl * (IH * (IWP / 4));
            const int iy = rem / (IWP / 4), q = rem - iy * (IWP / 4);
            const int gy = y0 + iy - PAD, gx = x0 - HPADC + 4 * q;
            const bool ok = v < XN / 4 && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;      // W % 4 == 0: all in or all out
            xr[i] = ok ? *reinterpret_cast<const float4 *>(xn + ((size_t)cl * d.H + gy) * d.W + gx) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NXV; ++i) {
            const int v = tid + 256 * i;
            if (v < XN / 4) reinterpret_cast<float4 *>(sx)[v] = xr[i];
        }
    } else {
        // a unit = 4 tile columns of the two planes (i, 0) and (i, 1) at tile row iy: 8 consecutive mosaic pixels of mosaic
        // row 2 gy + i, two 16-byte loads; even pixels go to plane 2i, odd ones to plane 2i + 1
        static_assert(CIN == 4, "space-to-depth staging is written for the 4-plane mosaic");
        constexpr int UNITS = 2 * IH * (IWP / 4), NU = (UNITS + 255) / 256;
        const float *xn = d.x + (size_t)n * 4 * hw;    // (2H) x (2W) mosaic of image n
        float4 a[NU], b[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = tid + 256 * i;
            const int pi = u / (IH * (IWP / 4)), rem = u - pi * (IH * (IWP / 4));
            const int iy = rem / (IWP / 4), q = rem - iy * (IWP / 4);
            const int gy = y0 + iy - PAD, gx = x0 - HPADC + 4 * q;
            const bool ok = u < UNITS && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
            const float *src = xn + (size_t)(2 * (ok ? gy : 0) + pi) * (2 * d.W) + 2 * (ok ? gx : 0);
            a[i] = ok ? *reinterpret_cast<const float4 *>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
            b[i] = ok ? *reinterpret_cast<const float4 *>(src + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = tid + 256 * i;
            if (u < UNITS) {
                const int pi = u / (IH * (IWP / 4)), rem = u - pi * (IH * (IWP / 4));     // rem = iy * (IWP/4) + q: the float4 slot in a plane
                reinterpret_cast<float4 *>(sx + (2 * pi) * IH * IWP)[rem] = make_float4(a[i].x, a[i].z, b[i].x, b[i].z);
                reinterpret_cast<float4 *>(sx + (2 * pi + 1) * IH * IWP)[rem] = make_float4(a[i].y, a[i].w, b[i].y, b[i].w);
            }
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

    // ---- K2 groups of RW x CB matrix instructions; the LDS operands of group j+1 are read before group j is issued
    const float *bx = sx + wrow * IWP + (HPADC - PAD) + l31;
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
    const int epi = d.epilogue, cbase = cb * CP;
    constexpr int NV = CP * 8 / 64;
    __syncthreads();                                   // every wave is done with the staged operands
    float *tile = smem + wave * (CP * 32);
    float bq[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int co = cbase + (lane >> 3) + 8 * i;
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
            const int co = cbase + (lane >> 3) + 8 * i;
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
            const int cl = (lane >> 3) + 8 * i, co = cbase + cl;
            const float4 v = *reinterpret_cast<const float4 *>(tile + cl * 32 + q4);
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

template <int KS, int CIN, bool UNSHUF, int CB>
int launch_lin(const risp_conv_desc &d, hipStream_t s) {
    constexpr int IH = TH + KS - 1, K2 = (CIN * KS * KS + 1) / 2;
    size_t lds = sizeof(float) * ((size_t)CIN * IH * IWP + (size_t)2 * K2 * 32 * CB);
    const size_t epi = sizeof(float) * 4 * 32 * CB * 32;          // the epilogue's four private transposition tiles
    if (lds < epi) lds = epi;
    const int ncb = (d.cout + 32 * CB - 1) / (32 * CB);
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_lin_kernel<KS, CIN, UNSHUF, CB>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        risp_set_error("risp_conv2d_k3: cannot raise the dynamic LDS limit to %zu bytes", lds);
        return 2;
    }
    dim3 grid((d.W + TW - 1) / TW, (d.H + TH - 1) / TH, d.N * ncb);
    hipLaunchKernelGGL((conv_lin_kernel<KS, CIN, UNSHUF, CB>), grid, dim3(256), lds, s, d, ncb);
    RISP_LAUNCH_CHECK("risp_conv2d_k3");
    return 0;
}

}  // namespace

extern "C" {

int risp_conv_k3_cout_block(int cin, int ksize) { return (cin * ksize * ksize > 256) ? 32 : 64; }

size_t risp_conv_k3_wpack_floats(int cin, int cout, int ksize) {
    const int cp = risp_conv_k3_cout_block(cin, ksize);
    return (size_t)2 * ((cin * ksize * ksize + 1) / 2) * cp * ((cout + cp - 1) / cp);
}

int risp_conv2d_k3(const risp_conv_desc *dp, void *stream) {
    RISP_CHECK_ARG(dp, "risp_conv2d_k3: null descriptor");
    const risp_conv_desc &d = *dp;
    RISP_CHECK_ARG(d.x && d.wpack && d.y, "risp_conv2d_k3: null tensor");
    RISP_CHECK_GROUP(d, "risp_conv2d_k3");
    const bool unshuf = d.load_mode == RISP_LOAD_UNSHUFFLE2;
    RISP_CHECK_ARG(d.N > 0 && d.H > 0 && d.W > 0 && d.W % 4 == 0 && d.cout > 0 && d.cout <= 64 && (size_t)d.N * 2 <= 65535 &&
                       (d.ksize == 9 || d.ksize == 3) && ((d.cin == 3 && !unshuf) || (d.cin == 4 && unshuf)) &&
                       (d.load_mode == RISP_LOAD_PLAIN || unshuf),
                   "risp_conv2d_k3: needs a 3x3 or 9x9 layer over 3 plain or 4 space-to-depth channels, cout <= 64, W %% 4 == 0 "
                   "(N=%d H=%d W=%d cin=%d cout=%d k=%d load=%d)", d.N, d.H, d.W, d.cin, d.cout, d.ksize, d.load_mode);
    RISP_CHECK_ARG(!(d.epilogue & ~(RISP_EPI_RELU | RISP_EPI_NOBIAS | RISP_EPI_CASEBIAS)), "risp_conv2d_k3: epilogue %d not supported",
                   d.epilogue);
    RISP_CHECK_ARG((d.epilogue & RISP_EPI_NOBIAS) || d.bias, "risp_conv2d_k3: bias missing");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_CASEBIAS) || (d.cvals && d.H >= d.ksize - 1 && d.W >= d.ksize - 1),
                   "risp_conv2d_k3: border-case bias needs its table in cvals and H, W >= k - 1");
    RISP_CHECK_ARG(((reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.y) | reinterpret_cast<uintptr_t>(d.wpack)) & 15) == 0,
                   "risp_conv2d_k3: tensors must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (d.ksize == 9 && !unshuf) return launch_lin<9, 3, false, 2>(d, s);
    if (d.ksize == 9) return launch_lin<9, 4, true, 1>(d, s);          // 324 x 64 weights do not fit twice per CU: cout blocks of 32
    if (!unshuf) return launch_lin<3, 3, false, 2>(d, s);
    return launch_lin<3, 4, true, 2>(d, s);
}

}  // extern "C"
