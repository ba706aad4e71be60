// Convolution layers of the learned proxies on the gfx950 fp32 matrix cores.
//
// Reference layers: SRCNNRes (srcnn_res_arch.py:15-24), SRCNNDemosaic
// (srcnn_demosaic_arch.py:14-25), Path14lBayer / Path14lBgr + ResidualBlock
// (path_14l_bayer_arch.py:6-57, path_14l_bgr_arch.py:6-56).  All are stride-1,
// zero-'same'-padded, odd square kernels in fp32; parity bar 1e-4 relative, so the
// exact-f32 MFMA (v_mfma_f32_32x32x2_f32, a k-ordered fmaf chain) is the only matrix
// instruction that qualifies (no xf32 on gfx950; bf16/fp8 would break parity).
//
// Implicit GEMM, per workgroup (256 threads = 4 waves, one per SIMD):
//     D[cout][pixel] += W[cout][k] * X[k][pixel],   k = (chunk, ky, kx, ci)
//   * output tile 16 rows x 32 pixels x all couts (<= 64); a wave owns 4 rows and keeps
//     4 x CB accumulator tiles of 32(cout) x 32(pixel) in VGPRs (D column = lane&31 = x,
//     so epilogue stores are 128-byte row segments of the NCHW planes);
//   * per chunk of CK input channels the zero-padded halo tile CK x (16+k-1) x (32+k-1)
//     and the matching weight slab [tap][ci][cout] are staged in LDS; the MFMA A operand
//     (weights) and B operand (pixels) are single conflict-free ds_read_b32 per lane
//     (lanes 0-31 -> channel ci, lanes 32-63 -> channel ci+1 of the same tap);
//   * space-to-depth (Bayer -> RGGB planes), the SRCNNRes broadcast planes, bias, residual
//     add, ReLU, ReLU-mask (backward) and PixelShuffle(2) are folded into the load / store
//     index maps, so no layer materialises a temporary.
// LDS per workgroup is 38-50 KB and VGPRs <= 256, so 2-3 workgroups share a CU and one
// workgroup's staging overlaps another's MFMA stream.
#include "risp_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TH = 16, TW = 32, RW = TH / 4;

struct ConvCfg {
    int ck, cb;
};

// chunk depth / cout blocks per (ksize, cin, cout) - shared by the packer and the launcher
ConvCfg conv_cfg(int cin, int cout, int ks) {
    ConvCfg c;
    c.cb = cout > 32 ? 2 : 1;
    if (ks == 9) c.ck = 2;
    else if (ks == 1) c.ck = 16;
    else if (ks == 5 && c.cb == 2) c.ck = 4;   // keeps the weight slab under 64 KB of LDS
    else c.ck = cin >= 8 ? 8 : 4;
    return c;
}

__device__ __forceinline__ float load_px(const risp_conv_desc &d, int n, int ci, int gy, int gx) {
    if (ci >= d.cin || gy < 0 || gy >= d.H || gx < 0 || gx >= d.W) return 0.f;
    if (d.load_mode == RISP_LOAD_PLAIN) return d.x[(((size_t)n * d.cin + ci) * d.H + gy) * d.W + gx];
    if (d.load_mode == RISP_LOAD_UNSHUFFLE2) {
        const int c = ci >> 2, i = (ci >> 1) & 1, j = ci & 1;
        return d.x[(((size_t)n * (d.cin >> 2) + c) * (2 * d.H) + 2 * gy + i) * (2 * (size_t)d.W) + 2 * gx + j];
    }
    // RISP_LOAD_CONSTCH
    if (ci < d.cin_img) return d.x[(((size_t)n * d.cin_img + ci) * d.H + gy) * d.W + gx];
    return d.cvals[n * (d.cin - d.cin_img) + (ci - d.cin_img)];
}

template <int KS, int CK, int CB>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const risp_conv_desc d) {
    constexpr int PAD = KS / 2, IH = TH + KS - 1, IW = TW + KS - 1, CP = 32 * CB, TAPS = KS * KS;
    constexpr int XN = CK * IH * IW, WN = TAPS * CK * CP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sx = smem;                          // [CK][IH][IW]
    float *sw = smem + ((XN + 3) & ~3);        // [TAPS][CK][CP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, n = blockIdx.z;
    const int wrow = wave * RW;

    f32x16 acc[RW][CB];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;

    const int nchunks = (d.cin + CK - 1) / CK;
    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();
        for (int idx = tid; idx < XN; idx += 256) {
            const int cl = idx / (IH * IW), rem = idx - cl * (IH * IW);
            const int iy = rem / IW, ix = rem - iy * IW;
            sx[idx] = load_px(d, n, ch * CK + cl, y0 + iy - PAD, x0 + ix - PAD);
        }
        {
            const float4 *src = reinterpret_cast<const float4 *>(d.wpack + (size_t)ch * WN);
            float4 *dst = reinterpret_cast<float4 *>(sw);
            for (int idx = tid; idx < WN / 4; idx += 256) dst[idx] = src[idx];
        }
        __syncthreads();

        const float *bx = sx + (half * IH + wrow) * IW + l31;   // channel `half` of pair 0, this wave's rows
        const float *aw = sw + half * CP + l31;
        for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
#pragma unroll
                for (int cp = 0; cp < CK / 2; ++cp) {
                    float av[CB], bv[RW];
#pragma unroll
                    for (int c = 0; c < CB; ++c) av[c] = aw[((ky * KS + kx) * CK + 2 * cp) * CP + c * 32];
#pragma unroll
                    for (int r = 0; r < RW; ++r) bv[r] = bx[(2 * cp * IH + r + ky) * IW + kx];
#pragma unroll
                    for (int r = 0; r < RW; ++r)
#pragma unroll
                        for (int c = 0; c < CB; ++c)
                            acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], bv[r], acc[r][c], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: lane holds D[cout = cb*32 + (e&3) + 8*(e>>2) + 4*half][x = x0 + l31]
    const int ox = x0 + l31;
    if (ox >= d.W) return;
    const int epi = d.epilogue;
    const size_t plane = (size_t)d.H * d.W;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int oy = y0 + wrow + r;
        if (oy >= d.H) continue;
#pragma unroll
        for (int c = 0; c < CB; ++c) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = c * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (co >= d.cout) continue;
                float v = acc[r][c][e];
                if (!(epi & RISP_EPI_NOBIAS)) v += d.bias[co];
                const size_t o = ((size_t)n * d.cout + co) * plane + (size_t)oy * d.W + ox;
                if ((epi & RISP_EPI_ADD) && co < d.add_c)
                    v += d.add[((size_t)n * d.add_c + co) * plane + (size_t)oy * d.W + ox];
                if (epi & RISP_EPI_RELU) v = v > 0.f ? v : 0.f;
                if (epi & RISP_EPI_MASK) v = d.mask[o] > 0.f ? v : 0.f;
                if (epi & RISP_EPI_SHUFFLE2) {
                    const int cc = co >> 2, i = (co >> 1) & 1, j = co & 1;
                    d.y[(((size_t)n * (d.cout >> 2) + cc) * (2 * d.H) + 2 * oy + i) * (2 * (size_t)d.W) + 2 * ox + j] = v;
                } else {
                    d.y[o] = v;
                }
            }
        }
    }
}

// wpack[chunk][tap][ci_l][co_pad]; zero outside (cin, cout)
__global__ void pack_kernel(const float *__restrict__ w, float *__restrict__ wp, int cin, int cout, int ks, int ck,
                            int cp, int transpose, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int taps = ks * ks;
    const int co = (int)(i % cp);
    size_t t = i / cp;
    const int cl = (int)(t % ck);
    t /= ck;
    const int tap = (int)(t % taps);
    const int chunk = (int)(t / taps);
    const int ci = chunk * ck + cl;
    float v = 0.f;
    if (ci < cin && co < cout) {
        if (!transpose)
            v = w[((size_t)co * cin + ci) * taps + tap];
        else  // w is (cin, cout, k, k) of the forward layer: swap roles, rotate taps by 180 degrees
            v = w[((size_t)ci * cout + co) * taps + (taps - 1 - tap)];
    }
    wp[i] = v;
}

template <int KS, int CK, int CB>
int launch_conv(const risp_conv_desc &d, hipStream_t s) {
    constexpr int IH = TH + KS - 1, IW = TW + KS - 1;
    constexpr size_t lds = ((((size_t)CK * IH * IW + 3) & ~(size_t)3) + (size_t)KS * KS * CK * 32 * CB) * sizeof(float);
    static_assert(lds <= 64 * 1024, "LDS tile too large");
    dim3 grid((d.W + TW - 1) / TW, (d.H + TH - 1) / TH, d.N);
    hipLaunchKernelGGL((conv_mfma_kernel<KS, CK, CB>), grid, dim3(256), lds, s, d);
    RISP_LAUNCH_CHECK("risp_conv2d");
    return 0;
}

}  // namespace

extern "C" {

size_t risp_conv_wpack_floats(int cin, int cout, int ksize) {
    const ConvCfg c = conv_cfg(cin, cout, ksize);
    const size_t nchunks = (cin + c.ck - 1) / c.ck;
    return nchunks * ksize * ksize * c.ck * 32 * c.cb;
}

int risp_conv_pack_weights(const float *w, int cin, int cout, int ksize, int transpose, float *wpack, void *stream) {
    RISP_CHECK_ARG(w && wpack && cin > 0 && cout > 0 && cout <= 64 && (ksize == 1 || ksize == 3 || ksize == 5 || ksize == 9),
                   "risp_conv_pack_weights: unsupported layer cin=%d cout=%d k=%d", cin, cout, ksize);
    const ConvCfg c = conv_cfg(cin, cout, ksize);
    const size_t total = risp_conv_wpack_floats(cin, cout, ksize);
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, wpack,
                       cin, cout, ksize, c.ck, 32 * c.cb, transpose, total);
    RISP_LAUNCH_CHECK("risp_conv_pack_weights");
    return 0;
}

int risp_conv2d(const risp_conv_desc *dp, void *stream) {
    RISP_CHECK_ARG(dp, "risp_conv2d: null descriptor");
    const risp_conv_desc &d = *dp;
    RISP_CHECK_ARG(d.x && d.wpack && d.y, "risp_conv2d: null tensor");
    RISP_CHECK_ARG(d.N > 0 && d.N <= 65535 && d.H > 0 && d.W > 0 && d.cin > 0 && d.cout > 0 && d.cout <= 64,
                   "risp_conv2d: bad shape N=%d H=%d W=%d cin=%d cout=%d", d.N, d.H, d.W, d.cin, d.cout);
    RISP_CHECK_ARG((d.epilogue & RISP_EPI_NOBIAS) || d.bias, "risp_conv2d: bias missing");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_ADD) || (d.add && d.add_c > 0), "risp_conv2d: add tensor missing");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_MASK) || d.mask, "risp_conv2d: mask tensor missing");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_SHUFFLE2) || (d.cout % 4 == 0 && !(d.epilogue & (RISP_EPI_MASK))),
                   "risp_conv2d: PixelShuffle store needs cout %% 4 == 0 and no mask");
    RISP_CHECK_ARG(d.load_mode != RISP_LOAD_UNSHUFFLE2 || d.cin % 4 == 0, "risp_conv2d: unshuffle load needs cin %% 4 == 0");
    RISP_CHECK_ARG(d.load_mode != RISP_LOAD_CONSTCH || (d.cvals && d.cin_img > 0 && d.cin_img <= d.cin),
                   "risp_conv2d: const-channel load needs cvals and cin_img");
    const ConvCfg c = conv_cfg(d.cin, d.cout, d.ksize);
    hipStream_t s = (hipStream_t)stream;
#define RISP_CONV_CASE(KS, CK, CB) \
    if (d.ksize == KS && c.ck == CK && c.cb == CB) return launch_conv<KS, CK, CB>(d, s);
    RISP_CONV_CASE(3, 8, 2)
    RISP_CONV_CASE(3, 8, 1)
    RISP_CONV_CASE(3, 4, 2)
    RISP_CONV_CASE(3, 4, 1)
    RISP_CONV_CASE(5, 8, 1)
    RISP_CONV_CASE(5, 4, 1)
    RISP_CONV_CASE(5, 4, 2)
    RISP_CONV_CASE(9, 2, 2)
    RISP_CONV_CASE(9, 2, 1)
    RISP_CONV_CASE(1, 16, 2)
    RISP_CONV_CASE(1, 16, 1)
#undef RISP_CONV_CASE
    risp_set_error("risp_conv2d: unsupported kernel size %d", d.ksize);
    return 1;
}

}  // extern "C"
