// Backward-weight convolution on the fp32 matrix cores: dW[co][ci][ky][kx] = sum over (n,y,x) of
// gy[n,co,y,x] * x[n,ci,y+ky-p,x+kx-p].  Needed only where proxy weights are trained - the online proxy
// fine-tuning of the search (reference: models/darts_ft_model.py:206-246, which back-propagates an MSE between
// an SRCNNRes proxy and its classical teacher into the proxy's three conv layers).
//
// Per filter tap this is a GEMM  D[co][ci] += Gy[co][pix] * Xs[pix][ci]  with the pixels as the reduction
// dimension.  Workgroup = one 16 x 8 pixel tile of one image x one (32 cout) x (32 cin) block; both tiles sit
// in LDS with an odd plane stride (lanes 0-31 read 32 different channels of the same pixel: conflict-free);
// the four waves split the k*k taps and run a 128-pixel v_mfma_f32_32x32x2_f32 chain per tap, then add their
// 32 x 32 result to a [tap][cout][cin] scratch with 128-byte contiguous float atomics.  A finishing kernel
// transposes the scratch into the (cout,cin,k,k) layout PyTorch uses.
#include "risp_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int GW = 16, GH = 8, GP = GW * GH;       // pixel tile
constexpr int GYS = GP + 1;                        // odd plane stride of the gy tile

__device__ __forceinline__ float wg_load_x(const risp_conv_desc &d, int n, int ci, int gy, int gx) {
    if (ci >= d.cin || gy < 0 || gy >= d.H || gx < 0 || gx >= d.W) return 0.f;
    if (d.load_mode == RISP_LOAD_CONSTCH) {
        if (ci < d.cin_img) return d.x[(((size_t)n * d.cin_img + ci) * d.H + gy) * d.W + gx];
        return d.cvals[n * (d.cin - d.cin_img) + (ci - d.cin_img)];
    }
    return d.x[(((size_t)n * d.cin + ci) * d.H + gy) * d.W + gx];
}

__global__ __launch_bounds__(256) void conv_wgrad_kernel(const risp_conv_desc d, const float *__restrict__ gy,
                                                         float *__restrict__ scratch, int cob, int cib) {
    extern __shared__ float lds[];
    const int K = d.ksize, P = K / 2, XW = GW + K - 1, XH = GH + K - 1;
    const int XS = (XW * XH) | 1;                  // odd plane stride
    float *sg = lds;                               // [32][GYS]
    float *sx = lds + 32 * GYS;                    // [32][XS]
    const int blk = blockIdx.z % (cob * cib), n = blockIdx.z / (cob * cib);
    const int co0 = (blk / cib) * 32, ci0 = (blk % cib) * 32;
    const int x0 = blockIdx.x * GW, y0 = blockIdx.y * GH;
    const size_t plane = (size_t)d.H * d.W;

    for (int idx = threadIdx.x; idx < 32 * GP; idx += 256) {
        const int c = idx / GP, p = idx - c * GP;
        const int py = y0 + p / GW, px = x0 + p % GW, co = co0 + c;
        sg[c * GYS + p] = (co < d.cout && py < d.H && px < d.W) ? gy[((size_t)n * d.cout + co) * plane + (size_t)py * d.W + px] : 0.f;
    }
    const int nci = d.cin < 32 ? d.cin : 32;           // LDS holds only the channels that exist
    for (int idx = threadIdx.x; idx < nci * XW * XH; idx += 256) {
        const int c = idx / (XW * XH), rem = idx - c * (XW * XH);
        const int ty = rem / XW, tx = rem - ty * XW;
        sx[c * XS + rem] = wg_load_x(d, n, ci0 + c, y0 + ty - P, x0 + tx - P);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    for (int tap = wave; tap < K * K; tap += 4) {
        const int ky = tap / K, kx = tap - ky * K;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const float *ga = sg + l31 * GYS + half;
        const bool has_ci = l31 < nci;                 // lanes of absent channels feed zeros
        const float *xb = sx + (has_ci ? l31 : 0) * XS + ky * XW + kx;
#pragma unroll 8
        for (int p = 0; p < GP; p += 2) {
            const int q = p + half;                              // this lane-half's pixel
            const float bv = has_ci ? xb[(q / GW) * XW + (q % GW)] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[p], bv, acc, 0, 0, 0);
        }
        // D[row = co][col = ci]; lanes 0-31 of a register hold 32 consecutive ci of one co: 128-byte atomics
        const int ci = ci0 + l31;
        if (ci < d.cin) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (co < d.cout) atomicAdd(&scratch[((size_t)tap * 64 + co) * 64 + ci], acc[e]);
            }
        }
    }
}

__global__ void wgrad_finish_kernel(const float *__restrict__ scratch, float *__restrict__ dw, int cin, int cout,
                                    int taps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cout * cin * taps) return;
    const int tap = i % taps, ci = (i / taps) % cin, co = i / (taps * cin);
    dw[i] = scratch[((size_t)tap * 64 + co) * 64 + ci];
}

}  // namespace

extern "C" {

size_t risp_conv_wgrad_scratch_floats(int ksize) { return (size_t)ksize * ksize * 64 * 64; }

int risp_conv2d_wgrad(const risp_conv_desc *dp, const float *gy, float *dw, float *scratch, void *stream) {
    RISP_CHECK_ARG(dp && gy && dw && scratch, "risp_conv2d_wgrad: null argument");
    const risp_conv_desc &d = *dp;
    RISP_CHECK_ARG(d.group_n == 0, "risp_conv2d_wgrad: grouped descriptors are not supported (one weight gradient per launch)");
    RISP_CHECK_ARG(d.x && d.N > 0 && d.H > 0 && d.W > 0 && d.cin > 0 && d.cin <= 64 && d.cout > 0 && d.cout <= 64 &&
                       (d.ksize == 1 || d.ksize == 3 || d.ksize == 5 || d.ksize == 9),
                   "risp_conv2d_wgrad: unsupported layer cin=%d cout=%d k=%d", d.cin, d.cout, d.ksize);
    RISP_CHECK_ARG(d.load_mode == RISP_LOAD_PLAIN || (d.load_mode == RISP_LOAD_CONSTCH && d.cvals && d.cin_img > 0),
                   "risp_conv2d_wgrad: load mode %d not supported", d.load_mode);
    const int cob = (d.cout + 31) / 32, cib = (d.cin + 31) / 32;
    const size_t xs = (size_t)(((GW + d.ksize - 1) * (GH + d.ksize - 1)) | 1);
    const int nci = d.cin < 32 ? d.cin : 32;           // channels of a cin block that actually hold data
    const size_t lds = sizeof(float) * (32 * (size_t)GYS + (size_t)nci * xs);
    RISP_CHECK_ARG(lds <= 64 * 1024, "risp_conv2d_wgrad: LDS tile too large");
    RISP_CHECK_ARG((size_t)d.N * cob * cib <= 65535, "risp_conv2d_wgrad: batch too large for one launch");
    hipStream_t s = (hipStream_t)stream;
    const size_t nscr = risp_conv_wgrad_scratch_floats(d.ksize);
    if (hipMemsetAsync(scratch, 0, sizeof(float) * nscr, s) != hipSuccess) {
        risp_set_error("risp_conv2d_wgrad: memset failed");
        return 2;
    }
    dim3 grid((d.W + GW - 1) / GW, (d.H + GH - 1) / GH, d.N * cob * cib);
    hipLaunchKernelGGL(conv_wgrad_kernel, grid, dim3(256), lds, s, d, gy, scratch, cob, cib);
    const int total = d.cout * d.cin * d.ksize * d.ksize;
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3((total + 255) / 256), dim3(256), 0, s, scratch, dw, d.cin, d.cout,
                       d.ksize * d.ksize);
    RISP_LAUNCH_CHECK("risp_conv2d_wgrad");
    return 0;
}

}  // extern "C"
