// Backward-weight convolution on the fp32 matrix cores: dW[co][ci][ky][kx] = sum over (n,y,x) of
// gy[n,co,y,x] * x[n,ci,y+ky-p,x+kx-p].  Needed only where proxy weights are trained - the online proxy
// fine-tuning of the search (reference: models/darts_ft_model.py:206-246, which back-propagates an MSE between
// an SRCNNRes proxy and its classical teacher into the proxy's three conv layers).
//
// Per filter tap this is a GEMM  D[co][ci] += Gy[co][pix] * Xs[pix][ci]  with the pixels as the reduction
// dimension.  A workgroup owns one (32 cout) x (32 cin) block and a fixed SLICE of the pixel tiles (16 x 8 pixels of one image:
// tiles slice, slice + S, ...); per tile both operands sit in LDS with an odd plane stride (lanes 0-31 read 32 different
// channels of the same pixel: conflict-free); the waves split the k*k taps, each keeps the 32 x 32 sums of its taps in registers
// over the whole walk (a 128-pixel v_mfma_f32_32x32x2_f32 chain per tile and tap) and stores them once, to its own slot of a
// [block][slice][tap][cout][cin] scratch.  A finishing kernel adds the slices IN INDEX ORDER and transposes into the
// (cout,cin,k,k) layout PyTorch uses: no atomics, the same bits on every run (round 4 added the tiles with float atomics).
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

// NW waves; wave w owns taps w, w + NW, ...: at most NTAP of them.  MODE packs the filter COLUMN into the matrix for thin layers:
//   0  rows = cout, columns = cin, one matrix chain per tap (ky, kx): k * k taps;
//   1  cin * k <= 32 (the 3 image channels of SRCNNRes' 9x9 first layer): columns = (ci, kx) - lane (ci, kx) reads its channel kx
//      pixels to the right -, one chain per filter ROW: k taps instead of k * k on columns that would be 3/32 full;
//   2  cout * k <= 32 (the 5x5 32 -> 3 last layer): rows = (co, kx) - the row reads gy kx pixels to the LEFT while the reduction
//      walks the input tile with its halo columns (x' = x + kx - p: gy[co][x' - kx + p] x[ci][x']) -, one chain per filter row.
template <int K, int NW, int MODE>
__global__ __launch_bounds__(64 * NW) void conv_wgrad_kernel(const risp_conv_desc d, const float *__restrict__ gy,
                                                               float *__restrict__ scratch, int cib, int tiles_x, int tiles_y) {
    extern __shared__ float lds[];
    constexpr int P = K / 2, XW = GW + K - 1, XH = GH + K - 1, XS = (XW * XH) | 1, TAPS = MODE == 0 ? K * K : K;
    constexpr int NTAP = (TAPS + NW - 1) / NW, NT = 64 * NW;
    float *sg = lds;                               // [32][GYS]
    float *sx = lds + 32 * GYS;                    // [32][XS]
    const int blk = blockIdx.y, slice = blockIdx.x, nslice = gridDim.x;
    const int co0 = (blk / cib) * 32, ci0 = (blk % cib) * 32;
    const size_t plane = (size_t)d.H * d.W;
    const int nci = d.cin < 32 ? d.cin : 32;           // LDS holds only the channels that exist
    const int nco = MODE == 2 ? d.cout : 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    // this lane's column: channel + column shift (MODE 1), or channel; lanes of absent columns feed zeros
    const int bci = MODE == 1 ? l31 / K : l31, bkx = MODE == 1 ? l31 - bci * K : 0;
    const bool has_b = bci < nci;
    // this lane's row: cout + column shift (MODE 2), or cout
    const int aco = MODE == 2 ? l31 / K : l31, akx = MODE == 2 ? l31 - aco * K : 0;
    const bool has_a = MODE == 2 ? aco < d.cout : true;
    f32x16 acc[NTAP];
#pragma unroll
    for (int i = 0; i < NTAP; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const int ntiles = d.N * tiles_x * tiles_y;
    for (int t = slice; t < ntiles; t += nslice) {
        const int n = t / (tiles_x * tiles_y), rem = t - n * (tiles_x * tiles_y);
        const int x0 = (rem % tiles_x) * GW, y0 = (rem / tiles_x) * GH;
        __syncthreads();                               // every wave has left the previous tile
        // staging: loads from clamped addresses, zeros selected afterwards - no branch around a load, so that the loads of an
        // unrolled group are in flight together (a tile is 30 loads per thread: one round trip each was 30 us per tile)
        const float *gyn = gy + (size_t)n * d.cout * plane;
#pragma unroll 4
        for (int idx = threadIdx.x; idx < nco * GP; idx += NT) {
            const int c = idx / GP, p = idx - c * GP;
            const int py = y0 + p / GW, px = x0 + p % GW, co = co0 + c;
            const bool ok = co < d.cout && py < d.H && px < d.W;
            const float v = gyn[ok ? (size_t)co * plane + (size_t)py * d.W + px : 0];
            sg[c * GYS + p] = ok ? v : 0.f;
        }
        if (d.load_mode == RISP_LOAD_PLAIN) {
            const float *xn = d.x + (size_t)n * d.cin * plane;
#pragma unroll 4
            for (int idx = threadIdx.x; idx < nci * XW * XH; idx += NT) {
                const int c = idx / (XW * XH), r2 = idx - c * (XW * XH);
                const int ty = r2 / XW, tx = r2 - ty * XW, py = y0 + ty - P, px = x0 + tx - P, ci = ci0 + c;
                const bool ok = ci < d.cin && py >= 0 && py < d.H && px >= 0 && px < d.W;
                const float v = xn[ok ? (size_t)ci * plane + (size_t)py * d.W + px : 0];
                sx[c * XS + r2] = ok ? v : 0.f;
            }
        } else {
            for (int idx = threadIdx.x; idx < nci * XW * XH; idx += NT) {
                const int c = idx / (XW * XH), r2 = idx - c * (XW * XH);
                const int ty = r2 / XW, tx = r2 - ty * XW;
                sx[c * XS + r2] = wg_load_x(d, n, ci0 + c, y0 + ty - P, x0 + tx - P);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NTAP; ++i) {
            const int tap = wave + NW * i;
            if (tap < TAPS) {
                if (MODE == 2) {
                    // reduction over the GH x XW positions x' of the input tile rows ky .. ky + GH - 1
                    const float *ga = sg + (has_a ? aco : 0) * GYS - akx;
                    const float *xb = sx + (has_b ? bci : 0) * XS + tap * XW;
#pragma unroll 8
                    for (int p = 0; p < GH * XW; p += 2) {
                        const int yy = p / XW, xx = p - yy * XW + half;          // (XW even: both halves of a pair sit in one row)
                        const float av = (has_a && xx >= akx && xx - akx < GW) ? ga[yy * GW + xx] : 0.f;
                        const float bv = has_b ? xb[p + half] : 0.f;
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
                    }
                } else {
                    const int ky = MODE == 0 ? tap / K : tap, kx = MODE == 0 ? tap - ky * K : bkx;
                    const float *ga = sg + l31 * GYS + half;
                    const float *xb = sx + (has_b ? bci : 0) * XS + ky * XW + kx;
#pragma unroll 8
                    for (int p = 0; p < GP; p += 2) {
                        const int q = p + half;                              // this lane-half's pixel
                        const float bv = has_b ? xb[(q / GW) * XW + (q % GW)] : 0.f;
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[p], bv, acc[i], 0, 0, 0);
                    }
                }
            }
        }
    }
    // D[row][col]; lanes 0-31 of a register hold the 32 columns of one row: 128-byte stores into this workgroup's slot
    float *out = scratch + ((size_t)blk * nslice + slice) * TAPS * 1024;
#pragma unroll
    for (int i = 0; i < NTAP; ++i) {
        const int tap = wave + NW * i;
        if (tap < TAPS) {
#pragma unroll
            for (int e = 0; e < 16; ++e) out[((size_t)tap * 32 + (e & 3) + 8 * (e >> 2) + 4 * half) * 32 + l31] = acc[i][e];
        }
    }
}

// dw[co][ci][ky][kx] = the slices of its block added in index order (four running sums, combined in a fixed tree)
template <int MODE>
__global__ void wgrad_finish_kernel(const float *__restrict__ scratch, float *__restrict__ dw, int cin, int cout, int K, int cib, int nslice) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, k2 = K * K;
    if (i >= cout * cin * k2) return;
    const int t = i % k2, ci = (i / k2) % cin, co = i / (k2 * cin), ky = t / K, kx = t - ky * K;
    const int taps = MODE == 0 ? k2 : K, tap = MODE == 0 ? t : ky;
    const int row = MODE == 2 ? co * K + kx : (co & 31), col = MODE == 1 ? ci * K + kx : (ci & 31);
    const int blk = MODE == 0 ? (co >> 5) * cib + (ci >> 5) : (MODE == 1 ? (co >> 5) : (ci >> 5));
    const float *src = scratch + (size_t)blk * nslice * taps * 1024 + ((size_t)tap * 32 + row) * 32 + col;
    const size_t step = (size_t)taps * 1024;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 4 <= nslice; k += 4) {
        s0 += src[(size_t)k * step];
        s1 += src[(size_t)(k + 1) * step];
        s2 += src[(size_t)(k + 2) * step];
        s3 += src[(size_t)(k + 3) * step];
    }
    for (; k < nslice; ++k) s0 += src[(size_t)k * step];
    dw[i] = (s0 + s1) + (s2 + s3);
}

constexpr int WG_TOTAL = 768;                      // most workgroups of a launch (the scratch holds a slot for each), dealt over the blocks of the matrix

template <int K, int NW, int MODE>
int launch_wgrad(const risp_conv_desc &d, const float *gy, float *dw, float *scratch, hipStream_t s) {
    constexpr int XS = ((GW + K - 1) * (GH + K - 1)) | 1;
    const int cob = MODE == 2 ? 1 : (d.cout + 31) / 32, cib = MODE == 1 ? 1 : (d.cin + 31) / 32, nci = d.cin < 32 ? d.cin : 32;
    const size_t lds = sizeof(float) * (32 * (size_t)GYS + (size_t)nci * XS);
    const int tx = (d.W + GW - 1) / GW, ty = (d.H + GH - 1) / GH;
    const long long ntiles = (long long)d.N * tx * ty;
    RISP_CHECK_ARG(lds <= 64 * 1024 && ntiles <= 0x7fffffff, "risp_conv2d_wgrad: tile too large for LDS, or too many tiles");
    // one round of resident workgroups: what the kernel's registers and LDS admit per CU (2 for the 25-tap 5x5 form, 3-4 for the rest)
    int occ = 0, cus = 0, dev = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, conv_wgrad_kernel<K, NW, MODE>, 64 * NW, lds) != hipSuccess || occ < 1) occ = 1;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    const int nwg = occ * cus < WG_TOTAL ? occ * cus : WG_TOTAL;
    const int per_block = nwg / (cob * cib) > 0 ? nwg / (cob * cib) : 1;
    const int nslice = ntiles < per_block ? (int)ntiles : per_block;
    hipLaunchKernelGGL((conv_wgrad_kernel<K, NW, MODE>), dim3(nslice, cob * cib), dim3(64 * NW), lds, s, d, gy, scratch, cib, tx, ty);
    const int total = d.cout * d.cin * K * K;
    hipLaunchKernelGGL(wgrad_finish_kernel<MODE>, dim3((total + 255) / 256), dim3(256), 0, s, scratch, dw, d.cin, d.cout, K, cib, nslice);
    return 0;
}

}  // namespace

extern "C" {

// slots of 1024 floats: one per workgroup and tap chain - k * k chains per workgroup in the plain form, k where the filter column is
// packed into the matrix (thin layers, see conv_wgrad_kernel)
size_t risp_conv_wgrad_scratch_floats(int cin, int cout, int ksize) {
    const bool thin = ksize > 1 && (cin * ksize <= 32 || cout * ksize <= 32);
    return (size_t)WG_TOTAL * (thin ? ksize : ksize * ksize) * 1024;
}

int risp_conv2d_wgrad(const risp_conv_desc *dp, const float *gy, float *dw, float *scratch, size_t scratch_floats, void *stream) {
    RISP_CHECK_ARG(dp && gy && dw && scratch, "risp_conv2d_wgrad: null argument");
    const risp_conv_desc &d = *dp;
    RISP_CHECK_ARG(scratch_floats >= risp_conv_wgrad_scratch_floats(d.cin, d.cout, d.ksize),
                   "risp_conv2d_wgrad: scratch holds %zu floats, needs risp_conv_wgrad_scratch_floats(cin, cout, ksize) = %zu", scratch_floats,
                   risp_conv_wgrad_scratch_floats(d.cin, d.cout, d.ksize));
    RISP_CHECK_ARG(d.group_n == 0, "risp_conv2d_wgrad: grouped descriptors are not supported (one weight gradient per launch)");
    RISP_CHECK_ARG(d.x && d.N > 0 && d.H > 0 && d.W > 0 && d.cin > 0 && d.cin <= 64 && d.cout > 0 && d.cout <= 64 &&
                       (d.ksize == 1 || d.ksize == 3 || d.ksize == 5 || d.ksize == 9),
                   "risp_conv2d_wgrad: unsupported layer cin=%d cout=%d k=%d", d.cin, d.cout, d.ksize);
    RISP_CHECK_ARG(d.load_mode == RISP_LOAD_PLAIN || (d.load_mode == RISP_LOAD_CONSTCH && d.cvals && d.cin_img > 0),
                   "risp_conv2d_wgrad: load mode %d not supported", d.load_mode);
    hipStream_t s = (hipStream_t)stream;
    int st;
    const bool thin_in = d.cin * d.ksize <= 32 && d.ksize > 1, thin_out = !thin_in && d.cout * d.ksize <= 32 && d.ksize > 1;
    switch (d.ksize) {
        case 1: st = launch_wgrad<1, 1, 0>(d, gy, dw, scratch, s); break;
        case 3: st = thin_in ? launch_wgrad<3, 1, 1>(d, gy, dw, scratch, s) : (thin_out ? launch_wgrad<3, 1, 2>(d, gy, dw, scratch, s) : launch_wgrad<3, 4, 0>(d, gy, dw, scratch, s)); break;
        case 5:         // plain: 7 taps = 112 accumulator registers per wave
            st = thin_in ? launch_wgrad<5, 4, 1>(d, gy, dw, scratch, s) : (thin_out ? launch_wgrad<5, 4, 2>(d, gy, dw, scratch, s) : launch_wgrad<5, 4, 0>(d, gy, dw, scratch, s)); break;
        default:        // plain: 8 waves x 11 taps = 176
            st = thin_in ? launch_wgrad<9, 4, 1>(d, gy, dw, scratch, s) : (thin_out ? launch_wgrad<9, 4, 2>(d, gy, dw, scratch, s) : launch_wgrad<9, 8, 0>(d, gy, dw, scratch, s)); break;
    }
    if (st) return st;
    RISP_LAUNCH_CHECK("risp_conv2d_wgrad");
    return 0;
}

}  // extern "C"
