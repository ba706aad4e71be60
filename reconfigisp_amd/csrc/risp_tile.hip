// Overlapped tiling of a full frame (utils/util_path_restore.py:47-134) on the device.
// gather = whole2patch's crop; blend = patch2whole with the linear edge-ramp mask and the
// count-map normalisation, evaluated per output pixel in tile order (same fp32 operation
// order as the reference's sequential "+=" loop).
#include "risp_common.h"

namespace {

__device__ __forceinline__ float ramp(int i, int size, int e) {
    // create_patch_mask (:56-63): (i+1)/(e+1) on the first e and (size-i)/(e+1) on the last e entries
    if (i < e) return (float)(i + 1) / (float)(e + 1);
    if (i >= size - e) return (float)(size - i) / (float)(e + 1);
    return 1.f;
}

__global__ __launch_bounds__(256) void tile_gather_kernel(const float *__restrict__ img, float *__restrict__ patches,
                                                          const int32_t *__restrict__ pos, int C, int H, int W, int h,
                                                          int w) {
    const int t = blockIdx.z, c = blockIdx.y;
    const int py = pos[2 * t], px = pos[2 * t + 1];
    const float *src = img + (size_t)c * H * W;
    float *dst = patches + ((size_t)t * C + c) * h * w;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < h * w; i += gridDim.x * blockDim.x) {
        const int y = i / w, x = i - y * w;
        dst[i] = src[(size_t)(py + y) * W + px + x];
    }
}

// One workgroup = 256 pixels of BLEND_ROWS image rows.  The tiles that reach into that block are listed first (in tile order, by the whole
// workgroup: a ballot per 64 tiles; index and corner in LDS) - a pixel then walks that list instead of all T tiles (63 tiles of a
// 4000 x 3000 frame: 1-6 reach a block): same adds in the same order, 396 -> ~1/3 of the time per frame.
constexpr int BLEND_ROWS = 4;
__global__ __launch_bounds__(256) void tile_blend_kernel(const float *__restrict__ patches, float *__restrict__ img,
                                                         const int32_t *__restrict__ pos, int T, int C, int H, int W,
                                                         int h, int w, int eh, int ew) {
    constexpr int LIST = 256;                          // listed tiles kept in LDS; a longer list falls back to the full walk
    __shared__ int list[LIST], list_y[LIST], list_x[LIST];
    __shared__ int wave_n[4];
    const int X0 = blockIdx.x * blockDim.x, X = X0 + threadIdx.x, Y0 = blockIdx.y * BLEND_ROWS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int n_list = 0;
    bool listed = true;
    for (int t0 = 0; t0 < T && listed; t0 += 256) {     // (uniform: every thread sees the same counts)
        const int t = t0 + threadIdx.x;
        bool in = false;
        int ty = 0, tx = 0;
        if (t < T) {
            ty = pos[2 * t];
            tx = pos[2 * t + 1];
            in = Y0 + BLEND_ROWS - 1 >= ty && Y0 < ty + h && X0 + 255 >= tx && X0 < tx + w;
        }
        const unsigned long long m = __ballot(in);
        if (lane == 0) wave_n[wave] = __popcll(m);
        __syncthreads();
        int before = n_list;
        for (int k = 0; k < wave; ++k) before += wave_n[k];
        const int all = n_list + wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3];
        if (all > LIST) {
            listed = false;
        } else if (in) {
            const int at = before + __popcll(m & ((1ull << lane) - 1ull));
            list[at] = t;
            list_y[at] = ty;
            list_x[at] = tx;
        }
        n_list = all;
        __syncthreads();
    }
    if (X >= W) return;
    const int n = listed ? n_list : T;
    for (int Y = Y0; Y < Y0 + BLEND_ROWS && Y < H; ++Y) {
        float cnt = 0.f;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < n; ++i) {
            const int t = listed ? list[i] : i;
            const int y = Y - (listed ? list_y[i] : pos[2 * t]), x = X - (listed ? list_x[i] : pos[2 * t + 1]);
            if (y < 0 || y >= h || x < 0 || x >= w) continue;
            const float m = fminf(ramp(y, h, eh), ramp(x, w, ew));
            cnt += m;
            const float *p = patches + ((size_t)t * C) * h * w + (size_t)y * w + x;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < C) acc[c] += p[(size_t)c * h * w] * m;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < C) img[((size_t)c * H + Y) * W + X] = acc[c] / cnt;
    }
}

}  // namespace

extern "C" {

int risp_tile_gather(const float *img, float *patches, const int32_t *pos_dev, int T, int C, int H, int W, int h, int w,
                     void *stream) {
    RISP_CHECK_ARG(img && patches && pos_dev && T > 0 && T <= 65535 && C > 0 && C <= 65535 && h > 0 && w > 0 && h <= H && w <= W,
                   "risp_tile_gather: bad arguments");
    int bx = (h * w + 255) / 256;
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(tile_gather_kernel, dim3(bx, C, T), dim3(256), 0, (hipStream_t)stream, img, patches, pos_dev, C, H,
                       W, h, w);
    RISP_LAUNCH_CHECK("risp_tile_gather");
    return 0;
}

int risp_tile_blend(const float *patches, float *img, const int32_t *pos_dev, int T, int C, int H, int W, int h, int w,
                    int eh, int ew, void *stream) {
    RISP_CHECK_ARG(patches && img && pos_dev && T > 0 && C > 0 && C <= 4 && h > 0 && w > 0 && h <= H && w <= W && H <= 65535,
                   "risp_tile_blend: bad arguments (C must be <= 4)");
    RISP_CHECK_ARG(eh >= 0 && ew >= 0 && eh <= h / 2 && ew <= w / 2, "risp_tile_blend: edge larger than half a tile");
    hipLaunchKernelGGL(tile_blend_kernel, dim3((W + 255) / 256, (H + BLEND_ROWS - 1) / BLEND_ROWS), dim3(256), 0, (hipStream_t)stream, patches, img,
                       pos_dev, T, C, H, W, h, w, eh, ew);
    RISP_LAUNCH_CHECK("risp_tile_blend");
    return 0;
}

}  // extern "C"
