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

__global__ __launch_bounds__(256) void tile_blend_kernel(const float *__restrict__ patches, float *__restrict__ img,
                                                         const int32_t *__restrict__ pos, int T, int C, int H, int W,
                                                         int h, int w, int eh, int ew) {
    const int X = blockIdx.x * blockDim.x + threadIdx.x, Y = blockIdx.y;
    if (X >= W) return;
    float cnt = 0.f;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < T; ++t) {
        const int y = Y - pos[2 * t], x = X - pos[2 * t + 1];
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
    hipLaunchKernelGGL(tile_blend_kernel, dim3((W + 255) / 256, H), dim3(256), 0, (hipStream_t)stream, patches, img,
                       pos_dev, T, C, H, W, h, w, eh, ew);
    RISP_LAUNCH_CHECK("risp_tile_blend");
    return 0;
}

}  // extern "C"
