// Input side of the path on the device: what the reference's dataset classes do on
// the host with numpy / cv2 before the first ISP stage sees a tensor -
//   * crop an even-aligned window out of a uint16 RGGB frame and scale it to fp32 [0,1]
//     (data/sid_sony_ratio_rggb2bgr_dataset.py:121-134: /16383; oneplus / s7isp: /1023),
//   * crop the uint8 HWC BGR ground truth to NCHW fp32 /255 (same lines),
//   * the OnePlus "resize by quad" (nearest-neighbour resize of the four colour planes + zero rows above and
//     below, data/oneplus_rggb2obj_dataset.py:109-145 / data/util.py:37-64).
// One thread per output element, 16-bit / 8-bit gathers; the frames stay in HBM as integers (2 B/pix).
#include "risp_common.h"

namespace {

// sel: (N,3) int32 = {frame, row, col}; out (N,1,h,w) = frames[frame][row+y][col+x] / divisor
__global__ __launch_bounds__(256) void raw_crop_kernel(const uint16_t *__restrict__ frames, float *__restrict__ out,
                                                       const int32_t *__restrict__ sel, int H0, int W0, int h, int w,
                                                       float divisor) {
    const int n = blockIdx.y;
    const int f = sel[3 * n], r = sel[3 * n + 1], c = sel[3 * n + 2];
    const uint16_t *src = frames + (size_t)f * H0 * W0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < h * w; i += gridDim.x * blockDim.x) {
        const int y = i / w, x = i - y * w;
        out[(size_t)n * h * w + i] = (float)src[(size_t)(r + y) * W0 + c + x] / divisor;
    }
}

// gt frames (F,H0,W0,3) uint8 HWC BGR -> out (N,3,h,w) fp32 / 255
__global__ __launch_bounds__(256) void gt_crop_kernel(const uint8_t *__restrict__ frames, float *__restrict__ out,
                                                      const int32_t *__restrict__ sel, int H0, int W0, int h, int w) {
    const int n = blockIdx.y;
    const int f = sel[3 * n], r = sel[3 * n + 1], c = sel[3 * n + 2];
    const uint8_t *src = frames + (size_t)f * H0 * W0 * 3;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < h * w; i += gridDim.x * blockDim.x) {
        const int y = i / w, x = i - y * w;
        const uint8_t *p = src + ((size_t)(r + y) * W0 + c + x) * 3;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) out[((size_t)n * 3 + ch) * h * w + i] = (float)p[ch] / 255.f;
    }
}

// src (H0,W0) RGGB uint16 -> dst (H,W) uint16: planes resized to (Hr/2, W/2) by nearest neighbour, placed
// pad_top/2 plane rows down, zero elsewhere
__global__ __launch_bounds__(256) void resize_rggb_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst,
                                                          int H0, int W0, int H, int W, int Hr, int pad_top) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i - y * W;
    const int py = (y >> 1) - (pad_top >> 1), px = x >> 1;        // plane coordinates inside the resized image
    const int ph = Hr >> 1, pw = W >> 1, sh = H0 >> 1, sw = W0 >> 1;
    uint16_t v = 0;
    if (py >= 0 && py < ph) {
        int sy = (int)floor((double)py * ((double)sh / (double)ph)), sx = (int)floor((double)px * ((double)sw / (double)pw));
        sy = sy < sh - 1 ? sy : sh - 1;
        sx = sx < sw - 1 ? sx : sw - 1;
        v = src[(size_t)(2 * sy + (y & 1)) * W0 + 2 * sx + (x & 1)];
    }
    dst[i] = v;
}

}  // namespace

extern "C" {

int risp_raw_crop(const uint16_t *frames, float *out, const int32_t *sel, int N, int H0, int W0, int h, int w,
                  float divisor, void *stream) {
    RISP_CHECK_ARG(frames && out && sel && N > 0 && N <= 65535 && h > 0 && w > 0 && h <= H0 && w <= W0 && divisor > 0.f,
                   "risp_raw_crop: bad arguments");
    int bx = (h * w + 256 * 8 - 1) / (256 * 8);
    bx = bx < 1 ? 1 : (bx > 256 ? 256 : bx);
    hipLaunchKernelGGL(raw_crop_kernel, dim3(bx, N), dim3(256), 0, (hipStream_t)stream, frames, out, sel, H0, W0, h, w,
                       divisor);
    RISP_LAUNCH_CHECK("risp_raw_crop");
    return 0;
}

int risp_gt_crop(const uint8_t *frames, float *out, const int32_t *sel, int N, int H0, int W0, int h, int w,
                 void *stream) {
    RISP_CHECK_ARG(frames && out && sel && N > 0 && N <= 65535 && h > 0 && w > 0 && h <= H0 && w <= W0,
                   "risp_gt_crop: bad arguments");
    int bx = (h * w + 256 * 8 - 1) / (256 * 8);
    bx = bx < 1 ? 1 : (bx > 256 ? 256 : bx);
    hipLaunchKernelGGL(gt_crop_kernel, dim3(bx, N), dim3(256), 0, (hipStream_t)stream, frames, out, sel, H0, W0, h, w);
    RISP_LAUNCH_CHECK("risp_gt_crop");
    return 0;
}

int risp_resize_rggb(const uint16_t *src, uint16_t *dst, int H0, int W0, int H, int W, int resized_h, int pad_top,
                     void *stream) {
    RISP_CHECK_ARG(src && dst && H0 >= 2 && W0 >= 2 && H0 % 2 == 0 && W0 % 2 == 0 && H % 2 == 0 && W % 2 == 0 &&
                       resized_h >= 2 && resized_h % 2 == 0 && pad_top >= 0 && pad_top % 2 == 0 && pad_top + resized_h <= H,
                   "risp_resize_rggb: bad arguments");
    hipLaunchKernelGGL(resize_rggb_kernel, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, dst, H0, W0,
                       H, W, resized_h, pad_top);
    RISP_LAUNCH_CHECK("risp_resize_rggb");
    return 0;
}

}  // extern "C"
