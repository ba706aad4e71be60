// The classical, non-differentiable "Origin" kernels used by OriginUniversal at test time
// (codes/models/modules/tools_origin.py:445-804; plugin call sites :457-468, :491-502, :526-543,
// :566-581, :604-623, :647-662, :686-710, :734-751, :775-797).  Their arithmetic lives in the absent
// ISP_Kernels package: what is computed here is the build-defined OPSPEC restated in
// oracle/isp_oracle.py (origin_demosaic / origin_tonemap / origin_whiteworld / origin_denoise).
//
// Domain: NCHW fp32 images scaled to 0..255; outputs are clipped and rounded to 8-bit codes.
// Neighbourhood filters stage a (32+2R) x (8+2R) reflect-101 halo tile per colour plane in LDS
// (coalesced row reads, every neighbour access afterwards is an LDS read); tone curves and white
// balance are float4 plane streams with per-image scalars.
#include "risp_common.h"

namespace {

constexpr int TX = 32, TY = 8;   // output pixels per 256-thread block

__device__ __forceinline__ float q8(float v) {
    return floorf(__builtin_amdgcn_fmed3f(v, 0.f, 255.f) + 0.5f);   // clamp in one instruction (v is never NaN here)
}
// Store form of a result v in 0..255: the 8-bit code / out_div; out_div < 0 is the diagnostic form that skips the
// clip-and-round (v / |out_div|), so that tests can compare the arithmetic BEFORE quantisation at float tolerance.
__device__ __forceinline__ float emit(float v, float so) {
    return so > 0.f ? q8(v) * (1.f / so) : v * (1.f / -so);
}
__device__ __forceinline__ int reflect101(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    // positions further than one reflection away only feed outputs that lie outside the image
    // (tile overhang); keep their loads in bounds
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

// planes: C contiguous HxW planes of one image -> lds[C][TY+2R][TX+2R]
template <bool QUANT>
__device__ __forceinline__ void stage_tile(const float *__restrict__ planes, float *lds, int C, int H, int W, int x0,
                                           int y0, int R, float in_scale) {
    const int tw = TX + 2 * R, th = TY + 2 * R, per = tw * th;
    for (int idx = threadIdx.x; idx < C * per; idx += blockDim.x) {
        const int c = idx / per, rem = idx - c * per;
        const int ty = rem / tw, tx = rem - ty * tw;
        const int gy = reflect101(y0 + ty - R, H), gx = reflect101(x0 + tx - R, W);
        const float v = planes[((size_t)c * H + gy) * W + gx] * in_scale;
        lds[idx] = QUANT ? q8(v) : v;
    }
    __syncthreads();
}

// ---------------------------------------------------------------- demosaic: bilinear / Malvar-He-Cutler
template <bool LAPLACIAN>
__global__ __launch_bounds__(256) void origin_demosaic_kernel(const float *__restrict__ x, float *__restrict__ y, int H,
                                                              int W, float si, float so) {
    extern __shared__ float lds[];
    constexpr int R = 2, tw = TX + 2 * R;
    const int n = blockIdx.z, x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    stage_tile<false>(x + (size_t)n * H * W, lds, 1, H, W, x0, y0, R, si);
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int px = x0 + lx, py = y0 + ly;
    if (px >= W || py >= H) return;
    auto s = [&](int dy, int dx) { return lds[(ly + R + dy) * tw + lx + R + dx]; };
    const float c = s(0, 0);
    const float cross = s(-1, 0) + s(1, 0) + s(0, -1) + s(0, 1);
    const float diag = s(-1, -1) + s(-1, 1) + s(1, -1) + s(1, 1);
    const float hor = s(0, -1) + s(0, 1), ver = s(-1, 0) + s(1, 0);
    float g_rb, rb_hor, rb_ver, rb_diag;
    if (LAPLACIAN) {
        const float fh = s(0, -2) + s(0, 2), fv = s(-2, 0) + s(2, 0), far = fh + fv;
        g_rb = (4.f * c + 2.f * cross - far) / 8.f;
        rb_hor = (5.f * c + 4.f * hor - diag - fh + 0.5f * fv) / 8.f;
        rb_ver = (5.f * c + 4.f * ver - diag - fv + 0.5f * fh) / 8.f;
        rb_diag = (6.f * c + 2.f * diag - 1.5f * far) / 8.f;
    } else {
        g_rb = cross / 4.f;
        rb_hor = hor / 2.f;
        rb_ver = ver / 2.f;
        rb_diag = diag / 4.f;
    }
    const bool er = (py & 1) == 0, ec = (px & 1) == 0;   // R at (even,even), B at (odd,odd)
    float R_, G_, B_;
    if (er && ec) { R_ = c; G_ = g_rb; B_ = rb_diag; }
    else if (er && !ec) { G_ = c; R_ = rb_hor; B_ = rb_ver; }
    else if (!er && ec) { G_ = c; R_ = rb_ver; B_ = rb_hor; }
    else { B_ = c; G_ = g_rb; R_ = rb_diag; }
    const size_t plane = (size_t)H * W, o = (size_t)n * 3 * plane + (size_t)py * W + px;
    y[o] = emit(B_, so);
    y[o + plane] = emit(G_, so);
    y[o + 2 * plane] = emit(R_, so);
}

// ---------------------------------------------------------------- bilateral
__global__ __launch_bounds__(256) void bilateral_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                        const int *__restrict__ win, const float *__restrict__ sig_c,
                                                        const float *__restrict__ sig_s, int H, int W, int R,
                                                        float si, float so) {
    extern __shared__ float lds[];
    const int tw = TX + 2 * R, per = tw * (TY + 2 * R);
    const int n = blockIdx.z, x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    stage_tile<false>(x + (size_t)n * 3 * H * W, lds, 3, H, W, x0, y0, R, si);
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int px = x0 + lx, py = y0 + ly;
    if (px >= W || py >= H) return;
    int r = win[n] / 2;
    r = r < 0 ? 0 : (r > R ? R : r);                   // never walk outside the staged halo
    const float ks = -1.f / (2.f * sig_s[n] * sig_s[n]), kc = -1.f / (2.f * sig_c[n] * sig_c[n]);
    const float ks2 = ks * 1.4426950408889634f, kc2 = kc * 1.4426950408889634f;     // base-2 exponent coefficients
    const float *ctr = lds + (ly + R) * tw + lx + R;
    const float cb = ctr[0], cg = ctr[per], cr = ctr[2 * per];
    float nb = 0.f, ng = 0.f, nr = 0.f, den = 0.f;
    for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) {
            const float *q = ctr + dy * tw + dx;
            const float qb = q[0], qg = q[per], qr = q[2 * per];
            if (dy == 0 && dx == 0) {                  // weight exp(0) = 1 exactly
                nb += qb; ng += qg; nr += qr; den += 1.f;
                continue;
            }
            const float dist = fabsf(qb - cb) + fabsf(qg - cg) + fabsf(qr - cr);
            // same expressions as the fused segment kernel (risp_fused.hip): base-2 exponent by one fma, fma sums
            const float wgt = __builtin_amdgcn_exp2f(__builtin_fmaf(dist * dist, kc2, (float)(dy * dy + dx * dx) * ks2));
            nb = __builtin_fmaf(wgt, qb, nb); ng = __builtin_fmaf(wgt, qg, ng); nr = __builtin_fmaf(wgt, qr, nr); den += wgt;
        }
    const size_t plane = (size_t)H * W, o = (size_t)n * 3 * plane + (size_t)py * W + px;
    const float rden = 1.f / den;                       // OPSPEC: normalise by one reciprocal, not three divisions
    y[o] = emit(nb * rden, so);
    y[o + plane] = emit(ng * rden, so);
    y[o + 2 * plane] = emit(nr * rden, so);
}

// ---------------------------------------------------------------- median on 8-bit codes (bisection on the code)
__global__ __launch_bounds__(256) void median_kernel(const float *__restrict__ x, float *__restrict__ y, int H, int W,
                                                     int R, float si, float so) {
    extern __shared__ float lds[];
    const int tw = TX + 2 * R, per = tw * (TY + 2 * R);
    const int n = blockIdx.z, x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    stage_tile<true>(x + (size_t)n * 3 * H * W, lds, 3, H, W, x0, y0, R, si);
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int px = x0 + lx, py = y0 + ly;
    if (px >= W || py >= H) return;
    const int k = 2 * R + 1, need = (k * k) / 2 + 1;   // rank of the middle element
    const size_t plane = (size_t)H * W, o = (size_t)n * 3 * plane + (size_t)py * W + px;
    for (int c = 0; c < 3; ++c) {
        const float *ctr = lds + c * per + (ly + R) * tw + lx + R;
        int lo = 0, hi = 255;                       // smallest code v with count(<= v) >= need
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const float fm = (float)mid;
            int cnt = 0;
            for (int dy = -R; dy <= R; ++dy)
                for (int dx = -R; dx <= R; ++dx) cnt += ctr[dy * tw + dx] <= fm ? 1 : 0;
            if (cnt >= need) hi = mid; else lo = mid + 1;
        }
        y[o + c * plane] = (float)lo * (1.f / fabsf(so));
    }
}

// ---------------------------------------------------------------- non-local means
__global__ __launch_bounds__(256) void fastnlm_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                      const int *__restrict__ blk, const int *__restrict__ srch,
                                                      const float *__restrict__ decay, int H, int W, int R,
                                                      float si, float so) {
    extern __shared__ float lds[];
    const int tw = TX + 2 * R, per = tw * (TY + 2 * R);
    const int n = blockIdx.z, x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    stage_tile<false>(x + (size_t)n * 3 * H * W, lds, 3, H, W, x0, y0, R, si);
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int px = x0 + lx, py = y0 + ly;
    if (px >= W || py >= H) return;
    const int b = blk[n];
    int rb = b / 2, rs = srch[n] / 2;
    rb = rb < 0 ? 0 : (rb > R ? R : rb);
    rs = rs < 0 ? 0 : (rs > R - rb ? R - rb : rs);      // never walk outside the staged halo
    const float scale = -1.f / (3.f * (float)(b * b) * decay[n] * decay[n]);
    const float *ctr = lds + (ly + R) * tw + lx + R;
    float nb = 0.f, ng = 0.f, nr = 0.f, den = 0.f;
    for (int sy = -rs; sy <= rs; ++sy)
        for (int sx = -rs; sx <= rs; ++sx) {
            float d2 = 0.f;
            for (int oy = -rb; oy <= rb; ++oy)
                for (int ox = -rb; ox <= rb; ++ox) {
                    const float *p = ctr + oy * tw + ox, *q = p + sy * tw + sx;
                    const float d0 = q[0] - p[0], d1 = q[per] - p[per], d2c = q[2 * per] - p[2 * per];
                    d2 += d0 * d0 + d1 * d1 + d2c * d2c;
                }
            const float wgt = __expf(d2 * scale);
            const float *q = ctr + sy * tw + sx;
            nb += wgt * q[0]; ng += wgt * q[per]; nr += wgt * q[2 * per]; den += wgt;
        }
    const size_t plane = (size_t)H * W, o = (size_t)n * 3 * plane + (size_t)py * W + px;
    const float rden = 1.f / den;                       // OPSPEC: normalise by one reciprocal, not three divisions
    y[o] = emit(nb * rden, so);
    y[o + plane] = emit(ng * rden, so);
    y[o + 2 * plane] = emit(nr * rden, so);
}

// ---------------------------------------------------------------- global tone curves / white-world (plane streams)
enum { TM_REINHARD = 0, TM_CRYSIS = 1, TM_FILMIC = 2, TM_GAIN = 3 };

__device__ __forceinline__ float hable(float t) {
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    return (t * (A * t + C * B) + D * E) / (t * (A * t + B) + D * F) - E / F;
}

// p: (N,4) per-image constants prepared by tonemap_prepare_kernel
template <int MODE>
__global__ __launch_bounds__(256) void tonemap_kernel(const float *__restrict__ x, const float *__restrict__ p,
                                                      float *__restrict__ y, int hw4, float si, float so) {
    const int n = blockIdx.y;
    const float p0 = p[n * 4], p1 = p[n * 4 + 1], p2 = p[n * 4 + 2];
    const float4 *xb = reinterpret_cast<const float4 *>(x) + (size_t)n * 3 * hw4;
    float4 *yb = reinterpret_cast<float4 *>(y) + (size_t)n * 3 * hw4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw4; i += gridDim.x * blockDim.x) {
        float4 v[3] = {xb[i], xb[hw4 + i], xb[2 * hw4 + i]};
        float *e = reinterpret_cast<float *>(v);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float b = e[k] * si / 255.f, g = e[4 + k] * si / 255.f, r = e[8 + k] * si / 255.f;
            if (MODE != TM_GAIN) {               // OPSPEC: the tone curves take max(sample, 0) - an upstream stage (a CNN
                b = fmaxf(b, 0.f);               // proxy, an unclipped gain) can hand over negative samples, for which
                g = fmaxf(g, 0.f);               // the log-average luminance and Hable's rational curve are undefined
                r = fmaxf(r, 0.f);
            }
            if (MODE == TM_REINHARD) {           // p0 = key / log-average luminance, p1 = 1 / Lwhite^2
                const float L = 0.114f * b + 0.587f * g + 0.299f * r;
                const float ls = p0 * L;
                const float s = ls * (1.f + ls * p1) / (1.f + ls) / fmaxf(L, 1e-6f);
                b *= s; g *= s; r *= s;
            } else if (MODE == TM_CRYSIS) {      // p0 = 0.5 / (lum_adapted + 0.05)
                b = 1.f - __expf(-b * p0); g = 1.f - __expf(-g * p0); r = 1.f - __expf(-r * p0);
            } else if (MODE == TM_FILMIC) {      // p0 = exposure bias, p1 = 1 / hable(W)
                b = hable(b * p0) * p1; g = hable(g * p0) * p1; r = hable(r * p0) * p1;
            } else {                             // per-channel gains p0,p1,p2 on the 0..255 values
                b *= p0; g *= p1; r *= p2;
            }
            e[k] = emit(b * 255.f, so); e[4 + k] = emit(g * 255.f, so); e[8 + k] = emit(r * 255.f, so);
        }
        yb[i] = v[0]; yb[hw4 + i] = v[1]; yb[2 * hw4 + i] = v[2];
    }
}

// per-image sum of log(L + 1e-4) of the 0..255 image (Reinhard's log-average luminance)
__global__ __launch_bounds__(256) void loglum_kernel(const float *__restrict__ x, float *__restrict__ out, int hw4,
                                                     float si) {
    __shared__ float red[4];
    const int n = blockIdx.y;
    const float4 *xb = reinterpret_cast<const float4 *>(x) + (size_t)n * 3 * hw4;
    float acc[1] = {0.f};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw4; i += gridDim.x * blockDim.x) {
        const float4 b = xb[i], g = xb[hw4 + i], r = xb[2 * hw4 + i];
        auto L = [si](float bb, float gg, float rr) {
            bb = fmaxf(bb * si, 0.f); gg = fmaxf(gg * si, 0.f); rr = fmaxf(rr * si, 0.f);     // as tonemap_kernel
            return __logf((0.114f * bb + 0.587f * gg + 0.299f * rr) / 255.f + 1e-4f);
        };
        acc[0] += (L(b.x, g.x, r.x) + L(b.y, g.y, r.y)) + (L(b.z, g.z, r.z) + L(b.w, g.w, r.w));
    }
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) out[n] = acc[0];               // one workgroup per image: bit-repeatable log-average
}

// raw per-image plugin parameters -> the constants tonemap_kernel consumes
__global__ void tonemap_prepare_kernel(int mode, const float *__restrict__ a, const float *__restrict__ b,
                                       const float *__restrict__ stats, float *__restrict__ p, int N, float inv_hw,
                                       float si) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
    if (mode == TM_REINHARD) {               // a = white_point, b = middle_grey, stats = sum log-lum
        const float lw = fmaxf(a[n], 0.01f) * 10.f;
        p0 = fmaxf(b[n], 0.01f) / __expf(stats[n] * inv_hw);
        p1 = 1.f / (lw * lw);
    } else if (mode == TM_CRYSIS) {          // a = lum_adapted
        p0 = 0.5f / (a[n] + 0.05f);
    } else if (mode == TM_FILMIC) {          // a = white_point, b = exposure_bias
        p0 = b[n];
        p1 = 1.f / hable(fmaxf(a[n], 0.01f) * 11.2f);
    } else {                                 // white-world: a = ratio, stats = channel stats (N,3,4) {min,sum,max,0}
        float mx[3], big = 0.f;
        for (int c = 0; c < 3; ++c) { mx[c] = fmaxf(stats[(n * 3 + c) * 4 + 2] * si, 1e-3f); big = fmaxf(big, mx[c]); }
        p0 = 1.f + a[n] * (big / mx[0] - 1.f);
        p1 = 1.f + a[n] * (big / mx[1] - 1.f);
        p2 = 1.f + a[n] * (big / mx[2] - 1.f);
    }
    p[n * 4] = p0; p[n * 4 + 1] = p1; p[n * 4 + 2] = p2; p[n * 4 + 3] = 0.f;
}

dim3 tile_grid(int N, int H, int W) { return dim3((W + TX - 1) / TX, (H + TY - 1) / TY, N); }
size_t tile_lds(int C, int R) { return sizeof(float) * C * (TX + 2 * R) * (TY + 2 * R); }

}  // namespace

extern "C" {

int risp_origin_demosaic(const float *bayer, float *bgr, int laplacian, int N, int H, int W, float in_scale,
                         float out_div, void *stream) {
    RISP_CHECK_ARG(bayer && bgr && N > 0 && N <= 65535 && H >= 4 && W >= 4 && H % 2 == 0 && W % 2 == 0,
                   "risp_origin_demosaic: bad arguments (N=%d H=%d W=%d)", N, H, W);
    if (laplacian)
        hipLaunchKernelGGL(origin_demosaic_kernel<true>, tile_grid(N, H, W), dim3(256), tile_lds(1, 2), (hipStream_t)stream,
                           bayer, bgr, H, W, in_scale, out_div);
    else
        hipLaunchKernelGGL(origin_demosaic_kernel<false>, tile_grid(N, H, W), dim3(256), tile_lds(1, 2), (hipStream_t)stream,
                           bayer, bgr, H, W, in_scale, out_div);
    RISP_LAUNCH_CHECK("risp_origin_demosaic");
    return 0;
}

int risp_origin_bilateral(const float *x, float *y, const int32_t *window, const float *sigma_color,
                          const float *sigma_space, int max_window, int N, int H, int W, float in_scale,
                          float out_div, void *stream) {
    RISP_CHECK_ARG(x && y && window && sigma_color && sigma_space && N > 0 && N <= 65535 && max_window >= 1 &&
                       max_window <= 17 && (max_window & 1) && H > max_window / 2 && W > max_window / 2,
                   "risp_origin_bilateral: bad arguments (window %d, H=%d W=%d)", max_window, H, W);
    const int R = max_window / 2;
    hipLaunchKernelGGL(bilateral_kernel, tile_grid(N, H, W), dim3(256), tile_lds(3, R), (hipStream_t)stream, x, y, window,
                       sigma_color, sigma_space, H, W, R, in_scale, out_div);
    RISP_LAUNCH_CHECK("risp_origin_bilateral");
    return 0;
}

int risp_origin_median(const float *x, float *y, int size, int N, int H, int W, float in_scale, float out_div,
                       void *stream) {
    RISP_CHECK_ARG(x && y && N > 0 && N <= 65535 && size >= 1 && size <= 17 && (size & 1) && H > size / 2 && W > size / 2,
                   "risp_origin_median: bad arguments (size %d, H=%d W=%d)", size, H, W);
    hipLaunchKernelGGL(median_kernel, tile_grid(N, H, W), dim3(256), tile_lds(3, size / 2), (hipStream_t)stream, x, y, H, W,
                       size / 2, in_scale, out_div);
    RISP_LAUNCH_CHECK("risp_origin_median");
    return 0;
}

int risp_origin_fastnlm(const float *x, float *y, const int32_t *block_size, const int32_t *search_block,
                        const float *decay, int max_block, int max_search, int N, int H, int W, float in_scale,
                        float out_div, void *stream) {
    const int R = max_block / 2 + max_search / 2;
    RISP_CHECK_ARG(x && y && block_size && search_block && decay && N > 0 && N <= 65535 && max_block >= 1 &&
                       max_block <= 17 && max_search >= 1 && max_search <= 17 && H > R && W > R,
                   "risp_origin_fastnlm: bad arguments (block %d search %d, H=%d W=%d)", max_block, max_search, H, W);
    RISP_CHECK_ARG(tile_lds(3, R) <= 64 * 1024, "risp_origin_fastnlm: window too large for the LDS tile");
    hipLaunchKernelGGL(fastnlm_kernel, tile_grid(N, H, W), dim3(256), tile_lds(3, R), (hipStream_t)stream, x, y, block_size,
                       search_block, decay, H, W, R, in_scale, out_div);
    RISP_LAUNCH_CHECK("risp_origin_fastnlm");
    return 0;
}

int risp_origin_tonemap(const float *x, float *y, int mode, const float *a, const float *b, const float *stats,
                        float *scratch, int N, int HW, float in_scale, float out_div, void *stream) {
    RISP_CHECK_ARG(x && y && a && scratch && N > 0 && N <= 65535 && HW > 0 && HW % 4 == 0 && mode >= TM_REINHARD &&
                       mode <= TM_GAIN,
                   "risp_origin_tonemap: bad arguments (mode %d)", mode);
    RISP_CHECK_ARG((mode != TM_REINHARD && mode != TM_FILMIC) || b, "risp_origin_tonemap: second parameter missing");
    RISP_CHECK_ARG(mode != TM_GAIN || stats, "risp_origin_tonemap: white-world needs the channel statistics");
    hipStream_t s = (hipStream_t)stream;
    const int hw4 = HW / 4;
    int bx = (hw4 + 255) / 256;
    if (bx > 64) bx = 64;
    float *p = scratch, *lsum = scratch + 4 * N;      // scratch: 5*N floats
    const float *st = stats;
    if (mode == TM_REINHARD) {
        if (hipMemsetAsync(lsum, 0, sizeof(float) * N, s) != hipSuccess) {
            risp_set_error("risp_origin_tonemap: memset failed");
            return 2;
        }
        hipLaunchKernelGGL(loglum_kernel, dim3(1, N), dim3(256), 0, s, x, lsum, hw4, in_scale);
        st = lsum;
    }
    hipLaunchKernelGGL(tonemap_prepare_kernel, dim3((N + 63) / 64), dim3(64), 0, s, mode, a, b, st, p, N, 1.0f / (float)HW,
                       in_scale);
    if (mode == TM_REINHARD) hipLaunchKernelGGL(tonemap_kernel<TM_REINHARD>, dim3(bx, N), dim3(256), 0, s, x, p, y, hw4, in_scale,
                                                out_div);
    else if (mode == TM_CRYSIS) hipLaunchKernelGGL(tonemap_kernel<TM_CRYSIS>, dim3(bx, N), dim3(256), 0, s, x, p, y, hw4, in_scale,
                                                out_div);
    else if (mode == TM_FILMIC) hipLaunchKernelGGL(tonemap_kernel<TM_FILMIC>, dim3(bx, N), dim3(256), 0, s, x, p, y, hw4, in_scale,
                                                out_div);
    else hipLaunchKernelGGL(tonemap_kernel<TM_GAIN>, dim3(bx, N), dim3(256), 0, s, x, p, y, hw4, in_scale,
                                                out_div);
    RISP_LAUNCH_CHECK("risp_origin_tonemap");
    return 0;
}

}  // extern "C"
