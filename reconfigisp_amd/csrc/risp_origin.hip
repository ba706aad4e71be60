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
    extern __shared__ __attribute__((aligned(16))) float lds[];
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
    extern __shared__ __attribute__((aligned(16))) float lds[];
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

// ---------------------------------------------------------------- 64 x 16 tiles, 4 adjacent pixels per thread
// The 32 x 8 one-pixel-per-thread kernels above and below stay as the general form (any W, any window).  For
// W % 4 == 0 the denoisers run on 64 x 16 tiles whose threads own 4 horizontally adjacent pixels: every LDS value a
// thread reads serves up to 4 windows, and the three output planes leave as 16-byte stores.  Per pixel the
// arithmetic is the SAME expression sequence as in the general kernels - the two forms give identical bits.
constexpr int QX = 64, QY = 16;
__host__ __device__ constexpr int tile4_pad(int R) { return (R + 3) & ~3; }             // halo columns kept left and right
__host__ __device__ constexpr int tile4_tw(int R) { return QX + 2 * tile4_pad(R); }

// planes: C contiguous HxW planes of one image -> lds[C][QY+2R][tile4_tw(R)], image column X at tile column
// X - x0 + tile4_pad(R): the 64 interior columns are 16-byte aligned in global memory and in LDS (one 16-byte load and one
// 16-byte LDS write per 4 values, row index arithmetic once per vector), only the 2R halo columns are scalar.  W % 4 == 0.
template <bool QUANT>
__device__ __forceinline__ void stage_tile4(const float *__restrict__ planes, float *lds, int C, int H, int W, int x0,
                                            int y0, int R, float in_scale) {
    const int RP = tile4_pad(R), tw = tile4_tw(R), th = QY + 2 * R;
    auto conv = [&](float v) { v *= in_scale; return QUANT ? q8(v) : v; };
#pragma unroll 4
    for (int idx = threadIdx.x; idx < C * th * (QX / 4); idx += 256) {
        const int rowi = idx / (QX / 4), v = idx - rowi * (QX / 4);
        const int c = rowi / th, ty = rowi - c * th;
        const int gy = reflect101(y0 + ty - R, H), gx = x0 + 4 * v;
        const float *src = planes + ((size_t)c * H + gy) * W;
        // the 16-byte load is issued unconditionally (from column 0 where the vector lies beyond the image): with a branch around it
        // hipcc waits for every load before it issues the next, and a tile is 4 .. 6 of them per thread - one memory round trip each
        const bool inside = gx < W;                                       // W % 4 == 0: entirely inside, or entirely outside
        float4 q = *reinterpret_cast<const float4 *>(src + (inside ? gx : 0));
        if (!inside) q = make_float4(src[reflect101(gx, W)], src[reflect101(gx + 1, W)], src[reflect101(gx + 2, W)], src[reflect101(gx + 3, W)]);
        *reinterpret_cast<float4 *>(lds + rowi * tw + RP + 4 * v) = make_float4(conv(q.x), conv(q.y), conv(q.z), conv(q.w));
    }
    for (int idx = threadIdx.x; idx < C * th * 2 * R; idx += 256) {
        const int rowi = idx / (2 * R), j = idx - rowi * (2 * R);
        const int c = rowi / th, ty = rowi - c * th;
        const int off = j < R ? j - R : QX + j - R;                          // tile-relative image column
        const int gy = reflect101(y0 + ty - R, H), gx = reflect101(x0 + off, W);
        lds[rowi * tw + RP + off] = conv(planes[((size_t)c * H + gy) * W + gx]);
    }
    __syncthreads();
}

typedef float of4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store3x4(float *y, size_t o, size_t plane, const float (&b)[4], const float (&g)[4],
                                         const float (&r)[4]) {
    *reinterpret_cast<of4 *>(y + o) = (of4){b[0], b[1], b[2], b[3]};
    *reinterpret_cast<of4 *>(y + o + plane) = (of4){g[0], g[1], g[2], g[3]};
    *reinterpret_cast<of4 *>(y + o + 2 * plane) = (of4){r[0], r[1], r[2], r[3]};
}

// bilinear / Malvar-He-Cutler demosaic on the 64 x 16 tile: per pixel the expressions of origin_demosaic_kernel, the
// column parity of each of the thread's 4 pixels known at compile time, three 16-byte stores.
template <bool LAPLACIAN>
__global__ __launch_bounds__(256) void demosaic4_kernel(const float *__restrict__ x, float *__restrict__ y, int H, int W,
                                                        float si, float so) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int R = 2, RP = tile4_pad(R), tw = tile4_tw(R);
    const int n = blockIdx.z, x0 = blockIdx.x * QX, y0 = blockIdx.y * QY;
    stage_tile4<false>(x + (size_t)n * H * W, lds, 1, H, W, x0, y0, R, si);
    const int lx = (threadIdx.x & 15) * 4, ly = threadIdx.x >> 4;
    const int px = x0 + lx, py = y0 + ly;
    if (px >= W || py >= H) return;
    const bool er = (py & 1) == 0;
    float ob[4], og[4], orr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        auto s = [&](int dy, int dx) { return lds[(ly + R + dy) * tw + lx + RP + i + dx]; };
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
        const bool ec = (i & 1) == 0;                  // px is a multiple of 4
        float R_, G_, B_;
        if (er && ec) { R_ = c; G_ = g_rb; B_ = rb_diag; }
        else if (er && !ec) { G_ = c; R_ = rb_hor; B_ = rb_ver; }
        else if (!er && ec) { G_ = c; R_ = rb_ver; B_ = rb_hor; }
        else { B_ = c; G_ = g_rb; R_ = rb_diag; }
        ob[i] = emit(B_, so); og[i] = emit(G_, so); orr[i] = emit(R_, so);
    }
    const size_t plane = (size_t)H * W;
    store3x4(y, (size_t)n * 3 * plane + (size_t)py * W + px, plane, ob, og, orr);
}

// RT > 0: images whose radius equals RT take a fully unrolled window (3x3 is what every reference configuration
// produces, tools_origin.py:698); any other radius <= R runs the loops.
template <int RT>
__global__ __launch_bounds__(256) void bilateral4_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                         const int *__restrict__ win, const float *__restrict__ sig_c,
                                                         const float *__restrict__ sig_s, int H, int W, int R,
                                                         float si, float so) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // (the launches of the unrolled forms pass R == RT: as a compile-time constant the tile's pitch and plane size fold into the
    // offset fields of the LDS reads - one vector add per read less)
    if (RT > 0) R = RT;
    const int tw = tile4_tw(R), per = tw * (QY + 2 * R), RP = tile4_pad(R);
    const int n = blockIdx.z, x0 = blockIdx.x * QX, y0 = blockIdx.y * QY;
    stage_tile4<false>(x + (size_t)n * 3 * H * W, lds, 3, H, W, x0, y0, R, si);
    const int lx = (threadIdx.x & 15) * 4, ly = threadIdx.x >> 4;
    const int px = x0 + lx, py = y0 + ly;
    if (px >= W || py >= H) return;                    // W % 4 == 0: the 4 pixels are in or out together
    int r = win[n] / 2;
    r = r < 0 ? 0 : (r > R ? R : r);
    const float ks = -1.f / (2.f * sig_s[n] * sig_s[n]), kc = -1.f / (2.f * sig_c[n] * sig_c[n]);
    const float ks2 = ks * 1.4426950408889634f, kc2 = kc * 1.4426950408889634f;
    const bool full = RT > 0 && r == RT;
    float ob[4], og[4], orr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float *ctr = lds + (ly + R) * tw + lx + RP + i;
        const float cb = ctr[0], cg = ctr[per], cr = ctr[2 * per];
        float nb = 0.f, ng = 0.f, nr = 0.f, den = 0.f;
        auto tap = [&](int dy, int dx) {
            const float *q = ctr + dy * tw + dx;
            const float qb = q[0], qg = q[per], qr = q[2 * per];
            if (dy == 0 && dx == 0) {
                nb += qb; ng += qg; nr += qr; den += 1.f;
                return;
            }
            const float dist = fabsf(qb - cb) + fabsf(qg - cg) + fabsf(qr - cr);
            const float wgt = __builtin_amdgcn_exp2f(__builtin_fmaf(dist * dist, kc2, (float)(dy * dy + dx * dx) * ks2));
            nb = __builtin_fmaf(wgt, qb, nb); ng = __builtin_fmaf(wgt, qg, ng); nr = __builtin_fmaf(wgt, qr, nr); den += wgt;
        };
        if (full) {
#pragma unroll
            for (int dy = -RT; dy <= RT; ++dy)
#pragma unroll
                for (int dx = -RT; dx <= RT; ++dx) tap(dy, dx);
        } else {
            for (int dy = -r; dy <= r; ++dy)
                for (int dx = -r; dx <= r; ++dx) tap(dy, dx);
        }
        const float rden = 1.f / den;
        ob[i] = emit(nb * rden, so); og[i] = emit(ng * rden, so); orr[i] = emit(nr * rden, so);
    }
    const size_t plane = (size_t)H * W;
    store3x4(y, (size_t)n * 3 * plane + (size_t)py * W + px, plane, ob, og, orr);
}

// 3x3 median of 8-bit codes, 4 pixels per thread: the six window columns a thread touches are sorted once
// (v_min3 / v_med3 / v_max3) and each pixel's median is med3(max of the column minima, med of the column medians,
// min of the column maxima) - the exact middle element, 25 instructions per pixel against ~650 for the bisection.
__global__ __launch_bounds__(256) void median3x4_kernel(const float *__restrict__ x, float *__restrict__ y, int H, int W,
                                                        float si, float so) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int R = 1, tw = tile4_tw(R), per = tw * (QY + 2 * R), RP = tile4_pad(R);
    const int n = blockIdx.z, x0 = blockIdx.x * QX, y0 = blockIdx.y * QY;
    stage_tile4<true>(x + (size_t)n * 3 * H * W, lds, 3, H, W, x0, y0, R, si);
    const int lx = (threadIdx.x & 15) * 4, ly = threadIdx.x >> 4;
    const int px = x0 + lx, py = y0 + ly;
    if (px >= W || py >= H) return;
    float o[3][4];
    const float inv = 1.f / fabsf(so);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float *top = lds + c * per + ly * tw + lx + RP - R;   // window rows ly .. ly+2, image columns px-1 .. px+4
        float lo[6], md[6], hi[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float a = top[j], b = top[tw + j], d = top[2 * tw + j];
            lo[j] = fminf(fminf(a, b), d);
            hi[j] = fmaxf(fmaxf(a, b), d);
            md[j] = __builtin_amdgcn_fmed3f(a, b, d);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a = fmaxf(fmaxf(lo[i], lo[i + 1]), lo[i + 2]);
            const float b = __builtin_amdgcn_fmed3f(md[i], md[i + 1], md[i + 2]);
            const float d = fminf(fminf(hi[i], hi[i + 1]), hi[i + 2]);
            o[c][i] = __builtin_amdgcn_fmed3f(a, b, d) * inv;
        }
    }
    const size_t plane = (size_t)H * W;
    store3x4(y, (size_t)n * 3 * plane + (size_t)py * W + px, plane, o[0], o[1], o[2]);
}

// ---------------------------------------------------------------- median on 8-bit codes (bisection on the code)
__global__ __launch_bounds__(256) void median_kernel(const float *__restrict__ x, float *__restrict__ y, int H, int W,
                                                     int R, float si, float so) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
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
    extern __shared__ __attribute__((aligned(16))) float lds[];
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

// K x K median (K = 5 .. 17; 9 is the reference's default, tools_origin.py:746 with p = 0.5) of 8-bit codes, 4 pixels per
// thread, on BYTES.  The tile is staged as packed codes (four pixels per LDS dword); a thread pulls the 12 (K <= 9) or
// 20 bytes per window row that its four windows share, cuts a pixel's row out with v_alignbyte and keeps that pixel's
// window in registers (K*K/4 dwords) through its bisection.  Rank counting uses the sum of absolute differences: with S(m) = sum |x_i - m| over a set of n
// codes, S(m+1) - S(m) = #(x <= m) - #(x > m), so #(x <= m) = (S(m+1) - S(m) + n) / 2 - and v_sad_u8 adds four
// |x - m| per instruction.  A bisection round costs 2 SADs per 4 codes (plus a compare for the last K*K % 4 codes)
// instead of a compare and an add per code and an LDS read per code: the same exact median (smallest code whose rank
// reaches the middle), about 4-5 x faster than the general kernel on 9 x 9.
template <int K>
__global__ __launch_bounds__(256) void median4_kernel(const float *__restrict__ x, float *__restrict__ y, int H, int W,
                                                      float si, float so) {
    constexpr int R = K / 2, RP = R <= 4 ? 4 : 8, TWB = QX + 2 * RP, TWD = TWB / 4, TH = QY + 2 * R;   // tile row: 72 / 80 bytes
    constexpr int NDW = (2 * RP + 4) / 4;              // dwords of a tile row that cover the thread's four windows
    constexpr int ND = K / 4, NE = K % 4;              // full dwords and leftover codes per window row
    constexpr int NL = K * NE, NLD = NL / 4, NLS = NL % 4;   // the leftovers of all rows, packed four to a dword again
    constexpr int NW = K * ND + NLD;
    static_assert(R <= RP && K <= 17, "window rows must fit the bytes a thread loads");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned *tile = reinterpret_cast<unsigned *>(lds);                  // [3][TH][TWD] packed codes
    const int n = blockIdx.z, x0 = blockIdx.x * QX, y0 = blockIdx.y * QY;
    const float *planes = x + (size_t)n * 3 * H * W;
    for (int idx = threadIdx.x; idx < 3 * TH * TWD; idx += 256) {
        const int rowi = idx / TWD, dj = idx - rowi * TWD;
        const int c = rowi / TH, ty = rowi - c * TH;
        const int gy = reflect101(y0 + ty - R, H), gx = x0 - RP + 4 * dj;
        const float *src = planes + ((size_t)c * H + gy) * W;
        float4 q;
        if (gx >= 0 && gx < W) q = *reinterpret_cast<const float4 *>(src + gx);       // W % 4 == 0: entirely inside
        else q = make_float4(src[reflect101(gx, W)], src[reflect101(gx + 1, W)], src[reflect101(gx + 2, W)], src[reflect101(gx + 3, W)]);
        tile[idx] = (unsigned)q8(q.x * si) | ((unsigned)q8(q.y * si) << 8) | ((unsigned)q8(q.z * si) << 16) | ((unsigned)q8(q.w * si) << 24);
    }
    __syncthreads();
    const int lxd = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int px = x0 + 4 * lxd, py = y0 + ly;
    if (px >= W || py >= H) return;
    const float inv = 1.f / fabsf(so);
    const size_t plane = (size_t)H * W, o = (size_t)n * 3 * plane + (size_t)py * W + px;
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        float res[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned wd[NW > 0 ? NW : 1];                      // the window as packed dwords: row pieces, then leftovers
            int ws[NLS > 0 ? NLS : 1], left[NL > 0 ? NL : 1];  // codes that fill no dword
            const int off = RP - R + i;                        // first byte of pixel i's window row
#pragma unroll
            for (int r = 0; r < K; ++r) {
                const unsigned *row = tile + (c * TH + ly + r) * TWD + lxd;     // bytes: image columns px-RP .. px+RP+3
                unsigned d[NDW];
#pragma unroll
                for (int j = 0; j < NDW; ++j) d[j] = row[j];
#pragma unroll
                for (int j = 0; j < ND; ++j) {
                    const int b = off + 4 * j;
                    wd[r * ND + j] = (b & 3) ? __builtin_amdgcn_alignbyte(d[(b >> 2) + 1 < NDW ? (b >> 2) + 1 : NDW - 1], d[b >> 2], b & 3)
                                             : d[b >> 2];
                }
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const int b = off + 4 * ND + e;
                    left[r * NE + e] = (int)((d[b >> 2] >> (8 * (b & 3))) & 0xffu);
                }
            }
#pragma unroll
            for (int g = 0; g < NLD; ++g)
                wd[K * ND + g] = (unsigned)left[4 * g] | ((unsigned)left[4 * g + 1] << 8) | ((unsigned)left[4 * g + 2] << 16) |
                                 ((unsigned)left[4 * g + 3] << 24);
#pragma unroll
            for (int e = 0; e < NLS; ++e) ws[e] = left[4 * NLD + e];
            int lo = 0, hi = 255;                      // smallest code v with #(x <= v) >= need
            constexpr int need = (K * K) / 2 + 1, n8 = 4 * NW;
#pragma unroll 1
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;        // <= 254
                const unsigned m4 = (unsigned)mid * 0x01010101u, m4p = m4 + 0x01010101u;
                unsigned sm = 0, sp = 0;
                int cnt = 0;
#pragma unroll
                for (int j = 0; j < NW; ++j) {
                    sm = __builtin_amdgcn_sad_u8(wd[j], m4, sm);
                    sp = __builtin_amdgcn_sad_u8(wd[j], m4p, sp);
                }
#pragma unroll
                for (int e = 0; e < NLS; ++e) cnt += ws[e] <= mid ? 1 : 0;
                cnt += ((int)sp - (int)sm + n8) >> 1;
                if (cnt >= need) hi = mid; else lo = mid + 1;
            }
            res[i] = (float)lo * inv;
        }
        *reinterpret_cast<of4 *>(y + o + c * plane) = (of4){res[0], res[1], res[2], res[3]};
    }
}

// Non-local means on the 64 x 16 tile, 4 pixels per thread.  FAST (the staged halo is exactly 2: block 3, search 3 - what
// every reference configuration produces, tools_origin.py:786-787): an image with those sizes keeps its 5 x 8 x 3
// neighbourhood in registers; the squared colour difference e_s(q) = |x[q+s] - x[q]|^2 of a search offset s is evaluated
// once per position q of the thread's 3 x 6 patch area and serves every patch that covers q (the general kernel
// evaluates it once per patch: 9 times), the patch distances are 3 x 3 sums of e_s in the general kernel's order.
template <bool FAST>
__global__ __launch_bounds__(256) void fastnlm4_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                       const int *__restrict__ blk, const int *__restrict__ srch,
                                                       const float *__restrict__ decay, int H, int W, int R,
                                                       float si, float so) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (FAST) R = 2;                                   // (launched for R == 2 only: the tile geometry as compile-time constants)
    const int tw = tile4_tw(R), per = tw * (QY + 2 * R), RP = tile4_pad(R);
    const int n = blockIdx.z, x0 = blockIdx.x * QX, y0 = blockIdx.y * QY;
    stage_tile4<false>(x + (size_t)n * 3 * H * W, lds, 3, H, W, x0, y0, R, si);
    const int lx = (threadIdx.x & 15) * 4, ly = threadIdx.x >> 4;
    const int px = x0 + lx, py = y0 + ly;
    if (px >= W || py >= H) return;
    const int b = blk[n];
    int rb = b / 2, rs = srch[n] / 2;
    rb = rb < 0 ? 0 : (rb > R ? R : rb);
    rs = rs < 0 ? 0 : (rs > R - rb ? R - rb : rs);
    const float scale = -1.f / (3.f * (float)(b * b) * decay[n] * decay[n]);
    float ob[4], og[4], orr[4];
    if (FAST && rb == 1 && rs == 1) {
        // v[row][col][ch]: image rows py-2 .. py+2, columns px-2 .. px+5 = tile columns lx+2 .. lx+9 (halo pad 4): an
        // 8-byte, a 16-byte and an 8-byte aligned read per row and plane
        float v[5][8][3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int rr = 0; rr < 5; ++rr) {
                const float *row = lds + c * per + (ly + rr) * tw + lx;
                const float2 a = *reinterpret_cast<const float2 *>(row + 2), e2 = *reinterpret_cast<const float2 *>(row + 8);
                const float4 d = *reinterpret_cast<const float4 *>(row + 4);
                v[rr][0][c] = a.x; v[rr][1][c] = a.y; v[rr][2][c] = d.x; v[rr][3][c] = d.y;
                v[rr][4][c] = d.z; v[rr][5][c] = d.w; v[rr][6][c] = e2.x; v[rr][7][c] = e2.y;
            }
        float nb[4] = {0.f, 0.f, 0.f, 0.f}, ng[4] = {0.f, 0.f, 0.f, 0.f}, nr[4] = {0.f, 0.f, 0.f, 0.f}, den[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sy = -1; sy <= 1; ++sy)
#pragma unroll
            for (int sx = -1; sx <= 1; ++sx) {
                float e[3][6];                         // q = (py + qy, px + qx), qy = -1..1, qx = -1..4
#pragma unroll
                for (int qy = 0; qy < 3; ++qy)
#pragma unroll
                    for (int qx = 0; qx < 6; ++qx) {
                        const float d0 = v[qy + 1 + sy][qx + 1 + sx][0] - v[qy + 1][qx + 1][0];
                        const float d1 = v[qy + 1 + sy][qx + 1 + sx][1] - v[qy + 1][qx + 1][1];
                        const float d2c = v[qy + 1 + sy][qx + 1 + sx][2] - v[qy + 1][qx + 1][2];
                        e[qy][qx] = d0 * d0 + d1 * d1 + d2c * d2c;
                    }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float d2 = 0.f;
#pragma unroll
                    for (int oy = 0; oy < 3; ++oy)
#pragma unroll
                        for (int ox = 0; ox < 3; ++ox) d2 += e[oy][i + ox];
                    const float wgt = __expf(d2 * scale);
                    nb[i] += wgt * v[2 + sy][i + 2 + sx][0];
                    ng[i] += wgt * v[2 + sy][i + 2 + sx][1];
                    nr[i] += wgt * v[2 + sy][i + 2 + sx][2];
                    den[i] += wgt;
                }
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float rden = 1.f / den[i];
            ob[i] = emit(nb[i] * rden, so); og[i] = emit(ng[i] * rden, so); orr[i] = emit(nr[i] * rden, so);
        }
    } else {
        for (int i = 0; i < 4; ++i) {
            const float *ctr = lds + (ly + R) * tw + lx + RP + i;
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
            const float rden = 1.f / den;
            ob[i] = emit(nb * rden, so); og[i] = emit(ng * rden, so); orr[i] = emit(nr * rden, so);
        }
    }
    const size_t plane = (size_t)H * W;
    store3x4(y, (size_t)n * 3 * plane + (size_t)py * W + px, plane, ob, og, orr);
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

// per-image sum of log(L + 1e-4) of the 0..255 image (Reinhard's log-average luminance): LOGLUM_PARTS workgroups per
// image, one partial sum each; tonemap_prepare_kernel adds the partials in index order (bit-repeatable).  (One
// workgroup per image, the first form, streamed 786 KB through 256 threads: 39 us on 64 x 256 x 256.)
constexpr int LOGLUM_PARTS = 8;
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
    if (threadIdx.x == 0) out[n * LOGLUM_PARTS + blockIdx.x] = acc[0];
}

// raw per-image plugin parameters -> the constants tonemap_kernel consumes
__global__ void tonemap_prepare_kernel(int mode, const float *__restrict__ a, const float *__restrict__ b,
                                       const float *__restrict__ stats, float *__restrict__ p, int N, float inv_hw,
                                       float si) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
    if (mode == TM_REINHARD) {               // a = white_point, b = middle_grey, stats = partial sums of log-lum
        const float lw = fmaxf(a[n], 0.01f) * 10.f;
        float lsum = 0.f;
        for (int k = 0; k < LOGLUM_PARTS; ++k) lsum += stats[n * LOGLUM_PARTS + k];
        p0 = fmaxf(b[n], 0.01f) / __expf(lsum * inv_hw);
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
dim3 tile4_grid(int N, int H, int W) { return dim3((W + QX - 1) / QX, (H + QY - 1) / QY, N); }
size_t tile4_lds(int C, int R) { return sizeof(float) * C * tile4_tw(R) * (QY + 2 * R); }
// the 4-pixels-per-thread form: 16-byte stores (rows and the output base 16-byte aligned) and an LDS tile within the 64 KB default
// the 4-pixel kernels read the input and write the output in 16-byte vectors: BOTH pointers must be aligned (a contiguous
// view with a storage offset that is not a multiple of 4 floats takes the general kernels)
bool tile4_ok(const float *x, const float *y, int W, int R) {
    return W % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0 && tile4_lds(3, R) <= 64 * 1024;
}

}  // namespace

extern "C" {

int risp_origin_demosaic(const float *bayer, float *bgr, int laplacian, int N, int H, int W, float in_scale,
                         float out_div, void *stream) {
    RISP_CHECK_ARG(bayer && bgr && N > 0 && N <= 65535 && H >= 4 && W >= 4 && H % 2 == 0 && W % 2 == 0,
                   "risp_origin_demosaic: bad arguments (N=%d H=%d W=%d)", N, H, W);
    if (W % 4 == 0 && ((reinterpret_cast<uintptr_t>(bgr) | reinterpret_cast<uintptr_t>(bayer)) & 15) == 0) {
        if (laplacian)
            hipLaunchKernelGGL(demosaic4_kernel<true>, tile4_grid(N, H, W), dim3(256), tile4_lds(1, 2), (hipStream_t)stream, bayer, bgr, H,
                               W, in_scale, out_div);
        else
            hipLaunchKernelGGL(demosaic4_kernel<false>, tile4_grid(N, H, W), dim3(256), tile4_lds(1, 2), (hipStream_t)stream, bayer, bgr, H,
                               W, in_scale, out_div);
    } else if (laplacian)
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
    if (tile4_ok(x, y, W, R)) {
        if (R == 1)
            hipLaunchKernelGGL(bilateral4_kernel<1>, tile4_grid(N, H, W), dim3(256), tile4_lds(3, R), (hipStream_t)stream, x, y,
                               window, sigma_color, sigma_space, H, W, R, in_scale, out_div);
        else if (R == 2)
            hipLaunchKernelGGL(bilateral4_kernel<2>, tile4_grid(N, H, W), dim3(256), tile4_lds(3, R), (hipStream_t)stream, x, y,
                               window, sigma_color, sigma_space, H, W, R, in_scale, out_div);
        else
            hipLaunchKernelGGL(bilateral4_kernel<0>, tile4_grid(N, H, W), dim3(256), tile4_lds(3, R), (hipStream_t)stream, x, y,
                               window, sigma_color, sigma_space, H, W, R, in_scale, out_div);
    } else
        hipLaunchKernelGGL(bilateral_kernel, tile_grid(N, H, W), dim3(256), tile_lds(3, R), (hipStream_t)stream, x, y, window,
                           sigma_color, sigma_space, H, W, R, in_scale, out_div);
    RISP_LAUNCH_CHECK("risp_origin_bilateral");
    return 0;
}

int risp_origin_median(const float *x, float *y, int size, int N, int H, int W, float in_scale, float out_div,
                       void *stream) {
    RISP_CHECK_ARG(x && y && N > 0 && N <= 65535 && size >= 1 && size <= 17 && (size & 1) && H > size / 2 && W > size / 2,
                   "risp_origin_median: bad arguments (size %d, H=%d W=%d)", size, H, W);
    auto med_lds = [](int k) { return sizeof(unsigned) * 3 * (QY + 2 * (k / 2)) * ((QX + (k <= 9 ? 8 : 16)) / 4); };
    const bool t4 = W % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
#define RISP_MEDIAN4(KK)                                                                                                      \
    hipLaunchKernelGGL(median4_kernel<KK>, tile4_grid(N, H, W), dim3(256), med_lds(KK), (hipStream_t)stream, x, y, H, W, in_scale, \
                       out_div)
    if (size == 3 && tile4_ok(x, y, W, 1))
        hipLaunchKernelGGL(median3x4_kernel, tile4_grid(N, H, W), dim3(256), tile4_lds(3, 1), (hipStream_t)stream, x, y, H, W,
                           in_scale, out_div);
    else if (size == 5 && t4) RISP_MEDIAN4(5);
    else if (size == 7 && t4) RISP_MEDIAN4(7);
    else if (size == 9 && t4) RISP_MEDIAN4(9);
    else if (size == 11 && t4) RISP_MEDIAN4(11);
    else if (size == 13 && t4) RISP_MEDIAN4(13);
    else if (size == 15 && t4) RISP_MEDIAN4(15);
    else if (size == 17 && t4) RISP_MEDIAN4(17);
#undef RISP_MEDIAN4
    else
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
    if (tile4_ok(x, y, W, R)) {
        if (R == 2)
            hipLaunchKernelGGL(fastnlm4_kernel<true>, tile4_grid(N, H, W), dim3(256), tile4_lds(3, R), (hipStream_t)stream, x, y,
                               block_size, search_block, decay, H, W, R, in_scale, out_div);
        else
            hipLaunchKernelGGL(fastnlm4_kernel<false>, tile4_grid(N, H, W), dim3(256), tile4_lds(3, R), (hipStream_t)stream, x, y,
                               block_size, search_block, decay, H, W, R, in_scale, out_div);
    } else
        hipLaunchKernelGGL(fastnlm_kernel, tile_grid(N, H, W), dim3(256), tile_lds(3, R), (hipStream_t)stream, x, y, block_size,
                           search_block, decay, H, W, R, in_scale, out_div);
    RISP_LAUNCH_CHECK("risp_origin_fastnlm");
    return 0;
}

size_t risp_origin_tonemap_scratch_floats(int N) { return (size_t)(4 + LOGLUM_PARTS) * (N > 0 ? N : 0); }

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
    float *p = scratch, *lsum = scratch + 4 * N;      // scratch: (4 + LOGLUM_PARTS) * N floats
    const float *st = stats;
    if (mode == TM_REINHARD) {                        // every partial is written (an empty share writes 0)
        hipLaunchKernelGGL(loglum_kernel, dim3(LOGLUM_PARTS, N), dim3(256), 0, s, x, lsum, hw4, in_scale);
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
