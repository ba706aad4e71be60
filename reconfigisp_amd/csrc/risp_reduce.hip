// Per-image reductions and the ops built on them (gfx950): channel statistics
// (SRCNNRes pre-pass, gray-world), the mixed-op combiner, plane sums, uint8 SSE.
//
// Pattern: float4 coalesced loads -> per-thread partials -> wave shuffles -> LDS ->
// one partial record per block in caller-provided scratch -> a one-wave finishing
// kernel.  No atomics on the statistics, so results are run-to-run deterministic.
#include "risp_common.h"

namespace {

constexpr int kStatBlocks = 64;  // max blocks per plane (= one wave in the finishing kernel)

struct MinMax {
    float mn, mx, sum;
    int imn, imx;
};

__device__ __forceinline__ void mm_take(MinMax &a, float v, int i) {
    if (v < a.mn || (v == a.mn && i < a.imn)) { a.mn = v; a.imn = i; }
    if (v > a.mx || (v == a.mx && i < a.imx)) { a.mx = v; a.imx = i; }
}
__device__ __forceinline__ void mm_merge(MinMax &a, float mn, int imn, float mx, int imx, float sum) {
    if (mn < a.mn || (mn == a.mn && imn < a.imn)) { a.mn = mn; a.imn = imn; }
    if (mx > a.mx || (mx == a.mx && imx < a.imx)) { a.mx = mx; a.imx = imx; }
    a.sum += sum;
}
__device__ __forceinline__ void mm_wave(MinMax &a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float mn = __shfl_down(a.mn, o, 64), mx = __shfl_down(a.mx, o, 64), sm = __shfl_down(a.sum, o, 64);
        int imn = __shfl_down(a.imn, o, 64), imx = __shfl_down(a.imx, o, 64);
        mm_merge(a, mn, imn, mx, imx, sm);
    }
}

// partial record: 5 words {min, max, sum, argmin, argmax}
__global__ __launch_bounds__(256) void stats_partial_kernel(const float *__restrict__ x, float *__restrict__ part,
                                                            int hw4, int nblk) {
    __shared__ float s_f[3][4];
    __shared__ int s_i[2][4];
    const int plane = blockIdx.y;
    const float4 *xb = reinterpret_cast<const float4 *>(x) + (size_t)plane * hw4;
    MinMax a = {INFINITY, -INFINITY, 0.f, 0x7fffffff, 0x7fffffff};
    // four vectors of a thread are requested together (one load per trip left 1 KB per wave in flight: 0.44 of the HBM rate at
    // 64 x 3 planes of 256 x 256); same elements in the same order per thread: the same bits
    const int step = gridDim.x * blockDim.x;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    auto take4 = [&](const float4 &v, int j) {
        mm_take(a, v.x, 4 * j);
        mm_take(a, v.y, 4 * j + 1);
        mm_take(a, v.z, 4 * j + 2);
        mm_take(a, v.w, 4 * j + 3);
        a.sum += (v.x + v.y) + (v.z + v.w);
    };
    for (; i + 3 * step < hw4; i += 4 * step) {
        const float4 v0 = xb[i], v1 = xb[i + step], v2 = xb[i + 2 * step], v3 = xb[i + 3 * step];
        take4(v0, i); take4(v1, i + step); take4(v2, i + 2 * step); take4(v3, i + 3 * step);
    }
    for (; i < hw4; i += step) take4(xb[i], i);
    mm_wave(a);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s_f[0][wave] = a.mn; s_f[1][wave] = a.mx; s_f[2][wave] = a.sum;
        s_i[0][wave] = a.imn; s_i[1][wave] = a.imx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        MinMax t = {s_f[0][0], s_f[1][0], s_f[2][0], s_i[0][0], s_i[1][0]};
        for (int w = 1; w < 4; ++w) mm_merge(t, s_f[0][w], s_i[0][w], s_f[1][w], s_i[1][w], s_f[2][w]);
        float *o = part + ((size_t)plane * nblk + blockIdx.x) * 5;
        o[0] = t.mn; o[1] = t.mx; o[2] = t.sum;
        o[3] = __int_as_float(t.imn); o[4] = __int_as_float(t.imx);
    }
}

__global__ __launch_bounds__(64) void stats_final_kernel(const float *__restrict__ part, float *__restrict__ stats,
                                                         int32_t *__restrict__ arg, int nblk) {
    const int plane = blockIdx.x, lane = threadIdx.x;
    MinMax a = {INFINITY, -INFINITY, 0.f, 0x7fffffff, 0x7fffffff};
    if (lane < nblk) {
        const float *p = part + ((size_t)plane * nblk + lane) * 5;
        a = {p[0], p[1], p[2], __float_as_int(p[3]), __float_as_int(p[4])};
    }
    mm_wave(a);
    if (lane == 0) {
        stats[plane * 4 + 0] = a.mn;
        stats[plane * 4 + 1] = a.sum;
        stats[plane * 4 + 2] = a.mx;
        stats[plane * 4 + 3] = 0.f;
        if (arg) {
            arg[plane * 2 + 0] = a.imn;
            arg[plane * 2 + 1] = a.imx;
        }
    }
}

int stat_blocks(int HW) {
    int b = HW / (4 * 256 * 8);
    return b < 1 ? 1 : (b > kStatBlocks ? kStatBlocks : b);
}

// ---- gray-world gains (OPSPEC in oracle/isp_oracle.py::grayworld)
constexpr float kGrayEps = 1e-6f;

__global__ void gray_gains_kernel(const float *__restrict__ stats, float *__restrict__ gains, int N, float inv_hw) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float m[3];
    for (int c = 0; c < 3; ++c) m[c] = stats[(n * 3 + c) * 4 + 1] * inv_hw;
    const float gray = (m[0] + m[1] + m[2]) / 3.f;
    for (int c = 0; c < 3; ++c) gains[n * 3 + c] = gray / fmaxf(m[c], kGrayEps);
}

// gk = dL/d gain (N,3) -> gm = dL/d mean (N,3)
__global__ void gray_gains_bwd_kernel(const float *__restrict__ stats, const float *__restrict__ gk,
                                      float *__restrict__ gm, int N, float inv_hw) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float m[3], mc[3];
    for (int c = 0; c < 3; ++c) {
        m[c] = stats[(n * 3 + c) * 4 + 1] * inv_hw;
        mc[c] = fmaxf(m[c], kGrayEps);
    }
    const float gray = (m[0] + m[1] + m[2]) / 3.f;
    float common = 0.f;
    for (int c = 0; c < 3; ++c) common += gk[n * 3 + c] / mc[c];
    common /= 3.f;
    for (int j = 0; j < 3; ++j) {
        // torch.clamp_min passes the gradient where m >= eps
        float own = (m[j] >= kGrayEps) ? gk[n * 3 + j] * gray / (mc[j] * mc[j]) : 0.f;
        gm[n * 3 + j] = common - own;
    }
}

// gx[plane] += g_mean[plane]/HW everywhere; gx[plane][argmin] += g_min; gx[plane][argmax] += g_max.  The gradient of
// plane n * C + c sits at g_*[n * gstride + c] (packed vectors: C = gstride = 1).
__global__ __launch_bounds__(256) void stats_bwd_kernel(float *__restrict__ gx, const float *__restrict__ g_min,
                                                        const float *__restrict__ g_mean,
                                                        const float *__restrict__ g_max,
                                                        const int32_t *__restrict__ arg, int hw4, float inv_hw, int C,
                                                        int gstride) {
    const int plane = blockIdx.y, gi = (plane / C) * gstride + plane % C;
    float4 *gb = reinterpret_cast<float4 *>(gx) + (size_t)plane * hw4;
    const float add = g_mean ? g_mean[gi] * inv_hw : 0.f;
    const int imn = (g_min && arg) ? arg[plane * 2] : -1, imx = (g_max && arg) ? arg[plane * 2 + 1] : -1;
    const float vmn = g_min ? g_min[gi] : 0.f, vmx = g_max ? g_max[gi] : 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw4; i += gridDim.x * blockDim.x) {
        float4 v = gb[i];
        float *e = reinterpret_cast<float *>(&v);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float t = e[k] + add;
            if (4 * i + k == imn) t += vmn;
            if (4 * i + k == imx) t += vmx;
            e[k] = t;
        }
        gb[i] = v;
    }
}

// ---- SRCNNRes broadcast-plane values (srcnn_res_arch.py:36-43): [min(3) | mean(3) | max(3) | params(P)]
__global__ void srcnn_cvals_kernel(const float *__restrict__ stats, const float *__restrict__ pv,
                                   float *__restrict__ cvals, int N, int P, float inv_hw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, row = 9 + P;
    if (i >= N * row) return;
    const int n = i / row, j = i - n * row;
    float v;
    if (j < 3) v = stats[(n * 3 + j) * 4 + 0];
    else if (j < 6) v = stats[(n * 3 + j - 3) * 4 + 1] * inv_hw;
    else if (j < 9) v = stats[(n * 3 + j - 6) * 4 + 2];
    else v = pv[n * P + (j - 9)];
    cvals[i] = v;
}

// ---- the same values applied to SRCNNRes' folded first layer: table[n][j] = sum_c cval[n][c] * rcase[c][j]
// (convnets.py::SrcnnResFold; j = cout x border case).  One thread per entry, the 9+P values rebuilt per thread from
// the statistics (wave-uniform loads), terms added in index order.
__global__ __launch_bounds__(256) void srcnn_case_table_kernel(const float *__restrict__ stats, const float *__restrict__ pv,
                                                               const float *__restrict__ rcase, float *__restrict__ table,
                                                               int P, int M, float inv_hw) {
    const int n = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    float acc = 0.f;
    for (int c = 0; c < 9 + P; ++c) {
        float v;
        if (c < 3) v = stats[(n * 3 + c) * 4 + 0];
        else if (c < 6) v = stats[(n * 3 + c - 3) * 4 + 1] * inv_hw;
        else if (c < 9) v = stats[(n * 3 + c - 6) * 4 + 2];
        else v = pv[n * P + (c - 9)];
        acc = __builtin_fmaf(v, rcase[(size_t)c * M + j], acc);
    }
    table[(size_t)n * M + j] = acc;
}

// ---- per-plane histogram, torch.histc(x, bins, 0, 1) semantics (tools_origin.py:120-128):
// values outside [0,1] (and NaN) are ignored, x == 1 lands in the last bin, raw counts.
// `copies` private histograms per workgroup (lane l counts into copy l % copies, stored bin-major so that lanes hitting the
// same bin land on different LDS banks): neighbouring pixels fall into the same bin, and on one shared histogram their
// LDS atomics serialise.  Counts are integers: the float atomics that merge the workgroups are exact in any order.
__global__ __launch_bounds__(256) void histc_kernel(const float *__restrict__ x, float *__restrict__ hist, int hw,
                                                    int bins, int copies) {
    extern __shared__ unsigned int sh[];
    const int plane = blockIdx.y, cp = threadIdx.x % copies;
    for (int b = threadIdx.x; b < bins * copies; b += blockDim.x) sh[b] = 0u;
    __syncthreads();
    const float *xb = x + (size_t)plane * hw;
    auto count = [&](float v) {
        if (v >= 0.f && v <= 1.f) {
            int pos = (int)(v * (float)bins);
            if (pos >= bins) pos = bins - 1;
            atomicAdd(&sh[pos * copies + cp], 1u);
        }
    };
    if ((hw & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {          // 16-byte loads need an aligned plane base
        const float4 *x4 = reinterpret_cast<const float4 *>(xb);
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw / 4; i += gridDim.x * blockDim.x) {
            const float4 v = x4[i];
            count(v.x); count(v.y); count(v.z); count(v.w);
        }
    } else {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) count(xb[i]);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < bins; b += blockDim.x) {
        unsigned int t = 0u;
        for (int k = 0; k < copies; ++k) t += sh[b * copies + k];
        if (t) atomicAdd(&hist[(size_t)plane * bins + b], (float)t);
    }
}

// ---- mixed-op combiner
struct MixArgs {
    const float *o[RISP_MAX_MIX];
    float *go[RISP_MAX_MIX];
    float w[RISP_MAX_MIX];
    int K;
};

__global__ __launch_bounds__(256) void mix_fwd_kernel(const MixArgs a, float *__restrict__ y, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < a.K; ++k) {
            const float4 v = reinterpret_cast<const float4 *>(a.o[k])[i];
            const float w = a.w[k];
            s.x += v.x * w; s.y += v.y * w; s.z += v.z * w; s.w += v.w * w;
        }
        reinterpret_cast<float4 *>(y)[i] = s;
    }
}

__global__ __launch_bounds__(256) void mix_bwd_kernel(const MixArgs a, const float *__restrict__ gy,
                                                      float *__restrict__ gw, size_t n4) {
    __shared__ float red[RISP_MAX_MIX * 4];
    float acc[RISP_MAX_MIX];
#pragma unroll
    for (int k = 0; k < RISP_MAX_MIX; ++k) acc[k] = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 g = reinterpret_cast<const float4 *>(gy)[i];
#pragma unroll
        for (int k = 0; k < RISP_MAX_MIX; ++k) {
            if (k < a.K) {
                const float4 v = reinterpret_cast<const float4 *>(a.o[k])[i];
                acc[k] += (g.x * v.x + g.y * v.y) + (g.z * v.z + g.w * v.w);
                if (a.go[k]) {
                    const float w = a.w[k];
                    reinterpret_cast<float4 *>(a.go[k])[i] = make_float4(g.x * w, g.y * w, g.z * w, g.w * w);
                }
            }
        }
    }
    block_sum<RISP_MAX_MIX>(acc, red);
    if (threadIdx.x == 0) {      // one partial row per workgroup (gw = scratch here); mix_finish_kernel adds them in order
#pragma unroll
        for (int k = 0; k < RISP_MAX_MIX; ++k)
            if (k < a.K) gw[(size_t)blockIdx.x * RISP_MAX_MIX + k] = acc[k];
    }
}

// one wave per operand: lane l adds the partial rows l, l + 64, ... in index order, a fixed shuffle tree adds the lanes
__global__ void mix_finish_kernel(const float *__restrict__ part, float *__restrict__ gw, int K, int blocks) {
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (k >= K) return;
    float s = 0.f;
    for (int b = lane; b < blocks; b += 64) s += part[(size_t)b * RISP_MAX_MIX + k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) gw[k] = s;
}

// ---- sum over H,W of selected channel planes
__global__ __launch_bounds__(256) void plane_sums_kernel(const float *__restrict__ x, float *__restrict__ out, int C,
                                                         int c0, int nc, int hw4) {
    __shared__ float red[4];
    const int n = blockIdx.y / nc, c = blockIdx.y % nc;
    const float4 *xb = reinterpret_cast<const float4 *>(x) + ((size_t)n * C + c0 + c) * hw4;
    float acc[1] = {0.f};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw4; i += gridDim.x * blockDim.x) {
        float4 v = xb[i];
        acc[0] += (v.x + v.y) + (v.z + v.w);
    }
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) out[n * nc + c] = acc[0];      // one workgroup per plane: no atomics, bit-repeatable
}

__global__ __launch_bounds__(256) void plane_sums_scalar_kernel(const float *__restrict__ x, float *__restrict__ out,
                                                                int C, int c0, int nc, int hw) {
    __shared__ float red[4];
    const int n = blockIdx.y / nc, c = blockIdx.y % nc;
    const float *xb = x + ((size_t)n * C + c0 + c) * hw;
    float acc[1] = {0.f};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) acc[0] += xb[i];
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) out[n * nc + c] = acc[0];
}

// ---- truncating uint8 conversion + squared error (utils/util.py:130-131,141-154)
__device__ __forceinline__ float to_u8(float v) {
    float t = v * 255.f;
    t = t < 0.f ? 0.f : (t > 255.f ? 255.f : t);
    return floorf(t);  // astype(uint8) truncates; t >= 0 so trunc == floor
}
__global__ __launch_bounds__(256) void sse_u8_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                     double *__restrict__ sse, size_t n) {
    __shared__ double red[4];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float d = to_u8(a[i]) / 255.f - to_u8(b[i]) / 255.f;
        acc += (double)(d * d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sse[1 + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);      // this workgroup's slot
}

// sse[0] = the workgroups' partial sums in index order (lane l: l, l + 64, ...; a fixed tree over the lanes): the same bits on every
// run (round 4 merged them with a double atomicAdd)
__global__ __launch_bounds__(64) void sse_finish_kernel(double *__restrict__ sse, int n) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) acc += sse[1 + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (threadIdx.x == 0) sse[0] = acc;
}

}  // namespace

extern "C" {

size_t risp_channel_stats_scratch_floats(int NC, int HW) { return (size_t)NC * stat_blocks(HW) * 5; }

int risp_channel_stats(const float *x, float *stats, int32_t *arg, float *scratch, int NC, int HW, void *stream) {
    RISP_CHECK_ARG(x && stats && scratch && NC > 0 && NC <= 65535 && HW > 0 && HW % 4 == 0,
                   "risp_channel_stats: bad arguments (NC=%d HW=%d)", NC, HW);
    const int nb = stat_blocks(HW);
    hipLaunchKernelGGL(stats_partial_kernel, dim3(nb, NC), dim3(256), 0, (hipStream_t)stream, x, scratch, HW / 4, nb);
    hipLaunchKernelGGL(stats_final_kernel, dim3(NC), dim3(64), 0, (hipStream_t)stream, scratch, stats, arg, nb);
    RISP_LAUNCH_CHECK("risp_channel_stats");
    return 0;
}

int risp_srcnn_cvals(const float *stats, const float *pv, float *cvals, int N, int P, int HW, void *stream) {
    RISP_CHECK_ARG(stats && cvals && N > 0 && P >= 0 && (P == 0 || pv) && HW > 0, "risp_srcnn_cvals: bad arguments");
    const int total = N * (9 + P);
    hipLaunchKernelGGL(srcnn_cvals_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, pv, cvals,
                       N, P, 1.0f / (float)HW);
    RISP_LAUNCH_CHECK("risp_srcnn_cvals");
    return 0;
}

int risp_srcnn_case_table(const float *stats, const float *pv, const float *rcase, float *table, int N, int P, int HW, int M,
                          void *stream) {
    RISP_CHECK_ARG(stats && rcase && table && N > 0 && N <= 65535 && P >= 0 && (P == 0 || pv) && HW > 0 && M > 0,
                   "risp_srcnn_case_table: bad arguments");
    hipLaunchKernelGGL(srcnn_case_table_kernel, dim3((M + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, stats, pv, rcase,
                       table, P, M, 1.0f / (float)HW);
    RISP_LAUNCH_CHECK("risp_srcnn_case_table");
    return 0;
}

int risp_histc(const float *x, float *hist, int NC, int HW, int bins, void *stream) {
    RISP_CHECK_ARG(x && hist && NC > 0 && NC <= 65535 && HW > 0 && bins > 0 && bins <= 4096, "risp_histc: bad arguments");
    if (hipMemsetAsync(hist, 0, sizeof(float) * NC * bins, (hipStream_t)stream) != hipSuccess) {
        risp_set_error("risp_histc: memset failed");
        return 2;
    }
    int bx = (HW + 256 * 16 - 1) / (256 * 16);
    if (bx > 64) bx = 64;
    int copies = 16;                                   // private histograms per workgroup, within 16 KB of LDS
    while (copies > 1 && bins * copies > 4096) copies >>= 1;
    hipLaunchKernelGGL(histc_kernel, dim3(bx, NC), dim3(256), bins * copies * sizeof(unsigned int), (hipStream_t)stream, x, hist,
                       HW, bins, copies);
    RISP_LAUNCH_CHECK("risp_histc");
    return 0;
}

static int stats_bwd_launch(const char *name, float *gx, const float *g_min, const float *g_mean, const float *g_max,
                            const int32_t *arg, int NC, int HW, int C, int gstride, void *stream) {
    RISP_CHECK_ARG(gx && NC > 0 && NC <= 65535 && HW > 0 && HW % 4 == 0, "%s: bad arguments", name);
    RISP_CHECK_ARG(arg || (!g_min && !g_max), "%s: arg indices required for min/max gradients", name);
    int bx = (HW / 4 + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(stats_bwd_kernel, dim3(bx, NC), dim3(256), 0, (hipStream_t)stream, gx, g_min, g_mean, g_max,
                       arg, HW / 4, 1.0f / (float)HW, C, gstride);
    RISP_LAUNCH_CHECK(name);
    return 0;
}

int risp_stats_bwd(float *gx, const float *g_min, const float *g_mean, const float *g_max, const int32_t *arg, int NC,
                   int HW, void *stream) {
    return stats_bwd_launch("risp_stats_bwd", gx, g_min, g_mean, g_max, arg, NC, HW, 1, 1, stream);
}

int risp_stats_bwd_rows(float *gx, const float *g_min, const float *g_mean, const float *g_max, const int32_t *arg, int N,
                        int C, int HW, int row_stride, void *stream) {
    RISP_CHECK_ARG(N > 0 && C > 0 && row_stride >= C, "risp_stats_bwd_rows: bad shape N=%d C=%d row_stride=%d", N, C, row_stride);
    return stats_bwd_launch("risp_stats_bwd_rows", gx, g_min, g_mean, g_max, arg, N * C, HW, C, row_stride, stream);
}

int risp_grayworld_gains_fwd(const float *stats, float *gains, int N, int HW, void *stream) {
    RISP_CHECK_ARG(stats && gains && N > 0 && HW > 0, "risp_grayworld_gains_fwd: bad arguments");
    hipLaunchKernelGGL(gray_gains_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, stats, gains, N,
                       1.0f / (float)HW);
    RISP_LAUNCH_CHECK("risp_grayworld_gains_fwd");
    return 0;
}

int risp_grayworld_gains_bwd(const float *stats, const float *g_gains, float *g_mean, int N, int HW, void *stream) {
    RISP_CHECK_ARG(stats && g_gains && g_mean && N > 0 && HW > 0, "risp_grayworld_gains_bwd: bad arguments");
    hipLaunchKernelGGL(gray_gains_bwd_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, stats, g_gains,
                       g_mean, N, 1.0f / (float)HW);
    RISP_LAUNCH_CHECK("risp_grayworld_gains_bwd");
    return 0;
}

static int mix_args(const char *name, const float *const *outs, const float *w, int K, float *const *go, MixArgs &a,
                    size_t numel) {
    RISP_CHECK_ARG(outs && w && K >= 1 && K <= RISP_MAX_MIX && numel > 0 && numel % 4 == 0,
                   "%s: bad arguments (K=%d numel=%zu)", name, K, numel);
    a.K = K;
    for (int k = 0; k < RISP_MAX_MIX; ++k) {
        a.o[k] = k < K ? outs[k] : nullptr;
        a.go[k] = (k < K && go) ? go[k] : nullptr;
        a.w[k] = k < K ? w[k] : 0.f;
        RISP_CHECK_ARG(k >= K || a.o[k], "%s: null operand %d", name, k);
    }
    return 0;
}

static int mix_grid(size_t n4) {
    size_t b = (n4 + 255) / 256;
    return (int)(b > 2048 ? 2048 : b);
}

int risp_mix_fwd(const float *const *outs, const float *w, int K, float *y, size_t numel, void *stream) {
    MixArgs a;
    if (int e = mix_args("risp_mix_fwd", outs, w, K, nullptr, a, numel)) return e;
    RISP_CHECK_ARG(y, "risp_mix_fwd: null output");
    hipLaunchKernelGGL(mix_fwd_kernel, dim3(mix_grid(numel / 4)), dim3(256), 0, (hipStream_t)stream, a, y, numel / 4);
    RISP_LAUNCH_CHECK("risp_mix_fwd");
    return 0;
}

size_t risp_mix_scratch_floats(void) { return (size_t)1024 * RISP_MAX_MIX; }

int risp_mix_bwd(const float *const *outs, const float *w, int K, const float *gy, float *const *go, float *gw,
                 float *scratch, size_t numel, void *stream) {
    MixArgs a;
    if (int e = mix_args("risp_mix_bwd", outs, w, K, go, a, numel)) return e;
    RISP_CHECK_ARG(gy && gw && scratch, "risp_mix_bwd: null argument");
    size_t b = (numel / 4 + 1023) / 1024;
    int grid = (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
    hipLaunchKernelGGL(mix_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, gy, scratch, numel / 4);
    hipLaunchKernelGGL(mix_finish_kernel, dim3(1), dim3(64 * RISP_MAX_MIX), 0, (hipStream_t)stream, scratch, gw, K, grid);
    RISP_LAUNCH_CHECK("risp_mix_bwd");
    return 0;
}

int risp_plane_sums(const float *x, float *out, int N, int C, int c0, int nc, int HW, void *stream) {
    RISP_CHECK_ARG(x && out && N > 0 && C > 0 && c0 >= 0 && nc > 0 && c0 + nc <= C && HW > 0 && N * nc <= 65535,
                   "risp_plane_sums: bad arguments");
    const int bx = 1;                                   // one workgroup per plane: deterministic, no atomics
    if (HW % 4 == 0)
        hipLaunchKernelGGL(plane_sums_kernel, dim3(bx, N * nc), dim3(256), 0, (hipStream_t)stream, x, out, C, c0, nc,
                           HW / 4);
    else
        hipLaunchKernelGGL(plane_sums_scalar_kernel, dim3(bx, N * nc), dim3(256), 0, (hipStream_t)stream, x, out, C, c0,
                           nc, HW);
    RISP_LAUNCH_CHECK("risp_plane_sums");
    return 0;
}

size_t risp_sse_uint8_doubles(void) { return 1 + 1024; }

int risp_sse_uint8(const float *a, const float *b, double *sse, size_t sse_doubles, size_t numel, void *stream) {
    RISP_CHECK_ARG(a && b && sse && numel > 0, "risp_sse_uint8: bad arguments");
    RISP_CHECK_ARG(sse_doubles >= risp_sse_uint8_doubles(), "risp_sse_uint8: sse holds %zu doubles, needs risp_sse_uint8_doubles() = %zu",
                   sse_doubles, risp_sse_uint8_doubles());
    size_t b_ = (numel + 256 * 16 - 1) / (256 * 16);
    int grid = (int)(b_ < 1 ? 1 : (b_ > 1024 ? 1024 : b_));
    hipLaunchKernelGGL(sse_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, sse, numel);
    hipLaunchKernelGGL(sse_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sse, grid);
    RISP_LAUNCH_CHECK("risp_sse_uint8");
    return 0;
}

}  // extern "C"
