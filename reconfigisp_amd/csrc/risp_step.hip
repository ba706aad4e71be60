// The glue of a DARTS iteration as own launches (round 4): the pixel loss with its gradient, and the reference's per-parameter
// Python loops - virtual step, finite-difference shifts, architecture gradient (models/darts_model.py:159-180, 204-222, 254-265,
// 299-323) - as ONE launch each over a table of <= RISP_MAX_LIST tiny tensors.  Every sum is taken in a fixed order: results
// repeat bit for bit.
#include "risp_common.h"

namespace {

constexpr int LOSS_BLOCKS = 256;

// pass 1: per-workgroup partial sums of (y - gt)^2 or |y - gt| over a grid-stride walk of 16-byte vectors, and - g != NULL -
// the gradient of the MEAN loss at upstream 1: 2 (y - gt) / numel or sign(y - gt) / numel
template <int KIND>
__global__ __launch_bounds__(256) void loss_partial_kernel(const float4 *__restrict__ y, const float4 *__restrict__ gt, float4 *__restrict__ g,
                                                           float *__restrict__ partial, size_t n4, float inv_n) {
    __shared__ float red[4];
    float s = 0.f;
    const float gs = (KIND == 0 ? 2.f : 1.f) * inv_n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 a = y[i], b = gt[i];
        const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
        if (KIND == 0) {
            s += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            if (g) g[i] = make_float4(d0 * gs, d1 * gs, d2 * gs, d3 * gs);
        } else {
            s += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
            auto sg = [&](float d) { return d > 0.f ? gs : (d < 0.f ? -gs : 0.f); };
            if (g) g[i] = make_float4(sg(d0), sg(d1), sg(d2), sg(d3));
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// pass 2: one wave adds the partial sums in index order (lane l: l, l + 64, ...), a fixed shuffle tree adds the lanes
__global__ __launch_bounds__(64) void loss_finish_kernel(const float *__restrict__ partial, int n, float inv_n, float *__restrict__ loss) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) loss[0] = s * inv_n;
}

// ---------------------------------------------------------------------------------- local / global loss (utils/util_loss.py:26-64)
// loss = MSE over the images flagged local (flag < 1) of (a * gain, b), gain[n][c] = clamp(mean b / (clamp(mean a, 0) + 1e-6), 0.5, 2)
// detached, + MSE over the images flagged global of the 1/4-scale bilinear down-samples (align_corners False, H % 4 == W % 4 == 0:
// sample (i, j) sits at source (4 i + 1.5, 4 j + 1.5): the mean of the 2 x 2 centre of its 4 x 4 cell).  Both branches on the device:
// no boolean indexing, no host read of the flags (the reference's img[glb_flag < 1] synchronises twice per evaluation).
// sums: [2][N * C] plane sums of a and b.  One workgroup per plane and slice of 4-row bands.
__global__ __launch_bounds__(256) void lg_partial_kernel(const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ flag,
                                                         const float *__restrict__ sums, float *__restrict__ g, float *__restrict__ partial,
                                                         int N, int C, int H, int W) {
    __shared__ float red[4];
    const int plane = blockIdx.y, n = plane / C, hw = H * W, w4 = W >> 2, bands = H >> 2;
    int nloc = 0;
    for (int i = 0; i < N; ++i) nloc += flag[i] < 1.f ? 1 : 0;
    const bool local = flag[n] < 1.f;
    const float *pa = a + (size_t)plane * hw, *pb = b + (size_t)plane * hw;
    float *pg = g ? g + (size_t)plane * hw : nullptr;
    float s = 0.f;
    if (local) {
        const float inv_hw = 1.f / (float)hw;
        float gain = (sums[N * C + plane] * inv_hw) / (fmaxf(sums[plane] * inv_hw, 0.f) + 1e-6f);
        gain = fminf(fmaxf(gain, 0.5f), 2.f);
        const float gs = 2.f * gain / ((float)nloc * (float)C * (float)hw);
        const float4 *a4 = reinterpret_cast<const float4 *>(pa), *b4 = reinterpret_cast<const float4 *>(pb);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < (hw >> 2); i += gridDim.x * 256) {
            const float4 x = a4[i], y = b4[i];
            const float d0 = x.x * gain - y.x, d1 = x.y * gain - y.y, d2 = x.z * gain - y.z, d3 = x.w * gain - y.w;
            s += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            if (pg) reinterpret_cast<float4 *>(pg)[i] = make_float4(d0 * gs, d1 * gs, d2 * gs, d3 * gs);
        }
    } else {
        // one thread per 4 x 4 cell: rows 4 i + 1, 4 i + 2, columns 4 j + 1, 4 j + 2 carry the sample, the other 12 a zero gradient
        const float gs = 2.f * 0.25f / ((float)(N - nloc) * (float)C * (float)(bands * w4));
        for (int cidx = blockIdx.x * 256 + threadIdx.x; cidx < bands * w4; cidx += gridDim.x * 256) {
            const int i = cidx / w4, j = cidx - i * w4;
            const size_t o = (size_t)(4 * i) * W + 4 * j;
            const float4 a1 = *reinterpret_cast<const float4 *>(pa + o + W), a2 = *reinterpret_cast<const float4 *>(pa + o + 2 * W);
            const float4 b1 = *reinterpret_cast<const float4 *>(pb + o + W), b2 = *reinterpret_cast<const float4 *>(pb + o + 2 * W);
            const float da = ((a1.y + a1.z) + (a2.y + a2.z)) * 0.25f, db = ((b1.y + b1.z) + (b2.y + b2.z)) * 0.25f;
            const float dd = da - db;
            s += dd * dd;
            if (pg) {
                const float v = dd * gs;
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f), m = make_float4(0.f, v, v, 0.f);
                *reinterpret_cast<float4 *>(pg + o) = z;
                *reinterpret_cast<float4 *>(pg + o + W) = m;
                *reinterpret_cast<float4 *>(pg + o + 2 * W) = m;
                *reinterpret_cast<float4 *>(pg + o + 3 * W) = z;
            }
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)plane * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// one wave: the partial sums of the local and of the global planes, each in index order, each divided by its own count
__global__ __launch_bounds__(64) void lg_finish_kernel(const float *__restrict__ partial, const float *__restrict__ flag, float *__restrict__ loss,
                                                        int N, int C, int H, int W, int per_plane) {
    float sl = 0.f, sg = 0.f;
    int nloc = 0;
    for (int i = 0; i < N; ++i) nloc += flag[i] < 1.f ? 1 : 0;
    for (int i = threadIdx.x; i < N * C * per_plane; i += 64) {
        const float v = partial[i];
        if (flag[(i / per_plane) / C] < 1.f) sl += v; else sg += v;
    }
    sl = wave_sum(sl);
    sg = wave_sum(sg);
    if (threadIdx.x == 0) {
        const float hw = (float)H * (float)W;
        const float ll = nloc > 0 ? sl / ((float)nloc * (float)C * hw) : 0.f;
        const float lg = nloc < N ? sg / ((float)(N - nloc) * (float)C * (hw / 16.f)) : 0.f;
        loss[0] = ll + lg;
    }
}

// ---------------------------------------------------------------------------------- tables of tiny tensors
// one wave per tensor (parameters of 1 .. 64 values, alphas of 2 .. 15)
__global__ __launch_bounds__(64) void virtual_step_kernel(const risp_list_desc d, float momentum, float lr_meta) {
    const int t = blockIdx.x;
    float *vp = d.a[t];
    const float *p = d.b[t], *g = d.c[t], *buf = d.e[t];
    for (int i = threadIdx.x; i < d.numel[t]; i += 64) {
        if (!g) {
            vp[i] = p[i];                                   // no gradient arrived (or an alpha): plain copy
        } else {
            float upd = buf ? buf[i] * momentum : 0.f;      // darts_model.py:208-218, operation by operation
            upd = upd + g[i];
            upd = upd * lr_meta;
            vp[i] = p[i] - upd;
        }
    }
}

// norm = || concatenation of c[t] ||_2 (one workgroup, index order), eps = norm < 1e-6 ? 0 : 0.01 / norm  (:276-277)
// first = 0: continue from the sum of squares a previous piece left in out[0]; last = 0: leave the running sum of squares there
__global__ __launch_bounds__(256) void norm_eps_kernel(const risp_list_desc d, float *__restrict__ out, int first, int last) {
    __shared__ float red[4];
    float s = 0.f;
    for (int t = 0; t < d.n; ++t) {
        const float *c = d.c[t];
        if (!c) continue;
        for (int i = threadIdx.x; i < d.numel[t]; i += 256) s += c[i] * c[i];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float sq = (red[0] + red[1]) + (red[2] + red[3]);
        if (!first) sq = out[0] + sq;
        if (!last) {
            out[0] = sq;
            return;
        }
        const float norm = sqrtf(sq);
        out[0] = norm;
        out[1] = norm < 1e-6f ? 0.f : 0.01f / norm;
    }
}

// a[t] += (factor * scalar[0]) * c[t]      (the +eps / -2 eps / +eps shifts of the parameters, :299-312)
__global__ __launch_bounds__(64) void axpy_scalar_kernel(const risp_list_desc d, const float *__restrict__ scalar, float factor) {
    const int t = blockIdx.x;
    float *a = d.a[t];
    const float *c = d.c[t];
    if (!c) return;
    const float s = factor * scalar[0];
    for (int i = threadIdx.x; i < d.numel[t]; i += 64) a[i] = a[i] + c[i] * s;
}

// a[t] = b[t] - lr_meta * ((c[t] - e[t]) / 2 * eps), zeros where b / c / e is missing or the Hessian term holds a NaN
// (:254-265, 313-323); flags[t] = 1 where a NaN was found
__global__ __launch_bounds__(64) void alpha_grad_kernel(const risp_list_desc d, const float *__restrict__ eps, float lr_meta,
                                                        int *__restrict__ flags) {
    const int t = blockIdx.x;
    float *out = d.a[t];
    const float *da = d.b[t], *pos = d.c[t], *neg = d.e[t];
    const int n = d.numel[t];
    const float e = eps[0];
    bool bad = false;
    float h[4];                                             // numel <= 256: up to 4 values per lane
    for (int k = 0, i = threadIdx.x; k < 4; ++k, i += 64) {
        h[k] = 0.f;
        if (i < n && da && pos && neg) {
            h[k] = (pos[i] - neg[i]) / 2.f * e;
            bad |= h[k] != h[k];
        }
    }
    const bool any_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
    for (int k = 0, i = threadIdx.x; k < 4; ++k, i += 64)
        if (i < n) out[i] = (da && pos && neg && !any_bad) ? da[i] - lr_meta * h[k] : 0.f;
    if (threadIdx.x == 0 && flags) flags[t] = any_bad ? 1 : 0;
}

// torch.optim.SGD with momentum (dampening 0, no weight decay / nesterov), one wave per tensor: buf = first ? g : buf * momentum + g;
// p = p - lr * buf  (a = p, c = g, e = momentum buffer, all updated in place)
__global__ __launch_bounds__(64) void sgd_momentum_kernel(const risp_list_desc d, float lr, float momentum, int first) {
    const int t = blockIdx.x;
    float *p = d.a[t], *buf = const_cast<float *>(d.e[t]);
    const float *g = d.c[t];
    if (!g) return;
    for (int i = threadIdx.x; i < d.numel[t]; i += 64) {
        const float b = first ? g[i] : buf[i] * momentum + g[i];
        buf[i] = b;
        p[i] = p[i] - lr * b;
    }
}

// torch.optim.Adam (no weight decay / amsgrad), one wave per tensor, as torch writes it:
//   exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
//   denom = exp_avg_sq.sqrt() / sqrt(1 - beta2^t) + eps; param.addcdiv_(exp_avg, denom, value = -lr / (1 - beta1^t))
// a = p, c = g, b = exp_avg, e = exp_avg_sq (b and e updated in place); w1 = (float)(1 - beta1), w2 = (float)(1 - beta2) formed in
// double by the caller, as torch forms them (within an ulp of torch.optim.Adam per step, not bit-equal: torch's CPU form multiplies
// by the reciprocal of sqrt(1 - beta2^t) where this divides, and its fused multiply-adds are the compiler's choice)
__global__ __launch_bounds__(64) void adam_kernel(const risp_list_desc d, float lr_step, float w1, float beta2, float w2, float bias2_sqrt, float eps) {
    const int t = blockIdx.x;
    float *p = d.a[t], *ea = const_cast<float *>(d.b[t]), *es = const_cast<float *>(d.e[t]);
    const float *g = d.c[t];
    if (!g) return;
    for (int i = threadIdx.x; i < d.numel[t]; i += 64) {
        const float grad = g[i];
        float m = ea[i], v = es[i];
        m = __builtin_fmaf(w1, grad - m, m);
        v = __builtin_fmaf(w2 * grad, grad, v * beta2);
        const float denom = __builtin_sqrtf(v) / bias2_sqrt + eps;
        ea[i] = m;
        es[i] = v;
        p[i] = p[i] - lr_step * (m / denom);
    }
}

int check_list(const risp_list_desc *d, const char *name, int max_numel) {
    RISP_CHECK_ARG(d && d->n >= 0 && d->n <= RISP_MAX_LIST, "%s: 0..%d tensors", name, RISP_MAX_LIST);
    for (int t = 0; t < d->n; ++t)
        RISP_CHECK_ARG(d->numel[t] > 0 && d->numel[t] <= max_numel && d->a[t], "%s: tensor %d: numel %d (1..%d) or null output", name, t,
                       d->numel[t], max_numel);
    return 0;
}
}  // namespace

extern "C" {

size_t risp_loss_scratch_floats(void) { return LOSS_BLOCKS; }

int risp_pixel_loss(const float *y, const float *gt, float *g, float *loss, float *scratch, size_t numel, int kind, void *stream) {
    RISP_CHECK_ARG(y && gt && loss && scratch && numel > 0 && numel % 4 == 0 && (kind == 0 || kind == 1),
                   "risp_pixel_loss: null tensor, numel %% 4 != 0 or kind not in {0 (mean squared), 1 (mean absolute)}");
    RISP_CHECK_ARG(((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gt) | reinterpret_cast<uintptr_t>(g)) & 15) == 0,
                   "risp_pixel_loss: tensors must be 16-byte aligned");
    const size_t n4 = numel / 4;
    const int blocks = (int)((n4 + 255) / 256 < (size_t)LOSS_BLOCKS ? (n4 + 255) / 256 : LOSS_BLOCKS);
    const float inv_n = 1.f / (float)numel;
    auto y4 = reinterpret_cast<const float4 *>(y), g4 = reinterpret_cast<const float4 *>(gt);
    if (kind == 0)
        hipLaunchKernelGGL(loss_partial_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y4, g4, reinterpret_cast<float4 *>(g), scratch, n4, inv_n);
    else
        hipLaunchKernelGGL(loss_partial_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y4, g4, reinterpret_cast<float4 *>(g), scratch, n4, inv_n);
    RISP_LAUNCH_CHECK("risp_pixel_loss");
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scratch, blocks, inv_n, loss);
    RISP_LAUNCH_CHECK("risp_pixel_loss");
    return 0;
}

size_t risp_local_global_scratch_floats(int N, int C) { return (size_t)N * C * (2 + 64); }

int risp_local_global_l2(const float *a, const float *b, const float *flag, float *g, float *loss, float *scratch, int N, int C, int H, int W,
                         void *stream) {
    RISP_CHECK_ARG(a && b && flag && loss && scratch && N > 0 && C > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0 && (long long)N * C <= 65535,
                   "risp_local_global_l2: null tensor, or H / W no multiple of 4 (N=%d C=%d H=%d W=%d)", N, C, H, W);
    RISP_CHECK_ARG(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(g)) & 15) == 0,
                   "risp_local_global_l2: tensors must be 16-byte aligned");
    if (int st = risp_plane_sums(a, scratch, N, C, 0, C, H * W, stream)) return st;
    if (int st = risp_plane_sums(b, scratch + (size_t)N * C, N, C, 0, C, H * W, stream)) return st;
    int bx = (H * W / 4 + 256 * 8 - 1) / (256 * 8);
    bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
    float *partial = scratch + 2 * (size_t)N * C;
    hipLaunchKernelGGL(lg_partial_kernel, dim3(bx, N * C), dim3(256), 0, (hipStream_t)stream, a, b, flag, scratch, g, partial, N, C, H, W);
    hipLaunchKernelGGL(lg_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, flag, loss, N, C, H, W, bx);
    RISP_LAUNCH_CHECK("risp_local_global_l2");
    return 0;
}

int risp_darts_virtual_step(const risp_list_desc *d, float momentum, float lr_meta, void *stream) {
    if (int st = check_list(d, "risp_darts_virtual_step", 1 << 20)) return st;
    for (int t = 0; t < d->n; ++t) RISP_CHECK_ARG(d->b[t], "risp_darts_virtual_step: tensor %d has no source", t);
    if (d->n == 0) return 0;
    hipLaunchKernelGGL(virtual_step_kernel, dim3(d->n), dim3(64), 0, (hipStream_t)stream, *d, momentum, lr_meta);
    RISP_LAUNCH_CHECK("risp_darts_virtual_step");
    return 0;
}

int risp_list_norm_eps_part(const risp_list_desc *d, float *out, int first, int last, void *stream) {
    RISP_CHECK_ARG(d && out && d->n >= 0 && d->n <= RISP_MAX_LIST, "risp_list_norm_eps: 0..%d tensors, an output of 2 floats", RISP_MAX_LIST);
    hipLaunchKernelGGL(norm_eps_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *d, out, first, last);
    RISP_LAUNCH_CHECK("risp_list_norm_eps");
    return 0;
}

int risp_list_norm_eps(const risp_list_desc *d, float *out, void *stream) { return risp_list_norm_eps_part(d, out, 1, 1, stream); }

int risp_list_axpy_scalar(const risp_list_desc *d, const float *scalar, float factor, void *stream) {
    if (int st = check_list(d, "risp_list_axpy_scalar", 1 << 20)) return st;
    RISP_CHECK_ARG(scalar, "risp_list_axpy_scalar: null scalar");
    if (d->n == 0) return 0;
    hipLaunchKernelGGL(axpy_scalar_kernel, dim3(d->n), dim3(64), 0, (hipStream_t)stream, *d, scalar, factor);
    RISP_LAUNCH_CHECK("risp_list_axpy_scalar");
    return 0;
}

int risp_sgd_momentum_step(const risp_list_desc *d, float lr, float momentum, int first, void *stream) {
    if (int st = check_list(d, "risp_sgd_momentum_step", 1 << 20)) return st;
    for (int t = 0; t < d->n; ++t) RISP_CHECK_ARG(d->e[t], "risp_sgd_momentum_step: tensor %d has no momentum buffer", t);
    if (d->n == 0) return 0;
    hipLaunchKernelGGL(sgd_momentum_kernel, dim3(d->n), dim3(64), 0, (hipStream_t)stream, *d, lr, momentum, first);
    RISP_LAUNCH_CHECK("risp_sgd_momentum_step");
    return 0;
}

int risp_adam_step(const risp_list_desc *d, float lr_step, float beta2, float one_minus_beta1, float one_minus_beta2, float bias2_sqrt,
                   float eps, void *stream) {
    if (int st = check_list(d, "risp_adam_step", 1 << 20)) return st;
    for (int t = 0; t < d->n; ++t) RISP_CHECK_ARG(d->b[t] && d->e[t], "risp_adam_step: tensor %d has no moment buffers", t);
    if (d->n == 0) return 0;
    hipLaunchKernelGGL(adam_kernel, dim3(d->n), dim3(64), 0, (hipStream_t)stream, *d, lr_step, one_minus_beta1, beta2, one_minus_beta2, bias2_sqrt, eps);
    RISP_LAUNCH_CHECK("risp_adam_step");
    return 0;
}

int risp_darts_alpha_grad(const risp_list_desc *d, const float *eps, float lr_meta, int *nan_flags, void *stream) {
    if (int st = check_list(d, "risp_darts_alpha_grad", 256)) return st;
    RISP_CHECK_ARG(eps, "risp_darts_alpha_grad: null eps");
    if (d->n == 0) return 0;
    hipLaunchKernelGGL(alpha_grad_kernel, dim3(d->n), dim3(64), 0, (hipStream_t)stream, *d, eps, lr_meta, nan_flags);
    RISP_LAUNCH_CHECK("risp_darts_alpha_grad");
    return 0;
}

}  // extern "C"
