// Host-logic helpers of the DARTS super-net as single launches (gfx950).  At a per-GPU batch of 4 the search step
// spent 7 % of its GPU time in ~1500 sub-5-us PyTorch kernels per iteration that do this bookkeeping:
//
//   mixture weights of a slot (super_prune_fifteen_demos_four_bayer_two.py:185-193):
//       prob = softmax(alpha); mask = prob.detach() < threshold * prob.detach().max(); post = prob.clone();
//       post[mask] = 0; post = post / post.sum().detach()
//     -> risp_prune_softmax_fwd / _bwd (one wave; ~10 element-wise / reduction kernels each way before)
//   per-image parameter blocks of the surviving ops of a slot (:204-209): sigmoid(par).repeat(N, 1) for each op
//     -> risp_param_blocks_fwd / _bwd (one launch for all ops of the slot; 2 + 3 kernels PER OP before)
#include "risp_common.h"

namespace {

__global__ __launch_bounds__(64) void prune_softmax_fwd_kernel(const float *__restrict__ alpha,
                                                               const unsigned char *__restrict__ unavailable, float threshold,
                                                               int K, float *__restrict__ probs, float *__restrict__ coef,
                                                               float *__restrict__ post) {
    const int k = threadIdx.x;
    float a = -INFINITY;
    if (k < K && !(unavailable && unavailable[k])) a = alpha[k];
    float mx = a;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const float e = k < K ? expf(a - mx) : 0.f;          // exp(-inf) = 0: an unavailable op has probability exactly 0
    float s = e;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float p = e / s;
    float pm = p;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pm = fmaxf(pm, __shfl_xor(pm, o, 64));
    const float kept = (k < K && !(p < threshold * pm)) ? p : 0.f;      // strict <: the reference's prune rule
    float ks = kept;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ks += __shfl_xor(ks, o, 64);
    if (k < K) {
        probs[k] = p;
        coef[k] = kept > 0.f || !(p < threshold * pm) ? 1.f / ks : 0.f;   // d post_k / d prob_k (mask and sum are detached)
        post[k] = kept / ks;
    }
}

// galpha_i = t_i - prob_i * sum_k t_k,  t_k = gpost_k * coef_k * prob_k   (softmax backward through the kept entries)
__global__ __launch_bounds__(64) void prune_softmax_bwd_kernel(const float *__restrict__ probs, const float *__restrict__ coef,
                                                               const float *__restrict__ gpost, int K,
                                                               float *__restrict__ galpha) {
    const int k = threadIdx.x;
    const float p = k < K ? probs[k] : 0.f;
    const float t = k < K ? gpost[k] * coef[k] * p : 0.f;
    float s = t;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (k < K) galpha[k] = t - p * s;
}

__device__ __forceinline__ float sigmoid_acc(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void param_blocks_fwd_kernel(const risp_param_blocks_desc d) {
    const int k = blockIdx.y, w = d.width[k], total = d.N * w;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x)
        d.block[k][i] = sigmoid_acc(d.raw[k][i % w]);
}

// graw[k][j] = sigmoid'(raw[k][j]) * sum_n gblock[k][n][j]   (the sum = .repeat's backward, images in index order)
__global__ __launch_bounds__(64) void param_blocks_bwd_kernel(const risp_param_blocks_desc d) {
    const int k = blockIdx.x, w = d.width[k], j = threadIdx.x;
    if (j >= w) return;
    float s = 0.f;
    if (d.gblock[k]) {
        const int row = d.gstride[k] ? d.gstride[k] : w;
        for (int n = 0; n < d.N; ++n) s += d.gblock[k][n * row + j];
    }
    const float y = sigmoid_acc(d.raw[k][j]);
    d.graw[k][j] = s * ((1.f - y) * y);
}

}  // namespace

extern "C" {

int risp_prune_softmax_fwd(const float *alpha, const unsigned char *unavailable, float threshold, int K, float *probs, float *coef,
                           float *post, void *stream) {
    RISP_CHECK_ARG(alpha && probs && coef && post && K >= 1 && K <= 64, "risp_prune_softmax_fwd: bad arguments (K=%d, 1..64)", K);
    hipLaunchKernelGGL(prune_softmax_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, alpha, unavailable, threshold, K, probs,
                       coef, post);
    RISP_LAUNCH_CHECK("risp_prune_softmax_fwd");
    return 0;
}

int risp_prune_softmax_bwd(const float *probs, const float *coef, const float *gpost, int K, float *galpha, void *stream) {
    RISP_CHECK_ARG(probs && coef && gpost && galpha && K >= 1 && K <= 64, "risp_prune_softmax_bwd: bad arguments (K=%d)", K);
    hipLaunchKernelGGL(prune_softmax_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, probs, coef, gpost, K, galpha);
    RISP_LAUNCH_CHECK("risp_prune_softmax_bwd");
    return 0;
}

static int check_blocks(const risp_param_blocks_desc *d, const char *who) {
    RISP_CHECK_ARG(d && d->n_ops >= 1 && d->n_ops <= RISP_MAX_PARAM_OPS && d->N >= 1, "%s: bad descriptor", who);
    for (int k = 0; k < d->n_ops; ++k)
        RISP_CHECK_ARG(d->raw[k] && d->width[k] >= 1 && d->width[k] <= 64, "%s: op %d: width %d (1..64) or null pointer", who, k,
                       d->width[k]);
    return 0;
}

int risp_param_blocks_fwd(const risp_param_blocks_desc *d, void *stream) {
    if (check_blocks(d, "risp_param_blocks_fwd")) return 1;
    for (int k = 0; k < d->n_ops; ++k) RISP_CHECK_ARG(d->block[k], "risp_param_blocks_fwd: block %d missing", k);
    int bx = (d->N * 64 + 255) / 256;
    if (bx > 16) bx = 16;
    hipLaunchKernelGGL(param_blocks_fwd_kernel, dim3(bx, d->n_ops), dim3(256), 0, (hipStream_t)stream, *d);
    RISP_LAUNCH_CHECK("risp_param_blocks_fwd");
    return 0;
}

int risp_param_blocks_bwd(const risp_param_blocks_desc *d, void *stream) {
    if (check_blocks(d, "risp_param_blocks_bwd")) return 1;
    for (int k = 0; k < d->n_ops; ++k)
        RISP_CHECK_ARG(d->graw[k] && (d->gstride[k] == 0 || d->gstride[k] >= d->width[k]),
                       "risp_param_blocks_bwd: op %d: gradient buffer missing or row stride %d < width", k, d->gstride[k]);
    hipLaunchKernelGGL(param_blocks_bwd_kernel, dim3(d->n_ops), dim3(64), 0, (hipStream_t)stream, *d);
    RISP_LAUNCH_CHECK("risp_param_blocks_bwd");
    return 0;
}

}  // extern "C"
