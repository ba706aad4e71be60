// The fully connected head of the conditional modules (sRGB pool 16-18): per image, the 3*bins histogram counts go
// through an MLP whose (in,out) row-major weights and biases are slices of ONE flat parameter vector, a scalar
// "global" entry is added and a sigmoid taken (models/modules/tools_origin.py:109-163).  Widths are tens of units,
// so this is launch-count work, not FLOPs: one launch forward, two backward instead of ~4 per layer.
//
//   risp_cond_fc_fwd   one workgroup per image; activations live in LDS; thread j owns output unit j of a layer
//                      (weights row-major (in,out): consecutive threads read consecutive floats).
//   risp_cond_fc_bwd   (1) per image: the deltas of every layer (LDS), written to a (N, sum widths) scratch;
//                      (2) per parameter: dW[i][j] = sum_n a_{l-1}[n,i] * delta_l[n,j], db[j] = sum_n delta_l[n,j],
//                          d(global scalar) = sum_{n,j} delta_L[n,j]; a fixed loop over n - deterministic.
// The histogram input carries no gradient (the reference detaches it, :125).
#include "risp_common.h"

namespace {

constexpr int FC_MAXL = 8;          // layers
constexpr int FC_MAXW = 1024;       // widest layer

struct FcShape {
    int n_layers;                   // number of weight matrices
    int w[FC_MAXL + 1];             // widths w[0] .. w[n_layers]
    int wofs[FC_MAXL], bofs[FC_MAXL], aofs[FC_MAXL + 1];   // offsets of W_l, b_l in flat; of layer l's units in an activation row
    int gofs, arow;                 // offset of the global scalar; activation row length (sum of w[0..L])
};

// acts (N, arow): [hist | a_1 (post-ReLU) | ... | z_L + global (pre-sigmoid)]
__global__ __launch_bounds__(256) void cond_fc_fwd_kernel(const float *__restrict__ hist, const float *__restrict__ flat,
                                                          float *__restrict__ acts, float *__restrict__ out, FcShape s) {
    __shared__ float a[2][FC_MAXW];
    const int n = blockIdx.x, t = threadIdx.x;
    float *row = acts + (size_t)n * s.arow;
    for (int i = t; i < s.w[0]; i += 256) row[i] = a[0][i] = hist[(size_t)n * s.w[0] + i];
    __syncthreads();
    for (int l = 0; l < s.n_layers; ++l) {
        const int fi = s.w[l], fo = s.w[l + 1];
        const float *W = flat + s.wofs[l], *b = flat + s.bofs[l];
        const float *src = a[l & 1];
        float *dst = a[(l + 1) & 1];
        const bool last = l == s.n_layers - 1;
        for (int j = t; j < fo; j += 256) {
            float z = 0.f;
            for (int i = 0; i < fi; ++i) z += src[i] * W[(size_t)i * fo + j];     // feat @ weight: k-ordered like the reference
            z += b[j];
            if (last) {
                z += flat[s.gofs];
                out[(size_t)n * fo + j] = 1.f / (1.f + __expf(-z));
            } else {
                z = z > 0.f ? z : 0.f;
            }
            dst[j] = z;
            row[s.aofs[l + 1] + j] = z;
        }
        __syncthreads();
    }
}

// deltas (N, arow): delta_l at aofs[l] for l = 1..L (slot 0 unused)
__global__ __launch_bounds__(256) void cond_fc_delta_kernel(const float *__restrict__ flat, const float *__restrict__ acts,
                                                            const float *__restrict__ out, const float *__restrict__ gout,
                                                            float *__restrict__ deltas, FcShape s) {
    __shared__ float d[2][FC_MAXW];
    const int n = blockIdx.x, t = threadIdx.x, L = s.n_layers;
    const float *row = acts + (size_t)n * s.arow;
    float *drow = deltas + (size_t)n * s.arow;
    for (int j = t; j < s.w[L]; j += 256) {                   // through the sigmoid
        const float y = out[(size_t)n * s.w[L] + j];
        const float v = gout[(size_t)n * s.w[L] + j] * y * (1.f - y);
        d[L & 1][j] = v;
        drow[s.aofs[L] + j] = v;
    }
    __syncthreads();
    for (int l = L - 1; l >= 1; --l) {                        // delta of layer l's output units from layer l+1's
        const int fo = s.w[l + 1], fi = s.w[l];
        const float *W = flat + s.wofs[l];
        const float *dn = d[(l + 1) & 1];
        for (int i = t; i < fi; i += 256) {
            float v = 0.f;
            for (int j = 0; j < fo; ++j) v += W[(size_t)i * fo + j] * dn[j];
            v = row[s.aofs[l] + i] > 0.f ? v : 0.f;           // ReLU
            d[l & 1][i] = v;
            drow[s.aofs[l] + i] = v;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void cond_fc_params_kernel(const float *__restrict__ acts, const float *__restrict__ deltas,
                                                             float *__restrict__ dflat, int total, int N, FcShape s) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    float g = 0.f;
    if (p == s.gofs) {                                        // the scalar added to every output unit
        const int L = s.n_layers;
        for (int n = 0; n < N; ++n)
            for (int j = 0; j < s.w[L]; ++j) g += deltas[(size_t)n * s.arow + s.aofs[L] + j];
    } else if (p < s.gofs) {
        int l = 0;
        while (l + 1 < s.n_layers && p >= s.wofs[l + 1]) ++l;
        const int fo = s.w[l + 1];
        if (p < s.bofs[l]) {
            const int i = (p - s.wofs[l]) / fo, j = (p - s.wofs[l]) - i * fo;
            for (int n = 0; n < N; ++n)
                g += acts[(size_t)n * s.arow + s.aofs[l] + i] * deltas[(size_t)n * s.arow + s.aofs[l + 1] + j];
        } else {
            const int j = p - s.bofs[l];
            for (int n = 0; n < N; ++n) g += deltas[(size_t)n * s.arow + s.aofs[l + 1] + j];
        }
    }                                                         // the unused tail of the "global" block: zero gradient
    dflat[p] = g;
}

int make_shape(const int *widths, int n_layers, FcShape &s, const char *who) {
    RISP_CHECK_ARG(widths && n_layers >= 1 && n_layers <= FC_MAXL, "%s: 1..%d layers", who, FC_MAXL);
    s.n_layers = n_layers;
    int at = 0, ao = 0;
    for (int l = 0; l <= n_layers; ++l) {
        RISP_CHECK_ARG(widths[l] >= 1 && widths[l] <= FC_MAXW, "%s: layer width %d (1..%d)", who, widths[l], FC_MAXW);
        s.w[l] = widths[l];
        s.aofs[l] = ao;
        ao += widths[l];
    }
    for (int l = 0; l < n_layers; ++l) {
        s.wofs[l] = at;
        at += widths[l] * widths[l + 1];
        s.bofs[l] = at;
        at += widths[l + 1];
    }
    s.gofs = at;
    s.arow = ao;
    return 0;
}

}  // namespace

extern "C" {

int risp_cond_fc_row_floats(const int *widths, int n_layers) {
    int r = 0;
    for (int l = 0; widths && l <= n_layers; ++l) r += widths[l];
    return r;
}

int risp_cond_fc_fwd(const float *hist, const float *flat, const int *widths, int n_layers, float *acts, float *out, int N,
                     void *stream) {
    RISP_CHECK_ARG(hist && flat && acts && out && N > 0, "risp_cond_fc_fwd: bad arguments");
    FcShape s;
    if (int e = make_shape(widths, n_layers, s, "risp_cond_fc_fwd")) return e;
    hipLaunchKernelGGL(cond_fc_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, hist, flat, acts, out, s);
    RISP_LAUNCH_CHECK("risp_cond_fc_fwd");
    return 0;
}

int risp_cond_fc_bwd(const float *flat, const int *widths, int n_layers, const float *acts, const float *out, const float *gout,
                     float *deltas, float *dflat, int total_params, int N, void *stream) {
    RISP_CHECK_ARG(flat && acts && out && gout && deltas && dflat && N > 0, "risp_cond_fc_bwd: bad arguments");
    FcShape s;
    if (int e = make_shape(widths, n_layers, s, "risp_cond_fc_bwd")) return e;
    RISP_CHECK_ARG(total_params > s.gofs, "risp_cond_fc_bwd: %d parameters, the layers alone need %d", total_params, s.gofs + 1);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(cond_fc_delta_kernel, dim3(N), dim3(256), 0, st, flat, acts, out, gout, deltas, s);
    hipLaunchKernelGGL(cond_fc_params_kernel, dim3((total_params + 255) / 256), dim3(256), 0, st, acts, deltas, dflat, total_params,
                       N, s);
    RISP_LAUNCH_CHECK("risp_cond_fc_bwd");
    return 0;
}

}  // extern "C"
