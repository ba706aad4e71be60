// One launch per LAYER of a super-net slot instead of one per operator (gfx950).
//
// The 15 operators of an sRGB slot (super_prune_fifteen_demos_four_bayer_two.py:35-52, looped at :183-212) include 8
// SRCNNRes proxies of identical geometry (srcnn_res_arch.py:15-53; only their parameter-channel count P differs, and P
// lives in the folded constant tables, not in the convolution shapes) and the demosaic slot 2 SRCNNDemosaic proxies.
// Run one by one, each is 6 launches forward and 6 backward whose grids, at the 4-image per-GPU batch of the 8-GPU
// search, are one or two rounds of workgroups.  Grouped, the members' images are stacked along N and every layer is ONE
// launch (risp_conv_desc.group_n: group index in the grid, per-group weights); this file holds the non-convolution
// pieces of that chain for a whole group at a time:
//
//   risp_srcnn_case_table_group   the folded-constant tables of all members (forward)
//   risp_srcnn_const_grad_group   the gradients of all members' constants (backward)
//   risp_group_sum                the members' input gradients added in member order, with the backward of the
//                                 min / mean / max statistics (srcnn_res_arch.py:36-40) applied on the way
//
// Every per-member value is computed by the expression sequence of the per-operator kernels (risp_reduce.hip,
// risp_conv_small.hip), so a grouped launch and G single launches give the same bits.
#include "risp_common.h"

namespace {

// table[g * N + n][j] = sum_c cval_g[n][c] * rcase_g[c][j], c in index order (risp_reduce.hip::srcnn_case_table_kernel)
__global__ __launch_bounds__(256) void case_table_group_kernel(const float *__restrict__ stats, const risp_srcnn_group_desc d,
                                                               float *__restrict__ table, float inv_hw) {
    const int g = blockIdx.z, n = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x, M = d.M, P = d.P[g];
    if (j >= M) return;
    const float *__restrict__ pv = d.pv[g];
    const float *__restrict__ rcase = d.rcase[g];
    float acc = 0.f;
    for (int c = 0; c < 9 + P; ++c) {
        float v;
        if (c < 3) v = stats[(n * 3 + c) * 4 + 0];
        else if (c < 6) v = stats[(n * 3 + c - 3) * 4 + 1] * inv_hw;
        else if (c < 9) v = stats[(n * 3 + c - 6) * 4 + 2];
        else v = pv[n * P + (c - 9)];
        acc = __builtin_fmaf(v, rcase[(size_t)c * M + j], acc);
    }
    table[((size_t)g * d.N + n) * M + j] = acc;
}

// gconst[g * N + n][c] = sum_j rs[g * N + n][j] * wconst_g[j][c]  (risp_conv_small.hip::const_grad_kernel: the same
// residue classes, chains and reduction tree); columns >= 9 + P_g of the `row`-wide output are written as zeros.
__global__ __launch_bounds__(256) void const_grad_group_kernel(const float *__restrict__ rs, const risp_srcnn_group_desc d,
                                                               float *__restrict__ gconst, int row) {
    const int c = blockIdx.x, n = blockIdx.y, g = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = d.M, C = 9 + d.P[g];
    const size_t img = (size_t)g * d.N + n;
    if (c >= C) {
        if (tid == 0) gconst[img * row + c] = 0.f;
        return;
    }
    const float *__restrict__ wconst = d.wconst[g];
    float a0 = 0.f, a1 = 0.f;
    int j = tid;
    for (; j + 256 < M; j += 512) {
        a0 = __builtin_fmaf(rs[img * M + j], wconst[(size_t)j * C + c], a0);
        a1 = __builtin_fmaf(rs[img * M + j + 256], wconst[(size_t)(j + 256) * C + c], a1);
    }
    if (j < M) a0 = __builtin_fmaf(rs[img * M + j], wconst[(size_t)j * C + c], a0);
    float v = a0 + a1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __shared__ float part[4];
    if (lane == 0) part[wave] = v;
    __syncthreads();
    if (tid == 0) gconst[img * row + c] = (part[0] + part[1]) + (part[2] + part[3]);
}

// out[plane][i] = sum over the members g (in index order) of  stack[g][plane][i] (+ stats backward of member g):
// t = value + g_mean / HW;  t += g_min at the plane's argmin;  t += g_max at its argmax   (risp_reduce.hip::stats_bwd_kernel).
// gstats: (G * N, row) with columns [0,C) = d/d min, [C,2C) = d/d mean, [2C,3C) = d/d max of image g * N + n; NULL = plain sum.
__global__ __launch_bounds__(256) void group_sum_kernel(const float *__restrict__ stack, float *__restrict__ out, int G, int NC,
                                                        int hw4, const float *__restrict__ gstats, int row, int C,
                                                        const int32_t *__restrict__ arg, float inv_hw) {
    const int plane = blockIdx.y, n = plane / C, c = plane - n * C;
    const size_t member = (size_t)NC * hw4;                  // float4 per member
    const float4 *sb = reinterpret_cast<const float4 *>(stack) + (size_t)plane * hw4;
    float4 *ob = reinterpret_cast<float4 *>(out) + (size_t)plane * hw4;
    const int imn = gstats ? arg[plane * 2] : -1, imx = gstats ? arg[plane * 2 + 1] : -1;
    const int N = NC / C;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw4; i += gridDim.x * blockDim.x) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int g = 0; g < G; ++g) {
            float4 v = sb[(size_t)g * member + i];
            if (gstats) {
                const float *gs = gstats + ((size_t)g * N + n) * row;
                const float add = gs[C + c] * inv_hw, vmn = gs[c], vmx = gs[2 * C + c];
                float *e = reinterpret_cast<float *>(&v);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float t = e[k] + add;
                    if (4 * i + k == imn) t += vmn;
                    if (4 * i + k == imx) t += vmx;
                    e[k] = t;
                }
            }
            if (g == 0) s = v;
            else { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        }
        ob[i] = s;
    }
}

int check_group(const risp_srcnn_group_desc *d, const char *who, bool fwd) {
    RISP_CHECK_ARG(d && d->G >= 1 && d->G <= RISP_MAX_GROUP && d->N >= 1 && d->N <= 65535 && d->M >= 1 && d->HW >= 1,
                   "%s: bad descriptor", who);
    for (int g = 0; g < d->G; ++g) {
        RISP_CHECK_ARG(d->P[g] >= 0 && d->P[g] <= 55, "%s: member %d: %d parameter channels", who, g, d->P[g]);
        if (fwd) RISP_CHECK_ARG(d->rcase[g] && (d->P[g] == 0 || d->pv[g]), "%s: member %d: null table or parameter block", who, g);
        else RISP_CHECK_ARG(d->wconst[g], "%s: member %d: null weight table", who, g);
    }
    return 0;
}

}  // namespace

extern "C" {

int risp_srcnn_case_table_group(const float *stats, const risp_srcnn_group_desc *d, float *table, void *stream) {
    if (check_group(d, "risp_srcnn_case_table_group", true)) return 1;
    RISP_CHECK_ARG(stats && table, "risp_srcnn_case_table_group: null tensor");
    hipLaunchKernelGGL(case_table_group_kernel, dim3((d->M + 255) / 256, d->N, d->G), dim3(256), 0, (hipStream_t)stream, stats, *d,
                       table, 1.0f / (float)d->HW);
    RISP_LAUNCH_CHECK("risp_srcnn_case_table_group");
    return 0;
}

int risp_srcnn_const_grad_group(const float *rs, const risp_srcnn_group_desc *d, float *gconst, int row, void *stream) {
    if (check_group(d, "risp_srcnn_const_grad_group", false)) return 1;
    int cmax = 0;
    for (int g = 0; g < d->G; ++g) cmax = 9 + d->P[g] > cmax ? 9 + d->P[g] : cmax;
    RISP_CHECK_ARG(rs && gconst && row >= cmax, "risp_srcnn_const_grad_group: null tensor or row stride %d < %d", row, cmax);
    hipLaunchKernelGGL(const_grad_group_kernel, dim3(row, d->N, d->G), dim3(256), 0, (hipStream_t)stream, rs, *d, gconst, row);
    RISP_LAUNCH_CHECK("risp_srcnn_const_grad_group");
    return 0;
}

int risp_group_sum(const float *stack, float *out, int G, int N, int C, int HW, const float *gstats, int row,
                   const int32_t *arg, void *stream) {
    RISP_CHECK_ARG(stack && out && G >= 1 && N >= 1 && C >= 1 && (size_t)N * C <= 65535 && HW > 0 && HW % 4 == 0,
                   "risp_group_sum: bad arguments (G=%d N=%d C=%d HW=%d)", G, N, C, HW);
    RISP_CHECK_ARG(!gstats || (arg && row >= 3 * C), "risp_group_sum: statistics gradients need arg indices and row >= 3 C");
    RISP_CHECK_ARG(((reinterpret_cast<uintptr_t>(stack) | reinterpret_cast<uintptr_t>(out)) & 15) == 0,
                   "risp_group_sum: tensors must be 16-byte aligned");
    int bx = (HW / 4 + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(group_sum_kernel, dim3(bx, N * C), dim3(256), 0, (hipStream_t)stream, stack, out, G, N * C, HW / 4, gstats,
                       row, C, arg, 1.0f / (float)HW);
    RISP_LAUNCH_CHECK("risp_group_sum");
    return 0;
}

}  // extern "C"
