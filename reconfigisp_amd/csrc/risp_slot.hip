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
#include "risp_ops.h"

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


// ---------------------------------------------------------------------------------------------------
// The mixture of a slot with its ELEMENT-WISE operators computed on the fly (super_prune_fifteen_demos_four_bayer_two.py:
// 195-210; tools_origin.py:53-73 gamma, 205-225 manual white balance, 27-45 gray world, 256-262 skip, 317-359
// WbQuadratic, 414-440 GtmManual).  Operand k of  y = sum_k w_k o_k  is either a materialised tensor (the CNN
// proxies) or one of those operators applied to the slot input x in registers: x is read once, y written once, and
// the five or six 12 B/pixel tensors the operators would write - and the mixture re-read - never exist.
// Per operand the value is ctx.fwd(x) of risp_ops.h, i.e. the bits the stand-alone kernel stores, and the sum runs in
// operand order with separate multiply and add like mix_fwd_kernel: y is bit-identical to the unfused slot.
using namespace risp_ops;

template <bool WBQ>
__device__ __forceinline__ void slot_contexts(const risp_slot_mix_desc &d, int n, WbManualCtx &wm, GammaCtx &ga, GtmCtx &gt,
                                              WbqCtx &wq, float (&g3)[3]) {
    for (int k = 0; k < d.K; ++k) {
        const float *p = d.ptr[k];
        switch (d.kind[k]) {
            case RISP_OP_WB_MANUAL:
#pragma unroll
                for (int c = 0; c < 3; ++c) wm.k[c] = p[n * 3 + c] * d.pmul[k];      // gain = params * 5, tools_origin.py:214
                break;
            case RISP_OP_GAMMA: ga = GammaCtx(p, n); break;
            case RISP_OP_GTM_MANUAL: gt = GtmCtx(p, n); break;
            case RISP_OP_WB_QUADRATIC:
                if constexpr (WBQ) wq = WbqCtx(p, n);
                break;
            case RISP_OP_GAIN3:
#pragma unroll
                for (int c = 0; c < 3; ++c) g3[c] = p[n * 3 + c];
                break;
            default: break;
        }
    }
}

__device__ __forceinline__ f3 gain3_fwd(const float (&k)[3], f3 v) { return {clamp01(v.b * k[0]), clamp01(v.g * k[1]), clamp01(v.r * k[2])}; }
__device__ __forceinline__ f3 gain3_bwd(const float (&k)[3], f3 x, f3 g, float *acc) {
    g.b *= gate01(x.b * k[0]);
    g.g *= gate01(x.g * k[1]);
    g.r *= gate01(x.r * k[2]);
    acc[0] += g.b * x.b;
    acc[1] += g.g * x.g;
    acc[2] += g.r * x.r;
    return {g.b * k[0], g.g * k[1], g.r * k[2]};
}

template <bool WBQ>
__global__ __launch_bounds__(256) void slot_mix_fwd_kernel(const risp_slot_mix_desc d, int hw4) {
    const int n = blockIdx.y;
    WbManualCtx wm; GammaCtx ga; GtmCtx gt; WbqCtx wq; float g3[3] = {0.f, 0.f, 0.f};
    slot_contexts<WBQ>(d, n, wm, ga, gt, wq, g3);
    const size_t base = (size_t)n * 3 * hw4;
    const float4 *xb = reinterpret_cast<const float4 *>(d.x) + base;
    float4 *yb = reinterpret_cast<float4 *>(d.y) + base;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw4; i += gridDim.x * blockDim.x) {
        const float4 b = xb[i], g = xb[hw4 + i], r = xb[2 * hw4 + i];
        const f3 px[4] = {{b.x, g.x, r.x}, {b.y, g.y, r.y}, {b.z, g.z, r.z}, {b.w, g.w, r.w}};
        f3 s[4] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        for (int k = 0; k < d.K; ++k) {
            const float w = d.w[k];
            f3 o[4];
            if (d.kind[k] == RISP_SLOT_TENSOR) {
                const float4 *ob = reinterpret_cast<const float4 *>(d.ptr[k]) + base;
                const float4 vb = ob[i], vg = ob[hw4 + i], vr = ob[2 * hw4 + i];
                o[0] = {vb.x, vg.x, vr.x}; o[1] = {vb.y, vg.y, vr.y}; o[2] = {vb.z, vg.z, vr.z}; o[3] = {vb.w, vg.w, vr.w};
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    switch (d.kind[k]) {
                        case RISP_OP_WB_MANUAL: o[q] = wm.fwd(px[q]); break;
                        case RISP_OP_GAMMA: o[q] = ga.fwd(px[q]); break;
                        case RISP_OP_GTM_MANUAL: o[q] = gt.fwd(px[q]); break;
                        case RISP_OP_WB_QUADRATIC:
                            if constexpr (WBQ) o[q] = wq.fwd(px[q]); else o[q] = px[q];
                            break;
                        case RISP_OP_GAIN3: o[q] = gain3_fwd(g3, px[q]); break;
                        default: o[q] = px[q]; break;                  // RISP_OP_SKIP
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s[q].b += o[q].b * w;
                s[q].g += o[q].g * w;
                s[q].r += o[q].r * w;
            }
        }
        yb[i] = make_float4(s[0].b, s[1].b, s[2].b, s[3].b);
        yb[hw4 + i] = make_float4(s[0].g, s[1].g, s[2].g, s[3].g);
        yb[2 * hw4 + i] = make_float4(s[0].r, s[1].r, s[2].r, s[3].r);
    }
}

// Backward of the same: per operand the architecture term <gy, o_k>; tensors receive go_k = w_k gy; the element-wise
// operators' input gradients ctx.bwd(x, w_k gy) are added in KIND order (below) into ONE gx; their parameter gradients are
// reduced registers -> lanes -> waves -> one row of RISP_SLOT_ROW floats per workgroup, which slot_mix_finish_kernel adds
// in index order (bit-repeatable; the block partition is that of the stand-alone backward kernels, so the parameter
// gradients are the bits they produce).  Row layout: [K architecture terms | wb 3 | gamma 1 | gtm 3 | wbq 30 | gain 3].
constexpr int SO_WM = RISP_MAX_MIX, SO_GA = SO_WM + 3, SO_GT = SO_GA + 1, SO_WQ = SO_GT + 3, SO_G3 = SO_WQ + 30;
static_assert(SO_G3 + 3 == RISP_SLOT_ROW, "row layout");

// WBQ: 0 = no quadratic white balance in the slot; 1 = with it, its 30 parameter sums carried here (256 registers: one wave per
// SIMD, 0.34 of the HBM rate); 2 = with it, input gradient only - the 30 sums come from slot_wbq_params_kernel, which reads x and
// gy once more (24 of ~250 B/pixel) at five waves per SIMD.  Same operations in the same order per thread and the same block
// partition: the same bits.
#ifndef RISP_SLOT_NT
#define RISP_SLOT_NT 0
#endif
#if RISP_SLOT_NT
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define SLOT_ST(p, v) do { const float4 v_ = (v); __builtin_nontemporal_store(f32x4_t{v_.x, v_.y, v_.z, v_.w}, reinterpret_cast<f32x4_t *>(p)); } while (0)
#else
#define SLOT_ST(p, v) (*(p) = (v))
#endif
template <int WBQ>
__global__ __launch_bounds__(256) void slot_mix_bwd_kernel(const risp_slot_mix_desc d, const float *__restrict__ gy,
                                                           float *__restrict__ gx, float *__restrict__ part, int hw4) {
    constexpr int NACC = WBQ == 1 ? RISP_SLOT_ROW : RISP_SLOT_ROW - 30;       // without WbQuadratic's sums its 30 slots are not carried
    constexpr int G3 = WBQ == 1 ? SO_G3 : SO_WQ;
    __shared__ float red[RISP_SLOT_ROW * 4];
    const int n = blockIdx.y;
    WbManualCtx wm; GammaCtx ga; GtmCtx gt; WbqCtx wq; float g3[3] = {0.f, 0.f, 0.f};
    slot_contexts<WBQ != 0>(d, n, wm, ga, gt, wq, g3);
    // operand index of each element-wise kind (-1: absent); the element-wise input gradients are added in KIND order
    // (skip, white balance, gamma, tone curve, quadratic, gray-world gain) - a fixed order, whatever the operand order
    int at[RISP_OP_GAIN3 + 1];
#pragma unroll
    for (int q = 0; q <= RISP_OP_GAIN3; ++q) at[q] = -1;
    for (int k = 0; k < d.K; ++k)
        if (d.kind[k] != RISP_SLOT_TENSOR) {
#pragma unroll
            for (int q = 0; q <= RISP_OP_GAIN3; ++q)
                if (d.kind[k] == q) at[q] = k;
        }
    const size_t base = (size_t)n * 3 * hw4;
    const float4 *xb = reinterpret_cast<const float4 *>(d.x) + base;
    const float4 *gb = reinterpret_cast<const float4 *>(gy) + base;
    float4 *ob = gx ? reinterpret_cast<float4 *>(gx) + base : nullptr;
    float acc[NACC];              // [0, RISP_MAX_MIX): architecture terms by operand; then the parameter-gradient slots
    float dotk[RISP_OP_GAIN3 + 1];   // architecture terms of the element-wise operands, by kind
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = 0.f;
#pragma unroll
    for (int q = 0; q <= RISP_OP_GAIN3; ++q) dotk[q] = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw4; i += gridDim.x * blockDim.x) {
        const float4 db = gb[i], dg = gb[hw4 + i], dr = gb[2 * hw4 + i];
        // ---- tensor operands: architecture term, and their gradient w_k gy.  (SLOT_TG > 1: the loads of that many operands issued together -
        // measured 2, 3, 4: 252 us each against 251-255: the launch is not short of bytes in flight; NT stores: 264)
#ifndef SLOT_TG
#define SLOT_TG 1
#endif
#pragma unroll
        for (int k0 = 0; k0 < RISP_MAX_MIX; k0 += SLOT_TG) {
            float4 vb[SLOT_TG], vg[SLOT_TG], vr[SLOT_TG];
#pragma unroll
            for (int j = 0; j < SLOT_TG; ++j) {
                const int k = k0 + j;
                if (k < RISP_MAX_MIX && k < d.K && d.kind[k] == RISP_SLOT_TENSOR) {
                    const float4 *tb = reinterpret_cast<const float4 *>(d.ptr[k]) + base;
                    vb[j] = tb[i]; vg[j] = tb[hw4 + i]; vr[j] = tb[2 * hw4 + i];
                }
            }
#pragma unroll
            for (int j = 0; j < SLOT_TG; ++j) {
                const int k = k0 + j;
                if (k < RISP_MAX_MIX && k < d.K && d.kind[k] == RISP_SLOT_TENSOR) {
                    const float w = d.w[k];
                    acc[k < RISP_MAX_MIX ? k : 0] += ((db.x * vb[j].x + db.y * vb[j].y) + (db.z * vb[j].z + db.w * vb[j].w)) +
                                                     ((dg.x * vg[j].x + dg.y * vg[j].y) + (dg.z * vg[j].z + dg.w * vg[j].w)) +
                                                     ((dr.x * vr[j].x + dr.y * vr[j].y) + (dr.z * vr[j].z + dr.w * vr[j].w));
                    if (d.go[k]) {
                        float4 *gk = reinterpret_cast<float4 *>(d.go[k]) + base;
                        SLOT_ST(gk + i, make_float4(db.x * w, db.y * w, db.z * w, db.w * w));
                        SLOT_ST(gk + hw4 + i, make_float4(dg.x * w, dg.y * w, dg.z * w, dg.w * w));
                        SLOT_ST(gk + 2 * hw4 + i, make_float4(dr.x * w, dr.y * w, dr.z * w, dr.w * w));
                    }
                }
            }
        }
        if (!ob) continue;                            // no element-wise operand
        // ---- element-wise operands, kind by kind, pixel by pixel (straight-line code: few live values)
        const float4 b = xb[i], g = xb[hw4 + i], r = xb[2 * hw4 + i];
        const float pb[4] = {b.x, b.y, b.z, b.w}, pg[4] = {g.x, g.y, g.z, g.w}, pr[4] = {r.x, r.y, r.z, r.w};
        const float qb[4] = {db.x, db.y, db.z, db.w}, qg[4] = {dg.x, dg.y, dg.z, dg.w}, qr[4] = {dr.x, dr.y, dr.z, dr.w};
        float ob_[4], og_[4], or_[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f3 px = {pb[q], pg[q], pr[q]}, gq = {qb[q], qg[q], qr[q]};
            f3 sx = {0.f, 0.f, 0.f};
            bool first = true;
            auto term = [&](int kind, f3 o, f3 t) {
                dotk[kind] += (gq.b * o.b + gq.g * o.g) + gq.r * o.r;
                if (first) sx = t;
                else { sx.b += t.b; sx.g += t.g; sx.r += t.r; }
                first = false;
            };
            auto up = [&](int kind) { const float w = d.w[at[kind]]; return f3{gq.b * w, gq.g * w, gq.r * w}; };
            if (at[RISP_OP_SKIP] >= 0) term(RISP_OP_SKIP, px, up(RISP_OP_SKIP));
            if (at[RISP_OP_WB_MANUAL] >= 0) term(RISP_OP_WB_MANUAL, wm.fwd(px), wm.bwd(px, up(RISP_OP_WB_MANUAL), acc + SO_WM));
            if (at[RISP_OP_GAMMA] >= 0) term(RISP_OP_GAMMA, ga.fwd(px), ga.bwd(px, up(RISP_OP_GAMMA), acc + SO_GA));
            if (at[RISP_OP_GTM_MANUAL] >= 0) term(RISP_OP_GTM_MANUAL, gt.fwd(px), gt.bwd(px, up(RISP_OP_GTM_MANUAL), acc + SO_GT));
            if constexpr (WBQ == 1) {
                if (at[RISP_OP_WB_QUADRATIC] >= 0)
                    term(RISP_OP_WB_QUADRATIC, wq.fwd(px), wq.bwd(px, up(RISP_OP_WB_QUADRATIC), acc + SO_WQ));
            } else if constexpr (WBQ == 2) {
                if (at[RISP_OP_WB_QUADRATIC] >= 0) term(RISP_OP_WB_QUADRATIC, wq.fwd(px), wq.bwd_gx(px, up(RISP_OP_WB_QUADRATIC)));
            }
            if (at[RISP_OP_GAIN3] >= 0) term(RISP_OP_GAIN3, gain3_fwd(g3, px), gain3_bwd(g3, px, up(RISP_OP_GAIN3), acc + G3));
            ob_[q] = sx.b; og_[q] = sx.g; or_[q] = sx.r;
        }
        ob[i] = make_float4(ob_[0], ob_[1], ob_[2], ob_[3]);
        ob[hw4 + i] = make_float4(og_[0], og_[1], og_[2], og_[3]);
        ob[2 * hw4 + i] = make_float4(or_[0], or_[1], or_[2], or_[3]);
    }
    // the element-wise operands' architecture terms go to their operand's slot (wave-uniform scatter, unrolled compare)
#pragma unroll
    for (int q = 0; q <= RISP_OP_GAIN3; ++q) {
#pragma unroll
        for (int k = 0; k < RISP_MAX_MIX; ++k)
            if (at[q] == k) acc[k] = dotk[q];
    }
    block_sum<NACC>(acc, red);
    if (threadIdx.x == 0) {
        float *row = part + ((size_t)n * gridDim.x + blockIdx.x) * RISP_SLOT_ROW;
#pragma unroll
        for (int j = 0; j < SO_WQ; ++j) row[j] = acc[j];
        if constexpr (WBQ == 1) {
#pragma unroll
            for (int j = 0; j < 30; ++j) row[SO_WQ + j] = acc[SO_WQ + j] * 10.f;       // WbqCtx::pscale
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) row[SO_G3 + j] = acc[G3 + j];
    }
}

// The 30 parameter sums of the slot's quadratic white balance (see slot_mix_bwd_kernel, WBQ = 2): slots SO_WQ .. SO_WQ + 29 of the
// same partial rows.  The loads run ahead of the arithmetic through LDS (bgr_walk_lds).
#ifndef RISP_WBQ_WAVES
#define RISP_WBQ_WAVES 2
#endif
#ifndef RISP_WBQ_AHEAD
#define RISP_WBQ_AHEAD 1
#endif
__global__ __launch_bounds__(256, RISP_WBQ_WAVES) void slot_wbq_params_kernel(const risp_slot_mix_desc d, const float *__restrict__ gy, float *__restrict__ part,
                                                              int hw4, int bx_rows) {
    __shared__ float red[30 * 4];
#if defined(RISP_WBQ_ABL) && RISP_WBQ_ABL == 3
    return;                                          /* timing only: the empty launch */
#endif
    const int n = blockIdx.y;
    __shared__ float4 stage[(RISP_WBQ_AHEAD + 1) * 6 * 256];
#if defined(RISP_WBQ_ABL) && (RISP_WBQ_ABL == 2 || RISP_WBQ_ABL == 4)
    if (hw4 > 0) hw4 = gridDim.x * blockDim.x;       /* timing only: one vector per thread - the kernel's fixed cost */
#endif
    const size_t base = (size_t)n * 3 * hw4;
    const float4 *xb = reinterpret_cast<const float4 *>(d.x) + base;
    const float4 *gb = reinterpret_cast<const float4 *>(gy) + base;
    const BgrWalkLds<RISP_WBQ_AHEAD> walk(xb, gb, hw4, stage);
    walk.prime();                                      // the first rows travel while the operand is looked up and its coefficients are formed
    int at = -1;
    for (int k = 0; k < d.K; ++k)
        if (d.kind[k] == RISP_OP_WB_QUADRATIC) at = k;
    if (at < 0) {
        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
        return;
    }
    const WbqCtx wq(d.ptr[at], n);
    const float w = d.w[at];
    float acc[30];
#pragma unroll
    for (int j = 0; j < 30; ++j) acc[j] = 0.f;
    walk.run([&](const BgrVec6 &c, int) {           // the walk of bgr_bwd_kernel<WbqCtx> (risp_pointwise.hip)
#ifdef RISP_WBQ_ABL            /* timing only: the walk without the arithmetic */
        acc[0] += ((c.b.x + c.g.x) + (c.r.x + c.db.x)) + (c.dg.x + c.dr.x) * w;
        acc[1] += ((c.b.y + c.g.y) + (c.r.y + c.db.y)) + (c.dg.y + c.dr.y) * w;
        acc[2] += ((c.b.z + c.g.z) + (c.r.z + c.db.z)) + (c.dg.z + c.dr.z) * w;
        acc[3] += ((c.b.w + c.g.w) + (c.r.w + c.db.w)) + (c.dg.w + c.dr.w) * w;
        return;
#endif
        wq.bwd_gp({c.b.x, c.g.x, c.r.x}, {c.db.x * w, c.dg.x * w, c.dr.x * w}, acc);
        wq.bwd_gp({c.b.y, c.g.y, c.r.y}, {c.db.y * w, c.dg.y * w, c.dr.y * w}, acc);
        wq.bwd_gp({c.b.z, c.g.z, c.r.z}, {c.db.z * w, c.dg.z * w, c.dr.z * w}, acc);
        wq.bwd_gp({c.b.w, c.g.w, c.r.w}, {c.db.w * w, c.dg.w * w, c.dr.w * w}, acc);
    });
#if defined(RISP_WBQ_ABL) && RISP_WBQ_ABL == 4
    if (acc[0] != 12345.f) return;                   /* timing only: one vector per thread, no reduction, no row */
#endif
#if defined(RISP_WBQ_ABL) && RISP_WBQ_ABL >= 4
    hw4 = 0;
#endif
    const float t = block_sum_dpp_lanes<30>(acc, red);
    if (threadIdx.x < 30) {
        // this workgroup's row of the main kernel's bx_rows partial rows per image; the rows beyond this kernel's own (coarser)
        // grid receive zeros - the finish kernel adds all bx_rows in index order, and adding 0 changes no bit
        for (int rb = blockIdx.x; rb < bx_rows; rb += gridDim.x)
            part[((size_t)n * bx_rows + rb) * RISP_SLOT_ROW + SO_WQ + threadIdx.x] = rb == (int)blockIdx.x ? t * 10.f : 0.f;      // WbqCtx::pscale
    }
}

// one wave per output element; lane l adds the partial rows l, l + 64, ... in index order, a fixed shuffle tree adds the lanes.
// Elements: K architecture terms (all rows), then per element-wise operand its (N, P) parameter-gradient block: the
// image's own bx rows (GtmManual: every row into image 0, zeros elsewhere - its knots come from params[0] only).
__global__ __launch_bounds__(256) void slot_mix_finish_kernel(const risp_slot_mix_desc d, const float *__restrict__ part,
                                                              float *__restrict__ gw, int N, int bx) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    int e = t;
    if (e < d.K) {
        float s = 0.f;
        for (int i = lane; i < N * bx; i += 64) s += part[(size_t)i * RISP_SLOT_ROW + e];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) gw[e] = s;
        return;
    }
    e -= d.K;
    for (int k = 0; k < d.K; ++k) {
        int np = 0, off = 0;
        switch (d.kind[k]) {
            case RISP_OP_WB_MANUAL: np = 3; off = SO_WM; break;
            case RISP_OP_GAMMA: np = 1; off = SO_GA; break;
            case RISP_OP_GTM_MANUAL: np = 3; off = SO_GT; break;
            case RISP_OP_WB_QUADRATIC: np = 30; off = SO_WQ; break;
            case RISP_OP_GAIN3: np = 3; off = SO_G3; break;
            default: break;
        }
        if (np == 0 || !d.gp[k]) continue;
        if (e < N * np) {
            const int row = e / np, j = e - row * np;
            const bool whole = d.kind[k] == RISP_OP_GTM_MANUAL;
            float s = 0.f;
            if (!whole || row == 0) {
                const int lo = whole ? 0 : row * bx, hi = whole ? N * bx : (row + 1) * bx;
                for (int i = lo + lane; i < hi; i += 64) s += part[(size_t)i * RISP_SLOT_ROW + off + j];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) d.gp[k][e] = d.kind[k] == RISP_OP_WB_MANUAL ? s * d.pmul[k] : s;
            return;
        }
        e -= N * np;
    }
}

int check_slot(const risp_slot_mix_desc *d, const char *who, bool &wbq, bool &pointwise) {
    RISP_CHECK_ARG(d && d->K >= 1 && d->K <= RISP_MAX_MIX && d->N >= 1 && d->N <= 65535 && d->HW > 0 && d->HW % 4 == 0,
                   "%s: bad descriptor", who);
    int seen = 0;
    wbq = pointwise = false;
    for (int k = 0; k < d->K; ++k) {
        const int kind = d->kind[k];
        RISP_CHECK_ARG(kind == RISP_SLOT_TENSOR || (kind >= RISP_OP_SKIP && kind <= RISP_OP_GAIN3 && kind != RISP_OP_DEMOSAIC_NEAREST),
                       "%s: operand %d: kind %d", who, k, kind);
        RISP_CHECK_ARG(kind == RISP_OP_SKIP || d->ptr[k], "%s: operand %d: null pointer", who, k);
        RISP_CHECK_ARG(kind != RISP_SLOT_TENSOR ? true : (reinterpret_cast<uintptr_t>(d->ptr[k]) & 15) == 0, "%s: operand %d unaligned", who, k);
        if (kind != RISP_SLOT_TENSOR) {
            RISP_CHECK_ARG(!(seen & (1 << kind)), "%s: two element-wise operands of kind %d (at most one each)", who, kind);
            seen |= 1 << kind;
            pointwise = true;
            wbq |= kind == RISP_OP_WB_QUADRATIC;
        }
    }
    RISP_CHECK_ARG(!pointwise || d->x, "%s: element-wise operands need the slot input x", who);
    RISP_CHECK_ARG(((reinterpret_cast<uintptr_t>(d->x) | reinterpret_cast<uintptr_t>(d->y)) & 15) == 0, "%s: unaligned x / y", who);
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

size_t risp_slot_mix_scratch_floats(int N, int HW) { return (size_t)N * risp_bwd_blocks(N, HW) * RISP_SLOT_ROW; }

int risp_slot_mix_fwd(const risp_slot_mix_desc *d, void *stream) {
    bool wbq, pw;
    if (check_slot(d, "risp_slot_mix_fwd", wbq, pw)) return 1;
    RISP_CHECK_ARG(d->y, "risp_slot_mix_fwd: null output");
    const int hw4 = d->HW / 4;
    int bx = (hw4 + 255) / 256;
    if (bx > 64) bx = 64;
    if (wbq) hipLaunchKernelGGL(slot_mix_fwd_kernel<true>, dim3(bx, d->N), dim3(256), 0, (hipStream_t)stream, *d, hw4);
    else hipLaunchKernelGGL(slot_mix_fwd_kernel<false>, dim3(bx, d->N), dim3(256), 0, (hipStream_t)stream, *d, hw4);
    RISP_LAUNCH_CHECK("risp_slot_mix_fwd");
    return 0;
}

int risp_slot_mix_bwd(const risp_slot_mix_desc *d, const float *gy, float *gx, float *gw, float *scratch, void *stream) {
    bool wbq, pw;
    if (check_slot(d, "risp_slot_mix_bwd", wbq, pw)) return 1;
    RISP_CHECK_ARG(gy && gw && scratch && (gx || !pw) && ((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(gx)) & 15) == 0,
                   "risp_slot_mix_bwd: null or unaligned argument");
    const int hw4 = d->HW / 4, bx = risp_bwd_blocks(d->N, d->HW);
#ifndef RISP_SLOT_WBQ_ONE_PASS
#define RISP_SLOT_WBQ_ONE_PASS 1      /* 1 = the 30 sums inside the main backward kernel (round 6: 250 us against 241 + 29 in two launches); 0: A/B */
#endif
    if (wbq && RISP_SLOT_WBQ_ONE_PASS) {
        hipLaunchKernelGGL(slot_mix_bwd_kernel<1>, dim3(bx, d->N), dim3(256), 0, (hipStream_t)stream, *d, gy, gx, scratch, hw4);
    } else if (wbq) {
        hipLaunchKernelGGL(slot_mix_bwd_kernel<2>, dim3(bx, d->N), dim3(256), 0, (hipStream_t)stream, *d, gy, gx, scratch, hw4);
        hipLaunchKernelGGL(slot_wbq_params_kernel, dim3(risp_bwd_blocks_wbq(d->N, d->HW), d->N), dim3(256), 0, (hipStream_t)stream, *d, gy, scratch, hw4, bx);
    } else {
        hipLaunchKernelGGL(slot_mix_bwd_kernel<0>, dim3(bx, d->N), dim3(256), 0, (hipStream_t)stream, *d, gy, gx, scratch, hw4);
    }
    int elems = d->K;
    for (int k = 0; k < d->K; ++k) {
        const int np = d->kind[k] == RISP_OP_WB_MANUAL || d->kind[k] == RISP_OP_GTM_MANUAL || d->kind[k] == RISP_OP_GAIN3 ? 3 :
                       d->kind[k] == RISP_OP_GAMMA ? 1 : d->kind[k] == RISP_OP_WB_QUADRATIC ? 30 : 0;
        if (np && d->gp[k]) elems += d->N * np;
    }
    hipLaunchKernelGGL(slot_mix_finish_kernel, dim3((elems + 3) / 4), dim3(256), 0, (hipStream_t)stream, *d, scratch, gw, d->N, bx);
    RISP_LAUNCH_CHECK("risp_slot_mix_bwd");
    return 0;
}

}  // extern "C"
