// Element-wise ISP operators and the fused element-wise pipeline segment (gfx950).
//
// All of these are HBM-bound: every thread moves 16-byte vectors (float4) of planar
// NCHW data, consecutive lanes touch consecutive 16-byte slots (1 KiB per wave
// instruction), per-image parameters are wave-uniform (scalar loads), and parameter
// gradients are reduced registers -> 64-lane shuffles -> LDS -> one partial row per workgroup in a caller-provided
// scratch buffer that a second launch adds in index order (bit-repeatable; no float atomics).
//
// Reference arithmetic (codes/models/modules/tools_origin.py): WbQuadratic :317-359,
// GtmManual :414-440; plugin-backed ops follow the build-defined OPSPEC restated in
// oracle/isp_oracle.py (gamma_manual, wb_manual, demosaic_nearest).
#include "risp_common.h"
#include "risp_ops.h"

namespace {

using namespace risp_ops;

// ---------------------------------------------------------------- planar BGR kernels
// grid = (blocks per image, N); each thread walks float4 slots of one image.
template <class Ctx>
__global__ __launch_bounds__(256) void bgr_fwd_kernel(const float *__restrict__ x, const float *__restrict__ p,
                                                      float *__restrict__ y, int hw4) {
    const int n = blockIdx.y;
    const Ctx ctx(p, n);
    const float4 *xb = reinterpret_cast<const float4 *>(x) + (size_t)n * 3 * hw4;
    float4 *yb = reinterpret_cast<float4 *>(y) + (size_t)n * 3 * hw4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw4; i += gridDim.x * blockDim.x) {
        float4 b = xb[i], g = xb[hw4 + i], r = xb[2 * hw4 + i];
        f3 o0 = ctx.fwd({b.x, g.x, r.x}), o1 = ctx.fwd({b.y, g.y, r.y});
        f3 o2 = ctx.fwd({b.z, g.z, r.z}), o3 = ctx.fwd({b.w, g.w, r.w});
        yb[i] = make_float4(o0.b, o1.b, o2.b, o3.b);
        yb[hw4 + i] = make_float4(o0.g, o1.g, o2.g, o3.g);
        yb[2 * hw4 + i] = make_float4(o0.r, o1.r, o2.r, o3.r);
    }
}

#ifndef RISP_WBQ_WAVES
#define RISP_WBQ_WAVES 2
#endif
#ifndef RISP_WBQ_AHEAD
#define RISP_WBQ_AHEAD 1
#endif
template <class Ctx>
__global__ __launch_bounds__(256, Ctx::NP >= 30 ? RISP_WBQ_WAVES : 1) void bgr_bwd_kernel(const float *__restrict__ x, const float *__restrict__ p,
                                                      const float *__restrict__ gy, float *__restrict__ gx,
                                                      float *__restrict__ gp, int hw4) {
    __shared__ float red[Ctx::NP * 4];
    __shared__ float4 stage[Ctx::NP >= 30 ? (RISP_WBQ_AHEAD + 1) * 6 * 256 : 1];
    const int n = blockIdx.y;
    const size_t base = (size_t)n * 3 * hw4;
    const float4 *xb = reinterpret_cast<const float4 *>(x) + base;
    const float4 *gb = reinterpret_cast<const float4 *>(gy) + base;
    float4 *ob = reinterpret_cast<float4 *>(gx) + base;
    const BgrWalkLds<RISP_WBQ_AHEAD> walk(xb, gb, hw4, stage);           // (used by the quadratic white balance only, see below)
    if constexpr (Ctx::NP >= 30) walk.prime();
    const Ctx ctx(p, n);
    float acc[Ctx::NP];
#pragma unroll
    for (int j = 0; j < Ctx::NP; ++j) acc[j] = 0.f;
    // the six 16-byte loads of the NEXT vector are issued before this one is worked on: the quadratic white balance holds 186
    // registers (30 parameter sums) = two waves per SIMD, too few to cover a load round trip per iteration by occupancy alone
    // (64 x 256 x 256: 63 us = 0.30 of the HBM rate with the loads at the top of the loop)
    const int step = gridDim.x * blockDim.x;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (Ctx::NP >= 30) {
        // ... and for it through LDS-DMA, RISP_WBQ_AHEAD vectors ahead (bgr_walk_lds, risp_common.h: the compiler sinks register
        // prefetches to their first use)
        walk.run([&](const BgrVec6 &c, int i) {
            f3 o0 = ctx.bwd({c.b.x, c.g.x, c.r.x}, {c.db.x, c.dg.x, c.dr.x}, acc);
            f3 o1 = ctx.bwd({c.b.y, c.g.y, c.r.y}, {c.db.y, c.dg.y, c.dr.y}, acc);
            f3 o2 = ctx.bwd({c.b.z, c.g.z, c.r.z}, {c.db.z, c.dg.z, c.dr.z}, acc);
            f3 o3 = ctx.bwd({c.b.w, c.g.w, c.r.w}, {c.db.w, c.dg.w, c.dr.w}, acc);
            ob[i] = make_float4(o0.b, o1.b, o2.b, o3.b);
            ob[hw4 + i] = make_float4(o0.g, o1.g, o2.g, o3.g);
            ob[2 * hw4 + i] = make_float4(o0.r, o1.r, o2.r, o3.r);
        });
    } else {
        float4 nb, ng, nr, ndb, ndg, ndr;
        if (i < hw4) {
            nb = xb[i]; ng = xb[hw4 + i]; nr = xb[2 * hw4 + i];
            ndb = gb[i]; ndg = gb[hw4 + i]; ndr = gb[2 * hw4 + i];
        }
        for (; i < hw4; i += step) {
            const float4 b = nb, g = ng, r = nr, db = ndb, dg = ndg, dr = ndr;
            const int j = i + step;
            if (j < hw4) {
                nb = xb[j]; ng = xb[hw4 + j]; nr = xb[2 * hw4 + j];
                ndb = gb[j]; ndg = gb[hw4 + j]; ndr = gb[2 * hw4 + j];
            }
            f3 o0 = ctx.bwd({b.x, g.x, r.x}, {db.x, dg.x, dr.x}, acc);
            f3 o1 = ctx.bwd({b.y, g.y, r.y}, {db.y, dg.y, dr.y}, acc);
            f3 o2 = ctx.bwd({b.z, g.z, r.z}, {db.z, dg.z, dr.z}, acc);
            f3 o3 = ctx.bwd({b.w, g.w, r.w}, {db.w, dg.w, dr.w}, acc);
            ob[i] = make_float4(o0.b, o1.b, o2.b, o3.b);
            ob[hw4 + i] = make_float4(o0.g, o1.g, o2.g, o3.g);
            ob[2 * hw4 + i] = make_float4(o0.r, o1.r, o2.r, o3.r);
        }
    }
    // one partial row per workgroup; param_finish_kernel adds them in index order
    if constexpr (Ctx::NP >= 30) {
        const float t = block_sum_dpp_lanes<Ctx::NP>(acc, red);
        if (threadIdx.x < Ctx::NP) gp[((size_t)n * gridDim.x + blockIdx.x) * Ctx::NP + threadIdx.x] = t * Ctx::pscale(threadIdx.x);
    } else {
        block_sum<Ctx::NP>(acc, red);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int j = 0; j < Ctx::NP; ++j) gp[((size_t)n * gridDim.x + blockIdx.x) * Ctx::NP + j] = acc[j] * Ctx::pscale(j);
        }
    }
}

// gp[row][j] = sum of the partial rows that map to `row` (Ctx::prow: the image itself, or row 0 for the whole batch).
// One wave per output element: lane l adds the partial rows l, l + 64, ... of the element in index order, a fixed
// shuffle tree adds the lanes - bit-repeatable, and no thread walks the whole list alone (the one-thread-per-element
// form took 63-75 us on 64 x 16 rows; this one a few).
template <class Ctx>
__global__ __launch_bounds__(256) void param_finish_kernel(const float *__restrict__ part, float *__restrict__ gp, int N, int bx,
                                                           int rows) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= rows * Ctx::NP) return;
    const int row = t / Ctx::NP, j = t - row * Ctx::NP;
    const bool own = Ctx::prow(1) == 1;                 // per-image rows: only the image's own bx partial rows count
    const int lo = own ? row * bx : 0, hi = own ? (row + 1) * bx : N * bx;
    float s = 0.f;
    for (int i = lo + lane; i < hi; i += 64)
        if (own || Ctx::prow(i / bx) == row) s += part[(size_t)i * Ctx::NP + j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) gp[t] = s;
}

template <class Ctx>
int launch_fwd(const char *name, const float *x, const float *p, float *y, int N, int HW, void *stream) {
    RISP_CHECK_ARG(x && p && y && N > 0 && HW > 0 && HW % 4 == 0, "%s: bad arguments (HW must be a multiple of 4)", name);
    const int hw4 = HW / 4;
    int bx = (hw4 + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(bgr_fwd_kernel<Ctx>, dim3(bx, N), dim3(256), 0, (hipStream_t)stream, x, p, y, hw4);
    RISP_LAUNCH_CHECK(name);
    return 0;
}

template <class Ctx>
int launch_bwd(const char *name, const float *x, const float *p, const float *gy, float *gx, float *gp, float *scratch,
               int N, int HW, void *stream) {
    RISP_CHECK_ARG(x && p && gy && gx && gp && scratch && N > 0 && HW > 0 && HW % 4 == 0, "%s: bad arguments", name);
    const int hw4 = HW / 4, bx = Ctx::NP >= 30 ? risp_bwd_blocks_wbq(N, HW) : risp_bwd_blocks(N, HW);
    hipLaunchKernelGGL(bgr_bwd_kernel<Ctx>, dim3(bx, N), dim3(256), 0, (hipStream_t)stream, x, p, gy, gx, scratch, hw4);
    const int rows = N;                                // gp is (N, NP): rows no image maps to (GtmManual: all but row 0) get 0
    hipLaunchKernelGGL(param_finish_kernel<Ctx>, dim3((rows * Ctx::NP + 3) / 4), dim3(256), 0, (hipStream_t)stream, scratch,
                       gp, N, bx, rows);
    RISP_LAUNCH_CHECK(name);
    return 0;
}

// ---------------------------------------------------------------- nearest-neighbour demosaic
// One thread = QW horizontally adjacent 2x2 quads (QW=2: float4 rows).
template <int QW>
struct RowVec;
template <>
struct RowVec<2> { using T = float4; };
template <>
struct RowVec<1> { using T = float2; };

template <int QW>
__device__ __forceinline__ void load_row(const float *p, float *v) {
    typename RowVec<QW>::T t = *reinterpret_cast<const typename RowVec<QW>::T *>(p);
    const float *s = reinterpret_cast<const float *>(&t);
#pragma unroll
    for (int i = 0; i < 2 * QW; ++i) v[i] = s[i];
}
template <int QW>
__device__ __forceinline__ void store_row(float *p, const float *v) {
    typename RowVec<QW>::T t;
    float *s = reinterpret_cast<float *>(&t);
#pragma unroll
    for (int i = 0; i < 2 * QW; ++i) s[i] = v[i];
    *reinterpret_cast<typename RowVec<QW>::T *>(p) = t;
}
// the same row as a streaming (non-temporal) store: for outputs no later kernel of the launch sequence re-reads
template <int QW>
__device__ __forceinline__ void store_row_nt(float *p, const float *v) {
    typedef float vt __attribute__((ext_vector_type(2 * QW)));
    vt t;
#pragma unroll
    for (int i = 0; i < 2 * QW; ++i) t[i] = v[i];
    __builtin_nontemporal_store(t, reinterpret_cast<vt *>(p));
}

// Bayer rows r0 (R G1 R G1 ..), r1 (G2 B G2 B ..) -> BGR pixels px[row][col]
template <int QW>
__device__ __forceinline__ void demosaic_quads(const float *r0, const float *r1, f3 (*px)[2 * QW]) {
#pragma unroll
    for (int q = 0; q < QW; ++q) {
        float R = r0[2 * q], G1 = r0[2 * q + 1], G2 = r1[2 * q], B = r1[2 * q + 1];
        px[0][2 * q] = px[0][2 * q + 1] = {B, G1, R};
        px[1][2 * q] = px[1][2 * q + 1] = {B, G2, R};
    }
}

template <int QW>
__global__ __launch_bounds__(256) void demosaic_nearest_bwd_kernel(const float *__restrict__ g, float *__restrict__ gb,
                                                                   int H, int W) {
    const int n = blockIdx.y, wq = W / (2 * QW), hq = H / 2;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= wq * hq) return;
    const int qy = t / wq, qx = t - qy * wq;
    const size_t plane = (size_t)H * W;
    const size_t off = (size_t)(2 * qy) * W + (size_t)qx * 2 * QW;
    float v[3][2][2 * QW];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 2; ++r) load_row<QW>(g + ((size_t)n * 3 + c) * plane + off + (size_t)r * W, v[c][r]);
    float o0[2 * QW], o1[2 * QW];
#pragma unroll
    for (int q = 0; q < QW; ++q) {
        const int a = 2 * q, b = 2 * q + 1;
        o0[a] = (v[2][0][a] + v[2][0][b]) + (v[2][1][a] + v[2][1][b]);  // R <- all four red grads
        o0[b] = v[1][0][a] + v[1][0][b];                                // G1 <- even-row greens
        o1[a] = v[1][1][a] + v[1][1][b];                                // G2 <- odd-row greens
        o1[b] = (v[0][0][a] + v[0][0][b]) + (v[0][1][a] + v[0][1][b]);  // B
    }
    store_row<QW>(gb + (size_t)n * plane + off, o0);
    store_row<QW>(gb + (size_t)n * plane + off + W, o1);
}

// ---------------------------------------------------------------- fused element-wise chain
struct ChainArgs {
    const float *in;
    int n_ops, N, H, W;
    int last_out;               // index of the last stage that stores an output
    int ops[RISP_MAX_CHAIN];
    const float *params[RISP_MAX_CHAIN];
    float *outs[RISP_MAX_CHAIN];
};

template <int QW, bool WBQ>
__global__ __launch_bounds__(256) void chain_kernel(const ChainArgs a) {
    const int n = blockIdx.y, W = a.W, H = a.H, wq = W / (2 * QW);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= wq * (H / 2)) return;
    const int qy = t / wq, qx = t - qy * wq;
    const size_t plane = (size_t)H * W;
    const size_t off = (size_t)(2 * qy) * W + (size_t)qx * 2 * QW;
    f3 px[2][2 * QW];
    int k0 = 0;
    if (a.ops[0] == RISP_OP_DEMOSAIC_NEAREST) {
        float r0[2 * QW], r1[2 * QW];
        load_row<QW>(a.in + (size_t)n * plane + off, r0);
        load_row<QW>(a.in + (size_t)n * plane + off + W, r1);
        demosaic_quads<QW>(r0, r1, px);
    } else {
        float v[3][2][2 * QW];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 2; ++r) load_row<QW>(a.in + ((size_t)n * 3 + c) * plane + off + (size_t)r * W, v[c][r]);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 2 * QW; ++i) px[r][i] = {v[0][r][i], v[1][r][i], v[2][r][i]};
    }
    for (int k = k0; k < a.n_ops; ++k) {
        const int op = a.ops[k];
        const float *p = a.params[k];
        apply_op<4 * QW, WBQ>(op, p, n, &px[0][0]);
        float *o = a.outs[k];
        if (o == nullptr) continue;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float vb[2 * QW], vg[2 * QW], vr[2 * QW];
#pragma unroll
            for (int i = 0; i < 2 * QW; ++i) {
                vb[i] = px[r][i].b;
                vg[i] = px[r][i].g;
                vr[i] = px[r][i].r;
            }
            float *ob = o + ((size_t)n * 3 + 0) * plane + off + (size_t)r * W;
            if (k < a.last_out) {       // an intermediate stage output: nothing downstream reads it soon - stream it
                store_row_nt<QW>(ob, vb);
                store_row_nt<QW>(ob + plane, vg);
                store_row_nt<QW>(ob + 2 * plane, vr);
            } else {                    // the segment's result feeds the next launch: default cache policy
                store_row<QW>(ob, vb);
                store_row<QW>(ob + plane, vg);
                store_row<QW>(ob + 2 * plane, vr);
            }
        }
    }
}

}  // namespace

// ---------------------------------------------------------------- C ABI
extern "C" {

int risp_wb_manual_fwd(const float *x, const float *p, float *y, int N, int HW, void *s) {
    return launch_fwd<WbManualCtx>("risp_wb_manual_fwd", x, p, y, N, HW, s);
}
int risp_wb_manual_bwd(const float *x, const float *p, const float *gy, float *gx, float *gp, float *scratch, int N, int HW,
                        void *s) {
    return launch_bwd<WbManualCtx>("risp_wb_manual_bwd", x, p, gy, gx, gp, scratch, N, HW, s);
}
int risp_gamma_fwd(const float *x, const float *p, float *y, int N, int HW, void *s) {
    return launch_fwd<GammaCtx>("risp_gamma_fwd", x, p, y, N, HW, s);
}
int risp_gamma_bwd(const float *x, const float *p, const float *gy, float *gx, float *gp, float *scratch, int N, int HW,
                    void *s) {
    return launch_bwd<GammaCtx>("risp_gamma_bwd", x, p, gy, gx, gp, scratch, N, HW, s);
}
int risp_gtm_manual_fwd(const float *x, const float *p, float *y, int N, int HW, void *s) {
    return launch_fwd<GtmCtx>("risp_gtm_manual_fwd", x, p, y, N, HW, s);
}
int risp_gtm_manual_bwd(const float *x, const float *p, const float *gy, float *gx, float *gp, float *scratch, int N, int HW,
                         void *s) {
    return launch_bwd<GtmCtx>("risp_gtm_manual_bwd", x, p, gy, gx, gp, scratch, N, HW, s);
}
int risp_wb_quadratic_fwd(const float *x, const float *p, float *y, int N, int HW, void *s) {
    return launch_fwd<WbqCtx>("risp_wb_quadratic_fwd", x, p, y, N, HW, s);
}
int risp_wb_quadratic_bwd(const float *x, const float *p, const float *gy, float *gx, float *gp, float *scratch, int N, int HW,
                           void *s) {
    return launch_bwd<WbqCtx>("risp_wb_quadratic_bwd", x, p, gy, gx, gp, scratch, N, HW, s);
}

int risp_gain3_fwd(const float *x, const float *k, float *y, int N, int HW, void *s) {
    return launch_fwd<Gain3Ctx>("risp_gain3_fwd", x, k, y, N, HW, s);
}
int risp_gain3_bwd(const float *x, const float *k, const float *gy, float *gx, float *gk, float *scratch, int N, int HW,
                   void *s) {
    return launch_bwd<Gain3Ctx>("risp_gain3_bwd", x, k, gy, gx, gk, scratch, N, HW, s);
}

size_t risp_param_grad_scratch_floats(int N) { return (size_t)(N > 0 ? N : 0) * 64 * 30; }     // risp_bwd_blocks <= 64 rows of <= 30

static int chain_launch(const ChainArgs &a, void *stream) {
    const bool wide = (a.W % 4 == 0);
    const int nq = (a.W / (wide ? 4 : 2)) * (a.H / 2);
    dim3 grid((nq + 255) / 256, a.N), block(256);
    bool wbq = false;
    for (int k = 0; k < a.n_ops; ++k) wbq |= a.ops[k] == RISP_OP_WB_QUADRATIC;
#ifdef RISP_CHAIN_ALWAYS_WBQ
    wbq = true;
#endif
    if (wide && wbq)
        hipLaunchKernelGGL((chain_kernel<2, true>), grid, block, 0, (hipStream_t)stream, a);
    else if (wide)
        hipLaunchKernelGGL((chain_kernel<2, false>), grid, block, 0, (hipStream_t)stream, a);
    else if (wbq)
        hipLaunchKernelGGL((chain_kernel<1, true>), grid, block, 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((chain_kernel<1, false>), grid, block, 0, (hipStream_t)stream, a);
    RISP_LAUNCH_CHECK("risp_chain_fwd");
    return 0;
}

int risp_chain_fwd(const float *in, int n_ops, const int *ops, const float *const *params, float *const *outs, int N,
                   int H, int W, void *stream) {
    RISP_CHECK_ARG(in && ops && params && outs, "risp_chain_fwd: null argument");
    RISP_CHECK_ARG(n_ops >= 1 && n_ops <= RISP_MAX_CHAIN, "risp_chain_fwd: n_ops %d out of range", n_ops);
    RISP_CHECK_ARG(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && N <= 65535,
                   "risp_chain_fwd: bad shape N=%d H=%d W=%d (H, W must be even)", N, H, W);
    ChainArgs a;
    a.in = in;
    a.n_ops = n_ops;
    a.last_out = -1;
    a.N = N;
    a.H = H;
    a.W = W;
    for (int k = 0; k < RISP_MAX_CHAIN; ++k) {
        a.ops[k] = RISP_OP_SKIP;
        a.params[k] = nullptr;
        a.outs[k] = nullptr;
    }
    for (int k = 0; k < n_ops; ++k) {
        const int op = ops[k];
        RISP_CHECK_ARG(op >= RISP_OP_SKIP && op <= RISP_OP_GAIN3, "risp_chain_fwd: unknown op %d", op);
        RISP_CHECK_ARG(op != RISP_OP_DEMOSAIC_NEAREST || k == 0, "risp_chain_fwd: demosaic must be the first op");
        const bool needs_p = op >= RISP_OP_WB_MANUAL;
        RISP_CHECK_ARG(!needs_p || params[k], "risp_chain_fwd: op %d at stage %d needs params", op, k);
        RISP_CHECK_ARG(op == RISP_OP_SKIP || outs[k], "risp_chain_fwd: stage %d needs an output buffer", k);
        a.ops[k] = op;
        a.params[k] = params[k];
        a.outs[k] = (op == RISP_OP_SKIP) ? nullptr : outs[k];
        if (a.outs[k]) a.last_out = k;
    }
#ifdef RISP_NT_ALL
    a.last_out = RISP_MAX_CHAIN;
#endif
    return chain_launch(a, stream);
}

int risp_demosaic_nearest_fwd(const float *bayer, float *bgr, int N, int H, int W, void *stream) {
    const int op = RISP_OP_DEMOSAIC_NEAREST;
    const float *p = nullptr;
    return risp_chain_fwd(bayer, 1, &op, &p, &bgr, N, H, W, stream);
}

int risp_demosaic_nearest_bwd(const float *g_bgr, float *g_bayer, int N, int H, int W, void *stream) {
    RISP_CHECK_ARG(g_bgr && g_bayer && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && N <= 65535,
                   "risp_demosaic_nearest_bwd: bad arguments");
    const bool wide = (W % 4 == 0);
    const int nq = (W / (wide ? 4 : 2)) * (H / 2);
    dim3 grid((nq + 255) / 256, N), block(256);
    if (wide)
        hipLaunchKernelGGL(demosaic_nearest_bwd_kernel<2>, grid, block, 0, (hipStream_t)stream, g_bgr, g_bayer, H, W);
    else
        hipLaunchKernelGGL(demosaic_nearest_bwd_kernel<1>, grid, block, 0, (hipStream_t)stream, g_bgr, g_bayer, H, W);
    RISP_LAUNCH_CHECK("risp_demosaic_nearest_bwd");
    return 0;
}

}  // extern "C"
