// Per-image contexts of the element-wise ISP operators (device-side, header-only): forward map,
// backward map and parameter-gradient accumulation for one BGR pixel.  Shared by the plane-stream
// kernels (risp_pointwise.hip) and the fused stencil segment (risp_fused.hip).
//
// Reference arithmetic (codes/models/modules/tools_origin.py): WbQuadratic :317-359, GtmManual :414-440;
// the plugin-backed ops follow the build-defined OPSPEC restated in oracle/isp_oracle.py.
#pragma once
#include "risp_common.h"

namespace risp_ops {

constexpr float kToe = 1.0f / 1024.0f;      // OPSPEC gamma toe (10 bits)
constexpr float kLog2Toe = -10.0f;          // log2(kToe)
constexpr float kLn2 = 0.6931471805599453f;

// ---------------------------------------------------------------- per-image op contexts
struct WbManualCtx {   // p = the per-image gain (N,3), i.e. what tools_origin.py:214 computes as params * 5
    static constexpr int NP = 3;
    float k[3];
    __device__ WbManualCtx() {}                      // members filled by the caller (a context kept across a pixel loop)
    __device__ WbManualCtx(const float *p, int n) {
#pragma unroll
        for (int c = 0; c < 3; ++c) k[c] = p[n * 3 + c];
    }
    __device__ f3 fwd(f3 v) const { return {v.b * k[0], v.g * k[1], v.r * k[2]}; }
    __device__ f3 bwd(f3 x, f3 g, float *acc) const {
        acc[0] += g.b * x.b;
        acc[1] += g.g * x.g;
        acc[2] += g.r * x.r;
        return {g.b * k[0], g.g * k[1], g.r * k[2]};
    }
    __device__ static float pscale(int) { return 1.f; }
    __host__ __device__ static int prow(int n) { return n; }
};

struct GammaCtx {
    static constexpr int NP = 1;
    float g, toe;  // toe = T^(g-1): slope of the linear segment below T
    __device__ GammaCtx() {}
    __device__ GammaCtx(const float *p, int n) {
        g = p[n];
        toe = __builtin_amdgcn_exp2f((g - 1.f) * kLog2Toe);
    }
    __device__ float f(float x) const {
        return x >= kToe ? __builtin_amdgcn_exp2f(g * __builtin_amdgcn_logf(x)) : x * toe;
    }
    __device__ f3 fwd(f3 v) const { return {f(v.b), f(v.g), f(v.r)}; }
    __device__ float b1(float x, float gy, float *acc) const {
        if (x >= kToe) {
            float l2 = __builtin_amdgcn_logf(x);
            float y = __builtin_amdgcn_exp2f(g * l2);
            acc[0] += gy * y * (l2 * kLn2);
            return gy * g * y / x;
        }
        acc[0] += gy * x * toe * (kLog2Toe * kLn2);
        return gy * toe;
    }
    __device__ f3 bwd(f3 x, f3 gy, float *acc) const {
        return {b1(x.b, gy.b, acc), b1(x.g, gy.g, acc), b1(x.r, gy.r, acc)};
    }
    __device__ static float pscale(int) { return 1.f; }
    __host__ __device__ static int prow(int n) { return n; }
};

struct GtmCtx {  // 4 segments; knots from row 0 of p only (tools_origin.py:423)
    static constexpr int NP = 3;
    float ys[4], sl[4];
    __device__ GtmCtx() {}
    __device__ GtmCtx(const float *p, int) {
        float k[5] = {0.f, p[0], p[1], p[2], 1.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            ys[s] = k[s];
            sl[s] = (k[s + 1] - k[s]) / 0.25f;
        }
    }
    __device__ float pre(float x, int &seg, float &slope) const {
        // half-open segments [k/4,(k+1)/4); anything else (x<0, x>=1, NaN) passes through
        seg = -1;
        slope = 1.f;
        float o = x;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float xs = 0.25f * s, xe = 0.25f * (s + 1);
            if (x >= xs && x < xe) {
                o = (x - xs) * sl[s] + ys[s];
                seg = s;
                slope = sl[s];
            }
        }
        return o;
    }
    __device__ float f(float x) const {
        int s;
        float m;
        return clamp01(pre(x, s, m));
    }
    __device__ f3 fwd(f3 v) const { return {f(v.b), f(v.g), f(v.r)}; }
    __device__ float b1(float x, float gy, float *acc) const {
        int s;
        float m;
        float o = pre(x, s, m);
        float g = gy * gate01(o);
        float t = (x - 0.25f * s) * 4.f;  // d out / d y_end ; (1-t) = d out / d y_start
#pragma unroll
        for (int j = 0; j < 3; ++j)     // knot j is the end of segment j and the start of segment j+1
            acc[j] += (s == j ? g * t : 0.f) + (s == j + 1 ? g * (1.f - t) : 0.f);
        return g * m;
    }
    __device__ f3 bwd(f3 x, f3 gy, float *acc) const {
        return {b1(x.b, gy.b, acc), b1(x.g, gy.g, acc), b1(x.r, gy.r, acc)};
    }
    __device__ static float pscale(int) { return 1.f; }
    __host__ __device__ static int prow(int) { return 0; }
};

struct WbqCtx {  // coef[ch][j] = 10 p[10ch+j] - 5 ; features B2 G2 R2 BG BR GR B G R 1
    static constexpr int NP = 30;
    float c[3][10];
    __device__ WbqCtx() {}
    __device__ WbqCtx(const float *p, int n) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                // (wave-uniform, but formed by vector arithmetic - gfx9 has no scalar float unit - and therefore kept in vector
                // registers: moved to scalar ones by hand, the 30 vector registers go to the kernels' prefetch depth and occupancy.
                // The empty asm hides the uniformity from the compiler, which would fold the readfirstlane away; the instruction
                // itself is the compiler's, with the wait states it needs)
                float v = p[n * 30 + ch * 10 + j] * 10.f - 5.f;
                asm("" : "+v"(v));
                c[ch][j] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
            }
    }
    __device__ float pre(const float *f, int ch) const {
        float s = f[0] * c[ch][0];
#pragma unroll
        for (int j = 1; j < 10; ++j) s += f[j] * c[ch][j];
        return s;
    }
    __device__ static void feats(f3 v, float *f) {
        f[0] = v.b * v.b; f[1] = v.g * v.g; f[2] = v.r * v.r;
        f[3] = v.b * v.g; f[4] = v.b * v.r; f[5] = v.g * v.r;
        f[6] = v.b; f[7] = v.g; f[8] = v.r; f[9] = 1.f;
    }
    __device__ f3 fwd(f3 v) const {
        float f[10];
        feats(v, f);
        return {clamp01(pre(f, 0)), clamp01(pre(f, 1)), clamp01(pre(f, 2))};
    }
    __device__ f3 bwd(f3 x, f3 gy, float *acc) const {
        float f[10];
        feats(x, f);
        float g[3] = {gy.b * gate01(pre(f, 0)), gy.g * gate01(pre(f, 1)), gy.r * gate01(pre(f, 2))};
        f3 o = {0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
#pragma unroll
            for (int j = 0; j < 10; ++j) acc[ch * 10 + j] = __builtin_fmaf(g[ch], f[j], acc[ch * 10 + j]);      // one instruction per term
            o.b += g[ch] * (2.f * x.b * c[ch][0] + x.g * c[ch][3] + x.r * c[ch][4] + c[ch][6]);
            o.g += g[ch] * (2.f * x.g * c[ch][1] + x.b * c[ch][3] + x.r * c[ch][5] + c[ch][7]);
            o.r += g[ch] * (2.f * x.r * c[ch][2] + x.b * c[ch][4] + x.g * c[ch][5] + c[ch][8]);
        }
        return o;
    }
    // the input gradient alone (the same operations in the same order as bwd, without the 30 parameter sums)
    __device__ f3 bwd_gx(f3 x, f3 gy) const {
        float f[10];
        feats(x, f);
        float g[3] = {gy.b * gate01(pre(f, 0)), gy.g * gate01(pre(f, 1)), gy.r * gate01(pre(f, 2))};
        f3 o = {0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            o.b += g[ch] * (2.f * x.b * c[ch][0] + x.g * c[ch][3] + x.r * c[ch][4] + c[ch][6]);
            o.g += g[ch] * (2.f * x.g * c[ch][1] + x.b * c[ch][3] + x.r * c[ch][5] + c[ch][7]);
            o.r += g[ch] * (2.f * x.r * c[ch][2] + x.b * c[ch][4] + x.g * c[ch][5] + c[ch][8]);
        }
        return o;
    }
    // ... and the parameter sums alone
    __device__ void bwd_gp(f3 x, f3 gy, float *acc) const {
        float f[10];
        feats(x, f);
        float g[3] = {gy.b * gate01(pre(f, 0)), gy.g * gate01(pre(f, 1)), gy.r * gate01(pre(f, 2))};
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
#pragma unroll
            for (int j = 0; j < 10; ++j) acc[ch * 10 + j] = __builtin_fmaf(g[ch], f[j], acc[ch * 10 + j]);
    }
    __device__ static float pscale(int) { return 10.f; }
    __host__ __device__ static int prow(int n) { return n; }
};

struct Gain3Ctx {  // y_c = clamp(x_c * p[n,c]) : gray-world apply with precomputed gains
    static constexpr int NP = 3;
    float k[3];
    __device__ Gain3Ctx(const float *p, int n) {
#pragma unroll
        for (int c = 0; c < 3; ++c) k[c] = p[n * 3 + c];
    }
    __device__ f3 fwd(f3 v) const { return {clamp01(v.b * k[0]), clamp01(v.g * k[1]), clamp01(v.r * k[2])}; }
    __device__ f3 bwd(f3 x, f3 g, float *acc) const {
        g.b *= gate01(x.b * k[0]);
        g.g *= gate01(x.g * k[1]);
        g.r *= gate01(x.r * k[2]);
        acc[0] += g.b * x.b;
        acc[1] += g.g * x.g;
        acc[2] += g.r * x.r;
        return {g.b * k[0], g.g * k[1], g.r * k[2]};
    }
    __device__ static float pscale(int) { return 1.f; }
    __host__ __device__ static int prow(int n) { return n; }
};

template <class Ctx, int NPX>
__device__ __forceinline__ void apply_all(const float *p, int n, f3 *px) {
    const Ctx ctx(p, n);
#pragma unroll
    for (int i = 0; i < NPX; ++i) px[i] = ctx.fwd(px[i]);
}

// one element-wise stage (RISP_OP_*) on NPX pixels held in registers; op is wave-uniform.  WBQ = false compiles the
// 30-coefficient WbQuadratic out: its live range alone sets the register count (and so the occupancy) of a kernel
// that contains it, so launchers pick the lean instantiation whenever the op list has no WbQuadratic.
template <int NPX, bool WBQ = true>
__device__ __forceinline__ void apply_op(int op, const float *p, int n, f3 *px) {
    switch (op) {
        case RISP_OP_WB_MANUAL: apply_all<WbManualCtx, NPX>(p, n, px); break;
        case RISP_OP_GAMMA: apply_all<GammaCtx, NPX>(p, n, px); break;
        case RISP_OP_GTM_MANUAL: apply_all<GtmCtx, NPX>(p, n, px); break;
        case RISP_OP_WB_QUADRATIC:
            if constexpr (WBQ) apply_all<WbqCtx, NPX>(p, n, px);
            break;
        case RISP_OP_GAIN3: apply_all<Gain3Ctx, NPX>(p, n, px); break;
        default: break;  // SKIP, DEMOSAIC_NEAREST (applied at load)
    }
}

}  // namespace risp_ops
