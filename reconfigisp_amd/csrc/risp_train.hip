// One training step of an element-wise fixed pipeline in two launches (gfx950):
//
//   IspModel.optimize_parameters (codes/models/isp_model.py:128-142):
//       output = netG(img); l_pix = cri_pix(output, gt); zero_grad(); l_pix.backward(); optimizer_G.step()
//   with netG = [nearest demosaic ->] a chain of WbManual / Gamma / GtmManual / WbQuadratic stages
//   (isp_universal.py:210-232: per stage sigmoid(raw).repeat(N,1), then the op), cri_pix = MSE or L1 (mean) and
//   optimizer_G = Adam.
//
// chain_train_kernel: a thread owns two horizontally adjacent pixels.  It runs the stages forward in registers keeping every stage's
// input (<= 6 stages x 2 pixels x 3 channels), stores the pipeline output, forms d loss / d output of its pixels and
// walks the stages backwards with the same per-pixel backward maps as the stand-alone kernels (risp_ops.h), adding
// the parameter gradients and the loss into per-thread accumulators.  Nothing but the input, the ground truth and
// the output touches HBM: 28 B/pixel from a mosaic (4 + 12 read, 12 written) against ~150 B/pixel for the
// op-by-op autograd graph.  Accumulators are reduced lanes -> waves -> one row per workgroup in the scratch buffer.
// train_finish_kernel: one wave per parameter adds the rows in a fixed order (bit-repeatable), applies the chain
// rule through gain = 5 * sigmoid(raw) / sigmoid(raw) and .repeat (sum over the images), performs the Adam update
// in place on (raw, exp_avg, exp_avg_sq) exactly as torch.optim.Adam writes it, and rebuilds the per-image
// parameter blocks for the next step.  The loss leaves as one device float.
#include "risp_common.h"
#include "risp_ops.h"

namespace {

using namespace risp_ops;

constexpr int MT = RISP_MAX_TRAIN_CHAIN;
constexpr int ROW = MT * 3 + 30 + 1;       // [stage][3] small-op gradients | 30 WbQuadratic gradients | loss
constexpr int BX_MAX = 32;

__device__ __forceinline__ int op_np(int op) {
    return op == RISP_OP_WB_QUADRATIC ? 30 : (op == RISP_OP_GAMMA ? 1 : 3);
}

template <bool WBQ>
__global__ __launch_bounds__(256) void chain_train_kernel(const risp_train_desc a, const float inv_count) {
    // accumulators of THIS instantiation: without a WbQuadratic stage the 30 quadratic slots are not carried (and not
    // reduced); the loss sits last either way, and the scratch row keeps the common ROW layout
    constexpr int NR = WBQ ? ROW : MT * 3 + 1;
    __shared__ float red[NR * 4];
    const int n = blockIdx.y, H = a.H, W = a.W, wq = W / 2, nq = wq * (H / 2);
    const size_t plane = (size_t)H * W;
    float acc[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[j] = 0.f;
    const float gscale = a.loss_kind == 0 ? 2.f * inv_count : inv_count;

    // The per-image op contexts (gains, gamma and its toe slope, tone-curve knots and slopes, the 30 quadratic
    // coefficients) are wave-uniform: the small ones are built ONCE before the pixel loop and pinned in scalar registers
    // (v_readfirstlane).  Built inside the loop - as the one-shot inference kernels do - they cost two constructions
    // per stage and iteration (30 loads + 30 fmas, four divisions, an exp2) and their share of the vector registers.
    auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    float sctx[MT][8];
#pragma unroll
    for (int k = 0; k < MT; ++k) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sctx[k][j] = 0.f;
        if (!WBQ && k < a.n_ops) {
            const int op = a.ops[k];
            if (op == RISP_OP_WB_MANUAL) {
                const WbManualCtx c(a.blocks[k], n);
#pragma unroll
                for (int j = 0; j < 3; ++j) sctx[k][j] = uni(c.k[j]);
            } else if (op == RISP_OP_GAMMA) {
                const GammaCtx c(a.blocks[k], n);
                sctx[k][0] = uni(c.g); sctx[k][1] = uni(c.toe);
            } else if (op == RISP_OP_GTM_MANUAL) {
                const GtmCtx c(a.blocks[k], n);
#pragma unroll
                for (int j = 0; j < 4; ++j) { sctx[k][j] = uni(c.ys[j]); sctx[k][4 + j] = uni(c.sl[j]); }
            }
        }
    }
    // The instantiation with a WbQuadratic stage (30 coefficients, 30 more gradient accumulators) sits at its register
    // budget: with pinned contexts it falls to one wave per SIMD (0.144 -> 0.20 ms per step), so it keeps the
    // per-use constructions; pipelines without WbQuadratic gain a third (0.148 -> 0.098 ms).
    auto wbm = [&](int k) {
        if (WBQ) return WbManualCtx(a.blocks[k], n);
        WbManualCtx c; c.k[0] = sctx[k][0]; c.k[1] = sctx[k][1]; c.k[2] = sctx[k][2]; return c;
    };
    auto gam = [&](int k) {
        if (WBQ) return GammaCtx(a.blocks[k], n);
        GammaCtx c; c.g = sctx[k][0]; c.toe = sctx[k][1]; return c;
    };
    auto gtm = [&](int k) {
        if (WBQ) return GtmCtx(a.blocks[k], n);
        GtmCtx c;
#pragma unroll
        for (int j = 0; j < 4; ++j) { c.ys[j] = sctx[k][j]; c.sl[j] = sctx[k][4 + j]; }
        return c;
    };
    auto wbq = [&](int k) { return WbqCtx(a.blocks[k], n); };

    // a thread owns ONE ROW of a 2x2 quad (2 pixels): t -> (quad row qy, quad column qx, row r of the quad)
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < 2 * nq; t += gridDim.x * blockDim.x) {
        const int hy = t / wq, qx = t - hy * wq;              // hy = image row
        const size_t off = (size_t)hy * W + 2 * qx;
        f3 px[2];
        if (a.from_bayer) {                                   // OPSPEC nearest demosaic: R, B of the quad, G of the own row
            const size_t q0 = (size_t)(hy & ~1) * W + 2 * qx;
            const float2 r0 = *reinterpret_cast<const float2 *>(a.in + (size_t)n * plane + q0);
            const float2 r1 = *reinterpret_cast<const float2 *>(a.in + (size_t)n * plane + q0 + W);
            px[0] = px[1] = {r1.y, (hy & 1) ? r1.x : r0.y, r0.x};
        } else {
            float2 v[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = *reinterpret_cast<const float2 *>(a.in + ((size_t)n * 3 + c) * plane + off);
            px[0] = {v[0].x, v[1].x, v[2].x};
            px[1] = {v[0].y, v[1].y, v[2].y};
        }
        // ---- forward, keeping each stage's input
        f3 xs[MT][2];
#pragma unroll
        for (int k = 0; k < MT; ++k) {
            if (k < a.n_ops) {
                xs[k][0] = px[0];
                xs[k][1] = px[1];
                const int op = a.ops[k];
                if (op == RISP_OP_WB_MANUAL) {
                    const WbManualCtx c = wbm(k);
                    px[0] = c.fwd(px[0]); px[1] = c.fwd(px[1]);
                } else if (op == RISP_OP_GAMMA) {
                    const GammaCtx c = gam(k);
                    px[0] = c.fwd(px[0]); px[1] = c.fwd(px[1]);
                } else if (op == RISP_OP_GTM_MANUAL) {
                    const GtmCtx c = gtm(k);
                    px[0] = c.fwd(px[0]); px[1] = c.fwd(px[1]);
                } else if (WBQ && op == RISP_OP_WB_QUADRATIC) {
                    const WbqCtx c = wbq(k);
                    px[0] = c.fwd(px[0]); px[1] = c.fwd(px[1]);
                }
            }
        }
        float2 gt[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) gt[c] = *reinterpret_cast<const float2 *>(a.gt + ((size_t)n * 3 + c) * plane + off);
        if (a.y) {
            float *o = a.y + (size_t)n * 3 * plane + off;
            *reinterpret_cast<float2 *>(o) = make_float2(px[0].b, px[1].b);
            *reinterpret_cast<float2 *>(o + plane) = make_float2(px[0].g, px[1].g);
            *reinterpret_cast<float2 *>(o + 2 * plane) = make_float2(px[0].r, px[1].r);
        }
        // ---- loss and d loss / d output
        f3 g[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float tb = i ? gt[0].y : gt[0].x, tg = i ? gt[1].y : gt[1].x, tr = i ? gt[2].y : gt[2].x;
            const float db = px[i].b - tb, dg = px[i].g - tg, dr = px[i].r - tr;
            if (a.loss_kind == 0) {
                acc[NR - 1] += (db * db + dg * dg) + dr * dr;
                g[i] = {db * gscale, dg * gscale, dr * gscale};
            } else {                                          // nn.L1Loss: sign(d) / count, sign(0) = 0
                acc[NR - 1] += (fabsf(db) + fabsf(dg)) + fabsf(dr);
                auto sg = [gscale](float d) { return d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f); };
                g[i] = {sg(db), sg(dg), sg(dr)};
            }
        }
        // ---- backward through the stages
#pragma unroll
        for (int k = MT - 1; k >= 0; --k) {
            if (k < a.n_ops) {
                const int op = a.ops[k];
                if (op == RISP_OP_WB_MANUAL) {
                    const WbManualCtx c = wbm(k);
#pragma unroll
                    for (int i = 0; i < 2; ++i) g[i] = c.bwd(xs[k][i], g[i], acc + 3 * k);
                } else if (op == RISP_OP_GAMMA) {
                    const GammaCtx c = gam(k);
#pragma unroll
                    for (int i = 0; i < 2; ++i) g[i] = c.bwd(xs[k][i], g[i], acc + 3 * k);
                } else if (op == RISP_OP_GTM_MANUAL) {
                    const GtmCtx c = gtm(k);
#pragma unroll
                    for (int i = 0; i < 2; ++i) g[i] = c.bwd(xs[k][i], g[i], acc + 3 * k);
                } else if (WBQ && op == RISP_OP_WB_QUADRATIC) {
                    const WbqCtx c = wbq(k);
#pragma unroll
                    for (int i = 0; i < 2; ++i) g[i] = c.bwd(xs[k][i], g[i], acc + 3 * MT);
                }
            }
        }
    }
    block_sum<NR>(acc, red);
    if (threadIdx.x == 0) {
        float *row = a.scratch + ((size_t)n * gridDim.x + blockIdx.x) * ROW;
#pragma unroll
        for (int j = 0; j < NR - 1; ++j) row[j] = acc[j];
        row[ROW - 1] = acc[NR - 1];
    }
}

// accurate sigmoid (torch.sigmoid in fp32: 1 / (1 + exp(-x)))
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// grid = one wave (64 threads) per parameter element + one wave for the loss; `rows` partial rows of ROW floats
__global__ __launch_bounds__(64) void train_finish_kernel(const risp_train_desc a, int rows, const float inv_count) {
    const int lane = threadIdx.x;
    int e = blockIdx.x, k = 0, j = 0, slot = ROW - 1;
    bool is_loss = true;
    for (k = 0; k < a.n_ops; ++k) {                  // element e -> (stage k, index j)
        const int np = op_np(a.ops[k]);
        if (e < np) { is_loss = false; j = e; break; }
        e -= np;
    }
    if (!is_loss) slot = a.ops[k] == RISP_OP_WB_QUADRATIC ? 3 * MT + j : 3 * k + j;
    float s = 0.f;
    for (int r = lane; r < rows; r += 64) s += a.scratch[(size_t)r * ROW + slot];
    s = wave_sum(s);                                  // fixed tree: bit-repeatable
    s = __shfl(s, 0, 64);
    if (is_loss) {
        if (lane == 0) a.loss[0] = s * inv_count;
        return;
    }
    const int op = a.ops[k], np = op_np(op);
    const float pre = op == RISP_OP_WB_MANUAL ? 5.f : 1.f;     // gain = 5 * sigmoid(raw)  (tools_origin.py:214)
    const float inner = op == RISP_OP_WB_QUADRATIC ? 10.f : 1.f; // coef = 10 p - 5        (tools_origin.py:340)
    float raw = a.raw[k][j];
    const float sg = sigmoidf(raw);
    const float grad = (s * inner * pre) * ((1.f - sg) * sg);    // .repeat backward = the sum over the images above
    // torch.optim.Adam (single step, no weight decay / amsgrad):
    //   exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    //   denom = exp_avg_sq.sqrt() / sqrt(1 - beta2^t) + eps; param.addcdiv_(exp_avg, denom, value = -lr / (1 - beta1^t))
    float m = a.exp_avg[k][j], v = a.exp_avg_sq[k][j];
    m = __builtin_fmaf(a.one_minus_beta1, grad - m, m);
    v = __builtin_fmaf(a.one_minus_beta2 * grad, grad, v * a.beta2);
    const float denom = __builtin_sqrtf(v) / a.bias2_sqrt + a.eps;
    raw = raw - a.lr_step * (m / denom);
    if (lane == 0) {
        a.grad[k][j] = grad;
        a.exp_avg[k][j] = m;
        a.exp_avg_sq[k][j] = v;
        a.raw[k][j] = raw;
    }
    const float blk = pre * sigmoidf(raw);             // what the next forward reads: sigmoid(raw).repeat(N,1) [* 5]
    for (int n = lane; n < a.N; n += 64) a.blocks[k][(size_t)n * np + j] = blk;
}

}  // namespace

extern "C" {

size_t risp_train_scratch_floats(int N) { return (size_t)(N > 0 ? N : 0) * BX_MAX * ROW; }

int risp_chain_train_step(const risp_train_desc *dp, void *stream) {
    RISP_CHECK_ARG(dp, "risp_chain_train_step: null descriptor");
    risp_train_desc a = *dp;
    RISP_CHECK_ARG(a.in && a.gt && a.loss && a.scratch, "risp_chain_train_step: null tensor");
    RISP_CHECK_ARG(a.N > 0 && a.N <= 65535 && a.H > 0 && a.W > 0 && a.H % 2 == 0 && a.W % 2 == 0,
                   "risp_chain_train_step: bad shape N=%d H=%d W=%d (H, W must be even)", a.N, a.H, a.W);
    RISP_CHECK_ARG(a.n_ops >= 1 && a.n_ops <= MT, "risp_chain_train_step: %d stages (1..%d)", a.n_ops, MT);
    RISP_CHECK_ARG(a.loss_kind == 0 || a.loss_kind == 1, "risp_chain_train_step: loss kind %d", a.loss_kind);
    int total = 0, n_wbq = 0;
    for (int k = 0; k < a.n_ops; ++k) {
        const int op = a.ops[k];
        RISP_CHECK_ARG(op == RISP_OP_WB_MANUAL || op == RISP_OP_GAMMA || op == RISP_OP_GTM_MANUAL || op == RISP_OP_WB_QUADRATIC,
                       "risp_chain_train_step: op %d has no fused training form", op);
        RISP_CHECK_ARG(a.blocks[k] && a.raw[k] && a.grad[k] && a.exp_avg[k] && a.exp_avg_sq[k],
                       "risp_chain_train_step: stage %d incomplete", k);
        n_wbq += op == RISP_OP_WB_QUADRATIC;
        total += op == RISP_OP_WB_QUADRATIC ? 30 : (op == RISP_OP_GAMMA ? 1 : 3);
    }
    RISP_CHECK_ARG(n_wbq <= 1, "risp_chain_train_step: at most one WbQuadratic stage");
    const float inv_count = 1.0f / ((float)a.N * 3.f * (float)a.H * (float)a.W);
    const int nq = (a.W / 2) * (a.H / 2);
    // 16 pixel pairs per thread - the measured optimum of both instantiations (64 x 256 x 256: 8 pairs +12 %, 12 +16 %,
    // 32 +27 % step time)
    int bx = (2 * nq + 256 * 16 - 1) / (256 * 16);
    if (bx < 1) bx = 1;
    if (bx > BX_MAX) bx = BX_MAX;
    hipStream_t s = (hipStream_t)stream;
    if (n_wbq) hipLaunchKernelGGL(chain_train_kernel<true>, dim3(bx, a.N), dim3(256), 0, s, a, inv_count);
    else hipLaunchKernelGGL(chain_train_kernel<false>, dim3(bx, a.N), dim3(256), 0, s, a, inv_count);
    hipLaunchKernelGGL(train_finish_kernel, dim3(total + 1), dim3(64), 0, s, a, a.N * bx, inv_count);
    RISP_LAUNCH_CHECK("risp_chain_train_step");
    return 0;
}

}  // extern "C"
