// Fused stencil segment of a fixed pipeline (inference):
//
//     [nearest demosaic] -> bilateral denoise -> element-wise chain (WB / gamma / tone curve ...)
//
// in ONE launch that reads the segment input once and still writes every stage output
// (intermediate_results is API: test.py:74 consumes every stage).  This is the literal ISP ordering
// of the headline benchmark (demosaic -> denoise -> white balance -> gamma -> tone map).
//
// Per workgroup (256 threads): a 64 x 16 output tile.  The BGR halo tile (reflect-101, radius = window/2)
// is staged in LDS - straight from the Bayer mosaic when the segment starts with the nearest-neighbour
// demosaic (index map only) - scaled to the 0..255 domain the classical bilateral is defined in.  Each
// thread then owns 4 consecutive pixels: bilateral from LDS, 8-bit rounding, back to [0,1], the
// element-wise stages in registers, and 16-byte stores of every stage's planes.
//
// Arithmetic: oracle/isp_oracle.py origin_denoise('bilateral') (build-defined OPSPEC, parity unpinned;
// call site tools_origin.py:686-710) and the element-wise contexts of risp_ops.h.
#include "risp_common.h"
#include "risp_ops.h"

namespace {

using namespace risp_ops;

constexpr int FX = 64, FY = 16, PXT = 4;   // tile and pixels per thread (FX/PXT * FY = 256 threads)

struct FusedArgs {
    const float *in;            // (N,1,H,W) mosaic if from_bayer else (N,3,H,W)
    float *out_dem;             // (N,3,H,W) demosaic output (from_bayer only)
    float *out_bil;             // (N,3,H,W) bilateral output
    const int *win;             // (N) odd window per image
    const float *sig_c, *sig_s; // (N)
    int n_ops, N, H, W, R;
    int last_out;               // index of the last chain stage that stores an output; -1 = the bilateral is the result
    int ops[RISP_MAX_CHAIN];
    const float *params[RISP_MAX_CHAIN];
    float *outs[RISP_MAX_CHAIN];
};

__device__ __forceinline__ int refl(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_nt(float *p, v4f v) {
#ifndef RISP_STREAM_POLICY
#define RISP_STREAM_POLICY "nt"            // tools/ab_fused.py: "sc1", "sc0 sc1", "sc1 nt" measured against it
#endif
    asm volatile("global_store_dwordx4 %0, %1, off " RISP_STREAM_POLICY ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ float q8f(float v) {
    return floorf(__builtin_amdgcn_fmed3f(v, 0.f, 255.f) + 0.5f);   // clamp in one instruction (v is never NaN here)
}

// RT: compile-time halo radius (1 = the 3x3 window every reference configuration produces, because
// `.int()` precedes `*7` at tools_origin.py:698); 0 = run-time radius a.R.
template <bool FROM_BAYER, int RT, bool WBQ>
#ifndef RISP_FUSED_WAVES
#define RISP_FUSED_WAVES 1
#endif
__global__ __launch_bounds__(256, RISP_FUSED_WAVES) void bilateral_chain_kernel(const FusedArgs a) {
    extern __shared__ float lds[];
    const int R = RT > 0 ? RT : a.R, H = a.H, W = a.W;
    const int tw = FX + 2 * R, th = FY + 2 * R, per = tw * th;
#ifndef RISP_FUSED_NO_XCD_MAP
    // XCD-aware tile order: the hardware deals consecutive workgroups round-robin to the 8 XCDs, so tile neighbours -
    // which share a halo ring - land on 8 different L2s.  Remapped, XCD k works through the k-th contiguous eighth of
    // the tile list and the neighbours' halo reads hit its own L2 (tools/ab_fused.py, 2000 launches rotating over 4
    // resident batches: 46.2 -> 45.5 us; -DRISP_FUSED_NO_XCD_MAP restores the plain order for A/B).
    int bxi = blockIdx.x, byi = blockIdx.y, bzi = blockIdx.z;
    {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        if ((total & 7u) == 0) {
            const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            const unsigned t = (lin & 7u) * (total >> 3) + (lin >> 3);
            bxi = t % gridDim.x;
            byi = (t / gridDim.x) % gridDim.y;
            bzi = t / (gridDim.x * gridDim.y);
        }
    }
    const int n = bzi, x0 = bxi * FX, y0 = byi * FY;
#else
    const int n = blockIdx.z, x0 = blockIdx.x * FX, y0 = blockIdx.y * FY;
#endif
    const size_t plane = (size_t)H * W;

    // ---- stage the BGR halo tile (raw [0,1] samples; the x255 of the bilateral's domain is applied on read).
    const int lx = (threadIdx.x & 15) * PXT, ly = threadIdx.x >> 4;
    const int px = x0 + lx, py = y0 + ly;
    if (FROM_BAYER && RT == 1) {
        // Quad staging: one 2x2 mosaic quad (two 8-byte loads) yields the BGR values of its four pixels -
        // R and B shared, G per row - so the index arithmetic is paid once per four tile entries.  Only quads
        // inside the image are loaded; the one-pixel ring outside the image is filled by reflection afterwards.
        constexpr int TWc = FX + 2, THc = FY + 2, PERc = TWc * THc, QW = FX / 2 + 2, QH = FY / 2 + 2;
        const float *bay = a.in + (size_t)n * plane;
        float2 top[2], bot[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int q = threadIdx.x + 256 * it;
            const int j = q / QW, i = q - j * QW;
            const int Y = y0 - 2 + 2 * j, X = x0 - 2 + 2 * i;
            const bool ok = q < QW * QH && Y >= 0 && Y < H && X >= 0 && X < W;
            top[it] = ok ? *reinterpret_cast<const float2 *>(bay + (size_t)Y * W + X) : make_float2(0.f, 0.f);
            bot[it] = ok ? *reinterpret_cast<const float2 *>(bay + (size_t)(Y + 1) * W + X) : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int q = threadIdx.x + 256 * it;
            const int j = q / QW, i = q - j * QW;
            const int Y = y0 - 2 + 2 * j, X = x0 - 2 + 2 * i;
            if (!(q < QW * QH && Y >= 0 && Y < H && X >= 0 && X < W)) continue;
            const float R_ = top[it].x, G1 = top[it].y, G2 = bot[it].x, B_ = bot[it].y;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int ty = 2 * j - 1 + dy, tx = 2 * i - 1 + dx;
                    if (ty >= 0 && ty < THc && tx >= 0 && tx < TWc) {
                        const int idx = ty * TWc + tx;
                        lds[idx] = B_;
                        lds[PERc + idx] = dy ? G2 : G1;
                        lds[2 * PERc + idx] = R_;
                    }
                }
        }
        if (x0 == 0 || y0 == 0 || x0 + FX >= W || y0 + FY >= H) {        // block-uniform: tile touches the image border
            __syncthreads();
            for (int idx = threadIdx.x; idx < PERc; idx += 256) {
                const int ty = idx / TWc, tx = idx - ty * TWc;
                const int gy = y0 - 1 + ty, gx = x0 - 1 + tx;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) continue;
                const int sy = refl(gy, H) - (y0 - 1), sx = refl(gx, W) - (x0 - 1);
                if (sy < 0 || sy >= THc || sx < 0 || sx >= TWc) continue;      // feeds only outputs outside the image
                const int src = sy * TWc + sx;
                lds[idx] = lds[src];
                lds[PERc + idx] = lds[PERc + src];
                lds[2 * PERc + idx] = lds[2 * PERc + src];
            }
        }
    } else if (FROM_BAYER) {
        const float *bay = a.in + (size_t)n * plane;
        for (int idx = threadIdx.x; idx < per; idx += 256) {
            const int ty = idx / tw, tx = idx - ty * tw;
            const int gy = refl(y0 + ty - R, H), gx = refl(x0 + tx - R, W);
            const int qy = gy & ~1, qx = gx & ~1;                       // quad origin
            lds[2 * per + idx] = bay[(size_t)qy * W + qx];
            lds[per + idx] = bay[(size_t)(qy + (gy & 1)) * W + qx + 1 - (gy & 1)];   // G1 on even rows, G2 on odd rows
            lds[idx] = bay[(size_t)(qy + 1) * W + qx + 1];
        }
    } else {
        const float *img = a.in + (size_t)n * 3 * plane;
        auto fetch1 = [&](int idx) {
            const int c = idx / per, rem = idx - c * per;
            const int ty = rem / tw, tx = rem - ty * tw;
            return img[(size_t)c * plane + (size_t)refl(y0 + ty - R, H) * W + refl(x0 + tx - R, W)];
        };
        if (RT > 0) {
            constexpr int PER3 = 3 * (FX + 2 * RT) * (FY + 2 * RT), NIT = (PER3 + 255) / 256;
            float v[NIT];
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int idx = threadIdx.x + 256 * i;
                v[i] = idx < PER3 ? fetch1(idx) : 0.f;
            }
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int idx = threadIdx.x + 256 * i;
                if (idx < PER3) lds[idx] = v[i];
            }
        } else {
            for (int idx = threadIdx.x; idx < 3 * per; idx += 256) lds[idx] = fetch1(idx);
        }
    }
    __syncthreads();

    if (px >= W || py >= H) return;                     // W % 4 == 0: the 4 pixels are in or out together
    const size_t o = (size_t)n * 3 * plane + (size_t)py * W + px;
    const float *ctr = lds + (ly + R) * tw + lx + R;

    if (FROM_BAYER) {                                   // the demosaic stage output: the staged samples themselves
        float4 vb, vg, vr;
        float *eb = reinterpret_cast<float *>(&vb), *eg = reinterpret_cast<float *>(&vg), *er = reinterpret_cast<float *>(&vr);
#pragma unroll
        for (int i = 0; i < PXT; ++i) {
            eb[i] = ctr[i];
            eg[i] = ctr[per + i];
            er[i] = ctr[2 * per + i];
        }
        st4_nt(a.out_dem + o, *reinterpret_cast<v4f *>(&vb));
        st4_nt(a.out_dem + o + plane, *reinterpret_cast<v4f *>(&vg));
        st4_nt(a.out_dem + o + 2 * plane, *reinterpret_cast<v4f *>(&vr));
    }

    // ---- bilateral on 4 pixels
    int r = a.win[n] / 2;
    r = r < 0 ? 0 : (r > R ? R : r);                   // never walk outside the staged halo
    const bool full = RT > 0 && r == RT;               // wave-uniform: unrolled window, LDS reads shared by the 4 pixels
    const float ks = -1.f / (2.f * a.sig_s[n] * a.sig_s[n]), kc = -1.f / (2.f * a.sig_c[n] * a.sig_c[n]);
    const float ks2 = ks * 1.4426950408889634f, kc2 = kc * 1.4426950408889634f;     // base-2 exponent coefficients
    f3 pix[PXT];
#pragma unroll
    for (int i = 0; i < PXT; ++i) {
        const float *c0 = ctr + i;
        const float cb = c0[0] * 255.f, cg = c0[per] * 255.f, cr = c0[2 * per] * 255.f;
        // the centre tap has weight exp(0) = 1 exactly: start from it instead of evaluating it
        float nb = 0.f, ng = 0.f, nr = 0.f, den = 0.f;
        auto tap = [&](int dy, int dx) {
            if (dy == 0 && dx == 0) {
                nb += cb; ng += cg; nr += cr; den += 1.f;
                return;
            }
            const float *q = c0 + dy * tw + dx;
            const float qb = q[0] * 255.f, qg = q[per] * 255.f, qr = q[2 * per] * 255.f;
            const float dist = fabsf(qb - cb) + fabsf(qg - cg) + fabsf(qr - cr);
            // exp(-|d|^2 / 2 sigma_s^2 - dist^2 / 2 sigma_c^2) as one fma feeding v_exp_f32 (log2 e folded into the two
            // coefficients) and fma accumulation: 5 vector instructions per tap fewer than the mul / add form
            // (measured -4 % launch time); bilateral_kernel of risp_origin.hip evaluates the same expressions.
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
        // x (1/255): what `output.float() / 255.` (tools_origin.py:716) evaluates to on the GPU (PyTorch divides by a
        // host scalar through its fp32 reciprocal)
        const float inv255 = 1.f / 255.f, rden = 1.f / den;     // OPSPEC: normalise by one reciprocal, not three divisions
        pix[i] = {q8f(nb * rden) * inv255, q8f(ng * rden) * inv255, q8f(nr * rden) * inv255};
    }
    // Stage outputs that nothing downstream reads soon (every one but the segment's result, which feeds the next
    // launch) are streamed with non-temporal stores: measured -11 % launch time once the working set exceeds the
    // Infinity Cache, unchanged when it fits.
    auto store = [&](float *dst, bool stream) {
        const v4f vb = {pix[0].b, pix[1].b, pix[2].b, pix[3].b}, vg = {pix[0].g, pix[1].g, pix[2].g, pix[3].g},
                  vr = {pix[0].r, pix[1].r, pix[2].r, pix[3].r};
        if (stream) {       // (asm: two branches storing the same value get merged into one plain store otherwise)
            st4_nt(dst + o, vb);
            st4_nt(dst + o + plane, vg);
            st4_nt(dst + o + 2 * plane, vr);
        } else {
            *reinterpret_cast<v4f *>(dst + o) = vb;
            *reinterpret_cast<v4f *>(dst + o + plane) = vg;
            *reinterpret_cast<v4f *>(dst + o + 2 * plane) = vr;
        }
    };
    store(a.out_bil, a.last_out >= 0);

    // ---- element-wise stages
    for (int k = 0; k < a.n_ops; ++k) {
        apply_op<PXT, WBQ>(a.ops[k], a.params[k], n, pix);
        if (a.outs[k]) store(a.outs[k], k < a.last_out);
    }
}

}  // namespace

extern "C" {

int risp_bilateral_chain_fwd(const float *in, int from_bayer, float *out_demosaic, float *out_bilateral,
                             const int32_t *window, const float *sigma_color, const float *sigma_space, int max_window,
                             int n_ops, const int *ops, const float *const *params, float *const *outs, int N, int H,
                             int W, void *stream) {
    RISP_CHECK_ARG(in && out_bilateral && window && sigma_color && sigma_space, "risp_bilateral_chain_fwd: null argument");
    RISP_CHECK_ARG(!from_bayer || out_demosaic, "risp_bilateral_chain_fwd: demosaic output buffer missing");
    RISP_CHECK_ARG(n_ops >= 0 && n_ops <= RISP_MAX_CHAIN && (n_ops == 0 || (ops && params && outs)),
                   "risp_bilateral_chain_fwd: bad op list");
    RISP_CHECK_ARG(max_window >= 1 && max_window <= 17 && (max_window & 1), "risp_bilateral_chain_fwd: window %d", max_window);
    RISP_CHECK_ARG(N > 0 && N <= 65535 && H % 2 == 0 && W % 4 == 0 && H > max_window / 2 && W > max_window / 2,
                   "risp_bilateral_chain_fwd: bad shape N=%d H=%d W=%d (W must be a multiple of 4)", N, H, W);
    FusedArgs a;
    a.in = in;
    a.out_dem = out_demosaic;
    a.out_bil = out_bilateral;
    a.win = window;
    a.sig_c = sigma_color;
    a.sig_s = sigma_space;
    a.n_ops = n_ops;
    a.N = N;
    a.H = H;
    a.W = W;
    a.R = max_window / 2;
    a.last_out = -1;
    for (int k = 0; k < RISP_MAX_CHAIN; ++k) {
        a.ops[k] = RISP_OP_SKIP;
        a.params[k] = nullptr;
        a.outs[k] = nullptr;
    }
    for (int k = 0; k < n_ops; ++k) {
        RISP_CHECK_ARG(ops[k] == RISP_OP_SKIP || (ops[k] >= RISP_OP_WB_MANUAL && ops[k] <= RISP_OP_GAIN3),
                       "risp_bilateral_chain_fwd: op %d not allowed after the stencil", ops[k]);
        RISP_CHECK_ARG(ops[k] == RISP_OP_SKIP || (params[k] && outs[k]), "risp_bilateral_chain_fwd: stage %d incomplete", k);
        a.ops[k] = ops[k];
        a.params[k] = params[k];
        a.outs[k] = ops[k] == RISP_OP_SKIP ? nullptr : outs[k];
        if (a.outs[k]) a.last_out = k;
    }
#ifdef RISP_NT_ALL
    a.last_out = RISP_MAX_CHAIN;
#endif
    const size_t lds = sizeof(float) * 3 * (FX + 2 * a.R) * (FY + 2 * a.R);
    dim3 grid((W + FX - 1) / FX, (H + FY - 1) / FY, N);
    hipStream_t s = (hipStream_t)stream;
    bool wbq = false;
    for (int k = 0; k < n_ops; ++k) wbq |= a.ops[k] == RISP_OP_WB_QUADRATIC;
#ifdef RISP_CHAIN_ALWAYS_WBQ
    wbq = true;
#endif
#define RISP_FUSED_LAUNCH(FB, RTV)                                                                            \
    do {                                                                                                      \
        if (wbq) hipLaunchKernelGGL((bilateral_chain_kernel<FB, RTV, true>), grid, dim3(256), lds, s, a);     \
        else hipLaunchKernelGGL((bilateral_chain_kernel<FB, RTV, false>), grid, dim3(256), lds, s, a);        \
    } while (0)
    if (a.R == 1) {
        if (from_bayer) RISP_FUSED_LAUNCH(true, 1);
        else RISP_FUSED_LAUNCH(false, 1);
    } else {
        if (from_bayer) RISP_FUSED_LAUNCH(true, 0);
        else RISP_FUSED_LAUNCH(false, 0);
    }
#undef RISP_FUSED_LAUNCH
    RISP_LAUNCH_CHECK("risp_bilateral_chain_fwd");
    return 0;
}

const char *risp_bilateral_chain_kernel(int from_bayer, int max_window, int with_wb_quadratic) {
    // the instance the dispatch above launches for these arguments, as a profiler prints it (profiles/traffic.json is keyed on it)
    static const char *names[2][2][2] = {
        {{"bilateral_chain_kernel<false,0,false>", "bilateral_chain_kernel<false,0,true>"},
         {"bilateral_chain_kernel<false,1,false>", "bilateral_chain_kernel<false,1,true>"}},
        {{"bilateral_chain_kernel<true,0,false>", "bilateral_chain_kernel<true,0,true>"},
         {"bilateral_chain_kernel<true,1,false>", "bilateral_chain_kernel<true,1,true>"}}};
    return names[from_bayer ? 1 : 0][max_window / 2 == 1 ? 1 : 0][with_wb_quadratic ? 1 : 0];
}

}  // extern "C"
