// 3x3 and 5x5 convolution layers on the fp32 matrix cores with a one-dimensional Winograd transform along x: F(4,3) - four adjacent
// outputs of a filter row from 6 multiplications, half the matrix work of the direct kernel (risp_conv.hip) - and F(4,5) - four outputs
// of a 5-tap row from 8, 0.4 of it.  All arithmetic stays fp32; the row scalings of the transforms sit in the packed weights.
// These kernels serve RISP_CONV_ARITH=f32 (bench.py's cnn_f32_* leg, the yardstick the split-precision kernels are measured against)
// and the layers the split-precision kernels do not take (cin % 16 != 0: 5x5 backward passes with 3 or 12 input channels).
// Round 5 removed the F(2,3) / F(2,5) kernels and the one-row F(4,5) form that no default route had reached since round 3 (their
// measurements: NOTES.md).
#include <cstdlib>
#include "risp_common.h"

#ifndef RISP_W43G_ABL
#define RISP_W43G_ABL 0     // tools/ab_wino43.py: timing-only ablations of the LDS-DMA F(4,3) kernel (wrong outputs)
#endif
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Diagnostic build only (-DRISP_CONV_STAMPS, tools/conv_stamps.py wino): per-wave cycle shares written to the
// (otherwise unused) mask buffer.  Never compiled into the product.
#ifdef RISP_CONV_STAMPS
#define WSTAMP(var)                                      \
    do {                                                 \
        __builtin_amdgcn_sched_barrier(0);               \
        var = __builtin_amdgcn_s_memtime();              \
        __builtin_amdgcn_s_waitcnt(0xC07F);              \
        __builtin_amdgcn_sched_barrier(0);               \
    } while (0)
#else
#define WSTAMP(var) do { } while (0)
#endif
constexpr int WTH = 4;                         // output rows of a tile
constexpr int WIH = WTH + 2;                    // staged rows


// ---------------------------------------------------------------------------------------------------
// 3x3 layers, F(4,3) along x: FOUR adjacent outputs of a filter row from 6 multiplications - 18 "taps" per output
// quad where F(2,3) above needs 24 and the direct kernel 36 (half its MFMA work).  Same interpolation points and
// scaled input transform as F(2,5); U_t = (G g)_t with G rows (1/4,0,0) (1/6,1/6,1/6) (1/6,-1/6,1/6)
// (1/24,1/12,1/6) (1/24,-1/12,1/6) (0,0,1);  y0 = m0+m1+m2+m3+m4, y1 = m1-m2+2m3-2m4, y2 = m1+m2+4m3+4m4,
// y3 = m1-m2+8m3-8m4+m5;  d_j = x[4q - 1 + j].  fp32 emulation of a 64 -> 64 layer: rms error 2.1e-7 of max|y|
// (direct fp32: 1.5e-7).  Wave = one output row of 128 pixels (MFMA column = pixel quad) x one cout block of 32
// (6 accumulator tiles); the lane's four pixels of a cout row leave as ONE 16-byte store, no LDS transposition.
constexpr int W43TW = 128, W43WP = W43TW + 8, W43TAPS = 18;


// ---- epilogue of the F(4,3) kernels: the lane owns 4 consecutive pixels of 16 cout rows - one 16-byte store each.
// All loads of a batch (bias, residual, mask) are issued unconditionally inside wave-uniform branches, one batch ahead
// of the arithmetic: written per element ("if (epi & ADD) load") hipcc branches around every load and waits
// vmcnt(0) after each, i.e. 16 x 3 dependent memory round trips per wave (r01: 10 % of a wave's life).
template <int EB = 4>
__device__ __forceinline__ void w43_epilogue(const risp_conv_desc &d, const f32x16 (&acc)[6], int n, int cb, int oy, int ox,
                                             int half) {
    if (!(oy < d.H && ox < d.W)) return;
    const int epi = d.epilogue;
    const size_t hw = (size_t)d.H * d.W, pix = (size_t)oy * d.W + ox;
    const bool has_add = (epi & RISP_EPI_ADD) != 0, has_mask = (epi & RISP_EPI_MASK) != 0;
    auto co_of = [&](int e) { return cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * half; };
    float bias[16];
    if (!(epi & RISP_EPI_NOBIAS)) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = co_of(e);
            bias[e] = d.bias[co < d.cout ? co : d.cout - 1];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) bias[e] = 0.f;
    }
    float4 av[2][EB], mv[2][EB];
    auto load_batch = [&](int b, int slot) {
        if (has_add) {
#pragma unroll
            for (int i = 0; i < EB; ++i) {
                const int co = co_of(EB * b + i);
                const int cc = co < d.add_c ? co : d.add_c - 1;               // clamped: always a valid address
                av[slot][i] = *reinterpret_cast<const float4 *>(d.add + ((size_t)n * d.add_c + cc) * hw + pix);
            }
        }
        if (has_mask) {
#pragma unroll
            for (int i = 0; i < EB; ++i) {
                const int co = co_of(EB * b + i);
                const int cc = co < d.cout ? co : d.cout - 1;
                mv[slot][i] = *reinterpret_cast<const float4 *>(d.mask + ((size_t)n * d.cout + cc) * hw + pix);
            }
        }
    };
    load_batch(0, 0);
#pragma unroll
    for (int b = 0; b < 16 / EB; ++b) {
        const int slot = b & 1;
        if (b + 1 < 16 / EB) load_batch(b + 1, slot ^ 1);
#pragma unroll
        for (int i = 0; i < EB; ++i) {
            const int e = EB * b + i, co = co_of(e);
            const float m0 = acc[0][e], m1 = acc[1][e], m2 = acc[2][e], m3 = acc[3][e], m4 = acc[4][e], m5 = acc[5][e];
            const float a12 = m1 + m2, s12 = m1 - m2, a34 = m3 + m4, s34 = m3 - m4;
            const float bb = bias[e];
            float4 o = make_float4(m0 + a12 + a34 + bb, s12 + 2.f * s34 + bb, a12 + 4.f * a34 + bb, s12 + 8.f * s34 + m5 + bb);
            if (has_add && co < d.add_c) {
                const float4 a = av[slot][i];
                o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
            }
            if (epi & RISP_EPI_RELU) {
                o.x = o.x > 0.f ? o.x : 0.f;
                o.y = o.y > 0.f ? o.y : 0.f;
                o.z = o.z > 0.f ? o.z : 0.f;
                o.w = o.w > 0.f ? o.w : 0.f;
            }
            if (has_mask) {
                const float4 m = mv[slot][i];
                o.x = m.x > 0.f ? o.x : 0.f;
                o.y = m.y > 0.f ? o.y : 0.f;
                o.z = m.z > 0.f ? o.z : 0.f;
                o.w = m.w > 0.f ? o.w : 0.f;
            }
            if (co < d.cout) *reinterpret_cast<float4 *>(d.y + ((size_t)n * d.cout + co) * hw + pix) = o;
        }
    }
}

#ifndef RISP_W43_WAVES
#define RISP_W43_WAVES 2
#endif
#ifndef RISP_W43_ABL
#define RISP_W43_ABL 0     // diagnostic builds (tools/ab_wino43.py): 1 no epilogue, 2 no global loads after the first chunk, 3 no barriers, 4 no LDS operand reads after the first chunk
#endif
template <int CK>
__global__ __launch_bounds__(256, RISP_W43_WAVES) void conv_wino43_kernel(const risp_conv_desc d, int ncb) {
    constexpr int CP = 32;
    constexpr int XN = CK * WIH * W43WP, WN = W43TAPS * CK * CP;
    constexpr int NXV = (XN / 4 + 255) / 256, NWV = (WN / 4 + 255) / 256;
    constexpr int NF = NXV + NWV;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sx = smem;                                  // [2][CK][WIH][W43WP]
    float *sw = smem + 2 * XN;                         // [2][W43TAPS][CK][CP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int x0 = blockIdx.x * W43TW, y0 = blockIdx.y * WTH, n = blockIdx.z / ncb, cb = blockIdx.z - n * ncb;
    const int nchunks = (d.cin + CK - 1) / CK;
    const float *__restrict__ wpack = d.wpack + (size_t)cb * nchunks * WN;
    unsigned long long t_k0 = 0, rt_k0 = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0, s_bar = 0, s_cmp = 0, s_pub = 0, t_begin = 0, t_loop_end = 0;
    (void)t_k0; (void)rt_k0; (void)t0; (void)t1; (void)t2; (void)t3; (void)s_bar; (void)s_cmp; (void)s_pub; (void)t_begin; (void)t_loop_end;
    WSTAMP(t_k0);
#ifdef RISP_CONV_STAMPS
    rt_k0 = __builtin_amdgcn_s_memrealtime();
#endif

    f32x16 acc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    const size_t hw = (size_t)d.H * d.W;
    const float *xn = d.x + (size_t)n * d.cin * hw;
    int xoff[NXV], xcl[NXV];
    float4 xr[NXV], wr[NWV];
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
        const int v = tid + 256 * i;
        const int cl = v / (WIH * (W43WP / 4)), rem = v - cl * (WIH * (W43WP / 4));
        const int iy = rem / (W43WP / 4), q = rem - iy * (W43WP / 4);
        const int gy = y0 + iy - 1, gx = x0 - 4 + 4 * q;
        const bool ok = v < XN / 4 && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
        xcl[i] = ok ? cl : -1;                         // -1: outside the image -> zeros
        xoff[i] = (cl * d.H + gy) * d.W + gx;
    }
    auto fetch_one = [&](int ch, int j) {              // j is a compile-time constant at every call site
        if (j < NXV) {
            const int ci = ch * CK + xcl[j];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (xcl[j] >= 0 && ci < d.cin) v = *reinterpret_cast<const float4 *>(xn + (size_t)ch * CK * hw + xoff[j]);
            xr[j] = v;
        } else if (j < NF) {
            const int v = tid + 256 * (j - NXV);
            wr[j - NXV] = (v < WN / 4) ? reinterpret_cast<const float4 *>(wpack + (size_t)ch * WN)[v]
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // slot j of the staging registers (0 .. NXV-1 input tile, NXV .. NF-1 weight slab) -> LDS buffer `buf`
    auto publish_one = [&](int buf, int j) {
        if (j < NXV) {
            const int v = tid + 256 * j;
            if (256 * (j + 1) <= XN / 4 || v < XN / 4) reinterpret_cast<float4 *>(sx + buf * XN)[v] = xr[j];   // full rounds: no branch
        } else if (j < NF) {
            const int v = tid + 256 * (j - NXV);
            if (256 * (j - NXV + 1) <= WN / 4 || v < WN / 4) reinterpret_cast<float4 *>(sw + buf * WN)[v] = wr[j - NXV];
        }
    };
    auto publish = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NF; ++j) publish_one(buf, j);
    };

    WSTAMP(t_begin);
#pragma unroll
    for (int j = 0; j < NF; ++j) fetch_one(0, j);
    publish(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        WSTAMP(t0);
#if RISP_W43_ABL != 3
        __syncthreads();                               // tile ch published, tile ch-1 no longer read
#endif
        WSTAMP(t1);
        const bool more = ch + 1 < nchunks;
#if RISP_W43_ABL != 2
        if (more) {
#pragma unroll
            for (int j = 0; j < NF; ++j) fetch_one(ch + 1, j);
        }
#endif
        // d0 of quad q sits at staged column 4q + 3 (image x0 + 4q - 1), d1..d4 in the 16-byte slot 4q + 4, d5 at 4q + 8.
        // The six operands come from THREE ds_read_b128 (slots q, q+1, q+2: conflict-free, 4 LDS cycles each) instead of
        // six scalar reads at a 16-byte lane stride (4-way bank conflicts on (a/4) mod 32: 8 cycles each, r01 PMC:
        // 53 % of the LDS-active cycles were conflict stalls).
        const f32x4 *bx = reinterpret_cast<const f32x4 *>(sx + buf * XN + (half * WIH + wave) * W43WP) + l31;
        const float *aw = sw + buf * WN + half * CP + l31;
        constexpr int NG = 3 * (CK / 2);
        float opa[2][6];
        f32x4 opd[2][3];
        auto load_group = [&](int g, int slot) {
            const int ky = g / (CK / 2), cp = g - ky * (CK / 2);
            const f32x4 *dp = bx + (2 * cp * WIH + ky) * (W43WP / 4);
#pragma unroll
            for (int j = 0; j < 3; ++j) opd[slot][j] = dp[j];
#pragma unroll
            for (int t = 0; t < 6; ++t) opa[slot][t] = aw[((ky * 6 + t) * CK + 2 * cp) * CP];
        };
#if RISP_W43_ABL == 4 || RISP_W43_ABL == 6
        if (ch == 0)
#endif
        load_group(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int slot = g & 1;
#if RISP_W43_ABL == 4 || RISP_W43_ABL == 6
            if (ch == 0)
#endif
            if (g + 1 < NG) load_group(g + 1, slot ^ 1);
            __builtin_amdgcn_sched_barrier(0);         // keep the reads above this group's MFMAs
            // (the empty asm consumes whole 16-byte tuples: without it the compiler narrows the outer two reads to the
            // single elements used and falls back to the conflicting scalar form)
            asm volatile("" : "+v"(opd[slot][0]), "+v"(opd[slot][2]));
            const float d0 = opd[slot][0].w, d1 = opd[slot][1].x, d2 = opd[slot][1].y, d3 = opd[slot][1].z,
                        d4 = opd[slot][1].w, d5 = opd[slot][2].x;
            const float s12 = d1 + d2, s34 = d3 + d4, m12 = d1 - d2, m34 = d3 - d4, m13 = d1 - d3, m24 = d2 - d4;
#if RISP_W43_ABL == 5 || RISP_W43_ABL == 6
            const float bv[6] = {d0, d1, d2, d3, d4, d5};      // diagnostic: no input transform
            (void)s12; (void)s34; (void)m12; (void)m34; (void)m13; (void)m24;
#else
            // fused multiply-adds (exact products, one rounding each): 14 vector instructions instead of 22
            const float bv[6] = {__builtin_fmaf(-5.f, d2, __builtin_fmaf(4.f, d0, d4)), __builtin_fmaf(4.f, s12, -s34),
                                 __builtin_fmaf(-4.f, m12, m34), __builtin_fmaf(-2.f, m13, -m24), __builtin_fmaf(2.f, m13, -m24),
                                 __builtin_fmaf(-5.f, d3, __builtin_fmaf(4.f, d1, d5))};
#endif
#pragma unroll
            for (int t = 0; t < 6; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[slot][t], bv[t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        WSTAMP(t2);
#if RISP_W43_ABL != 7
        if (more) publish(buf ^ 1);
#endif
        WSTAMP(t3);
#ifdef RISP_CONV_STAMPS
        s_bar += t1 - t0;
        s_cmp += t2 - t1;
        s_pub += t3 - t2;
#endif
    }
    WSTAMP(t_loop_end);

#if RISP_W43_ABL == 1
    if (acc[0][0] == 123.456f)
#endif
    w43_epilogue(d, acc, n, cb, y0 + wave, x0 + 4 * l31, half);
#ifdef RISP_CONV_STAMPS
    if (lane == 0 && d.mask && !(d.epilogue & RISP_EPI_MASK)) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<float *>(d.mask)) +
                                8 * ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave);
        unsigned long long t_end;
        __builtin_amdgcn_s_waitcnt(0x0070);
        WSTAMP(t_end);
        o[0] = s_bar; o[1] = s_cmp; o[2] = s_pub;
        o[3] = t_begin - t_k0;
        o[4] = t_loop_end - t_begin;
        o[5] = t_end - t_loop_end;
        o[6] = rt_k0;
        o[7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// ---------------------------------------------------------------------------------------------------
// F(4,3) with LDS-DMA staging (global_load_lds_dwordx4): the form risp_conv2d_wino43 launches whenever cin % 4 == 0.
// Same tile, operand reads, MFMA stream and epilogue as conv_wino43_kernel; the input tile and the weight slab of a
// chunk go global -> LDS directly: no staging registers and no ds_write pass, which (with a one-deep epilogue
// prefetch) brings the kernel to 166 registers = THREE workgroups per CU on 2 x 24.5 KB of LDS each.  The staging
// area is zeroed once; lanes whose element lies outside the image - the same lanes in every chunk, hence the
// cin % 4 == 0 requirement - are masked out of the DMA by EXEC (inside the asm statement, so every wave issues
// exactly 6 DMA instructions per chunk: 24 wave-instructions of 64 x 16 bytes, 14 for the 896-float4 padded input
// tile, 10 for the 640-float4 padded weight slab) and keep their zeros.  (Pointing such lanes at 16 zero bytes in
// global memory instead costs 60 %: every CU then hammers one cache line.)  The DMA is inline asm: with the builtin
// hipcc degrades the counted lgkmcnt waits of the operand reads to lgkmcnt(0).
// Measured (tools/ab_wino43.py, 64 -> 64 on 64 x 128 x 128): 397 us register-staged, 384 us this form at 2 workgroups
// per CU, **368 us** at 3 (3 LDS stages with the transfer two chunks ahead: 392 us); residual + ReLU layer 422 -> 396 us.
template <int STAGES, int WGS>
__global__ __launch_bounds__(256, WGS) void conv_wino43_glds_kernel(const risp_conv_desc d, int ncb) {
    constexpr int CK = 4, CP = 32;
    constexpr int XN = CK * WIH * W43WP, WN = W43TAPS * CK * CP;       // floats: 3264, 2304
    constexpr int XI = 14, WI = 10, PER_WAVE = (XI + WI) / 4;            // wave-instructions per chunk
    constexpr int XPAD = XI * 64 * 4, STAGE = (XI + WI) * 64 * 4;        // floats per padded input tile / per stage
    static_assert(XN <= XPAD && WN <= WI * 64 * 4 && (XI + WI) % 4 == 0, "staging layout");
    extern __shared__ __attribute__((aligned(16))) float smem[];        // [STAGES][STAGE]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int x0 = blockIdx.x * W43TW, y0 = blockIdx.y * WTH, n = blockIdx.z / ncb, cb = blockIdx.z - n * ncb;
    const int nchunks = d.cin / CK;
    const float *__restrict__ wpack = d.wpack + (size_t)cb * nchunks * WN;
    const size_t hw = (size_t)d.H * d.W;
    const float *xn = d.x + (size_t)n * d.cin * hw;

    f32x16 acc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // zero the whole staging area once (masked lanes never write their slots)
    for (int v = tid; v < STAGES * STAGE / 4; v += 256) reinterpret_cast<float4 *>(smem)[v] = make_float4(0.f, 0.f, 0.f, 0.f);

    // this wave's 6 DMA slots: id = wave + 4 j; ids 0..13 input tile, 14..23 weight slab
    const float *src0[PER_WAVE];
    unsigned long long mask[PER_WAVE];
    int step[PER_WAVE];
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
        const int id = wave + 4 * j;
        bool ok;
        if (id < XI) {
            const int v = id * 64 + lane;
            const int cl = v / (WIH * (W43WP / 4)), rem = v - cl * (WIH * (W43WP / 4));
            const int iy = rem / (W43WP / 4), q = rem - iy * (W43WP / 4);
            const int gy = y0 + iy - 1, gx = x0 - 4 + 4 * q;
            ok = v < XN / 4 && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
            src0[j] = xn + ((size_t)cl * d.H + gy) * d.W + gx;
            step[j] = CK * (int)hw;
        } else {
            const int v = (id - XI) * 64 + lane;
            ok = v < WN / 4;
            src0[j] = wpack + 4 * v;
            step[j] = WN;
        }
        mask[j] = __builtin_amdgcn_ballot_w64(ok);
    }
    auto issue = [&](int ch, int buf) {
        float *stage = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) {
            const int id = wave + 4 * j;
            const float *src = src0[j] + (size_t)ch * step[j];
            float *dst = stage + (id < XI ? id * 256 : XPAD + (id - XI) * 256);      // wave-uniform; the lane's slot is +16 B * lane
            lds_dma16(src, dst, mask[j]);
        }
    };

    __syncthreads();                                   // zeros in place before the first DMA lands
#pragma unroll
    for (int c = 0; c < STAGES - 1; ++c)
        if (c < nchunks) issue(c, c);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch % STAGES;
        __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): every DMA this wave has issued (chunks <= ch + STAGES - 2) has landed
#if RISP_W43G_ABL != 1
        __builtin_amdgcn_s_barrier();                  // ... for every wave; buffer (ch-1) % STAGES is free
#endif
#if RISP_W43G_ABL != 3
        if (ch + STAGES - 1 < nchunks) issue(ch + STAGES - 1, (ch + STAGES - 1) % STAGES);
#endif
        const float *sx = smem + buf * STAGE, *sw = sx + XPAD;
        const f32x4 *bx = reinterpret_cast<const f32x4 *>(sx + (half * WIH + wave) * W43WP) + l31;
        const float *aw = sw + half * CP + l31;
        constexpr int NG = 3 * (CK / 2);
        float opa[2][6];
        f32x4 opd[2][3];
        auto load_group = [&](int g, int slot) {
            const int ky = g / (CK / 2), cp = g - ky * (CK / 2);
            const f32x4 *dp = bx + (2 * cp * WIH + ky) * (W43WP / 4);
#pragma unroll
            for (int j = 0; j < 3; ++j) opd[slot][j] = dp[j];
#pragma unroll
            for (int t = 0; t < 6; ++t) opa[slot][t] = aw[((ky * 6 + t) * CK + 2 * cp) * CP];
        };
        load_group(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int slot = g & 1;
            if (g + 1 < NG) load_group(g + 1, slot ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" : "+v"(opd[slot][0]), "+v"(opd[slot][2]));
            const float d0 = opd[slot][0].w, d1 = opd[slot][1].x, d2 = opd[slot][1].y, d3 = opd[slot][1].z,
                        d4 = opd[slot][1].w, d5 = opd[slot][2].x;
            const float s12 = d1 + d2, s34 = d3 + d4, m12 = d1 - d2, m34 = d3 - d4, m13 = d1 - d3, m24 = d2 - d4;
            const float bv[6] = {__builtin_fmaf(-5.f, d2, __builtin_fmaf(4.f, d0, d4)), __builtin_fmaf(4.f, s12, -s34),
                                 __builtin_fmaf(-4.f, m12, m34), __builtin_fmaf(-2.f, m13, -m24), __builtin_fmaf(2.f, m13, -m24),
                                 __builtin_fmaf(-5.f, d3, __builtin_fmaf(4.f, d1, d5))};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#if RISP_W43G_ABL == 5
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[slot][t], opd[slot][t >> 1][t & 1], acc[t], 0, 0, 0);
#else
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[slot][t], bv[t], acc[t], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#if RISP_W43G_ABL == 2
    if (acc[0][0] == 123.456f)
#endif
    w43_epilogue<WGS == 3 ? 1 : 4>(d, acc, n, cb, y0 + wave, x0 + 4 * l31, half);   // 3 workgroups per CU: 168 registers
}

// ---------------------------------------------------------------------------------------------------
// F(4,3), BOTH cout blocks of a 33..64-cout layer in one wave (round 3).  tools/mfma_valu.hip: a vector instruction beside
// the fp32 matrix stream costs its own 4-5 cycles of matrix time whatever the tile form (the fp32 matrix and vector
// pipes do not overlap), and an LDS-DMA piece ~60; so what is worth halving is the work PER MATRIX INSTRUCTION.  Here a
// wave keeps 12 accumulator tiles (one output row of 128 pixels x 64 couts): the transformed B operands of a group
// (14 vector instructions) now feed 12 matrix instructions instead of 6, and the input tile is staged once per 64 couts
// (13 + 2 x 9 = 31 pieces per chunk where two workgroups of conv_wino43_glds_kernel take 48).  192 accumulator registers:
// two workgroups per CU, so the operand registers are kept to one set - the A operands of the second cout block are read
// while the first block's six matrix instructions run, the next group's rows and first-block A operands during the
// second's.  Same packed weights, same arithmetic per output as conv_wino43_glds_kernel: bit-identical results.
#ifndef RISP_W43_B2
#define RISP_W43_B2 1
#endif
#ifndef RISP_W43B2_ABL
#define RISP_W43B2_ABL 0     // diagnostic builds (tools/ab_wino43.py; outputs wrong, only the time matters): 1 no barrier, 2 no epilogue, 3 no transfers after the first chunk, 5 no input transform
#endif
#ifndef RISP_W43_B2_EB
#define RISP_W43_B2_EB 1       // epilogue prefetch depth of cout block 0 (all 192 accumulator registers still live)
#endif
#ifndef RISP_W43_B2_EB1
#define RISP_W43_B2_EB1 4      // ... of cout block 1 (block 0's accumulators are dead by then)
#endif
__global__ __launch_bounds__(256, 2) void conv_wino43_b2_kernel(const risp_conv_desc d) {
    constexpr int CK = 4, CP = 32;
    constexpr int XN = CK * WIH * W43WP, WN = W43TAPS * CK * CP;       // floats: 3264, 2304
    constexpr int XI = 13, WI = 9, NP = XI + 2 * WI, PER_WAVE = (NP + 3) / 4;       // 31 wave-instructions per chunk
    constexpr int STAGE = NP * 256;                                     // floats per stage: tile pieces, slab of block 0, slab of block 1
    static_assert(XN <= XI * 256 && WN == WI * 256, "staging layout");
    extern __shared__ __attribute__((aligned(16))) float smem[];        // [2][STAGE]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int x0 = blockIdx.x * W43TW, y0 = blockIdx.y * WTH, n = blockIdx.z;
    const int nchunks = d.cin / CK;
    const size_t hw = (size_t)d.H * d.W;
    const float *xn = d.x + (size_t)n * d.cin * hw;

    f32x16 acc[2][6];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[b][t][e] = 0.f;

    // zero the input-tile part of both stages once: lanes outside the image are masked out of every DMA and keep it
    for (int v = tid; v < 2 * XI * 64; v += 256)
        reinterpret_cast<float4 *>(smem + (v >= XI * 64 ? STAGE - XI * 256 : 0))[v] = make_float4(0.f, 0.f, 0.f, 0.f);

    const float *src0[PER_WAVE];
    unsigned long long mask[PER_WAVE];
    int step[PER_WAVE];
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
        const int id = wave + 4 * j;
        bool ok;
        if (id < XI) {
            const int v = id * 64 + lane;
            const int cl = v / (WIH * (W43WP / 4)), rem = v - cl * (WIH * (W43WP / 4));
            const int iy = rem / (W43WP / 4), q = rem - iy * (W43WP / 4);
            const int gy = y0 + iy - 1, gx = x0 - 4 + 4 * q;
            ok = v < XN / 4 && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
            src0[j] = xn + ((size_t)cl * d.H + gy) * d.W + gx;
            step[j] = CK * (int)hw;
        } else {
            const int wb = id - XI < WI ? 0 : 1;                       // cout block of this slab piece
            ok = id < NP;
            src0[j] = d.wpack + (size_t)wb * nchunks * WN + 4 * ((id - XI - wb * WI) * 64 + lane);
            step[j] = WN;
        }
        mask[j] = __builtin_amdgcn_ballot_w64(ok);
    }
    auto issue = [&](int ch, int buf) {
        float *stage = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) {
            const int id = wave + 4 * j;
            if (id < NP) lds_dma16(src0[j] + (size_t)ch * step[j], stage + id * 256, mask[j]);      // wave-uniform guard (piece 31 does not exist)
        }
    };

    __syncthreads();                                   // zeros in place before the first DMA lands
    issue(0, 0);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): this wave's pieces of chunk ch have landed
#if RISP_W43B2_ABL != 1
        __builtin_amdgcn_s_barrier();                  // ... every wave's; the other stage is free
#endif
#if RISP_W43B2_ABL != 3
        if (ch + 1 < nchunks) issue(ch + 1, buf ^ 1);
#endif
        const float *sx = smem + buf * STAGE, *sw = sx + XI * 256;
        const f32x4 *bx = reinterpret_cast<const f32x4 *>(sx + (half * WIH + wave) * W43WP) + l31;
        const float *aw = sw + half * CP + l31;
        constexpr int NG = 3 * (CK / 2);
        float opa[2][6];
        f32x4 opd[3];
        auto load_rows = [&](int g) {
            const int ky = g / (CK / 2), cp = g - ky * (CK / 2);
            const f32x4 *dp = bx + (2 * cp * WIH + ky) * (W43WP / 4);
#pragma unroll
            for (int j = 0; j < 3; ++j) opd[j] = dp[j];
        };
        auto load_a = [&](int g, int b) {
            const int ky = g / (CK / 2), cp = g - ky * (CK / 2);
#pragma unroll
            for (int t = 0; t < 6; ++t) opa[b][t] = aw[b * WN + ((ky * 6 + t) * CK + 2 * cp) * CP];
        };
        load_rows(0);
        load_a(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            load_a(g, 1);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" : "+v"(opd[0]), "+v"(opd[2]));
            const float d0 = opd[0].w, d1 = opd[1].x, d2 = opd[1].y, d3 = opd[1].z, d4 = opd[1].w, d5 = opd[2].x;
            const float s12 = d1 + d2, s34 = d3 + d4, m12 = d1 - d2, m34 = d3 - d4, m13 = d1 - d3, m24 = d2 - d4;
            const float bv[6] = {__builtin_fmaf(-5.f, d2, __builtin_fmaf(4.f, d0, d4)), __builtin_fmaf(4.f, s12, -s34),
                                 __builtin_fmaf(-4.f, m12, m34), __builtin_fmaf(-2.f, m13, -m24), __builtin_fmaf(2.f, m13, -m24),
                                 __builtin_fmaf(-5.f, d3, __builtin_fmaf(4.f, d1, d5))};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#if RISP_W43B2_ABL == 5
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[0][t], opd[t >> 1][t & 1], acc[0][t], 0, 0, 0);
#else
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[0][t], bv[t], acc[0][t], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < NG) {
                load_rows(g + 1);
                load_a(g + 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 6; ++t)
#if RISP_W43B2_ABL == 5
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[1][t], opd[t >> 1][t & 1], acc[1][t], 0, 0, 0);
#else
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[1][t], bv[t], acc[1][t], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#if RISP_W43B2_ABL == 2
    if (acc[0][0][0] == 123.456f)
#endif
    w43_epilogue<RISP_W43_B2_EB>(d, acc[0], n, 0, y0 + wave, x0 + 4 * l31, half);
#if RISP_W43B2_ABL == 2
    if (acc[1][0][0] == 123.456f)
#endif
    w43_epilogue<RISP_W43_B2_EB1>(d, acc[1], n, 1, y0 + wave, x0 + 4 * l31, half);
}

// ---------------------------------------------------------------------------------------------------
// F(4,5) along x with LDS-DMA staging (round 3): four outputs of a 5-tap filter row from 8 products - 40 "taps" per pixel
// QUAD where F(2,5) issues 30 per pixel PAIR, i.e. 2/3 of its matrix instructions (SRCNNRes' 64 -> 32 layer and its
// backward are a third of a search step, srcnn_res_arch.py:20).  Interpolation points 0, +-1, +-2, +-1/2, inf; the row
// scalings sit in the packed weights so that the on-the-fly input transform has small coefficients and shares its even
// and odd halves between a point and its negative (26 vector instructions per 8 matrix instructions):
//     V0 = (d0 - d6) + 5.25 (d4 - d2)                      V7 = (d7 - d1) + 5.25 (d3 - d5)
//     V1,2 = E1 +- O1,  E1 = 4 (d2 + d6) - 17 d4,           O1 = 4 (d1 + d5) - 17 d3
//     V3,4 = E3 +- 2 O3, E3 = d2 - 5 d4 + 4 d6,             O3 = d1 - 5 d3 + 4 d5
//     V5,6 = 2 E5 +- O5, E5 = 4 d2 - 5 d4 + d6,             O5 = 4 d1 - 5 d3 + d5        (d_j = x[4q - 2 + j])
//     y0 = m0 + (m1+m2) + (m3+m4) + (m5+m6)                 y1 = (m1-m2) + 2 (m3-m4) + (m5-m6) / 2
//     y2 = (m1+m2) + 4 (m3+m4) + (m5+m6) / 4                y3 = (m1-m2) + 8 (m3-m4) + (m5-m6) / 8 + m7
// fp32 emulation of a 64-channel row (random weights, activations in [0,1)): rms / max error 1.7e-7 / 1.8e-6 of max|y|
// against 1.5e-7 / 1.1e-6 for F(2,5) and 1.0e-7 / 8e-7 for the direct convolution.
// Tile, operand reads (three ds_read_b128 per lane and group) and store epilogue (four pixels of a cout row per lane = one
// 16-byte store) are those of conv_wino43_glds_kernel; 8 accumulator tiles (128 registers) -> two workgroups per CU;
// per chunk of 4 input channels 17 pieces of input tile (4 x 8 x 136 floats) + 20 of weight slab (40 x 4 x 32).
constexpr int W45IH = WTH + 4, W45TAPS = 40;

template <int EB = 4>
__device__ __forceinline__ void w45_epilogue(const risp_conv_desc &d, const f32x16 (&acc)[8], int n, int cb, int oy, int ox,
                                             int half) {
    if (!(oy < d.H && ox < d.W)) return;
    const int epi = d.epilogue;
    const size_t hw = (size_t)d.H * d.W, pix = (size_t)oy * d.W + ox;
    const bool has_add = (epi & RISP_EPI_ADD) != 0, has_mask = (epi & RISP_EPI_MASK) != 0;
    auto co_of = [&](int e) { return cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * half; };
    float bias[16];
    if (!(epi & RISP_EPI_NOBIAS)) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = co_of(e);
            bias[e] = d.bias[co < d.cout ? co : d.cout - 1];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) bias[e] = 0.f;
    }
    float4 av[2][EB], mv[2][EB];
    auto load_batch = [&](int b, int slot) {
        if (has_add) {
#pragma unroll
            for (int i = 0; i < EB; ++i) {
                const int co = co_of(EB * b + i);
                const int cc = co < d.add_c ? co : d.add_c - 1;               // clamped: always a valid address
                av[slot][i] = *reinterpret_cast<const float4 *>(d.add + ((size_t)n * d.add_c + cc) * hw + pix);
            }
        }
        if (has_mask) {
#pragma unroll
            for (int i = 0; i < EB; ++i) {
                const int co = co_of(EB * b + i);
                const int cc = co < d.cout ? co : d.cout - 1;
                mv[slot][i] = *reinterpret_cast<const float4 *>(d.mask + ((size_t)n * d.cout + cc) * hw + pix);
            }
        }
    };
    load_batch(0, 0);
#pragma unroll
    for (int b = 0; b < 16 / EB; ++b) {
        const int slot = b & 1;
        if (b + 1 < 16 / EB) load_batch(b + 1, slot ^ 1);
#pragma unroll
        for (int i = 0; i < EB; ++i) {
            const int e = EB * b + i, co = co_of(e);
            const float a12 = acc[1][e] + acc[2][e], s12 = acc[1][e] - acc[2][e], a34 = acc[3][e] + acc[4][e],
                        s34 = acc[3][e] - acc[4][e], a56 = acc[5][e] + acc[6][e], s56 = acc[5][e] - acc[6][e];
            const float bb = bias[e];
            float4 o = make_float4(acc[0][e] + a12 + a34 + a56 + bb, s12 + 2.f * s34 + 0.5f * s56 + bb,
                                   a12 + 4.f * a34 + 0.25f * a56 + bb, s12 + 8.f * s34 + 0.125f * s56 + acc[7][e] + bb);
            if (has_add && co < d.add_c) {
                const float4 a = av[slot][i];
                o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
            }
            if (epi & RISP_EPI_RELU) {
                o.x = o.x > 0.f ? o.x : 0.f;
                o.y = o.y > 0.f ? o.y : 0.f;
                o.z = o.z > 0.f ? o.z : 0.f;
                o.w = o.w > 0.f ? o.w : 0.f;
            }
            if (has_mask) {
                const float4 m = mv[slot][i];
                o.x = m.x > 0.f ? o.x : 0.f;
                o.y = m.y > 0.f ? o.y : 0.f;
                o.z = m.z > 0.f ? o.z : 0.f;
                o.w = m.w > 0.f ? o.w : 0.f;
            }
            if (co < d.cout) *reinterpret_cast<float4 *>(d.y + ((size_t)n * d.cout + co) * hw + pix) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// F(4,5) with TWO output rows per wave on 16x16x4 tiles (round 3).  A vector instruction costs its own 4-5 cycles of matrix
// time (tools/mfma_valu.hip) and F(4,5)'s input transform is 26 of them per staged row: in conv_wino45_glds_kernel a wave
// owns one output row and transforms the 5 input rows under it - 26 x 5 per 5 x 8 matrix instructions of 64 cycles, ~23 %
// on top of the matrix time.  Here the wave owns two output rows x 64 pixels x 32 couts (v_mfma_f32_16x16x4_f32: 2 rows x
// 8 points x 2 cout blocks of 16 = 32 accumulator tiles of 4 registers) and walks its SIX input rows once: each
// transformed row feeds output row 0 with filter row ky = i and output row 1 with ky = i - 1, 26 x 6 vector instructions
// per 160 matrix instructions of 32 cycles - 40 % fewer per matrix cycle.  Same workgroup tile (4 rows x 128 pixels), same
// staging (17 + 20 LDS-DMA pieces per 4-channel chunk), same interpolation points and output transform; K = 4 per
// instruction instead of 2 + 2, so results differ from the one-row kernel by fp32 rounding only.
// Weight slab of a chunk (risp.h, layout 1): [ky 5][point group 2][cout block of 16: 2][cin 4][cout 16][4 points] - a lane
// (cout m = lane & 15, cin k = lane >> 4) reads the A operands of four points with one conflict-free ds_read_b128.
#ifndef RISP_W45R2_ABL
#define RISP_W45R2_ABL 0     // diagnostic builds (tools/ab_wino43.py, RISP_AB_ENTRY=wino45; outputs wrong): 2 no epilogue, 3 no transfers after the first chunk, 5 no input transform
#endif
#ifndef RISP_W45_R2
#define RISP_W45_R2 1
#endif
template <int EB>
__device__ __forceinline__ void w45r2_epilogue(const risp_conv_desc &d, const f32x4 (&acc)[2][8][2], int n, int cb, int oy0, int ox,
                                               int kk) {
    if (!(oy0 < d.H && ox < d.W)) return;
    const int epi = d.epilogue;
    const size_t hw = (size_t)d.H * d.W;
    const bool has_add = (epi & RISP_EPI_ADD) != 0, has_mask = (epi & RISP_EPI_MASK) != 0;
    // item i = (rr * 2 + mb) * 4 + e: output row oy0 + rr, cout cb * 32 + mb * 16 + 4 * kk + e
    auto co_of = [&](int i) { return cb * 32 + ((i >> 2) & 1) * 16 + 4 * kk + (i & 3); };
    auto pix_of = [&](int i) {
        const int oy = oy0 + (i >> 3);
        return (size_t)(oy < d.H ? oy : d.H - 1) * d.W + ox;                // clamped: always a valid address
    };
    float bias[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int co = co_of(j);
        bias[j] = (epi & RISP_EPI_NOBIAS) ? 0.f : d.bias[co < d.cout ? co : d.cout - 1];
    }
    float4 av[2][EB], mv[2][EB];
    auto load_batch = [&](int b, int slot) {
        if (has_add) {
#pragma unroll
            for (int j = 0; j < EB; ++j) {
                const int i = EB * b + j, co = co_of(i);
                av[slot][j] = *reinterpret_cast<const float4 *>(d.add + ((size_t)n * d.add_c + (co < d.add_c ? co : d.add_c - 1)) * hw + pix_of(i));
            }
        }
        if (has_mask) {
#pragma unroll
            for (int j = 0; j < EB; ++j) {
                const int i = EB * b + j, co = co_of(i);
                mv[slot][j] = *reinterpret_cast<const float4 *>(d.mask + ((size_t)n * d.cout + (co < d.cout ? co : d.cout - 1)) * hw + pix_of(i));
            }
        }
    };
    load_batch(0, 0);
#pragma unroll
    for (int b = 0; b < 16 / EB; ++b) {
        const int slot = b & 1;
        if (b + 1 < 16 / EB) load_batch(b + 1, slot ^ 1);
#pragma unroll
        for (int j = 0; j < EB; ++j) {
            const int i = EB * b + j, rr = i >> 3, mb = (i >> 2) & 1, e = i & 3, co = co_of(i);
            const float m0 = acc[rr][0][mb][e], m1 = acc[rr][1][mb][e], m2 = acc[rr][2][mb][e], m3 = acc[rr][3][mb][e],
                        m4 = acc[rr][4][mb][e], m5 = acc[rr][5][mb][e], m6 = acc[rr][6][mb][e], m7 = acc[rr][7][mb][e];
            const float a12 = m1 + m2, s12 = m1 - m2, a34 = m3 + m4, s34 = m3 - m4, a56 = m5 + m6, s56 = m5 - m6;
            const float bb = bias[i & 7];
            float4 o = make_float4(m0 + a12 + a34 + a56 + bb, s12 + 2.f * s34 + 0.5f * s56 + bb, a12 + 4.f * a34 + 0.25f * a56 + bb,
                                   s12 + 8.f * s34 + 0.125f * s56 + m7 + bb);
            if (has_add && co < d.add_c) {
                const float4 a = av[slot][j];
                o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
            }
            if (epi & RISP_EPI_RELU) {
                o.x = o.x > 0.f ? o.x : 0.f;
                o.y = o.y > 0.f ? o.y : 0.f;
                o.z = o.z > 0.f ? o.z : 0.f;
                o.w = o.w > 0.f ? o.w : 0.f;
            }
            if (has_mask) {
                const float4 m = mv[slot][j];
                o.x = m.x > 0.f ? o.x : 0.f;
                o.y = m.y > 0.f ? o.y : 0.f;
                o.z = m.z > 0.f ? o.z : 0.f;
                o.w = m.w > 0.f ? o.w : 0.f;
            }
            if (co < d.cout && oy0 + rr < d.H) *reinterpret_cast<float4 *>(d.y + ((size_t)n * d.cout + co) * hw + pix_of(i)) = o;
        }
    }
}

__global__ __launch_bounds__(256, 2) void conv_wino45_r2_kernel(const risp_conv_desc d_in, int ncb) {
    constexpr int CK = 4, CP = 32;
    constexpr int XN = CK * W45IH * W43WP, WN = W45TAPS * CK * CP;      // floats: 4352, 5120
    constexpr int XI = XN / 256, WI = WN / 256, PIECES = XI + WI, PER_WAVE = (PIECES + 3) / 4;     // 17 + 20 wave-instructions per chunk
    constexpr int STAGE = PIECES * 256;
    static_assert(XN % 256 == 0 && WN % 256 == 0, "staging layout");
    extern __shared__ __attribute__((aligned(16))) float smem[];        // [2][STAGE]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane & 15, kk = lane >> 4, rp = wave >> 1, xh = wave & 1;
    const int x0 = blockIdx.x * W43TW, y0 = blockIdx.y * WTH, n = blockIdx.z / ncb, cb = blockIdx.z - n * ncb;
    const risp_conv_desc d = risp_conv_group_view(d_in, n);
    const int nchunks = (d.cin + CK - 1) / CK;         // cin % 4 != 0 only with a single chunk (cin < 4): see the masks below
    const float *__restrict__ wpack = d.wpack + (size_t)cb * nchunks * WN;
    const size_t hw = (size_t)d.H * d.W;
    const float *xn = d.x + (size_t)n * d.cin * hw;

    f32x4 acc[2][8][2];                                 // [output row][point][cout block of 16]
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[r][t][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // zero the input-tile part of both stages once (lanes outside the image are masked out of every transfer and keep it)
    for (int v = tid; v < 2 * XI * 64; v += 256)
        reinterpret_cast<float4 *>(smem + (v >= XI * 64 ? STAGE - XI * 256 : 0))[v] = make_float4(0.f, 0.f, 0.f, 0.f);

    const float *src0[PER_WAVE];
    unsigned long long mask[PER_WAVE];
    int step[PER_WAVE];
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
        const int id = wave + 4 * j;
        bool ok = false;
        src0[j] = xn;
        step[j] = 0;
        if (id < XI) {
            const int v = id * 64 + lane;
            const int cl = v / (W45IH * (W43WP / 4)), rem = v - cl * (W45IH * (W43WP / 4));
            const int iy = rem / (W43WP / 4), qq = rem - iy * (W43WP / 4);
            const int gy = y0 + iy - 2, gx = x0 - 4 + 4 * qq;
            ok = gy >= 0 && gy < d.H && gx >= 0 && gx < d.W && cl < d.cin;
            src0[j] = xn + ((size_t)(cl < d.cin ? cl : 0) * d.H + gy) * d.W + gx;
            step[j] = CK * (int)hw;
        } else if (id < PIECES) {
            ok = true;
            src0[j] = wpack + 4 * ((id - XI) * 64 + lane);
            step[j] = WN;
        }
        mask[j] = __builtin_amdgcn_ballot_w64(ok);
    }
    auto issue = [&](int ch, int buf) {
        float *stage = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) {
            const int id = wave + 4 * j;
            if (id < PIECES) lds_dma16(src0[j] + (size_t)ch * step[j], stage + id * 256, mask[j]);     // wave-uniform branch
        }
    };

    __syncthreads();                                   // zeros in place before the first DMA lands
    if (nchunks > 0) issue(0, 0);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): this wave's transfers of chunk ch have landed
        __builtin_amdgcn_s_barrier();                  // ... for every wave; the other stage is free
#if RISP_W45R2_ABL != 3
        if (ch + 1 < nchunks) issue(ch + 1, buf ^ 1);
#endif
        const float *sx = smem + buf * STAGE;
        // the lane's channel kk, quad 16 xh + q: staged row 2 rp + i, 16-byte slots Q, Q + 1, Q + 2 (d_j = x[4 Q - 2 + j] = slot value 2 + j)
        const f32x4 *bx = reinterpret_cast<const f32x4 *>(sx + (kk * W45IH + 2 * rp) * W43WP) + 16 * xh + q;
        const f32x4 *aw = reinterpret_cast<const f32x4 *>(sx + XN) + lane;          // [ky][point group][cout block][64 lanes] x 4 points
        f32x4 opa[2][2][2];                              // [set][point group][cout block]: filter rows ky = i (set i & 1) and i - 1
        f32x4 opd[3];
        auto load_row = [&](int i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) opd[j] = bx[i * (W43WP / 4) + j];
        };
        auto load_a = [&](int ky) {
#pragma unroll
            for (int pg = 0; pg < 2; ++pg)
#pragma unroll
                for (int b = 0; b < 2; ++b) opa[ky & 1][pg][b] = aw[((ky * 2 + pg) * 2 + b) * 64];
        };
        load_row(0);
        load_a(0);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" : "+v"(opd[0]), "+v"(opd[2]));
            const float d0 = opd[0].z, d1 = opd[0].w, d2 = opd[1].x, d3 = opd[1].y, d4 = opd[1].z, d5 = opd[1].w, d6 = opd[2].x,
                        d7 = opd[2].y;
            const float e1 = __builtin_fmaf(4.f, d2 + d6, -17.f * d4), o1 = __builtin_fmaf(4.f, d1 + d5, -17.f * d3);
            const float e3 = __builtin_fmaf(4.f, d6, __builtin_fmaf(-5.f, d4, d2)), o3 = __builtin_fmaf(4.f, d5, __builtin_fmaf(-5.f, d3, d1));
            const float e5 = __builtin_fmaf(4.f, d2, __builtin_fmaf(-5.f, d4, d6)), o5 = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
#if RISP_W45R2_ABL == 5
            const float bv[8] = {d0, d1, d2, d3, d4, d5, d6, d7};
#else
            const float bv[8] = {__builtin_fmaf(5.25f, d4 - d2, d0 - d6), e1 + o1, e1 - o1, __builtin_fmaf(2.f, o3, e3),
                                 __builtin_fmaf(-2.f, o3, e3), __builtin_fmaf(2.f, e5, o5), __builtin_fmaf(2.f, e5, -o5),
                                 __builtin_fmaf(5.25f, d3 - d5, d7 - d1)};
#endif
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < 6) load_row(i + 1);              // the raw row is consumed: the next one lands during the matrix instructions
            __builtin_amdgcn_sched_barrier(0);
            if (i >= 1) {                                // output row 1, filter row ky = i - 1
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[1][t][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(opa[(i - 1) & 1][t >> 2][b][t & 3], bv[t], acc[1][t][b], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < 5) load_a(i + 1);                // into the set ky = i - 1 just left
            __builtin_amdgcn_sched_barrier(0);
            if (i < 5) {                                 // output row 0, filter row ky = i
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[0][t][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(opa[i & 1][t >> 2][b][t & 3], bv[t], acc[0][t][b], 0, 0, 0);
            }
        }
    }
#if RISP_W45R2_ABL == 2
    if (acc[0][0][0][0] == 123.456f)
#endif
    w45r2_epilogue<4>(d, acc, n, cb, y0 + 2 * rp, x0 + 4 * (16 * xh + q), kk);
}

}  // namespace

extern "C" {

constexpr int W43CK = 4;
int risp_conv_wino43_chunk(void) { return W43CK; }

size_t risp_conv_wino43_wpack_floats(int cin, int cout) {
    return (size_t)((cout + 31) / 32) * ((cin + W43CK - 1) / W43CK) * W43TAPS * W43CK * 32;
}

int risp_conv2d_wino43(const risp_conv_desc *dp, void *stream) {
    RISP_CHECK_ARG(dp, "risp_conv2d_wino43: null descriptor");
    const risp_conv_desc &d = *dp;
    RISP_CHECK_ARG(d.x && d.wpack && d.y, "risp_conv2d_wino43: null tensor");
    RISP_CHECK_ARG(d.group_n == 0, "risp_conv2d_wino43: grouped launches are not supported by the 3x3 Winograd kernels");
    RISP_CHECK_ARG(d.N > 0 && d.H > 0 && d.W > 0 && d.W % 4 == 0 && d.cin > 0 && d.cout > 0 && d.cout <= 64 && d.ksize == 3 &&
                       (size_t)d.N * ((d.cout + 31) / 32) <= 65535,
                   "risp_conv2d_wino43: needs a 3x3 layer, cout <= 64, W %% 4 == 0 (N=%d H=%d W=%d cin=%d cout=%d k=%d)", d.N, d.H,
                   d.W, d.cin, d.cout, d.ksize);
    RISP_CHECK_ARG(d.load_mode == RISP_LOAD_PLAIN, "risp_conv2d_wino43: only plain loads");
    RISP_CHECK_ARG(!(d.epilogue & ~(RISP_EPI_RELU | RISP_EPI_ADD | RISP_EPI_MASK | RISP_EPI_NOBIAS)),
                   "risp_conv2d_wino43: epilogue %d not supported", d.epilogue);
    RISP_CHECK_ARG((d.epilogue & RISP_EPI_NOBIAS) || d.bias, "risp_conv2d_wino43: bias missing");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_ADD) || (d.add && d.add_c > 0), "risp_conv2d_wino43: add tensor missing");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_MASK) || d.mask, "risp_conv2d_wino43: mask tensor missing");
    RISP_CHECK_ARG(((reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.y) | reinterpret_cast<uintptr_t>(d.add) |
                     reinterpret_cast<uintptr_t>(d.mask) | reinterpret_cast<uintptr_t>(d.wpack)) & 15) == 0,
                   "risp_conv2d_wino43: tensors must be 16-byte aligned");
    const int ncb = (d.cout + 31) / 32;
    constexpr int XN = W43CK * WIH * W43WP, WN = W43TAPS * W43CK * 32;
    dim3 grid((d.W + W43TW - 1) / W43TW, (d.H + WTH - 1) / WTH, d.N * ncb);
#ifndef RISP_W43_NO_GLDS
    if (RISP_W43_B2 && d.cin % 4 == 0 && ncb == 2) {            // both cout blocks per wave: 12 accumulator tiles, 2 workgroups per CU
        hipLaunchKernelGGL(conv_wino43_b2_kernel, dim3(grid.x, grid.y, d.N), dim3(256), sizeof(float) * 2 * 31 * 256, (hipStream_t)stream, d);
        RISP_LAUNCH_CHECK("risp_conv2d_wino43");
        return 0;
    }
    if (d.cin % 4 == 0) {                              // LDS-DMA staging, 2 LDS stages, 3 workgroups per CU (see the kernel)
#ifndef RISP_W43_GLDS
#define RISP_W43_GLDS 2
#endif
#ifndef RISP_W43_GLDS_WGS
#define RISP_W43_GLDS_WGS 3
#endif
        constexpr int ST = RISP_W43_GLDS;
        const size_t lds_g = sizeof(float) * ST * 24 * 64 * 4;
        if (lds_g > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wino43_glds_kernel<ST, RISP_W43_GLDS_WGS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_g) != hipSuccess) {
            risp_set_error("risp_conv2d_wino43: cannot raise the dynamic LDS limit to %zu bytes", lds_g);
            return 2;
        }
        hipLaunchKernelGGL((conv_wino43_glds_kernel<ST, RISP_W43_GLDS_WGS>), grid, dim3(256), lds_g, (hipStream_t)stream, d, ncb);
        RISP_LAUNCH_CHECK("risp_conv2d_wino43");
        return 0;
    }
#endif
    const size_t lds = sizeof(float) * 2 * (XN + WN);
    hipLaunchKernelGGL(conv_wino43_kernel<W43CK>, grid, dim3(256), lds, (hipStream_t)stream, d, ncb);
    RISP_LAUNCH_CHECK("risp_conv2d_wino43");
    return 0;
}

int risp_conv_wino45_chunk(void) { return 4; }

int risp_conv_wino45_layout(void) { return 1; }

size_t risp_conv_wino45_wpack_floats(int cin, int cout) {
    return (size_t)((cout + 31) / 32) * ((cin + 3) / 4) * W45TAPS * 4 * 32;
}

int risp_conv2d_wino45(const risp_conv_desc *dp, void *stream) {
    RISP_CHECK_ARG(dp, "risp_conv2d_wino45: null descriptor");
    const risp_conv_desc &d = *dp;
    RISP_CHECK_ARG(d.x && d.wpack && d.y, "risp_conv2d_wino45: null tensor");
    RISP_CHECK_GROUP(d, "risp_conv2d_wino45");
    RISP_CHECK_ARG(d.N > 0 && d.H > 0 && d.W > 0 && d.W % 4 == 0 && d.cin > 0 && (d.cin % 4 == 0 || d.cin < 4) && d.cout > 0 && d.cout <= 64 &&
                       d.ksize == 5 && (size_t)d.N * ((d.cout + 31) / 32) <= 65535,
                   "risp_conv2d_wino45: needs a 5x5 layer, cin %% 4 == 0 or cin < 4, cout <= 64, W %% 4 == 0 (N=%d H=%d W=%d cin=%d cout=%d k=%d)",
                   d.N, d.H, d.W, d.cin, d.cout, d.ksize);
    RISP_CHECK_ARG(d.load_mode == RISP_LOAD_PLAIN, "risp_conv2d_wino45: only plain loads");
    RISP_CHECK_ARG(!(d.epilogue & ~(RISP_EPI_RELU | RISP_EPI_ADD | RISP_EPI_MASK | RISP_EPI_NOBIAS)),
                   "risp_conv2d_wino45: epilogue %d not supported", d.epilogue);
    RISP_CHECK_ARG((d.epilogue & RISP_EPI_NOBIAS) || d.bias, "risp_conv2d_wino45: bias missing");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_ADD) || (d.add && d.add_c > 0), "risp_conv2d_wino45: add tensor missing");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_MASK) || d.mask, "risp_conv2d_wino45: mask tensor missing");
    RISP_CHECK_ARG(((reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.y) | reinterpret_cast<uintptr_t>(d.add) |
                     reinterpret_cast<uintptr_t>(d.mask) | reinterpret_cast<uintptr_t>(d.wpack)) & 15) == 0,
                   "risp_conv2d_wino45: tensors must be 16-byte aligned");
    const int ncb = (d.cout + 31) / 32;
    const size_t lds = sizeof(float) * 2 * (4 * W45IH * W43WP + W45TAPS * 4 * 32);
    const auto kernel = &conv_wino45_r2_kernel;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        risp_set_error("risp_conv2d_wino45: cannot raise the dynamic LDS limit to %zu bytes", lds);
        return 2;
    }
    dim3 grid((d.W + W43TW - 1) / W43TW, (d.H + WTH - 1) / WTH, d.N * ncb);
    hipLaunchKernelGGL(kernel, grid, dim3(256), lds, (hipStream_t)stream, d, ncb);
    RISP_LAUNCH_CHECK("risp_conv2d_wino45");
    return 0;
}

}  // extern "C"
