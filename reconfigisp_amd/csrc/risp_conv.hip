// Convolution layers of the learned proxies on the gfx950 fp32 matrix cores.
//
// Reference layers: SRCNNRes (srcnn_res_arch.py:15-24), SRCNNDemosaic
// (srcnn_demosaic_arch.py:14-25), Path14lBayer / Path14lBgr + ResidualBlock
// (path_14l_bayer_arch.py:6-57, path_14l_bgr_arch.py:6-56).  All are stride-1,
// zero-'same'-padded, odd square kernels in fp32; parity bar 1e-4 relative, so the
// exact-f32 MFMA (v_mfma_f32_32x32x2_f32, a k-ordered fmaf chain) is the only matrix
// instruction that qualifies (no xf32 on gfx950; bf16/fp8 would break parity).
//
// Implicit GEMM, per workgroup (256 threads = 4 waves, one per SIMD):
//     D[cout][pixel] += W[cout][k] * X[k][pixel],   k = (chunk, ky, kx, ci)
//   * output tile 16 rows x 32 pixels x all couts (<= 64); a wave owns 4 rows and keeps
//     4 x CB accumulator tiles of 32(cout) x 32(pixel) in VGPRs (D column = lane&31 = x,
//     so epilogue stores are 128-byte row segments of the NCHW planes);
//   * per chunk of CK input channels the zero-padded halo tile CK x (16+k-1) x (32+k-1)
//     and the matching weight slab [tap][ci][cout] are staged in LDS; the MFMA A operand
//     (weights) and B operand (pixels) are single conflict-free ds_read_b32 per lane
//     (lanes 0-31 -> channel ci, lanes 32-63 -> channel ci+1 of the same tap);
//   * space-to-depth (Bayer -> RGGB planes), the SRCNNRes broadcast planes, bias, residual
//     add, ReLU, ReLU-mask (backward) and PixelShuffle(2) are folded into the load / store
//     index maps, so no layer materialises a temporary.
//   * the global loads of the next channel chunk are issued into registers before the MFMA loop
//     of the current chunk and written to LDS after it (register-staged software pipeline).
// LDS per workgroup is 38-55 KB and VGPRs <= 256, so two workgroups share a CU.
#include "risp_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Diagnostic build only (-DRISP_CONV_STAMPS, tools/conv_stamps.py): per-wave cycle shares of the
// chunk loop's phases, written to the (otherwise unused) mask buffer.  Never compiled into the product.
#ifdef RISP_CONV_STAMPS
#define RISP_STAMP(var)                                  \
    do {                                                 \
        __builtin_amdgcn_sched_barrier(0);               \
        var = __builtin_amdgcn_s_memtime();              \
        __builtin_amdgcn_s_waitcnt(0xC07F);              \
        __builtin_amdgcn_sched_barrier(0);               \
    } while (0)
#else
#define RISP_STAMP(var) do { } while (0)
#endif

constexpr int TH = 16, TW = 32, RW = TH / 4;

struct ConvCfg {
    int ck, cb;
};

// chunk depth / cout blocks per (ksize, cin, cout) - shared by the packer and the launcher
ConvCfg conv_cfg(int cin, int cout, int ks) {
    ConvCfg c;
    c.cb = cout > 32 ? 2 : 1;
    if (ks == 9) c.ck = 2;                      // 81 taps x 2 channels per slab; single LDS buffer
    else if (ks == 1) c.ck = 8;
    else if (ks == 5 && c.cb == 2) c.ck = 2;    // keeps both ping-pong buffers under 64 KB
    else c.ck = 4;                              // 3x3 / 5x5: ping-pong LDS buffers of 4 channels
    return c;
}

// Which taps of a k-wide window centred on coordinate v fall inside [0, L): case P = all of them (interior),
// case i < P = the first P - i are cut off, case P + j = the last j are cut off.  Needs L >= 2 P.
__device__ __forceinline__ int border_case(int v, int L, int P) {
    return v < P ? v : (v >= L - P ? 2 * P - (L - 1 - v) : P);
}

__device__ __forceinline__ float load_px(const risp_conv_desc &d, int n, int ci, int gy, int gx) {
    if (ci >= d.cin || gy < 0 || gy >= d.H || gx < 0 || gx >= d.W) return 0.f;
    if (d.load_mode == RISP_LOAD_PLAIN) return d.x[(((size_t)n * d.cin + ci) * d.H + gy) * d.W + gx];
    if (d.load_mode == RISP_LOAD_UNSHUFFLE2) {
        const int c = ci >> 2, i = (ci >> 1) & 1, j = ci & 1;
        return d.x[(((size_t)n * (d.cin >> 2) + c) * (2 * d.H) + 2 * gy + i) * (2 * (size_t)d.W) + 2 * gx + j];
    }
    // RISP_LOAD_CONSTCH
    if (ci < d.cin_img) return d.x[(((size_t)n * d.cin_img + ci) * d.H + gy) * d.W + gx];
    return d.cvals[n * (d.cin - d.cin_img) + (ci - d.cin_img)];
}

// FAST path: 16-byte loads of 4 consecutive pixels (W % 4 == 0, plain / const-channel inputs); the
// float4 lies entirely inside or entirely outside the image because gx and W are multiples of 4.
__device__ __forceinline__ float4 load_px4(const risp_conv_desc &d, int n, int ci, int gy, int gx) {
    if (ci >= d.cin || gy < 0 || gy >= d.H || gx < 0 || gx >= d.W) return make_float4(0.f, 0.f, 0.f, 0.f);
    const int planes = d.load_mode == RISP_LOAD_CONSTCH ? d.cin_img : d.cin;
    if (ci < planes)
        return *reinterpret_cast<const float4 *>(d.x + (((size_t)n * planes + ci) * d.H + gy) * d.W + gx);
    const float v = d.cvals[n * (d.cin - d.cin_img) + (ci - d.cin_img)];
    return make_float4(v, v, v, v);
}

// One workgroup = 16x32 output pixels x all couts.  Software pipeline over chunks of CK input
// channels: the global loads of chunk c+1 are issued into registers from INSIDE chunk c's MFMA
// stream (one 16-byte load per filter tap, so their address arithmetic fills MFMA issue gaps) and are
// written to LDS after it.  PP = ping-pong LDS buffers (one barrier per chunk); otherwise one buffer
// and two barriers.
template <int KS, int CK, int CB, bool FAST, bool PP>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const risp_conv_desc d_in) {
    constexpr int PAD = KS / 2, IH = TH + KS - 1, IW = TW + KS - 1, CP = 32 * CB, TAPS = KS * KS;
    constexpr int IWP = FAST ? (KS == 1 ? 32 : 40) : IW;      // LDS row stride (FAST: 16-byte aligned rows)
    constexpr int XOFF = FAST ? (KS == 1 ? 0 : 4 - PAD) : 0;  // first needed column inside a staged row
    constexpr int XN = CK * IH * IWP, WN = TAPS * CK * CP;
    constexpr int XSZ = (XN + 3) & ~3;
    constexpr int NXV = FAST ? (XN / 4 + 255) / 256 : 0;      // per-thread prefetch registers (float4)
    constexpr int NWV = (WN / 4 + 255) / 256;
    constexpr int NF = NXV + NWV, PER_TAP = (NF + TAPS - 1) / TAPS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sx = smem;                                  // [NBUF][CK][IH][IWP]
    float *sw = smem + (PP ? 2 : 1) * XSZ;             // [NBUF][TAPS][CK][CP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, n = blockIdx.z;
    const risp_conv_desc d = risp_conv_group_view(d_in, n);
    const int wrow = wave * RW;
    const int nchunks = (d.cin + CK - 1) / CK;
#ifdef RISP_CONV_STAMPS
    unsigned long long t_k0, rt_k0;
    RISP_STAMP(t_k0);
    rt_k0 = __builtin_amdgcn_s_memrealtime();
#endif

    f32x16 acc[RW][CB];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;

    // ---- chunk-invariant part of the staging addresses (FAST path)
    const int planes = d.load_mode == RISP_LOAD_CONSTCH ? d.cin_img : d.cin;
    const size_t hw = (size_t)d.H * d.W;
    const float *xn = d.x + (size_t)n * planes * hw;
    int xoff[NXV > 0 ? NXV : 1], xcl[NXV > 0 ? NXV : 1];
    float4 xr[NXV > 0 ? NXV : 1];
    float4 wr[NWV];
    if constexpr (FAST) {
#pragma unroll
        for (int i = 0; i < NXV; ++i) {
            const int v = tid + 256 * i;
            const int cl = v / (IH * (IWP / 4)), rem = v - cl * (IH * (IWP / 4));
            const int iy = rem / (IWP / 4), q = rem - iy * (IWP / 4);
            const int gy = y0 + iy - PAD, gx = x0 - XOFF - PAD + 4 * q;
            const bool ok = v < XN / 4 && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
            xcl[i] = ok ? cl : -1;                     // -1: outside the image -> zeros
            xoff[i] = (cl * d.H + gy) * d.W + gx;
        }
    }
    auto fetch_one = [&](int ch, int j) {              // j is a compile-time constant at every call site
        if (j < NXV) {
            if constexpr (FAST) {
                const int ci = ch * CK + xcl[j];
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (xcl[j] >= 0 && ci < d.cin) {
                    if (ci < planes) {
                        v = *reinterpret_cast<const float4 *>(xn + (size_t)ch * CK * hw + xoff[j]);
                    } else {
                        const float c = d.cvals[n * (d.cin - d.cin_img) + (ci - d.cin_img)];
                        v = make_float4(c, c, c, c);
                    }
                }
                xr[j] = v;
            }
        } else if (j < NF) {
            const int v = tid + 256 * (j - NXV);
            wr[j - NXV] = (v < WN / 4) ? reinterpret_cast<const float4 *>(d.wpack + (size_t)ch * WN)[v]
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto publish = [&](int ch, int buf) {
        float *sxb = sx + buf * XSZ, *swb = sw + buf * WN;
        if constexpr (FAST) {
#pragma unroll
            for (int i = 0; i < NXV; ++i) {
                const int v = tid + 256 * i;
                if (v < XN / 4) reinterpret_cast<float4 *>(sxb)[v] = xr[i];
            }
        } else {
            // generic path (strided space-to-depth reads, W % 4 != 0): element loads in batches of 8,
            // all issued before the first is consumed; not prefetched across chunks
            constexpr int G = 8;
            for (int base = 0; base < XN; base += 256 * G) {
                float t[G];
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    const int v = base + tid + 256 * i;
                    const int cl = v / (IH * IW), rem = v - cl * (IH * IW);
                    const int iy = rem / IW, ix = rem - iy * IW;
                    t[i] = (v < XN) ? load_px(d, n, ch * CK + cl, y0 + iy - PAD, x0 + ix - PAD) : 0.f;
                }
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    const int v = base + tid + 256 * i;
                    if (v < XN) sxb[v] = t[i];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int v = tid + 256 * i;
            if (v < WN / 4) reinterpret_cast<float4 *>(swb)[v] = wr[i];
        }
    };

    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, s_bar = 0, s_cmp = 0, s_pub = 0, t_begin = 0;
    (void)t0; (void)t1; (void)t2; (void)t3; (void)s_bar; (void)s_cmp; (void)s_pub; (void)t_begin;
    RISP_STAMP(t_begin);
#pragma unroll
    for (int j = 0; j < NF; ++j) fetch_one(0, j);
    if constexpr (PP) publish(0, 0);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = PP ? (ch & 1) : 0;
        RISP_STAMP(t0);
        __syncthreads();          // PP: tile ch published, tile ch-1 no longer read.  !PP: tile ch-1 no longer read
        if constexpr (!PP) {
            publish(ch, 0);
            __syncthreads();
        }
        RISP_STAMP(t1);
        const bool more = ch + 1 < nchunks;
        const float *bx = sx + buf * XSZ + (half * IH + wrow) * IWP + XOFF + l31;   // channel `half` of pair 0
        const float *aw = sw + buf * WN + half * CP + l31;
        // Register double-buffering of the LDS operands: the reads of group g+1 = (tap, channel pair) are issued
        // before the RW x CB MFMAs of group g (the scheduler otherwise sinks each read next to its use and the wave
        // eats the LDS latency once per group: measured 0.69 -> 0.56 ms on the Winograd twin of this loop).
        constexpr int NG = TAPS * (CK / 2);
        float opa[2][CB], opb[2][RW];
        auto load_group = [&](int g, int slot) {
            const int tap = g / (CK / 2), cp = g - tap * (CK / 2);
            const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
            for (int c = 0; c < CB; ++c) opa[slot][c] = aw[(tap * CK + 2 * cp) * CP + c * 32];
#pragma unroll
            for (int r = 0; r < RW; ++r) opb[slot][r] = bx[(2 * cp * IH + r + ky) * IWP + kx];
        };
        load_group(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int slot = g & 1;
            if (g + 1 < NG) load_group(g + 1, slot ^ 1);
            if (more && g % (CK / 2) == 0) {           // one filter tap's share of the next chunk's global loads
#pragma unroll
                for (int j = (g / (CK / 2)) * PER_TAP; j < (g / (CK / 2) + 1) * PER_TAP; ++j) fetch_one(ch + 1, j);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
                for (int c = 0; c < CB; ++c)
                    acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[slot][c], opb[slot][r], acc[r][c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        RISP_STAMP(t2);
        if constexpr (PP) {
            if (more) publish(ch + 1, buf ^ 1);
        }
        RISP_STAMP(t3);
#ifdef RISP_CONV_STAMPS
        s_bar += t1 - t0;
        s_cmp += t2 - t1;
        s_pub += t3 - t2;
#endif
    }
#ifdef RISP_CONV_STAMPS
    unsigned long long t_loop_end;
    RISP_STAMP(t_loop_end);
#endif

    // ---- epilogue: lane holds D[cout = cb*32 + (e&3) + 8*(e>>2) + 4*half][x = x0 + l31].
    const int epi = d.epilogue;
    const size_t plane = (size_t)d.H * d.W;
    const float *__restrict__ pbias = d.bias;
    const float *__restrict__ padd = d.add;
    const float *__restrict__ pmask = d.mask;
    float *__restrict__ py = d.y;

    // Vector path: one output row of the wave (CP couts x 32 pixels) is transposed through LDS so that
    // each lane owns 4 consecutive pixels of one cout plane; residual / mask loads and the stores are
    // 16 bytes per lane (a quarter of the memory instructions of the lane-per-pixel form, which made the
    // store tail ~25 % of a workgroup's lifetime).
    const bool vec = (d.W % 4 == 0) && !(epi & RISP_EPI_SHUFFLE2) &&
                     ((reinterpret_cast<uintptr_t>(py) | reinterpret_cast<uintptr_t>(padd) |
                       reinterpret_cast<uintptr_t>(pmask)) & 15) == 0;
    if (vec) {
        constexpr int NV = CP * 8 / 64;                  // float4 per lane per row
        __syncthreads();                                 // every wave is done with the staging tiles
        float *tile = smem + wave * (CP * 32);           // [CP][32] floats, private to the wave
        float bq[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int co = (lane >> 3) + 8 * i;
            bq[i] = 0.f;
            if (!(epi & RISP_EPI_NOBIAS)) bq[i] = pbias[co < d.cout ? co : d.cout - 1];      // uniform branch, clamped address
        }
        const int q4 = 4 * (lane & 7);
        // RISP_EPI_CASEBIAS: a (cout, KS, KS) table per image (d.cvals) indexed by how close the pixel is to the
        // image border - the contribution of spatially constant input channels that were folded out of the layer.
        // Interior tiles see one case only, which joins the bias.
        const bool caseb = (epi & RISP_EPI_CASEBIAS) != 0;
        const float *__restrict__ ctab = d.cvals + (size_t)n * d.cout * TAPS;
        const bool case_edge = caseb && (x0 < PAD || y0 < PAD || x0 + TW > d.W - PAD || y0 + TH > d.H - PAD);
        if (caseb && !case_edge) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int co = (lane >> 3) + 8 * i;
                if (co < d.cout) bq[i] += ctab[co * TAPS + PAD * KS + PAD];
            }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int oy = y0 + wrow + r;
#pragma unroll
            for (int c = 0; c < CB; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    tile[(c * 32 + (e & 3) + 8 * (e >> 2) + 4 * half) * 32 + l31] = acc[r][c][e];
            // the tile is private to this wave and a wave's LDS operations execute in order: no workgroup
            // barrier, only keep the compiler from moving the reads above the writes
            __builtin_amdgcn_wave_barrier();
            const bool row_ok = oy < d.H && x0 + q4 < d.W;
            float4 v[NV], av[NV], mv[NV];
            // Residual and mask values: unconditional loads from clamped (always valid) addresses inside wave-uniform
            // branches, all NV issued before the first use.  Written as per-element "cond ? load : 0" hipcc branches
            // around every load and waits vmcnt(0) behind each - and behind the previous row's stores.
            const size_t pix = row_ok ? (size_t)oy * d.W + x0 + q4 : 0;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                v[i] = *reinterpret_cast<const float4 *>(tile + ((lane >> 3) + 8 * i) * 32 + q4);
                av[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                mv[i] = make_float4(1.f, 1.f, 1.f, 1.f);
            }
            if (epi & RISP_EPI_ADD) {
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int co = (lane >> 3) + 8 * i, cc = co < d.add_c ? co : d.add_c - 1;
                    const float4 a = *reinterpret_cast<const float4 *>(padd + ((size_t)n * d.add_c + cc) * plane + pix);
                    if (co < d.add_c) av[i] = a;
                }
            }
            if (epi & RISP_EPI_MASK) {
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int co = (lane >> 3) + 8 * i, cc = co < d.cout ? co : d.cout - 1;
                    mv[i] = *reinterpret_cast<const float4 *>(pmask + ((size_t)n * d.cout + cc) * plane + pix);
                }
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int co = (lane >> 3) + 8 * i;
                float4 o;
                o.x = v[i].x + bq[i] + av[i].x;
                o.y = v[i].y + bq[i] + av[i].y;
                o.z = v[i].z + bq[i] + av[i].z;
                o.w = v[i].w + bq[i] + av[i].w;
                if (case_edge && row_ok && co < d.cout) {
                    const float *ct = ctab + co * TAPS + border_case(oy, d.H, PAD) * KS;
                    const int ox4 = x0 + q4;
                    o.x += ct[border_case(ox4, d.W, PAD)];
                    o.y += ct[border_case(ox4 + 1, d.W, PAD)];
                    o.z += ct[border_case(ox4 + 2, d.W, PAD)];
                    o.w += ct[border_case(ox4 + 3, d.W, PAD)];
                }
                if (epi & RISP_EPI_RELU) {
                    o.x = o.x > 0.f ? o.x : 0.f;
                    o.y = o.y > 0.f ? o.y : 0.f;
                    o.z = o.z > 0.f ? o.z : 0.f;
                    o.w = o.w > 0.f ? o.w : 0.f;
                }
                o.x = mv[i].x > 0.f ? o.x : 0.f;
                o.y = mv[i].y > 0.f ? o.y : 0.f;
                o.z = mv[i].z > 0.f ? o.z : 0.f;
                o.w = mv[i].w > 0.f ? o.w : 0.f;
                if (row_ok && co < d.cout)
                    *reinterpret_cast<float4 *>(py + ((size_t)n * d.cout + co) * plane + (size_t)oy * d.W + x0 + q4) = o;
            }
            __builtin_amdgcn_wave_barrier();            // tile is rewritten by the next row (same wave, in order)
        }
#ifdef RISP_CONV_STAMPS
        if (lane == 0 && d.mask && !(d.epilogue & RISP_EPI_MASK)) {
            unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<float *>(d.mask)) +
                                    8 * ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave);
            unsigned long long t_end;
            __builtin_amdgcn_s_waitcnt(0x0070);
            RISP_STAMP(t_end);
            o[0] = s_bar; o[1] = s_cmp; o[2] = s_pub;
            o[3] = t_begin - t_k0;
            o[4] = t_loop_end - t_begin;
            o[5] = t_end - t_loop_end;
            o[6] = rt_k0;
            o[7] = __builtin_amdgcn_s_memrealtime();
        }
#endif
        return;
    }

    // Scalar path (PixelShuffle stores, W % 4 != 0): all loads of a row are issued before its stores.
    const int ox = x0 + l31;
    if (ox >= d.W) return;
    float bv[CB][16];
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = c * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
            bv[c][e] = (!(epi & RISP_EPI_NOBIAS) && co < d.cout) ? pbias[co] : 0.f;
        }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int oy = y0 + wrow + r;
        if (oy >= d.H) continue;
        const size_t pix = (size_t)oy * d.W + ox;
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            float av[16], mv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = c * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                av[e] = ((epi & RISP_EPI_ADD) && co < d.add_c) ? padd[((size_t)n * d.add_c + co) * plane + pix] : 0.f;
                mv[e] = ((epi & RISP_EPI_MASK) && co < d.cout) ? pmask[((size_t)n * d.cout + co) * plane + pix] : 1.f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = c * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (co >= d.cout) continue;
                float v = acc[r][c][e] + bv[c][e] + av[e];
                if (epi & RISP_EPI_CASEBIAS)
                    v += d.cvals[((size_t)n * d.cout + co) * TAPS + border_case(oy, d.H, PAD) * KS + border_case(ox, d.W, PAD)];
                if (epi & RISP_EPI_RELU) v = v > 0.f ? v : 0.f;
                v = mv[e] > 0.f ? v : 0.f;
                if (epi & RISP_EPI_SHUFFLE2) {
                    const int cc = co >> 2, i = (co >> 1) & 1, j = co & 1;
                    py[(((size_t)n * (d.cout >> 2) + cc) * (2 * d.H) + 2 * oy + i) * (2 * (size_t)d.W) + 2 * ox + j] = v;
                } else {
                    py[((size_t)n * d.cout + co) * plane + pix] = v;
                }
            }
        }
    }
#ifdef RISP_CONV_STAMPS
    if (lane == 0 && d.mask && !(d.epilogue & RISP_EPI_MASK)) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<float *>(d.mask)) +
                                8 * ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave);
        unsigned long long t_end;
        __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0): the epilogue stores have left the wave
        RISP_STAMP(t_end);
        o[0] = s_bar; o[1] = s_cmp; o[2] = s_pub;
        o[3] = t_begin - t_k0;            // prologue (accumulator init, address setup)
        o[4] = t_loop_end - t_begin;      // chunk loop incl. first fetch/publish
        o[5] = t_end - t_loop_end;        // epilogue
        o[6] = rt_k0;                     // absolute 100 MHz ticks at entry
        o[7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// wpack[chunk][tap][ci_l][co_pad]; zero outside (cin, cout)
__global__ void pack_kernel(const float *__restrict__ w, float *__restrict__ wp, int cin, int cout, int ks, int ck,
                            int cp, int transpose, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int taps = ks * ks;
    const int co = (int)(i % cp);
    size_t t = i / cp;
    const int cl = (int)(t % ck);
    t /= ck;
    const int tap = (int)(t % taps);
    const int chunk = (int)(t / taps);
    const int ci = chunk * ck + cl;
    float v = 0.f;
    if (ci < cin && co < cout) {
        if (!transpose)
            v = w[((size_t)co * cin + ci) * taps + tap];
        else  // w is (cin, cout, k, k) of the forward layer: swap roles, rotate taps by 180 degrees
            v = w[((size_t)ci * cout + co) * taps + (taps - 1 - tap)];
    }
    wp[i] = v;
}

template <int KS, int CK, int CB, bool FAST>
int launch_conv_impl(const risp_conv_desc &d, hipStream_t s) {
    constexpr bool PP = KS != 9;
    constexpr int IH = TH + KS - 1, IW = TW + KS - 1;
    constexpr int IWP = FAST ? (KS == 1 ? 32 : 40) : IW;
    constexpr size_t lds = (PP ? 2 : 1) * ((((size_t)CK * IH * IWP + 3) & ~(size_t)3) + (size_t)KS * KS * CK * 32 * CB) * sizeof(float);
    static_assert(lds <= 64 * 1024, "LDS tile too large");
    dim3 grid((d.W + TW - 1) / TW, (d.H + TH - 1) / TH, d.N);
    hipLaunchKernelGGL((conv_mfma_kernel<KS, CK, CB, FAST, PP>), grid, dim3(256), lds, s, d);
    RISP_LAUNCH_CHECK("risp_conv2d");
    return 0;
}

template <int KS, int CK, int CB>
int launch_conv(const risp_conv_desc &d, hipStream_t s) {
    const bool fast = d.load_mode != RISP_LOAD_UNSHUFFLE2 && d.W % 4 == 0 &&
                      (reinterpret_cast<uintptr_t>(d.x) & 15) == 0;
    return fast ? launch_conv_impl<KS, CK, CB, true>(d, s) : launch_conv_impl<KS, CK, CB, false>(d, s);
}

}  // namespace

extern "C" {

size_t risp_conv_wpack_floats(int cin, int cout, int ksize) {
    const ConvCfg c = conv_cfg(cin, cout, ksize);
    const size_t nchunks = (cin + c.ck - 1) / c.ck;
    return nchunks * ksize * ksize * c.ck * 32 * c.cb;
}

int risp_conv_pack_weights(const float *w, int cin, int cout, int ksize, int transpose, float *wpack, void *stream) {
    RISP_CHECK_ARG(w && wpack && cin > 0 && cout > 0 && cout <= 64 && (ksize == 1 || ksize == 3 || ksize == 5 || ksize == 9),
                   "risp_conv_pack_weights: unsupported layer cin=%d cout=%d k=%d", cin, cout, ksize);
    const ConvCfg c = conv_cfg(cin, cout, ksize);
    const size_t total = risp_conv_wpack_floats(cin, cout, ksize);
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, wpack,
                       cin, cout, ksize, c.ck, 32 * c.cb, transpose, total);
    RISP_LAUNCH_CHECK("risp_conv_pack_weights");
    return 0;
}

int risp_conv2d(const risp_conv_desc *dp, void *stream) {
    RISP_CHECK_ARG(dp, "risp_conv2d: null descriptor");
    const risp_conv_desc &d = *dp;
    RISP_CHECK_ARG(d.x && d.wpack && d.y, "risp_conv2d: null tensor");
    RISP_CHECK_GROUP(d, "risp_conv2d");
    RISP_CHECK_ARG(d.N > 0 && d.N <= 65535 && d.H > 0 && d.W > 0 && d.cin > 0 && d.cout > 0 && d.cout <= 64,
                   "risp_conv2d: bad shape N=%d H=%d W=%d cin=%d cout=%d", d.N, d.H, d.W, d.cin, d.cout);
    RISP_CHECK_ARG((d.epilogue & RISP_EPI_NOBIAS) || d.bias, "risp_conv2d: bias missing");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_ADD) || (d.add && d.add_c > 0), "risp_conv2d: add tensor missing");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_MASK) || d.mask, "risp_conv2d: mask tensor missing");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_SHUFFLE2) || (d.cout % 4 == 0 && !(d.epilogue & (RISP_EPI_MASK))),
                   "risp_conv2d: PixelShuffle store needs cout %% 4 == 0 and no mask");
    RISP_CHECK_ARG(d.load_mode != RISP_LOAD_UNSHUFFLE2 || d.cin % 4 == 0, "risp_conv2d: unshuffle load needs cin %% 4 == 0");
    RISP_CHECK_ARG(d.load_mode != RISP_LOAD_CONSTCH || (d.cvals && d.cin_img > 0 && d.cin_img <= d.cin),
                   "risp_conv2d: const-channel load needs cvals and cin_img");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_CASEBIAS) ||
                       (d.cvals && d.load_mode != RISP_LOAD_CONSTCH && d.H >= 2 * (d.ksize / 2) && d.W >= 2 * (d.ksize / 2)),
                   "risp_conv2d: border-case bias needs its table in cvals, a non-const-channel load and H, W >= k - 1");
    const ConvCfg c = conv_cfg(d.cin, d.cout, d.ksize);
    hipStream_t s = (hipStream_t)stream;
#define RISP_CONV_CASE(KS, CK, CB) \
    if (d.ksize == KS && c.ck == CK && c.cb == CB) return launch_conv<KS, CK, CB>(d, s);
    RISP_CONV_CASE(3, 4, 2)
    RISP_CONV_CASE(3, 4, 1)
    RISP_CONV_CASE(5, 4, 1)
    RISP_CONV_CASE(5, 2, 2)
    RISP_CONV_CASE(9, 2, 2)
    RISP_CONV_CASE(9, 2, 1)
    RISP_CONV_CASE(1, 8, 2)
    RISP_CONV_CASE(1, 8, 1)
#undef RISP_CONV_CASE
    risp_set_error("risp_conv2d: unsupported kernel size %d", d.ksize);
    return 1;
}

}  // extern "C"
