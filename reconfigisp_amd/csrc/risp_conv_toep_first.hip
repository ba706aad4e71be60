// Split-precision convolution for the 9x9 FIRST layers of the proxies - few input channels, 64 output channels - on the f16
// matrix pipe (round 4):
//   9x9,  3 -> 64   SRCNNRes once its 9+P broadcast planes are folded out (srcnn_res_arch.py:18, 41-46; SrcnnResFold)
//   9x9,  4 -> 64   SRCNNDemosaic on the space-to-depth mosaic (srcnn_demosaic_arch.py:14-16, 39-43)
// With 3 input channels the reduction index of a matrix instruction cannot be channels.  As in risp_conv_toep.hip it is a WINDOW
// of 16 input pixels of one channel and one filter row, the columns are the 32 blocks of 8 pixels of a 256-pixel row; here the
// rows are 32 OUTPUT CHANNELS and each of the 8 pixel positions j of a block has its own accumulator:
//        D_j[co][b] = sum_u A_j[co][u] B[u][b],   B[u][b] = x[ci][y + ky - 4][8 b - 4 + u],   A_j[co][u] = w[co][ci][ky][u - j].
// The 8 operands A_j are 8-slot windows, one slot apart, of ONE zero-padded filter row per lane: the lane reads its row once
// (two 16-byte LDS reads per part), even shifts are register pairs as they lie, odd shifts four v_alignbit each - 6 LDS reads
// and 32 vector instructions per 24 matrix instructions.  9 of the 16 reduction slots carry a tap (the fp32 matrix-core kernel
// risp_conv2d_k3 uses 243 of 244 - at a sixteenth of the rate).  Arithmetic as in risp_conv_f16x2.hip: two f16 halves per fp32
// operand, three products, fp32 accumulation; weights split at pack time, activations per workgroup tile and channel.
//
// Kernel.  Persistent workgroups (2 per CU) of 4 waves; a tile = 4 rows x 256 pixels x 32 couts, a wave owns one row (8
// accumulators of 16 registers).  A chunk = one input channel: halo tile (12 rows x 264 pixels) staged through registers as f16
// rows (slot = 8 pixels, stored 4 pixels in), the channel's 9 filter rows of the cout block (18 KB: per cout the taps 0-7 and the
// ninth tap in two 16-byte slots, hi and lo) by LDS-DMA one channel ahead.  216 matrix instructions between two barriers.
// Epilogue: bias, border-case table (RISP_EPI_CASEBIAS, staged in LDS per tile), ReLU; a lane holds 8 consecutive pixels of a
// cout: two 16-byte stores, a wave-instruction pair = 1 KB of a cout row.
#include "risp_f16x2.h"

namespace {
constexpr int TF_KS = 9, TF_P = 4;
constexpr int TF_WST = TF_KS * 2 * 2 * 32;                           // weight slots of (cout block, ci): [ky][part][taps 0-7 | tap 8][cout]
constexpr int TF_PW = (TF_WST / 64 + 3) / 4;
static_assert(TF_WST % 64 == 0, "weights in whole LDS-DMA pieces");
// Tile geometry.  The 32 columns of a matrix instruction are FOLD rows x 32 / FOLD blocks of 8 pixels: one 256-pixel strip of a
// row, or - planes of at most 128 pixels, the Bayer-domain proxies on 256 x 256 patches - two rows of a 128-pixel strip, so that
// narrow planes do not leave half of every instruction idle.  A wave owns FOLD rows, a tile 4 FOLD.
template <int FOLD>
struct TFG {
    // 16-byte slots and 4-pixel quads of a staged row (folded: 32 slots, not 17 - the lanes of the second row must sit a multiple of
    // 256 bytes from those of the first for a conflict-free 16-byte LDS read, tools/lds_bank_probe.hip)
    static constexpr int NBLK = 32 / FOLD, TW = 8 * NBLK, RS = FOLD == 1 ? NBLK + 1 : 32, Q = 2 * NBLK + 2;
    static constexpr int TH = 4 * FOLD, IH = TH + 2 * TF_P;
    static constexpr int PART = IH * RS, TILE = 2 * PART;            // 16-byte slots of the staged channel: hi part, lo part
    static constexpr int NTASK = (IH * Q + 255) / 256;
    static constexpr int CT = TH * 32 * TF_KS;                       // floats of the border-case table of a tile: [row][cout][x case]
    static constexpr int LDS_BYTES = (TILE + 2 * TF_WST + 1) * 16 + 64 + 2 * CT * 4;      // ... + the border-case table, twice (by tile parity)
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

__device__ __forceinline__ int tf_border_case(int v, int L) { return v < TF_P ? v : (v >= L - TF_P ? 2 * TF_P - (L - 1 - v) : TF_P); }

// UNSHUF: x is the (N, 1, 2H, 2W) mosaic, input channel 2i+j = x[2y+i][2x+j] (RISP_LOAD_UNSHUFFLE2, cin == 4)
template <bool UNSHUF, bool CASEB, int FOLD>
__global__ __launch_bounds__(256, 2) void conv_toep_first_kernel(const risp_conv_desc d, int tiles_x, int tiles_y, int ncb, int ntiles,
                                                                 unsigned *__restrict__ ties, unsigned max_ties) {
    using G = TFG<FOLD>;
    constexpr int P = TF_P, KS = TF_KS, IH = G::IH, RS = G::RS, WST = TF_WST, PW = TF_PW, NTASK = G::NTASK, NV = UNSHUF ? 2 : 1;
    constexpr int TF_TILE = G::TILE, TF_PART = G::PART, TF_Q = G::Q, TF_CT = G::CT, NBLK = G::NBLK;
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    uint4 *tile = smem, *wl = smem + TF_TILE, *zero = wl + 2 * WST;
    float *red = reinterpret_cast<float *>(zero + 1);
    float *ctl = red + 16;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hl = lane >> 5;
    const size_t hw = (size_t)d.H * d.W;
    const unsigned hw4 = (unsigned)hw * 4u;
    const int nwg = gridDim.x;
    const int wg = (nwg & 7) == 0 ? (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    if (tid == 0) *zero = make_uint4(0u, 0u, 0u, 0u);

    auto task_id = [&](int k) { return tid + 256 * k < IH * TF_Q ? tid + 256 * k : tid + 256 * (k - 1); };
    static_assert((NTASK - 1) * 256 <= IH * TF_Q, "only the last task may be missing");
    int dst[NTASK];
    unsigned off[NTASK];
    bool ok[NTASK];
#pragma unroll
    for (int k = 0; k < NTASK; ++k) {
        const int id = task_id(k), row = id / TF_Q, q = id - row * TF_Q;
        dst[k] = (row * RS + (q >> 1)) * 16 + (q & 1) * 8;
    }
    struct TileRef {
        int n, cb, x0, y0;
        const uint4 *w;
    };
    TileRef cur;
    __amdgpu_buffer_rsrc_t rx;
    auto locate = [&](int t, TileRef &r) {
        r.cb = t % ncb;
        const int q0 = t / ncb, tx = q0 % tiles_x, q = q0 / tiles_x, ty = q % tiles_y;
        r.n = q / tiles_y;
        r.x0 = tx * G::TW;
        r.y0 = ty * G::TH;
        const int g = d.group_n > 0 ? r.n / d.group_n : 0;
        r.w = reinterpret_cast<const uint4 *>(d.wpack + (size_t)g * d.wpack_gs);
    };
    auto setup = [&](const TileRef &r) {
        const int g = d.group_n > 0 ? r.n / d.group_n : 0;
        const int nx = (d.group_flags & RISP_GROUP_SHARED_X) ? r.n - g * d.group_n : r.n;
        rx = h2_rsrc(d.x + (size_t)nx * d.cin * hw);
#pragma unroll
        for (int k = 0; k < NTASK; ++k) {
            const int id = task_id(k), row = id / TF_Q, q = id - row * TF_Q;
            const int gy = r.y0 - P + row, gx = r.x0 - 4 + 4 * q;
            ok[k] = gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
            // plain: byte offset inside a plane; mosaic: of element (2 gy, 2 gx) of the (2H, 2W) image
            off[k] = ok[k] ? (UNSHUF ? 4u * (unsigned)(2 * gy * 2 * d.W + 2 * gx) : 4u * (unsigned)(gy * d.W + gx)) : 0u;
        }
    };
    float4 v[NTASK][NV];
    auto fetch = [&](int ci) {
        // mosaic: row 2 gy + i, 8 elements from column 2 gx - the quad of plane (i, j) is elements j, j + 2, j + 4, j + 6
        const unsigned so = UNSHUF ? (unsigned)(ci >> 1) * 8u * (unsigned)d.W : (unsigned)ci * hw4;
#pragma unroll
        for (int k = 0; k < NTASK; ++k) {
            v[k][0] = h2_load16(rx, off[k], so);
            if (UNSHUF) v[k][NV - 1] = h2_load16(rx, off[k], so + 16u);
        }
    };
    auto quad = [&](int k, int ci) {
        if constexpr (UNSHUF) {
            const float4 a = v[k][0], b = v[k][NV - 1];
            return (ci & 1) ? make_float4(a.y, a.w, b.y, b.w) : make_float4(a.x, a.z, b.x, b.z);
        } else {
            return v[k][0];
        }
    };
    unsigned wvoff[PW], wlds[PW];                                     // lane offset in the pack; LDS byte address in buffer 0 (scalar)
    const unsigned lds_wl = lds_addr_of(wl);
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        const int piece = (wave + 4 * p) % (WST / 64);
        wvoff[p] = 16u * (unsigned)(piece * 64 + lane);
        wlds[p] = lds_wl + 16u * (unsigned)(piece * 64);
    }
    auto issue_weights = [&](int ci, int slot, const TileRef &r) {
        const uint4 *src = r.w + 1 + ((size_t)r.cb * d.cin + ci) * WST;          // slot 0 of the pack = header
#pragma unroll
        for (int p = 0; p < PW; ++p) lds_dma16_m(src, wvoff[p], wlds[p] + (unsigned)slot * (WST * 16u));
    };
    // operands.  B: lane (b, half) of input row r reads slot b + half of that row.  A: the lane's zero-padded filter row
    // R[0..15] = Wp[8 half ..], Wp = (8 zeros, taps 0-8, zeros): slots (zero, taps 0-7) for half 0, (taps 0-7, tap 8) for half 1.
    const int srow = FOLD * wave + l31 / NBLK, sblk = l31 % NBLK;     // the lane's row inside the tile and its pixel block
    const int bbase = srow * RS + sblk + hl;
    // slot indices relative to smem; (ring, ky, part) adds ring * WST + (ky * 2 + part) * 64 to the weight slots
    constexpr int ZI = TF_TILE + 2 * WST;
    const int a1 = TF_TILE + (hl ? 32 + l31 : l31);

    int t_cur = wg;
    if (t_cur >= ntiles) return;
    locate(t_cur, cur);
    setup(cur);
    issue_weights(0, 0, cur);
    fetch(0);
    int ring = 0, parity = 0;
    for (;;) {
        f32x16 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        int se = 0;
        float bsum = 0.f;                              // sum over the channels of the tile's largest input magnitude (see `ties`)
        const int t_next = t_cur + nwg;
        const bool more = t_next < ntiles;
        TileRef nxt = cur;
        if (more) locate(t_next, nxt);
        // border-case table of this tile: [row][cout of the block][x case], read in the epilogue.  Two buffers by tile parity: the
        // waves still in the previous tile's epilogue read the other one.
        parity ^= 1;
        if constexpr (CASEB) {
            const float *ctab = d.cvals + (size_t)cur.n * d.cout * (KS * KS);
            float creg[(TF_CT + 255) / 256];
#pragma unroll
            for (int i = 0; i < (TF_CT + 255) / 256; ++i) {
                const int idx = tid + 256 * i, r = idx / (32 * KS), rem = idx - r * (32 * KS), co = rem / KS, kx = rem - co * KS;
                const int oy = cur.y0 + r, cog = cur.cb * 32 + co;
                creg[i] = (idx < TF_CT && oy < d.H && cog < d.cout) ? ctab[(size_t)cog * (KS * KS) + tf_border_case(oy, d.H) * KS + kx] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < (TF_CT + 255) / 256; ++i)
                if (tid + 256 * i < TF_CT) ctl[parity * TF_CT + tid + 256 * i] = creg[i];
        }
        for (int ci = 0; ci < d.cin; ++ci) {
            float m = 0.f;
#pragma unroll
            for (int k = 0; k < NTASK; ++k)
                if (ok[k]) m = amax4(m, quad(k, ci));
            m = h2_wave_max(m);
            if (lane == 0) red[wave] = m;
            __syncthreads();                           // A
            const float4 mx = *reinterpret_cast<const float4 *>(red);
            const float tmax = fmaxf(fmaxf(mx.x, mx.y), fmaxf(mx.z, mx.w));
            bsum += tmax;
            int eb = (int)(__builtin_bit_cast(unsigned, tmax) >> 23);
            eb = __builtin_amdgcn_readfirstlane(eb);
            int want = 141 - eb;
            want = want > 100 ? 100 : want;
            if (ci == 0) {
                se = want;
            } else if (want < se) {
                const int fe = 127 + want - se;
                const float f = fe > 0 ? __builtin_bit_cast(float, (unsigned)fe << 23) : 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[j][e] *= f;
                se = want;
            }
            const float s = __builtin_bit_cast(float, (unsigned)(127 + se) << 23);
#pragma unroll
            for (int k = 0; k < NTASK; ++k) {
                const float4 q4 = quad(k, ci);
                const float b0 = ok[k] ? q4.x * s : 0.f, b1 = ok[k] ? q4.y * s : 0.f, b2 = ok[k] ? q4.z * s : 0.f, b3 = ok[k] ? q4.w * s : 0.f;
                const h2 h01 = {(_Float16)b0, (_Float16)b1}, h23 = {(_Float16)b2, (_Float16)b3};
                const h2 l01 = {(_Float16)(b0 - (float)h01[0]), (_Float16)(b1 - (float)h01[1])};
                const h2 l23 = {(_Float16)(b2 - (float)h23[0]), (_Float16)(b3 - (float)h23[1])};
                char *base = reinterpret_cast<char *>(tile) + dst[k];
                *reinterpret_cast<uint2 *>(base) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
                *reinterpret_cast<uint2 *>(base + TF_PART * 16) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
            }
            H2_WAIT_VM(0);
            __syncthreads();                           // B
            if (ci + 1 < d.cin) {
                issue_weights(ci + 1, ring ^ 1, cur);
                fetch(ci + 1);
            } else if (more) {
                issue_weights(0, ring ^ 1, nxt);
                setup(nxt);
                fetch(0);
            }
            // ---- matrix phase: 9 filter rows x 8 positions x 3 products
            const int wo = ring * WST;
            const uint4 *ts = tile + bbase;
            u32x4 ra[2][2];                             // [part][low / high half of the lane's 16-slot row]
            h8 bv[2][2];
            auto load_a = [&](int ky) {
#pragma unroll
                for (int part = 0; part < 2; ++part) {
                    const int o = wo + (ky * 2 + part) * 64;
                    ra[part][0] = __builtin_bit_cast(u32x4, smem[hl ? TF_TILE + o + l31 : ZI]);
                    ra[part][1] = __builtin_bit_cast(u32x4, smem[a1 + o]);
                }
            };
            auto load_b = [&](int ky, int buf) {
                bv[buf][0] = __builtin_bit_cast(h8, ts[ky * RS]);
                bv[buf][1] = __builtin_bit_cast(h8, ts[TF_PART + ky * RS]);
            };
            load_a(0);
            load_b(0, 0);
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int buf = ky & 1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // A_j = slots 8 - j .. 15 - j of the row: dwords (8 - j) / 2 .., shifted by 16 bits when 8 - j is odd
                    const int s0 = 8 - j, dw = s0 >> 1;
                    h8 ap[2];
#pragma unroll
                    for (int part = 0; part < 2; ++part) {
                        unsigned r8[8];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            r8[i] = ra[part][0][i];
                            r8[4 + i] = ra[part][1][i];
                        }
                        u32x4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            o[i] = (s0 & 1) ? __builtin_amdgcn_alignbit(r8[dw + i + 1], r8[dw + i], 16) : r8[dw + i];
                        ap[part] = __builtin_bit_cast(h8, o);
                    }
                    if (j == 7 && ky + 1 < KS) {
                        // the row of the next filter row replaces this one as soon as its last window is formed: its reads run
                        // under the three products below (one register set instead of two: the accumulators take 128 of the 256)
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        load_a(ky + 1);
                        load_b(ky + 1, buf ^ 1);
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[0], bv[buf][1], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[1], bv[buf][0], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[0], bv[buf][0], acc[j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            ring ^= 1;
        }
        // ---- epilogue.  Lane (b, half), accumulator j, element e: cout 32 cb + 8 (e >> 2) + 4 half + (e & 3), pixel 8 b + j
        {
            const int g = d.group_n > 0 ? cur.n / d.group_n : 0;
            const float inv_sw = *reinterpret_cast<const float *>(cur.w);
            const float fin = inv_sw * __builtin_bit_cast(float, (unsigned)(127 - se) << 23);
            const int epi = d.epilogue;
            const float floor_ = (epi & RISP_EPI_RELU) ? 0.f : -__builtin_inff();
            // |error| of a sum here <~ 2^-22 x (sum of |terms|) <= 2^-22 x 81 x max|w| x bsum; inv_sw 2^15 >= max|w|: a margin of 2 more bits
            const float tau = bsum * inv_sw * (81.f * 32768.f / 1048576.f);
            const int oy = cur.y0 + srow, ox = cur.x0 + 8 * sblk;
            const float *bias = (epi & RISP_EPI_NOBIAS) ? nullptr : d.bias + (size_t)g * d.bias_gs;
            int cxo[8];
            if constexpr (CASEB) {
#pragma unroll
                for (int j = 0; j < 8; ++j) cxo[j] = tf_border_case(ox + j < d.W ? ox + j : d.W - 1, d.W);
            }
            if (oy < d.H && ox < d.W) {
                const size_t pix = (size_t)oy * d.W + ox;
                const bool second = ox + 4 < d.W;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int cl = 8 * (e >> 2) + 4 * hl + (e & 3), co = cur.cb * 32 + cl;
                    if (co < d.cout) {
                        const float bb = bias ? bias[co] : 0.f;
                        float o[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            o[j] = acc[j][e] * fin + bb;
                            if constexpr (CASEB) o[j] += ctl[parity * TF_CT + (srow * 32 + cl) * KS + cxo[j]];
                            // a pre-activation this close to zero has no reliable sign in fp32: listed for toep_first_ties_kernel
                            if (ties && fabsf(o[j]) < tau && ox + j < d.W) {
                                const unsigned slot = atomicAdd(ties, 1u);
                                if (slot < max_ties) ties[1 + slot] = (unsigned)((((size_t)cur.n * d.cout + co) * hw + pix + j));
                            }
                            o[j] = o[j] < floor_ ? floor_ : o[j];
                        }
                        float *yp = d.y + ((size_t)cur.n * d.cout + co) * hw + pix;
                        *reinterpret_cast<float4 *>(yp) = make_float4(o[0], o[1], o[2], o[3]);
                        if (second) *reinterpret_cast<float4 *>(yp + 4) = make_float4(o[4], o[5], o[6], o[7]);
                    }
                    __builtin_amdgcn_sched_barrier(0);      // (hipcc would hoist the table reads of all 16 couts and spill them)
                }
            }
        }
        if (!more) break;
        cur = nxt;
        t_cur = t_next;
    }
}

// The outputs listed by conv_toep_first_kernel (pre-activation within the arithmetic's own error of zero), recomputed EXACTLY: every
// product in double (exact), summed in double in a fixed order, bias and border-case value added in double, rounded once.  One
// wave per listed output.  ReLU decisions of the layer then no longer depend on the summation order of whichever fp32-accurate
// kernel computed them (tools/dbg_flips.py: one such activation of the DARTS golden scenario moved iteration 1 by 9e-4).
template <bool UNSHUF>
__global__ __launch_bounds__(256) void toep_first_ties_kernel(const risp_conv_desc d, const float *__restrict__ w32, long long w32_gs,
                                                              const unsigned *__restrict__ ties, unsigned max_ties) {
    constexpr int KS = TF_KS, P = TF_P;
    const unsigned count = ties[0] < max_ties ? ties[0] : max_ties;
    const int lane = threadIdx.x & 63;
    const size_t hw = (size_t)d.H * d.W;
    const int K = d.cin * KS * KS;
    for (unsigned t = blockIdx.x * 4 + (threadIdx.x >> 6); t < count; t += gridDim.x * 4) {
        const size_t idx = ties[1 + t];
        const int n = (int)(idx / ((size_t)d.cout * hw)), co = (int)(idx / hw) - n * d.cout;
        const size_t pix = idx - ((size_t)n * d.cout + co) * hw;
        const int y = (int)(pix / d.W), x = (int)(pix - (size_t)y * d.W);
        const int g = d.group_n > 0 ? n / d.group_n : 0;
        const int nx = (d.group_flags & RISP_GROUP_SHARED_X) ? n - g * d.group_n : n;
        const float *xn = d.x + (size_t)nx * d.cin * hw;
        const float *wr = w32 + (size_t)g * w32_gs + (size_t)co * K;
        double a = 0.0;
        for (int k = lane; k < K; k += 64) {
            const int ci = k / (KS * KS), r = k - ci * (KS * KS), ky = r / KS, kx = r - ky * KS;
            const int sy = y + ky - P, sx = x + kx - P;
            if (sy >= 0 && sy < d.H && sx >= 0 && sx < d.W) {
                // mosaic: channel 2 i + j of plane pixel (sy, sx) = element (2 sy + i, 2 sx + j) of the (2H, 2W) image
                const float xv = UNSHUF ? xn[(size_t)(2 * sy + (ci >> 1)) * (2 * d.W) + 2 * sx + (ci & 1)] : xn[(size_t)ci * hw + (size_t)sy * d.W + sx];
                a += (double)xv * (double)wr[k];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if (lane == 0) {
            if (!(d.epilogue & RISP_EPI_NOBIAS)) a += (double)d.bias[(size_t)g * d.bias_gs + co];
            if (d.epilogue & RISP_EPI_CASEBIAS)
                a += (double)d.cvals[((size_t)n * d.cout + co) * (KS * KS) + tf_border_case(y, d.H) * KS + tf_border_case(x, d.W)];
            float o = (float)a;
            if ((d.epilogue & RISP_EPI_RELU) && !(o > 0.f)) o = 0.f;
            d.y[idx] = o;
        }
    }
}

#ifndef RISP_TF_WGS
#define RISP_TF_WGS 2
#endif

template <bool UNSHUF, bool CASEB, int FOLD>
int launch_toep_first_g(const risp_conv_desc &d, unsigned *ties, unsigned max_ties, void *stream) {
    using G = TFG<FOLD>;
    constexpr int TF_LDS_BYTES = G::LDS_BYTES;
    auto kern = &conv_toep_first_kernel<UNSHUF, CASEB, FOLD>;
    if (TF_LDS_BYTES > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, TF_LDS_BYTES) != hipSuccess) {
        risp_set_error("risp_conv2d_toep_first: cannot raise the dynamic LDS limit to %d bytes", TF_LDS_BYTES);
        return 2;
    }
    const int tx = (d.W + G::TW - 1) / G::TW, ty = (d.H + G::TH - 1) / G::TH, ncb = (d.cout + 31) / 32;
    const long long ntiles = (long long)tx * ty * d.N * ncb;
    if (ntiles > 0x7fffffff) {
        risp_set_error("risp_conv2d_toep_first: too many tiles");
        return 1;
    }
    const int slots = RISP_TF_WGS * h2_cu_count();
    const int grid = ntiles < slots ? (int)ntiles : slots;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), TF_LDS_BYTES, (hipStream_t)stream, d, tx, ty, ncb, (int)ntiles, ties, max_ties);
    RISP_LAUNCH_CHECK("risp_conv2d_toep_first");
    return 0;
}
template <bool UNSHUF, bool CASEB>
int launch_toep_first(const risp_conv_desc &d, unsigned *ties, unsigned max_ties, void *stream) {
    return d.W <= 128 ? launch_toep_first_g<UNSHUF, CASEB, 2>(d, ties, max_ties, stream) : launch_toep_first_g<UNSHUF, CASEB, 1>(d, ties, max_ties, stream);
}
}  // namespace

extern "C" {

size_t risp_conv_toep_first_wpack_bytes(int cin, int cout) { return 16 + (size_t)((cout + 31) / 32) * cin * TF_WST * 16; }

}  // extern "C"
int risp_launch_xwin(const risp_conv_desc &d, unsigned *ties, unsigned max_ties, void *stream);      // risp_conv_xwin.hip
extern "C" {

static int toep_first_impl(const risp_conv_desc *dp, unsigned *ties, unsigned max_ties, void *stream) {
    RISP_CHECK_ARG(dp, "risp_conv2d_toep_first: null descriptor");
    const risp_conv_desc &d = *dp;
    RISP_CHECK_ARG(d.x && d.wpack && d.y, "risp_conv2d_toep_first: null tensor");
    RISP_CHECK_GROUP(d, "risp_conv2d_toep_first");
    const bool unshuf = d.load_mode == RISP_LOAD_UNSHUFFLE2;
    RISP_CHECK_ARG(d.load_mode == RISP_LOAD_PLAIN || (unshuf && d.cin == 4), "risp_conv2d_toep_first: plain loads, or the mosaic with cin == 4");
    RISP_CHECK_ARG(d.N > 0 && d.H > 0 && d.W > 0 && d.W % 4 == 0 && d.cin > 0 && d.cin <= 16 && d.cout > 0 && d.ksize == TF_KS &&
                       (unsigned long long)d.cin * d.H * d.W < (1ull << 30),
                   "risp_conv2d_toep_first: needs a 9x9 layer with cin <= 16, W %% 4 == 0, fewer than 2^30 input elements per image "
                   "(N=%d H=%d W=%d cin=%d cout=%d k=%d)",
                   d.N, d.H, d.W, d.cin, d.cout, d.ksize);
    RISP_CHECK_ARG(!(d.epilogue & ~(RISP_EPI_RELU | RISP_EPI_NOBIAS | RISP_EPI_CASEBIAS)), "risp_conv2d_toep_first: epilogue %d not supported",
                   d.epilogue);
    RISP_CHECK_ARG((d.epilogue & RISP_EPI_NOBIAS) || d.bias, "risp_conv2d_toep_first: bias missing");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_CASEBIAS) || (d.cvals && d.H >= d.ksize - 1 && d.W >= d.ksize - 1),
                   "risp_conv2d_toep_first: border-case bias needs its table in cvals and H, W >= k - 1");
    RISP_CHECK_ARG(((reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.y) | reinterpret_cast<uintptr_t>(d.wpack)) & 15) == 0,
                   "risp_conv2d_toep_first: tensors must be 16-byte aligned");
    const bool cb = (d.epilogue & RISP_EPI_CASEBIAS) != 0;
    // 3 plain input channels: the (channel, tap) reduction index of risp_conv_xwin.hip - two matrix steps per filter row instead of three
#ifndef RISP_XWIN_OFF                                        /* (A/B builds: tools/ab_xwin.py) */
    if (!unshuf && d.cin == 3 && d.cout <= 64 && (unsigned long long)d.cout * d.H * d.W < (1ull << 29)) return risp_launch_xwin(d, ties, max_ties, stream);
#endif
    if (unshuf) return cb ? launch_toep_first<true, true>(d, ties, max_ties, stream) : launch_toep_first<true, false>(d, ties, max_ties, stream);
    return cb ? launch_toep_first<false, true>(d, ties, max_ties, stream) : launch_toep_first<false, false>(d, ties, max_ties, stream);
}

int risp_conv2d_toep_first(const risp_conv_desc *dp, void *stream) { return toep_first_impl(dp, nullptr, 0, stream); }

/* The same layer with EXACT ReLU decisions (training forwards): outputs whose pre-activation lies within the arithmetic's own error
 * of zero are listed (ties: [0] = count, then up to max_ties output indices; zeroed here) and recomputed in double by a second
 * launch from the layer's fp32 weights w32 (cout, cin, 9, 9) - members of a grouped launch w32_gs floats apart. */
int risp_conv2d_toep_first_exact(const risp_conv_desc *dp, const float *w32, long long w32_gs, unsigned *ties, unsigned max_ties, void *stream) {
    RISP_CHECK_ARG(dp && w32 && ties && max_ties > 0 && w32_gs >= 0, "risp_conv2d_toep_first_exact: needs the fp32 weights and the tie list");
    RISP_CHECK_ARG((unsigned long long)dp->N * dp->cout * dp->H * dp->W < (1ull << 32), "risp_conv2d_toep_first_exact: more than 2^32 outputs");
    if (hipMemsetAsync(ties, 0, sizeof(unsigned), (hipStream_t)stream) != hipSuccess) {
        risp_set_error("risp_conv2d_toep_first_exact: cannot clear the tie counter");
        return 2;
    }
    const int rc = toep_first_impl(dp, ties, max_ties, stream);
    if (rc) return rc;
    if (dp->load_mode == RISP_LOAD_UNSHUFFLE2)
        hipLaunchKernelGGL(toep_first_ties_kernel<true>, dim3(512), dim3(256), 0, (hipStream_t)stream, *dp, w32, w32_gs, ties, max_ties);
    else
        hipLaunchKernelGGL(toep_first_ties_kernel<false>, dim3(512), dim3(256), 0, (hipStream_t)stream, *dp, w32, w32_gs, ties, max_ties);
    RISP_LAUNCH_CHECK("risp_conv2d_toep_first_exact");
    return 0;
}

}  // extern "C"
