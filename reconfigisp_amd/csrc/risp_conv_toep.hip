// Split-precision convolution for layers with FEW output channels and a wide filter row, on the f16 matrix pipe (round 4):
//   * SRCNNRes conv 9x9 (12+P) -> 64 (srcnn_res_arch.py:18), backward-data restricted to the 3 image channels   (64 -> 3)
//   * SRCNNRes conv 5x5 32 -> 3 (srcnn_res_arch.py:22), forward
//   * SRCNNDemosaic conv 9x9 4 -> 64 backward-data through PixelShuffle (srcnn_demosaic_arch.py:14-16)          (64 -> 4)
// The matrix instruction wants 32 rows of outputs; a 3-cout layer has 3.  Here the rows are (cout, j) with j = the pixel's
// position inside a block of 8 pixels, the columns are the 32 blocks of a 256-pixel row, and the reduction index is a
// WINDOW of 16 input pixels of one input channel and one filter row:
//        D[(co, j)][b] = sum_u A[(co, j)][u] B[u][b],   B[u][b] = x[ci][y + ky - P][8 b - 4 + u],
//        A[(co, j)][u] = w[co][ci][ky][u - j + P - 4]  (0 outside the filter row)                      - a Toeplitz band.
// 4 couts x 8 positions fill the 32 rows; of the 16 reduction slots a 9-tap row uses 9 and a 5-tap row 5 - against 3 of 32 rows
// when couts alone index the rows.  One v_mfma_f32_32x32x16_f16 = one (ci, ky) of one 256-pixel output row.  Arithmetic as in
// risp_conv_f16x2.hip: two f16 halves per fp32 operand, three products, fp32 accumulation; the weights (the bands, hi and lo) are
// packed once, the activations are scaled per workgroup tile and input channel by the tile's own largest magnitude.
//
// Kernel.  Persistent workgroups (2 per CU) of 4 waves; a tile = 4 T rows x 256 pixels, a wave owns T consecutive rows (T
// accumulators of 16 registers).  A chunk = ONE input channel: its halo tile (4 T + 2 P rows x 264 pixels) is staged through
// registers as f16 rows - a 16-byte LDS slot = 8 consecutive pixels = one lane's B operand, no shifting at read time because
// the window of block b starts at pixel 8 b - 4 and the rows are stored 4 pixels in - , its KS bands (2 KB each) arrive by
// LDS-DMA one channel ahead.  Within a chunk the wave walks the T + KS - 1 input rows it needs: each row's operand is read
// once and serves every (output row t, filter row ky) pair with t + ky = row; the KS band operands stay in registers.
// Two barriers per chunk, 3 KS T matrix instructions (108 for 9x9 at T = 4) between them.
#include "risp_f16x2.h"

namespace {
// Tile geometry.  The 32 columns of a matrix instruction are FOLD rows x 32 / FOLD blocks of 8 pixels: one 256-pixel strip of a row,
// or - planes of at most 128 pixels - two rows of a 128-pixel strip (a wave then owns 2 T rows, a tile 8 T).
template <int KS, int NB, int T, int FOLD>
struct TP {
    static constexpr int P = KS / 2, NBLK = 32 / FOLD, TW = 8 * NBLK, TH = 4 * T * FOLD, IH = TH + 2 * P;
    // 16-byte slots and 4-pixel quads of a staged row (pixels x0 - 4 .. x0 + TW + 3); folded: 32 slots per row, so that the lanes of the
    // second row sit a multiple of 256 bytes from those of the first (conflict-free 16-byte LDS reads, tools/lds_bank_probe.hip)
    static constexpr int RS = FOLD == 1 ? NBLK + 1 : 32, Q = 2 * NBLK + 2;
    static constexpr int PART = IH * RS;                          // slots of one part (hi or lo) of the staged channel
    static constexpr int TILE = 2 * PART;
    static constexpr int WST = KS * 2 * 2 * NB * 32;              // band slots of one input channel: [ky][part][window half][row (cout, j)]
    static constexpr int PW = (WST / 64 + 3) / 4;                 // LDS-DMA instructions per wave and channel
    static constexpr int NTASK = (IH * Q + 255) / 256;            // staging tasks (row, quad) per thread
    static constexpr int LDS_BYTES = (TILE + 2 * WST) * 16 + 64;  // tile, two band buffers, the row of maxima
    static_assert(WST % 64 == 0, "bands in whole LDS-DMA pieces");
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

// SHUF: PixelShuffle(2) store (cout % 4 == 0).  HAS_ADD: y += add[:, :add_c].
template <int KS, int NB, int T, int FOLD, bool HAS_ADD, bool SHUF>
__global__ __launch_bounds__(256, 2) void conv_toep_kernel(const risp_conv_desc d, int tiles_x, int tiles_y, int ntiles, float *__restrict__ psum) {
    using C = TP<KS, NB, T, FOLD>;
    constexpr int P = C::P, IH = C::IH, RS = C::RS, WST = C::WST, PW = C::PW, NTASK = C::NTASK, TP_Q = C::Q, TP_TW = C::TW, NBLK = C::NBLK;
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    uint4 *tile = smem, *wl = smem + C::TILE;
    float *red = reinterpret_cast<float *>(wl + 2 * WST);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hl = lane >> 5;
    const size_t hw = (size_t)d.H * d.W;
    const unsigned hw4 = (unsigned)hw * 4u;                           // bytes of a plane (cin * H * W < 2^30: checked by the entry point)
    const int nwg = gridDim.x;
    const int wg = (nwg & 7) == 0 ? (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3) : blockIdx.x;

    // staging tasks: (tile row, quad of 4 pixels) -> one 16-byte load, 8 bytes of hi + 8 bytes of lo
    int dst[NTASK];                                                   // byte offset into the hi part
    unsigned off[NTASK];
    bool ok[NTASK];
    // (a thread without an NTASK-th task of its own repeats its previous one - same bytes to the same place - so that the staging
    // code has no branches)
    auto task_id = [&](int k) { return tid + 256 * k < IH * TP_Q ? tid + 256 * k : tid + 256 * (k - 1); };
    static_assert((NTASK - 1) * 256 <= IH * TP_Q, "only the last task may be missing");
    bool inner[NTASK];                                                 // the task's quad belongs to the tile itself (not to its halo)
#pragma unroll
    for (int k = 0; k < NTASK; ++k) {
        const int id = task_id(k), row = id / TP_Q, q = id - row * TP_Q;
        dst[k] = (row * RS + (q >> 1)) * 16 + (q & 1) * 8;
        inner[k] = tid + 256 * k < IH * TP_Q && row >= P && row < P + C::TH && q >= 1 && q <= TP_TW / 4;
    }
    struct TileRef {
        int n, x0, y0, ti;                                                // image, corner, tile index inside the image
        const uint4 *w;
    };
    TileRef cur;
    __amdgpu_buffer_rsrc_t rx;
    auto locate = [&](int t, TileRef &r) {
        const int tx = t % tiles_x, q = t / tiles_x, ty = q % tiles_y;
        r.n = q / tiles_y;
        r.x0 = tx * TP_TW;
        r.y0 = ty * C::TH;
        r.ti = ty * tiles_x + tx;
        const int g = d.group_n > 0 ? r.n / d.group_n : 0;
        r.w = reinterpret_cast<const uint4 *>(d.wpack + (size_t)g * d.wpack_gs);
    };
    auto setup = [&](const TileRef &r) {
        const int g = d.group_n > 0 ? r.n / d.group_n : 0;
        const int nx = (d.group_flags & RISP_GROUP_SHARED_X) ? r.n - g * d.group_n : r.n;
        rx = h2_rsrc(d.x + (size_t)nx * d.cin * hw);
#pragma unroll
        for (int k = 0; k < NTASK; ++k) {
            const int id = task_id(k), row = id / TP_Q, q = id - row * TP_Q;
            const int gy = r.y0 - P + row, gx = r.x0 - 4 + 4 * q;
            ok[k] = gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
            off[k] = ok[k] ? 4u * (unsigned)(gy * d.W + gx) : 0u;
        }
    };
    float4 v[NTASK];
    auto fetch = [&](int ci) {
        const unsigned so = (unsigned)ci * hw4;
#pragma unroll
        for (int k = 0; k < NTASK; ++k) v[k] = h2_load16(rx, off[k], so);
    };
    // band pieces of 64 slots; every wave issues PW transfers (a wave without a piece of its own repeats an earlier one)
    unsigned wvoff[PW], wlds[PW];                                     // lane offset in the pack; LDS byte address in buffer 0 (scalar)
    const unsigned lds_wl = lds_addr_of(wl);
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        const int piece = (wave + 4 * p) % (WST / 64);
        wvoff[p] = 16u * (unsigned)(piece * 64 + lane);
        wlds[p] = lds_wl + 16u * (unsigned)(piece * 64);
    }
    auto issue_bands = [&](int ci, int slot, const TileRef &r) {
        const uint4 *src = r.w + 1 + (size_t)ci * WST;                 // slot 0 of the pack = header
#pragma unroll
        for (int p = 0; p < PW; ++p) lds_dma16_m(src, wvoff[p], wlds[p] + (unsigned)slot * (WST * 16u));
    };
    // operands.  B: lane (b = lane & 31, half) of input row r reads slot b + half of that row (pixels 8 b - 4 + 8 half ..);
    // A: lane (m = lane & 31, half) reads row m, window half `half` of a band.
    const int srow = FOLD * T * wave + (l31 / NBLK) * T, sblk = l31 % NBLK;    // the lane's first row inside the tile, its pixel block
    const int bbase = srow * RS + sblk + hl;
    const int abase = hl * NB * 32 + l31;

#ifdef RISP_TP_STAMPS
    unsigned long long t_start = __builtin_amdgcn_s_memtime(), t_top = 0, t_stage = 0, t_feed = 0, t_mat = 0, t_epi = 0, t0, t1;
#define TPSTAMP(acc_) do { __builtin_amdgcn_s_waitcnt(0xC07F); t1 = __builtin_amdgcn_s_memtime(); acc_ += t1 - t0; t0 = t1; } while (0)
#else
#define TPSTAMP(acc_) do { } while (0)
#endif
    int t_cur = wg;
    if (t_cur >= ntiles) return;
    locate(t_cur, cur);
    setup(cur);
    issue_bands(0, 0, cur);
    fetch(0);
    int ring = 0;
    for (;;) {
        f32x16 acc[T][NB];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][b][e] = 0.f;
        int se = 0;                                    // running exponent: the accumulators hold sum * 2^se * s_w
        const int t_next = t_cur + nwg;
        const bool more = t_next < ntiles;
        TileRef nxt = cur;
        if (more) locate(t_next, nxt);
        for (int ci = 0; ci < d.cin; ++ci) {
#ifdef RISP_TP_STAMPS
            t0 = __builtin_amdgcn_s_memtime();
#endif
            float m = 0.f;
#pragma unroll
            for (int k = 0; k < NTASK; ++k)
                if (ok[k]) m = amax4(m, v[k]);
            m = h2_wave_max(m);
            if (lane == 0) red[wave] = m;
            if (psum) {                                // the channel's sum over the tile's own pixels, in a fixed order (risp_rect_sums_tiles)
                float sm = 0.f;
#pragma unroll
                for (int k = 0; k < NTASK; ++k)
                    if (inner[k] && ok[k]) sm += (v[k].x + v[k].y) + (v[k].z + v[k].w);
                sm = h2_wave_sum(sm);
                if (lane == 0) red[4 + wave] = sm;
            }
            __syncthreads();                           // A: the maxima are visible; every wave has left the previous channel's tile and bands
            TPSTAMP(t_top);
            if (psum && tid == 0) psum[((size_t)cur.n * (tiles_x * tiles_y) + cur.ti) * d.cin + ci] = (red[4] + red[5]) + (red[6] + red[7]);
            const float4 mx = *reinterpret_cast<const float4 *>(red);
            const float tmax = fmaxf(fmaxf(mx.x, mx.y), fmaxf(mx.z, mx.w));
            int eb = (int)(__builtin_bit_cast(unsigned, tmax) >> 23);
            eb = __builtin_amdgcn_readfirstlane(eb);
            int want = 141 - eb;                        // exponent of s_x: tmax s_x in [2^14, 2^15)
            want = want > 100 ? 100 : want;             // an all-zero or denormal tile: any scale will do
            if (ci == 0) {
                se = want;
            } else if (want < se) {                     // larger values than before: rescale the running sums (exact)
                const int fe = 127 + want - se;
                const float f = fe > 0 ? __builtin_bit_cast(float, (unsigned)fe << 23) : 0.f;
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int b = 0; b < NB; ++b)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[t][b][e] *= f;
                se = want;
            }
            const float s = __builtin_bit_cast(float, (unsigned)(127 + se) << 23);
#pragma unroll
            for (int k = 0; k < NTASK; ++k) {
                const float a0 = ok[k] ? v[k].x * s : 0.f, a1 = ok[k] ? v[k].y * s : 0.f, a2 = ok[k] ? v[k].z * s : 0.f, a3 = ok[k] ? v[k].w * s : 0.f;
                const h2 h01 = {(_Float16)a0, (_Float16)a1}, h23 = {(_Float16)a2, (_Float16)a3};
                const h2 l01 = {(_Float16)(a0 - (float)h01[0]), (_Float16)(a1 - (float)h01[1])};
                const h2 l23 = {(_Float16)(a2 - (float)h23[0]), (_Float16)(a3 - (float)h23[1])};
                char *base = reinterpret_cast<char *>(tile) + dst[k];
                *reinterpret_cast<uint2 *>(base) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
                *reinterpret_cast<uint2 *>(base + C::PART * 16) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
            }
            H2_WAIT_VM(0);                             // this wave's band pieces of the channel (issued a channel ago) have landed
            __syncthreads();                           // B: tile and bands complete
            TPSTAMP(t_stage);
            // the next channel - or the next tile's first - is requested now and is in flight during the matrix phase: bands into
            // the buffer every wave has left, the tile into registers
            if (ci + 1 < d.cin) {
                issue_bands(ci + 1, ring ^ 1, cur);
#ifdef RISP_TP_STAMPS
                TPSTAMP(t_epi);                        // (diagnostic: the band transfers alone are booked on the epilogue's counter)
#endif
                fetch(ci + 1);
            } else if (more) {
                issue_bands(0, ring ^ 1, nxt);
                setup(nxt);
                fetch(0);
            }
            TPSTAMP(t_feed);
            // ---- matrix phase
            const uint4 *ws = wl + ring * WST + abase;
            const uint4 *ts = tile + bbase;
            if constexpr (NB == 1) {
                // the band operands of all KS filter rows stay in registers; the wave walks the T + KS - 1 input rows it needs: each
                // row's operand is read once and serves every (output row t, filter row ky) pair with t + ky = row
                h8 a[KS][2], bv[3][2];
                auto load_a = [&](int ky) {
#pragma unroll
                    for (int part = 0; part < 2; ++part) a[ky][part] = __builtin_bit_cast(h8, ws[((ky * 2 + part) * 2) * 32]);
                };
                auto load_b = [&](int r, int buf) {
                    bv[buf][0] = __builtin_bit_cast(h8, ts[r * RS]);
                    bv[buf][1] = __builtin_bit_cast(h8, ts[C::PART + r * RS]);
                };
                load_a(0);
                load_b(0, 0);
                load_b(1, 1);
#pragma unroll
                for (int r = 0; r < T + KS - 1; ++r) {
                    if (r + 2 < T + KS - 1) load_b(r + 2, (r + 2) % 3);
                    if (r + 1 < KS) load_a(r + 1);
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        const int ky = r - t;
                        if (ky >= 0 && ky < KS) {
                            acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ky][0], bv[r % 3][1], acc[t][0], 0, 0, 0);
                            acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ky][1], bv[r % 3][0], acc[t][0], 0, 0, 0);
                            acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ky][0], bv[r % 3][0], acc[t][0], 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // several row blocks (5x5 32 -> 12): the bands of ONE filter row at a time (2 NB operands, double-buffered), steps
                // (ky, t) with the pixel operand of the next step read under the 3 NB products of this one
                h8 a[2][NB][2], bv[2][2];
                auto load_a = [&](int ky, int buf) {
#pragma unroll
                    for (int b = 0; b < NB; ++b)
#pragma unroll
                        for (int part = 0; part < 2; ++part) a[buf][b][part] = __builtin_bit_cast(h8, ws[((ky * 2 + part) * 2) * NB * 32 + b * 32]);
                };
                auto load_b = [&](int r, int buf) {
                    bv[buf][0] = __builtin_bit_cast(h8, ts[r * RS]);
                    bv[buf][1] = __builtin_bit_cast(h8, ts[C::PART + r * RS]);
                };
                load_a(0, 0);
                load_b(0, 0);
#pragma unroll
                for (int step = 0; step < KS * T; ++step) {
                    const int ky = step / T, t = step - ky * T;
                    if (step + 1 < KS * T) {
                        load_b((step + 1) / T + (step + 1) % T, (step + 1) & 1);
                        if (t == T - 1) load_a(ky + 1, (ky + 1) & 1);
                    }
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ky & 1][b][0], bv[step & 1][1], acc[t][b], 0, 0, 0);
                        acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ky & 1][b][1], bv[step & 1][0], acc[t][b], 0, 0, 0);
                        acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ky & 1][b][0], bv[step & 1][0], acc[t][b], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            ring ^= 1;
            TPSTAMP(t_mat);
        }
#ifdef RISP_TP_STAMPS
        t0 = __builtin_amdgcn_s_memtime();
#endif
        // ---- epilogue: y = epilogue(acc * 2^-se / s_w + bias).  Lane (b, half), accumulator element e of block nb: cout 4 nb + (e >> 2),
        // pixel 8 b + 4 half + (e & 3) - four consecutive pixels per cout: 16-byte stores, a wave row = 1 KB contiguous.
        {
            const int g = d.group_n > 0 ? cur.n / d.group_n : 0;
            const int na = (d.group_flags & RISP_GROUP_SHARED_ADD) ? cur.n - g * d.group_n : cur.n;
            const float inv_sw = *reinterpret_cast<const float *>(cur.w);
            const float fin = inv_sw * __builtin_bit_cast(float, (unsigned)(127 - se) << 23);
            const int epi = d.epilogue;
            const float floor_ = (epi & RISP_EPI_RELU) ? 0.f : -__builtin_inff();
            const int ox = cur.x0 + 8 * sblk + 4 * hl;
            const float *bias = (epi & RISP_EPI_NOBIAS) ? nullptr : d.bias + (size_t)g * d.bias_gs;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int oy = cur.y0 + srow + t;
                if (oy >= d.H || ox >= d.W) continue;
                const size_t pix = (size_t)oy * d.W + ox;
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if constexpr (SHUF) {               // PixelShuffle(2): cout 4 g + 2 i + j -> plane g, pixel (2 y + i, 2 x + j)
                        if (4 * b < d.cout) {
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                const float b0 = bias ? bias[4 * b + 2 * i] : 0.f, b1 = bias ? bias[4 * b + 2 * i + 1] : 0.f;
                                float e8[8];
#pragma unroll
                                for (int p = 0; p < 4; ++p) {
                                    e8[2 * p] = acc[t][b][4 * (2 * i) + p] * fin + b0;
                                    e8[2 * p + 1] = acc[t][b][4 * (2 * i + 1) + p] * fin + b1;
                                }
                                float *yp = d.y + (((size_t)cur.n * (d.cout >> 2) + b) * 2 * d.H + 2 * oy + i) * (2 * (size_t)d.W) + 2 * ox;
                                *reinterpret_cast<float4 *>(yp) = make_float4(e8[0], e8[1], e8[2], e8[3]);
                                *reinterpret_cast<float4 *>(yp + 4) = make_float4(e8[4], e8[5], e8[6], e8[7]);
                            }
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int co = 4 * b + c;
                            if (co < d.cout) {
                                const float bb = bias ? bias[co] : 0.f;
                                float4 o = make_float4(acc[t][b][4 * c] * fin + bb, acc[t][b][4 * c + 1] * fin + bb, acc[t][b][4 * c + 2] * fin + bb,
                                                       acc[t][b][4 * c + 3] * fin + bb);
                                if (HAS_ADD && co < d.add_c) {
                                    const float4 a4 = *reinterpret_cast<const float4 *>(d.add + ((size_t)na * d.add_c + co) * hw + pix);
                                    o.x += a4.x; o.y += a4.y; o.z += a4.z; o.w += a4.w;
                                }
                                o.x = o.x < floor_ ? floor_ : o.x;            // ReLU, or nothing (floor = -inf); a NaN stays a NaN
                                o.y = o.y < floor_ ? floor_ : o.y;
                                o.z = o.z < floor_ ? floor_ : o.z;
                                o.w = o.w < floor_ ? floor_ : o.w;
                                *reinterpret_cast<float4 *>(d.y + ((size_t)cur.n * d.cout + co) * hw + pix) = o;
                            }
                        }
                    }
                }
            }
        }
        TPSTAMP(t_epi);
        if (!more) break;
        cur = nxt;
        t_cur = t_next;
    }
#ifdef RISP_TP_STAMPS
    if (lane == 0 && d.cvals) {                        // diagnostic build: cycle shares of a wave's life (tools/ab_toep.py)
        unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<float *>(d.cvals)) + 6 * ((size_t)blockIdx.x * 4 + wave);
        o[0] = t_top; o[1] = t_stage; o[2] = t_feed; o[3] = t_mat; o[4] = t_epi; o[5] = __builtin_amdgcn_s_memtime() - t_start;
    }
#endif
}

#ifndef RISP_TP_WGS
#define RISP_TP_WGS 2        // persistent workgroups per CU
#endif

template <int KS, int NB, int T, int FOLD, bool HAS_ADD, bool SHUF>
int launch_toep_g(const risp_conv_desc &d, float *psum, void *stream) {
    using C = TP<KS, NB, T, FOLD>;
    auto kern = &conv_toep_kernel<KS, NB, T, FOLD, HAS_ADD, SHUF>;
    if (C::LDS_BYTES > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess) {
        risp_set_error("risp_conv2d_toep: cannot raise the dynamic LDS limit to %d bytes", C::LDS_BYTES);
        return 2;
    }
    const int tx = (d.W + C::TW - 1) / C::TW, ty = (d.H + C::TH - 1) / C::TH;
    const long long ntiles = (long long)tx * ty * d.N;
    if (ntiles > 0x7fffffff) {
        risp_set_error("risp_conv2d_toep: too many tiles");
        return 1;
    }
    const int slots = RISP_TP_WGS * h2_cu_count();
    const int grid = ntiles < slots ? (int)ntiles : slots;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), C::LDS_BYTES, (hipStream_t)stream, d, tx, ty, (int)ntiles, psum);
    RISP_LAUNCH_CHECK("risp_conv2d_toep");
    return 0;
}

template <int KS, int NB, int T, bool HAS_ADD, bool SHUF>
int launch_toep(const risp_conv_desc &d, float *psum, void *stream) {
    // (three row blocks: the bands of a channel take 30 KB twice - no room for the folded tile: narrow planes run half empty there)
    if constexpr (NB == 1) {
        if (d.W <= 128) return launch_toep_g<KS, NB, T, 2, HAS_ADD, SHUF>(d, psum, stream);
    }
    return launch_toep_g<KS, NB, T, 1, HAS_ADD, SHUF>(d, psum, stream);
}

template <int KS, int NB, int T>
int launch_toep_epi(const risp_conv_desc &d, float *psum, void *stream) {
    if (d.epilogue & RISP_EPI_SHUFFLE2) return launch_toep<KS, NB, T, false, true>(d, psum, stream);
    return (d.epilogue & RISP_EPI_ADD) ? launch_toep<KS, NB, T, true, false>(d, psum, stream) : launch_toep<KS, NB, T, false, false>(d, psum, stream);
}
}  // namespace

extern "C" {

size_t risp_conv_toep_wpack_bytes(int cin, int cout, int ksize) {
    const int nb = cout <= 4 ? 1 : 3;
    return 16 + (size_t)cin * ksize * 2 * 2 * nb * 32 * 16;
}

static int conv2d_toep_impl(const risp_conv_desc *dp, float *psum, void *stream) {
    RISP_CHECK_ARG(dp, "risp_conv2d_toep: null descriptor");
    const risp_conv_desc &d = *dp;
    RISP_CHECK_ARG(d.x && d.wpack && d.y, "risp_conv2d_toep: null tensor");
    RISP_CHECK_GROUP(d, "risp_conv2d_toep");
    RISP_CHECK_ARG(d.N > 0 && d.H > 0 && d.W > 0 && d.W % 4 == 0 && d.cin > 0 && d.cout > 0 && (d.cout <= 4 || (d.cout <= 12 && d.ksize == 5)) && (d.ksize == 5 || d.ksize == 9) &&
                       (unsigned long long)d.cin * d.H * d.W < (1ull << 30),
                   "risp_conv2d_toep: needs a 9x9 layer with cout <= 4 or a 5x5 layer with cout <= 12, W %% 4 == 0, fewer than 2^30 input elements per image "
                   "(N=%d H=%d W=%d cin=%d cout=%d k=%d)",
                   d.N, d.H, d.W, d.cin, d.cout, d.ksize);
    RISP_CHECK_ARG(d.load_mode == RISP_LOAD_PLAIN, "risp_conv2d_toep: only plain loads");
    RISP_CHECK_ARG(!(d.epilogue & ~(RISP_EPI_RELU | RISP_EPI_ADD | RISP_EPI_NOBIAS | RISP_EPI_SHUFFLE2)), "risp_conv2d_toep: epilogue %d not supported",
                   d.epilogue);
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_SHUFFLE2) || (d.cout % 4 == 0 && !(d.epilogue & (RISP_EPI_RELU | RISP_EPI_ADD))),
                   "risp_conv2d_toep: PixelShuffle store needs cout %% 4 == 0 and no other epilogue");
    RISP_CHECK_ARG((d.epilogue & RISP_EPI_NOBIAS) || d.bias, "risp_conv2d_toep: bias missing");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_ADD) || (d.add && d.add_c > 0), "risp_conv2d_toep: add tensor missing");
    RISP_CHECK_ARG(((reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.y) | reinterpret_cast<uintptr_t>(d.add) |
                     reinterpret_cast<uintptr_t>(d.wpack)) & 15) == 0,
                   "risp_conv2d_toep: tensors must be 16-byte aligned");
    if (d.ksize == 9) return launch_toep_epi<9, 1, 4>(d, psum, stream);
    // three row blocks: 2 output rows per wave (8-row tiles).  (3 rows: 1.4 x the tile time for 1.5 x the rows, but the tile shape must
    // not depend on the launch - the per-tile scale would make an inference result depend on the batch it travels in)
    if (d.cout > 4) return launch_toep_epi<5, 3, 2>(d, psum, stream);
    return launch_toep_epi<5, 1, 4>(d, psum, stream);      // (6 rows per wave: no faster, 8 spill)
}

int risp_conv2d_toep(const risp_conv_desc *dp, void *stream) { return conv2d_toep_impl(dp, nullptr, stream); }

/* ... and, on the way, the sum of every input channel over every tile's own pixels: psum [N][tiles per image][cin] floats, tile
 * t = (y / 16) * ceil(W / 256) + x / 256 (risp_conv_toep_tiles per image).  What risp_rect_sums_tiles finishes into the
 * rectangle sums of the constant-plane gradient (srcnn_res_arch.py:41-46) without reading the 64-channel tensor again. */
int risp_conv_toep_tiles(int H, int W) {           /* cout <= 4: tiles of 16 x 256, or 32 x 128 on planes of at most 128 pixels */
    return W <= 128 ? ((H + 31) / 32) * ((W + 127) / 128) : ((H + 15) / 16) * ((W + 255) / 256);
}

int risp_conv2d_toep_sums(const risp_conv_desc *dp, float *psum, void *stream) {
    RISP_CHECK_ARG(psum && dp && dp->cout <= 4, "risp_conv2d_toep_sums: needs the buffer of partial sums and a layer with cout <= 4 (16-row tiles)");
    return conv2d_toep_impl(dp, psum, stream);
}

}  // extern "C"
