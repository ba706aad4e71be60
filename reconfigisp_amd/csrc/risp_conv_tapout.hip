// Split-precision convolution for layers with at most 3 OUTPUT channels and many input channels, with the filter ROWS moved into the
// rows of the matrix instruction (round 6; replaces the Toeplitz-band kernel of risp_conv_toep.hip for these layers):
//   * SRCNNRes conv 9x9 (12+P) -> 64 (srcnn_res_arch.py:18), backward-data restricted to the 3 image channels   (64 -> 3)
//   * SRCNNRes conv 5x5 32 -> 3 (srcnn_res_arch.py:22), forward
// With 3 couts the 32 rows of v_mfma_f32_32x32x16_f16 cannot be couts; the band kernel made them (cout, pixel position) and paid
// for it with a reduction index of which 9 of 16 slots carry a tap (2.4 x the useful products for 9x9 64 -> 3).  Here the reduction
// index is what it is in a wide layer - 16 INPUT CHANNELS, dense - and the rows are (cout, filter row ky):
//        D[(c, ky)][y'][x] = sum_kx sum_ci  w[c][ci][ky][kx]  in[ci][y'][x + kx - P]          one input row y', 27 of 32 rows (9x9)
//        out[c][y][x]      = sum_ky D[(c, ky)][y + ky - P][x]                                  a vertical shift-add
// The filter COLUMN kx is a shift of the pixel operand - an LDS address offset, free - so one input row costs K x cin / 16 steps of
// 3 products for every 32 pixels (9x9, 64 channels: 108 matrix instructions per 32 pixels, the band form 216), and every input row
// is staged exactly once: a persistent workgroup owns a strip of 128 columns and WALKS DOWN the image, 4 input rows per step.  A
// consumer wave owns a block of 32 columns and all 4 rows of the step, so a column of the output is only ever touched by one wave -
// by two lanes of it, which hold the even and the odd filter rows: the vertical shift-add runs in REGISTERS, a ring of 13 partial
// output rows per cout and lane that moves up by 4 rows per step; the 4 rows whose last input row the step held leave it - the two
// lanes' halves added through one lane exchange - through the epilogue.  No LDS traffic, no atomics, one fixed order of additions.
// (The first build kept the ring in LDS and added into it with ds_add_f32: correct, and 35 % of the kernel's time - LDS float adds
// retire at about one lane per clock and CU.)  Arithmetic as in risp_conv_f16x2.hip: two f16 halves per fp32 operand, three
// products, fp32 accumulation, the activations scaled per (step tile, chunk of 16 channels) by the tile's own largest magnitude with
// a running exponent in the accumulators; each input row's D is converted to true scale before it enters the ring.
//
// Kernel.  ONE workgroup of 8 waves per CU, wave-specialised like risp_conv_f16x2_ws.hip.  A work item = (image, strip of 128
// columns, segment of rows); a step = 4 input rows; a chunk = 16 input channels of a step: its tile (4 rows x (128 + 2 P) pixels, one
// 16-byte LDS slot = the hi (or lo) halves of 8 channels of a pixel) and its weights (K x 2 KB).
//   waves 4-7  PRODUCERS (one input row each): phase p, between barriers p - 1 and p, scales, splits and writes chunk p into tile[p & 1]
//              (thread = (4 pixels, channel half): 8 16-byte loads, 8 16-byte LDS writes; the 2 P halo columns by 128 threads), sends
//              its weights by LDS-DMA into wl[p & 1], takes the largest magnitude of chunk p + 1 (already in registers) and
//              publishes it in red[(p + 1) & 3], requests chunk p + 2.  Chunks run on across steps and work items.
//   waves 0-3  CONSUMERS (32 columns each): after barrier q the matrix instructions of chunk q out of tile[q & 1] / wl[q & 1] -
//              per filter column 12 products, the operands of the next column read meanwhile -, after a step's last chunk the
//              ring, the retiring rows and their stores.
// One barrier per chunk, 12 K matrix instructions per consumer wave between two of them.
#include "risp_f16x2.h"
#include <type_traits>

namespace {
constexpr int TO_TW = 128, TO_RS = 4;

template <int KS>
struct TOG {
    static constexpr int P = KS / 2, TWH = TO_TW + 2 * P;
    static constexpr int NR = TO_RS + 2 * P + 1;                  // ring rows per cout and lane (the odd-row lanes sit one row lower)
    static constexpr int PART = 2 * TO_RS * TWH;                  // slots of one part (hi or lo): [channel half][row][column]
    static constexpr int TILE = 2 * PART;
    static constexpr int WST = KS * 2 * 2 * 32;                   // weight slots of a chunk: [kx][part][channel half][row m]
    static constexpr int NPIECE = WST / 64, PW = (NPIECE + 3) / 4;
    static constexpr int NLOAD = 8 + (P == 4 ? 1 : P);            // vector-memory loads of a chunk per producer thread
    static constexpr int LDS_BYTES = (2 * TILE + 2 * WST) * 16 + 64 + 4 * 64 * 4;      // ... + the maxima (4 chunks x 4 waves) + the channel sums of 4 waves
    static_assert(WST % 64 == 0, "weights in whole LDS-DMA pieces");
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
    static_assert((2 * P) % TO_RS == 0, "output rows complete in whole groups of a step");
};

#define TO_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

struct ToItem {
    int n, x0, ys, ye, g;                                           // image, strip corner, output rows [ys, ye), group member
};

// exponent a chunk asks for: its tile's largest magnitude (per producer wave in red4[0..3]) into [2^14, 2^15)
__device__ __forceinline__ int to_want(const float *red4) {
    const float4 mx = *reinterpret_cast<const float4 *>(red4);
    const float tmax = fmaxf(fmaxf(mx.x, mx.y), fmaxf(mx.z, mx.w));
    int eb = (int)(__builtin_bit_cast(unsigned, tmax) >> 23);
    eb = __builtin_amdgcn_readfirstlane(eb);
    const int want = 141 - eb;
    return want > 100 ? 100 : want;                                 // an all-zero or denormal tile: any scale will do
}

// c ? a : b of two REGISTER values (hipcc turns `c ? x[i + 1] : x[i]` into x[i + c] and moves the whole array to scratch memory)
__device__ __forceinline__ float to_sel(bool c, float a, float b) {
    asm volatile("" : "+v"(a));
    asm volatile("" : "+v"(b));
    return c ? a : b;
}

#ifdef RISP_TO_STAMPS
#define TO_T() __builtin_amdgcn_s_memtime()
#else
#define TO_T() 0ull
#endif

// PSUM: also the sum of every input channel over the item's own pixels (cin == 64), for risp_rect_sums_tiles
template <int KS, bool HAS_ADD, bool PSUM>
__global__ __launch_bounds__(512, 2) void conv_tapout_kernel(const risp_conv_desc d, int strips, int segs, int seg_rows, int nitems,
                                                             float *__restrict__ psum) {
    using G = TOG<KS>;
    constexpr int P = G::P, TWH = G::TWH, NR = G::NR, PART = G::PART, TILE = G::TILE, WST = G::WST, PW = G::PW, TW = TO_TW, RS = TO_RS, LAG = 2 * P / RS;
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    uint4 *tile = smem, *wl = smem + 2 * TILE;
    float *red = reinterpret_cast<float *>(wl + 2 * WST);               // [chunk & 3][producer wave]
    float *psred = red + 16;                                             // [producer wave][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hl = lane >> 5;
    const size_t hw = (size_t)d.H * d.W;
    const unsigned hw4 = (unsigned)hw * 4u;
    const int nch = d.cin >> 4;
    const int nwg = gridDim.x;
    const int wg = (nwg & 7) == 0 ? (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    if (wg >= nitems) return;
    const int per = strips * segs;
    auto locate = [&](int t, ToItem &r) {
        r.n = t / per;
        const int q = t - r.n * per, sg = q / strips, st = q - sg * strips;
        r.x0 = st * TW;
        r.ys = sg * seg_rows;
        r.ye = r.ys + seg_rows < d.H ? r.ys + seg_rows : d.H;
        r.g = d.group_n > 0 ? r.n / d.group_n : 0;
    };
    // steps of an item: s = 0 .. ns - 1 cover input rows ys - P + RS s ..; those that meet the image are computed (s_lo .. s_hi), every
    // step retires the group of 4 output rows whose last input row it held (LAG steps behind)
    auto first_step = [&](const ToItem &r) { return r.ys - P < 0 ? (P - r.ys) / RS : 0; };
    auto last_step = [&](const ToItem &r) { return ((r.ye + P < d.H ? r.ye + P : d.H) - 1 - (r.ys - P)) / RS; };
#ifndef TO_PSPLIT
#define TO_PSPLIT 0
#endif
#ifdef RISP_TO_STAMPS
    unsigned long long t_wait = 0, t_work = 0, t_tail = 0, t0 = TO_T(), t1;
    const unsigned long long t_start = t0;
#define TOSTAMP(acc_) do { t1 = TO_T(); acc_ += t1 - t0; t0 = t1; } while (0)
#else
#define TOSTAMP(acc_) do { } while (0)
#endif

    if (wave >= 4) {
        // =============================================================================================== producers
        // staging tasks.  Main: row = producer wave, pixels 4 (lane & 31) .., channels 8 (lane >> 5) .. of the chunk.  Halo (the first 128
        // producer threads): row = (tid >> 5) & 3, side = (tid >> 4) & 1, channel tid & 15; P pixels left of the strip or right of it.
        const int pt = tid - 256, pw = wave - 4;
        const int hrow = (pt >> 5) & 3, hside = (pt >> 4) & 1, hch = pt & 15;
        const bool htask = pt < 128;
        const int mdst = ((hl * RS + pw) * TWH + P + 4 * l31) * 16;                                  // byte offset in the hi part
        const int hdst = ((((hch >> 3) * RS + hrow) * TWH) + (hside ? TW + P : 0)) * 16 + 2 * (hch & 7);
        unsigned wvoff[PW], wlds[PW];
        const unsigned lds_wl = lds_addr_of(wl);
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            const int piece = (pw + 4 * p) % G::NPIECE;
            wvoff[p] = 16u * (unsigned)(piece * 64 + lane);
            wlds[p] = lds_wl + 16u * (unsigned)(piece * 64);
        }
        // the chunk sequence of this workgroup: items wg, wg + nwg, ..; steps s_lo .. s_hi; chunks 0 .. nch - 1
        int cu_t = wg, cu_s = 0, cu_shi = 0, cu_c = 0;
        bool cu_valid = false;
        ToItem cu_it;
        __amdgpu_buffer_rsrc_t rx;
        unsigned mxoff = 0, hxoff = 0;
        bool mxok = false, hxok = false;
        auto enter_item = [&](int t) {
            cu_t = t;
            cu_valid = t < nitems;
            if (!cu_valid) return;
            locate(t, cu_it);
            cu_s = first_step(cu_it);
            cu_shi = last_step(cu_it);
            cu_c = 0;
            const int nx = (d.group_flags & RISP_GROUP_SHARED_X) ? cu_it.n - cu_it.g * d.group_n : cu_it.n;
            rx = h2_rsrc(d.x + (size_t)nx * d.cin * hw);
            const int gx = cu_it.x0 + 4 * l31;
            mxok = gx < d.W;
            mxoff = 4u * (unsigned)gx + (unsigned)(8 * hl) * hw4;
            const int hx = hside ? cu_it.x0 + TW : cu_it.x0 - P;
            hxok = htask && hx >= 0 && hx < d.W;
            hxoff = 4u * (unsigned)(hx < 0 ? 0 : hx) + (unsigned)hch * hw4;
        };
        auto advance = [&]() {
            if (++cu_c < nch) return;
            cu_c = 0;
            if (++cu_s <= cu_shi) return;
            enter_item(cu_t + nwg);
        };
        // a chunk in registers: a lane outside the image carries an out-of-range offset (the buffer returns zeros)
        struct Set {
            float4 v[8];
            float hv[P];
            const uint4 *w;                                         // its weights (group member's pack, slot 0 = header)
            int c, t;
            bool valid, own, last;                                  // own: the wave's row belongs to the item's segment; last: the item's last chunk
        };
        auto fetch = [&](Set &z) {
            z.valid = cu_valid;
            if (!cu_valid) return;
            z.c = cu_c;
            z.t = cu_t;
            z.w = reinterpret_cast<const uint4 *>(d.wpack + (size_t)cu_it.g * d.wpack_gs);
            z.last = cu_c == nch - 1 && cu_s == cu_shi;
            const int y = cu_it.ys - P + RS * cu_s + pw;
            z.own = y >= cu_it.ys && y < cu_it.ye;
            const unsigned mo = (mxok && y >= 0 && y < d.H) ? mxoff + 4u * (unsigned)(y * d.W) : 0x80000000u;
            const unsigned so = (unsigned)(16 * cu_c) * hw4;
#pragma unroll
            for (int k = 0; k < 8; ++k) z.v[k] = h2_load16(rx, mo, so + (unsigned)k * hw4);
            const int yh = cu_it.ys - P + RS * cu_s + hrow;
            const unsigned ho = (hxok && yh >= 0 && yh < d.H) ? hxoff + 4u * (unsigned)(yh * d.W) : 0x80000000u;
            if constexpr (P == 4) {
                const float4 q = h2_load16(rx, ho, so);
                z.hv[0] = q.x; z.hv[1] = q.y; z.hv[2] = q.z; z.hv[3] = q.w;
            } else {
#pragma unroll
                for (int k = 0; k < P; ++k) z.hv[k] = h2_load4(rx, ho + 4u * k, so);
            }
            advance();
        };
        float psacc[PSUM ? 4 : 1][8];
        if (PSUM) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int k = 0; k < 8; ++k) psacc[c][k] = 0.f;
        }
        int ps_pending = -1;                                            // item whose channel sums sit in psred, to be finished after the next barrier
        // largest magnitude of a chunk -> red[slot][pw]; its own pixels into the channel sums
        // (`ctag`: the chunk's index among the item step's 4 as a compile-time constant - the four phases of a step are four instances of the
        // code; indexed with z.c the sums' registers were addressed through s_set_gpr_idx_on / _off, 64 mode switches per chunk.  The sums
        // themselves: 32 additions per chunk - as v_add_f32.  Paired by the SLP vectoriser into 16 v_pk_add_f32 they took 1700 cycles
        // beside the consumers' matrix instructions (350 unpaired): the library is built with -fno-slp-vectorize, see the Makefile.)
        auto amax = [&](Set &z, int slot, auto ctag) {
            constexpr int C = decltype(ctag)::value;
            if (!z.valid) return;
            float m = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) m = amax4(m, z.v[k]);
#pragma unroll
            for (int k = 0; k < P; ++k) m = fmaxf(m, fabsf(z.hv[k]));
            m = h2_wave_max(m);
            if (lane == 0) red[slot * 4 + pw] = m;
            if (PSUM) {
                if (z.own) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) psacc[PSUM ? C : 0][k] += (z.v[k].x + z.v[k].y) + (z.v[k].z + z.v[k].w);
                }
                if (z.last) {
                    // the item's channel sums: lanes of one half hold the same 8 channels of each chunk; rows of 16 lanes by DPP, the two rows of a
                    // half by readlane, the four producer waves through LDS after the next barrier - a fixed order
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            float x = psacc[c][k];
                            x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
                            x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
                            x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
                            x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));
                            const int xi = __builtin_bit_cast(int, x);
                            const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 16));
                            const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 48));
                            if (lane == 0) {
                                psred[pw * 64 + c * 16 + k] = s0;
                                psred[pw * 64 + c * 16 + 8 + k] = s1;
                            }
                            psacc[c][k] = 0.f;
                        }
                    ps_pending = z.t;
                }
            }
        };
        int se = 0;                                                     // the step's running exponent, as the consumers track it
        // chunk z (its maxima in red[slot]) -> LDS buffer `buf`, its weights by LDS-DMA
        auto stage = [&](Set &z, int slot, int buf) {
            if (!z.valid) return;
            const int want = to_want(red + slot * 4);
            se = (z.c == 0 || want < se) ? want : se;
            const float sc = __builtin_bit_cast(float, (unsigned)(127 + se) << 23);
            {
                const uint4 *src = z.w + 1 + (size_t)z.c * WST;
#pragma unroll
                for (int p = 0; p < PW; ++p) lds_dma16_m(src, wvoff[p], wlds[p] + (unsigned)buf * (WST * 16u));
            }
            char *tb = reinterpret_cast<char *>(tile + buf * TILE);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a8[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) a8[k] = comp(z.v[k], i);
                uint4 hi, lo;
                split8(a8, sc, hi, lo);
                *reinterpret_cast<uint4 *>(tb + mdst + 16 * i) = hi;
                *reinterpret_cast<uint4 *>(tb + mdst + 16 * i + PART * 16) = lo;
            }
            if (htask) {
#pragma unroll
                for (int k = 0; k < P; ++k) {
                    const float a = z.hv[k] * sc;
                    const _Float16 hh = (_Float16)a, ll = (_Float16)(a - (float)hh);
                    *reinterpret_cast<_Float16 *>(tb + hdst + 16 * k) = hh;
                    *reinterpret_cast<_Float16 *>(tb + hdst + 16 * k + PART * 16) = ll;
                }
            }
            // every lane has requested the halo values, the halo lanes alone read them: for the others hipcc keeps the request "in flight"
            // and, where it next reuses those registers (the addresses of the following request), waits for EVERY vector-memory operation -
            // the weight pieces above included.  A use by all lanes, here where the next wait asks for more anyway:
#pragma unroll
            for (int k = 0; k < P; ++k) asm volatile("" :: "v"(z.hv[k]));
        };
        auto finish_psum = [&]() {
            if (PSUM && ps_pending >= 0) {
                if (pt < 64) {
                    const int n = ps_pending / per;
                    psum[((size_t)n * per + (ps_pending - n * per)) * 64 + pt] = (psred[pt] + psred[64 + pt]) + (psred[128 + pt] + psred[192 + pt]);
                }
                // the store retires HERE, once per item, with a wait the compiler can see (where this branch joins the phase it would
                // otherwise assume the store in flight when it next reuses the store's registers)
                __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
                ps_pending = -1;
            }
        };
        Set za, zb;
        enter_item(wg);
        fetch(za);                                                      // chunk 0
        fetch(zb);                                                      // chunk 1
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(G::NLOAD) : "memory");      // chunk 0 has landed (chunk 1's loads still in flight)
        amax(za, 0, std::integral_constant<int, 0>{});
        TO_BARRIER_LDS();                                               // barrier "-1": the maxima of chunk 0 are visible
        // phase p: stage chunk p (set p & 1), take the maxima of chunk p + 1 (the other set), request chunk p + 2 (into the set just freed)
        auto phase = [&](Set &zs, Set &zn, int p, auto ctag) {
            TOSTAMP(t_wait);
            finish_psum();
            stage(zs, p & 3, p & 1);
#if defined(RISP_TO_STAMPS) && TO_PSPLIT == 1                /* (tools/tapout_stamps.py -DTO_PSPLIT=n: the producers' phase up to a point, in t_tail) */
            TOSTAMP(t_tail);
#endif
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PW) : "memory");      // chunk p + 1 has landed (behind it: this phase's weight pieces)
#if defined(RISP_TO_STAMPS) && TO_PSPLIT == 2
            TOSTAMP(t_tail);
#endif
            amax(zn, (p + 1) & 3, ctag);
#if defined(RISP_TO_STAMPS) && TO_PSPLIT == 3
            TOSTAMP(t_tail);
#endif
            fetch(zs);
#if defined(RISP_TO_STAMPS) && TO_PSPLIT == 4
            TOSTAMP(t_tail);
#endif
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(G::NLOAD) : "memory");      // the weight pieces have landed (behind them: chunk p + 2)
            TOSTAMP(t_work);
            TO_BARRIER_LDS();
        };
        for (int p = 0;; p += 4) {                                      // (with the channel sums: 64 channels = 4 chunks, chunk of phase p = p & 3)
            if (!za.valid) break;
            phase(za, zb, p, std::integral_constant<int, 1>{});
            if (!zb.valid) break;
            phase(zb, za, p + 1, std::integral_constant<int, 2>{});
            if (!za.valid) break;
            phase(za, zb, p + 2, std::integral_constant<int, 3>{});
            if (!zb.valid) break;
            phase(zb, za, p + 3, std::integral_constant<int, 0>{});
        }
        finish_psum();
    } else {
        // =============================================================================================== consumers
        const int ox_l = 32 * wave + l31;                               // this lane's column inside the strip
#ifdef TO_PRIO
        __builtin_amdgcn_s_setprio(TO_PRIO);
#endif
        int q = 0;                                                      // chunk counter of this workgroup: buffers q & 1, maxima q & 3
        TO_BARRIER_LDS();                                               // barrier "-1"
        for (int t = wg; t < nitems; t += nwg) {
            ToItem it;
            locate(t, it);
            const int rows = it.ye - it.ys, ns = (rows + 2 * P + RS - 1) / RS, s_lo = first_step(it), s_hi = last_step(it);
            const uint4 *wp = reinterpret_cast<const uint4 *>(d.wpack + (size_t)it.g * d.wpack_gs);
            const float inv_sw = *reinterpret_cast<const float *>(wp);
            // the ring: ring[c][r] = this lane's share of output row (ys + RS s - 2 P + r - half) of cout c at the lane's column (the lanes of the
            // upper half hold the odd filter rows and sit one output row lower, so that a contribution's index is the same in both halves)
            float ring[3][NR];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < NR; ++r) ring[c][r] = 0.f;
            const int na = (d.group_flags & RISP_GROUP_SHARED_ADD) ? it.n - it.g * d.group_n : it.n;
            const int ox = it.x0 + ox_l;
            // the residual values of the rows a step retires (rows ys + RS (s - LAG) + 2 half + {0, 1}, this lane's column) are requested
            // BEFORE the step's last matrix phase: behind it, the wait for them would also be a wait for the previous step's stores
            float addv[3][2];
            auto request_add = [&](int s) {
                const int gi = s - LAG;
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) {
                        const int oy = it.ys + RS * gi + 2 * hl + r2;
                        const bool ok = HAS_ADD && gi >= 0 && c < d.add_c && oy < it.ye && ox < d.W;
                        addv[c][r2] = ok ? d.add[((size_t)na * d.add_c + c) * hw + (size_t)oy * d.W + ox] : 0.f;
                    }
            };
            for (int s = s_lo; s < ns; ++s) {
                if (s > s_hi) request_add(s);
                if (s <= s_hi) {
                    f32x16 acc[RS];
#pragma unroll
                    for (int a = 0; a < RS; ++a)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
                    int se = 0;
                    for (int c = 0; c < nch; ++c, ++q) {
                        TOSTAMP(t_tail);
                        TO_BARRIER_LDS();                                // chunk q is in tile[q & 1] / wl[q & 1], its maxima in red[q & 3]
                        TOSTAMP(t_wait);
                        const int want = to_want(red + (q & 3) * 4);
                        if (c == 0) {
                            se = want;
                        } else if (want < se) {                          // larger values than before: rescale the running sums (exact)
                            const int fe = 127 + want - se;
                            const float f = fe > 0 ? __builtin_bit_cast(float, (unsigned)fe << 23) : 0.f;
#pragma unroll
                            for (int a = 0; a < RS; ++a)
#pragma unroll
                                for (int e = 0; e < 16; ++e) acc[a][e] *= f;
                            se = want;
                        }
                        if (c == nch - 1) request_add(s);
                        // ---- matrix phase: filter columns kx; the 2 weight and 8 pixel operands of a column are read while the previous column's
                        // 12 products run; consecutive products go to different accumulators
                        const uint4 *ws = wl + (q & 1) * WST + hl * 32 + l31;
                        const uint4 *ts = tile + (q & 1) * TILE + hl * RS * TWH + 32 * wave + l31;
                        h8 av[2][2], bv[2][RS][2];
                        auto load_ops = [&](int kx, int buf) {
                            av[buf][0] = __builtin_bit_cast(h8, ws[(kx * 2 + 0) * 64]);
                            av[buf][1] = __builtin_bit_cast(h8, ws[(kx * 2 + 1) * 64]);
#pragma unroll
                            for (int a = 0; a < RS; ++a) {
                                bv[buf][a][0] = __builtin_bit_cast(h8, ts[a * TWH + kx]);
                                bv[buf][a][1] = __builtin_bit_cast(h8, ts[PART + a * TWH + kx]);
                            }
                        };
                        load_ops(0, 0);
#pragma unroll
                        for (int kx = 0; kx < KS; ++kx) {
                            const int b = kx & 1;
                            if (kx + 1 < KS) load_ops(kx + 1, b ^ 1);
                            asm volatile("" ::: "memory");
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int a = 0; a < RS; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[b][0], bv[b][a][1], acc[a], 0, 0, 0);
#pragma unroll
                            for (int a = 0; a < RS; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[b][1], bv[b][a][0], acc[a], 0, 0, 0);
#pragma unroll
                            for (int a = 0; a < RS; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[b][0], bv[b][a][0], acc[a], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        TOSTAMP(t_work);
                    }
                    // ---- the step's D rows enter the ring at true scale.  Accumulator a = input row RS s + a (relative to ys - P), element e:
                    // matrix row m = 4 j + i with j = 2 (e >> 2) + half, i = e & 3 - i < 3: (cout i, filter row j); i == 3: (cout j, filter row
                    // 8) for j < 3.  Output row u = RS s + a - ky (relative to ys) = ring index a - ky + 2 P + half.
                    const float fin = inv_sw * __builtin_bit_cast(float, (unsigned)(127 - se) << 23);
#pragma unroll
                    for (int a = 0; a < RS; ++a) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int i = e & 3, g2 = 2 * (e >> 2);
                            const float val = acc[a][e] * fin;
                            if (i < 3) {
                                if (g2 < KS) {                           // filter row g2 + half (5 taps: row 5 of the upper half carries zero weights)
                                    ring[i][a - g2 + 2 * P] += val;
                                }
                            } else if (KS == 9) {
                                // filter row 8: element (g = 0): cout 0 in the lower half, cout 1 in the upper one; (g = 1): cout 2, lower half only
                                if (g2 == 0) {
                                    ring[0][a - 8 + 2 * P] += hl ? 0.f : val;
                                    ring[1][a - 8 + 2 * P + 1] += hl ? val : 0.f;
                                } else if (g2 == 2) {
                                    ring[2][a - 8 + 2 * P] += hl ? 0.f : val;
                                }
                            }
                        }
                    }
                }
                // ---- the group of output rows whose last input row this step held: rows ys + RS (s - LAG) + {0 .. 3} = ring rows 0 .. 3 of the
                // lower half + 1 .. 4 of the upper half, exchanged between the two lanes of a column; the lower half stores rows 0, 1, the
                // upper one rows 2, 3.  Then the ring moves up by RS rows.
                const int gi = s - LAG;
                if (gi >= 0 && RS * gi < rows) {
                    const int epi = d.epilogue;
                    const float floor_ = (epi & RISP_EPI_RELU) ? 0.f : -__builtin_inff();
                    const float *bias = (epi & RISP_EPI_NOBIAS) ? nullptr : d.bias + (size_t)it.g * d.bias_gs;
                    float tot[3][RS];
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int r = 0; r < RS; ++r) {
                            const float mine = to_sel(hl, ring[c][r + 1], ring[c][r]);
                            const float other = __shfl_xor(mine, 32);
                            tot[c][r] = hl ? other + mine : mine + other;       // (lower half's share first, in both lanes)
                        }
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int r2 = 0; r2 < 2; ++r2) {
                            const int oy = it.ys + RS * gi + 2 * hl + r2;
                            if (c < d.cout && oy < it.ye && ox < d.W) {
                                const size_t pix = (size_t)oy * d.W + ox;
                                float o = to_sel(hl, tot[c][2 + r2], tot[c][r2]) + (bias ? bias[c] : 0.f);
                                if (HAS_ADD) o += addv[c][r2];
                                o = o < floor_ ? floor_ : o;             // ReLU, or nothing (floor = -inf); a NaN stays a NaN
                                d.y[((size_t)it.n * d.cout + c) * hw + pix] = o;
                            }
                        }
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
#pragma unroll
                    for (int r = 0; r + RS < NR; ++r) ring[c][r] = ring[c][r + RS];
#pragma unroll
                    for (int r = NR - RS; r < NR; ++r) ring[c][r] = 0.f;
                }
            }
        }
        TOSTAMP(t_tail);
    }
#ifdef RISP_TO_STAMPS
    if (lane == 0 && d.cvals) {                        // diagnostic build: cycle shares of a wave's life (tools/tapout_stamps.py)
        unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<float *>(d.cvals)) + 4 * ((size_t)blockIdx.x * 8 + wave);
        o[0] = t_wait; o[1] = t_work; o[2] = t_tail; o[3] = TO_T() - t_start;
    }
#endif
}

// rows of a segment: the whole image when there are work items enough for the chip without cutting it (every cut costs 2 P halo rows of
// staging and products), else segments of at least 32 rows - or what the caller fixes (a multiple of 4: inference launches, where a
// result must not depend on the batch it travels in - the per-tile scales follow the segment's row phase)
int tapout_seg_rows(int N, int H, int W, int seg_rows) {
    if (seg_rows > 0) return seg_rows >= H ? H : seg_rows;
    const int strips = (W + TO_TW - 1) / TO_TW, slots = h2_cu_count();
    long long items = (long long)N * strips;
    int segs = 1;
    while (items * segs < slots && (H + 2 * segs - 1) / (2 * segs) >= 32) segs *= 2;
    int s = ((H + segs - 1) / segs + 3) & ~3;
    return s >= H ? H : s;
}

template <int KS, bool HAS_ADD, bool PSUM>
int launch_tapout(const risp_conv_desc &d, int seg_rows, float *psum, void *stream) {
    using G = TOG<KS>;
    auto kern = &conv_tapout_kernel<KS, HAS_ADD, PSUM>;
    if (G::LDS_BYTES > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES) != hipSuccess) {
        risp_set_error("risp_conv2d_tapout: cannot raise the dynamic LDS limit to %d bytes", G::LDS_BYTES);
        return 2;
    }
    const int S = tapout_seg_rows(d.N, d.H, d.W, seg_rows);
    const int strips = (d.W + TO_TW - 1) / TO_TW, segs = (d.H + S - 1) / S;
    const long long nitems = (long long)d.N * strips * segs;
    if (nitems > 0x7fffffff) {
        risp_set_error("risp_conv2d_tapout: too many work items");
        return 1;
    }
    const int slots = h2_cu_count();
    const int grid = nitems < slots ? (int)nitems : slots;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), G::LDS_BYTES, (hipStream_t)stream, d, strips, segs, S, (int)nitems, psum);
    RISP_LAUNCH_CHECK("risp_conv2d_tapout");
    return 0;
}
}  // namespace

extern "C" {

size_t risp_conv_tapout_wpack_bytes(int cin, int ksize) { return 16 + (size_t)((cin + 15) / 16) * ksize * 2 * 2 * 32 * 16; }

int risp_conv_tapout_seg_rows(int N, int H, int W) { return tapout_seg_rows(N, H, W, 0); }

int risp_conv_tapout_items(int N, int H, int W, int seg_rows) {
    const int S = tapout_seg_rows(N, H, W, seg_rows);
    return ((W + TO_TW - 1) / TO_TW) * ((H + S - 1) / S);
}

static int conv2d_tapout_impl(const risp_conv_desc *dp, int seg_rows, float *psum, void *stream) {
    RISP_CHECK_ARG(dp, "risp_conv2d_tapout: null descriptor");
    const risp_conv_desc &d = *dp;
    RISP_CHECK_ARG(d.x && d.wpack && d.y, "risp_conv2d_tapout: null tensor");
    RISP_CHECK_GROUP(d, "risp_conv2d_tapout");
    RISP_CHECK_ARG(d.N > 0 && d.H > 0 && d.W > 0 && d.W % 4 == 0 && d.cin > 0 && d.cin % 16 == 0 && d.cout > 0 && d.cout <= 3 &&
                       (d.ksize == 5 || d.ksize == 9) && (unsigned long long)d.cin * d.H * d.W < (1ull << 30) &&
                       (unsigned long long)d.H * d.W < (1ull << 24),
                   "risp_conv2d_tapout: needs a 5x5 or 9x9 layer with cout <= 3, cin %% 16 == 0, W %% 4 == 0, fewer than 2^30 input elements and 2^24 "
                   "pixels per image (N=%d H=%d W=%d cin=%d cout=%d k=%d)",
                   d.N, d.H, d.W, d.cin, d.cout, d.ksize);
    RISP_CHECK_ARG(seg_rows >= 0 && seg_rows % 4 == 0, "risp_conv2d_tapout: seg_rows must be 0 (chosen by the launch) or a multiple of 4");
    RISP_CHECK_ARG(d.load_mode == RISP_LOAD_PLAIN, "risp_conv2d_tapout: only plain loads");
    RISP_CHECK_ARG(!(d.epilogue & ~(RISP_EPI_RELU | RISP_EPI_ADD | RISP_EPI_NOBIAS)), "risp_conv2d_tapout: epilogue %d not supported", d.epilogue);
    RISP_CHECK_ARG((d.epilogue & RISP_EPI_NOBIAS) || d.bias, "risp_conv2d_tapout: bias missing");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_ADD) || (d.add && d.add_c > 0), "risp_conv2d_tapout: add tensor missing");
    RISP_CHECK_ARG(((reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.wpack)) & 15) == 0, "risp_conv2d_tapout: x and wpack must be 16-byte aligned");
    RISP_CHECK_ARG(!psum || (d.cin == 64 && d.ksize == 9), "risp_conv2d_tapout_sums: needs a 9x9 layer with 64 input channels");
    const bool add = (d.epilogue & RISP_EPI_ADD) != 0;
    if (d.ksize == 9) {
        if (psum) return add ? launch_tapout<9, true, true>(d, seg_rows, psum, stream) : launch_tapout<9, false, true>(d, seg_rows, psum, stream);
        return add ? launch_tapout<9, true, false>(d, seg_rows, psum, stream) : launch_tapout<9, false, false>(d, seg_rows, psum, stream);
    }
    return add ? launch_tapout<5, true, false>(d, seg_rows, psum, stream) : launch_tapout<5, false, false>(d, seg_rows, psum, stream);
}

int risp_conv2d_tapout(const risp_conv_desc *dp, int seg_rows, void *stream) { return conv2d_tapout_impl(dp, seg_rows, nullptr, stream); }

/* ... and, on the way, the sum of every input channel over every work item's own pixels: psum [N][risp_conv_tapout_items][64] floats -
 * what risp_rect_sums_tiles finishes into the rectangle sums of the constant-plane gradient (srcnn_res_arch.py:41-46) */
int risp_conv2d_tapout_sums(const risp_conv_desc *dp, int seg_rows, float *psum, void *stream) {
    RISP_CHECK_ARG(psum, "risp_conv2d_tapout_sums: needs the buffer of partial sums");
    return conv2d_tapout_impl(dp, seg_rows, psum, stream);
}

}  // extern "C"
