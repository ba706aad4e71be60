// Split-precision convolution, wave-specialised form (round 5).  Same arithmetic, same pack, same tile geometry and the same bits
// as conv_f16x2_kernel<KS, NT> (risp_conv_f16x2.hip: read that file's header for the arithmetic and the LDS slot layout); what changes
// is WHO does what.  Layers: the 64 -> 64 3x3 layers of Path-Restore (path_14l_bgr_arch.py:6-21, 58-86; path_14l_bayer_arch.py:59-88)
// and the 5x5 64 -> 32 layer of SRCNNRes (srcnn_res_arch.py:20), forward and backward-data.
//
// Why.  In the round-4 kernel every wave staged its share of the input tile (load, scale, split, 8 ds_write_b128), waited at two
// barriers per chunk and one per filter row, issued the LDS-DMA of the weights and ran the matrix instructions: the step loops
// alone ran at 98 % of the matrix pipe's rate but were 48 % of a wave's life, and two such workgroups per CU reached 0.59
// matrix-pipe busy (profiles/r04_conv_pmc.txt).  Here ONE workgroup of 8 waves owns a CU:
//   waves 0-3  CONSUMERS: matrix instructions and the store epilogue, nothing else.  A wave owns rows (w, w + 4) x 64 pixels x 32 NT
//              couts as before.  3x3: one barrier per CHUNK of 16 input channels (216 matrix instructions at NT = 2) instead of two per
//              chunk plus one per filter row; 5x5: two per chunk (filter rows 0-2 | 3-4: 180 | 120 matrix instructions).
//   waves 4-7  PRODUCERS: two chunks ahead they load the halo tile to registers (8 + 2 P x 16-byte buffer loads per thread) and find its
//              largest magnitude; one chunk ahead they scale, split and write it into the other LDS tile buffer and send the filter rows
//              by LDS-DMA - their vector instructions issue beside the consumers' matrix instructions (separate pipes), and the HBM
//              latency of a chunk runs beside a whole phase.
// 3x3, phase g (between barrier g and g + 1): consumers multiply chunk g out of tile[g & 1] and weight slots 3 (g & 1) ..; producers
// write chunk g + 1 (its maxima published in red[(g + 1) & 3] before barrier g) into tile[(g + 1) & 1] - the consumers left that
// buffer at barrier g -, send the three filter rows of chunk g + 1 into the other three weight slots, load chunk g + 2 and publish
// its per-wave maxima in red[(g + 2) & 3].  5x5: five weight slots, filter row ky in slot ky; while the consumers are in rows 0-2 of
// chunk g the producers send rows 3-4 of chunk g (slots the consumers left at the chunk's start), while they are in rows 3-4 the
// producers send rows 0-2 of chunk g + 1.  Chunks run on across tiles: the staging of a tile's first chunk overlaps the previous
// tile's last matrix phase and its epilogue.  LDS: 3x3 NT = 2: 2 x 43.5 KB of tile + 6 x 12 KB of weights = 157.6 KB; 5x5: 2 x 51 + 5 x
// 10 = 152.6 KB of the CU's 160.
// The consumers read every distinct pixel operand of a filter row ONCE (column shift u = pixel tile + tap: 6 for 3 taps - the
// round-4 3x3 loop read 12 -, 8 for 5) as one stream over the filter rows of a phase, pixel and weight operands two steps ahead of
// their first use, across filter rows.  Every accumulator still receives its products in the round-4 order (chunk, filter row, tap;
// x_lo w_hi, x_hi w_lo, x_hi w_hi): results are bit-identical (tests/test_gpu_f16x2.py).
//
// SPLIT epilogue (3x3, 64 couts, with a residual and / or a mask to read: the launches whose epilogue is a quarter to a third of a
// consumer's life).  The tile's two cout blocks are multiplied one after the other in the tile's first and last chunk (the pixel
// operands are read once per block there), and the epilogue travels in 32 PIECES (one cout row per lane, one 16-byte store): pieces
// 0-15 (block 0) ride on the 18 steps of block 1's products in the last chunk, pieces 16-31 (block 1) on block 0's products of the
// NEXT tile's first chunk, whose sums start from a zero operand instead of zeroed registers.  Residual / mask rows come through a
// ring of 8 (4 + 4) pieces, requested in front of the store of the piece that frees the slot; a lane outside the image carries an
// out-of-range buffer offset instead of a branch.  Per accumulator the order of the products is unchanged: same bits.  Measured
// (tools/ws_stamps.py, tools/ab_ws_build.py, 32 x 256 x 256): consumer cycles per tile 42.5k -> 38.7k (residual + ReLU), 51.5k -> 44.7k
// (residual + mask), the epilogue's share of their life 0.25 / 0.37 -> 0.01 - but the launch gains 2.5-5.5 % only (460 -> 436 us, 533 ->
// 509 us): the chip is POWER-limited under this kernel and answers the denser instruction stream with a lower clock (1.58 -> 1.47
// GHz, 1.73 -> 1.54 GHz).  A launch of 4 images gains 8-9 %.  Without a residual or mask the form LOSES 3 % (the extra operand reads
// of the two block-wise chunks cost more than the 0.12 of the life they hide): those launches keep the tile's epilogue behind its
// last chunk.
#include "risp_f16x2.h"

namespace {
constexpr int WS_TH = 8, WS_TW = 64, WS_S = 17, WS_RS = 4 * WS_S, WS_CK = 16;

template <int KS, int NT>
struct WS {
    static constexpr int P = KS / 2, IH = WS_TH + 2 * P, NU = KS + 3;   // NU: column shifts u = t + kx of a filter row
    static constexpr int PART = 2 * IH * WS_RS;                  // 16-byte slots of one part (hi or lo): [channel half][row][slot]
    static constexpr int TILE = 2 * PART;
    static constexpr int WST = KS * 2 * 2 * NT * 32;             // weight slots of one filter row: [kx][part][channel half][cout]
    static constexpr int NSLOT = KS == 3 ? 6 : 5;                // filter-row slots in LDS
    static constexpr int NH = KS == 3 ? 1 : 2;                   // phases (barriers) per chunk
    static constexpr int LDS_BYTES = (2 * TILE + NSLOT * WST) * 16 + 4 * 16 + 2 * 256;     // + maxima (4 chunks x 4 waves) + bias (x 2)
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
    static_assert(WST % 64 == 0, "filter rows in whole LDS-DMA pieces");
    static_assert(KS == 3 || (KS == 5 && NT == 1), "3x3 with one or two cout blocks per wave, 5x5 with one");
};

#ifdef RISP_WS_STAMPS
#define WS_T() __builtin_amdgcn_s_memtime()
#else
#define WS_T() 0ull
#endif
#ifndef WS_SPLIT
#define WS_SPLIT 1                                                  /* 0: diagnostic builds - the 3x3 NT = 2 epilogue behind the tile's last chunk */
#endif
#define WS_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define WS_BARRIER_ALL() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")

struct WsTile {
    int n, cb, x0, y0, g;                                               // image, cout block, corner, group member
};

// exponent a chunk asks for: its tile's largest magnitude (per producer wave in red4[0..3]) into [2^14, 2^15)
__device__ __forceinline__ int ws_want(const float *red4) {
    const float4 mx = *reinterpret_cast<const float4 *>(red4);
    const float tmax = fmaxf(fmaxf(mx.x, mx.y), fmaxf(mx.z, mx.w));
    int eb = (int)(__builtin_bit_cast(unsigned, tmax) >> 23);
    eb = __builtin_amdgcn_readfirstlane(eb);
    const int want = 141 - eb;
    return want > 100 ? 100 : want;                                     // an all-zero or denormal tile: any scale will do
}

template <int KS, int NT, bool HAS_ADD, bool HAS_MASK>
__global__ __launch_bounds__(512, 2) void conv_f16x2_ws_kernel(const risp_conv_desc d, int tiles_x, int tiles_y, int ncb, int ntiles) {
    using C = WS<KS, NT>;
    constexpr int P = C::P, IH = C::IH, NU = C::NU, S = WS_S, RS = WS_RS, WST = C::WST, NH = C::NH;
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    uint4 *tile = smem, *wl = smem + 2 * C::TILE;
    float *red = reinterpret_cast<float *>(wl + C::NSLOT * WST);       // [chunk & 3][producer wave]
    float *lbias = red + 16;                                           // [tile parity][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunks = d.cin / WS_CK;
    const size_t hw = (size_t)d.H * d.W;
    const unsigned hw4 = (unsigned)hw * 4u;                            // max(cin, cout) * H * W * 4 < 2^31: checked by the entry point
    const int nwg = gridDim.x;
    const int wg = (nwg & 7) == 0 ? (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    if (wg >= ntiles) return;
    const int my_tiles = (ntiles - wg + nwg - 1) / nwg;
    const int G = my_tiles * nchunks;                                  // chunks of this workgroup: NH (G + 2) barriers
    auto locate = [&](int t, WsTile &r) {
        r.cb = t % ncb;
        const int q = t / ncb;
        r.n = q / (tiles_x * tiles_y);
        const int rem = q - r.n * (tiles_x * tiles_y), ty = rem / tiles_x;
        r.x0 = (rem - ty * tiles_x) * WS_TW;
        r.y0 = ty * WS_TH;
        r.g = d.group_n > 0 ? r.n / d.group_n : 0;
    };

    if (wave >= 4) {
        // =============================================================================================== producers
        // staging tasks of a thread, as in conv_f16x2_kernel.  Pass 1 = the 8 interior rows: (8 channels, row, quad) = 8 x 16-byte loads,
        // 8 slots.  Pass 2 = the 2 P halo rows as (channel pair, row, quad) tasks and the halo columns as (channel pair, row, column) tasks.
        const int pt = tid - 256, pw = wave - 4;
        constexpr int NC2 = (IH * 2 * P * 8 + 255) / 256;
        const int g1 = pt >> 7, r1 = (pt >> 4) & 7, q1 = pt & 15;
        const int dst1 = (g1 * IH + r1 + P) * RS;
        unsigned off1, offr[P], offc[NC2];
        bool ok1, okr[P], okc[NC2];
        int dstr[P], dstc[NC2];
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int id = pt + 256 * k, hr = (id >> 4) % (2 * P), cp = id / (32 * P);
            dstr[k] = (((cp >> 2) * IH + (hr < P ? hr : WS_TH + hr)) * RS) * 16 + (cp & 3) * 4;
        }
#pragma unroll
        for (int k = 0; k < NC2; ++k) {
            const int id = pt + 256 * k;
            const int cp = id / (IH * 2 * P), rem = id - cp * (IH * 2 * P), ir = rem / (2 * P), cc = rem - ir * (2 * P);
            const int c = cc < P ? cc : WS_TW + cc;
            dstc[k] = id < IH * 2 * P * 8 ? (((cp >> 2) * IH + ir) * RS + (c & 3) * S + (c >> 2)) * 16 + (cp & 3) * 4 : -1;
        }
        __amdgpu_buffer_rsrc_t rx;
        auto setup_load = [&](const WsTile &r) {                       // staging addresses of tile r
            const int nx = (d.group_flags & RISP_GROUP_SHARED_X) ? r.n - r.g * d.group_n : r.n;
            rx = h2_rsrc(d.x + (size_t)nx * d.cin * hw);
            const int x0 = r.x0, y0 = r.y0;
            const int gy1 = y0 + r1, gx1 = x0 + 4 * q1;
            ok1 = gy1 < d.H && gx1 < d.W;
            off1 = ok1 ? 8u * g1 * hw4 + 4u * (unsigned)(gy1 * d.W + gx1) : 0u;
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const int id = pt + 256 * k, q = id & 15, hr = (id >> 4) % (2 * P), cp = id / (32 * P);
                const int gy = y0 - P + (hr < P ? hr : WS_TH + hr), gx = x0 + 4 * q;
                okr[k] = gy >= 0 && gy < d.H && gx < d.W;
                offr[k] = okr[k] ? 2u * cp * hw4 + 4u * (unsigned)(gy * d.W + gx) : 0u;
            }
#pragma unroll
            for (int k = 0; k < NC2; ++k) {
                const int id = pt + 256 * k;
                const int cp = id / (IH * 2 * P), rem2 = id - cp * (IH * 2 * P), ir = rem2 / (2 * P), cc = rem2 - ir * (2 * P);
                const int gy = y0 - P + ir, gx = x0 - P + (cc < P ? cc : WS_TW + cc);
                okc[k] = dstc[k] >= 0 && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
                offc[k] = okc[k] ? 2u * cp * hw4 + 4u * (unsigned)(gy * d.W + gx) : 0u;
            }
        };
        // two register sets: the loads of chunk g + 2 are issued at the START of chunk g into one while chunk g + 1 is split and written
        // out of the other
        struct Raw {
            float4 v1[8], vr[P][2];
            float vc[NC2][2];
        };
        Raw raw0, raw1;
        auto fetch = [&](Raw &R, int ch) {
            const unsigned off = (unsigned)ch * WS_CK * hw4;
#pragma unroll
            for (int j = 0; j < 8; ++j) R.v1[j] = h2_load16(rx, off1, off + j * hw4);
#pragma unroll
            for (int k = 0; k < P; ++k) {
                R.vr[k][0] = h2_load16(rx, offr[k], off);
                R.vr[k][1] = h2_load16(rx, offr[k], off + hw4);
            }
#pragma unroll
            for (int k = 0; k < NC2; ++k) {
                R.vc[k][0] = h2_load4(rx, offc[k], off);
                R.vc[k][1] = h2_load4(rx, offc[k], off + hw4);
            }
        };
        auto slot_of = [&](int c) { return (c & 3) * S + (c >> 2); };
        // (the registers hold zeros wherever the tile reaches outside the image: masked when the loads have arrived - the chunk is written a
        // phase later, possibly after the addresses moved on to the next tile)
        auto put_quad = [&](uint4 *tb, const float4 (&v)[8], int dst, int c0, float s) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = comp(v[e], j);
                uint4 hi, lo;
                split8(a, s, hi, lo);
                const int sl = dst + slot_of(c0 + j);
                tb[sl] = hi;
                tb[C::PART + sl] = lo;
            }
        };
        auto put_pair = [&](uint4 *tb, float a0, float a1, float s, int byte_off) {
            a0 *= s;
            a1 *= s;
            const h2 hh = {(_Float16)a0, (_Float16)a1};
            const h2 ll = {(_Float16)(a0 - (float)hh[0]), (_Float16)(a1 - (float)hh[1])};
            char *base = reinterpret_cast<char *>(tb) + byte_off;
            *reinterpret_cast<unsigned *>(base) = __builtin_bit_cast(unsigned, hh);
            *reinterpret_cast<unsigned *>(base + C::PART * 16) = __builtin_bit_cast(unsigned, ll);
        };
        // filter rows ky0 .. ky0 + nky - 1 of chunk ch into consecutive weight slots from slot0: pieces of 64 LDS slots dealt over the four
        // producer waves; slot L = 64 piece + lane of the run sits in row L / (32 NT) of [filter row][kx][part][channel half] and takes
        // this tile's cout block out of the pack's rows (which hold all ncb of them)
        const int row_slots = ncb * NT * 32;
        const unsigned lds_wl = lds_addr_of(wl);
        auto issue_weights = [&](const WsTile &r, int ch, int ky0, int nky, int slot0) {
            const uint4 *src = reinterpret_cast<const uint4 *>(d.wpack + (size_t)r.g * d.wpack_gs) + 1 +
                               (size_t)(ch * KS + ky0) * (KS * 4) * row_slots + r.cb * NT * 32;
            const int npiece = nky * (WST / 64);
            for (int piece = pw; piece < npiece; piece += 4) {
                const int L = piece * 64 + lane, row = L / (NT * 32), col = L - row * (NT * 32);
                lds_dma16_m(src, 16u * (unsigned)(row * row_slots + col), lds_wl + 16u * (unsigned)(slot0 * WST + piece * 64));
            }
        };

        WsTile tl, tw;                                                 // tile of the chunk being loaded / written
        int kl = 0, chl = 0, kw = 0, chw = 0, se = 0;
        locate(wg, tl);
        tw = tl;
        setup_load(tl);
#ifdef RISP_WS_STAMPS
        unsigned long long t_work = 0, t_bar = 0, t_a = WS_T(), t_b;
#endif
        auto close_phase = [&]() {
#ifdef RISP_WS_STAMPS
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            t_b = WS_T(); t_work += t_b - t_a;
#endif
            WS_BARRIER_ALL();                                          // pieces landed, tile and maxima written
#ifdef RISP_WS_STAMPS
            t_a = WS_T(); t_bar += t_a - t_b;
#endif
        };
        // the producers' share of chunk g (g = -2, -1: filling the pipeline): RL receives chunk g + 2, RW (loaded during chunk g - 1) is
        // written as chunk g + 1
        auto chunk = [&](int g, Raw &RL, Raw &RW) {
            const int w = g + 1, l = g + 2;
            const bool wr = w >= 0 && w < G, ld = l < G;
            if (KS == 3) {
                if (wr) issue_weights(tw, chw, 0, 3, 3 * (w & 1));     // in flight while the tile is split and written
            } else if (g >= 0) {
                // rows 3-4 of chunk g, whose tile is tw's predecessor in the walk: (tile, chunk) of g = (tw, chw) stepped back by one
                WsTile tg = tw;
                int chg = chw - 1;
                if (chg < 0) {
                    chg = nchunks - 1;
                    locate(wg + (kw - 1) * nwg, tg);
                }
                issue_weights(tg, chg, 3, 2, 3);
            }
            if (ld) fetch(RL, chl);
            if (wr) {
                const int want = ws_want(red + 4 * (w & 3));
                se = (chw == 0 || want < se) ? want : se;              // the running exponent of the tile (the consumers keep the same)
                const float s = __builtin_bit_cast(float, (unsigned)(127 + se) << 23);
                uint4 *tb = tile + (w & 1) * C::TILE;
#ifndef RISP_WS_ABL_NO_QUAD
                put_quad(tb, RW.v1, dst1, 4 * q1 + P, s);
#endif
                // The halo travels as channel PAIRS: 4-byte LDS writes, four lanes to a bank - ALL of the kernel's bank conflicts (13 % of
                // the LDS's active cycles, tools/ws_conflicts.sh: 0 with this block compiled out).  Nobody waits for them: the LDS is a
                // quarter busy and the producers have a third of every phase to spare.  Whole-slot halo tasks (8 channels per lane, 16-byte
                // writes: conflict share 0.006) were built and measured - every producer wave then runs the split / write sequence twice
                // (the second time on 16 lanes), the producers become the critical path (0.66 -> 0.90 of their life) and the launch is
                // 6-8 % SLOWER (400 -> 434 us): not kept.
#ifndef RISP_WS_ABL_NO_HALO       /* PMC diagnostics only (wrong results) */
#pragma unroll
                for (int k = 0; k < P; ++k) {
                    const int q = (pt + 256 * k) & 15;
#pragma unroll
                    for (int j = 0; j < 4; ++j) put_pair(tb, comp(RW.vr[k][0], j), comp(RW.vr[k][1], j), s, dstr[k] + 16 * slot_of(4 * q + j + P));
                }
#pragma unroll
                for (int k = 0; k < NC2; ++k)
                    if (dstc[k] >= 0) put_pair(tb, RW.vc[k][0], RW.vc[k][1], s, dstc[k]);
#endif
                if (chw == 0 && pt < 32 * NT) {
                    const int co = tw.cb * 32 * NT + pt;
                    lbias[(kw & 1) * 64 + pt] = (d.epilogue & RISP_EPI_NOBIAS) || co >= d.cout ? 0.f : d.bias[(size_t)tw.g * d.bias_gs + co];
                }
            }
            if (KS == 5) {
                close_phase();                                         // ---- the consumers move on to filter rows 3-4 of chunk g
                if (wr) issue_weights(tw, chw, 0, 3, 0);               // rows 0-2 of chunk g + 1
            }
            if (wr && ++chw == nchunks) {
                chw = 0;
                if (++kw < my_tiles) locate(wg + kw * nwg, tw);
            }
            if (ld) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the chunk's values have arrived (issued a phase or more ago)
                float m = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    RL.v1[j] = ok1 ? RL.v1[j] : make_float4(0.f, 0.f, 0.f, 0.f);
                    m = amax4(m, RL.v1[j]);
                }
#pragma unroll
                for (int k = 0; k < P; ++k) {
                    RL.vr[k][0] = okr[k] ? RL.vr[k][0] : make_float4(0.f, 0.f, 0.f, 0.f);
                    RL.vr[k][1] = okr[k] ? RL.vr[k][1] : make_float4(0.f, 0.f, 0.f, 0.f);
                    m = amax4(amax4(m, RL.vr[k][0]), RL.vr[k][1]);
                }
#pragma unroll
                for (int k = 0; k < NC2; ++k) {
                    RL.vc[k][0] = okc[k] ? RL.vc[k][0] : 0.f;
                    RL.vc[k][1] = okc[k] ? RL.vc[k][1] : 0.f;
                    m = fmaxf(m, fmaxf(fabsf(RL.vc[k][0]), fabsf(RL.vc[k][1])));
                }
                m = h2_wave_max(m);
                if (lane == 0) red[4 * (l & 3) + pw] = m;
                if (++chl == nchunks) {
                    chl = 0;
                    if (++kl < my_tiles) {
                        locate(wg + kl * nwg, tl);
                        setup_load(tl);
                    }
                }
            }
            close_phase();
        };
        for (int g = -2; g < G; g += 2) {
            chunk(g, raw0, raw1);
            if (g + 1 < G) chunk(g + 1, raw1, raw0);
        }
#ifdef RISP_WS_STAMPS
        if (lane == 0 && d.cvals) {
            unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<float *>(d.cvals)) + 8 * ((size_t)blockIdx.x * 8 + wave);
            o[0] = t_work; o[1] = t_bar; o[2] = 0; o[3] = 0; o[4] = 0;
        }
#endif
        return;
    }

    // =================================================================================================== consumers
    const int l31 = lane & 31, hl = lane >> 5;
    // B = pixels: lane (n = lane & 31, hl) of pixel tile t holds row wave + 4 (n >> 4), column 4 (n & 15) + t; at column shift
    // u = t + kx its slot is (u & 3) * S + (n & 15) + (u >> 2).  A = weights: lane (m = lane & 31, hl): cout m of a block, channels 8 hl ...
    // (a wave's two rows are FOUR apart: a 16-byte LDS read is free of bank conflicts only if lanes 16-31 sit a multiple of 256 bytes
    // from lanes 0-15 - tools/lds_bank_probe.hip)
    const int bbase = (hl * IH + wave + 4 * (l31 >> 4)) * RS + (l31 & 15);
    const int abase = hl * NT * 32 + l31;
#ifdef RISP_WS_STAMPS
    unsigned long long c_bar = 0, c_mat = 0, c_epi = 0, c_head = 0, c0 = WS_T(), c1;
    const unsigned long long c_start = c0, rt0 = __builtin_amdgcn_s_memrealtime();
#define WS_LAP(acc) do { __builtin_amdgcn_sched_barrier(0); c1 = WS_T(); acc += c1 - c0; c0 = c1; } while (0)
#else
#define WS_LAP(acc) do { } while (0)
#endif
    // ---- epilogue: y = epilogue(acc * 2^-se / s_w + bias), as in conv_f16x2_kernel, in PIECES: piece p = one cout row per lane (cout
    // 32 (p >> 4) + 8 ((p >> 2) & 3) + (p & 3) + 4 hl of the tile's block), one 16-byte store of four consecutive pixels, 16 lanes = 256
    // contiguous bytes.  gfx9 counts loads and stores in ONE in-order counter: a load issued behind a store returns only when that store
    // has completed.  The residual / mask rows therefore travel through a ring of R pieces, piece p + R requested in front of the store
    // of piece p.
    constexpr bool SPLIT = WS_SPLIT && KS == 3 && NT == 2 && (HAS_ADD || HAS_MASK || WS_SPLIT > 1);            // the tile's two cout blocks finish half a chunk apart (header)
    constexpr int NPIECE = 16 * NT, R = (HAS_ADD && HAS_MASK) ? 4 : 8;
    unsigned hw4e = hw4;
    asm volatile("" : "+s"(hw4e));
    const float floor_ = (d.epilogue & RISP_EPI_RELU) ? 0.f : -__builtin_inff();
    // the tile the pieces belong to.  A lane outside the image carries an offset beyond the buffer's range (h2_rsrc: 2^31 - 1 bytes):
    // its loads return zeros and its stores are dropped by the range check - no branch in the instruction stream.  SPLIT: until the
    // first tile's last chunk "no tile": every lane out of range
    unsigned e_loff = 0x80000000u, e_hw4 = 0;
    float e_fin = 0.f;
    const float *e_bias = lbias + 4 * hl;
    __amdgpu_buffer_rsrc_t e_ry = h2_rsrc(d.x), e_ra = e_ry, e_rm = e_ry;
    float4 av[HAS_ADD ? R : 1], mv[HAS_MASK ? R : 1];
#pragma unroll
    for (int i = 0; i < (HAS_ADD ? R : 1); ++i) av[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < (HAS_MASK ? R : 1); ++i) mv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto epi_setup = [&](const WsTile &cur, int k, int se) {
        const int oy = cur.y0 + wave + 4 * (l31 >> 4), ox = cur.x0 + 4 * (l31 & 15);
        const int na = (d.group_flags & RISP_GROUP_SHARED_ADD) ? cur.n - cur.g * d.group_n : cur.n;
        const float inv_sw = *reinterpret_cast<const float *>(d.wpack + (size_t)cur.g * d.wpack_gs);
        e_fin = inv_sw * __builtin_bit_cast(float, (unsigned)(127 - se) << 23);
        e_hw4 = hw4e;
        e_loff = oy < d.H && ox < d.W ? 4u * (unsigned)(oy * d.W + ox) + (unsigned)(cur.cb * 32 * NT + 4 * hl) * hw4 : 0x80000000u;
        e_ry = h2_rsrc(d.y + (size_t)cur.n * d.cout * hw);
        e_ra = h2_rsrc(HAS_ADD ? d.add + (size_t)na * d.add_c * hw : d.x);
        e_rm = h2_rsrc(HAS_MASK ? d.mask + (size_t)cur.n * d.cout * hw : d.x);
        e_bias = lbias + (k & 1) * 64 + 4 * hl;
    };
    auto epi_fetch = [&](int p) {
        unsigned h = e_hw4;
        asm volatile("" : "+s"(h));                                     // (the offsets of 32 pieces are not worth 32 registers)
        const unsigned co = (unsigned)((p >> 4) * 32 + 8 * ((p >> 2) & 3) + (p & 3)) * h;
        if (HAS_ADD) av[p % R] = h2_load16(e_ra, e_loff, co);
        if (HAS_MASK) mv[p % R] = h2_load16(e_rm, e_loff, co);
    };
    // (the bias of piece p is read from LDS two pieces ahead: LDS reads return in order, a piece that waited for its own read would
    // wait for every operand read in flight)
    float bq[3];
    auto epi_bias = [&](int p) { bq[p % 3] = e_bias[(p >> 4) * 32 + 8 * ((p >> 2) & 3) + (p & 3)]; };
    auto epi_piece = [&](const f32x16 (&acc)[4][NT], int p) {
        const int b = p >> 4, e = p & 15, cu = b * 32 + 8 * (e >> 2) + (e & 3);
        const float bb = bq[p % 3];
        float4 o = make_float4(acc[0][b][e] * e_fin + bb, acc[1][b][e] * e_fin + bb, acc[2][b][e] * e_fin + bb, acc[3][b][e] * e_fin + bb);
        float4 a4, mk;
        if (HAS_ADD) a4 = av[p % R];
        if (HAS_MASK) mk = mv[p % R];
        if ((HAS_ADD || HAS_MASK) && p + R < NPIECE) epi_fetch(p + R);
        if (HAS_ADD) {
            o.x += a4.x; o.y += a4.y; o.z += a4.z; o.w += a4.w;
        }
        o.x = o.x < floor_ ? floor_ : o.x;                              // ReLU, or nothing (floor = -inf); a NaN stays a NaN (torch.relu)
        o.y = o.y < floor_ ? floor_ : o.y;
        o.z = o.z < floor_ ? floor_ : o.z;
        o.w = o.w < floor_ ? floor_ : o.w;
        if (HAS_MASK) {
            o.x = mk.x > 0.f ? o.x : 0.f;
            o.y = mk.y > 0.f ? o.y : 0.f;
            o.z = mk.z > 0.f ? o.z : 0.f;
            o.w = mk.w > 0.f ? o.w : 0.f;
        }
        // (plane offset in the VECTOR offset: see risp_conv_f16x2.hip - a 16-byte buffer store reads its data registers late)
        unsigned h = e_hw4;
        asm volatile("" : "+s"(h));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), e_ry, e_loff + (unsigned)cu * h, 0, 0);
    };

#pragma unroll
    for (int i = 0; i < 2 * NH; ++i) WS_BARRIER_LDS();                 // chunks -2 and -1: the producers fill the pipeline
    WS_LAP(c_bar);
    int g = 0;
    int want_next = ws_want(red);                                      // chunk 0's exponent; later ones are read a phase ahead (below)
    f32x16 acc[4][NT];
    for (int k = 0; k < my_tiles; ++k) {
        WsTile cur;
        locate(wg + k * nwg, cur);
        if constexpr (!SPLIT) {                                        // (SPLIT: the first product of every sum starts from a zero operand)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int b = 0; b < NT; ++b)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[t][b][e] = 0.f;
        }
        int se = 0;                                                    // running exponent: the accumulators hold sum * 2^se * s_w
        const uint4 *wsb, *tsb;
        h8 a[KS][NT][2], bv[3][2];
        // a chunk's head: its exponent (published by the producers a phase ago), the running sums rescaled if it is smaller, its buffers
        auto head = [&](int ch) {
            const int want = want_next;
            // the next chunk's maxima were published a phase ago (visible since the barrier that opened this phase): read them now, off
            // the critical path of the next phase's head (after the walk's last chunk: a stale row, unused)
            want_next = ws_want(red + 4 * ((g + 1) & 3));
            if (ch == 0) {
                se = want;
            } else if (want < se) {                                    // larger values than before: rescale the running sums (exact)
                const int fe = 127 + want - se;
                const float f = fe > 0 ? __builtin_bit_cast(float, (unsigned)fe << 23) : 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int b = 0; b < NT; ++b)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[t][b][e] *= f;
                se = want;
            }
            wsb = wl + (KS == 3 ? 3 * (g & 1) * WST : 0) + abase;      // filter row ky of the chunk: slot ky from here
            tsb = tile + (g & 1) * C::TILE + bbase;
        };
        // filter rows [ky0, ky0 + nky): steps q = NU (ky - ky0) + u; step q multiplies the pixel operand of shift u with the taps
        // kx = u - t of the pixel tiles t.  Operands are read two steps ahead of their first use: the pixel operand of step q + 2, and
        // tap kx of filter row ky in front of step NU (ky - ky0) + kx - 2 (a tap's registers are free again after shift kx + 3).
        // BSEL: both cout blocks (-1) or one; ZERO: the sums start here (a tile's first chunk); EPI: piece EPI - 1 + q of the
        // epilogue rides on step q (the pieces belong to the OTHER block, or to the previous tile).
        auto rows = [&](auto KY0, auto NKY, auto BSEL_, auto ZERO_, auto EPI_) {
            constexpr int ky0 = decltype(KY0)::value, nq = decltype(NKY)::value * NU, BSEL = decltype(BSEL_)::value, EPI = decltype(EPI_)::value;
            constexpr bool ZERO = decltype(ZERO_)::value;
            auto load_a = [&](int ky, int kx) {
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    if (BSEL < 0 || b == BSEL)
#pragma unroll
                        for (int part = 0; part < 2; ++part)
                            a[kx][b][part] = __builtin_bit_cast(h8, wsb[ky * WST + ((kx * 2 + part) * 2) * NT * 32 + b * 32]);
            };
            auto load_b = [&](int q) {
                const int ky = ky0 + q / NU, u = q % NU, sl = ky * RS + (u & 3) * S + (u >> 2);
                bv[q % 3][0] = __builtin_bit_cast(h8, tsb[sl]);
                bv[q % 3][1] = __builtin_bit_cast(h8, tsb[C::PART + sl]);
            };
            load_a(ky0, 0);
            load_b(0);
            load_a(ky0, 1);
            load_b(1);
            if constexpr (EPI > 0) {
                epi_bias(EPI - 1);
                epi_bias(EPI);
            }
            __builtin_amdgcn_sched_barrier(0);
            WS_LAP(c_head);
#pragma unroll
            for (int q = 0; q < nq; ++q) {
                const int u = q % NU;
                if (q + 2 < nq) {
                    load_b(q + 2);
                    if ((q + 2) % NU < KS) load_a(ky0 + (q + 2) / NU, (q + 2) % NU);
                }
                if constexpr (EPI > 0)
                    if (q + 2 < 16) epi_bias(EPI - 1 + q + 2);
                // keep the reads of the later steps in front of this step's products (hipcc sinks them to their first use otherwise)
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int kx = u - t;
                    if (kx >= 0 && kx < KS) {
                        const bool first = ZERO && q < NU && kx == 0;       // filter row 0, tap 0: the first product of acc[t][.]
#pragma unroll
                        for (int b = 0; b < NT; ++b)
                            if (BSEL < 0 || b == BSEL) {
                                f32x16 c0;
                                if (first) {
#pragma unroll
                                    for (int e = 0; e < 16; ++e) c0[e] = 0.f;
                                } else {
                                    c0 = acc[t][b];
                                }
                                acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kx][b][0], bv[q % 3][1], c0, 0, 0, 0);
                            }
#pragma unroll
                        for (int b = 0; b < NT; ++b)
                            if (BSEL < 0 || b == BSEL) acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kx][b][1], bv[q % 3][0], acc[t][b], 0, 0, 0);
#pragma unroll
                        for (int b = 0; b < NT; ++b)
                            if (BSEL < 0 || b == BSEL) acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kx][b][0], bv[q % 3][0], acc[t][b], 0, 0, 0);
                    }
                }
                if constexpr (EPI > 0)
                    if (q < 16) epi_piece(acc, EPI - 1 + q);
                __builtin_amdgcn_sched_barrier(0);
            }
            WS_LAP(c_mat);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I3 = std::integral_constant<int, 3>;
        using IALL = std::integral_constant<int, -1>;
        using NO = std::false_type;
        if constexpr (SPLIT) {
            // One cout block after the other in a tile's first and last chunk (the pixel operands are read once per block there): the
            // epilogue of block 0 rides on block 1's products of the last chunk, that of block 1 on block 0's products of the NEXT
            // tile's first chunk - the matrix pipe does not wait for the stores.  (nchunks >= 2: the entry point sees to it.)  Straight
            // code per tile - first chunk, loop over the middle ones, last chunk - so that the accumulators never meet at a join.
            head(0);
            rows(I0{}, I3{}, I0{}, std::true_type{}, std::integral_constant<int, 17>{});            // pieces 16 .. 31 of the previous tile
            rows(I0{}, I3{}, I1{}, std::true_type{}, I0{});
            ++g;
            WS_BARRIER_LDS();
            WS_LAP(c_bar);
            for (int ch = 1; ch + 1 < nchunks; ++ch, ++g) {
                head(ch);
                rows(I0{}, I3{}, IALL{}, NO{}, I0{});
                WS_BARRIER_LDS();
                WS_LAP(c_bar);
            }
            head(nchunks - 1);
            epi_setup(cur, k, se);
            if (HAS_ADD || HAS_MASK) {
#pragma unroll
                for (int p = 0; p < R; ++p) epi_fetch(p);
            }
            rows(I0{}, I3{}, I0{}, NO{}, I0{});
            rows(I0{}, I3{}, I1{}, NO{}, I1{});                                                     // pieces 0 .. 15
            ++g;
            WS_BARRIER_LDS();
            WS_LAP(c_bar);
        } else {
            for (int ch = 0; ch < nchunks; ++ch, ++g) {
                head(ch);
                if constexpr (KS == 3) {
                    rows(I0{}, I3{}, IALL{}, NO{}, I0{});
                } else {
                    rows(I0{}, I3{}, IALL{}, NO{}, I0{});
                    WS_BARRIER_LDS();                                  // rows 3-4 of this chunk have landed; the producers may refill slots 0-2
                    WS_LAP(c_bar);
                    rows(I3{}, std::integral_constant<int, 2>{}, IALL{}, NO{}, I0{});
                }
                if (ch + 1 < nchunks) {                                // (a tile's last chunk: behind the epilogue)
                    WS_BARRIER_LDS();
                    WS_LAP(c_bar);
                }
            }
        }
        if constexpr (!SPLIT) {
            // ---- epilogue: y = epilogue(acc * 2^-se / s_w + bias), as in conv_f16x2_kernel: one 16-byte store per cout and lane, 16 lanes =
            // 256 contiguous bytes of a cout row
            {
                const int oy = cur.y0 + wave + 4 * (l31 >> 4), ox = cur.x0 + 4 * (l31 & 15);
                const int na = (d.group_flags & RISP_GROUP_SHARED_ADD) ? cur.n - cur.g * d.group_n : cur.n;
                const float inv_sw = *reinterpret_cast<const float *>(d.wpack + (size_t)cur.g * d.wpack_gs);
                const float fin = inv_sw * __builtin_bit_cast(float, (unsigned)(127 - se) << 23);
                const float floor_ = (d.epilogue & RISP_EPI_RELU) ? 0.f : -__builtin_inff();
                const bool pixok = oy < d.H && ox < d.W;
                unsigned hw4e = hw4;
                asm volatile("" : "+s"(hw4e));
                const unsigned loff = (pixok ? 4u * (unsigned)(oy * d.W + ox) : 0u) + (unsigned)(cur.cb * 32 * NT + 4 * hl) * hw4;
                const __amdgpu_buffer_rsrc_t ry = h2_rsrc(d.y + (size_t)cur.n * d.cout * hw);
                const __amdgpu_buffer_rsrc_t ra = h2_rsrc(HAS_ADD ? d.add + (size_t)na * d.add_c * hw : d.x);
                const __amdgpu_buffer_rsrc_t rm = h2_rsrc(HAS_MASK ? d.mask + (size_t)cur.n * d.cout * hw : d.x);
                const float *bias_row = lbias + (k & 1) * 64 + 4 * hl;
                // gfx9 counts loads and stores in ONE in-order counter: a load issued behind a store returns only when that store has
                // completed.  The residual / mask rows therefore come in batches through two register sets, the loads of batch k + 1 issued
                // IN FRONT of the stores of batch k: they queue behind the stores of batch k - 1 only, and that round trip passes while batch
                // k is formed and stored (the round-4 epilogue loaded a batch behind the previous batch's stores and waited it out).
                constexpr int EB = (HAS_ADD && HAS_MASK) ? 4 : 8, NB = 16 * NT / EB;
                float4 av[2][HAS_ADD ? EB : 1], mv[2][HAS_MASK ? EB : 1];
                auto fetch_rows = [&](int g2, int buf) {
#pragma unroll
                    for (int kk = 0; kk < EB; ++kk) {
                        const int c = g2 * EB + kk, cu = (c >> 4) * 32 + 8 * ((c >> 2) & 3) + (c & 3);
                        if (HAS_ADD) av[buf][kk] = h2_load16(ra, loff, (unsigned)cu * hw4e);
                        if (HAS_MASK) mv[buf][kk] = h2_load16(rm, loff, (unsigned)cu * hw4e);
                    }
                };
                if (HAS_ADD || HAS_MASK) fetch_rows(0, 0);
#pragma unroll
                for (int g2 = 0; g2 < NB; ++g2) {
                    if ((HAS_ADD || HAS_MASK) && g2 + 1 < NB) fetch_rows(g2 + 1, (g2 + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kk = 0; kk < EB; ++kk) {
                        const int c = g2 * EB + kk, b = c >> 4, j = (c >> 2) & 3, i = c & 3, e = 4 * j + i, cu = b * 32 + 8 * j + i;
                        const float bb = bias_row[cu];
                        float4 o = make_float4(acc[0][b][e] * fin + bb, acc[1][b][e] * fin + bb, acc[2][b][e] * fin + bb, acc[3][b][e] * fin + bb);
                        if (HAS_ADD) {
                            const float4 a4 = av[g2 & 1][kk];
                            o.x += a4.x; o.y += a4.y; o.z += a4.z; o.w += a4.w;
                        }
                        o.x = o.x < floor_ ? floor_ : o.x;                  // ReLU, or nothing (floor = -inf); a NaN stays a NaN (torch.relu)
                        o.y = o.y < floor_ ? floor_ : o.y;
                        o.z = o.z < floor_ ? floor_ : o.z;
                        o.w = o.w < floor_ ? floor_ : o.w;
                        if (HAS_MASK) {
                            const float4 mk = mv[g2 & 1][kk];
                            o.x = mk.x > 0.f ? o.x : 0.f;
                            o.y = mk.y > 0.f ? o.y : 0.f;
                            o.z = mk.z > 0.f ? o.z : 0.f;
                            o.w = mk.w > 0.f ? o.w : 0.f;
                        }
                        // (plane offset in the VECTOR offset: see risp_conv_f16x2.hip - a 16-byte buffer store reads its data registers late)
                        if (pixok) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry, loff + (unsigned)cu * hw4e, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            WS_LAP(c_epi);
            WS_BARRIER_LDS();                                          // end of the tile's last phase (the stores drain on their own)
            WS_LAP(c_bar);
        }
    }
    if constexpr (SPLIT) {                                             // the last tile's second block
        epi_bias(16);
        epi_bias(17);
#pragma unroll
        for (int p = 16; p < 32; ++p) {
            if (p + 2 < 32) epi_bias(p + 2);
            epi_piece(acc, p);
        }
        WS_LAP(c_epi);
    }
#ifdef RISP_WS_STAMPS
    if (lane == 0 && d.cvals) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<float *>(d.cvals)) + 8 * ((size_t)blockIdx.x * 8 + wave);
        o[0] = c_bar; o[1] = c_mat; o[2] = c_epi; o[3] = c_head; o[4] = WS_T() - c_start; o[5] = rt0; o[6] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

template <int KS, int NT, bool HAS_ADD, bool HAS_MASK>
int launch_ws(const risp_conv_desc &d, void *stream) {
    using C = WS<KS, NT>;
    auto kern = &conv_f16x2_ws_kernel<KS, NT, HAS_ADD, HAS_MASK>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess) {
        risp_set_error("risp_conv2d_f16x2: cannot raise the dynamic LDS limit to %d bytes", C::LDS_BYTES);
        return 2;
    }
    const int tx = (d.W + WS_TW - 1) / WS_TW, ty = (d.H + WS_TH - 1) / WS_TH, ncb = d.cout / (32 * NT);
    const long long ntiles = (long long)tx * ty * d.N * ncb;
    if (ntiles > 0x7fffffff) {
        risp_set_error("risp_conv2d_f16x2: too many tiles");
        return 1;
    }
    const int slots = h2_cu_count();                                   // one persistent workgroup per CU
    const int grid = ntiles < slots ? (int)ntiles : slots;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), C::LDS_BYTES, (hipStream_t)stream, d, tx, ty, ncb, (int)ntiles);
    RISP_LAUNCH_CHECK("risp_conv2d_f16x2");
    return 0;
}

template <int KS, int NT>
int launch_ws_epi(const risp_conv_desc &d, void *stream) {
    const bool a = (d.epilogue & RISP_EPI_ADD) != 0, m = (d.epilogue & RISP_EPI_MASK) != 0;
    return a ? (m ? launch_ws<KS, NT, true, true>(d, stream) : launch_ws<KS, NT, true, false>(d, stream))
             : (m ? launch_ws<KS, NT, false, true>(d, stream) : launch_ws<KS, NT, false, false>(d, stream));
}
}  // namespace

// the launches of risp_conv2d_f16x2 (arguments checked there)
int risp_launch_f16x2_ws(const risp_conv_desc &d, void *stream) {
    if (d.ksize == 5) return launch_ws_epi<5, 1>(d, stream);           // one cout block per tile: 64 couts = two tiles per pixel tile
    return d.cout == 64 ? launch_ws_epi<3, 2>(d, stream) : launch_ws_epi<3, 1>(d, stream);
}
