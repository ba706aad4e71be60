// Split-precision 9x9 FIRST layer over 3 input channels (SRCNNRes once its broadcast planes are folded out, srcnn_res_arch.py:18,
// 41-46) with the filter row's taps as the reduction index - all of it (round 6; serves risp_conv2d_toep_first[_exact] for cin == 3).
// The band kernel of risp_conv_toep_first.hip slides 8 windows of a zero-padded filter row over a 16-pixel window of ONE channel:
// 9 of its 16 reduction slots carry a tap.  Here the reduction index of one filter row is (channel, tap): 27 values in 32 slots,
//        out[co][y][x] = sum_ky sum_q W[co][ky][q] E[q][y + ky - 4][x],   E[(c, kx)][r][x] = in[c][r][x + kx - 4],
// ONE v_mfma_f32_16x16x32_f16 per filter row, 16 pixels, 16 couts and product.  The expanded input E - per pixel four 16-byte slots: the
// 8-tap windows of the three channels and the three ninth taps - costs 128 bytes per pixel and row, so it lives in a RING of 16 input
// rows of a 32-pixel strip in LDS and a persistent workgroup walks DOWN its strip: every input row is expanded once.  The weights of the
// layer (72 KB) stay in LDS.  Matrix work: 0.67 of the band form's.  One scale per work item - the largest magnitude of the image's strip -
// because an accumulator sums over rows that were staged at different times.
//
// Kernel.  ONE workgroup of 8 waves per CU in two TEAMS of four; waves w and w + 4 share a SIMD.  Work item = (image, strip of 32 columns,
// segment of rows); row block b = 4 input rows; group g = 4 output rows, needs blocks g .. g + 2; phase p computes group p.
//   the team p & 1 COMPUTES group p: wave = a quarter of the couts (16) x 4 rows x 32 pixels = 8 accumulators of 4 registers; it walks the
//       group's 12 input rows - a row's two pixel operands are read once and serve the (output row, filter row) pairs with t + ky = r, the
//       filter rows' operands roll through 5 register sets: 72 LDS reads per 216 matrix instructions;
//   the other team SERVES: the epilogue of the group it computed in the previous phase (a lane holds 4 consecutive pixels of a cout: 16-byte
//       stores) and the expansion of block p + 3 (thread = (row, pixel, slot pair): 16 loads requested two phases earlier, 4 16-byte LDS
//       writes) - vector work that issues beside its SIMD partner's matrix instructions.
// The teams swap roles every phase; one barrier per phase.  (s_setprio for the computing team: no change, 1.92 against 1.88 ms.  The first form of this kernel - 4 consumer waves of 32x32x16 instructions and
// 4 producer waves - ran every SIMD's matrix pipe half of the time at best: a consumer's epilogue cost as much as its products.  2.13 ms
// per grouped launch of config 3 against the band form's 2.97.)
#include "risp_f16x2.h"
#include <type_traits>

namespace {
constexpr int XW_TW = 32, XW_RING = 16, XW_KS = 9, XW_P = 4;
constexpr int XW_ROW = 2 * 4 * XW_TW;                               // slots of a ring row: [part][slot][pixel]
constexpr int XW_WST = 4 * XW_KS * 2 * 4 * 16;                     // weight slots: [cout quarter][ky][part][slot][cout]
constexpr int XW_TS = 36;                                            // floats of a cout row of a wave's transposition scratch (32 pixels + 4: bank spread)
constexpr int XW_TL = 511;                                          // entries of the workgroup's own tie list (behind its counter)
constexpr int XW_LDS_BYTES = (XW_RING * XW_ROW + XW_WST) * 16 + 64 + 2 * 64 * 4 + 8 * 16 * XW_TS * 4 + (1 + XW_TL) * 4 + 64 * XW_KS * 4;      // ... + the maxima + per cout: bias, interior border-case value + the scratch + the tie list + the border-case values of an interior row
static_assert(XW_LDS_BYTES <= 160 * 1024, "one workgroup per CU");
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define XW_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define XW_BARRIER_ALL() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ int xw_border_case(int v, int L) { return v < XW_P ? v : (v >= L - XW_P ? 2 * XW_P - (L - 1 - v) : XW_P); }

struct XwItem {
    int n, x0, ys, ye, g;                                            // image, strip corner, output rows [ys, ye), group member
};

template <bool CASEB, bool TIES>
__global__ __launch_bounds__(512, 2) void conv_xwin_kernel(const risp_conv_desc d, int strips, int segs, int seg_rows, int nitems,
                                                           unsigned *__restrict__ ties, unsigned max_ties) {
    constexpr int TW = XW_TW, RING = XW_RING, KS = XW_KS, P = XW_P, ROW = XW_ROW;
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    uint4 *ring = smem, *wl = smem + RING * ROW;
    float *red = reinterpret_cast<float *>(wl + XW_WST);               // [0..7] the waves' maxima
    float *btab = red + 16;                                             // [64] bias, [64] the interior border-case value of the item's image

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int team = wave >> 2, cq = wave & 3;
    float *tsc = btab + 128 + wave * (16 * XW_TS);                      // this wave's [16 couts][32 pixels] scratch: the epilogue's transposition
    // ties (TIES): an item's near-zero pre-activations are collected in LDS and handed to the global list once per item.  (One global
    // atomic per tie made a serving wave wait out a device-scope round trip, and its team's phase with it: 1.84 -> 2.39 ms at the ~5e-5
    // ties per output of a training forward.)
    unsigned *tl = reinterpret_cast<unsigned *>(btab + 128 + 8 * (16 * XW_TS));
    float *bt9 = reinterpret_cast<float *>(tl + 1 + XW_TL);             // [64 couts][9 column cases] of row case 4 (every row but the image's first and last 4)
    if (TIES && tid == 0) tl[0] = 0u;                                   // (the item loop opens with a barrier)
    const size_t hw = (size_t)d.H * d.W;
    const unsigned hw4 = (unsigned)hw * 4u;
    // a workgroup takes a run of consecutive items: the strips of an image, the images of a member - its weights stay in LDS
    const int per_wg = (nitems + gridDim.x - 1) / gridDim.x;
    const int t_first = blockIdx.x * per_wg, t_end = t_first + per_wg < nitems ? t_first + per_wg : nitems;
    if (t_first >= nitems) return;
    const int per = strips * segs;
    auto locate = [&](int t, XwItem &r) {
        r.n = t / per;
        const int q = t - r.n * per, st = q / segs, sg = q - st * segs;
        r.x0 = st * TW;
        r.ys = sg * seg_rows;
        r.ye = r.ys + seg_rows < d.H ? r.ys + seg_rows : d.H;
        r.g = d.group_n > 0 ? r.n / d.group_n : 0;
    };
    int g_loaded = -1;                                                  // member whose weights are in LDS
#ifdef RISP_XW_STAMPS
    unsigned long long t_pre = 0, t_wait = 0, t_work = 0, t_epi = 0, t0 = __builtin_amdgcn_s_memtime(), t1;
    const unsigned long long t_start = t0;
#define XWSTAMP(acc_) do { t1 = __builtin_amdgcn_s_memtime(); acc_ += t1 - t0; t0 = t1; } while (0)
#else
#define XWSTAMP(acc_) do { } while (0)
#endif
    // staging task of a thread inside its team: row j of a block, pixel, slot pair sp (0: the windows of channels 0, 1; 1: the window of
    // channel 2 and the ninth taps of the three channels)
    const int st = tid & 255, sj = st >> 6, spx = st & 31, sp = (st >> 5) & 1;

    for (int t = t_first; t < t_end; ++t) {
        XwItem it;
        locate(t, it);
        const int rows = it.ye - it.ys, ng = (rows + 3) >> 2, nblocks = ng + 2;
        const int nx = (d.group_flags & RISP_GROUP_SHARED_X) ? it.n - it.g * d.group_n : it.n;
        const float *xin = d.x + (size_t)nx * 3 * hw;
        const uint4 *wp = reinterpret_cast<const uint4 *>(d.wpack + (size_t)it.g * d.wpack_gs);
        XW_BARRIER_ALL();                                               // every wave has left the previous item: ring, weights and maxima are free
        // ---- the member's weights into LDS (from the pack of risp_conv_toep_first_wpack_bytes: [cout block][ci][ky][part][taps 0-7 | tap
        // 8][cout][8]) and the largest magnitude of the strip (all rows of the image, the strip's columns + 4 each side)
        if (it.g != g_loaded) {
            for (int s = tid; s < XW_WST; s += 512) {
                const int c = s & 15, slot = (s >> 4) & 3, part = (s >> 6) & 1, rest = s >> 7, ky = rest % KS, q4 = rest / KS;
                const int co = 16 * q4 + c, cb = co >> 5, cl = co & 31;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (co >= d.cout) {
                } else if (slot < 3) {
                    v = wp[1 + ((((size_t)cb * 3 + slot) * KS + ky) * 2 + part) * 64 + cl];
                } else {                                                // the ninth taps of the three channels
                    unsigned short q[3];
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch)
                        q[ch] = *reinterpret_cast<const unsigned short *>(wp + 1 + ((((size_t)cb * 3 + ch) * KS + ky) * 2 + part) * 64 + 32 + cl);
                    v = make_uint4((unsigned)q[0] | ((unsigned)q[1] << 16), (unsigned)q[2], 0u, 0u);
                }
                wl[s] = v;
            }
            g_loaded = it.g;
        }
        if (tid < 64) {
            const bool has = tid < d.cout;
            btab[tid] = (has && !(d.epilogue & RISP_EPI_NOBIAS)) ? d.bias[(size_t)it.g * d.bias_gs + tid] : 0.f;
            btab[64 + tid] = (CASEB && has) ? d.cvals[((size_t)it.n * d.cout + tid) * (KS * KS) + XW_P * KS + XW_P] : 0.f;
        }
        if (CASEB) {
            for (int s = tid; s < 64 * KS; s += 512) {
                const int co = s / KS, xc = s - co * KS;
                bt9[s] = co < d.cout ? d.cvals[((size_t)it.n * d.cout + co) * (KS * KS) + XW_P * KS + xc] : 0.f;
            }
        }
        {
            float m = 0.f;
            const int cols = TW + 2 * P;                                // 40 columns = 10 quads per row and channel
            for (int i = tid; i < 3 * d.H * (cols / 4); i += 512) {
                const int q = i % (cols / 4), rc = i / (cols / 4), y = rc % d.H, c = rc / d.H;
                const int gx = it.x0 - P + 4 * q;
                if (gx >= 0 && gx < d.W) m = amax4(m, *reinterpret_cast<const float4 *>(xin + (size_t)c * hw + (size_t)y * d.W + gx));
            }
            m = h2_wave_max(m);
            if (lane == 0) red[wave] = m;
        }
        XW_BARRIER_ALL();
        float tmax = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) tmax = fmaxf(tmax, red[i]);
        int se = 141 - (int)(__builtin_bit_cast(unsigned, tmax) >> 23);
        se = __builtin_amdgcn_readfirstlane(se);
        se = se > 100 ? 100 : se;                                       // an all-zero or denormal strip: any scale will do
        XWSTAMP(t_pre);

        // ---- staging: byte offsets (column + channel plane) of the thread's two slots; out of range -> 0x80000000: the buffer returns zeros
        const float sc = __builtin_bit_cast(float, (unsigned)(127 + se) << 23);
        const int gx0 = it.x0 + spx - P;
        unsigned offa[8], offb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool in = gx0 + e >= 0 && gx0 + e < d.W, in8 = gx0 + 8 >= 0 && gx0 + 8 < d.W;
            offa[e] = in ? 4u * (unsigned)(gx0 + e) + (sp ? 2u * hw4 : 0u) : 0x80000000u;
            offb[e] = sp ? ((e < 3 && in8) ? 4u * (unsigned)(gx0 + 8) + (unsigned)e * hw4 : 0x80000000u) : (in ? 4u * (unsigned)(gx0 + e) + hw4 : 0x80000000u);
        }
        const __amdgpu_buffer_rsrc_t rx = h2_rsrc(xin);
        float va[8], vb[8];
        auto fetch = [&](int b) {                                       // block b: input rows ys - 4 + 4 b + j
            // (the row is the wave's: its byte offset travels as the load's scalar offset, a row outside the image takes a uniform branch
            // around the 16 loads.  As a per-lane select of the address hipcc built a divergent branch pair around EVERY load: 144
            // instructions of the ~450 a serving phase issues beside the other team's products)
            const int y = it.ys - P + 4 * b + (wave & 3);
            if (b < nblocks && y >= 0 && y < d.H) {
                const unsigned ro = 4u * (unsigned)(y * d.W);
#pragma unroll
                for (int e = 0; e < 8; ++e) {                           // (an invalid column stays >= 2^31: the range check does not see the scalar offset)
                    va[e] = h2_load4(rx, offa[e], ro);
                    vb[e] = h2_load4(rx, offb[e], ro);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) va[e] = vb[e] = 0.f;
            }
        };
        auto stage = [&](int b) {
            if (b >= nblocks) return;
            uint4 h0, l0, h1, l1;
            split8(va, sc, h0, l0);
            split8(vb, sc, h1, l1);
            uint4 *row = ring + ((4 * b + sj) % RING) * ROW + spx;
            row[(0 * 4 + 2 * sp) * TW] = h0;
            row[(0 * 4 + 2 * sp + 1) * TW] = h1;
            row[(1 * 4 + 2 * sp) * TW] = l0;
            row[(1 * 4 + 2 * sp + 1) * TW] = l1;
        };
        // ---- epilogue constants
        const float inv_sw = *reinterpret_cast<const float *>(wp);
        const float fin = inv_sw * __builtin_bit_cast(float, (unsigned)(127 - se) << 23);
        // |error| of a sum here <~ 2^-22 x (sum of |terms|) <= 2^-22 x 81 x max|w| x 3 tmax; inv_sw 2^15 >= max|w|: a margin of 2 more bits
        const float tau = 3.f * tmax * inv_sw * (81.f * 32768.f / 1048576.f);
        const float floor_ = (d.epilogue & RISP_EPI_RELU) ? 0.f : -__builtin_inff();
        const __amdgpu_buffer_rsrc_t rc = h2_rsrc(CASEB ? d.cvals + (size_t)it.n * d.cout * (KS * KS) : d.x);
        const __amdgpu_buffer_rsrc_t ry = h2_rsrc(d.y + (size_t)it.n * d.cout * hw);
        const unsigned tbase = (unsigned)((size_t)it.n * d.cout * hw);
        const int q8 = lane & 7, c8 = lane >> 3, px_ = it.x0 + 4 * q8;   // the epilogue's lane: 4 pixels of cout c8 of a pass
        int xcase[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xcase[i] = xw_border_case(px_ + i < d.W ? px_ + i : d.W - 1, d.W);
        const uint4 *ws = wl + (size_t)cq * (KS * 2 * 4 * 16) + lane;   // + (ky * 2 + part) * 64: slot lane >> 4, cout lane & 15
        f32x4 acc[4][2];                                                // [output row][half of the strip]; a lane: pixels 16 hp + 4 (lane >> 4) + {0 .. 3}

        // prologue: blocks 0 and 1 by the two teams, block 2 by team 0; team 1 serves in phase 0 (block 3), team 0 in phase 1 (block 4)
        fetch(team);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stage(team);
        if (team == 0) {
            fetch(2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stage(2);
            fetch(4);
        } else {
            fetch(3);
        }
        XW_BARRIER_LDS();
        XWSTAMP(t_work);
        for (int p = 0; p <= ng; ++p) {
            if ((p & 1) == team) {
                if (p < ng) {
                    // ======================================================================================= compute group p
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int hp = 0; hp < 2; ++hp) acc[a][hp] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const int rbase = (4 * p) % RING;
                    h8 av[5][2], bv[2][2][2];              // (5 weight sets: the row ahead lands while the 4 in use are read); [buffer][half][part]
                    auto load_a = [&](int ky) {
                        av[ky % 5][0] = __builtin_bit_cast(h8, ws[(ky * 2 + 0) * 64]);
                        av[ky % 5][1] = __builtin_bit_cast(h8, ws[(ky * 2 + 1) * 64]);
                    };
                    auto load_b = [&](int r, int buf) {
                        int rr = rbase + r;
                        rr = rr >= RING ? rr - RING : rr;
                        const uint4 *row = ring + rr * ROW + kg * TW + l15;
#pragma unroll
                        for (int hp = 0; hp < 2; ++hp) {
                            bv[buf][hp][0] = __builtin_bit_cast(h8, row[16 * hp]);
                            bv[buf][hp][1] = __builtin_bit_cast(h8, row[4 * TW + 16 * hp]);
                        }
                    };
                    load_a(0);
                    load_b(0, 0);
#pragma unroll
                    for (int r = 0; r < 12; ++r) {
                        if (r + 1 < 12) load_b(r + 1, (r + 1) & 1);
                        if (r + 1 < KS) load_a(r + 1);
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        // rows of the product = pixels (the expanded input is the first operand), columns = couts
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            const int ky = r - a;
                            if (ky >= 0 && ky < KS) {
#pragma unroll
                                for (int hp = 0; hp < 2; ++hp) acc[a][hp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bv[r & 1][hp][1], av[ky % 5][0], acc[a][hp], 0, 0, 0);
                            }
                        }
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            const int ky = r - a;
                            if (ky >= 0 && ky < KS) {
#pragma unroll
                                for (int hp = 0; hp < 2; ++hp) acc[a][hp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bv[r & 1][hp][0], av[ky % 5][1], acc[a][hp], 0, 0, 0);
                            }
                        }
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            const int ky = r - a;
                            if (ky >= 0 && ky < KS) {
#pragma unroll
                                for (int hp = 0; hp < 2; ++hp) acc[a][hp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bv[r & 1][hp][0], av[ky % 5][0], acc[a][hp], 0, 0, 0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                XWSTAMP(t_work);
            } else {
                // =========================================================================================== serve
                // ---- the expansion of block p + 3 (requested two phases ago: nothing younger than its loads is in flight except this team's
                // stores of two phases ago), then the request of block p + 5 - ahead of the epilogue's stores
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef XW_NO_STAGE                                          /* (diagnostic builds, wrong results: what a part of the serving phase costs) */
                stage(p + 3);
                fetch(p + 5);
#endif
                XWSTAMP(t_work);
#ifndef XW_NO_EPI
                if (p >= 1 && p - 1 < ng) {
                    // ---- epilogue of group g = p - 1 (computed by this team in the previous phase).  The product leaves a lane 4 consecutive
                    // pixels of cout (lane & 15) - 64-byte pieces of 16 cout rows per store; through the wave's LDS scratch a lane takes 4 pixels
                    // of cout (lane >> 3) instead: 8 lanes = the strip's 32 pixels = one 128-byte line per cout (the 16-byte stores in 64-byte
                    // pieces cost 0.6 of the kernel's 2.17 ms).
                    // The border-case value of a pixel: from the LDS table of the interior row case - unless the group holds one of the image's
                    // first or last 4 rows (a wave-uniform test), whose values are read from memory.  Two instances of the code: hipcc waits
                    // where the paths of a branch around a load meet, for every vector-memory operation in flight - the previous STORE included
                    // (800 cycles per store, 6500 of a serving phase's 9000 while the other team's products take 4000).
                    const int g = p - 1;
                    const bool gedge = CASEB && (it.ys + 4 * g < XW_P || it.ys + 4 * g + 3 >= d.H - XW_P);
                    auto epilogue = [&](auto edge_tag) {
                        constexpr bool EDGE = decltype(edge_tag)::value;
                        // per cout of the lane's two passes: bias, and (rows of the interior case) the border-case values of its 4 pixels
                        float bb[2], tvi[2][4];
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            const int cc = 16 * cq + 8 * half + c8;
                            bb[half] = btab[cc];
#pragma unroll
                            for (int i = 0; i < 4; ++i) tvi[half][i] = (CASEB && !EDGE) ? bt9[cc * KS + xcase[i]] : 0.f;
                        }
                        // the transposition, one row ahead: row a + 1 goes into the scratch behind the reads of row a (a wave's LDS operations
                        // complete in order) while row a is finished and stored - one exposed LDS round trip per phase instead of four
                        f32x4 raw[2][2];
                        auto put_get = [&](int a) {
#pragma unroll
                            for (int hp = 0; hp < 2; ++hp) *reinterpret_cast<f32x4 *>(tsc + l15 * XW_TS + 16 * hp + 4 * kg) = acc[a][hp];
#pragma unroll
                            for (int half = 0; half < 2; ++half) raw[a & 1][half] = *reinterpret_cast<const f32x4 *>(tsc + (8 * half + c8) * XW_TS + 4 * q8);
                        };
                        put_get(0);
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            if (a + 1 < 4) put_get(a + 1);
                            const int oy = it.ys + 4 * g + a;
                            const bool rok = oy < it.ye;
                            const int ycase = EDGE ? xw_border_case(oy < d.H ? oy : d.H - 1, d.H) : XW_P;
#pragma unroll
                            for (int half = 0; half < 2; ++half) {
                                const int cc = 16 * cq + 8 * half + c8;                   // this lane's cout of the pass
                                const bool ck = cc < d.cout;
                                float tv[4];
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    if (CASEB && EDGE) tv[i] = h2_load4(rc, (ck && px_ + i < d.W) ? 4u * (unsigned)(cc * (KS * KS) + ycase * KS + xcase[i]) : 0x80000000u, 0u);
                                    else tv[i] = tvi[half][i];
                                }
                                float o[4];
#pragma unroll
                                for (int i = 0; i < 4; ++i) {      // (product, + bias, + border-case value: each rounded - the order of the band kernel; a fused
                                    float v = raw[a & 1][half][i] * fin + bb[half];      // form moved darts_step_kf5's it0_alpha_grad1 across its 6.1e-6 budget)
                                    if (CASEB) v += tv[i];
                                    o[i] = v;
                                }
                                // a pre-activation this close to zero has no reliable sign in fp32: listed for toep_first_ties_kernel
                                const bool near = TIES && fminf(fminf(fabsf(o[0]), fabsf(o[1])), fminf(fabsf(o[2]), fabsf(o[3]))) < tau;
                                if (TIES && __builtin_amdgcn_ballot_w64(near && rok && ck) != 0) {      // (rare: a wave-uniform test first)
#pragma unroll
                                    for (int i = 0; i < 4; ++i)
                                        if (rok && ck && px_ + i < d.W && fabsf(o[i]) < tau) {
                                            const unsigned idx = tbase + (unsigned)cc * (unsigned)hw + (unsigned)(oy * d.W + px_ + i);   // (< 2^32 outputs: checked by the entry point)
                                            const unsigned ls = __hip_atomic_fetch_add(tl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                            if (ls < (unsigned)XW_TL) {
                                                tl[1 + ls] = idx;
                                            } else {                    // the item's list is full (a flat image region): straight to the global list
                                                const unsigned slot = atomicAdd(ties, 1u);
                                                if (slot < max_ties) ties[1 + slot] = idx;
                                            }
                                        }
                                }
#pragma unroll
                                for (int i = 0; i < 4; ++i) o[i] = o[i] < floor_ ? floor_ : o[i];
                                const unsigned vo = (rok && ck && px_ < d.W) ? 4u * (unsigned)(oy * d.W + px_) + (unsigned)cc * hw4 : 0x80000000u;
#ifndef XW_NO_STORE
                                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__builtin_bit_cast(unsigned, o[0]), __builtin_bit_cast(unsigned, o[1]), __builtin_bit_cast(unsigned, o[2]),
                                                                             __builtin_bit_cast(unsigned, o[3])}, ry, vo, 0u, 0);
#else
                                if (o[0] + o[1] + o[2] + o[3] == 123.456f) __builtin_amdgcn_raw_buffer_store_b32(0u, ry, vo, 0u, 0);
#endif
                            }
                        }
                    };
                    if (gedge) epilogue(std::true_type{});
                    else epilogue(std::false_type{});
                }
#else
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    asm volatile("" :: "v"(acc[a][0]), "v"(acc[a][1]));      // (the products stay alive)
                }
#endif
                XWSTAMP(t_epi);
            }
            XW_BARRIER_LDS();
            XWSTAMP(t_wait);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (TIES && wave == 0) {                                        // (behind the last phase's barrier: every append of the item is in LDS)
            const unsigned all = tl[0], cnt = all < (unsigned)XW_TL ? all : (unsigned)XW_TL;
            if (cnt) {
                unsigned base = 0u;
                if (lane == 0) base = atomicAdd(ties, cnt);
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                for (unsigned i = lane; i < cnt; i += 64u)
                    if (base + i < max_ties) ties[1 + base + i] = tl[1 + i];
            }
            if (lane == 0) tl[0] = 0u;                                  // (LDS operations of a wave complete in order; the next item opens with a barrier)
        }
    }
#ifdef RISP_XW_STAMPS
    if (lane == 0 && max_ties == 0xABCDu && ties) {      // diagnostic build: cycle shares of a wave's life (tools/xwin_stamps.py)
        unsigned long long *o = reinterpret_cast<unsigned long long *>(ties) + 8 * ((size_t)blockIdx.x * 8 + wave);
        o[0] = t_pre; o[1] = t_wait; o[2] = t_work; o[3] = t_epi; o[4] = __builtin_amdgcn_s_memtime() - t_start;
    }
#endif
}

// rows of a segment: whole strips when the launch has work items enough for the chip, else segments of at least 32 rows (a multiple
// of 4).  A result does not depend on the choice: the scale is the image strip's, the order of a sum the filter's.
int xwin_seg_rows(int N, int H, int W) {
    const int strips = (W + XW_TW - 1) / XW_TW, slots = h2_cu_count();
    long long items = (long long)N * strips;
    int segs = 1;
    while (items * segs < slots && (H + 2 * segs - 1) / (2 * segs) >= 32) segs *= 2;
    int s = ((H + segs - 1) / segs + 3) & ~3;
    return s >= H ? H : s;
}
}  // namespace

int risp_launch_xwin(const risp_conv_desc &d, unsigned *ties, unsigned max_ties, void *stream) {
    const bool cb = (d.epilogue & RISP_EPI_CASEBIAS) != 0;
    auto kern = ties ? (cb ? &conv_xwin_kernel<true, true> : &conv_xwin_kernel<false, true>) : (cb ? &conv_xwin_kernel<true, false> : &conv_xwin_kernel<false, false>);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, XW_LDS_BYTES) != hipSuccess) {
        risp_set_error("risp_conv2d_toep_first: cannot raise the dynamic LDS limit to %d bytes", XW_LDS_BYTES);
        return 2;
    }
    const int S = xwin_seg_rows(d.N, d.H, d.W);
    const int strips = (d.W + XW_TW - 1) / XW_TW, segs = (d.H + S - 1) / S;
    const long long nitems = (long long)d.N * strips * segs;
    if (nitems > 0x7fffffff) {
        risp_set_error("risp_conv2d_toep_first: too many work items");
        return 1;
    }
    const int slots = h2_cu_count();
    const int grid = nitems < slots ? (int)nitems : slots;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), XW_LDS_BYTES, (hipStream_t)stream, d, strips, segs, S, (int)nitems, ties, max_ties);
    RISP_LAUNCH_CHECK("risp_conv2d_toep_first");
    return 0;
}
