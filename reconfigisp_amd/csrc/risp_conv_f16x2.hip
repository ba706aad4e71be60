// Split-precision convolution on the f16 matrix pipe (round 4): fp32 in, fp32 out, fp32-level accuracy at 16/3 of the
// fp32 matrix-instruction rate.  Layers: the 64 -> 64 3x3 layers of Path-Restore (path_14l_bgr_arch.py:6-21, 58-86;
// path_14l_bayer_arch.py:59-88) and the 5x5 64 -> 32 layer of SRCNNRes with its backward (srcnn_res_arch.py:20).
//
// Arithmetic.  Every fp32 operand v is cut into two halves of 11 significant bits each, hi = rn_f16(v s), lo = rn_f16(v s - hi)
// (s a power of two), so that v s = hi + lo up to 2^-22 |v s|, and a product of two operands is taken as
//        x w  ~  (x_lo w_hi + x_hi w_lo + x_hi w_hi) / (s_x s_w)
// - three v_mfma_f32_32x32x16_f16 instructions (products of f16 values are exact in fp32; accumulation in fp32) instead of
// eight v_mfma_f32_32x32x2_f32 of twice the duration.  The dropped term x_lo w_lo is 2^-22 of the product.  Against float64
// (tests/test_pack_algebra_cpu.py::test_f16x2_split_numerics: 64-channel 3x3 and 5x5 rows, activations in [0,1) and
// gradient-like inputs of magnitude 1e-5) the emulated scheme has HALF the rms error of the fp32 FMA chain the fp32 matrix
// instruction is (5.3e-8 vs 1.1e-7 of max|y|), because its error is per product (random walk) where the chain rounds the
// running sum at every step; tests/test_gpu_f16x2.py compares the kernel itself with float64, beside the fp32 kernels.
//
// Range.  f16 has 5 exponent bits: hi overflows above 65504 and lo loses bits below 2^-14.  Weights are scaled per LAYER at
// pack time (s_w = 2^k puts max|w| into [2^14, 2^15)).  Activations / upstream gradients are scaled per WORKGROUP TILE and
// CHUNK of 16 input channels at staging time: the tile's largest magnitude (DPP butterflies + one LDS row, no extra barrier)
// picks s_x = 2^e with max|x| s_x in [2^14, 2^15); the accumulators carry the running exponent and are rescaled (exact:
// a power of two) only when a later chunk needs a smaller one.  Everything down to 2^-17 of the tile's maximum keeps its 22
// bits, below that the absolute error is 2^-39 of the maximum.  No inter-kernel state, deterministic, and a gradient tensor
// of magnitude 1e-8 is as exact as an activation tensor of magnitude 1.
//
// Kernel.  Workgroup = 4 waves = 8 rows x 64 pixels x NT cout blocks of 32; a wave owns 2 rows (w, w + 4) x 64 pixels = 4 pixel tiles
// of 32 (B operand) x NT cout blocks (A operand): lane (n, half) of pixel tile t holds row w + 4 (n >> 4), column 4 (n & 15) + t, so that
// the four tiles give a lane four CONSECUTIVE pixels of a cout row - 16-byte stores, 16 lanes = 256 contiguous bytes, no LDS
// transposition.  Per chunk of 16 input channels the halo tile is staged through registers (8 x 16-byte loads per thread = 4
// pixels x 8 channels, scaled, split, 8 x ds_write_b128); the weights arrive pre-split by LDS-DMA, one STAGE = one filter row
// of one chunk at a time, through a ring of LDS buffers that runs ahead of the matrix instructions.
// LDS layout: one 16-byte slot = the hi (or lo) halves of 8 channels of one pixel = exactly one lane's B operand.  Within a
// tile row the slot of column c is (c & 3) * 17 + (c >> 2): the staging lanes (4 consecutive columns each) write consecutive
// slots, and the 16 lanes of a pixel tile read 16 consecutive slots for every tap shift: both directions are free of bank
// conflicts.
// PERSISTENT workgroups, 2 per CU, each walks tiles id, id + gridDim.x, ...: a tile's stores are never waited for (they drain
// while the next tile is staged), and nothing runs in chip-wide lockstep rounds.  Consecutive workgroup ids sit on different
// XCDs (8 L2s): XCD k takes the k-th contiguous eighth of each sweep, so that tiles sharing halo rows share an L2.
#include "risp_f16x2.h"

namespace {
constexpr int H2_TH = 8, H2_TW = 64, H2_S = 17, H2_RS = 4 * H2_S, H2_CK = 16;

template <int KS, int NT>
struct H2 {
    static constexpr int P = KS / 2, IH = H2_TH + 2 * P;
    static constexpr int PART = 2 * IH * H2_RS;                  // 16-byte slots of one part (hi or lo): [channel half][row][slot]
    static constexpr int TILE = 2 * PART;
    static constexpr int WST = KS * 2 * 2 * NT * 32;             // weight slots of one stage = one filter row: [kx][part][channel half][cout]
    // weight stages in LDS; the transfers run RING - 1 stages ahead.  3x3: three (80 KB per workgroup, two workgroups per CU);
    // 5x5 (12 tile rows): two.
    static constexpr int RING = KS == 3 ? 3 : 2;
    static constexpr int PW = (WST / 64 + 3) / 4;                 // LDS-DMA instructions per wave and stage (the same for every wave)
    static constexpr int LDS_BYTES = (TILE + RING * WST) * 16 + 64 + 2 * 256;    // tile, weight ring, the row of maxima, the bias (x 2)
    static_assert(WST % 64 == 0, "weight stage in whole LDS-DMA pieces");
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

#ifdef RISP_H2_STAMPS
#define H2STAMP(v) do { __builtin_amdgcn_s_waitcnt(0xC07F); v = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define H2STAMP(v) do { } while (0)
#endif

// HAS_ADD / HAS_MASK: epilogue flags as template parameters - with run-time flags the epilogue is a chain of wave-uniform
// branches around its loads, hipcc spills what the loads return and waits vmcnt(0) behind every block.
// ncb: cout blocks of 32 NT in the tile index (a 64-cout layer on the NT = 1 kernel is two tiles per pixel tile).
template <int KS, int NT, bool HAS_ADD, bool HAS_MASK>
__global__ __launch_bounds__(256, 2) void conv_f16x2_kernel(const risp_conv_desc d, int tiles_x, int tiles_y, int ncb, int ntiles) {
    using C = H2<KS, NT>;
    constexpr int P = C::P, IH = C::IH, S = H2_S, RS = H2_RS, WST = C::WST, RING = C::RING, AHEAD = RING - 1;
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    uint4 *tile = smem, *wl = smem + C::TILE;
    float *red = reinterpret_cast<float *>(wl + RING * WST);         // 4 floats: per-wave maxima of the chunk being staged
    float *lbias = red + 16;                                         // 2 x 64 floats, by tile parity (LDS reads do not queue behind the stores)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hl = lane >> 5;
    const int nchunks = d.cin / H2_CK, nstages = nchunks * KS;
    const size_t hw = (size_t)d.H * d.W;
    const int nwg = gridDim.x;
    const int wg = (nwg & 7) == 0 ? (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3) : blockIdx.x;

    // ---- staging tasks of this thread.  Pass 1 = the 8 interior rows: (8 channels, row, quad) = 8 x 16-byte loads, 8 slots.
    // Pass 2 = the 2 P halo rows as (channel pair, row, quad) tasks - 2 x 16-byte loads, P tasks per thread - and the halo columns
    // as (channel pair, row, column) tasks - 2 scalar loads, NC2 tasks per thread.  (Whole 8-channel slots per thread, as in
    // pass 1, would keep 32 more registers in flight through the matrix phase.)
    constexpr int NC2 = (IH * 2 * P * 8 + 255) / 256;
    const int g1 = tid >> 7, r1 = (tid >> 4) & 7, q1 = tid & 15;
    const int dst1 = (g1 * IH + r1 + P) * RS;                               // + slot of column 4 q + j + P
    unsigned off1, offr[P], offc[NC2];                                      // byte offsets of the tasks inside the image
    bool ok1, okr[P], okc[NC2];
    int dstr[P], dstc[NC2];                                                 // byte offsets into the hi part
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const int id = tid + 256 * k, hr = (id >> 4) % (2 * P), cp = id / (32 * P);
        dstr[k] = (((cp >> 2) * IH + (hr < P ? hr : H2_TH + hr)) * RS) * 16 + (cp & 3) * 4;        // + 16 * slot of column 4 q + j + P
    }
#pragma unroll
    for (int k = 0; k < NC2; ++k) {
        const int id = tid + 256 * k;
        const int cp = id / (IH * 2 * P), rem = id - cp * (IH * 2 * P), ir = rem / (2 * P), cc = rem - ir * (2 * P);
        const int c = cc < P ? cc : H2_TW + cc;
        dstc[k] = id < IH * 2 * P * 8 ? (((cp >> 2) * IH + ir) * RS + (c & 3) * S + (c >> 2)) * 16 + (cp & 3) * 4 : -1;
    }
    // bytes of a plane.  max(cin, cout) * H * W * 4 < 2^31 (checked by the entry point): every lane offset below - pixel + up to 16
    // planes on the way in, pixel + cout plane on the way out - stays inside the 2^31 - 1 bytes that h2_rsrc's range check covers
    const unsigned hw4 = (unsigned)hw * 4u;
    // the tile being staged: image / member, cout block, corner; its tensors as the launch's group layout has them
    struct TileRef {
        int n, cb, x0, y0;
        const uint4 *w;                                                     // the member's pack
    };
    TileRef cur;
    __amdgpu_buffer_rsrc_t rx;
    int parity = 0;
    auto locate = [&](int t, TileRef &r) {
        r.cb = t % ncb;
        const int q = t / ncb;
        r.n = q / (tiles_x * tiles_y);
        const int rem = q - r.n * (tiles_x * tiles_y), ty = rem / tiles_x;
        r.x0 = (rem - ty * tiles_x) * H2_TW;
        r.y0 = ty * H2_TH;
        const int g = d.group_n > 0 ? r.n / d.group_n : 0;
        r.w = reinterpret_cast<const uint4 *>(d.wpack + (size_t)g * d.wpack_gs);
    };
    auto setup = [&](const TileRef &r) {                                    // staging addresses + the bias row of tile r
        const int g = d.group_n > 0 ? r.n / d.group_n : 0;
        const int nx = (d.group_flags & RISP_GROUP_SHARED_X) ? r.n - g * d.group_n : r.n;
        rx = h2_rsrc(d.x + (size_t)nx * d.cin * hw);
        const int x0 = r.x0, y0 = r.y0;
        const int gy1 = y0 + r1, gx1 = x0 + 4 * q1;
        ok1 = gy1 < d.H && gx1 < d.W;
        off1 = ok1 ? 8u * g1 * hw4 + 4u * (unsigned)(gy1 * d.W + gx1) : 0u;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int id = tid + 256 * k, q = id & 15, hr = (id >> 4) % (2 * P), cp = id / (32 * P);
            const int gy = y0 - P + (hr < P ? hr : H2_TH + hr), gx = x0 + 4 * q;
            okr[k] = gy >= 0 && gy < d.H && gx < d.W;
            offr[k] = okr[k] ? 2u * cp * hw4 + 4u * (unsigned)(gy * d.W + gx) : 0u;
        }
#pragma unroll
        for (int k = 0; k < NC2; ++k) {
            const int id = tid + 256 * k;
            const int cp = id / (IH * 2 * P), rem2 = id - cp * (IH * 2 * P), ir = rem2 / (2 * P), cc = rem2 - ir * (2 * P);
            const int gy = y0 - P + ir, gx = x0 - P + (cc < P ? cc : H2_TW + cc);
            okc[k] = dstc[k] >= 0 && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
            offc[k] = okc[k] ? 2u * cp * hw4 + 4u * (unsigned)(gy * d.W + gx) : 0u;
        }
        parity ^= 1;                                    // visible after the tile's first barrier; the previous tile's epilogue reads the other row
        if (tid < 32 * NT) {
            const int co = r.cb * 32 * NT + tid;
            lbias[parity * 64 + tid] = (d.epilogue & RISP_EPI_NOBIAS) || co >= d.cout ? 0.f : d.bias[(size_t)g * d.bias_gs + co];
        }
    };

    float4 v1[8], vr[P][2];
    float vc[NC2][2];
    auto fetch = [&](int ch) {
        const unsigned off = (unsigned)ch * H2_CK * hw4;
#pragma unroll
        for (int j = 0; j < 8; ++j) v1[j] = h2_load16(rx, off1, off + j * hw4);
#pragma unroll
        for (int k = 0; k < P; ++k) {
            vr[k][0] = h2_load16(rx, offr[k], off);
            vr[k][1] = h2_load16(rx, offr[k], off + hw4);
        }
#pragma unroll
        for (int k = 0; k < NC2; ++k) {
            vc[k][0] = h2_load4(rx, offc[k], off);
            vc[k][1] = h2_load4(rx, offc[k], off + hw4);
        }
    };
    auto slot_of = [&](int c) { return (c & 3) * S + (c >> 2); };
    auto put_quad = [&](const float4 (&v)[8], bool ok, int dst, int c0, float s) {       // 4 pixels x 8 channels: 8 slots
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = ok ? comp(v[e], j) : 0.f;
            uint4 hi, lo;
            split8(a, s, hi, lo);
            const int sl = dst + slot_of(c0 + j);
            tile[sl] = hi;
            tile[C::PART + sl] = lo;
        }
    };
    auto put_pair = [&](float a0, float a1, float s, int byte_off) {     // one pixel, one channel pair: 4 bytes of hi, 4 of lo
        a0 *= s;
        a1 *= s;
        const h2 hh = {(_Float16)a0, (_Float16)a1};
        const h2 ll = {(_Float16)(a0 - (float)hh[0]), (_Float16)(a1 - (float)hh[1])};
        char *base = reinterpret_cast<char *>(tile) + byte_off;
        *reinterpret_cast<unsigned *>(base) = __builtin_bit_cast(unsigned, hh);
        *reinterpret_cast<unsigned *>(base + C::PART * 16) = __builtin_bit_cast(unsigned, ll);
    };

    // operand addresses.  B = pixels: lane (n = lane & 31, hl) of pixel tile t holds row wave + 4 (n >> 4), column 4 (n & 15) + t;
    // at tap (ky, kx) its slot is row + ky, ((t + kx) & 3) * S + (n & 15) + ((t + kx) >> 2): a compile-time offset from bbase.
    // A = weights: lane (m = lane & 31, hl) holds cout m of a block, channels 8 hl .. 8 hl + 7.
    // (a wave's two rows are FOUR apart - wave, wave + 4: a 16-byte LDS read is free of bank conflicts only if lanes 16-31 sit a
    // multiple of 256 bytes from lanes 0-15 (tools/lds_bank_probe.hip: 3.9 conflict cycles per read at a stride of one 1088-byte
    // tile row, 0 at four rows = 17 x 256 bytes; the offset of the upper half-wave does not matter))
    const int bbase = (hl * IH + wave + 4 * (l31 >> 4)) * RS + (l31 & 15);
    const int abase = hl * NT * 32 + l31;
    // Weight stages (one filter row of one chunk each) go through a ring of LDS buffers, the LDS-DMA AHEAD stages in front of the
    // matrix instructions.  The wait in front of a stage's closing barrier is a COUNTED wait for the pieces of the NEXT stage that
    // leaves what is younger in flight (loads, stores and LDS-DMA return in issue order).
    // A piece = 64 consecutive LDS slots of the stage image [kx][part][channel half][32 NT couts]; in the pack a row holds all
    // 32 NT ncb couts of the layer: the lane offsets below pick this tile's cout block.
    constexpr int PW = C::PW, LOADS = 8 + 2 * P + 2 * NC2;           // vector-memory instructions of issue_weights / fetch
    const int row_slots = ncb * NT * 32;                             // pack slots per (kx, part, channel half) row
    unsigned wvoff[PW], wlds[PW];                                    // lane offset in the pack; LDS byte address in ring slot 0 (scalar)
    const unsigned lds_wl = lds_addr_of(wl);
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        // every wave issues PW instructions, so that the counted vmcnt waits are the same arithmetic for all of them: a wave
        // without a piece of its own repeats an earlier one (same bytes to the same place).  A transfer with EXEC = 0 is not
        // a substitute - it does not count in vmcnt, and the waves issuing it then waited for one transfer too few.
        const int piece = (wave + 4 * p) % (WST / 64);
        const int L = piece * 64 + lane, row = L / (NT * 32), col = L - row * (NT * 32);
        wvoff[p] = 16u * (unsigned)(row * row_slots + col);
        wlds[p] = lds_wl + 16u * (unsigned)(piece * 64);
    }
    int ring = 0;                                                    // ring slot of the stage being multiplied
    auto issue_weights = [&](int stage, int slot, const TileRef &r) {       // LDS-DMA of stage (chunk, ky) = stage / KS, stage % KS
        const uint4 *src = r.w + 1 + (size_t)stage * (KS * 4) * row_slots + r.cb * NT * 32;          // slot 0 of the pack = header
        // (scalar LDS addresses: the pointer form of the transfer cost ~140 cycles of dependent scalar instructions each)
#pragma unroll
        for (int p = 0; p < PW; ++p) lds_dma16_m(src, wvoff[p], wlds[p] + (unsigned)slot * (WST * 16u));
    };
    auto ring_next = [&](int r, int k) { return r + k >= RING ? r + k - RING : r + k; };

#ifdef RISP_H2_STAMPS
    unsigned long long t_start = __builtin_amdgcn_s_memtime(), t_stage = 0, t_mat = 0, t_wait = 0, t_epi = 0, t_steps = 0, t0, t1;
    const unsigned long long rt_start = __builtin_amdgcn_s_memrealtime();
#endif
    int t_cur = wg;
    if (t_cur >= ntiles) return;
    locate(t_cur, cur);
    setup(cur);
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) issue_weights(s, s, cur);
    fetch(0);
    for (;;) {
        f32x16 acc[4][NT];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][b][e] = 0.f;
        int se = 0;                                    // running exponent: the accumulators hold sum * 2^se * s_w
        const int t_next = t_cur + nwg;
        const bool more = t_next < ntiles;
        TileRef nxt = cur;
        if (more) locate(t_next, nxt);
        for (int ch = 0; ch < nchunks; ++ch) {
            H2STAMP(t0);
            // largest magnitude of the chunk's tile -> red[wave]
            float m = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) m = amax4(m, ok1 ? v1[j] : make_float4(0.f, 0.f, 0.f, 0.f));
#pragma unroll
            for (int k = 0; k < P; ++k)
                if (okr[k]) m = amax4(amax4(m, vr[k][0]), vr[k][1]);
#pragma unroll
            for (int k = 0; k < NC2; ++k)
                if (okc[k]) m = fmaxf(m, fmaxf(fabsf(vc[k][0]), fabsf(vc[k][1])));
            m = h2_wave_max(m);
            if (lane == 0) red[wave] = m;
#ifdef RISP_H2_STAMPS
            H2STAMP(t1); t_wait += t1 - t0; t0 = t1;   // waiting for the prefetched tile (vmcnt) + the reduction
#endif
            __syncthreads();                           // A: the maxima are visible; every wave has left the previous chunk's tile
            const float4 mx = *reinterpret_cast<const float4 *>(red);
            const float tmax = fmaxf(fmaxf(mx.x, mx.y), fmaxf(mx.z, mx.w));
            // exponent that puts tmax into [2^14, 2^15): biased exponent eb of tmax -> 2^(141 - eb)
            int eb = (int)(__builtin_bit_cast(unsigned, tmax) >> 23);
            eb = __builtin_amdgcn_readfirstlane(eb);
            int want = 141 - eb;                        // exponent of s_x (unbiased)
            want = want > 100 ? 100 : want;             // an all-zero or denormal tile: any scale will do
            if (ch == 0) {
                se = want;
            } else if (want < se) {                     // larger values than before: rescale the running sums (exact)
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
            const float s = __builtin_bit_cast(float, (unsigned)(127 + se) << 23);
            put_quad(v1, ok1, dst1, 4 * q1 + P, s);
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const int q = (tid + 256 * k) & 15;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    put_pair(okr[k] ? comp(vr[k][0], j) : 0.f, okr[k] ? comp(vr[k][1], j) : 0.f, s, dstr[k] + 16 * slot_of(4 * q + j + P));
            }
#pragma unroll
            for (int k = 0; k < NC2; ++k)
                if (dstc[k] >= 0) put_pair(okc[k] ? vc[k][0] : 0.f, okc[k] ? vc[k][1] : 0.f, s, dstc[k]);
            // (the weight pieces of this chunk's first stage were waited for in front of the previous stage's last barrier, or - first
            // tile - are older than the tile just consumed)
            __syncthreads();                           // B: tile and weights complete
#ifdef RISP_H2_STAMPS
            H2STAMP(t1); t_stage += t1 - t0; t0 = t1;
#endif

            // ---- matrix phase: KS stages (filter rows) x KS taps x 4 pixel tiles x NT cout blocks x 3 products
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int stage = ch * KS + ky;
                const bool dma = stage + AHEAD < nstages || more;
                const bool fetched = ky == 0 && ch + 1 < nchunks;
                auto feed = [&]() {                     // AHEAD stages ahead, into the slot every wave has left; then the tile prefetch
                    if (stage + AHEAD < nstages) issue_weights(stage + AHEAD, ring_next(ring, AHEAD), cur);
                    else if (more) issue_weights(stage + AHEAD - nstages, ring_next(ring, AHEAD), nxt);
                    if (fetched) fetch(ch + 1);         // behind the pieces: in flight during the whole matrix phase
                };
                const uint4 *ws = wl + ring * WST + abase;
                const uint4 *ts = tile + bbase + ky * RS;
#ifdef RISP_H2_STAMPS
                unsigned long long ts0, ts1;
#endif
                if constexpr (KS == 3) {
                    // 12 steps (kx, t): the B operand of step + 1 (and the A operands of the next tap) are read while the six
                    // products of this step run
                    h8 a[2][NT][2], bv[2][2];
                    auto load_a = [&](int kx, int buf) {
#pragma unroll
                        for (int b = 0; b < NT; ++b)
#pragma unroll
                            for (int part = 0; part < 2; ++part)
                                a[buf][b][part] = __builtin_bit_cast(h8, ws[((kx * 2 + part) * 2) * NT * 32 + b * 32]);
                    };
                    auto load_b = [&](int u, int buf) {
                        const int sl = (u & 3) * S + (u >> 2);
                        bv[buf][0] = __builtin_bit_cast(h8, ts[sl]);
                        bv[buf][1] = __builtin_bit_cast(h8, ts[C::PART + sl]);
                    };
                    load_a(0, 0);
                    load_b(0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    feed();                             // behind the first operand reads: issued while those are in flight
                    __builtin_amdgcn_sched_barrier(0);
                    H2STAMP(ts0);
#pragma unroll
                    for (int step = 0; step < 4 * KS; ++step) {
                        const int kx = step >> 2, t = step & 3;
                        if (step + 1 < 4 * KS) {
                            load_b(((step + 1) & 3) + ((step + 1) >> 2), (step + 1) & 1);
                            if (((step + 1) & 3) == 0) load_a((step + 1) >> 2, ((step + 1) >> 2) & 1);
                        }
                        // keep the reads of step + 1 in front of the products of this step (hipcc sinks them to their first use and
                        // waits lgkmcnt(0) in front of every group otherwise) and apart from later reads of the same slots
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        const int ab = kx & 1, bb = step & 1;
#pragma unroll
                        for (int b = 0; b < NT; ++b) acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ab][b][0], bv[bb][1], acc[t][b], 0, 0, 0);
#pragma unroll
                        for (int b = 0; b < NT; ++b) acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ab][b][1], bv[bb][0], acc[t][b], 0, 0, 0);
#pragma unroll
                        for (int b = 0; b < NT; ++b) acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ab][b][0], bv[bb][0], acc[t][b], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    // 5 taps: pixel tile t at tap kx reads the slots of column shift u = t + kx - 8 distinct B operands serve the
                    // 20 (t, kx) pairs of the row (one cout block per wave: read pair by pair the LDS would be the limit).  The A
                    // operands of all taps stay in registers; B runs two shifts ahead.
                    static_assert(KS == 3 || NT == 1, "the 5x5 form holds one cout block per wave");
                    h8 a[KS][2], bv[3][2];
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                        for (int part = 0; part < 2; ++part) a[kx][part] = __builtin_bit_cast(h8, ws[((kx * 2 + part) * 2) * NT * 32]);
                    auto load_b = [&](int u, int buf) {
                        const int sl = (u & 3) * S + (u >> 2);
                        bv[buf][0] = __builtin_bit_cast(h8, ts[sl]);
                        bv[buf][1] = __builtin_bit_cast(h8, ts[C::PART + sl]);
                    };
                    load_b(0, 0);
                    load_b(1, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    feed();
                    __builtin_amdgcn_sched_barrier(0);
                    H2STAMP(ts0);
#pragma unroll
                    for (int u = 0; u < KS + 3; ++u) {
                        if (u + 2 < KS + 3) load_b(u + 2, (u + 2) % 3);
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int kx = u - t;
                            if (kx >= 0 && kx < KS) {
                                acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kx][0], bv[u % 3][1], acc[t][0], 0, 0, 0);
                                acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kx][1], bv[u % 3][0], acc[t][0], 0, 0, 0);
                                acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kx][0], bv[u % 3][0], acc[t][0], 0, 0, 0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#ifdef RISP_H2_STAMPS
                ts1 = __builtin_amdgcn_s_memtime();
                t_steps += ts1 - ts0;
#endif
                ring = ring_next(ring, 1);
                // the NEXT stage's pieces must have landed: issued AHEAD - 1 stages ago (in front of this stage's feed when AHEAD is 1).
                // Younger than them: the pieces of the feeds since, and the tile prefetch of this chunk's stage 0 where it came later.
                if (!dma) {
                    H2_WAIT_VM(0);
                } else if (AHEAD == 2) {
                    if (ky + 1 < KS && ch + 1 < nchunks) H2_WAIT_VM(PW + LOADS);
                    else H2_WAIT_VM(PW);
                } else {
                    if (fetched) H2_WAIT_VM(LOADS);
                    else H2_WAIT_VM(0);
                }
                if (ky + 1 < KS) __syncthreads();
            }
#ifdef RISP_H2_STAMPS
            __builtin_amdgcn_sched_barrier(0);
            H2STAMP(t1); t_mat += t1 - t0;
#endif
        }
        // ---- epilogue: y = epilogue(acc * 2^-se / s_w + bias).  Lane (n, hl): couts 32 b + 8 (e >> 2) + 4 hl + (e & 3) of the
        // tile's cout block(s), the four pixels 4 (n & 15) .. + 3 of row wave + 4 (n >> 4) sit in the four pixel tiles: one 16-byte
        // store per cout, 16 lanes = 256 contiguous bytes of a cout row.
#ifdef RISP_H2_STAMPS
        const unsigned long long t_loop_end = __builtin_amdgcn_s_memtime();
#endif
        {
            const int oy = cur.y0 + wave + 4 * (l31 >> 4), ox = cur.x0 + 4 * (l31 & 15);
            const int g = d.group_n > 0 ? cur.n / d.group_n : 0;
            const int na = (d.group_flags & RISP_GROUP_SHARED_ADD) ? cur.n - g * d.group_n : cur.n;
            const float inv_sw = *reinterpret_cast<const float *>(cur.w);
            const float fin = inv_sw * __builtin_bit_cast(float, (unsigned)(127 - se) << 23);
            const int epi = d.epilogue;
            constexpr bool has_add = HAS_ADD, has_mask = HAS_MASK;
            const float floor_ = (epi & RISP_EPI_RELU) ? 0.f : -__builtin_inff();
            const bool pixok = oy < d.H && ox < d.W;
            // addresses = wave-uniform base (image: scalar registers) + one 32-bit lane offset (pixel, + 4 couts for the upper
            // half-wave) + the cout plane
            unsigned hw4e = hw4;                        // opaque copy: the plane offsets below are tile-invariant, and hipcc would keep
            asm volatile("" : "+s"(hw4e));              // them all in scalar registers across the persistent loop (spilling them)
            const unsigned loff = (pixok ? 4u * (unsigned)(oy * d.W + ox) : 0u) + (unsigned)(cur.cb * 32 * NT + 4 * hl) * hw4;
            const __amdgpu_buffer_rsrc_t ry = h2_rsrc(d.y + (size_t)cur.n * d.cout * hw);
            const __amdgpu_buffer_rsrc_t ra = h2_rsrc(has_add ? d.add + (size_t)na * d.add_c * hw : d.x);
            const __amdgpu_buffer_rsrc_t rm = h2_rsrc(has_mask ? d.mask + (size_t)cur.n * d.cout * hw : d.x);
            const float *bias_row = lbias + parity * 64 + 4 * hl;
            // gfx9 counts loads and stores in one in-order counter: a load issued after a store returns only when that store has
            // completed.  So the residual / mask rows are loaded in few, large batches (64 registers: 16 couts, 8 with both
            // tensors), each batch in front of its own stores - one store round trip per batch instead of one per cout pair.
            // (Forming all results first and storing at the end would need none, but hipcc cannot reuse the accumulator registers
            // element by element and spills the results.)
            constexpr int EB = (HAS_ADD && HAS_MASK) ? 8 : 16, NB = 16 * NT / EB;
            float4 av[HAS_ADD ? EB : 1], mv[HAS_MASK ? EB : 1];
#pragma unroll
            for (int g2 = 0; g2 < NB; ++g2) {
                // batch = couts [g2 * EB, g2 * EB + EB) in units of (b, j, i): cout = 32 b + 8 j + 4 hl + i
#pragma unroll
                for (int k = 0; k < EB; ++k) {
                    const int c = g2 * EB + k, cu = (c >> 4) * 32 + 8 * ((c >> 2) & 3) + (c & 3);
                    if (has_add) av[k] = h2_load16(ra, loff, (unsigned)cu * hw4e);
                    if (has_mask) mv[k] = h2_load16(rm, loff, (unsigned)cu * hw4e);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < EB; ++k) {
                    const int c = g2 * EB + k, b = c >> 4, j = (c >> 2) & 3, i = c & 3, e = 4 * j + i, cu = b * 32 + 8 * j + i;
                    const float bb = bias_row[cu];
                    float4 o = make_float4(acc[0][b][e] * fin + bb, acc[1][b][e] * fin + bb, acc[2][b][e] * fin + bb, acc[3][b][e] * fin + bb);
                    if (has_add) {
                        const float4 a4 = av[k];
                        o.x += a4.x; o.y += a4.y; o.z += a4.z; o.w += a4.w;
                    }
                    o.x = o.x < floor_ ? floor_ : o.x;            // ReLU, or nothing (floor = -inf); a NaN stays a NaN (torch.relu)
                    o.y = o.y < floor_ ? floor_ : o.y;
                    o.z = o.z < floor_ ? floor_ : o.z;
                    o.w = o.w < floor_ ? floor_ : o.w;
                    if (has_mask) {
                        const float4 mk = mv[k];
                        o.x = mk.x > 0.f ? o.x : 0.f;
                        o.y = mk.y > 0.f ? o.y : 0.f;
                        o.z = mk.z > 0.f ? o.z : 0.f;
                        o.w = mk.w > 0.f ? o.w : 0.f;
                    }
                    // (plane offset in the VECTOR offset, scalar offset 0: a 16-byte buffer store reads its data registers late, and
                    // hipcc pads the following overwrite of them with wait states only for this form - with the plane in the scalar
                    // offset the last four lanes of every 16 stored the NEXT cout's values on gfx950)
                    if (pixok) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry, loff + (unsigned)cu * hw4e, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#ifdef RISP_H2_STAMPS
        t_epi += __builtin_amdgcn_s_memtime() - t_loop_end;
#endif
        if (!more) break;
        cur = nxt;
        setup(cur);
        fetch(0);
        t_cur = t_next;
    }
#ifdef RISP_H2_STAMPS
    if (lane == 0 && d.cvals) {                        // diagnostic build: cycle shares of a wave's life (tools/ab_f16x2.py)
        unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<float *>(d.cvals)) + 10 * ((size_t)blockIdx.x * 4 + wave);
        __builtin_amdgcn_s_waitcnt(0x0070);
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        o[0] = t_wait; o[1] = t_stage; o[2] = t_mat; o[3] = t_epi; o[4] = t_end - t_start;
        o[5] = rt_start; o[6] = __builtin_amdgcn_s_memrealtime(); o[8] = t_steps; o[9] = 0;
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[7] = hwid | ((unsigned long long)xcc << 32);
    }
#endif
}

#ifndef RISP_H2_WGS
#define RISP_H2_WGS 2        // persistent workgroups per CU (1: diagnostic builds, tools/ab_f16x2.py)
#endif

template <int KS, int NT, bool HAS_ADD, bool HAS_MASK>
int launch_f16x2(const risp_conv_desc &d, void *stream) {
    using C = H2<KS, NT>;
    auto kern = &conv_f16x2_kernel<KS, NT, HAS_ADD, HAS_MASK>;
    if (C::LDS_BYTES > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess) {
        risp_set_error("risp_conv2d_f16x2: cannot raise the dynamic LDS limit to %d bytes", C::LDS_BYTES);
        return 2;
    }
    const int tx = (d.W + H2_TW - 1) / H2_TW, ty = (d.H + H2_TH - 1) / H2_TH, ncb = d.cout / (32 * NT);
    const long long ntiles = (long long)tx * ty * d.N * ncb;
    if (ntiles > 0x7fffffff) {
        risp_set_error("risp_conv2d_f16x2: too many tiles");
        return 1;
    }
    const int slots = RISP_H2_WGS * h2_cu_count();
    const int grid = ntiles < slots ? (int)ntiles : slots;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), C::LDS_BYTES, (hipStream_t)stream, d, tx, ty, ncb, (int)ntiles);
    RISP_LAUNCH_CHECK("risp_conv2d_f16x2");
    return 0;
}

template <int KS, int NT>
int launch_f16x2_epi(const risp_conv_desc &d, void *stream) {
#ifdef RISP_H2_NOMASK      /* timing experiment only (wrong results): what the fp32 mask reads of the backward-data passes cost */
    const bool a = (d.epilogue & RISP_EPI_ADD) != 0, m = false;
#else
    const bool a = (d.epilogue & RISP_EPI_ADD) != 0, m = (d.epilogue & RISP_EPI_MASK) != 0;
#endif
    return a ? (m ? launch_f16x2<KS, NT, true, true>(d, stream) : launch_f16x2<KS, NT, true, false>(d, stream))
             : (m ? launch_f16x2<KS, NT, false, true>(d, stream) : launch_f16x2<KS, NT, false, false>(d, stream));
}
}  // namespace

int risp_launch_f16x2_ws(const risp_conv_desc &d, void *stream);      // risp_conv_f16x2_ws.hip

extern "C" {

#ifdef RISP_H2_STAMPS
int risp_conv_f16x2_occupancy(void) {                 // diagnostic builds only: resident workgroups per CU
    int nb = -1;
    using C = H2<3, 2>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_f16x2_kernel<3, 2, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_f16x2_kernel<3, 2, false, false>, 256, C::LDS_BYTES);
    return nb;
}
#endif

size_t risp_conv_f16x2_wpack_bytes(int cin, int cout, int ksize) {
    const int nt = (cout + 31) / 32, nch = (cin + H2_CK - 1) / H2_CK;
    return 16 + (size_t)nch * ksize * ksize * 2 * 2 * nt * 32 * 16;
}

static int f16x2_check(const risp_conv_desc *dp) {
    RISP_CHECK_ARG(dp, "risp_conv2d_f16x2: null descriptor");
    const risp_conv_desc &d = *dp;
    RISP_CHECK_ARG(d.x && d.wpack && d.y, "risp_conv2d_f16x2: null tensor");
    RISP_CHECK_GROUP(d, "risp_conv2d_f16x2");
    RISP_CHECK_ARG(d.N > 0 && d.H > 0 && d.W > 0 && d.W % 4 == 0 && d.cin > 0 && d.cin % H2_CK == 0 && (d.cout == 32 || d.cout == 64) &&
                       (d.ksize == 3 || d.ksize == 5) && (unsigned long long)(d.cout > d.cin ? d.cout : d.cin) * d.H * d.W * 4ull < (1ull << 31) &&
                       (!(d.epilogue & RISP_EPI_ADD) || d.add_c == d.cout),
                   "risp_conv2d_f16x2: needs a 3x3 or 5x5 layer, cin %% 16 == 0, cout 32 or 64 (= add_c), W %% 4 == 0, fewer than 2^31 "
                   "bytes per image on either side (N=%d H=%d W=%d cin=%d cout=%d k=%d)",
                   d.N, d.H, d.W, d.cin, d.cout, d.ksize);
    RISP_CHECK_ARG(d.load_mode == RISP_LOAD_PLAIN, "risp_conv2d_f16x2: only plain loads");
    RISP_CHECK_ARG(!(d.epilogue & ~(RISP_EPI_RELU | RISP_EPI_ADD | RISP_EPI_MASK | RISP_EPI_NOBIAS)),
                   "risp_conv2d_f16x2: epilogue %d not supported", d.epilogue);
    RISP_CHECK_ARG((d.epilogue & RISP_EPI_NOBIAS) || d.bias, "risp_conv2d_f16x2: bias missing");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_ADD) || (d.add && d.add_c > 0), "risp_conv2d_f16x2: add tensor missing");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_MASK) || d.mask, "risp_conv2d_f16x2: mask tensor missing");
    RISP_CHECK_ARG(((reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.y) | reinterpret_cast<uintptr_t>(d.add) |
                     reinterpret_cast<uintptr_t>(d.mask) | reinterpret_cast<uintptr_t>(d.wpack)) & 15) == 0,
                   "risp_conv2d_f16x2: tensors must be 16-byte aligned");
    return 0;
}

/* the form in which every wave stages, waits and multiplies (round 4): two 4-wave workgroups per CU */
int risp_conv2d_f16x2_uniform(const risp_conv_desc *dp, void *stream) {
    const int rc = f16x2_check(dp);
    if (rc) return rc;
    const risp_conv_desc &d = *dp;
    if (d.ksize == 5) return launch_f16x2_epi<5, 1>(d, stream);         // one cout block per tile: 64 couts = two tiles per pixel tile
    return d.cout == 64 ? launch_f16x2_epi<3, 2>(d, stream) : launch_f16x2_epi<3, 1>(d, stream);
}

int risp_conv2d_f16x2(const risp_conv_desc *dp, void *stream) {
    const int rc = f16x2_check(dp);
    if (rc) return rc;
    // the wave-specialised form (same bits).  Its 64-cout 3x3 kernel spreads a tile's epilogue over the last chunk and the next tile's
    // first one: a one-chunk layer (16 input channels: none in the proxies) takes the uniform form
    if (dp->ksize == 3 && dp->cout == 64 && dp->cin < 32) return risp_conv2d_f16x2_uniform(dp, stream);
    return risp_launch_f16x2_ws(*dp, stream);
}

}  // extern "C"
