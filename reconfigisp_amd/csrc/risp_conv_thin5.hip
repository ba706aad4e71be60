// risp_conv2d_thin5: 5x5 layers with at most 3 INPUT channels and 32 / 64 output channels in split precision on the f16 matrix pipe -
// the backward-data pass of SRCNNRes' last layer (srcnn_res_arch.py:22: the upstream gradient's 3 image channels -> the 32 hidden
// channels, masked by the ReLU of the layer before, :20).  Until round 6 it ran on the fp32 Winograd kernel F(4,5), which pads the 3
// channels to 4 and walks chunks of channels it does not have: 1.12 ms per grouped launch of 8 x 32 x 256 x 256, for a layer that moves
// 4.5 GB (the 32-channel mask in, the 32-channel result out) - an HBM-bound launch.
//
// Reduction index of ONE 32 x 32 x 16 matrix instruction = (filter row ky, channel c): 15 of 16 slots; the filter column kx is a shift of
// the pixel operand by whole 16-byte slots.  Per output row a wave builds, for each of its 32 + 4 columns, the column's 5 x 3 input
// values (a register ring that walks down the image: 3 new values per row) as hi / lo halves in its own LDS row, then 5 columns x 3
// split-precision products give 32 pixels x 32 couts.  The product leaves a lane ONE pixel of 16 couts: a store instruction writes
// two whole 128-byte lines (32 consecutive pixels of two cout rows) - no transposition.  One scale per work item (image, 128-column
// strip, 32-row segment, block of 32 couts): a sum has one chunk, so no running exponent.  Fixed order of every sum: bit-repeatable;
// a result does not depend on the batch an image travels in.
#include "risp_common.h"
#include "risp_f16x2.h"

namespace {
constexpr int T5_TW = 128, T5_SEG = 32, T5_P = 2, T5_KS = 5;
constexpr int T5_EC = 40;                                            // slots of a wave's operand row (36 columns used)

template <bool HAS_MASK, bool HAS_BIAS>
__global__ __launch_bounds__(256) void conv_thin5_kernel(const risp_conv_desc d_in, int strips, int segs, int nb) {
    __shared__ __attribute__((aligned(16))) uint4 ebuf[4][2 * 2 * T5_EC];      // per wave: [part: hi, lo][half of the reduction index][column]
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hl = lane >> 5;
    int t = blockIdx.x;
    const int cb = t % nb;
    t /= nb;
    const int sg = t % segs;
    t /= segs;
    const int st = t % strips, n = t / strips;
    const risp_conv_desc d = risp_conv_group_view(d_in, n);
    const int H = d.H, W = d.W;
    const size_t hw = (size_t)H * W;
    const unsigned hw4 = (unsigned)hw * 4u;
    const int x0 = st * T5_TW, ys = sg * T5_SEG, ye = ys + T5_SEG < H ? ys + T5_SEG : H;
    const float *xin = d.x + (size_t)n * d.cin * hw;

    // ---- the item's scale: largest magnitude of its input region (rows ys - 2 .. ye + 1, columns x0 - 2 .. x0 + 129)
    float m = 0.f;
    {
        const int r0 = ys - T5_P < 0 ? 0 : ys - T5_P, r1 = ye + T5_P < H ? ye + T5_P : H;
        const int c0 = x0 - T5_P < 0 ? 0 : x0 - T5_P, c1 = x0 + T5_TW + T5_P < W ? x0 + T5_TW + T5_P : W;
        for (int q = wave; q < d.cin * (r1 - r0); q += 4) {
            const int c = q / (r1 - r0), r = r0 + q - c * (r1 - r0);
            const float *row = xin + (size_t)c * hw + (size_t)r * W;
            for (int xx = c0 + lane; xx < c1; xx += 64) m = fmaxf(m, fabsf(row[xx]));
        }
    }
    m = h2_wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    const float tmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (x0 + 32 * wave >= W) return;                                  // (a strip's waves beyond the image's last column; no barrier below)
    int se = 141 - (int)(__builtin_bit_cast(unsigned, tmax) >> 23);                     // the largest magnitude into [2^14, 2^15)
    se = __builtin_amdgcn_readfirstlane(se);
    se = se > 100 ? 100 : se;                                         // an all-zero or denormal region: any scale will do
    const float sc = __builtin_bit_cast(float, (unsigned)(127 + se) << 23);
    const uint4 *wp = reinterpret_cast<const uint4 *>(d.wpack);
    const float fin = *reinterpret_cast<const float *>(wp) * __builtin_bit_cast(float, (unsigned)(127 - se) << 23);

    // ---- the cout block's weights: per filter column the hi and lo halves of [cout l31][(ky, c) 8 hl ..]
    h8 wa[T5_KS][2];
#pragma unroll
    for (int kx = 0; kx < T5_KS; ++kx)
#pragma unroll
        for (int part = 0; part < 2; ++part) wa[kx][part] = __builtin_bit_cast(h8, wp[1 + (((cb * T5_KS + kx) * 2 + part) * 2 + hl) * 32 + l31]);

    // ---- input: lane L < 36 owns column x0 + 32 wave - 2 + L of the wave's operand row; a column or row outside the image reads zeros
    const __amdgpu_buffer_rsrc_t rx = h2_rsrc(xin);
    const int cx = x0 + 32 * wave - T5_P + lane;
    const bool cok = lane < 32 + 2 * T5_P && cx >= 0 && cx < W;
    unsigned voff[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) voff[c] = (cok && c < d.cin) ? 4u * (unsigned)cx + (unsigned)c * hw4 : 0x80000000u;
    auto load_row = [&](int row, float (&v)[3]) {                     // (the row is the wave's: a uniform branch, the offset a scalar)
        if (row >= 0 && row < H) {
            const unsigned ro = 4u * (unsigned)(row * W);
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = h2_load4(rx, voff[c], ro);
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = 0.f;
        }
    };
    float ring[T5_KS][3], nxt[3];                                     // rows y - 2 .. y + 2 of the lane's column; row y + 3 on its way
#pragma unroll
    for (int r = 0; r < T5_KS; ++r) load_row(ys - T5_P + r, ring[r]);
    load_row(ys + T5_P + 1, nxt);

    // ---- output: lane (pixel l31, half hl) holds couts 32 cb + 8 (e >> 2) + 4 hl + (e & 3), e = 0 .. 15
    const int ox = x0 + 32 * wave + l31;
    const unsigned vo = ox < W ? 4u * (unsigned)ox + (unsigned)(4 * hl) * hw4 : 0x80000000u;
    const __amdgpu_buffer_rsrc_t ry = h2_rsrc(d.y + (size_t)n * d.cout * hw);
    const __amdgpu_buffer_rsrc_t rm = h2_rsrc(HAS_MASK ? d.mask + (size_t)n * d.cout * hw : d.y);
    const bool relu = (d.epilogue & RISP_EPI_RELU) != 0;
    float bias[HAS_BIAS ? 16 : 1];                                    // (a template parameter: 16 registers the backward-data launch does not hold - 4 waves per SIMD)
    if (HAS_BIAS) {
#pragma unroll
        for (int e = 0; e < 16; ++e) bias[e] = d.bias[32 * cb + 8 * (e >> 2) + 4 * hl + (e & 3)];
    }
    uint4 *eb = ebuf[wave];
    // the mask row of the NEXT output row is requested before this row's stores: memory operations retire in order, a wait for loads
    // issued behind stores would wait for the stores too
    float mk[16];
    auto load_mask = [&](int y, float (&v)[16]) {
        const unsigned so_ = 4u * (unsigned)(y * W) + (unsigned)(32 * cb) * hw4;
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = h2_load4(rm, vo, so_ + (unsigned)(8 * (e >> 2) + (e & 3)) * hw4);
    };
    if (HAS_MASK) load_mask(ys, mk);

    for (int y = ys; y < ye; ++y) {
        // ---- the operand row of output row y: reduction index k = 3 ky + c, slot hl = k >> 3
        {
            const float a0[8] = {ring[0][0], ring[0][1], ring[0][2], ring[1][0], ring[1][1], ring[1][2], ring[2][0], ring[2][1]};
            const float a1[8] = {ring[2][2], ring[3][0], ring[3][1], ring[3][2], ring[4][0], ring[4][1], ring[4][2], 0.f};
            uint4 h0, l0, h1, l1;
            split8(a0, sc, h0, l0);
            split8(a1, sc, h1, l1);
            if (lane < T5_EC) {
                eb[0 * T5_EC + lane] = h0;
                eb[1 * T5_EC + lane] = h1;
                eb[2 * T5_EC + lane] = l0;
                eb[3 * T5_EC + lane] = l1;
            }
        }
        const unsigned so = 4u * (unsigned)(y * W) + (unsigned)(32 * cb) * hw4;
        float mkn[16];
        if (HAS_MASK && y + 1 < ye) load_mask(y + 1, mkn);
        // ---- 5 filter columns x 3 products (a wave's LDS operations complete in order: the reads see the row just written)
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int kx = 0; kx < T5_KS; ++kx) {
            const h8 bh = __builtin_bit_cast(h8, eb[(0 + hl) * T5_EC + l31 + kx]), bl = __builtin_bit_cast(h8, eb[(2 + hl) * T5_EC + l31 + kx]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[kx][1], bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[kx][0], bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[kx][0], bh, acc, 0, 0, 0);
        }
        // ---- the ring walks one row down; the row after next is requested
#pragma unroll
        for (int r = 0; r + 1 < T5_KS; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) ring[r][c] = ring[r + 1][c];
#pragma unroll
        for (int c = 0; c < 3; ++c) ring[T5_KS - 1][c] = nxt[c];
        load_row(y + T5_P + 2, nxt);
        // ---- epilogue: 16 stores of two 128-byte lines each
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float v = acc[e] * fin;
            if (HAS_BIAS) v += bias[e];
            if (HAS_MASK) v = mk[e] > 0.f ? v : 0.f;
            if (relu) v = v < 0.f ? 0.f : v;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, vo, so + (unsigned)(8 * (e >> 2) + (e & 3)) * hw4, 0);
        }
        if (HAS_MASK) {
#pragma unroll
            for (int e = 0; e < 16; ++e) mk[e] = mkn[e];
        }
    }
}
}  // namespace

extern "C" {

size_t risp_conv_thin5_wpack_bytes(int cout) { return 16 + (size_t)((cout + 31) / 32) * T5_KS * 2 * 2 * 32 * 16; }

int risp_conv2d_thin5(const risp_conv_desc *dp, void *stream) {
    RISP_CHECK_ARG(dp, "risp_conv2d_thin5: null descriptor");
    const risp_conv_desc &d = *dp;
    RISP_CHECK_ARG(d.x && d.wpack && d.y, "risp_conv2d_thin5: null tensor");
    RISP_CHECK_GROUP(d, "risp_conv2d_thin5");
    RISP_CHECK_ARG(d.N > 0 && d.H > 0 && d.W > 0 && d.ksize == 5 && d.cin >= 1 && d.cin <= 3 && d.cout >= 32 && d.cout <= 64 && d.cout % 32 == 0,
                   "risp_conv2d_thin5: a 5x5 layer with 1 .. 3 input and 32 or 64 output channels (ksize=%d cin=%d cout=%d)", d.ksize, d.cin, d.cout);
    RISP_CHECK_ARG(d.load_mode == RISP_LOAD_PLAIN && !(d.epilogue & ~(RISP_EPI_RELU | RISP_EPI_MASK | RISP_EPI_NOBIAS)),
                   "risp_conv2d_thin5: plain loads; epilogue RELU | MASK | NOBIAS");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_MASK) || d.mask, "risp_conv2d_thin5: mask tensor missing");
    RISP_CHECK_ARG((d.epilogue & RISP_EPI_NOBIAS) || d.bias, "risp_conv2d_thin5: bias missing");
    RISP_CHECK_ARG((unsigned long long)d.cout * d.H * d.W * 4 < (1ull << 31), "risp_conv2d_thin5: an image side of 2 GiB or more (buffer addressing)");
    RISP_CHECK_ARG((reinterpret_cast<uintptr_t>(d.wpack) & 15) == 0, "risp_conv2d_thin5: the weight pack must be 16-byte aligned");
    const int strips = (d.W + T5_TW - 1) / T5_TW, segs = (d.H + T5_SEG - 1) / T5_SEG, nb = d.cout / 32;
    const long long items = (long long)d.N * strips * segs * nb;
    RISP_CHECK_ARG(items <= 0x7fffffff, "risp_conv2d_thin5: too many work items");
    const bool mask = (d.epilogue & RISP_EPI_MASK) != 0, bias = !(d.epilogue & RISP_EPI_NOBIAS);
    auto kern = mask ? (bias ? &conv_thin5_kernel<true, true> : &conv_thin5_kernel<true, false>) : (bias ? &conv_thin5_kernel<false, true> : &conv_thin5_kernel<false, false>);
    hipLaunchKernelGGL(kern, dim3((unsigned)items), dim3(256), 0, (hipStream_t)stream, d, strips, segs, nb);
    RISP_LAUNCH_CHECK("risp_conv2d_thin5");
    return 0;
}

}  // extern "C"
