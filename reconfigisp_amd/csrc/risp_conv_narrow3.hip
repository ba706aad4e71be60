// risp_conv2d_narrow3: 3x3 layers with at most 4 OUTPUT channels and 16 .. 64 input channels in split precision on the f16 matrix pipe -
// Path-Restore's last layer (path_14l_bgr_arch.py / path_14l_bayer_arch.py: 64 -> 3, 64 -> 4 + PixelShuffle) and the backward-data pass
// of its first layer (the same shape).  The vector-FMA kernel risp_conv2d_small stages 4 channels at a time between two barriers and
// reads its 256 B / pixel at 2.6 TB/s; the arithmetic of these layers is nothing, so the kernel here is built around the loads.
//
// Rows of a 32 x 32 x 16 matrix instruction = (filter row ky, cout): 4 ky + co, 12 of 32; reduction index = 16 input channels; the filter
// column kx is a shift of the pixel operand by whole 16-byte slots; columns = 32 pixels of a wave's strip.  A wave walks down its segment
// of rows: per INPUT row it loads the row's channels (lane = (column, channel half): 32 columns of which the first and the last are the
// neighbours' - 30 output columns per wave, no halo pass; 4-byte loads of consecutive pixels, requested one row ahead), scales (one
// scale per wave and row), splits and writes them chunk by chunk as its own LDS operand row (no barrier: a wave's LDS operations
// complete in order), 3 columns x cin / 16 chunks x 3 products give the row's contribution to THREE output
// rows, which meet in registers: out[y] = (D[y - 1][ky 0] + D[y][ky 1]) + D[y + 1][ky 2] (the ky 1 part crosses from the upper lane half
// by one shuffle).  Every sum in a fixed order; a result depends neither on the batch nor on how the launch cuts its segments.
#include "risp_common.h"
#include "risp_f16x2.h"

namespace {
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#ifndef N3_WAVES
#define N3_WAVES 4                 /* waves of a workgroup (8 and 16 measured: 266 / 211 us against 218 at 32 x 256 x 256 - no DRAM-page effect) */
#endif
constexpr int N3_WO = 30, N3_TW = N3_WAVES * N3_WO, N3_MAXCH = 4;            // output columns of a wave (its 32 loaded columns minus one each side), of a workgroup; chunks of 16 channels

template <bool SHUF>
__global__ __launch_bounds__(64 * N3_WAVES) void conv_narrow3_kernel(const risp_conv_desc d_in, int strips, int segs, int seg_rows) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];      // weights [chunk][kx][part][64 lanes], then per wave the operand row of ONE chunk [part][hl][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hl = lane >> 5;
    int t = blockIdx.x;
    const int sg = t % segs;
    t /= segs;
    const int st = t % strips, n = t / strips;
    const risp_conv_desc d = risp_conv_group_view(d_in, n);
    const int H = d.H, W = d.W, nch = d.cin >> 4;
    const size_t hw = (size_t)H * W;
    const unsigned hw4 = (unsigned)hw * 4u;
    const int x0 = st * N3_TW + N3_WO * wave, ys = sg * seg_rows, ye = ys + seg_rows < H ? ys + seg_rows : H;
    const uint4 *wp = reinterpret_cast<const uint4 *>(d.wpack);
    uint4 *wl = smem, *eb = smem + nch * 3 * 2 * 64 + wave * (2 * 2 * 32);
    for (int s = tid; s < nch * 3 * 2 * 64; s += 64 * N3_WAVES) wl[s] = wp[1 + s];
    __syncthreads();
    if (x0 >= W) return;                                              // (a strip's waves beyond the image's last column; no barrier below)
    const float inv_sw = *reinterpret_cast<const float *>(wp);
#ifdef N3_WREG                                               /* (experiment: the weights in registers instead of 24 LDS reads per row - 196 registers, two
                                                                waves per SIMD: 290 us against 213 at 32 x 256 x 256; the kernel lives on its occupancy) */
    h8 wr[N3_MAXCH][3][2];
#pragma unroll
    for (int c = 0; c < N3_MAXCH; ++c)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int part = 0; part < 2; ++part) wr[c][kx][part] = __builtin_bit_cast(h8, wl[((c < nch ? c : 0) * 3 + kx) * 2 * 64 + part * 64 + lane]);
#endif

    // ---- input: lane (l31, hl) owns column x0 - 1 + l31 and channels 8 hl .. 8 hl + 7 of every chunk: all 64 lanes load and convert; the
    // wave's first and last column are its neighbours' (30 output columns per wave).  A column or row outside the image reads zeros.
    const __amdgpu_buffer_rsrc_t rx = h2_rsrc(d.x + (size_t)n * d.cin * hw);
    const int cx = x0 - 1 + l31;
    const unsigned voff = (cx >= 0 && cx < W) ? 4u * (unsigned)cx + (unsigned)(8 * hl) * hw4 : 0x80000000u;
    float v[N3_MAXCH][8];
    auto request = [&](int row) {                                     // (the row is the wave's: a uniform branch, the offset a scalar)
        if (row >= 0 && row < H) {
            const unsigned ro = 4u * (unsigned)(row * W);
#pragma unroll
            for (int c = 0; c < N3_MAXCH; ++c)
                if (c < nch) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[c][k] = h2_load4(rx, voff, ro + (unsigned)(16 * c + k) * hw4);
                }
        } else {
#pragma unroll
            for (int c = 0; c < N3_MAXCH; ++c)
#pragma unroll
                for (int k = 0; k < 8; ++k) v[c][k] = 0.f;
        }
    };
    // ---- output: lanes 1 .. 30 of the LOWER half store; couts e & 3 of filter rows 0 (e < 4) and 2 (e >= 4), the upper half holds row 1
    const int ox = x0 - 1 + l31;
    const bool ook = hl == 0 && l31 >= 1 && l31 <= N3_WO && ox < W;
    const __amdgpu_buffer_rsrc_t ry = h2_rsrc(SHUF ? d.y + (size_t)n * (d.cout >> 2) * 4 * hw : d.y + (size_t)n * d.cout * hw);
    const bool relu = (d.epilogue & RISP_EPI_RELU) != 0;
    float bias[4];
#pragma unroll
    for (int co = 0; co < 4; ++co) bias[co] = (!(d.epilogue & RISP_EPI_NOBIAS) && co < d.cout) ? d.bias[co] : 0.f;
    float P[4], Q[4];                                                 // out[r - 1] so far: D[r - 2][ky 0] + D[r - 1][ky 1];  out[r] so far: D[r - 1][ky 0]
#pragma unroll
    for (int co = 0; co < 4; ++co) P[co] = Q[co] = 0.f;
    const int bcol[3] = {(l31 + 31) & 31, l31, (l31 + 1) & 31};       // operand column of filter column kx (the wrapped ones feed the two unused outputs)

    request(ys - 1);
    for (int r = ys - 1; r <= ye; ++r) {
        // ---- one scale for the wave's 32 columns x cin channels of input row r
        float m = 0.f;
#pragma unroll
        for (int c = 0; c < N3_MAXCH; ++c)
            if (c < nch) {
#pragma unroll
                for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf(v[c][k]));
            }
        m = h2_wave_max(m);
        int se = 141 - (int)(__builtin_bit_cast(unsigned, m) >> 23);                    // the largest magnitude into [2^14, 2^15)
        se = __builtin_amdgcn_readfirstlane(se);
        se = se > 100 ? 100 : se;                                     // an all-zero or denormal row: any scale will do
        const float sc = __builtin_bit_cast(float, (unsigned)(127 + se) << 23);
        const float fin = inv_sw * __builtin_bit_cast(float, (unsigned)(127 - se) << 23);
        // ---- chunk by chunk: split, the wave's own LDS operand row (no barrier: a wave's LDS operations complete in order, the row of
        // chunk c + 1 is written behind the reads of chunk c), 3 filter columns x 3 products into two alternating accumulators
        f32x16 acc[2];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][e] = acc[1][e] = 0.f;
#pragma unroll
        for (int c = 0; c < N3_MAXCH; ++c)
            if (c < nch) {
                uint4 hi, lo;
                split8(v[c], sc, hi, lo);
                eb[hl * 32 + l31] = hi;
                eb[64 + hl * 32 + l31] = lo;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const h8 bh = __builtin_bit_cast(h8, eb[hl * 32 + bcol[kx]]), bl = __builtin_bit_cast(h8, eb[64 + hl * 32 + bcol[kx]]);
#ifdef N3_WREG
                    const h8 ah = wr[c][kx][0], al = wr[c][kx][1];
#else
                    const h8 ah = __builtin_bit_cast(h8, wl[((c * 3 + kx) * 2 + 0) * 64 + lane]), al = __builtin_bit_cast(h8, wl[((c * 3 + kx) * 2 + 1) * 64 + lane]);
#endif
                    f32x16 &a = acc[(c * 3 + kx) & 1];
                    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, a, 0, 0, 0);
                }
            }
        request(r + 1);                                               // (the values are in LDS: their registers take the next row)
        // ---- the three output rows this input row reaches
        const int y = r - 1;
        float val[4];
#pragma unroll
        for (int co = 0; co < 4; ++co) {
            const float d0 = (acc[0][co] + acc[1][co]) * fin, d2 = (acc[0][4 + co] + acc[1][4 + co]) * fin;      // lower half: filter rows 0 and 2; upper half (in d0): row 1
            const float d1 = __shfl_xor(d0, 32);                      // (the lower half receives row 1)
            float o = (P[co] + d2) + bias[co];                        // out[r - 1] complete
            P[co] = Q[co] + d1;
            Q[co] = d0;
            val[co] = relu ? (o < 0.f ? 0.f : o) : o;
        }
        if (y >= ys && y < ye) {
            if (SHUF) {                                               // PixelShuffle(2): cout 2 i + j -> pixel (2 y + i, 2 x + j) of a (2H, 2W) plane
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned off = ook ? 4u * (unsigned)((2 * y + i) * (2 * W) + 2 * ox) : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, val[2 * i]), __builtin_bit_cast(unsigned, val[2 * i + 1])}, ry, off, 0u, 0);
                }
            } else {
                const unsigned off = ook ? 4u * (unsigned)(y * W + ox) : 0x80000000u;
#pragma unroll
                for (int co = 0; co < 4; ++co)
                    if (co < d.cout) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val[co]), ry, off, (unsigned)co * hw4, 0);
            }
        }
    }
}
}  // namespace

extern "C" {

size_t risp_conv_narrow3_wpack_bytes(int cin) { return 16 + (size_t)(cin / 16) * 3 * 2 * 64 * 16; }

int risp_conv2d_narrow3(const risp_conv_desc *dp, void *stream) {
    RISP_CHECK_ARG(dp, "risp_conv2d_narrow3: null descriptor");
    const risp_conv_desc &d = *dp;
    RISP_CHECK_ARG(d.x && d.wpack && d.y, "risp_conv2d_narrow3: null tensor");
    RISP_CHECK_GROUP(d, "risp_conv2d_narrow3");
    RISP_CHECK_ARG(d.N > 0 && d.H > 0 && d.W > 0 && d.ksize == 3 && d.cin >= 16 && d.cin <= 16 * N3_MAXCH && d.cin % 16 == 0 && d.cout >= 1 && d.cout <= 4,
                   "risp_conv2d_narrow3: a 3x3 layer with 16 .. 64 input channels (a multiple of 16) and 1 .. 4 output channels (ksize=%d cin=%d cout=%d)",
                   d.ksize, d.cin, d.cout);
    RISP_CHECK_ARG(d.load_mode == RISP_LOAD_PLAIN && !(d.epilogue & ~(RISP_EPI_RELU | RISP_EPI_SHUFFLE2 | RISP_EPI_NOBIAS)),
                   "risp_conv2d_narrow3: plain loads; epilogue RELU | SHUFFLE2 | NOBIAS");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_SHUFFLE2) || d.cout == 4, "risp_conv2d_narrow3: PixelShuffle(2) needs 4 output channels");
    RISP_CHECK_ARG((d.epilogue & RISP_EPI_NOBIAS) || d.bias, "risp_conv2d_narrow3: bias missing");
    RISP_CHECK_ARG((unsigned long long)d.cin * d.H * d.W * 4 < (1ull << 31), "risp_conv2d_narrow3: an image side of 2 GiB or more (buffer addressing)");
    RISP_CHECK_ARG((reinterpret_cast<uintptr_t>(d.wpack) & 15) == 0, "risp_conv2d_narrow3: the weight pack must be 16-byte aligned");
    const int strips = (d.W + N3_TW - 1) / N3_TW;
    // rows of a segment: 32, shorter while the launch has fewer workgroups than twice the CUs (a result does not depend on the cut)
    int seg = 32;
#ifndef N3_MIN_WGS
#define N3_MIN_WGS 512
#endif
    while (seg > 4 && (long long)d.N * strips * ((d.H + seg - 1) / seg) < N3_MIN_WGS) seg >>= 1;
    const int segs = (d.H + seg - 1) / seg;
    const long long items = (long long)d.N * strips * segs;
    RISP_CHECK_ARG(items <= 0x7fffffff, "risp_conv2d_narrow3: too many work items");
    const size_t lds = ((size_t)(d.cin / 16) * 3 * 2 * 64 + N3_WAVES * (2 * 2 * 32)) * 16;
    if (d.epilogue & RISP_EPI_SHUFFLE2)
        hipLaunchKernelGGL(conv_narrow3_kernel<true>, dim3((unsigned)items), dim3(64 * N3_WAVES), lds, (hipStream_t)stream, d, strips, segs, seg);
    else
        hipLaunchKernelGGL(conv_narrow3_kernel<false>, dim3((unsigned)items), dim3(64 * N3_WAVES), lds, (hipStream_t)stream, d, strips, segs, seg);
    RISP_LAUNCH_CHECK("risp_conv2d_narrow3");
    return 0;
}

}  // extern "C"
