// Direct (vector-ALU) convolution for layers with at most 4 output channels, and the rectangle sums that replace
// a convolution over spatially constant channels.
//
// Why: the fp32 matrix-core kernel (risp_conv.hip) tiles the output channels in blocks of 32, so the 3-channel
// heads and tails of the proxies issue 8-10x the useful work there:
//   * SRCNNRes conv 5x5 32->3            (srcnn_res_arch.py:22)              forward
//   * SRCNNRes conv 9x9 (12+P)->64       (srcnn_res_arch.py:18)              backward-data, image channels only
// On CDNA4 the packed fp32 vector FMA has the same peak as the fp32 MFMA (157 TFLOP/s), so a direct kernel with
// 4 output channels per pixel loses nothing to padding: weights are wave-uniform (scalar loads feed
// v_pk_fma_f32 straight from SGPR pairs), activations come from an LDS halo tile as 3 x 128-bit reads per filter
// row and are reused across the k taps of the row and the 4 pixels of the thread.
//
// The 9+P broadcast channels of SRCNNRes (srcnn_res_arch.py:41-46) are per-image constants inside the image and
// zero in the padding.  Forward, their contribution is a per-(image, cout, border case) constant (risp_conv2d
// RISP_EPI_CASEBIAS); backward, the gradient of constant c is sum_{co,tap} W[co][c][tap] * S[co][tap] with S the
// sum of the upstream gradient over the pixels whose tap lands inside the image - risp_rect_sums below.
#include "risp_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int SX = 64, SPX = 4;              // tile width; SPX x SPY pixels per thread, tile height 16 * SPY
constexpr int HPAD = 4;                      // halo columns kept left and right of the tile (>= k/2, multiple of 4)
constexpr int STW = SX + 2 * HPAD;           // LDS row stride: tile column t <-> image x0 - HPAD + t

template <int KS> struct SmallCfg { static constexpr int CCH = KS == 9 ? 3 : 4; };   // input channels per LDS stage

// Thread = 4 x 2 output pixels x 4 couts (16 packed accumulators).  Per input channel the thread walks the
// KS + 1 tile rows its two output rows need: each row is 3 x 128-bit LDS reads, used by the taps of BOTH output
// rows, and each filter row (KS x 4 wave-uniform weights, scalar loads) serves both output rows too.
// NP = pairs of output channels held per pixel: 2 (cout <= 4) or 6 (cout <= 12: SRCNNDemosaic's 5x5 32 -> 12 tail,
// srcnn_demosaic_arch.py:21, which the matrix-core kernel pads to 32).  The weight pack is [cin][k][k][2 NP].
// SPY = output rows per thread: 2 (64 x 32 tiles) when the grid fills the chip, 1 (64 x 16 tiles, twice the
// workgroups) for small batches - the per-GPU batch of the 8-GPU search is 4 images.
// Occupancy: the 3x3 and 9x9 forms with 4 couts are held to 128 registers = FOUR workgroups per CU (a few spilled
// values outside the tap loops): the weights arrive by scalar loads whose latency three waves per SIMD do not cover, and
// the 1024-tile grid of a grouped launch at the per-GPU batch of the 8-GPU search is then ONE round of workgroups
// instead of 768 + 256 (9x9 64 -> 3 on 32 x 256 x 256: 1050 -> 873 us).  The 5x5 form loses (44 bytes of scratch in
// the loop: 215 -> 277 us) and keeps its 165 registers.
// C3 (cout == 3, two output rows per thread): the fourth lane of the packed FMAs is not left idle.  The layer's pack then
// has k + 1 filter rows and its fourth slot holds the THIRD cout's weight of the row above - (w0, w1, w2[ky], w2[ky - 1]),
// row k = (0, 0, 0, w2[k - 1]) - so that at tile row r ONE packed FMA per pixel and tap serves cout 2 of both output rows
// (lane x: row 0 with w2[r], lane y: row 1 with w2[r - 1]; the zero slots add 0) beside the two (c0, c1) FMAs of the
// rows: 12 instead of 16 per pixel quad and tap, the same products in the same order per output.  The other forms read
// such a pack too (their "cout 3" collects w2[ky - 1] products and is never stored).
#ifndef RISP_SMALL_C3_5X5_WAVES
#define RISP_SMALL_C3_5X5_WAVES 4
#endif
template <int KS, int NP, int SPY, bool C3 = false>
__global__ __launch_bounds__(256, (NP == 2 && (KS != 5 || (C3 && RISP_SMALL_C3_5X5_WAVES == 4)) ? 4 : (NP == 2 ? 3 : 2))) void conv_small_kernel(const risp_conv_desc d_in, int groups, float *__restrict__ partial) {
    static_assert(!C3 || (NP == 2 && SPY == 2), "C3: four-lane form of a 3-cout layer on two output rows");
    constexpr int SY = 16 * SPY;
    constexpr int P = KS / 2, TH_ = SY + 2 * P, CCH = SmallCfg<KS>::CCH, ROWV = STW / 4;
    extern __shared__ float4 lds4[];                       // [CCH][TH_][STW]
    // groups > 1 (small grids): the input channels are split over `groups` workgroups per tile, each writes its raw
    // partial sums to `partial` [group][N][cout][H][W] and small_reduce_kernel finishes (fixed order: deterministic)
    const int tid = threadIdx.x, n = blockIdx.z / groups, grp = blockIdx.z - n * groups;
    const risp_conv_desc d = risp_conv_group_view(d_in, n);
    const int x0 = blockIdx.x * SX, y0 = blockIdx.y * SY;
    const int cpg = (d.cin + groups - 1) / groups, cbeg = grp * cpg, cend = (cbeg + cpg < d.cin) ? cbeg + cpg : d.cin;
    const int lx = (tid & 15) * SPX, ly = (tid >> 4) * SPY;
    const int H = d.H, W = d.W, cin = d.cin;
    const size_t plane = (size_t)H * W;
    const float *__restrict__ xin = d.x + (size_t)n * cin * plane;
    const f32x2 *__restrict__ wp = reinterpret_cast<const f32x2 *>(d.wpack);        // [cin][KS][KS][NP] cout pairs
    const bool vecw = (W & 3) == 0;

    const int wrows = (NP == 2 && d.cout == 3) ? KS + 1 : KS;                      // filter rows in the pack (see C3)
    // C3: acc[r][p][0] = (c0, c1) of output row r; acc[0][p][1] = c2 of (row 0, row 1); acc[1][p][1] unused
    f32x2 acc[SPY][SPX][NP];
#pragma unroll
    for (int r = 0; r < SPY; ++r)
#pragma unroll
        for (int p = 0; p < SPX; ++p)
#pragma unroll
            for (int q = 0; q < NP; ++q) acc[r][p][q] = (f32x2){0.f, 0.f};

    // Staging: all global loads of a stage are issued (into registers) before the first LDS write, so a stage costs
    // one memory round trip, which the other resident workgroups of the CU cover with their FMAs.
    constexpr int NIT = (CCH * TH_ * ROWV + 255) / 256;
    float4 pf[NIT];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + 256 * it;
            const int c = idx / (TH_ * ROWV), rem = idx - c * (TH_ * ROWV);
            const int ty = rem / ROWV, v = rem - ty * ROWV;
            const int gy = y0 - P + ty, gx = x0 - HPAD + 4 * v, ci = c0 + c;
            const bool row_ok = idx < CCH * TH_ * ROWV && ci < cend && gy >= 0 && gy < H;
            const float *src = xin + (size_t)(row_ok ? ci : 0) * plane + (size_t)(row_ok ? gy : 0) * W;
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (vecw) {                                    // W % 4 == 0: the float4 is entirely in or out
                if (row_ok && gx >= 0 && gx < W) q = *reinterpret_cast<const float4 *>(src + gx);
            } else {
                float *e = reinterpret_cast<float *>(&q);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (row_ok && gx + j >= 0 && gx + j < W) e[j] = src[gx + j];
            }
            pf[it] = q;
        }
    };
    // SPY == 1 (small grids, few waves per CU, half the accumulators): the loads of stage s+1 are issued before the
    // FMAs of stage s, so their latency does not sit between barriers.
    constexpr bool PIPE = SPY == 1;
    if (PIPE) fetch(cbeg);
    for (int c0 = cbeg; c0 < cend; c0 += CCH) {
        if (!PIPE) fetch(c0);
        __syncthreads();                                   // the previous stage has been consumed
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + 256 * it;
            if (idx < CCH * TH_ * ROWV) lds4[idx] = pf[it];
        }
        __syncthreads();
        if (PIPE && c0 + CCH < cend) fetch(c0 + CCH);
        const int cn = cend - c0 < CCH ? cend - c0 : CCH;
        for (int c = 0; c < cn; ++c) {
            const f32x2 *wch = wp + (size_t)(c0 + c) * wrows * KS * NP;            // wave-uniform -> scalar loads
#pragma unroll
            for (int r = 0; r < KS + SPY - 1; ++r) {                               // tile row ly + r
                const float4 *row = lds4 + ((c * TH_ + ly + r) * STW + lx) / 4;
                const float4 r0 = row[0], r1 = row[1], r2 = row[2];
                const float a[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
                if constexpr (C3) {
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        const f32x2 w2p = wch[(r * KS + kx) * 2 + 1];              // (w2[r], w2[r - 1]); rows 0 and KS carry the zeros
                        f32x2 wa = {0.f, 0.f}, wb = {0.f, 0.f};
                        if (r < KS) wa = wch[(r * KS + kx) * 2];                   // (c0, c1) of filter row r     -> output row 0
                        if (r >= 1) wb = wch[((r - 1) * KS + kx) * 2];             // (c0, c1) of filter row r - 1 -> output row 1
#pragma unroll
                        for (int p = 0; p < SPX; ++p) {
                            const float av = a[HPAD - P + kx + p];
                            const f32x2 a2 = {av, av};
                            if (r < KS) acc[0][p][0] = __builtin_elementwise_fma(a2, wa, acc[0][p][0]);
                            if (r >= 1) acc[1][p][0] = __builtin_elementwise_fma(a2, wb, acc[1][p][0]);
                            acc[0][p][1] = __builtin_elementwise_fma(a2, w2p, acc[0][p][1]);
                        }
                    }
                    continue;
                }
#pragma unroll
                for (int o = 0; o < SPY; ++o) {                                    // output row o sees it as filter row r - o
                    const int ky = r - o;
                    if (ky < 0 || ky >= KS) continue;
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        f32x2 wq[NP];
#pragma unroll
                        for (int q = 0; q < NP; ++q) wq[q] = wch[(ky * KS + kx) * NP + q];
#pragma unroll
                        for (int p = 0; p < SPX; ++p) {
                            const float av = a[HPAD - P + kx + p];
                            const f32x2 a2 = {av, av};
#pragma unroll
                            for (int q = 0; q < NP; ++q) acc[o][p][q] = __builtin_elementwise_fma(a2, wq[q], acc[o][p][q]);
                        }
                    }
                }
            }
        }
    }

    // ---- epilogue: bias, residual add, ReLU, ReLU mask; one 16-byte store per output plane and row
    const int ox = x0 + lx, epi = d.epilogue;
    if (ox >= W) return;
    const bool full = vecw || ox + SPX <= W;
    if (groups > 1) {                           // raw partial sums of this channel group
#pragma unroll
        for (int o = 0; o < SPY; ++o) {
            const int oy = y0 + ly + o;
            if (oy < H) {
#pragma unroll
                for (int co = 0; co < 2 * NP; ++co) {
                    if (co < d.cout) {
                        float *pp = partial + (((size_t)grp * d.N + n) * d.cout + co) * plane + (size_t)oy * W + ox;
#pragma unroll
                        for (int p = 0; p < SPX; ++p)
                            if (full || ox + p < W) pp[p] = (C3 && co == 2) ? (o ? acc[0][p][1].y : acc[0][p][1].x) : (co & 1 ? acc[o][p][co >> 1].y : acc[o][p][co >> 1].x);
                    }
                }
            }
        }
        return;
    }
    if (epi & RISP_EPI_SHUFFLE2) {          // PixelShuffle(2): (N,4G,H,W) -> (N,G,2H,2W), cout pair 2g+i = row i of group g
        const int groups = d.cout >> 2;
#pragma unroll
        for (int o = 0; o < SPY; ++o) {
            const int oy = y0 + ly + o;
            if (oy >= H) break;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int g = q >> 1, i = q & 1;
                if (g < groups) {
                const float b0 = (epi & RISP_EPI_NOBIAS) ? 0.f : d.bias[2 * q], b1 = (epi & RISP_EPI_NOBIAS) ? 0.f : d.bias[2 * q + 1];
                float *yp = d.y + (((size_t)n * groups + g) * 2 * H + 2 * oy + i) * (2 * (size_t)W) + 2 * ox;
                float e[2 * SPX];
#pragma unroll
                for (int p = 0; p < SPX; ++p) {
                    e[2 * p] = acc[o][p][q].x + b0;
                    e[2 * p + 1] = acc[o][p][q].y + b1;
                }
                if (vecw) {
                    *reinterpret_cast<float4 *>(yp) = make_float4(e[0], e[1], e[2], e[3]);
                    *reinterpret_cast<float4 *>(yp + 4) = make_float4(e[4], e[5], e[6], e[7]);
                } else {
#pragma unroll
                    for (int p = 0; p < SPX; ++p)
                        if (ox + p < W) { yp[2 * p] = e[2 * p]; yp[2 * p + 1] = e[2 * p + 1]; }
                }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int o = 0; o < SPY; ++o) {
        const int oy = y0 + ly + o;
        if (oy >= H) break;
        const size_t pix = (size_t)oy * W + ox;
#pragma unroll
        for (int co = 0; co < 2 * NP; ++co) {
            if (co < d.cout) {
            const float b = (epi & RISP_EPI_NOBIAS) ? 0.f : d.bias[co];
            float v[SPX];
#pragma unroll
            for (int p = 0; p < SPX; ++p) v[p] = ((C3 && co == 2) ? (o ? acc[0][p][1].y : acc[0][p][1].x) : (co & 1 ? acc[o][p][co >> 1].y : acc[o][p][co >> 1].x)) + b;
            if ((epi & RISP_EPI_ADD) && co < d.add_c) {
                const float *ap = d.add + ((size_t)n * d.add_c + co) * plane + pix;
#pragma unroll
                for (int p = 0; p < SPX; ++p)
                    if (full || ox + p < W) v[p] += ap[p];
            }
            if (epi & RISP_EPI_RELU) {
#pragma unroll
                for (int p = 0; p < SPX; ++p) v[p] = v[p] > 0.f ? v[p] : 0.f;
            }
            if (epi & RISP_EPI_MASK) {
                const float *mp = d.mask + ((size_t)n * d.cout + co) * plane + pix;
#pragma unroll
                for (int p = 0; p < SPX; ++p)
                    if (full || ox + p < W) v[p] = mp[p] > 0.f ? v[p] : 0.f;
            }
            float *yp = d.y + ((size_t)n * d.cout + co) * plane + pix;
            if (vecw) {
                *reinterpret_cast<float4 *>(yp) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int p = 0; p < SPX; ++p)
                    if (ox + p < W) yp[p] = v[p];
            }
            }
        }
    }
}

// S[plane][ky][kx] = sum of g over the pixels q of the plane with q + (ky - P, kx - P) inside the image.
// One workgroup per plane.  Pass 1 streams the plane once with 16-byte loads into per-column sums (no cross-lane
// traffic); the K row ranges differ from "all rows" only by up to P rows at the top or bottom, the K column ranges
// only by up to P columns at the left or right, so everything else happens on W-long vectors in LDS.
constexpr int RK_MAX = 9, RW_MAX = 8192;
__global__ __launch_bounds__(256) void rect_sums_kernel(const float *__restrict__ g, float *__restrict__ out, int H, int W,
                                                        int K) {
    const int P = K / 2;
    const float *gp = g + (size_t)blockIdx.x * H * W;
    extern __shared__ float col[];                         // [phases][W] partial column sums; row 0 becomes the total
    __shared__ float sums[RK_MAX][RK_MAX];
    const int tid = threadIdx.x;
    if ((W & 3) == 0) {
        const int wq = W >> 2;                             // float4 per row
        // thread -> (column group, row phase): consecutive threads read consecutive 16-byte pieces of a row
        const int groups = wq < 256 ? wq : 256, phases = 256 / groups;
        const int cg = tid % groups, ph = tid / groups;
        if (ph < phases) {
            for (int q = cg; q < wq; q += groups) {
                // four independent accumulator chains: four 16-byte loads of the column are in flight together (one
                // chain made this pass latency-bound: 233 us for 64 planes of 256 x 256 per image at batch 32)
                float4 a[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                int y = ph;
                for (; y + 3 * phases < H; y += 4 * phases) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float4 v = *reinterpret_cast<const float4 *>(gp + (size_t)(y + u * phases) * W + 4 * q);
                        a[u].x += v.x; a[u].y += v.y; a[u].z += v.z; a[u].w += v.w;
                    }
                }
                for (; y < H; y += phases) {
                    const float4 v = *reinterpret_cast<const float4 *>(gp + (size_t)y * W + 4 * q);
                    a[0].x += v.x; a[0].y += v.y; a[0].z += v.z; a[0].w += v.w;
                }
                float4 t;
                t.x = (a[0].x + a[1].x) + (a[2].x + a[3].x); t.y = (a[0].y + a[1].y) + (a[2].y + a[3].y);
                t.z = (a[0].z + a[1].z) + (a[2].z + a[3].z); t.w = (a[0].w + a[1].w) + (a[2].w + a[3].w);
                *reinterpret_cast<float4 *>(col + (size_t)ph * W + 4 * q) = t;
            }
        }
        __syncthreads();
        for (int x = tid; x < W; x += 256) {               // fixed order: the result is deterministic
            float a = col[x];
            for (int p2 = 1; p2 < phases; ++p2) a += col[(size_t)p2 * W + x];
            col[x] = a;
        }
    } else {
        for (int x = tid; x < W; x += 256) {
            float a = 0.f;
            for (int y = 0; y < H; ++y) a += gp[(size_t)y * W + x];
            col[x] = a;
        }
    }
    __syncthreads();
    // row case i (dy = i - P): dy < 0 drops rows 0 .. -dy-1, dy > 0 drops the last dy rows.  One wave per case.
    const int lane = tid & 63, wave = tid >> 6;
    for (int i = wave; i < K; i += 4) {
        const int dy = i - P;
        float t = 0.f, edge[2 * (RK_MAX / 2)];             // edge[k-1]: sum of the first k columns, edge[P+k-1]: last k
#pragma unroll
        for (int k = 0; k < 2 * (RK_MAX / 2); ++k) edge[k] = 0.f;
        for (int x = lane; x < W; x += 64) {
            float v = col[x];
            for (int k = 0; k < -dy; ++k) v -= gp[(size_t)k * W + x];
            for (int k = 0; k < dy; ++k) v -= gp[(size_t)(H - 1 - k) * W + x];
            t += v;
            for (int k = 1; k <= P; ++k) {
                if (x < k) edge[k - 1] += v;
                if (x >= W - k) edge[P + k - 1] += v;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            t += __shfl_xor(t, o);
#pragma unroll
            for (int k = 0; k < 2 * (RK_MAX / 2); ++k) edge[k] += __shfl_xor(edge[k], o);
        }
        if (lane == 0) {
            sums[i][P] = t;
            for (int k = 1; k <= P; ++k) {
                sums[i][P - k] = t - edge[k - 1];          // dx = -k: the first k columns have no source
                sums[i][P + k] = t - edge[P + k - 1];      // dx = +k: the last k columns
            }
        }
    }
    __syncthreads();
    if (tid < K * K) out[(size_t)blockIdx.x * K * K + tid] = sums[tid / K][tid % K];
}

// The same rectangle sums of a 9x9 layer from per-tile sums of the planes (risp_conv2d_toep_sums wrote them while it staged the
// planes for the backward-data convolution) plus the 4 border rows and 4 border columns on each side - an eighth of the plane
// instead of all of it:  S(dy, dx) = T - (dropped rows) - (dropped columns) + (dropped rows x dropped columns), where dy < 0 drops
// the first -dy rows, dy > 0 the last dy, and the same for columns.  One workgroup per plane; every sum in a fixed order.
__global__ __launch_bounds__(256) void rect_sums_tiles_kernel(const float *__restrict__ g, const float *__restrict__ psum,
                                                              float *__restrict__ out, int C, int H, int W, int tiles) {
    constexpr int P = 4, K = 9;
    const int plane = blockIdx.x, n = plane / C, c = plane - n * C;
    const float *gp = g + (size_t)plane * H * W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ float part[4][9], tot[9], rowsum[8], corner[4][P][P];
    float t = 0.f, cl[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) cl[k] = 0.f;
    for (int i = tid; i < tiles; i += 256) t += psum[((size_t)n * tiles + i) * C + c];
    for (int y = tid; y < H; y += 256) {                  // the 4 first and 4 last columns of every row
        const float4 a = *reinterpret_cast<const float4 *>(gp + (size_t)y * W), b = *reinterpret_cast<const float4 *>(gp + (size_t)y * W + W - 4);
        cl[0] += a.x; cl[1] += a.y; cl[2] += a.z; cl[3] += a.w;
        cl[4] += b.x; cl[5] += b.y; cl[6] += b.z; cl[7] += b.w;
    }
    {                                                       // the 4 first and 4 last rows: 32 threads each
        const int r = tid >> 5, y = r < P ? r : H - 2 * P + r;
        float rs = 0.f;
        for (int q = tid & 31; q < W / 4; q += 32) {
            const float4 v = *reinterpret_cast<const float4 *>(gp + (size_t)y * W + 4 * q);
            rs += (v.x + v.y) + (v.z + v.w);
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) rs += __shfl_xor(rs, o);
        if ((tid & 31) == 0) rowsum[r] = rs;
    }
    if (tid < 64) {                                         // the four 4 x 4 corners, element by element
        const int q = tid >> 4, r = (tid >> 2) & 3, cc = tid & 3;
        corner[q][r][cc] = gp[(size_t)((q & 2) ? H - P + r : r) * W + ((q & 1) ? W - P + cc : cc)];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        t += __shfl_xor(t, o);
#pragma unroll
        for (int k = 0; k < 8; ++k) cl[k] += __shfl_xor(cl[k], o);
    }
    if (lane == 0) {
        part[wave][0] = t;
#pragma unroll
        for (int k = 0; k < 8; ++k) part[wave][1 + k] = cl[k];
    }
    __syncthreads();
    if (tid < 9) tot[tid] = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    __syncthreads();
    if (tid < K * K) {
        const int i = tid / K, j = tid - i * K, dy = i - P, dx = j - P;
        const int nr = dy < 0 ? -dy : dy, nc = dx < 0 ? -dx : dx;           // dropped rows / columns
        const int r0 = dy < 0 ? 0 : 2 * P - nr, c0 = dx < 0 ? 0 : 2 * P - nc;        // first dropped entry of rowsum[] / tot[1 ..]
        float rd = 0.f, cd = 0.f, xd = 0.f;
        for (int k = 0; k < nr; ++k) rd += rowsum[r0 + k];
        for (int k = 0; k < nc; ++k) cd += tot[1 + c0 + k];
        const int q = (dy > 0 ? 2 : 0) + (dx > 0 ? 1 : 0);
        for (int a = 0; a < nr; ++a)
            for (int b = 0; b < nc; ++b) xd += corner[q][(dy < 0 ? 0 : P - nr) + a][(dx < 0 ? 0 : P - nc) + b];
        out[(size_t)plane * (K * K) + tid] = ((tot[0] - rd) - cd) + xd;
    }
}

// gconst[n][c] = sum_j rs[n][j] * wconst[j][c]: the rectangle sums applied to the constant-plane weights (the backward
// of the folded constants, C = 9 + P columns).  One workgroup per (image, column); every thread owns a fixed residue
// class of j, the partial sums meet by wave shuffles and one LDS pass in a fixed order (deterministic).
__global__ __launch_bounds__(256) void const_grad_kernel(const float *__restrict__ rs, const float *__restrict__ wconst,
                                                         float *__restrict__ gconst, int M, int C) {
    const int c = blockIdx.x, n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float a0 = 0.f, a1 = 0.f;
    int j = tid;
    for (; j + 256 < M; j += 512) {                      // two independent chains: the loads of both are in flight together
        a0 = __builtin_fmaf(rs[(size_t)n * M + j], wconst[(size_t)j * C + c], a0);
        a1 = __builtin_fmaf(rs[(size_t)n * M + j + 256], wconst[(size_t)(j + 256) * C + c], a1);
    }
    if (j < M) a0 = __builtin_fmaf(rs[(size_t)n * M + j], wconst[(size_t)j * C + c], a0);
    float v = a0 + a1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __shared__ float part[4];
    if (lane == 0) part[wave] = v;
    __syncthreads();
    if (tid == 0) gconst[(size_t)n * C + c] = (part[0] + part[1]) + (part[2] + part[3]);
}

// y = epilogue(sum over channel groups of the partial sums), the groups added in index order
__global__ __launch_bounds__(256) void small_reduce_kernel(const risp_conv_desc d_in, int groups, const float *__restrict__ partial) {
    const size_t plane = (size_t)d_in.H * d_in.W, total = (size_t)d_in.N * d_in.cout * plane;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int co = (int)((i / plane) % d_in.cout);
    const size_t n = i / (plane * d_in.cout), pix = i % plane;
    const risp_conv_desc d = risp_conv_group_view(d_in, (int)n);
    float v = partial[i];
    for (int g = 1; g < groups; ++g) v += partial[(size_t)g * total + i];
    const int epi = d.epilogue;
    if (!(epi & RISP_EPI_NOBIAS)) v += d.bias[co];
    if (epi & RISP_EPI_SHUFFLE2) {          // PixelShuffle(2): cout 4g + 2i + j -> plane g, pixel (2y + i, 2x + j)
        const size_t oy = pix / d.W, ox = pix - oy * d.W;
        d.y[((n * (d.cout >> 2) + (co >> 2)) * 2 * d.H + 2 * oy + ((co >> 1) & 1)) * (2 * (size_t)d.W) + 2 * ox + (co & 1)] = v;
        return;
    }
    if ((epi & RISP_EPI_ADD) && co < d.add_c) v += d.add[(n * d.add_c + co) * plane + pix];
    if (epi & RISP_EPI_RELU) v = v > 0.f ? v : 0.f;
    if (epi & RISP_EPI_MASK) v = d.mask[i] > 0.f ? v : 0.f;
    d.y[i] = v;
}

#ifndef RISP_SMALL_C3
#define RISP_SMALL_C3 1       // A/B switch (tools/ab_build.sh): 0 = 3-cout layers on the general form (same pack)
#endif
template <int KS, int NP>
int launch_small(const risp_conv_desc &d, float *scratch, int groups, hipStream_t s) {
    constexpr int P = KS / 2;
    const int tiles_x = (d.W + SX - 1) / SX;
    if (groups <= 1 && (size_t)tiles_x * ((d.H + 31) / 32) * d.N >= 384) {          // enough 64 x 32 tiles for every CU
        const size_t lds = sizeof(float) * SmallCfg<KS>::CCH * (32 + 2 * P) * STW;
        if (RISP_SMALL_C3 && NP == 2 && d.cout == 3)
            hipLaunchKernelGGL((conv_small_kernel<KS, NP, 2, NP == 2>), dim3(tiles_x, (d.H + 31) / 32, d.N), dim3(256), lds, s, d, 1, nullptr);
        else
            hipLaunchKernelGGL((conv_small_kernel<KS, NP, 2>), dim3(tiles_x, (d.H + 31) / 32, d.N), dim3(256), lds, s, d, 1, nullptr);
    } else if (groups > 1 && (size_t)tiles_x * ((d.H + 31) / 32) * d.N * groups >= 256) {
        // channel groups on the 64 x 32 tile (two output rows per thread: each LDS row read feeds both rows' taps):
        // 9x9 64 -> 3 on 4 x 256 x 256, 4 groups: 127 -> 118 us, 8 groups: 136 -> 114 us (tools/ab_small.py)
        const size_t lds = sizeof(float) * SmallCfg<KS>::CCH * (32 + 2 * P) * STW;
        if (RISP_SMALL_C3 && NP == 2 && d.cout == 3)
            hipLaunchKernelGGL((conv_small_kernel<KS, NP, 2, NP == 2>), dim3(tiles_x, (d.H + 31) / 32, d.N * groups), dim3(256), lds, s, d, groups,
                               scratch);
        else
            hipLaunchKernelGGL((conv_small_kernel<KS, NP, 2>), dim3(tiles_x, (d.H + 31) / 32, d.N * groups), dim3(256), lds, s, d, groups,
                               scratch);
        const size_t total = (size_t)d.N * d.cout * d.H * d.W;
        hipLaunchKernelGGL(small_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d, groups, scratch);
    } else {
        if (groups < 1) groups = 1;
        const size_t lds = sizeof(float) * SmallCfg<KS>::CCH * (16 + 2 * P) * STW;
        hipLaunchKernelGGL((conv_small_kernel<KS, NP, 1>), dim3(tiles_x, (d.H + 15) / 16, d.N * groups), dim3(256), lds, s, d, groups,
                           scratch);
        if (groups > 1) {
            const size_t total = (size_t)d.N * d.cout * d.H * d.W;
            hipLaunchKernelGGL(small_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d, groups, scratch);
        }
    }
    RISP_LAUNCH_CHECK("risp_conv2d_small");
    return 0;
}

}  // namespace

extern "C" {

int risp_conv_small_cout_pad(int cout) { return cout <= 4 ? 4 : 12; }

size_t risp_conv_small_wpack_floats(int cin, int cout, int ksize) {
    return (size_t)cin * (ksize + (cout == 3 ? 1 : 0)) * ksize * risp_conv_small_cout_pad(cout);     // cout == 3: k + 1 filter rows (see C3)
}

// Channel groups worth using for this layer on this grid (1 = none): small grids with many input channels, where one
// workgroup per tile leaves most of the chip idle and a lone wave per SIMD cannot hide its scalar-load latency.
int risp_conv_small_groups(const risp_conv_desc *dp) {
    if (!dp || dp->cin < 32) return 1;
    const size_t tiles = (size_t)((dp->W + SX - 1) / SX) * ((dp->H + 15) / 16) * dp->N;
    if (tiles >= 768) return 1;
    const size_t tiles32 = (size_t)((dp->W + SX - 1) / SX) * ((dp->H + 31) / 32) * dp->N;   // the split runs on 64 x 32 tiles
    int g = (int)(1024 / (tiles32 ? tiles32 : 1));
    if (g > 8) g = 8;
    while (g > 1 && dp->cin / g < 8) --g;
    return g < 1 ? 1 : g;
}

static int conv2d_small_impl(const risp_conv_desc *dp, float *scratch, int groups, void *stream);

int risp_conv2d_small(const risp_conv_desc *dp, void *stream) { return conv2d_small_impl(dp, nullptr, 1, stream); }

/* scratch: groups * N * cout * H * W floats, groups = risp_conv_small_groups(d) (or fewer); groups <= 1 ignores it */
int risp_conv2d_small_split(const risp_conv_desc *dp, float *scratch, int groups, void *stream) {
    return conv2d_small_impl(dp, scratch, groups, stream);
}

static int conv2d_small_impl(const risp_conv_desc *dp, float *scratch, int groups, void *stream) {
    RISP_CHECK_ARG(dp, "risp_conv2d_small: null descriptor");
    RISP_CHECK_ARG(groups <= 1 || (scratch && groups <= 16 && (size_t)dp->N * groups <= 65535),
                   "risp_conv2d_small_split: %d groups need a scratch buffer and N * groups <= 65535", groups);
    const risp_conv_desc &d = *dp;
    RISP_CHECK_ARG(d.x && d.wpack && d.y, "risp_conv2d_small: null tensor");
    RISP_CHECK_GROUP(d, "risp_conv2d_small");
    RISP_CHECK_ARG(d.N > 0 && d.N <= 65535 && d.H > 0 && d.W > 0 && d.cin > 0 && d.cout > 0 && d.cout <= 12,
                   "risp_conv2d_small: bad shape N=%d H=%d W=%d cin=%d cout=%d (cout <= 12)", d.N, d.H, d.W, d.cin, d.cout);
    RISP_CHECK_ARG(d.load_mode == RISP_LOAD_PLAIN, "risp_conv2d_small: only plain loads");
    RISP_CHECK_ARG(!(d.epilogue & ~(RISP_EPI_RELU | RISP_EPI_ADD | RISP_EPI_MASK | RISP_EPI_NOBIAS | RISP_EPI_SHUFFLE2)),
                   "risp_conv2d_small: epilogue %d not supported", d.epilogue);
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_SHUFFLE2) ||
                       (d.cout % 4 == 0 && !(d.epilogue & (RISP_EPI_RELU | RISP_EPI_ADD | RISP_EPI_MASK))),
                   "risp_conv2d_small: PixelShuffle store needs cout %% 4 == 0 and no other epilogue");
    RISP_CHECK_ARG((d.epilogue & RISP_EPI_NOBIAS) || d.bias, "risp_conv2d_small: bias missing");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_ADD) || (d.add && d.add_c > 0), "risp_conv2d_small: add tensor missing");
    RISP_CHECK_ARG(!(d.epilogue & RISP_EPI_MASK) || d.mask, "risp_conv2d_small: mask tensor missing");
    RISP_CHECK_ARG((reinterpret_cast<uintptr_t>(d.wpack) & 15) == 0, "risp_conv2d_small: wpack must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (d.cout <= 4) {
        if (d.ksize == 3) return launch_small<3, 2>(d, scratch, groups, s);
        if (d.ksize == 5) return launch_small<5, 2>(d, scratch, groups, s);
        if (d.ksize == 9) return launch_small<9, 2>(d, scratch, groups, s);
    } else {
        if (d.ksize == 3) return launch_small<3, 6>(d, scratch, groups, s);
        if (d.ksize == 5) return launch_small<5, 6>(d, scratch, groups, s);
    }
    risp_set_error("risp_conv2d_small: unsupported kernel size %d", d.ksize);
    return 1;
}

int risp_rect_sums(const float *g, float *out, int planes, int H, int W, int ksize, void *stream) {
    RISP_CHECK_ARG(g && out && planes > 0 && H > 0 && W > 0, "risp_rect_sums: bad arguments");
    RISP_CHECK_ARG((ksize & 1) && ksize >= 1 && ksize <= RK_MAX && H >= ksize / 2 && W >= ksize / 2 && W <= RW_MAX,
                   "risp_rect_sums: window %d on a %dx%d plane (W <= %d)", ksize, H, W, RW_MAX);
    const int wq = W >> 2, groups = wq < 256 ? (wq > 0 ? wq : 1) : 256, phases = (W & 3) ? 1 : 256 / groups;
    const size_t lds = sizeof(float) * (size_t)phases * W;
    hipLaunchKernelGGL(rect_sums_kernel, dim3(planes), dim3(256), lds, (hipStream_t)stream, g, out, H, W, ksize);
    RISP_LAUNCH_CHECK("risp_rect_sums");
    return 0;
}

int risp_rect_sums_tiles(const float *g, const float *psum, float *out, int N, int C, int H, int W, int tiles, void *stream) {
    RISP_CHECK_ARG(g && psum && out && N > 0 && C > 0 && tiles > 0 && H >= 4 && W >= 4 && W % 4 == 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0,
                   "risp_rect_sums_tiles: bad arguments (9x9 windows: H, W >= 4, W %% 4 == 0, 16-byte aligned planes)");
    hipLaunchKernelGGL(rect_sums_tiles_kernel, dim3((unsigned)((size_t)N * C)), dim3(256), 0, (hipStream_t)stream, g, psum, out, C, H, W, tiles);
    RISP_LAUNCH_CHECK("risp_rect_sums_tiles");
    return 0;
}

int risp_srcnn_const_grad(const float *rs, const float *wconst, float *gconst, int N, int M, int C, void *stream) {
    RISP_CHECK_ARG(rs && wconst && gconst && N > 0 && N <= 65535 && M > 0 && C >= 1, "risp_srcnn_const_grad: bad arguments (N=%d M=%d C=%d)",
                   N, M, C);
    hipLaunchKernelGGL(const_grad_kernel, dim3(C, N), dim3(256), 0, (hipStream_t)stream, rs, wconst, gconst, M, C);
    RISP_LAUNCH_CHECK("risp_srcnn_const_grad");
    return 0;
}

}  // extern "C"
