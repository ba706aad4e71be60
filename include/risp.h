/*
 * risp.h - C ABI of libreconfigisp_hip.so, the MI355X (gfx950) implementation of
 * ReconfigISP's per-image ISP forward path.
 *
 * What this boundary replaces (paths relative to the reference's codes/):
 *   - the absent /DATA/ISP_Kernels plugin that models/modules/tools_origin.py:8-17
 *     imports and calls as  Kernel().run(img, option, params_dict)  (14 call sites,
 *     SURVEY.md section 8b);
 *   - the torch-eager bodies of the in-tree operators (WbQuadratic, GtmManual, the
 *     SRCNN / Path-Restore proxies, the mixed-op combiner, whole2patch/patch2whole).
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch types.  Every pointer is DEVICE memory
 *     owned by the caller (PyTorch); the library keeps no reference past the call and
 *     allocates nothing; scratch space is passed in explicitly.
 *   - tensors are contiguous NCHW fp32; colour images are BGR (ch0=B, ch1=G, ch2=R),
 *     Bayer mosaics are 1-channel RGGB; H and W are even.
 *   - `stream` is a hipStream_t (NULL = default stream); launches are asynchronous.
 *   - return value 0 = OK; non-zero = error, text via risp_last_error() (thread local).
 *   - re-entrant: no mutable global state.
 */
#ifndef RISP_H
#define RISP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RISP_VERSION 100

int risp_version(void);
const char *risp_last_error(void);

/* ---------------------------------------------------------------------------
 * Element-wise operators.  `p` is the (N,P) per-image parameter block in [0,1]
 * (what the reference passes as sigmoid(par).repeat(N,1)); gp is (N,P) and is
 * fully written by the call.
 * ------------------------------------------------------------------------- */

/* demosaic.Demosaic().run(img,'nearestneighbor',..) - tools_origin.py:278-284.
 * (N,1,H,W) RGGB -> (N,3,H,W) BGR.  OPSPEC: R,B replicated over the 2x2 quad,
 * G takes the green sample of its own row. */
int risp_demosaic_nearest_fwd(const float *bayer, float *bgr, int N, int H, int W, void *stream);
int risp_demosaic_nearest_bwd(const float *g_bgr, float *g_bayer, int N, int H, int W, void *stream);

/* whitebalance.WhiteBalance().run(img,'manual',{'gain'}) - tools_origin.py:211-221.  p = the gain
 * itself, (N,3) in [0,5] (the wrapper's params * 5, :214); y_c = x_c * gain_c, no clipping. */
int risp_wb_manual_fwd(const float *x, const float *p, float *y, int N, int HW, void *stream);
/* All *_bwd of the element-wise ops: gp is fully written (no pre-zeroing); scratch holds
 * risp_param_grad_scratch_floats(N) floats - one partial row per workgroup, added in index order by a second
 * launch, so parameter gradients are bit-repeatable. */
size_t risp_param_grad_scratch_floats(int N);
int risp_wb_manual_bwd(const float *x, const float *p, const float *gy, float *gx, float *gp,
        float *scratch, int N, int HW, void *stream);

/* gamma.Gamma().run(img,'manual',{'gamma':g}) - tools_origin.py:59-69. P=1.
 * OPSPEC: y = x^g (x >= 1/1024), y = x * (1/1024)^(g-1) below. */
int risp_gamma_fwd(const float *x, const float *p, float *y, int N, int HW, void *stream);
int risp_gamma_bwd(const float *x, const float *p, const float *gy, float *gx, float *gp,
        float *scratch, int N, int HW, void *stream);

/* GtmManual(4).forward - tools_origin.py:414-440. P=3; knots come from p[0,:] only,
 * so gp[0,:] holds the whole-batch gradient and gp[1:,:] = 0. */
int risp_gtm_manual_fwd(const float *x, const float *p, float *y, int N, int HW, void *stream);
int risp_gtm_manual_bwd(const float *x, const float *p, const float *gy, float *gx, float *gp,
        float *scratch, int N, int HW, void *stream);

/* WbQuadratic.forward - tools_origin.py:317-359. P=30, coef[n,ch,j] = 10 p[n,10ch+j] - 5 */
int risp_wb_quadratic_fwd(const float *x, const float *p, float *y, int N, int HW, void *stream);
int risp_wb_quadratic_bwd(const float *x, const float *p, const float *gy, float *gx, float *gp,
        float *scratch, int N, int HW, void *stream);

/* Per-image per-channel statistics of an (N,C,H,W) tensor (NC = N*C planes):
 * stats[plane*4+{0,1,2}] = {min, sum, max}; arg[plane*2+{0,1}] = first (row-major) index of
 * the min / max (may be NULL).  Used by SRCNNRes (srcnn_res_arch.py:36-40) and gray-world.
 * scratch: risp_channel_stats_scratch_floats(NC,HW) floats.  Deterministic (no atomics). */
size_t risp_channel_stats_scratch_floats(int NC, int HW);
int risp_channel_stats(const float *x, float *stats, int32_t *arg, float *scratch, int NC, int HW,
                       void *stream);
/* backward of the statistics, in place on gx (NC planes): gx += g_mean[plane]/HW everywhere,
 * gx[argmin] += g_min[plane], gx[argmax] += g_max[plane]; any of the three may be NULL. */
int risp_stats_bwd(float *gx, const float *g_min, const float *g_mean, const float *g_max,
                   const int32_t *arg, int NC, int HW, void *stream);
/* the same with the three gradients as column ranges of a row-major (N, row_stride) matrix - plane n * C + c reads
 * g_*[n * row_stride + c] - so that SRCNNRes' folded backward (srcnn_res_arch.py:36-46) passes slices of its
 * (N, 9+P) constant-plane gradient without copying them out. */
int risp_stats_bwd_rows(float *gx, const float *g_min, const float *g_mean, const float *g_max,
                        const int32_t *arg, int N, int C, int HW, int row_stride, void *stream);

/* Per-plane histogram with torch.histc(x, bins, min=0, max=1) semantics (raw counts as fp32;
 * out-of-range values ignored; x == 1 in the last bin) - the conditional heads' feature,
 * tools_origin.py:120-128 (computed on the CPU there).  hist is (NC,bins), fully written. */
int risp_histc(const float *x, float *hist, int NC, int HW, int bins, void *stream);

/* SRCNNRes broadcast-plane values (srcnn_res_arch.py:36-43): cvals (N,9+P) =
 * [min_b,min_g,min_r | mean | max | pv (N,P)] from the stats of the (N,3,H,W) input. */
int risp_srcnn_cvals(const float *stats, const float *pv, float *cvals, int N, int P, int HW, void *stream);
/* the same values applied to a weight-only table: table (N,M) = cvals (N,9+P) @ rcase (9+P,M), terms in index order -
 * the per-(image, cout, border case) constants that replace the 9+P broadcast planes of SRCNNRes' 9x9 first layer
 * (srcnn_res_arch.py:41-46; M = 64 * 81, consumed by risp_conv2d with RISP_EPI_CASEBIAS). */
int risp_srcnn_case_table(const float *stats, const float *pv, const float *rcase, float *table, int N, int P, int HW,
                          int M, void *stream);

/* whitebalance 'grayworld' - tools_origin.py:33-41 ("output is clipped to [0, 1]", :22).
 * OPSPEC: gains[n,c] = gray_n / max(mean[n,c],1e-6), gray = mean over c of the channel means;
 * y = clamp(x * gains).  Composition: risp_channel_stats -> risp_grayworld_gains_fwd ->
 * risp_gain3_fwd; backward: risp_gain3_bwd -> risp_grayworld_gains_bwd -> risp_stats_bwd. */
int risp_grayworld_gains_fwd(const float *stats, float *gains, int N, int HW, void *stream);
int risp_grayworld_gains_bwd(const float *stats, const float *g_gains, float *g_mean, int N, int HW,
                             void *stream);
/* y[n,c] = clamp(x[n,c] * k[n,c], 0, 1); k is (N,3) */
int risp_gain3_fwd(const float *x, const float *k, float *y, int N, int HW, void *stream);
int risp_gain3_bwd(const float *x, const float *k, const float *gy, float *gx, float *gk, float *scratch, int N,
                   int HW, void *stream);

/* ---------------------------------------------------------------------------
 * Fused element-wise segment of a fixed pipeline (isp_universal.py:210-232): one
 * launch reads the segment input once and writes EVERY stage output
 * (intermediate_results is API).  ops[k] in RISP_OP_*; params[k] = (N,P_k) or NULL;
 * outs[k] = (N,3,H,W) (NULL for RISP_OP_SKIP, which aliases its input).
 * The first op may be RISP_OP_DEMOSAIC_NEAREST (input (N,1,H,W)); all others take BGR.
 * ------------------------------------------------------------------------- */
enum {
    RISP_OP_SKIP = 0,
    RISP_OP_DEMOSAIC_NEAREST = 1,
    RISP_OP_WB_MANUAL = 2,
    RISP_OP_GAMMA = 3,
    RISP_OP_GTM_MANUAL = 4,
    RISP_OP_WB_QUADRATIC = 5,
    RISP_OP_GAIN3 = 6 /* y_c = clamp(x_c * p[n,c]): gray-world apply with precomputed gains */
};
#define RISP_MAX_CHAIN 8
int risp_chain_fwd(const float *in, int n_ops, const int *ops, const float *const *params,
                   float *const *outs, int N, int H, int W, void *stream);

/* ---------------------------------------------------------------------------
 * Mixed-op combiner (super_prune_fifteen_demos_four_bayer_two.py:183-212):
 * y = sum_k w[k] * o_k  over K op outputs of one slot (w = pruned, renormalised probs,
 * host array).  bwd: go_k = w[k]*gy (written only where go[k] != NULL) and
 * gw[k] = <gy, o_k> (device, K floats, fully written).
 * ------------------------------------------------------------------------- */
#define RISP_MAX_MIX 16
int risp_mix_fwd(const float *const *outs, const float *w, int K, float *y, size_t numel, void *stream);
/* scratch: risp_mix_scratch_floats() floats (one partial row per workgroup, added in index order: bit-repeatable) */
size_t risp_mix_scratch_floats(void);
int risp_mix_bwd(const float *const *outs, const float *w, int K, const float *gy,
                 float *const *go, float *gw, float *scratch, size_t numel, void *stream);

/* The mixture of a slot with its ELEMENT-WISE operators computed on the fly: operand k of y = sum_k w[k] o_k is either a
 * materialised (N,3,H,W) tensor (kind RISP_SLOT_TENSOR, ptr = the tensor: the CNN proxies) or o_k = op(x, params) for
 * kind RISP_OP_SKIP / WB_MANUAL / GAMMA / GTM_MANUAL / WB_QUADRATIC / GAIN3 (ptr = the (N,P) parameter block; at most one
 * operand of each kind) evaluated in registers from the slot input x: x is read once and the operators' outputs never
 * touch HBM.  pmul[k] multiplies a WB_MANUAL block on the way in (the wrapper's params * 5, tools_origin.py:214) and its
 * gradient on the way out.  y is bit-identical to running the operators one by one and risp_mix_fwd.
 * Backward: gw (K) = <gy, o_k>; go[k] (tensor operands, may be NULL) = w[k] gy; gx = sum over the element-wise operands of
 * their input gradients at upstream w[k] gy, added in KIND order (skip, manual white balance, gamma, tone curve, quadratic
 * white balance, gain - whatever the operand order; NULL allowed only when there is none); gp[k] = (N,P) parameter-gradient
 * block of operand k (GTM_MANUAL: whole batch in row 0), fully written, may be NULL.  Deterministic.  Identical bits to the
 * unfused slot: y, gp (and go); gw and gx differ from it by summation order (~1e-7). */
#define RISP_SLOT_TENSOR (-1)
#define RISP_SLOT_ROW (RISP_MAX_MIX + 40)
typedef struct risp_slot_mix_desc {
    int K, N, HW;
    int kind[RISP_MAX_MIX];
    float w[RISP_MAX_MIX], pmul[RISP_MAX_MIX];
    const float *ptr[RISP_MAX_MIX];
    float *go[RISP_MAX_MIX];                      /* backward */
    float *gp[RISP_MAX_MIX];                      /* backward */
    const float *x;
    float *y;                                     /* forward */
} risp_slot_mix_desc;
int risp_slot_mix_fwd(const risp_slot_mix_desc *d, void *stream);
size_t risp_slot_mix_scratch_floats(int N, int HW);
int risp_slot_mix_bwd(const risp_slot_mix_desc *d, const float *gy, float *gx, float *gw, float *scratch, void *stream);

/* ---------------------------------------------------------------------------
 * Convolution layers of the learned proxies on fp32 MFMA
 * (srcnn_res_arch.py:15-24, srcnn_demosaic_arch.py:14-25, path_14l_*_arch.py:6-57).
 * Stride 1, 'same' zero padding, odd square kernels.
 *
 *   y = epilogue( conv(load(x), W) + bias )
 *
 * wpack: weights repacked on the device by risp_conv_pack_weights into the
 * [chunk][tap][ci][cout_pad] slabs the kernel stages in LDS.
 * ------------------------------------------------------------------------- */
enum { /* how the kernel reads its input tensor */
    RISP_LOAD_PLAIN = 0,      /* x is (N,cin,H,W) */
    RISP_LOAD_UNSHUFFLE2 = 1, /* x is (N,cin/4,2H,2W); plane 4c+2i+j = x[c][2y+i][2x+j].  cin=4: the
                                 Bayer -> [R,G1,G2,B] split (path_14l_bayer_arch.py:70-75); also the
                                 backward of a PixelShuffle(2) store */
    RISP_LOAD_CONSTCH = 2     /* channels >= cin_img are per-image constants cvals[n,ci-cin_img]
                                 inside the image and 0 in the padding (SRCNNRes broadcast planes,
                                 srcnn_res_arch.py:41-46) */
};
enum { /* epilogue flags */
    RISP_EPI_RELU = 1,       /* y = max(y,0) */
    RISP_EPI_ADD = 2,        /* y[:, :add_c] += add (N,add_c,H,W), before the activation */
    RISP_EPI_MASK = 4,       /* y = (mask[n,co,h,w] > 0) ? y : 0   (backward through a ReLU) */
    RISP_EPI_SHUFFLE2 = 8,   /* store through PixelShuffle(2): (N,cout,H,W) -> (N,cout/4,2H,2W) */
    RISP_EPI_NOBIAS = 16,
    RISP_EPI_CASEBIAS = 32   /* y[n,co,h,w] += cvals[n,co,case(h),case(w)]: cvals is a (N,cout,k,k) table, case(v) = v
                                for v < k/2, k/2 in the interior, k-1-(L-1-v) for the last k/2 coordinates - the
                                contribution of spatially constant input channels (zero in the padding) folded out
                                of the layer: SRCNNRes broadcast planes, srcnn_res_arch.py:41-46.  Needs H,W >= k-1 */
};
typedef struct {
    int N, H, W;             /* H,W = conv resolution (half the source for RISP_LOAD_UNSHUFFLE2) */
    int cin, cout, ksize;    /* cout <= 64; ksize in {1,3,5,9} */
    int load_mode, cin_img;  /* cin_img: image channels when load_mode == RISP_LOAD_CONSTCH */
    int epilogue, add_c;
    const float *x, *wpack, *bias, *cvals, *add, *mask;
    float *y;
    /* Grouped launch (group_n = 0: off).  The same-geometry operators of a super-net slot - the 8 SRCNNRes proxies of an
     * sRGB slot, the 2 proxy demosaics (super_prune_fifteen_demos_four_bayer_two.py:35-52, looped at :183-212) - run each
     * layer as ONE launch: the N images are N / group_n consecutive groups of group_n images, group g convolves with
     * wpack + g * wpack_gs and bias + g * bias_gs (floats); y, mask, cvals (and x, add unless shared) are the groups'
     * tensors stacked along N.  RISP_GROUP_SHARED_X / _ADD: x / add hold ONE group's group_n images, read by every
     * group (the slot input).  Per image the arithmetic is that of the ungrouped launch: results are bit-identical. */
    int group_n, group_flags;
    long long wpack_gs, bias_gs;
} risp_conv_desc;
enum { RISP_GROUP_SHARED_X = 1, RISP_GROUP_SHARED_ADD = 2 };
size_t risp_conv_wpack_floats(int cin, int cout, int ksize);
/* w: device (cout,cin,k,k) torch layout.  transpose=1: w is the FORWARD layer's (cin,cout,k,k)
 * tensor and the pack holds the backward-data convolution (roles swapped, taps rotated 180). */
int risp_conv_pack_weights(const float *w, int cin, int cout, int ksize, int transpose, float *wpack,
                           void *stream);
int risp_conv2d(const risp_conv_desc *d, void *stream);

/* The same operator for layers with cout <= 12 (SRCNNRes conv 5x5 32->3, srcnn_res_arch.py:22; backward-data of its
 * 9x9 first layer :18 restricted to the 3 image channels; Path-Restore tails; SRCNNDemosaic conv 5x5 32->12,
 * srcnn_demosaic_arch.py:21): a direct vector-FMA kernel - the matrix-core kernel above pads cout to 32.
 * wpack: [cin][k][k][P] floats, P = risp_conv_small_cout_pad(cout) = 4 or 12 (couts zero-padded), 16-byte
 * aligned - the layer's weights w[co][ci][ky][kx] for a forward layer, w[ci_b][co_b][k-1-ky][k-1-kx] for a
 * backward-data layer.  cout == 3 (every proxy tail): [cin][k + 1][k][4] = (w0, w1, w2[ky], w2[ky - 1]) with w2[-1] = 0 and
 * row k = (0, 0, 0, w2[k - 1]) - the fourth lane of the packed FMAs then serves the third cout of a second output row.  load_mode PLAIN; epilogue RELU | ADD | MASK | NOBIAS, or SHUFFLE2 [| NOBIAS] with cout % 4 == 0;
 * ksize in {3,5,9} (9 only for cout <= 4). */
int risp_conv_small_cout_pad(int cout);
size_t risp_conv_small_wpack_floats(int cin, int cout, int ksize);
int risp_conv2d_small(const risp_conv_desc *d, void *stream);
/* Small grids (the per-GPU batch of the 8-GPU search is 4 images): the input channels are split over `groups`
 * workgroups per tile, partial sums go to scratch (groups * N * cout * H * W floats) and a second launch adds them
 * in index order and applies the epilogue (deterministic).  risp_conv_small_groups: the split worth using (1 = none). */
int risp_conv_small_groups(const risp_conv_desc *d);
int risp_conv2d_small_split(const risp_conv_desc *d, float *scratch, int groups, void *stream);

/* The same operator for 3x3 layers with a one-dimensional Winograd transform F(4,3) along x, fp32 throughout (half the
 * matrix-core work of risp_conv2d; Path-Restore's 64->64 layers, path_14l_bayer_arch.py:6-21, when RISP_CONV_ARITH=f32):
 * wpack [cout block of 32][chunk of risp_conv_wino43_chunk() cin][ky][t][ci][32], U_t = (G g)_t, G rows (1/4,0,0)
 * (1/6,1/6,1/6) (1/6,-1/6,1/6) (1/24,1/12,1/6) (1/24,-1/12,1/6) (0,0,1) applied to filter row g = w[co][ci][ky][0..2]
 * (backward-data: the forward weight with roles swapped and taps rotated by 180 degrees).  load_mode PLAIN, W % 4 == 0,
 * 16-byte aligned tensors, cout <= 64; epilogue RELU | ADD | MASK | NOBIAS; no grouped launches.  Layers with 33..64 couts
 * and cin % 4 == 0 run both cout blocks in one wave (same results bit for bit).  The library reads no environment. */
int risp_conv_wino43_chunk(void);
size_t risp_conv_wino43_wpack_floats(int cin, int cout);
int risp_conv2d_wino43(const risp_conv_desc *d, void *stream);

/* The same operator for 5x5 layers with at most 3 INPUT channels and 32 or 64 output channels in split precision on the f16 matrix
 * pipe (round 6, risp_conv_thin5.hip): the backward-data pass of SRCNNRes' last layer (srcnn_res_arch.py:22 - the upstream gradient's 3
 * image channels -> 32 hidden channels, masked by the ReLU of the layer before, :20), which the fp32 Winograd kernel served at 4 padded
 * channels.  Reduction index of a matrix instruction = (filter row, channel): 15 of 16 slots; the filter column shifts the pixel operand.
 * wpack (risp_conv_thin5_wpack_bytes(cout) bytes, 16-byte aligned): a 16-byte header whose first float is 1 / s_w, then [cout block of
 * 32][kx][part: hi, lo][half of the reduction index][cout][8] halves of w * s_w, reduction index k = 3 ky + c (k = 15 and missing
 * channels zero) - the layer's weights w[co][c][ky][kx] (backward-data: the forward weight with roles swapped and taps rotated by 180
 * degrees).  load_mode PLAIN; epilogue RELU | MASK | NOBIAS; cout * H * W * 4 < 2^31; any W; grouped launches.  One scale per work item
 * (image, 128-column strip, 32-row segment): a result does not depend on the batch; every sum in a fixed order. */
size_t risp_conv_thin5_wpack_bytes(int cout);
int risp_conv2d_thin5(const risp_conv_desc *d, void *stream);

/* The same operator for 3x3 layers with at most 4 OUTPUT channels and 16 .. 64 input channels (a multiple of 16) in split precision on
 * the f16 matrix pipe (round 6, risp_conv_narrow3.hip): Path-Restore's last layer (path_14l_bgr_arch.py, path_14l_bayer_arch.py: 64 -> 3,
 * 64 -> 4 + PixelShuffle) and the backward-data pass of its first layer.  Rows of a matrix instruction = (filter row, cout), reduction
 * index = 16 channels, the filter column shifts the pixel operand; a wave walks down the rows and the three filter rows' contributions
 * meet in registers.  wpack (risp_conv_narrow3_wpack_bytes(cin) bytes, 16-byte aligned): a 16-byte header whose first float is 1 / s_w,
 * then [chunk of 16 cin][kx][part: hi, lo][channel half][row m, 32][8 channels] halves of w * s_w, row m = 4 ky + co (other rows zero) -
 * the layer's weights w[co][ci][ky][kx] (backward-data: the forward weight with roles swapped and taps rotated by 180 degrees).
 * load_mode PLAIN; epilogue RELU | SHUFFLE2 (cout == 4) | NOBIAS; cin * H * W * 4 < 2^31; any W; grouped launches.  One scale per wave
 * and input row: a result depends neither on the batch nor on how the launch cuts its row segments; every sum in a fixed order. */
size_t risp_conv_narrow3_wpack_bytes(int cin);
int risp_conv2d_narrow3(const risp_conv_desc *d, void *stream);

/* The same operator for the FIRST layers of the proxies - few input channels, the whole weight matrix staged once per
 * workgroup - with a LINEAR reduction index k = (ci * ksize + ky) * ksize + kx, so that the matrix instruction's two k-slots
 * hold consecutive k (3 channels of a 9x9 layer: 122 instructions per tile instead of the 162 that channel PAIRS cost in
 * risp_conv2d):  9x9 3 -> cout (SRCNNRes with its broadcast planes folded out, srcnn_res_arch.py:18, 41-46), 3x3 3 -> cout
 * (path_14l_bgr_arch.py:40-43) with load_mode PLAIN;  9x9 4 -> cout (srcnn_demosaic_arch.py:14-16, 39-43) and 3x3 4 -> cout
 * (path_14l_bayer_arch.py:37-40, 70-75) with load_mode UNSHUFFLE2 (x = the (N,1,2H,2W) mosaic).  cout <= 64.
 * wpack: [cout block][2 ceil(cin k k / 2)][B] floats, B = risp_conv_k3_cout_block(cin, ksize) (32 or 64), entry [b][k][c] =
 * w[b B + c][ci][ky][kx], zero beyond the layer, 16-byte aligned.  W % 4 == 0, 16-byte aligned tensors; epilogue RELU |
 * NOBIAS | CASEBIAS. */
int risp_conv_k3_cout_block(int cin, int ksize);
size_t risp_conv_k3_wpack_floats(int cin, int cout, int ksize);
int risp_conv2d_k3(const risp_conv_desc *d, void *stream);

/* 5x5 layers with F(4,5) along x (0.4 of the matrix-core work of risp_conv2d; SRCNNRes' 64->32 layer and its backward,
 * srcnn_res_arch.py:20, when RISP_CONV_ARITH=f32, and the 5x5 backward passes with 3 or 12 input channels): U_t = (G g)_t / s_t,
 * G rows (1,0,0,0,0) (1,1,1,1,1) (1,-1,1,-1,1) (1,2,4,8,16) (1,-2,4,-8,16) (1,1/2,1/4,1/8,1/16) (1,-1/2,1/4,-1/8,1/16)
 * (0,0,0,0,1), s = (1, -18, -18, 360, 360, 45/16, 45/16, 1), applied to filter row g = w[co][ci][ky][0..4].  wpack (layout 1,
 * what risp_conv_wino45_layout() returns): [cout block of 32][chunk of 4 cin][ky][point group 2][cout block of 16: 2][ci 4]
 * [cout 16][4 points] (two output rows per wave).  cin % 4 == 0 or cin < 4; otherwise the restrictions of risp_conv2d_wino43;
 * grouped launches. */
int risp_conv_wino45_chunk(void);
int risp_conv_wino45_layout(void);
size_t risp_conv_wino45_wpack_floats(int cin, int cout);
int risp_conv2d_wino45(const risp_conv_desc *d, void *stream);

/* The same operator on the f16 matrix pipe at fp32-level accuracy (round 4; the 64 -> 64 3x3 layers of Path-Restore,
 * path_14l_bgr_arch.py:6-21, 58-86; path_14l_bayer_arch.py:59-88; the 5x5 64 -> 32 layer of SRCNNRes and its backward,
 * srcnn_res_arch.py:20): every fp32 operand is cut into two f16 halves,
 * hi = rn_f16(v s), lo = rn_f16(v s - hi), and x w is taken as (x_lo w_hi + x_hi w_lo + x_hi w_hi) / (s_x s_w) - three
 * v_mfma_f32_32x32x16_f16 with fp32 accumulation instead of eight fp32 matrix instructions of twice the duration; the
 * dropped term is 2^-22 of the product.  x, y, add, mask stay fp32 tensors: the activations are scaled (per workgroup tile
 * and chunk of 16 input channels, by the tile's own largest magnitude - so gradients of magnitude 1e-8 are as exact as
 * activations of magnitude 1) and split at staging time; the weights are scaled and split once, at pack time.
 * wpack: risp_conv_f16x2_wpack_bytes() bytes, 16-byte aligned: a 16-byte header whose first float is 1 / s_w (s_w = 2^k with
 * max|w| s_w in [2^14, 2^15)), then [chunk of 16 cin][tap][part: hi, lo][channel half][cout padded to 32 or 64][8 channels]
 * _Float16 (reconfigisp_amd/convnets.py::f16x2_weights).  ksize 3 or 5, cin % 16 == 0, cout 32 or 64, add_c == cout, W % 4 == 0,
 * max(cin, cout) * H * W * 4 < 2^31 bytes per image, 16-byte aligned tensors, load_mode PLAIN; epilogue RELU | ADD | MASK | NOBIAS; grouped
 * launches (group_n, strides in floats).  A tile's result does not depend on the batch it travels in. */
size_t risp_conv_f16x2_wpack_bytes(int cin, int cout, int ksize);
int risp_conv2d_f16x2(const risp_conv_desc *d, void *stream);
/* The same layer, same pack, same bits by the form risp_conv2d_f16x2 had in round 4: two 4-wave workgroups per CU in which every
 * wave stages, waits and multiplies (risp_conv2d_f16x2 = one 8-wave workgroup per CU, four waves stage tiles and weights a chunk
 * ahead, four issue the matrix instructions).  Serves the one-chunk 3x3 64-cout shape the wave-specialised form does not take;
 * tests/test_gpu_f16x2.py holds the two to identical bits on every shape and epilogue.  The library keeps no switch between them. */
int risp_conv2d_f16x2_uniform(const risp_conv_desc *d, void *stream);

/* The same arithmetic for layers with at most 4 output channels and a 5- or 9-tap filter row (round 4; SRCNNRes conv 5x5 32 -> 3,
 * srcnn_res_arch.py:22; backward-data of its 9x9 first layer restricted to the 3 image channels, :18; backward-data of
 * SRCNNDemosaic's 9x9 4 -> 64 through PixelShuffle, srcnn_demosaic_arch.py:14-16): the rows of the matrix instruction are
 * (cout, position j inside a block of 8 pixels), its columns the 32 blocks of a 256-pixel row, its reduction index a window of 16
 * input pixels of one input channel and filter row - a Toeplitz band of the filter row as the A operand.
 * wpack: risp_conv_toep_wpack_bytes() bytes, 16-byte aligned: a 16-byte header whose first float is 1 / s_w, then
 * [cin][ky][part: hi, lo][window half][row m = 8 cout + j, padded to 32][8 window slots] _Float16 with
 * band[m][u] = w[co][ci][ky][u - j + k/2 - 4] s_w (0 outside the filter row; reconfigisp_amd/convnets.py::toep_weights).
 * cout <= 4 (ksize 9 or 5), or 5 .. 12 with ksize 5 (SRCNNDemosaic's 5x5 32 -> 12 tail, srcnn_demosaic_arch.py:21: three row blocks,
 * rows padded to 96); any cin, W % 4 == 0, fewer than 2^30 input elements per image, 16-byte aligned tensors, load_mode PLAIN;
 * epilogue RELU | ADD (add_c <= cout) | NOBIAS, or SHUFFLE2 [| NOBIAS] with cout % 4 == 0; grouped launches. */
size_t risp_conv_toep_wpack_bytes(int cin, int cout, int ksize);
int risp_conv2d_toep(const risp_conv_desc *d, void *stream);
/* The same launch also writing, per tile of 16 rows x 256 columns, the sum of every input channel over the tile's own pixels:
 * psum [N][risp_conv_toep_tiles(H, W)][cin] floats.  risp_rect_sums_tiles (below, next to risp_rect_sums) finishes them into the
 * rectangle sums of SRCNNRes' constant planes (srcnn_res_arch.py:41-46) from the border rows and columns alone - the 64-channel
 * upstream gradient is read once, by the backward-data convolution. */
int risp_conv_toep_tiles(int H, int W);
int risp_conv2d_toep_sums(const risp_conv_desc *d, float *psum, void *stream);

/* The layers with at most 3 output channels and cin % 16 == 0 (SRCNNRes conv 5x5 32 -> 3 forward, srcnn_res_arch.py:22; backward-data of
 * its 9x9 first layer restricted to the 3 image channels, :18) with the filter ROWS moved into the rows of the matrix instruction
 * (round 6, risp_conv_tapout.hip): reduction index = 16 input channels (dense), rows = (cout, ky), the filter column kx = a shift of the
 * pixel operand; one matrix pass per INPUT row, whose 3 x k results are added into a ring of output rows (every address by one lane
 * in program order: bit-repeatable).  Half the matrix instructions of risp_conv2d_toep for 9x9 64 -> 3.  Split precision as in
 * risp_conv2d_f16x2, activations scaled per (4 rows x 136 columns, 16 channels).
 * wpack: risp_conv_tapout_wpack_bytes() bytes, 16-byte aligned: a 16-byte header whose first float is 1 / s_w, then [chunk of 16
 * cin][kx][part: hi, lo][channel half][row m, 32][8 channels] _Float16 with row m = 4 ky + co (ky < 8) or 4 co + 3 (ky = 8)
 * (reconfigisp_amd/convnets.py::tapout_weights).  ksize 5 or 9, cout <= 3, cin % 16 == 0, W % 4 == 0, H * W < 2^24, cin * H * W < 2^30;
 * load_mode PLAIN; epilogue RELU | ADD (add_c <= cout) | NOBIAS; grouped launches.  seg_rows: rows of a work item's segment - 0 = chosen by the launch to fill
 * the chip (training launches), else a multiple of 4 fixed by the caller: an image's result then does not depend on the batch it
 * travels in (the scales follow the segment's row phase). */
size_t risp_conv_tapout_wpack_bytes(int cin, int ksize);
int risp_conv_tapout_seg_rows(int N, int H, int W);          /* what seg_rows = 0 chooses for a launch of N images (a pure function) */
int risp_conv2d_tapout(const risp_conv_desc *d, int seg_rows, void *stream);
/* ... and, on the way, the sum of every input channel over every work item's own pixels (ksize 9, cin 64):
 * psum [N][risp_conv_tapout_items(N, H, W, seg_rows)][64] floats, finished by risp_rect_sums_tiles. */
int risp_conv_tapout_items(int N, int H, int W, int seg_rows);
int risp_conv2d_tapout_sums(const risp_conv_desc *d, int seg_rows, float *psum, void *stream);

/* ... and for the 9x9 FIRST layers (few input channels, 64 couts: SRCNNRes 3 -> 64 with its broadcast planes folded out,
 * srcnn_res_arch.py:18, 41-46; SRCNNDemosaic 4 -> 64 on the mosaic, srcnn_demosaic_arch.py:14-16, 39-43 - what risp_conv2d_k3
 * does on the fp32 matrix pipe): rows = 32 couts, one accumulator per pixel position j of a block of 8, the 8 A operands are
 * windows of one zero-padded filter row per lane.  wpack: risp_conv_toep_first_wpack_bytes() bytes, 16-byte aligned: a 16-byte
 * header whose first float is 1 / s_w, then [cout block of 32][cin][ky][part: hi, lo][taps 0-7 | tap 8 and 7 zeros][cout][8] _Float16
 * (reconfigisp_amd/convnets.py::toep_first_weights).  ksize 9, cin <= 16, W % 4 == 0, 16-byte aligned tensors; load_mode PLAIN, or
 * UNSHUFFLE2 with cin == 4 (x = the (N,1,2H,2W) mosaic); epilogue RELU | NOBIAS | CASEBIAS; grouped launches. */
size_t risp_conv_toep_first_wpack_bytes(int cin, int cout);
int risp_conv2d_toep_first(const risp_conv_desc *d, void *stream);
/* ... with exact ReLU decisions, for training forwards: outputs whose pre-activation lies within the arithmetic's own error of zero
 * (|z| < 2^-20 x 81 x max|w| x the tile's input magnitude summed over the channels) are listed in `ties` ([0] = count, then up to
 * max_ties linear output indices; cleared by this call) and a second launch recomputes them in double - exact products, fixed
 * order, one rounding - from the layer's fp32 weights w32 (cout,cin,9,9), the members of a grouped launch w32_gs floats apart.
 * Fewer than 2^32 outputs.  Which way such an activation falls otherwise depends on the summation order of whichever fp32-accurate
 * kernel computed it, and one ReLU mask bit is a finite step of every gradient behind it.  (Opt-in: RISP_CONV_TOEP_FIRST=train.) */
int risp_conv2d_toep_first_exact(const risp_conv_desc *d, const float *w32, long long w32_gs, unsigned *ties, unsigned max_ties,
                                 void *stream);

/* out[p][ky][kx] = sum of g[p] (planes x H x W) over the pixels q with q + (ky - k/2, kx - k/2) inside the plane:
 * what the backward of a k x k convolution over a spatially CONSTANT input channel needs from the upstream gradient
 * (d loss / d constant = sum_{co,tap} w[co][c][tap] * out[co][tap]).  k odd <= 9, H, W >= k/2. */
int risp_rect_sums(const float *g, float *out, int planes, int H, int W, int ksize, void *stream);
/* out as risp_rect_sums with ksize 9, from psum [N][tiles][C] (risp_conv2d_toep_sums) and the 4 border rows / columns of the planes
 * g (N,C,H,W); H, W >= 4, W % 4 == 0.  Another summation order than risp_rect_sums (agreement ~1e-6 of the plane's sum). */
int risp_rect_sums_tiles(const float *g, const float *psum, float *out, int N, int C, int H, int W, int tiles, void *stream);
/* ... and that product: gconst (N,C) = rs (N,M) @ wconst (M,C) (SRCNNRes: M = 64 * 81 rectangle sums per
 * image, C = 9+P constant planes, srcnn_res_arch.py:41-46).  Deterministic. */
int risp_srcnn_const_grad(const float *rs, const float *wconst, float *gconst, int N, int M, int C, void *stream);

/* ---------------------------------------------------------------------------
 * The same-geometry proxies of a super-net slot as ONE launch per layer (the 8 SRCNNRes of an sRGB slot,
 * super_prune_fifteen_demos_four_bayer_two.py:35-52 / :183-212; srcnn_res_arch.py:15-53).  The convolutions use
 * risp_conv_desc.group_n; these are the non-convolution links of the chain for all members at once.  Member g has P[g]
 * parameter channels; tensors of the group are the members' tensors stacked along N (member-major: row g * N + n).
 * Per member every value is computed exactly as by the single-operator entry point named beside it.
 * ------------------------------------------------------------------------- */
#define RISP_MAX_GROUP 16
typedef struct risp_srcnn_group_desc {
    int G, N, HW, M;                              /* members, images per member, H*W, table width (cout * k * k) */
    int P[RISP_MAX_GROUP];
    const float *pv[RISP_MAX_GROUP];              /* (N, P[g]) parameter blocks (forward) */
    const float *rcase[RISP_MAX_GROUP];           /* (9 + P[g], M) folded first-layer tables (forward) */
    const float *wconst[RISP_MAX_GROUP];          /* (M, 9 + P[g]) constant-plane weights (backward) */
} risp_srcnn_group_desc;
/* table (G*N, M): risp_srcnn_case_table per member; stats (N,3,4) of the shared input */
int risp_srcnn_case_table_group(const float *stats, const risp_srcnn_group_desc *d, float *table, void *stream);
/* gconst (G*N, row), row >= max(9 + P): risp_srcnn_const_grad per member into columns [0, 9 + P[g]), zeros beyond */
int risp_srcnn_const_grad_group(const float *rs, const risp_srcnn_group_desc *d, float *gconst, int row, void *stream);
/* out (N,C,H,W) = sum over the members, in member order, of stack (G,N,C,H,W).  gstats != NULL: member g's term first
 * receives risp_stats_bwd_rows with g_min / g_mean / g_max = columns [0,C) / [C,2C) / [2C,3C) of gstats (G*N, row) and
 * the argmin / argmax indices `arg` (N,C,2) of the shared input. */
int risp_group_sum(const float *stack, float *out, int G, int N, int C, int HW, const float *gstats, int row,
                   const int32_t *arg, void *stream);

/* Backward-weight of the same layer: dw (cout,cin,k,k) = sum_{n,y,x} gy[n,co,y,x] * load(x)[n,ci,y+ky-p,x+kx-p]
 * (fully written).  Uses d->x, load_mode (PLAIN / CONSTCH), cin_img, cvals, N, H, W, cin, cout, ksize; gy is
 * (N,cout,H,W).  scratch: scratch_floats >= risp_conv_wgrad_scratch_floats(cin, cout, ksize) floats, checked (per-workgroup partial sums:
 * 768 slices of the pixel tiles, dealt over the 32 x 32 blocks of (cout, cin), k * k slots each - k for the thin layers whose filter column
 * is packed into the matrix -, added in index order by a finishing launch - no atomics, the same bits on every run).
 * Only the proxy fine-tuning path (darts_ft_model.py:206-246) needs weight gradients. */
size_t risp_conv_wgrad_scratch_floats(int cin, int cout, int ksize);
int risp_conv2d_wgrad(const risp_conv_desc *d, const float *gy, float *dw, float *scratch, size_t scratch_floats, void *stream);

/* sum over H,W of channels [c0, c0+nc) of an (N,C,H,W) tensor -> out (N,nc) (SRCNNRes
 * gradient of the broadcast parameter planes). */
int risp_plane_sums(const float *x, float *out, int N, int C, int c0, int nc, int HW, void *stream);

/* ---------------------------------------------------------------------------
 * Conditional modules (sRGB pool 16-18): the fully connected head, ConditionalModuleBGR._fc_forward
 * (tools_origin.py:109-163).  hist (N, widths[0]) raw histogram counts (risp_histc; no gradient);
 * flat: per layer an (in,out) row-major weight then a bias, then the "global" block whose FIRST entry is
 * added to every output unit (:158-160); ReLU between layers, sigmoid at the end.
 * acts / deltas: (N, risp_cond_fc_row_floats) scratch rows kept by the caller between forward and backward.
 * out (N, widths[n_layers]); dflat (total_params) is fully written (deterministic, no atomics).
 * ------------------------------------------------------------------------- */
int risp_cond_fc_row_floats(const int *widths, int n_layers);
int risp_cond_fc_fwd(const float *hist, const float *flat, const int *widths, int n_layers, float *acts, float *out, int N,
                     void *stream);
int risp_cond_fc_bwd(const float *flat, const int *widths, int n_layers, const float *acts, const float *out, const float *gout,
                     float *deltas, float *dflat, int total_params, int N, void *stream);

/* ---------------------------------------------------------------------------
 * Bookkeeping of one super-net slot as single launches (models/modules/super_prune_fifteen_demos_four_bayer_two.py).
 * ------------------------------------------------------------------------- */
/* mixture weights, :185-193: prob = softmax(alpha) (entries flagged in `unavailable` get probability exactly 0);
 * entries with prob < threshold * max(prob) (strict) are pruned; post = kept / sum(kept).  The mask and the sum are
 * detached in the reference, so d post_k / d prob_k = coef_k (1 / sum for kept entries, 0 for pruned ones).
 * K <= 64.  probs, coef, post: K floats each (probs and coef feed the backward). */
int risp_prune_softmax_fwd(const float *alpha, const unsigned char *unavailable, float threshold, int K, float *probs,
                           float *coef, float *post, void *stream);
int risp_prune_softmax_bwd(const float *probs, const float *coef, const float *gpost, int K, float *galpha, void *stream);

/* per-image parameter blocks of the ops of a slot, :204-209 / isp_universal.py:226-228:
 * block[k] (N, width[k]) = sigmoid(raw[k]).repeat(N, 1); backward: graw[k] = sigmoid' * sum over the images of
 * gblock[k] (NULL = no gradient arrived: zeros), images in index order (deterministic). */
#define RISP_MAX_PARAM_OPS 16
typedef struct risp_param_blocks_desc {
    int n_ops, N;
    int width[RISP_MAX_PARAM_OPS];                /* 1..64 */
    const float *raw[RISP_MAX_PARAM_OPS];         /* (width) */
    float *block[RISP_MAX_PARAM_OPS];             /* forward: (N, width) out */
    const float *gblock[RISP_MAX_PARAM_OPS];      /* backward: (N, width) in, may be NULL */
    float *graw[RISP_MAX_PARAM_OPS];              /* backward: (width) out */
    int gstride[RISP_MAX_PARAM_OPS];              /* backward: floats between the rows of gblock[k] (0 = width: packed) */
} risp_param_blocks_desc;
int risp_param_blocks_fwd(const risp_param_blocks_desc *d, void *stream);
int risp_param_blocks_bwd(const risp_param_blocks_desc *d, void *stream);

/* ---------------------------------------------------------------------------
 * The glue of a DARTS iteration as own launches (risp_step.hip).
 * ------------------------------------------------------------------------- */
/* loss[0] = mean((y - gt)^2) (kind 0: nn.MSELoss, models/darts_model.py:58-63, isp_model.py:29-34) or mean(|y - gt|) (kind 1:
 * nn.L1Loss) over numel values, numel % 4 == 0, 16-byte aligned; g != NULL: also the gradient of that mean with respect to y
 * at upstream 1 (2 (y - gt) / numel, sign(y - gt) / numel) - the backward pass is then a scaling.  Two launches (partial sums
 * per workgroup into scratch = risp_loss_scratch_floats() floats, added in index order): deterministic. */
size_t risp_loss_scratch_floats(void);
int risp_pixel_loss(const float *y, const float *gt, float *g, float *loss, float *scratch, size_t numel, int kind, void *stream);

/* local_global_loss with the mean-squared loss (utils/util_loss.py:26-64, pixel_criterion local_global_l2): images with flag[n] < 1
 * ("local"): MSE of (a * gain, b), gain[n][c] = clamp(mean b / (clamp(mean a, 0) + 1e-6), 0.5, 2), no gradient through the gain; the
 * others ("global"): MSE of the 1/4-scale bilinear down-samples (align_corners false; H, W % 4 == 0); loss[0] = the sum of the two means
 * (an empty branch contributes 0), g (may be NULL) = d loss / d a.  flag (N) floats ON THE DEVICE: no host read, no boolean indexing.
 * scratch: risp_local_global_scratch_floats(N, C) floats; four launches, partial sums added in index order. */
size_t risp_local_global_scratch_floats(int N, int C);
int risp_local_global_l2(const float *a, const float *b, const float *flag, float *g, float *loss, float *scratch, int N, int C, int H, int W,
                         void *stream);

/* The reference's per-parameter Python loops over the <= 216 floats of a super-net as ONE launch over a table of tensors. */
#define RISP_MAX_LIST 64
typedef struct risp_list_desc {
    int n;                                        /* tensors, <= RISP_MAX_LIST */
    int numel[RISP_MAX_LIST];
    float *a[RISP_MAX_LIST];                      /* written */
    const float *b[RISP_MAX_LIST];
    const float *c[RISP_MAX_LIST];
    const float *e[RISP_MAX_LIST];
} risp_list_desc;
/* virtual step, darts_model.py:204-222: a = b - lr_meta * (momentum * e + c); e NULL: no momentum buffer yet; c NULL: a = b
 * (no gradient arrived, or an alpha copied to the twin network).  Operation by operation the reference's arithmetic. */
int risp_darts_virtual_step(const risp_list_desc *d, float momentum, float lr_meta, void *stream);
/* out[0] = 2-norm of the concatenation of the c[t] (NULL entries skipped), out[1] = eps = out[0] < 1e-6 ? 0 : 0.01 / out[0]
 * (:274-277); fixed summation order.  Lists longer than RISP_MAX_LIST go through risp_list_norm_eps_part in pieces: first = 0
 * continues from the sum of squares the previous piece left in out[0]; last = 0 leaves the running sum of squares there instead of
 * finishing (risp_list_norm_eps = one piece with first = last = 1). */
int risp_list_norm_eps(const risp_list_desc *d, float *out, void *stream);
int risp_list_norm_eps_part(const risp_list_desc *d, float *out, int first, int last, void *stream);
/* a[t] += (factor * scalar[0]) * c[t] (c NULL: untouched): the +eps, -2 eps, +eps shifts of the parameters (:299-312) with
 * eps on the device. */
int risp_list_axpy_scalar(const risp_list_desc *d, const float *scalar, float factor, void *stream);
/* The optimizers of the search (darts_model.py:78-81) over the table, in place, as torch.optim writes them (no weight decay,
 * dampening, nesterov, amsgrad): SGD with momentum - e = momentum buffers: e = first ? c : e * momentum + c; a -= lr * e - and
 * Adam - b = exp_avg, e = exp_avg_sq (both updated), lr_step = lr / (1 - beta1^t), bias2_sqrt = sqrt(1 - beta2^t),
 * one_minus_beta = (float)(1 - beta) formed in double like torch's lerp_ / addcmul_ weights.  The denominator is
 * sqrt(exp_avg_sq) / bias2_sqrt + eps with a true division (torch's list-wide CUDA form; its CPU form multiplies by the
 * reciprocal: 1 ulp apart).  Rows whose gradient c is NULL are left alone. */
int risp_sgd_momentum_step(const risp_list_desc *d, float lr, float momentum, int first, void *stream);
int risp_adam_step(const risp_list_desc *d, float lr_step, float beta2, float one_minus_beta1, float one_minus_beta2, float bias2_sqrt,
                   float eps, void *stream);
/* architecture gradient, :254-265 with :313-323: a = b - lr_meta * ((c - e) / 2 * eps[0]); zeros where b, c or e is NULL or
 * the finite-difference term holds a NaN (nan_flags[t] = 1 there; may be NULL).  numel <= 256. */
int risp_darts_alpha_grad(const risp_list_desc *d, const float *eps, float lr_meta, int *nan_flags, void *stream);

/* ---------------------------------------------------------------------------
 * One training step of an element-wise fixed pipeline in two launches - replaces the body of
 * IspModel.optimize_parameters (models/isp_model.py:128-142: output = netG(img); l_pix = cri_pix(output, gt);
 * zero_grad(); l_pix.backward(); optimizer_G.step()) when netG is [nearest demosaic ->] a chain of WbManual / Gamma /
 * GtmManual / WbQuadratic stages (isp_universal.py:210-232), cri_pix is nn.MSELoss or nn.L1Loss (mean) and
 * optimizer_G is torch.optim.Adam without weight decay / amsgrad.  Forward, loss, backward (stage inputs kept in
 * registers), the parameter-gradient reduction (deterministic), the chain rule through sigmoid(raw).repeat(N,1)
 * [* 5 for WbManual] and the Adam update all happen on the device; only in, gt and y touch HBM.
 * ------------------------------------------------------------------------- */
#define RISP_MAX_TRAIN_CHAIN 6
typedef struct risp_train_desc {
    const float *in;                              /* (N,1,H,W) RGGB mosaic if from_bayer, else (N,3,H,W) BGR */
    const float *gt;                              /* (N,3,H,W) */
    float *y;                                     /* (N,3,H,W) pipeline output (IspModel.output); may be NULL */
    int from_bayer;                               /* a nearest-neighbour demosaic (tools_origin.py:278-284) in front */
    int n_ops;                                    /* parametrised stages, 1..RISP_MAX_TRAIN_CHAIN */
    int ops[RISP_MAX_TRAIN_CHAIN];                /* RISP_OP_WB_MANUAL / GAMMA / GTM_MANUAL / WB_QUADRATIC (at most one) */
    float *blocks[RISP_MAX_TRAIN_CHAIN];          /* (N,P_k) per-image parameter blocks the stages read = sigmoid(raw)
                                                     .repeat(N,1) (x 5 for WbManual); REWRITTEN for the next step */
    float *raw[RISP_MAX_TRAIN_CHAIN];             /* (P_k) learnable vectors (param_step<k>_<name>): updated in place */
    float *grad[RISP_MAX_TRAIN_CHAIN];            /* (P_k) d loss / d raw: what .grad holds after backward() */
    float *exp_avg[RISP_MAX_TRAIN_CHAIN];         /* (P_k) Adam first moment, updated in place */
    float *exp_avg_sq[RISP_MAX_TRAIN_CHAIN];      /* (P_k) Adam second moment, updated in place */
    int loss_kind;                                /* 0 = nn.MSELoss, 1 = nn.L1Loss (mean reduction) */
    int N, H, W;                                  /* H, W even */
    float lr_step;                                /* lr / (1 - beta1^t), t = this step's number (1-based) */
    float beta1, beta2;
    float one_minus_beta1, one_minus_beta2;       /* (float)(1 - beta): formed in double and rounded once, as torch forms the
                                                     weights of lerp_ / addcmul_ (1.f - 0.999f is 1.3e-5 off float(0.001)) */
    float bias2_sqrt;                             /* sqrt(1 - beta2^t) */
    float eps;
    float *loss;                                  /* 1 float: the mean loss of this step */
    float *scratch;                               /* risp_train_scratch_floats(N) floats */
} risp_train_desc;
size_t risp_train_scratch_floats(int N);
int risp_chain_train_step(const risp_train_desc *d, void *stream);

/* ---------------------------------------------------------------------------
 * Overlapped tiling (utils/util_path_restore.py:47-134), NCHW on device.
 * positions: host int32 [T][2] (y,x).
 * ------------------------------------------------------------------------- */
int risp_tile_gather(const float *img, float *patches, const int32_t *pos_dev, int T, int C,
                     int H, int W, int h, int w, void *stream);
int risp_tile_blend(const float *patches, float *img, const int32_t *pos_dev, int T, int C,
                    int H, int W, int h, int w, int eh, int ew, void *stream);

/* ---------------------------------------------------------------------------
 * Classical, non-differentiable "Origin" kernels of OriginUniversal (tools_origin.py:445-804).
 * Build-defined OPSPEC (parity unpinned), see oracle/isp_oracle.py origin_*.  Images are NCHW fp32
 * scaled to 0..255; outputs are clipped and rounded to 8-bit codes; reflect-101 borders.
 * in_scale / out_div: every input sample is multiplied by in_scale first and every output code is
 * multiplied by the fp32 reciprocal of out_div last (how PyTorch divides a GPU tensor by a scalar) - (1,1) for the plugin boundary, which receives x255 images and divides the
 * result itself (tools_origin.py:455,471), (255,255) when the fused pipeline works on [0,1] tensors
 * directly (bit-identical: the same fp32 multiply and divide, minus two passes over the image).
 * out_div < 0 (diagnostic): the clip-and-round is skipped and the unquantised value / |out_div| is stored, so that
 * tests can compare the arithmetic in front of the 8-bit rounding at float tolerance (the median always returns codes).
 * ------------------------------------------------------------------------- */
/* demosaic 'bilinear' (laplacian=0) / 'laplacian' (Malvar-He-Cutler, laplacian=1) - :457-468, :491-502 */
int risp_origin_demosaic(const float *bayer, float *bgr, int laplacian, int N, int H, int W, float in_scale,
                         float out_div, void *stream);
/* spatialnoisereduction 'bilateral' - :686-710.  window (N) odd <= 17, sigmas (N) in 0..255 units */
int risp_origin_bilateral(const float *x, float *y, const int32_t *window, const float *sigma_color,
                          const float *sigma_space, int max_window, int N, int H, int W, float in_scale,
                          float out_div, void *stream);
/* 'median' - :734-751; one odd size <= 17 for the whole batch; works on 8-bit codes */
int risp_origin_median(const float *x, float *y, int size, int N, int H, int W, float in_scale, float out_div,
                       void *stream);
/* 'fastnlm' - :775-797; block_size / search_block (N) odd, decay (N) */
int risp_origin_fastnlm(const float *x, float *y, const int32_t *block_size, const int32_t *search_block,
                        const float *decay, int max_block, int max_search, int N, int H, int W, float in_scale,
                        float out_div, void *stream);
/* globaltonemapping 'reinhard' (mode 0: a=white_point, b=middle_grey), 'crysisengine' (1: a=lum_adapted),
 * 'filmic' (2: a=white_point, b=exposure_bias) - :526-543, :566-581, :604-623; whitebalance
 * 'whiteworld' (3: a=white_point_ratio, stats from risp_channel_stats) - :647-662.
 * a, b: (N) device arrays; scratch: risp_origin_tonemap_scratch_floats(N) floats (per-image constants and the
 * partial sums of Reinhard's log-average luminance, added in a fixed order: deterministic). */
size_t risp_origin_tonemap_scratch_floats(int N);
int risp_origin_tonemap(const float *x, float *y, int mode, const float *a, const float *b, const float *stats,
                        float *scratch, int N, int HW, float in_scale, float out_div, void *stream);

/* Fused stencil segment (inference): [nearest demosaic ->] bilateral -> element-wise chain in one launch;
 * the BGR halo tile is staged in LDS (straight from the mosaic when from_bayer), every stage output is
 * written ([0,1] domain; the bilateral works on x255 values and returns codes/255 like the reference
 * wrapper, tools_origin.py:690,716).  ops/params/outs as in risp_chain_fwd (no demosaic op). W % 4 == 0. */
int risp_bilateral_chain_fwd(const float *in, int from_bayer, float *out_demosaic, float *out_bilateral,
                             const int32_t *window, const float *sigma_color, const float *sigma_space,
                             int max_window, int n_ops, const int *ops, const float *const *params,
                             float *const *outs, int N, int H, int W, void *stream);

/* ---------------------------------------------------------------------------
 * Input side on the device (what the reference's dataset classes do with numpy / cv2 on the host).
 * sel is a device (N,3) int32 array {frame, row, col}; rows / cols even keep the RGGB phase.
 * ------------------------------------------------------------------------- */
/* uint16 RGGB frames (F,H0,W0) -> (N,1,h,w) fp32 = sample / divisor (1023 / 16383:
 * data/oneplus_rggb2obj_dataset.py:201, data/sid_sony_ratio_rggb2bgr_dataset.py:121-134) */
int risp_raw_crop(const uint16_t *frames, float *out, const int32_t *sel, int N, int H0, int W0, int h, int w,
                  float divisor, void *stream);
/* uint8 HWC BGR ground truth (F,H0,W0,3) -> (N,3,h,w) fp32 / 255 */
int risp_gt_crop(const uint8_t *frames, float *out, const int32_t *sel, int N, int H0, int W0, int h, int w,
                 void *stream);
/* OnePlus "resize by quad" (data/util.py:37-64, oneplus_rggb2obj_dataset.py:109-145): each colour plane of
 * src (H0,W0) is resized to (resized_h/2, W/2) by nearest neighbour and placed pad_top rows down in the
 * zero-filled dst (H,W). */
int risp_resize_rggb(const uint16_t *src, uint16_t *dst, int H0, int W0, int H, int W, int resized_h, int pad_top,
                     void *stream);

/* tensor2bgr + psnr on device (utils/util.py:118-154): truncating uint8 conversion of both images, squared error accumulated
 * in fp64 into sse[0].  sse: risp_sse_uint8_doubles() doubles - sse[1 ..] receive the workgroups' partial sums, added in index
 * order by a finishing launch (no atomics: the same bits on every run). */
size_t risp_sse_uint8_doubles(void);
int risp_sse_uint8(const float *a, const float *b, double *sse, size_t sse_doubles, size_t numel, void *stream);

/* Diagnostics: the kernel instance risp_bilateral_chain_fwd launches for these arguments, named as rocprofv3 prints it
 * (bench.py binds the committed counter readings of profiles/traffic.json to the kernel it actually launches). */
const char *risp_bilateral_chain_kernel(int from_bayer, int max_window, int with_wb_quadratic);

#ifdef __cplusplus
}
#endif
#endif /* RISP_H */
