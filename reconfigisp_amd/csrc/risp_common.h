// Shared helpers for the libreconfigisp_hip kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "risp.h"

#define RISP_WAVE 64

void risp_set_error(const char *fmt, ...);

#define RISP_CHECK_ARG(cond, ...)            \
    do {                                     \
        if (!(cond)) {                       \
            risp_set_error(__VA_ARGS__);     \
            return 1;                        \
        }                                    \
    } while (0)

#define RISP_LAUNCH_CHECK(name)                                                   \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            risp_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return 2;                                                             \
        }                                                                         \
    } while (0)

// NaN-preserving clamp to [0,1] (torch.clamp semantics).
__device__ __forceinline__ float clamp01(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }
// gradient gate of torch.clamp(v,0,1): passes where 0 <= v <= 1
__device__ __forceinline__ float gate01(float v) { return (v >= 0.f && v <= 1.f) ? 1.f : 0.f; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Block-wide sum of NV per-thread partials; thread 0 of the block receives the totals in v[].
// lds must hold NV * (blockDim.x/64) floats.  Contains barriers: call from all threads.
// thread 0's half of a block sum: the waves' partial sums of every value, added in wave order.  With four waves (every launch of
// this library) the NV x 4 LDS reads are issued together and waited for once: a loop over a run-time wave count made them NV x 4
// dependent round trips - 9 us at the end of a workgroup with 30 values, which a launch of ONE resident round of workgroups (the
// quadratic white balance's backward kernels) paid in full (tools/ab_wbq.sh: 16.1 -> 6.7 us of fixed cost).
template <int NV>
__device__ __forceinline__ void block_sum_finish(float (&v)[NV], const float *lds, int nw) {
    if (nw == 4) {
        float q[NV][4];
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int w = 0; w < 4; ++w) q[i][w] = lds[i * 4 + w];
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = (((0.f + q[i][0]) + q[i][1]) + q[i][2]) + q[i][3];
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float s = 0.f;
            for (int w = 0; w < nw; ++w) s += lds[i * nw + w];
            v[i] = s;
        }
    }
}

template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float *lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = wave_sum(v[i]);
        if (lane == 0) lds[i * nw + wave] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) block_sum_finish<NV>(v, lds, nw);
    __syncthreads();
}

// Grouped convolution launch (risp_conv_desc.group_n > 0): the descriptor as image n's group sees it - weights and
// bias of group g = n / group_n; shared tensors rebased so that the kernels' usual indexing by n lands on image
// n - g * group_n.  Wave-uniform (n comes from blockIdx): scalar arithmetic, once per workgroup.
__device__ __forceinline__ risp_conv_desc risp_conv_group_view(risp_conv_desc d, int n) {
    if (d.group_n > 0) {
        const int g = n / d.group_n;
        const size_t hw = (size_t)d.H * d.W, back = (size_t)g * d.group_n;
        d.wpack += (size_t)g * d.wpack_gs;
        if (d.bias) d.bias += (size_t)g * d.bias_gs;
        if (d.group_flags & RISP_GROUP_SHARED_X) d.x -= back * (d.load_mode == RISP_LOAD_CONSTCH ? d.cin_img : d.cin) * hw;
        if ((d.group_flags & RISP_GROUP_SHARED_ADD) && d.add) d.add -= back * d.add_c * hw;
    }
    return d;
}

#define RISP_CHECK_GROUP(d, name)                                                                                        \
    RISP_CHECK_ARG((d).group_n == 0 || ((d).group_n > 0 && (d).N % (d).group_n == 0 && (d).wpack_gs >= 0 && (d).bias_gs >= 0 && \
                                        (d).wpack_gs % 4 == 0),                                                           \
                   name ": grouped launch needs N %% group_n == 0 and non-negative strides, wpack_gs %% 4 == 0 (N=%d group_n=%d)", \
                   (d).N, (d).group_n)

typedef __attribute__((address_space(3))) void lptr_t;
// One LDS-DMA wave-instruction: lanes whose bit is set in `mask` copy 16 bytes from their `src` to LDS byte address
// lds_dst + 16 * lane; the others are switched off by EXEC inside the statement (no compiler branch around it, so
// every wave issues the same number of DMAs) and leave their slot untouched.  M0 and EXEC are saved and restored in
// the same statement.  Inline asm rather than __builtin_amdgcn_global_load_lds: beside the builtin hipcc turns the
// counted lgkmcnt(N) waits of the following operand reads into lgkmcnt(0).
__device__ __forceinline__ void lds_dma16(const float *src, float *dst_wave_base, unsigned long long mask) {
    const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t *)dst_wave_base);
    unsigned long long keep_exec;
    unsigned keep_m0;
    asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %4\n\ts_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %1\n\ts_mov_b64 exec, %0"
                 : "=&s"(keep_exec), "=&s"(keep_m0) : "v"(src), "s"(lds_dst), "s"(mask) : "memory");
}

// The same with a wave-uniform 64-bit base in scalar registers and one 32-bit byte offset per lane (no 64-bit address pair
// per piece in vector registers).  `mask` as in lds_dma16 (a wave issues the instruction even when no lane copies, so every
// wave of a workgroup counts the same number of transfers in vmcnt).
__device__ __forceinline__ void lds_dma16_s(const void *sbase, unsigned voff, void *dst_wave_base, unsigned long long mask = ~0ull) {
    const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t *)dst_wave_base);
    unsigned long long keep_exec;
    unsigned keep_m0;
    asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %5\n\ts_mov_b32 %1, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %1\n\ts_mov_b64 exec, %0"
                 : "=&s"(keep_exec), "=&s"(keep_m0) : "v"(voff), "s"(sbase), "s"(lds_dst), "s"(mask) : "memory");
}

// The lean form for the persistent convolution kernels: every lane copies, the LDS byte address is an integer the caller keeps in
// scalar registers (lds_addr_of(base) once, scalar adds per piece).  The pointer form above converts a generic pointer per call
// (readfirstlane pair, null check, select) and masks EXEC: ~20 dependent scalar instructions, 140 cycles per transfer measured in
// risp_conv_toep.hip - 8 % of a tile where a stage is a handful of transfers.
__device__ __forceinline__ unsigned lds_addr_of(const void *lds_ptr) {
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t *)lds_ptr);
}
__device__ __forceinline__ void lds_dma16_m(const void *sbase, unsigned voff, unsigned lds_addr) {
    unsigned keep_m0;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep_m0) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

// Workgroups per image of the element-wise BACKWARD kernels (stand-alone risp_*_bwd and the fused slot mixture share it, so
// both cut an image into the same partial sums and give the same parameter-gradient bits): >= 4 vectors per thread so the
// block reduction amortises, but enough workgroups for the chip when the batch is small (the per-GPU batch of the 8-GPU
// search is 4 images: 16 per image were 64 workgroups on 256 CUs).
inline int risp_bwd_blocks(int N, int HW) {
    const int hw4 = HW / 4;
    int bx = (hw4 + 1023) / 1024;
#ifndef RISP_BWD_WGS
#define RISP_BWD_WGS 512
#endif
    const int want = (RISP_BWD_WGS + N - 1) / (N > 0 ? N : 1);          // ~512 workgroups in flight
    if (bx < want) bx = want;
    const int most = (hw4 + 255) / 256;                        // at least one vector per thread
    if (bx > most) bx = most;
    if (bx > 64) bx = 64;
    return bx < 1 ? 1 : bx;
}

// ... of the kernels that form WbQuadratic's 30 parameter sums (stand-alone backward and the second launch of the fused slot
// backward: the same partition, so the same bits).  Their per-workgroup cost is dominated by what does not scale with the pixels -
// 30 parameter loads, a 30-value block reduction, the partial row - worth ~8 vectors of pixel work: ONE resident round of
// workgroups (512 = 2 per CU at their 2 waves per SIMD) with as many vectors per thread as that leaves, instead of 4 per thread.
inline int risp_bwd_blocks_wbq(int N, int HW) {
    const int hw4 = HW / 4;
#ifndef RISP_WBQ_WGS
#define RISP_WBQ_WGS 512
#endif
    int bx = (RISP_WBQ_WGS + N - 1) / (N > 0 ? N : 1);
    const int most = (hw4 + 255) / 256;                        // at least one vector per thread
    if (bx > most) bx = most;
    if (bx > 64) bx = 64;
    return bx < 1 ? 1 : bx;
}

// one vector (4 pixels) of a BGR image and of its upstream gradient: six 16-byte plane loads
struct BgrVec6 {
    float4 b, g, r, db, dg, dr;
    __device__ __forceinline__ void load(const float4 *xb, const float4 *gb, int hw4, int i) {
        b = xb[i]; g = xb[hw4 + i]; r = xb[2 * hw4 + i];
        db = gb[i]; dg = gb[hw4 + i]; dr = gb[2 * hw4 + i];
    }
};

// work(vector, index) over the vectors i = blockIdx.x * blockDim.x + threadIdx.x + k * gridDim.x * blockDim.x < hw4 of a thread of a
// 256-thread workgroup, k = 0, 1, .., with the loads running D vectors ahead of the work THROUGH LDS (D + 1 slots of 6 x 256 x 16
// bytes in `stage`): every wave sends the six plane rows of its 64 threads by LDS-DMA (global_load_lds_dwordx4: no registers, and
// nothing the compiler could sink behind the arithmetic - hipcc moves plain prefetch loads down to their first use and then waits
// with the counter at zero; loads in inline asm are no way out: the compiler copies and spills their registers before the data has
// landed) and reads its own 16 bytes back when the counter says the row has landed.  The pipelined part covers the K iterations
// EVERY thread of the workgroup takes part in (a uniform trip count); the ragged end (at most one more vector for some of the
// threads) follows with plain loads.  A slot is overwritten one iteration after it was read; work() may issue vector-memory
// stores (they share the in-order counter: waited for with the row, a little early).  prime() then run(work).
template <int D>
struct BgrWalkLds {
    const float4 *xb, *gb;
    float4 *stage;
    int hw4, step, tid, i0, K;
    unsigned lds0, plane;
    __device__ __forceinline__ BgrWalkLds(const float4 *xb_, const float4 *gb_, int hw4_, float4 *stage_)
        : xb(xb_), gb(gb_), stage(stage_), hw4(hw4_) {
        step = gridDim.x * blockDim.x;
        tid = threadIdx.x;
        i0 = blockIdx.x * blockDim.x + tid;
        const int lastlane = blockIdx.x * blockDim.x + blockDim.x - 1;
        K = lastlane < hw4 ? (hw4 - 1 - lastlane) / step + 1 : 0;
        lds0 = lds_addr_of(stage) + 16u * (unsigned)(__builtin_amdgcn_readfirstlane(tid) & ~63);
        plane = 16u * (unsigned)hw4;
    }
    __device__ __forceinline__ void issue(int slot, int k) const {        // (k past the last iteration: the last vector once more, unused)
        const unsigned off = 16u * (unsigned)(i0 + (k < K ? k : K - 1) * step);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            lds_dma16_m(xb, off + p * plane, lds0 + 16u * 256u * (unsigned)(slot * 6 + p));
            lds_dma16_m(gb, off + p * plane, lds0 + 16u * 256u * (unsigned)(slot * 6 + 3 + p));
        }
    }
    // the first D vectors: as early in the kernel as the pointers are known - their round trip runs beside the kernel's prologue
    // (parameter loads, coefficient arithmetic)
    __device__ __forceinline__ void prime() const {
        if (K > 0) {
#pragma unroll
            for (int j = 0; j < D; ++j) issue(j, j);
        }
    }
    template <class F>
    __device__ __forceinline__ void run(F &&work) const {
        if (K > 0) {
            int slot = 0;
            for (int k = 0; k < K; ++k) {
                issue(slot + D > D ? slot - 1 : slot + D, k + D);          // slot (k + D) mod (D + 1)
                asm volatile("s_waitcnt vmcnt(%0)" : : "n"(6 * D) : "memory");
                const float4 *sl = stage + slot * 6 * 256 + tid;
                BgrVec6 v;
                v.b = sl[0]; v.g = sl[256]; v.r = sl[512]; v.db = sl[768]; v.dg = sl[1024]; v.dr = sl[1280];
                work(v, i0 + k * step);
                slot = slot + 1 > D ? 0 : slot + 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" : : : "memory");             // the rows past the last iteration
        }
        const int it = i0 + K * step;
        if (it < hw4) {
            BgrVec6 t;
            t.load(xb, gb, hw4, it);
            work(t, it);
        }
    }
};

// sum of a wave's values, in every lane's row leader: butterflies inside the rows of 16 by DPP (no LDS round trips), then the four
// rows by readlane - a fixed order
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));       // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));       // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));      // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));      // row_mirror
    const int x = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 48)));
}

// block_sum with the wave stage on DPP: thread 0 receives the totals in v[] (waves added in index order)
template <int NV>
__device__ __forceinline__ void block_sum_dpp(float (&v)[NV], float *lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float s = wave_sum_dpp(v[i]);
        if (lane == 0) lds[i * nw + wave] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) block_sum_finish<NV>(v, lds, nw);
    __syncthreads();
}

// ... and with the totals spread over the lanes: thread j < NV returns the total of value j (0 elsewhere) - NV lanes add four
// partials each and can store their element of a row side by side, instead of thread 0 adding 4 NV partials and storing NV scalars.
// Same adds in the same order as block_sum_dpp.  256 threads.
template <int NV>
__device__ __forceinline__ float block_sum_dpp_lanes(const float (&v)[NV], float *lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float s = wave_sum_dpp(v[i]);
        if (lane == 0) lds[i * 4 + wave] = s;
    }
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x < NV) {
        const float *q = lds + 4 * threadIdx.x;
        t = (((0.f + q[0]) + q[1]) + q[2]) + q[3];
    }
    return t;
}

struct f3 {
    float b, g, r;
};
