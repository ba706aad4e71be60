// Shared pieces of the split-precision (two f16 halves per fp32 operand) convolution kernels: risp_conv_f16x2.hip (channels in
// the reduction index) and risp_conv_toep.hip (a row of taps in the reduction index).
#pragma once
#include "risp_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Buffer addressing (a 128-bit resource in scalar registers + one 32-bit byte offset per lane + a scalar byte offset): a tensor
// plane costs a scalar add instead of a 64-bit address pair per lane and access.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t h2_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 h2_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float h2_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// largest of a wave's non-negative values, in every lane: butterflies inside the rows of 16 by DPP, then the four rows by
// readlane (non-negative floats order like their bit patterns) - the shuffle form (six ds_bpermute round trips) sat on every
// chunk's critical path
__device__ __forceinline__ float h2_wave_max(float v) {
    int x = __builtin_bit_cast(int, v);
    x = max(x, __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
    x = max(x, __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
    x = max(x, __builtin_amdgcn_mov_dpp(x, 0x141, 0xF, 0xF, true));     // row_half_mirror
    x = max(x, __builtin_amdgcn_mov_dpp(x, 0x140, 0xF, 0xF, true));     // row_mirror
    const int r = max(max(__builtin_amdgcn_readlane(x, 0), __builtin_amdgcn_readlane(x, 16)),
                      max(__builtin_amdgcn_readlane(x, 32), __builtin_amdgcn_readlane(x, 48)));
    return __builtin_bit_cast(float, r);
}

// sum of a wave's values in a fixed order (the same butterflies by DPP, then the four rows): deterministic
__device__ __forceinline__ float h2_wave_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    const int x = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 48)));
}

__device__ __forceinline__ float amax4(float m, const float4 &v) {
    return fmaxf(fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fabsf(v.z))), fabsf(v.w));
}

__device__ __forceinline__ float comp(const float4 &v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }

// hi / lo halves of 8 scaled values -> two 16-byte slots
__device__ __forceinline__ void split8(const float (&a)[8], float s, uint4 &hi, uint4 &lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a0 = a[2 * e] * s, a1 = a[2 * e + 1] * s;
        const h2 hh = {(_Float16)a0, (_Float16)a1};
        const h2 ll = {(_Float16)(a0 - (float)hh[0]), (_Float16)(a1 - (float)hh[1])};
        h[e] = __builtin_bit_cast(unsigned, hh);
        l[e] = __builtin_bit_cast(unsigned, ll);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

#define H2_WAIT_VM(keep) __builtin_amdgcn_s_waitcnt(0x0F70 | ((keep) & 15) | (((keep) >> 4) << 14))      /* s_waitcnt vmcnt(keep) */

inline int h2_cu_count() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return cus;
}

