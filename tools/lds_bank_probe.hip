// LDS bank-conflict probe for the access patterns of risp_conv_f16x2.hip (run under rocprofv3 --pmc SQ_LDS_BANK_CONFLICT
// SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS; one kernel name per pattern).  hipcc -O3 --offload-arch=gfx950 tools/lds_bank_probe.hip -o /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int S = 17, RS = 68, IH = 10, PART = 2 * IH * RS;
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define PROBE(name, ADDR_EXPR, KIND)                                                               \
    __global__ __launch_bounds__(256) void name(float *out) {                                      \
        extern __shared__ __attribute__((aligned(16))) uint4 smem[];                               \
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hl = lane >> 5; \
        (void)wave; (void)l31; (void)hl;                                                           \
        for (int i = tid; i < 4096; i += 256) smem[i] = make_uint4(i, i, i, i);                   \
        __syncthreads();                                                                           \
        unsigned acc = 0;                                                                          \
        const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void *)smem;    \
        for (int it = 0; it < 1024; ++it) {                                                        \
            const unsigned off = base + (unsigned)(ADDR_EXPR);                                     \
            if (KIND == 0) { u4 v; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(off)); acc += v.x; } \
            else if (KIND == 1) { u4 v = {acc, 1u, 2u, 3u}; asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(off), "v"(v)); } \
            else { asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(off), "v"(acc)); } \
        }                                                                                          \
        out[blockIdx.x * 256 + tid] = (float)acc;                                                  \
    }
// B operand read: lane (n = l31, hl): row 2 wave + (n >> 4), slot (n & 15) + shift
PROBE(read_b128_B_operand, 16 * (((hl * IH + 2 * wave + (l31 >> 4)) * RS + (l31 & 15) + (it & 1) * S + (it & 7))), 0)
// B operand read variants: row stride RSV slots, channel-half offset HV slots
#define BVAR(name, RSV, HV) PROBE(name, 16 * ((hl * (HV) + (2 * wave + (l31 >> 4)) * (RSV) + (l31 & 15) + (it & 1) * S + (it & 7))), 0)
BVAR(readB_rs64_h640, 64, 640)
BVAR(readB_rs72_h720, 72, 720)
BVAR(readB_rs68_h688, 68, 688)
BVAR(readB_rs68_h696, 68, 696)
BVAR(readB_rs76_h760, 76, 760)
BVAR(readB_rs80_h800, 80, 800)
BVAR(readB_rs72_h728, 72, 728)
BVAR(readB_rs72_h736, 72, 736)
// only two distinct runs: lanes 0-15 | 16-31 one row apart, upper half-wave the same addresses as the lower
PROBE(readB_rows_only_rs68, 16 * (((l31 >> 4)) * 68 + (l31 & 15) + (it & 7)), 0)
PROBE(readB_rows_only_rs72, 16 * (((l31 >> 4)) * 72 + (l31 & 15) + (it & 7)), 0)
PROBE(readB_rows_only_rs64, 16 * (((l31 >> 4)) * 64 + (l31 & 15) + (it & 7)), 0)
PROBE(readB_half_only_h680, 16 * (hl * 680 + l31 + (it & 7)), 0)
PROBE(readB_half_only_h688, 16 * (hl * 688 + l31 + (it & 7)), 0)
// A operand read: lane (m = l31, hl): slot hl * 64 + l31 of a row
PROBE(read_b128_A_operand, 16 * (2720 + (it % 12) * 128 + hl * 64 + l31), 0)
// A operand read with the upper half rotated by 8 slots
PROBE(read_b128_A_rotated, 16 * (2720 + (it % 12) * 128 + hl * 64 + ((l31 + 8 * hl) & 63)), 0)
// all 64 lanes consecutive slots (the fp32 Winograd kernels' pattern)
PROBE(read_b128_consecutive, 16 * (lane + (it & 63)), 0)
// staging write: thread (g = tid >> 7, r = (tid >> 4) & 7, q = tid & 15): slot ((g IH + r + 1) RS + (c & 3) S + (c >> 2)), c = 4 q + j + 1
PROBE(write_b128_put_quad, 16 * ((((tid >> 7) * IH + ((tid >> 4) & 7) + 1) * RS) + (((4 * (tid & 15) + (it & 3) + 1) & 3) * S) + ((4 * (tid & 15) + (it & 3) + 1) >> 2)) + (it & 4 ? PART * 16 : 0), 1)
// halo-row write: thread (q = tid & 15, hr = (tid >> 4) & 1, cp = tid >> 5): 4 bytes at slot of column 4 q + j + 1, row 0 / 9
PROBE(write_b32_put_pair, ((((tid >> 5) >> 2) * IH + (((tid >> 4) & 1) ? 9 : 0)) * RS + (((4 * (tid & 15) + (it & 3) + 1) & 3) * S) + ((4 * (tid & 15) + (it & 3) + 1) >> 2)) * 16 + ((tid >> 5) & 3) * 4, 2)
int main() {
    float *out;
    (void)hipMalloc(&out, 1024 * 256 * 4);
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(read_b128_B_operand, dim3(1024), dim3(256), 65536, 0, out);
        hipLaunchKernelGGL(read_b128_A_operand, dim3(1024), dim3(256), 65536, 0, out);
#define RUN(k) hipLaunchKernelGGL(k, dim3(1024), dim3(256), 65536, 0, out)
        RUN(readB_rs64_h640); RUN(readB_rs72_h720); RUN(readB_rs68_h688); RUN(readB_rs68_h696); RUN(readB_rs76_h760); RUN(readB_rs80_h800);
        RUN(readB_rs72_h728); RUN(readB_rs72_h736); RUN(readB_rows_only_rs68); RUN(readB_rows_only_rs72); RUN(readB_rows_only_rs64);
        RUN(readB_half_only_h680); RUN(readB_half_only_h688);
        hipLaunchKernelGGL(read_b128_A_rotated, dim3(1024), dim3(256), 65536, 0, out);
        hipLaunchKernelGGL(read_b128_consecutive, dim3(1024), dim3(256), 65536, 0, out);
        hipLaunchKernelGGL(write_b128_put_quad, dim3(1024), dim3(256), 65536, 0, out);
        hipLaunchKernelGGL(write_b32_put_pair, dim3(1024), dim3(256), 65536, 0, out);
    }
    (void)hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
