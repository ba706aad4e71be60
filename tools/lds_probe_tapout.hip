// LDS bank-conflict probe for the tile of risp_conv_tapout.hip: the producers' 16-byte slot writes (a lane = 4 consecutive pixels of 8
// channels: one slot per pixel) and the consumers' 16-byte operand reads (32 consecutive columns per lane half, shifted by the filter
// column).  Layouts: `lin` = slot of column c is c (round 6's first form), `q<SQ>` = slot (c & 3) * SQ + (c >> 2) (the quarter
// interleave of risp_conv_f16x2_ws.hip: staging lanes write consecutive slots).  Run under rocprofv3 --pmc SQ_LDS_BANK_CONFLICT
// SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS (tools/lds_probe_tapout.sh); one kernel name per pattern.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int RS = 4;
#define PROBE(name, ADDR_EXPR, KIND)                                                               \
    __global__ __launch_bounds__(256) void name(float *out) {                                      \
        extern __shared__ __attribute__((aligned(16))) uint4 smem[];                               \
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hl = lane >> 5; \
        (void)wave; (void)l31; (void)hl;                                                           \
        for (int i = tid; i < 8192; i += 256) smem[i] = make_uint4(i, i, i, i);                   \
        __syncthreads();                                                                           \
        unsigned acc = 0;                                                                          \
        const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void *)smem;    \
        for (int it = 0; it < 1024; ++it) {                                                        \
            const unsigned off = base + 16u * (unsigned)(ADDR_EXPR);                               \
            if (KIND == 0) { u4 v; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(off)); acc += v.x; } \
            else { u4 v = {acc, 1u, 2u, 3u}; asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(off), "v"(v)); } \
        }                                                                                          \
        out[blockIdx.x * 256 + tid] = (float)acc;                                                  \
    }
// column of a producer lane's i-th pixel / of a consumer lane at filter column kx (P = 4: the 9x9 layer; it & 3 = the pixel, it % 9 = kx)
#define WCOL (4 + 4 * l31 + (it & 3))
#define RCOL (32 * wave + l31 + (it % 9))
#define QSLOT(c, SQ) ((((c) & 3) * (SQ)) + ((c) >> 2))
// row of 136 columns; the upper lane half is the other channel half, RS rows further; producer wave = row
PROBE(write_lin, (hl * RS + wave) * 136 + WCOL, 1)
PROBE(read_lin, (hl * RS + (it & 3)) * 136 + RCOL, 0)
#define LAYOUT(SQ)                                                                                  \
    PROBE(write_q##SQ, (hl * RS + wave) * (4 * SQ) + QSLOT(WCOL, SQ), 1)                            \
    PROBE(read_q##SQ, (hl * RS + (it & 3)) * (4 * SQ) + QSLOT(RCOL, SQ), 0)
LAYOUT(34) LAYOUT(35) LAYOUT(36) LAYOUT(37) LAYOUT(38) LAYOUT(40) LAYOUT(44) LAYOUT(48)
int main() {
    float *out;
    (void)hipMalloc(&out, 1024 * 256 * 4);
#define RUN(k) hipLaunchKernelGGL(k, dim3(1024), dim3(256), 131072, 0, out)
#define RUNL(SQ) RUN(write_q##SQ); RUN(read_q##SQ)
    for (int r = 0; r < 2; ++r) {
        RUN(write_lin); RUN(read_lin);
        RUNL(34); RUNL(35); RUNL(36); RUNL(37); RUNL(38); RUNL(40); RUNL(44); RUNL(48);
    }
    (void)hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
