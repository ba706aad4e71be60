// What read bandwidth does a walk over P channel planes deliver?  Every wave walks down a segment of rows of a 256 x 256 image strip and,
// per row, loads its columns of P planes (planes 256 KB apart, as the NCHW tensors of the convolution kernels are) - 4 bytes per lane (32
// columns, two planes per instruction) or 16 bytes per lane.  Three forms:
//   probe         `acc += x[..]` in the unrolled loop: hipcc keeps the additions in order and with them the loads - ONE load in flight per
//                 wave, ~2.5 loads per microsecond and wave whatever P; 64 planes from 512 workgroups: 1.2 TB/s (NOT a property of the memory)
//   probe_chunked the same with the planes in chunks of 16 (rows inside): the same
//   probe_deep    a row's 32 loads (64 planes) issued before the first is consumed: 5.1 TB/s from the same 512 workgroups, 5.8 from 2048
// => a 64-plane walk is not HBM-bound at the 2.6-2.9 TB/s the few-channel kernels read at (tap-row, narrow 3x3, vector kernel): their
// waves spend the row period in their own LDS round trips and dependent matrix instructions (NOTES.md, round 6).
// hipcc -O3 --offload-arch=gfx950 tools/plane_read_probe.hip -o /tmp/plane_read_probe && /tmp/plane_read_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int P, int WIDE, int SEG>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ x, float *__restrict__ out, int H, int W) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x;
    const int segs = H / SEG, strips = WIDE ? W / 512 + (W % 512 != 0) : W / 128;
    const int sg = t % segs; t /= segs;
    const int st = t % strips, n = t / strips;
    const size_t hw = (size_t)H * W;
    const float *base = x + (size_t)n * P * hw;
    float acc = 0.f;
    if (WIDE) {
        const int cx = st * 512 + 128 * wave + 4 * (lane & 31);
        if (cx >= W) return;
        const int ph = lane >> 5;                       // two planes per instruction as well: lanes 32-63 take plane p + P / 2
        for (int y = sg * SEG; y < sg * SEG + SEG; ++y)
#pragma unroll
            for (int p = 0; p < (P + 1) / 2; ++p) {
                const int pl = P == 1 ? 0 : p + ph * (P / 2);
                const float4 v = *reinterpret_cast<const float4 *>(base + (size_t)pl * hw + (size_t)y * W + cx);
                acc += (v.x + v.y) + (v.z + v.w);
            }
    } else {
        const int cx = st * 128 + 32 * wave + (lane & 31), ph = lane >> 5;
        for (int y = sg * SEG; y < sg * SEG + SEG; ++y)
#pragma unroll
            for (int p = 0; p < (P + 1) / 2; ++p) {
                const int pl = P == 1 ? 0 : p + ph * (P / 2);
                acc += base[(size_t)pl * hw + (size_t)y * W + cx];
            }
    }
    if (acc == 123.456f) out[0] = acc;
}
// 64 planes, DEPTH rows requested before the first is consumed (registers: 32 x DEPTH values per lane)
template <int DEPTH, int SEG>
__global__ __launch_bounds__(256) void probe_deep(const float *__restrict__ x, float *__restrict__ out, int H, int W) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x;
    const int segs = H / SEG, strips = W / 128;
    const int sg = t % segs; t /= segs;
    const int st = t % strips, n = t / strips;
    const size_t hw = (size_t)H * W;
    const float *base = x + (size_t)n * 64 * hw + (size_t)(lane >> 5) * 32 * hw + st * 128 + 32 * wave + (lane & 31);
    float acc = 0.f, v[DEPTH][32];
    for (int y = sg * SEG; y < sg * SEG + SEG; y += DEPTH) {
#pragma unroll
        for (int dd = 0; dd < DEPTH; ++dd)
#pragma unroll
            for (int p = 0; p < 32; ++p) v[dd][p] = base[(size_t)p * hw + (size_t)(y + dd) * W];
#pragma unroll
        for (int dd = 0; dd < DEPTH; ++dd)
#pragma unroll
            for (int p = 0; p < 32; ++p) acc += v[dd][p];
    }
    if (acc == 123.456f) out[0] = acc;
}
template <int DEPTH, int SEG>
void run_deep(const float *x, float *out, int N, int H, int W) {
    const int grid = N * (W / 128) * (H / SEG);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe_deep<DEPTH, SEG>), dim3(grid), dim3(256), 0, 0, x, out, H, W);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((probe_deep<DEPTH, SEG>), dim3(grid), dim3(256), 0, 0, x, out, H, W);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("64 planes, %d rows requested together, 4 bytes per lane, %4d images of %d x %d, %2d-row segments (%5d workgroups): %7.1f us per launch, %6.0f GB/s\n",
           DEPTH, N, H, W, SEG, grid, ms * 100, (double)N * 64 * H * W * 4 / (ms / 10 * 1e-3) / 1e9);
}
// the same bytes with the planes walked in chunks of 16: all rows of the segment for planes 0 .. 15, then for 16 .. 31, ...
template <int P, int SEG>
__global__ __launch_bounds__(256) void probe_chunked(const float *__restrict__ x, float *__restrict__ out, int H, int W) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x;
    const int segs = H / SEG, strips = W / 128;
    const int sg = t % segs; t /= segs;
    const int st = t % strips, n = t / strips;
    const size_t hw = (size_t)H * W;
    const float *base = x + (size_t)n * P * hw;
    float acc = 0.f;
    const int cx = st * 128 + 32 * wave + (lane & 31), ph = lane >> 5;
    for (int c = 0; c < P / 16; ++c)
        for (int y = sg * SEG; y < sg * SEG + SEG; ++y)
#pragma unroll
            for (int p = 0; p < 8; ++p) acc += base[(size_t)(16 * c + p + 8 * ph) * hw + (size_t)y * W + cx];
    if (acc == 123.456f) out[0] = acc;
}
template <int P, int SEG>
void run_chunked(const float *x, float *out, int N, int H, int W) {
    const int grid = N * (W / 128) * (H / SEG);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe_chunked<P, SEG>), dim3(grid), dim3(256), 0, 0, x, out, H, W);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((probe_chunked<P, SEG>), dim3(grid), dim3(256), 0, 0, x, out, H, W);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%2d planes in chunks of 16 (rows inside), 4 bytes per lane, %4d images of %d x %d, %2d-row segments (%5d workgroups): %7.1f us per launch, %6.0f GB/s\n",
           P, N, H, W, SEG, grid, ms * 100, (double)N * P * H * W * 4 / (ms / 10 * 1e-3) / 1e9);
}
template <int P, int WIDE, int SEG = 32>
void run(const float *x, float *out, int N, int H, int W) {
    const int strips = WIDE ? (W + 511) / 512 : W / 128, grid = N * strips * (H / SEG);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<P, WIDE, SEG>), dim3(grid), dim3(256), 0, 0, x, out, H, W);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((probe<P, WIDE, SEG>), dim3(grid), dim3(256), 0, 0, x, out, H, W);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)N * P * H * W * 4 * (WIDE && P == 1 ? 0.5 : 1.0) * (P == 1 ? 1.0 : 1.0);
    printf("%2d planes, %2d bytes per lane, %4d images of %d x %d, %2d-row segments (%5d workgroups): %7.1f us per launch, %6.0f GB/s\n", P, WIDE ? 16 : 4, N, H, W, SEG, grid, ms * 100, bytes / (ms / 10 * 1e-3) / 1e9);
}
int main() {
    const int H = 256, W = 256;
    float *x, *out;
    const size_t total = (size_t)2048 * H * W;                 // 512 MB: N x P = 2048 planes for every P
    hipMalloc(&x, total * 4); hipMalloc(&out, 64);
    hipMemset(x, 0, total * 4);
    run<2, 0>(x, out, 1024, H, W);  run<4, 0>(x, out, 512, H, W);  run<16, 0>(x, out, 128, H, W); run<64, 0>(x, out, 32, H, W);
    run<2, 1>(x, out, 1024, H, W);  run<4, 1>(x, out, 512, H, W);  run<16, 1>(x, out, 128, H, W); run<64, 1>(x, out, 32, H, W);
    // the same bytes per wave and row with more, shorter walks (more workgroups), and with planes that are no power of two
    run<64, 0, 8>(x, out, 32, H, W); run<64, 0, 4>(x, out, 32, H, W); run<16, 0, 8>(x, out, 128, H, W);
    run<2, 0, 32>(x, out, 32, H, W); run<16, 0, 32>(x, out, 32, H, W);          // few workgroups with few planes
    run_deep<1, 32>(x, out, 32, H, W); run_deep<2, 32>(x, out, 32, H, W); run_deep<4, 32>(x, out, 32, H, W); run_deep<2, 8>(x, out, 32, H, W); run_deep<4, 8>(x, out, 32, H, W);
    run_chunked<64, 32>(x, out, 32, H, W); run_chunked<64, 8>(x, out, 32, H, W);
    run<64, 0>(x, out, 21, H, 384); run<16, 0>(x, out, 85, H, 384); run<64, 0, 8>(x, out, 21, H, 384);
    return 0;
}
