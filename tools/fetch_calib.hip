// FETCH_SIZE calibration (GPU box, under rocprofv3 --pmc FETCH_SIZE): three streaming-read kernels that each read the
// same 1 GiB buffer (4x the Infinity Cache) exactly once with 4-, 8- and 16-byte loads per lane, plus a quad-pattern
// kernel that reads it the way bilateral_chain_kernel stages its mosaic (two 8-byte loads from adjacent rows per lane).
// bytes read / (FETCH_SIZE x 1 KiB) is the correction factor of that access width on this chip.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib tools/fetch_calib.hip && rocprofv3 --pmc FETCH_SIZE ... -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>

template <typename V>
__global__ __launch_bounds__(256) void read_kernel(const V *__restrict__ p, size_t n, float *out) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        V v = p[i];
        const float *f = reinterpret_cast<const float *>(&v);
        for (unsigned k = 0; k < sizeof(V) / 4; ++k) acc += f[k];
    }
    if (acc == 12345.678f) out[0] = acc;       // never true: keeps the loads alive
}

// rows of W floats; a lane reads the 2x2 quad at (2*qy, 2*qx) as two float2
__global__ __launch_bounds__(256) void quad_kernel(const float *__restrict__ p, int W, int H, float *out) {
    float acc = 0.f;
    const int wq = W / 2;
    const size_t nq = (size_t)wq * (H / 2);
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < nq; t += (size_t)gridDim.x * blockDim.x) {
        const size_t qy = t / wq, qx = t - qy * wq;
        const float2 a = *reinterpret_cast<const float2 *>(p + 2 * qy * W + 2 * qx);
        const float2 b = *reinterpret_cast<const float2 *>(p + (2 * qy + 1) * W + 2 * qx);
        acc += (a.x + a.y) + (b.x + b.y);
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    float *buf, *out;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, 4);
    hipMemset(buf, 0, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_kernel<float>, dim3(4096), dim3(256), 0, 0, buf, bytes / 4, out);
        hipLaunchKernelGGL(read_kernel<float2>, dim3(4096), dim3(256), 0, 0, (const float2 *)buf, bytes / 8, out);
        hipLaunchKernelGGL(read_kernel<float4>, dim3(4096), dim3(256), 0, 0, (const float4 *)buf, bytes / 16, out);
        hipLaunchKernelGGL(quad_kernel, dim3(4096), dim3(256), 0, 0, buf, 16384, 16384, out);
    }
    hipDeviceSynchronize();
    printf("fetch_calib: every kernel read %zu bytes once\n", bytes);
    return 0;
}
