// Developer probe: issue rate of v_pk_fma_f32 against v_fma_f32 on gfx950 (same flops, same dependency depth).
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/pk_probe tools/probes/pk_probe.hip && gpurun_out/pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_scalar(float* out, float a, float b, int iters)
{
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (float)threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_packed(float* out, float a, float b, int iters)
{
    f2 x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = f2{(float)threadIdx.x + 2 * i, (float)threadIdx.x + 2 * i + 1};
    f2 av = {a, a}, bv = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(av), "v"(bv));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    float* d;
    hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2048;
    for (int blocks : {256, 1024, 2048, 4096}) {
        for (int which = 0; which < 2; ++which) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (which == 0) k_scalar<<<blocks, 256>>>(d, 1.0001f, 0.5f, iters);
                else k_packed<<<blocks, 256>>>(d, 1.0001f, 0.5f, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep == 2) {
                    double flops = (double)blocks * 256 * iters * 64 * 2;
                    printf("%s blocks %4d  %.3f ms  %.1f TFLOP/s\n", which ? "v_pk_fma_f32" : "v_fma_f32   ", blocks, ms, flops / ms * 1e-9);
                }
            }
        }
    }
    return 0;
}
