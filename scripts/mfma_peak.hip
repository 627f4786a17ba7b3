// Achievable fp32 matrix rate of this GPU: every wave issues independent v_mfma_f32_32x32x2_f32 back to back.
// hipcc --offload-arch=gfx950 -O3 scripts/mfma_peak.hip -o variants/mfma_peak && variants/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = (float)(threadIdx.x + c);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS>
void run(int wgs, const char* tag) {
    float* out;
    (void)hipMalloc(&out, (size_t)wgs * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    k<CHAINS><<<wgs, 256>>>(out, 1000, 1.0f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<CHAINS><<<wgs, 256>>>(out, iters, 1.0f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)wgs * 4 * iters * CHAINS * 4096.0;
    printf("%s: %d WGs x 4 waves, %d chains: %.3f ms, %.1f TFLOP/s, implied clock at 64 cyc/MFMA %.3f GHz\n", tag, wgs, CHAINS, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 2.4);
    (void)hipFree(out);
}
int main() {
    run<4>(256, "1 wave/SIMD");
    run<4>(512, "2 waves/SIMD");
    run<2>(512, "2 waves/SIMD");
    run<1>(1024, "4 waves/SIMD, 1 chain");
    return 0;
}
