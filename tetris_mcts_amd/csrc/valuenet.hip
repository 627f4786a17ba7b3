// Value network forward for gfx950 (model/model_vv.py:13-52; Model_VV.inference 210-217).
//
// Numerics contract (shared with oracle/valuenet_oracle.c): every output element is ONE fp32 fma chain,
// acc = bias; for k ascending: acc = fma(x_k, w_k, acc), with k = ci*9 + ky*3 + kx for the convolutions and
// the flat input index for the linear layers; sigmoid through tm_exp (explicit fma polynomial).  The MFMA
// kernels (v_mfma_f32_32x32x2_f32 is exactly such a k-ordered fma chain) and the plain kernels below
// therefore produce identical bits.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tetris_mcts_hip.h"

namespace tmcts_vn {

constexpr int A1 = 32 * 18 * 8, A2 = 32 * 16 * 6, A3 = 32 * 14 * 4, HID = 256;
constexpr int OFF_C1W = 0, OFF_C1B = 288, OFF_C2W = 320, OFF_C2B = 9536, OFF_C3W = 9568, OFF_C3B = 18784,
              OFF_F1W = 18816, OFF_F1B = 477568, OFF_FOW = 477824, OFF_FOB = 478336, OFF_UB = 478338, OFF_LB = 478340;

__device__ inline double tm_exp(double x) {
    if (x > 700.0) x = 700.0;
    if (x < -700.0) x = -700.0;
    const double inv_ln2 = 1.4426950408889634074, ln2_hi = 6.93147180369123816490e-01,
                 ln2_lo = 1.90821492927058770002e-10;
    double n = rint(x * inv_ln2);
    double r = fma(-n, ln2_hi, x);
    r = fma(-n, ln2_lo, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    long long bits = __double_as_longlong(p);
    bits += ((long long)n) << 52;
    return __longlong_as_double(bits);
}

// ---- reference-order plain kernels: one thread per output element ----
template <int CIN, int H, int W>
__global__ void k_conv3x3_relu(const float* __restrict__ in, int in_stride, const int8_t* __restrict__ in8,
                               const float* __restrict__ wt, const float* __restrict__ bias, float* __restrict__ out,
                               int out_stride, int n) {
    constexpr int OH = H - 2, OW = W - 2, PER = 32 * OH * OW;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * PER) return;
    int s = t / PER, e = t - s * PER;
    int co = e / (OH * OW), p = e - co * (OH * OW), y = p / OW, x = p - y * OW;
    float acc = bias[co];
    for (int ci = 0; ci < CIN; ++ci)
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                int ii = (ci * H + y + ky) * W + x + kx;
                float xv = in8 ? (float)in8[(size_t)s * 200 + ii] : in[(size_t)s * in_stride + ii];
                acc = fmaf(xv, wt[((co * CIN + ci) * 3 + ky) * 3 + kx], acc);
            }
    out[(size_t)s * out_stride + e] = acc > 0.0f ? acc : 0.0f;
}

__global__ void k_fc1_relu(const float* __restrict__ a3, int stride, const float* __restrict__ w,
                           const float* __restrict__ b, float* __restrict__ h, int hstride, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * HID) return;
    int s = t / HID, j = t - s * HID;
    float acc = b[j];
    const float* x = a3 + (size_t)s * stride;
    const float* wr = w + (size_t)j * A3;
    for (int i = 0; i < A3; ++i) acc = fmaf(x[i], wr[i], acc);
    h[(size_t)s * hstride + j] = acc > 0.0f ? acc : 0.0f;
}

__global__ void k_fc_out(const float* __restrict__ h, int hstride, const float* __restrict__ P, float* __restrict__ v,
                         float* __restrict__ var, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * 2) return;
    int s = t >> 1, j = t & 1;
    float acc = P[OFF_FOB + j];
    const float* x = h + (size_t)s * hstride;
    const float* wr = P + OFF_FOW + j * HID;
    for (int i = 0; i < HID; ++i) acc = fmaf(x[i], wr[i], acc);
    double e = tm_exp(-(double)acc);
    float sg = (float)(1.0 / (1.0 + e));
    float tt = sg * P[OFF_UB + j];
    float o = tt + P[OFF_LB + j];
    if (j == 0) v[s] = o; else var[s] = o;
}

}  // namespace tmcts_vn

using namespace tmcts_vn;

extern "C" int tm_valuenet_forward(const float* P, const int8_t* states, int n, float* v, float* var, float* scratch,
                                   void* stream_) {
    // scratch: n x TM_VALUENET_SCRATCH floats
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0) return 0;
    constexpr int SS = TM_VALUENET_SCRATCH;
    float* a1 = scratch;            // [n][SS]: a1 at 0, a2 at A1, a3 at A1+A2, h at A1+A2+A3
    const int T = 256;
    hipLaunchKernelGGL((k_conv3x3_relu<1, 20, 10>), dim3((n * A1 + T - 1) / T), dim3(T), 0, stream, (const float*)nullptr,
                       0, states, P + OFF_C1W, P + OFF_C1B, a1, SS, n);
    hipLaunchKernelGGL((k_conv3x3_relu<32, 18, 8>), dim3((n * A2 + T - 1) / T), dim3(T), 0, stream, a1, SS,
                       (const int8_t*)nullptr, P + OFF_C2W, P + OFF_C2B, a1 + A1, SS, n);
    hipLaunchKernelGGL((k_conv3x3_relu<32, 16, 6>), dim3((n * A3 + T - 1) / T), dim3(T), 0, stream, a1 + A1, SS,
                       (const int8_t*)nullptr, P + OFF_C3W, P + OFF_C3B, a1 + A1 + A2, SS, n);
    hipLaunchKernelGGL(k_fc1_relu, dim3((n * HID + T - 1) / T), dim3(T), 0, stream, a1 + A1 + A2, SS, P + OFF_F1W,
                       P + OFF_F1B, a1 + A1 + A2 + A3, SS, n);
    hipLaunchKernelGGL(k_fc_out, dim3((n * 2 + T - 1) / T), dim3(T), 0, stream, a1 + A1 + A2 + A3, SS, P, v, var, n);
    return (int)hipGetLastError();
}
