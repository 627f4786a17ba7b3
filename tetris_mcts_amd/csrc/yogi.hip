// The Yogi step of the online fit as ONE kernel over a flat parameter / gradient / moment buffer (model/yogi.py:39-90).
//
// The reference's step is, per parameter tensor, eight elementwise tensor operations (yogi.py:69-88); on the device that is
// 12 parameters x 8 launches per iteration of a 478 342-parameter net on a batch of 1 024 - the fit was launch-bound (r05:
// 2.3 ms an iteration, 57 % of the online loop's wall-clock).  Here: the same operations per element in the same order, fp32
// where the reference's tensor operations are fp32 (a Python scalar enters a float32 tensor operation as a float), no
// contraction (the library is built with -ffp-contract=off), one launch; the step count and the powers of the betas live on
// the device (advanced by a one-thread kernel in front), so that an iteration is the same sequence of launches every time and
// can be replayed from a HIP graph (tetris_mcts_amd/train.py).
//   state (doubles): [0] step t, [1] beta1^t, [2] beta2^t, [3] lr / (1 - beta1^t), [4] sqrt(1 - beta2^t)
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/tetris_mcts_hip.h"

namespace tmcts {

__global__ void k_yogi_tick(double* st, double lr, double b1, double b2) {
    const double t = st[0] + 1.0;
    const double p1 = t == 1.0 ? b1 : st[1] * b1, p2 = t == 1.0 ? b2 : st[2] * b2;
    st[0] = t; st[1] = p1; st[2] = p2;
    st[3] = lr / (1.0 - p1);            // step_size = lr / bias_correction1              (yogi.py:84-85)
    st[4] = sqrt(1.0 - p2);             // math.sqrt(bias_correction2)                     (yogi.py:82)
}

__global__ __launch_bounds__(256) void k_yogi_step(float* __restrict__ p, const float* __restrict__ g_, float* __restrict__ m,
                                                   float* __restrict__ v, const double* __restrict__ st, int n, float b1, float omb1,
                                                   float nomb2, float eps, float wd) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const bool first = st[0] == 1.0;
    const float step = (float)st[3], bc2 = (float)st[4];
    float g = g_[i];
    const float pi = p[i];
    float mi = first ? 0.f : m[i];                        // exp_avg = zeros, exp_avg_sq = grad * grad       (yogi.py:62-66)
    float vi = first ? g * g : v[i];
    if (wd != 0.f) g = g + wd * pi;                       // grad.add(p, alpha=weight_decay)                 (yogi.py:69-70)
    mi = mi * b1;                                         // exp_avg.mul_(beta1)
    mi = mi + omb1 * g;                                   //        .add_(grad, alpha=1 - beta1)             (yogi.py:73)
    const float g2 = g * g;                               // grad.mul(grad)
    const float d = vi - g2;
    const float sg = d > 0.f ? 1.f : d < 0.f ? -1.f : d;  // torch.sign (0 stays 0, NaN stays NaN)
    vi = vi + (nomb2 * sg) * g2;                          // exp_avg_sq.addcmul_(sign, grad_squared, value=-(1 - beta2))   (yogi.py:75-78)
    const float den = sqrtf(vi) / bc2 + eps;              // (exp_avg_sq.sqrt() / math.sqrt(bias_correction2)).add_(eps)  (yogi.py:82)
    p[i] = pi + (-step * mi) / den;                       // p.addcdiv_(exp_avg, denom, value=-step_size)    (yogi.py:87)
    m[i] = mi;
    v[i] = vi;
}

}  // namespace tmcts

extern "C" int tm_yogi_step(float* p, const float* g, float* m, float* v, double* state, int n, double lr, double beta1, double beta2,
                            double eps, double weight_decay, void* stream) {
    if (!p || !g || !m || !v || !state || n < 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(tmcts::k_yogi_tick, dim3(1), dim3(1), 0, (hipStream_t)stream, state, lr, beta1, beta2);
    if (n > 0)
        hipLaunchKernelGGL(tmcts::k_yogi_step, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, g, m, v, state, n,
                           (float)beta1, (float)(1.0 - beta1), (float)(-(1.0 - beta2)), (float)eps, (float)weight_decay);      // (Python scalars: doubles, float32 in the tensor operation)
    return (int)hipGetLastError();
}
