// Distributional value head for gfx950 (model/model_distributional.py:18-57 `Net`; Model_Dist.inference 100-107): the leaf
// evaluator of DistValueSim (BASELINE configs[4]).
//   input 22 x 10 (the 20 visible rows under two empty ones: the reference's net is hard-wired to 22 rows, :27)
//   conv 4x4 (1 -> 32) + LeakyReLU -> conv 4x4 (32 -> 32) + LeakyReLU -> flatten 2048 -> FC 128 + LeakyReLU -> FC atoms -> softmax
//
// Numerics contract (shared with oracle/distnet_oracle.c): every pre-activation is ONE fp32 fma chain, acc = bias; for k
// ascending: acc = fma(x_k, w_k, acc), k = ci*16 + ky*4 + kx for the convolutions and the flat input index for the linear
// layers - exactly what v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 compute per output element; LeakyReLU = x > 0 ? x :
// x * 0.01f; softmax: m = max logit, e_b = tm_exp((double)x_b - (double)m), sum over b ascending in double, p_b =
// (float)(e_b / sum).  Held to 1e-6 relative against the reference's own Net (tests/golden/ref_distnet.npz).
//
// Two kernels per evaluation of all pending leaves:
//   k_dn_conv : one wave per state (render + conv1 + conv2, activations in LDS, conv2 = 2 position tiles x 256 MFMA steps),
//               two workgroups of four waves per CU; 4096 states = exactly four per SIMD
//   k_dn_fc   : 16 states x all 128 hidden units per workgroup (8 waves, one 16 x 16 tile each, K = 2048 staged through LDS),
//               then FC atoms on the matrix cores and the softmax in the same workgroup - no hand-off between workgroups
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>
#include "../../include/tetris_mcts_hip.h"
#include "engine.h"

namespace tmcts_dn {

constexpr int C1P = 19 * 7, C2P = 16 * 4, A2 = 32 * C2P, HID = 128, KFC = A2;
constexpr int OFF_C1W = 0, OFF_C1B = 512, OFF_C2W = 544, OFF_C2B = 16928, OFF_F1W = 16960, OFF_F1B = 279104, OFF_FVW = 279232;
constexpr int PREP_W2 = 0, PREP_W1 = 16384, PREP_TOTAL = 16384 + HID * KFC;
static_assert(PREP_TOTAL == TM_DISTNET_PREPARED, "prepared stream size");
static_assert(OFF_FVW + 50 * HID + 50 == TM_DISTNET_PARAMS_50, "parameter blob size at 50 atoms");
constexpr int A1CS = C1P;                         // conv1-output channel stride in LDS
constexpr int WAVE_LDS = (32 * A1CS + 220 + 3) & ~3;   // floats per wave: a1, the 22 x 10 input

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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

__device__ __forceinline__ float leaky(float v) { return v > 0.0f ? v : v * 0.01f; }

// Operand streams ("T4", as valuenet.hip): for MFMA step s the A operand of lane l is W[row][k(s, l)]; four consecutive steps
// are stored together so one 16-byte load per lane feeds four MFMAs: T4[(s/4)*64 + l][s%4].
//   conv2 (32x32x2): row = l & 31, k = 2 s + (l >> 5), 256 steps
//   fc1   (16x16x4): per 16-row hidden tile ht: row = 16 ht + (l & 15), k = 4 s + (l >> 4), 512 steps
__global__ void k_dn_prepare(const float* __restrict__ P, float* __restrict__ prep) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < PREP_W1) {
        int q = t / 256, l = (t / 4) % 64, r = t % 4;
        int s = 4 * q + r;
        int k = 2 * s + (l >> 5), row = l & 31;
        prep[t] = P[OFF_C2W + row * 512 + k];
    } else if (t < PREP_TOTAL) {
        int e = t - PREP_W1;
        int ht = e / (128 * 256), e2 = e % (128 * 256);
        int q = e2 / 256, l = (e2 / 4) % 64, r = e2 % 4;
        int s = 4 * q + r;
        int k = 4 * s + (l >> 4), row = 16 * ht + (l & 15);
        prep[t] = P[OFF_F1W + (size_t)row * KFC + k];
    }
}

__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// conv2: 4x4 valid convolution 32 -> 32 channels over the 19 x 7 map = 64 output positions = two tiles of 32 on the matrix
// cores; K = 512 = 256 steps of two taps, 8 steps per input channel.  in: LDS activations [32][A1CS]; boff[t][j]: this lane's
// LDS float offset for tile t and step j of an input channel; W: prepared T4 stream (+lane).
// Software pipeline as valuenet.hip's conv_mfma: the B operands (LDS) of quad q+1 and the weights (global) of quad q+2 are
// requested while the 8 MFMAs of quad q issue.
constexpr int NQ2 = 64;
__device__ __forceinline__ void conv2_mfma(const float* __restrict__ in, const int (&boff)[2][8],
                                           const float4* __restrict__ W, f32x16 (&acc)[2]) {
    float4 wq[3];
    float bb[2][4][2];
    auto load_b = [&](int q, int buf) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = 4 * q + r, ci = s / 8, j = s % 8;
#pragma unroll
            for (int t = 0; t < 2; ++t) bb[buf][r][t] = in[boff[t][j] + ci * A1CS];
        }
    };
    wq[0] = W[0];
    wq[1] = W[64];
    load_b(0, 0);
#pragma unroll
    for (int q = 0; q < NQ2; ++q) {
        if (q + 2 < NQ2) wq[(q + 2) % 3] = W[(q + 2) * 64];
        if (q + 1 < NQ2) load_b(q + 1, (q + 1) & 1);
        const float4 w4 = wq[q % 3];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = (r == 0) ? w4.x : (r == 1) ? w4.y : (r == 2) ? w4.z : w4.w;
#pragma unroll
            for (int t = 0; t < 2; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb[q & 1][r][t], acc[t], 0, 0, 0);
        }
        // issue order inside the quad: the weight load up front, one LDS read behind every MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (q % 6 == 5) __builtin_amdgcn_sched_barrier(0);   // bounded scheduling regions (compile time)
    }
}

// Input: either int8 states [n][200] (0 empty, 1 locked, -1 falling piece), or (states == nullptr) the tree engine's
// evaluation requests: request s names NODE eval_obs[s] of game s (TM_KIND_DIST: no observation projection), rendered here
// from the node's packed game (ENGINE_SPEC.md section 2; the rendering of tree.hip k_eval_render: pack_obs + obs_cell).
// Request 0 = no request (finished leaf, collecting game): skipped, its outputs are never read.
__global__ __launch_bounds__(256, 2) void k_dn_conv(const float* __restrict__ P, const float* __restrict__ prep,
                                                    const int8_t* __restrict__ states, const uint32_t* __restrict__ node_game,
                                                    const int32_t* __restrict__ eval_obs, int max_nodes, int n,
                                                    float* __restrict__ a2out, int a2stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
    float* a1 = smem + w * WAVE_LDS;
    float* x0 = a1 + 32 * A1CS;
    // ---- per-lane constants ----
    int boff2[2][8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int p = 32 * t + l31, y = p >> 2, x = p & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 2 * j + half;
            boff2[t][j] = (y + (k >> 2)) * 7 + x + (k & 3);
        }
    }
    float bias2[16], bias1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
        bias2[r] = P[OFF_C2B + i];
        bias1[r] = P[OFF_C1B + i];
    }
    float w1[8];      // conv1 weights: A operand of step st = W1[co = l31][k = 2 st + half]
    int koff1[8];
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        const int k = 2 * st + half;
        w1[st] = P[OFF_C1W + l31 * 16 + k];
        koff1[st] = (k >> 2) * 10 + (k & 3);
    }
    const float4* W2 = reinterpret_cast<const float4*>(prep + PREP_W2) + lane;

    // the two dependent global reads per request (request -> packed game) are issued one state ahead
    const int stride = gridDim.x * 4;
    int s = blockIdx.x * 4 + w;
    int o_next = (!states && s < n) ? eval_obs[s] : 0;
    uint32_t gw_next = 0;
    if (!states && s < n && lane < 16) gw_next = node_game[((size_t)s * max_nodes + o_next) * tmcts::GAME_DW + lane];
    // the two hidden rows never change
    if (lane < 20) x0[lane] = 0.0f;
    for (; s < n; s += stride) {
        int zoff = 0;                        // weight stream re-read per state (keeps it out of ~130 hoisted registers)
        asm volatile("" : "+s"(zoff));
        const float4* W2s = W2 + zoff;
        const int o = o_next;
        const uint32_t gw = gw_next;
        const int sn = s + stride;
        if (!states) {
            o_next = (sn < n) ? eval_obs[sn] : 0;
            if (o == 0) {
                gw_next = (sn < n && lane < 16) ? node_game[((size_t)sn * max_nodes + o_next) * tmcts::GAME_DW + lane] : 0u;
                continue;
            }
        }
        // ---- input: rows 2..21 of x0 ----
        if (states) {
            for (int i = lane; i < 200; i += 64) x0[20 + i] = (float)states[(size_t)s * 200 + i];
        } else {
            const uint32_t pa = (uint32_t)__shfl((int)gw, 10, 64), pb = (uint32_t)__shfl((int)gw, 11, 64);
            const bool ended = (pb >> 8) & 1u;
            const int piece = pa & 0xFF, rot = (pa >> 8) & 0xFF;
            const int px = (int)(int8_t)((pa >> 16) & 0xFF), py = (int)(int8_t)((pa >> 24) & 0xFF);
            const uint32_t mask = (piece < 7 && rot < 4) ? tmcts::PIECE_MASK[piece][rot] : 0u;
#pragma unroll
            for (int it = 0; it < 4; ++it) {   // uniform trip count: the shuffles need every lane active
                const int i = lane + 64 * it, ic = min(i, 199);
                const int r = ic / 10, c = ic - 10 * r;
                const uint32_t w2 = (uint32_t)__shfl((int)gw, r >> 1, 64);
                float v = (float)((w2 >> (16 * (r & 1) + c)) & 1u);
                const int dr = r - py, dc = c - px;
                const bool pc = dr >= 0 && dr < 4 && dc >= 0 && dc < 4 && ((mask >> (4 * dr + dc)) & 1u);
                if (!ended && pc) v = -1.0f;
                if (i < 200) x0[20 + i] = v;
            }
            gw_next = (sn < n && lane < 16) ? node_game[((size_t)sn * max_nodes + o_next) * tmcts::GAME_DW + lane] : 0u;
        }
        lds_fence();
        // ---- conv1 (K = 16 = 8 steps of two taps): 133 positions = 5 tiles (the last one 5 positions) ----
        {
#pragma unroll 1
            for (int t = 0; t < 5; ++t) {
                const int p = 32 * t + l31, pc = min(p, C1P - 1), y = pc / 7, base = y * 10 + (pc - 7 * y);
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = bias1[r];
#pragma unroll
                for (int st = 0; st < 8; ++st)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[st], x0[base + koff1[st]], acc, 0, 0, 0);
                if (p < C1P) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
                        a1[i * A1CS + p] = leaky(acc[r]);
                    }
                }
            }
        }
        lds_fence();
        // ---- conv2: 64 positions = 2 tiles, straight to the scratch row of the state (flatten order co*64 + y*4 + x) ----
        {
            f32x16 acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = bias2[r];
            conv2_mfma(a1, boff2, W2s, acc);
            float* dst = a2out + (size_t)s * a2stride;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
                    dst[i * C2P + 32 * t + l31] = leaky(acc[t][r]);
                }
        }
        lds_fence();
    }
}

// fc1 (2048 -> 128) + LeakyReLU, fc_v (128 -> atoms) and the softmax of 16 states per workgroup.  v_mfma_f32_16x16x4_f32:
// lane l supplies A[i = l&15][k = l>>4], B[k = l>>4][j = l&15], holds D[i = (l>>4)*4 + r][j = l&15]; i = hidden unit / atom
// (weights are the A operand), j = state.  Wave w owns hidden units 16 w .. 16 w + 15; activations staged through LDS in
// FC_KC-wide K chunks, weight quads double-buffered in registers.
constexpr int FC_KC = 256, FC_PITCH = FC_KC + 4, FC_ST = 16;
constexpr int HS_PITCH = HID + 4, LG_PITCH = 65;
__global__ __launch_bounds__(512) void k_dn_fc(const float* __restrict__ P, const float* __restrict__ prep,
                                               const float* __restrict__ a2, int a2stride, int n, int atoms,
                                               const int32_t* __restrict__ eval_obs, float* __restrict__ out, int out_stride) {
    __shared__ __attribute__((aligned(16))) float bt[2][FC_ST * FC_PITCH];
    __shared__ float lg[FC_ST * LG_PITCH];
    __shared__ double ex[FC_ST * 64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, kk = lane >> 4, l15 = lane & 15;
    const int s0 = blockIdx.x * FC_ST;
    if (eval_obs) {
        // a tile none of whose 16 slots carries a request: nothing to do
        const bool mine = threadIdx.x < FC_ST && s0 + (int)threadIdx.x < n && eval_obs[s0 + threadIdx.x] != 0;
        if (!__syncthreads_or(mine)) return;
    }
    const float4* W = reinterpret_cast<const float4*>(prep + PREP_W1) + (size_t)w * 128 * 64 + lane;
    f32x4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = P[OFF_F1B + 16 * w + kk * 4 + r];
    // fc_v's A operands (waves 0..3: atoms 16 w .. 16 w + 15), requested now, used after the K loop
    float fv[32];
    f32x4 accv;
    {
        const int o = 16 * (w & 3) + l15;
#pragma unroll
        for (int st = 0; st < 32; ++st) fv[st] = (w < 4 && o < atoms) ? P[OFF_FVW + o * HID + 4 * st + kk] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oo = 16 * (w & 3) + kk * 4 + r;
            accv[r] = (w < 4 && oo < atoms) ? P[OFF_FVW + atoms * HID + oo] : 0.0f;
        }
    }
    // staging: 16 rows x FC_KC floats per chunk, 64 threads per row, 16-byte pieces, two passes of 8 rows
    constexpr int TPR = FC_KC / 4, RPP = 512 / TPR, NPASS = FC_ST / RPP;
    const int row0 = threadIdx.x / TPR, c4 = (threadIdx.x % TPR) * 4;
    // the activations of chunk c + 2 are requested while chunk c is multiplied (two register sets): the rows come from beyond
    // this XCD's L2 (written by convolution waves all over the chip), a round trip one chunk of MFMAs does not cover
    float4 st[2][NPASS];
    auto gload = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int sa = s0 + row0 + RPP * i;
            st[chunk & 1][i] = (sa < n) ? *reinterpret_cast<const float4*>(a2 + (size_t)sa * a2stride + chunk * FC_KC + c4)
                                        : make_float4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i)
            *reinterpret_cast<float4*>(&bt[chunk & 1][(row0 + RPP * i) * FC_PITCH + c4]) = st[chunk & 1][i];
    };
    constexpr int NCH = KFC / FC_KC, QPC = FC_KC / 16;
    // weights: a ring of WRING groups of four quads (16 MFMA steps each), the group WRING - 1 ahead requested while a group is
    // multiplied (one workgroup per CU: the lead of five groups covers the L2 round trip, valuenet.hip k_vn_fc1)
    constexpr int GRP = 16, NGRP = FC_KC / 4 / GRP, WRING = 6;
    static_assert(QPC == 4 * NGRP, "a group = four weight quads");
    float4 wring[WRING][4];
    auto wload = [&](int G) {          // G = chunk * NGRP + group
#pragma unroll
        for (int q = 0; q < 4; ++q) wring[G % WRING][q] = W[((size_t)G * 4 + q) * 64];
    };
    gload(0);
    gload(1);
#pragma unroll
    for (int G0 = 0; G0 < WRING - 1; ++G0) wload(G0);
    lstore(0);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const float* b0 = &bt[c & 1][l15 * FC_PITCH + kk];
        // the B operands (LDS) of the next 16 MFMA steps are requested while this group's 16 MFMAs issue (valuenet.hip k_vn_fc1)
        float bb[2][GRP];
        auto load_b = [&](int grp, int buf) {
#pragma unroll
            for (int i = 0; i < GRP; ++i) bb[buf][i] = b0[4 * (GRP * grp + i)];
        };
        load_b(0, 0);
#pragma unroll
        for (int gq = 0; gq < NGRP; ++gq) {
            const int G = c * NGRP + gq;
            if (G + WRING - 1 < NCH * NGRP) wload(G + WRING - 1);
            if (gq + 1 < NGRP) load_b(gq + 1, (gq + 1) & 1);
#pragma unroll
            for (int i = 0; i < GRP; ++i) {
                const float4 w4 = wring[G % WRING][i / 4];
                const int r = i & 3;
                const float a = (r == 0) ? w4.x : (r == 1) ? w4.y : (r == 2) ? w4.z : w4.w;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb[gq & 1][i], acc, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < GRP / 2; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);      // the next group's reads stay in this group's region: a whole group ahead of their use
        }
        if (c + 1 < NCH) lstore(c + 1);
        if (c + 2 < NCH) gload(c + 2);
        __syncthreads();
    }
    // hidden layer of the 16 states -> LDS hs[state][unit] (the staging buffers are free now)
    float* hs = &bt[0][0];
    static_assert(FC_ST * HS_PITCH <= 2 * FC_ST * FC_PITCH, "hidden tile fits the staging buffers");
    {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = leaky(acc[r]);
        *reinterpret_cast<f32x4*>(&hs[l15 * HS_PITCH + 16 * w + kk * 4]) = o;
    }
    __syncthreads();
    // fc_v: logits[atom][state], K = 128 = 32 steps, four 16-atom tiles on waves 0..3
    if (w < 4) {
        const float* b = &hs[l15 * HS_PITCH + kk];
#pragma unroll
        for (int st = 0; st < 32; ++st) accv = __builtin_amdgcn_mfma_f32_16x16x4f32(fv[st], b[4 * st], accv, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) lg[l15 * LG_PITCH + 16 * w + kk * 4 + r] = accv[r];
    }
    __syncthreads();
    // softmax: wave w takes states 2 w and 2 w + 1, lane b = atom b
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        const int j = 2 * w + h, sidx = s0 + j;
        const bool on = lane < atoms;
        const float x = on ? lg[j * LG_PITCH + lane] : -__builtin_inff();
        float m = x;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
        const double e = on ? tm_exp((double)x - (double)m) : 0.0;
        ex[j * 64 + lane] = e;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double sum = 0.0;
        for (int b = 0; b < atoms; ++b) sum = sum + ex[j * 64 + b];
        if (on && sidx < n && !(eval_obs && eval_obs[sidx] == 0)) out[(size_t)sidx * out_stride + lane] = (float)(e / sum);
    }
}

static std::once_flag g_attr_once;
static int g_attr_err = 0;

static int dn_forward_impl(const float* P, const float* prepared, const int8_t* states, const uint32_t* node_game,
                           const int32_t* eval_obs, int max_nodes, int n, int atoms, float* out, int out_stride,
                           float* scratch, hipStream_t stream) {
    if (n <= 0) return 0;
    if (atoms < 1 || atoms > 64) return (int)hipErrorInvalidValue;
    const int lds = 4 * WAVE_LDS * (int)sizeof(float);
    std::call_once(g_attr_once, [&] {
        g_attr_err = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_dn_conv),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    });
    if (g_attr_err) return g_attr_err;
    int blocks = (n + 3) / 4;
    if (blocks > 512) blocks = 512;      // two resident workgroups per CU, waves stride over the states
    hipLaunchKernelGGL(k_dn_conv, dim3(blocks), dim3(256), lds, stream, P, prepared, states, node_game, eval_obs, max_nodes,
                       n, scratch, TM_DISTNET_SCRATCH);
    hipLaunchKernelGGL(k_dn_fc, dim3((n + FC_ST - 1) / FC_ST), dim3(512), 0, stream, P, prepared, scratch, TM_DISTNET_SCRATCH,
                       n, atoms, eval_obs, out, out_stride);
    return (int)hipGetLastError();
}

}  // namespace tmcts_dn

using namespace tmcts_dn;

extern "C" {

int tm_distnet_prepare(const float* P, float* prepared, void* stream_) {
    hipLaunchKernelGGL(k_dn_prepare, dim3((PREP_TOTAL + 255) / 256), dim3(256), 0, (hipStream_t)stream_, P, prepared);
    return (int)hipGetLastError();
}

int tm_distnet_forward(const float* P, const float* prepared, const int8_t* states, int n, int atoms, float* dist,
                       int dist_stride, float* scratch, void* stream_) {
    if (dist_stride < atoms) return (int)hipErrorInvalidValue;
    return dn_forward_impl(P, prepared, states, nullptr, nullptr, 0, n, atoms, dist, dist_stride, scratch, (hipStream_t)stream_);
}

int tm_distnet_forward_requests(const float* P, const float* prepared, const tm_store* s, float* scratch, void* stream_) {
    if (s->kind != TM_KIND_DIST || s->eval_slots != 1) return (int)hipErrorInvalidValue;
    return dn_forward_impl(P, prepared, nullptr, s->node_game, s->eval_obs, s->max_nodes, s->n_games, s->dist_bins,
                           s->eval_dist, TM_DIST_ROW, scratch, (hipStream_t)stream_);
}

}  // extern "C"
