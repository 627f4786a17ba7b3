// Value network forward for gfx950 (model/model_vv.py:13-52; Model_VV.inference 210-217).
//
// Numerics contract (shared with oracle/valuenet_oracle.c): every output element is ONE fp32 fma chain,
// acc = bias; for k ascending: acc = fma(x_k, w_k, acc), with k = ci*9 + ky*3 + kx for the convolutions and
// the flat input index for the linear layers; sigmoid through tm_exp (explicit fma polynomial).  The MFMA
// kernels (v_mfma_f32_32x32x2_f32 is exactly such a k-ordered fma chain) and the plain kernels below
// therefore produce identical bits.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>
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
                         float* __restrict__ var, int n, const int32_t* __restrict__ eval_obs) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * 2) return;
    int s = t >> 1, j = t & 1;
    if (eval_obs && eval_obs[s] == 0) return;   // unused evaluation slot: its outputs are never read
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
namespace tmcts_vn {

// ===================================================================================================
// MFMA path.  v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2] * B[2x32]; lane l supplies A[i=l&31][k=l>>5] and
// B[k=l>>5][j=l&31]; lane l holds D[i=(r&3)+8*(r>>2)+4*(l>>5)][j=l&31] in register r of 16.  Per output it is
// the chain fma(a_k1,b_k1, fma(a_k0,b_k0, C)): stepping k in pairs, ascending, reproduces the reference order.
// Orientation: i = output channel / hidden unit (weights are the A operand), j = output position / state.
//
// Prepared weights ("T4" streams): for MFMA step s the A operand of lane l is W[row=l&31][k=2s+(l>>5)];
// four consecutive steps are stored together so one global_load_dwordx4 per lane feeds four MFMAs:
//   T4[(s/4)*64 + l][s%4]
// ===================================================================================================
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int SCHED_REGION_QUADS = 6;   // conv_mfma closes its scheduling region every this many quads of MFMA steps
constexpr int PREP_W2 = 0, PREP_W3 = 9216, PREP_W1 = 18432, PREP_TOTAL = 18432 + 458752;
constexpr int A1CS = 145;   // conv1-output channel stride in LDS (18*8 = 144, +1 against bank conflicts)
constexpr int A2CS = 97;    // conv2-output channel stride in LDS (16*6 = 96, +1)
// Two workgroups per CU: conv2's outputs live in registers until its last LDS read of a1 has been issued, so a2 can
// take a1's place; a wave then needs 19.4 KB instead of 31.8 KB and two workgroups (8 waves) fit a CU's 160 KB: the
// vector-ALU phases of one wave (render, conv1, epilogues) run under the other wave's MFMAs (measured, r01: 118 us -> 104 us
// per launch of 4096 states).
constexpr int CONV_WG_PER_CU = 2;
constexpr int WAVE_LDS = 32 * A1CS + 200;               // floats per wave: a1 (a2 overlays it), input

__global__ void k_vn_prepare(const float* __restrict__ P, float* __restrict__ prep) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 2 * 9216) {
        const float* W = P + (t < 9216 ? OFF_C2W : OFF_C3W);
        int e = t % 9216;
        int q = e / 256, l = (e / 4) % 64, r = e % 4;
        int s = 4 * q + r;
        int k = 2 * s + (l >> 5), row = l & 31;
        prep[t] = W[row * 288 + k];
    } else if (t < PREP_TOTAL) {
        // fc1 runs on v_mfma_f32_16x16x4_f32: lane l supplies A[row = l&15][k = 4s + (l>>4)] for step s of a 16-row tile
        int e = t - 18432;
        int ht = e / (112 * 256), e2 = e % (112 * 256);
        int q = e2 / 256, l = (e2 / 4) % 64, r = e2 % 4;
        int s = 4 * q + r;
        int k = 4 * s + (l >> 4), row = 16 * ht + (l & 15);
        prep[t] = P[OFF_F1W + (size_t)row * A3 + k];
    }
}

__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 3x3 valid convolution, 32 -> 32 channels, as TILES position tiles of 32 on the matrix cores.
// in: LDS activations [32][IN_CS] (rows of IN_W), boff[t][j]: this lane's LDS float offset for tile t and the
// j-th step of an 18-k (two input channel) block; W: prepared T4 stream (+lane already applied).
template <int TILES, int IN_CS>
__device__ __forceinline__ void conv_mfma(const float* __restrict__ in, const int (&boff)[TILES][9],
                                          const float4* __restrict__ W, f32x16 (&acc)[TILES]) {
    // Software pipeline: the B operands (LDS) of quad q+1 and the weights (global) of quad q+2 are requested
    // while the 4*TILES MFMAs of quad q issue; accumulators rotate so consecutive MFMAs are independent.
    float4 wq[3];
    float bb[2][4][TILES];
    auto load_b = [&](int q, int buf) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = 4 * q + r, cp = s / 9, j = s % 9;   // 18 k per pair of input channels
#pragma unroll
            for (int t = 0; t < TILES; ++t) bb[buf][r][t] = in[boff[t][j] + cp * 2 * IN_CS];
        }
    };
    wq[0] = W[0];
    wq[1] = W[64];
    load_b(0, 0);
#pragma unroll
    for (int q = 0; q < 36; ++q) {
        if (q + 2 < 36) wq[(q + 2) % 3] = W[(q + 2) * 64];
        if (q + 1 < 36) load_b(q + 1, (q + 1) & 1);
        const float4 w4 = wq[q % 3];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = (r == 0) ? w4.x : (r == 1) ? w4.y : (r == 2) ? w4.z : w4.w;
#pragma unroll
            for (int t = 0; t < TILES; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb[q & 1][r][t], acc[t], 0, 0, 0);
        }
        // issue order inside the quad: one LDS read behind every MFMA, the weight load up front
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#pragma unroll
        for (int i = 0; i < 4 * TILES; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        // Close the scheduling region every few quads: the group solver's cost grows steeply with the number of
        // groups in one region (all 36 quads in one region take minutes to compile).
        if (q % SCHED_REGION_QUADS == SCHED_REGION_QUADS - 1) __builtin_amdgcn_sched_barrier(0);
    }
}

// Input: either int8 states [n][200], or (states == nullptr) the evaluation requests of the tree engine's dense list (ReqList
// below): entry p = (request slot, packed observation of the slot's game, ENGINE_SPEC.md section 7), rendered here on the fly
// (0 empty, 1 locked, -1 falling piece).
constexpr int CONV_WAVES_PER_SIMD = 2;   // two workgroups per CU = two waves per SIMD: 256 registers each
// The tree engine's dense request list (include/tetris_mcts_hip.h, tm_store::eval_list): `segs` segments with one counter
// each under the current parity; entry d of segment sg is list[(sg + segs * (d / slots)) * slots + d % slots] = (request slot,
// observation index).  Dense position p counts through the segments in order.
struct ReqList {
    const int2* list;
    const int32_t* cnt;
    int parity, segs, slots;
};
// lane i: inclusive prefix of the segments' counts; total = the number of requests
__device__ __forceinline__ int req_prefix(const ReqList& rq, int lane, int& total) {
    int incl = lane < rq.segs ? rq.cnt[lane * 2 + rq.parity] : 0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    total = __shfl(incl, 63, 64);
    return incl;
}
// index into rq.list of dense position p (wave-uniform, p < total; every lane of the wave calls it)
__device__ __forceinline__ int req_at(const ReqList& rq, int incl, int p, int lane) {
    const int sg = __popcll(__ballot(lane < rq.segs && incl <= p));     // the first segment whose inclusive prefix exceeds p
    const int before = __shfl(incl, sg > 0 ? sg - 1 : 0, 64);
    const int d = p - (sg > 0 ? before : 0);
    return (sg + rq.segs * (d / rq.slots)) * rq.slots + d % rq.slots;
}
#include "valuenet_conv.inc"

// fc1 (1792 -> 256) + ReLU on v_mfma_f32_16x16x4_f32 (D[16x16] += A[16x4] B[4x16]; lane l: A[i=l&15][k=l>>4],
// B[k=l>>4][j=l&15], D[i=(l>>4)*4+r][j=l&15]; per output a k-ordered fma chain, 4 terms per instruction).
// A workgroup of 8 waves owns a tile of ROWS = 16 RT states x 256 / NY hidden units (NY workgroups per tile): wave w one
// 16-row hidden tile and NST 16-state tiles, two waves per SIMD so one wave's waits hide behind the other's MFMAs.
// Activations are staged through LDS in KC-wide K chunks (requested two chunks ahead), the B operands read from LDS a group
// of steps ahead, the weights held in a ring of WRING groups.  Two shapes, the same arithmetic and the same bits (every
// output element is its own k-ordered chain whatever the tiling):
//   <2, 4, 256, 6>  32-state tiles, four workgroups of 64 units each, 152 registers, one workgroup per CU: the single-leaf
//                   kinds, whose dense request list leaves 60-100 tiles a launch - a latency problem (r03 gave a tile two
//                   workgroups of 128 units: half of the CUs without a workgroup, 40 us whatever the number of tiles; r04 calls
//                   B-G: 51 -> 33 us);
//   <4, 4, 128, 3>  64-state tiles, four workgroups of 64 units each, two workgroups per CU: the leaf-parallel kinds' hundreds
//                   of tiles - an L2-bandwidth problem (every tile streams the 1.8 MB of weights: twice the states per tile =
//                   half the stream).  (<4, 2, 128, 3>, two workgroups of 128 units per tile, reads the activations half as
//                   often and was measured at 121 us against 80: 224 workgroups of four dependent accumulators each.)
typedef float f32x4 __attribute__((ext_vector_type(4)));
// The output layer (256 -> 2, sigmoid, affine) is folded in: a tile's NY workgroups (parts of the hidden units) store
// their part of h with write-through (sc1) stores, wait for them, and arrive on the tile's counter; the LAST to arrive reads
// the other parts with sc1 loads (the valid hand-off form of MI355X_MICROARCH.md: 16-byte sc1 stores and loads, no fences)
// and runs the 2 x 256 fma chains of its states - the same chain, in the same order, as k_fc_out.  Nobody waits for
// anybody.  `cnt`: one int per tile (the first pad word of the scratch row of the tile's first state), zeroed by k_vn_conv at
// the start of every evaluation (the kernel also leaves it zero).
typedef float f32x4v __attribute__((ext_vector_type(4)));
#ifdef TM_FC1_TIMELINE   // (measurement builds, scripts/fc1_timeline.py: thread 0's s_memrealtime at the kernel's stations, in the pad words of
                         //  scratch row s0 + blockIdx.y)
#define FC1_STAMP(i) do { if (threadIdx.x == 0 && s0 + (int)blockIdx.y < n) reinterpret_cast<int*>(hout)[(size_t)(s0 + blockIdx.y) * hstride + HID + 1 + (i)] = (int)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FC1_STAMP(i) do { } while (0)
#endif
template <int RT, int NY, int KC, int WRING, int PF>
__global__ __launch_bounds__(512) void k_vn_fc1(const float* __restrict__ P, const float* __restrict__ prep,
                                                const float* __restrict__ a3, int a3stride, int n,
                                                float* __restrict__ hout, int hstride,
                                                ReqList rq, int32_t* __restrict__ cnt, int cnt_stride,
                                                float* __restrict__ v_out, float* __restrict__ var_out) {
    constexpr int ROWS = 16 * RT, PITCH = KC + 4, HS_PITCH = HID + 4;
    constexpr int UNITS = HID / NY, HT = UNITS / 16, SG = 8 / HT, NST = RT / SG;      // hidden tiles per workgroup, groups of state tiles, state tiles per wave
    static_assert(HT * SG == 8 && NST * SG == RT && NST >= 1 && ROWS <= 64, "eight waves = hidden tiles x groups of state tiles; one lane per row");
    constexpr int BT = ROWS * PITCH > (ROWS * HS_PITCH + 1) / 2 ? ROWS * PITCH : (ROWS * HS_PITCH + 1) / 2;     // floats per staging buffer
    __shared__ __attribute__((aligned(16))) float bt[2][BT];
    __shared__ int row_slot[ROWS];        // request mode: where row j of the tile delivers its outputs
    __shared__ int last_flag;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, kk = lane >> 4, l15 = lane & 15;
    const int s0 = blockIdx.x * ROWS;
    int my_slot = 0;
    FC1_STAMP(0);
    if (rq.list) {
        // rows = dense positions of the request list; a tile past its end has nothing to do (all of its workgroups leave,
        // before they have asked the memory system for anything else)
        const int incl = req_prefix(rq, lane, n);
        if (s0 >= n) return;
        if (w == 0) {
            // lane j resolves row j (all list entries in flight together; the slot is needed after the K loop only)
            const int p = min(s0 + (lane & (ROWS - 1)), n - 1);
            int sg = 0;
            for (int k = 0; k < rq.segs; ++k) sg += __builtin_amdgcn_readlane(incl, k) <= p ? 1 : 0;     // first segment whose inclusive prefix exceeds p
            const int before = __shfl(incl, sg > 0 ? sg - 1 : 0, 64);
            const int d = p - (sg > 0 ? before : 0);
            my_slot = rq.list[(sg + rq.segs * (d / rq.slots)) * rq.slots + d % rq.slots].x;
        }
    }
    const int ht = blockIdx.y * HT + (w % HT);       // 16-row hidden tile 0..15
    const int st0 = (w / HT) * NST;                   // this wave's first 16-state tile
    const float4* W = reinterpret_cast<const float4*>(prep + PREP_W1) + (size_t)ht * 112 * 64 + lane;
    f32x4 acc[NST];
#pragma unroll
    for (int t = 0; t < NST; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = P[OFF_F1B + 16 * ht + kk * 4 + r];
    constexpr int NCH = A3 / KC;          // chunks of KC k = KC/4 MFMA steps
    // a group = GRP MFMA steps (per state tile) = GRP/4 weight quads; weights: a ring of WRING groups, the group WRING - 1 ahead
    // requested while a group is multiplied (measured, r04 calls C-E per launch of ~60 32-state tiles: chunk-deep weights without
    // the LDS pipeline 45 us, with it 35 us; a ring of three groups 45 us again - the weights need the longer lead when nothing
    // else is resident to cover it)
    constexpr int GRP = 16 / NST, QPG = GRP / 4, NGRP = KC / 4 / GRP;
    static_assert(A3 % KC == 0 && NCH >= 2 && (KC / 4) % GRP == 0, "chunking");
    float4 wring[WRING][QPG];
    auto wload = [&](int G) {          // G = chunk * NGRP + group
#pragma unroll
        for (int q = 0; q < QPG; ++q) wring[G % WRING][q] = W[((size_t)G * QPG + q) * 64];
    };
    // The first weights (and the biases above) are requested BEFORE the activations: a wave's loads return in order, and the first
    // MFMA needs both.  (Requested before the request list's counters are read, by every workgroup of the grid: 34.4 us against 31.5.)
#pragma unroll
    for (int G0 = 0; G0 < WRING - 1; ++G0) wload(G0);
    // staging: ROWS rows x KC floats per chunk, KC/4 threads per row, 16-byte pieces
    constexpr int TPR = KC / 4, RPP = 512 / TPR, NPASS = ROWS / RPP;
    const int row0 = threadIdx.x / TPR, c4 = (threadIdx.x % TPR) * 4;
    // The activations of chunk c + 2 are requested while chunk c is multiplied (two register sets): a tile's rows were written
    // by convolution waves all over the chip, so they come from beyond this XCD's L2, and one chunk of MFMAs (under 2 us) does
    // not cover that round trip.  (Kept as a select: a plain 16-byte copy into the array sent the whole array to scratch
    // memory - tests/test_abi.py holds every hot kernel to a private segment of 0.)
    // r06 (scripts/fc1_timeline.py, the kernel's stations in time): PF register sets = PF chunks in flight, the request for chunk
    // c + PF issued at the START of chunk c's MFMAs (until then it went out after them: one chunk of cover, not two), and the
    // first weights requested before the first activations (a wave's loads return in order).  1 867 states, entry to outputs:
    // 34.2 us before; PF = 2 / 3 / 4 / 7 (the whole tile up front): 28.9 / 29.6 / 30.2 / 32.6 - the more is asked for at once, the
    // later the first chunk is there (4.5 ... 9.4 us), and the K loop does not shorten (18 - 19 us for 12 of matrix time).
    float4 st[PF][NPASS];
    auto gload = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int sa = s0 + row0 + RPP * i;
            st[chunk % PF][i] = (sa < n) ? *reinterpret_cast<const float4*>(a3 + (size_t)sa * a3stride + chunk * KC + c4)
                                         : make_float4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i)
            *reinterpret_cast<float4*>(&bt[chunk & 1][(row0 + RPP * i) * PITCH + c4]) = st[chunk % PF][i];
    };
    static_assert(PF >= 2 && PF <= A3 / KC, "chunks in flight");
#pragma unroll
    for (int c = 0; c < PF; ++c) gload(c);
    FC1_STAMP(1);
    lstore(0);
    __syncthreads();
    FC1_STAMP(2);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + PF < NCH) gload(c + PF);         // into the register set chunk c's staging has left (lstore(c): before the last barrier)
        const float* b0 = &bt[c & 1][(16 * st0 + l15) * PITCH + kk];
        // The B operands (LDS) of the next group of MFMA steps are requested while this group's MFMAs issue (r03 - r04 call C:
        // the compiler's own order was read, wait for it, two MFMAs, read ...: an LDS round trip per pair of MFMAs, 45 us per
        // launch whatever the number of tiles, against 12 us of matrix-core time).
        float bb[2][NST][GRP];
        auto load_b = [&](int grp, int buf) {
#pragma unroll
            for (int t = 0; t < NST; ++t)
#pragma unroll
                for (int i = 0; i < GRP; ++i) bb[buf][t][i] = b0[16 * t * PITCH + 4 * (GRP * grp + i)];
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
#pragma unroll
                for (int t = 0; t < NST; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb[gq & 1][t][i], acc[t], 0, 0, 0);
            }
            // issue order inside the group: an LDS read behind every other MFMA (the reads pair up into ds_read2_b32)
#pragma unroll
            for (int i = 0; i < NST * GRP / 2; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);      // the next group's reads stay in this group's region: a whole group ahead of their use
        }
        if (c + 1 < NCH) lstore(c + 1);          // (requested PF - 1 chunks ago)
        __syncthreads();
        if (c == NCH - 1) FC1_STAMP(3);
    }
    // D[i = kk*4 + r][j = l15]: four consecutive hidden units of one state per lane -> one 16-byte write-through store,
    // and a copy into LDS (hs[state][hidden unit], the layout the output layer below reads)
    const int i0 = 16 * ht + kk * 4;
    float* hs = &bt[0][0];                      // ROWS x HS_PITCH floats: the K loop is over, its staging buffers are free
    static_assert(ROWS * HS_PITCH <= 2 * BT, "hidden tile fits the staging buffers");
#pragma unroll
    for (int t = 0; t < NST; ++t) {
        f32x4v o0;
#pragma unroll
        for (int r = 0; r < 4; ++r) o0[r] = acc[t][r] > 0.f ? acc[t][r] : 0.f;
        const int row = 16 * (st0 + t) + l15;
        if (s0 + row < n) {
            float* dst = hout + (size_t)(s0 + row) * hstride + i0;
            asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(o0) : "memory");
        }
        *reinterpret_cast<f32x4v*>(&hs[row * HS_PITCH + i0]) = o0;
    }
    if (rq.list && w == 0 && lane < ROWS) row_slot[lane] = my_slot;      // (the list entry has had the whole K loop to arrive)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    FC1_STAMP(4);
    if (threadIdx.x == 0) {
        // (relaxed on purpose: the hand-off is carried by the write-through sc1 stores + s_waitcnt before the arrival and the sc1
        // loads after it - MI355X_MICROARCH.md's measured form.  With __ATOMIC_ACQ_REL here the compiler adds buffer_wbl2 sc1 /
        // buffer_inv sc1 around the atomic: 33.4 -> 35.8 us per launch, r04; tests/test_gpu_valuenet.py holds the folded output
        // layer to the oracle's bits on every run)
        const int old = __hip_atomic_fetch_add(&cnt[(size_t)blockIdx.x * cnt_stride], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = old;
        if (old == NY - 1) __hip_atomic_store(&cnt[(size_t)blockIdx.x * cnt_stride], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
    }
    __syncthreads();
    FC1_STAMP(5);
    if (last_flag != NY - 1) return;
    // ---- this workgroup arrived last: the other parts of h past the caches (ROWS states x (256 - UNITS) units) ----
    {
        // all of a thread's 16-byte loads in flight together, one wait (the memory clobbers keep the LDS stores behind it)
        constexpr int OQ = (HID - UNITS) / 4, CNT = ROWS * OQ / 512;      // 16-byte pieces per row of the other parts; per thread
        static_assert(ROWS * OQ % 512 == 0, "whole passes");
        f32x4v wv[CNT];
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            const int e = i * 512 + threadIdx.x, row = e / OQ, c = (e % OQ) * 4;
            const int col = c < (int)blockIdx.y * UNITS ? c : c + UNITS;      // skipping this workgroup's own units
            const float* src = hout + (size_t)(s0 + (s0 + row < n ? row : 0)) * hstride + col;
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(wv[i]) : "v"(src) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            const int e = i * 512 + threadIdx.x, row = e / OQ, c = (e % OQ) * 4;
            const int col = c < (int)blockIdx.y * UNITS ? c : c + UNITS;
            f32x4v t = wv[i];
            asm volatile("" : "+v"(t));          // (a use the compiler cannot hoist above the wait)
            if (s0 + row < n) *reinterpret_cast<f32x4v*>(&hs[row * HS_PITCH + col]) = t;
        }
    }
    __syncthreads();
    FC1_STAMP(6);
    if (threadIdx.x < 2 * ROWS) {
        // one chain per lane: state j = t >> 1, output o = t & 1 (k_fc_out's thread t = 2 s + o); fma over the 256 hidden units
        // in order
        const int j = threadIdx.x >> 1, o = threadIdx.x & 1;
        int sidx = s0 + j;
        const bool live = sidx < n;
        if (rq.list) sidx = row_slot[j];          // (written by wave 0 before the barrier above)
        if (live) {
            float a = P[OFF_FOB + o];
            const float* x = &hs[j * HS_PITCH];
            const float* wr = P + OFF_FOW + o * HID;
            // the same chain in the same order, its operands fetched four at a time (r06, scripts/fc1_timeline.py: this loop was
            // 4.5 us of the tile's last workgroup - a scalar LDS read and a scalar global read in front of every fma)
            if ((reinterpret_cast<uintptr_t>(wr) & 15) == 0) {
                const float4* x4 = reinterpret_cast<const float4*>(x);
                const float4* w4 = reinterpret_cast<const float4*>(wr);
#pragma unroll 8
                for (int i = 0; i < HID / 4; ++i) {
                    const float4 xv = x4[i], wv = w4[i];
                    a = fmaf(xv.x, wv.x, a); a = fmaf(xv.y, wv.y, a); a = fmaf(xv.z, wv.z, a); a = fmaf(xv.w, wv.w, a);
                }
            } else {
#pragma unroll 8
                for (int i = 0; i < HID; ++i) a = fmaf(x[i], wr[i], a);
            }
            const double e = tm_exp(-(double)a);
            const float sg = (float)(1.0 / (1.0 + e));
            const float tt = sg * P[OFF_UB + o];
            const float res = tt + P[OFF_LB + o];
            if (o == 0) v_out[sidx] = res; else var_out[sidx] = res;
        }
    }
    FC1_STAMP(7);
}

}  // namespace tmcts_vn

using namespace tmcts_vn;

extern "C" {

int tm_valuenet_prepare(const float* P, float* prepared, void* stream_) {
    hipLaunchKernelGGL(k_vn_prepare, dim3((PREP_TOTAL + 255) / 256), dim3(256), 0, (hipStream_t)stream_, P, prepared);
    return (int)hipGetLastError();
}

// reference-order plain kernels (one thread per output); scratch: n x TM_VALUENET_SCRATCH floats
int tm_valuenet_forward_plain(const float* P, const int8_t* states, int n, float* v, float* var, float* scratch,
                              void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0) return 0;
    constexpr int SS = TM_VALUENET_SCRATCH;
    float* a1 = scratch;            // per state: a1 at 0, a2 at A1, a3 at A1+A2, h at A1+A2+A3
    const int T = 256;
    hipLaunchKernelGGL((k_conv3x3_relu<1, 20, 10>), dim3((n * A1 + T - 1) / T), dim3(T), 0, stream, (const float*)nullptr,
                       0, states, P + OFF_C1W, P + OFF_C1B, a1, SS, n);
    hipLaunchKernelGGL((k_conv3x3_relu<32, 18, 8>), dim3((n * A2 + T - 1) / T), dim3(T), 0, stream, a1, SS,
                       (const int8_t*)nullptr, P + OFF_C2W, P + OFF_C2B, a1 + A1, SS, n);
    hipLaunchKernelGGL((k_conv3x3_relu<32, 16, 6>), dim3((n * A3 + T - 1) / T), dim3(T), 0, stream, a1 + A1, SS,
                       (const int8_t*)nullptr, P + OFF_C3W, P + OFF_C3B, a1 + A1 + A2, SS, n);
    hipLaunchKernelGGL(k_fc1_relu, dim3((n * HID + T - 1) / T), dim3(T), 0, stream, a1 + A1 + A2, SS, P + OFF_F1W,
                       P + OFF_F1B, a1 + A1 + A2 + A3, SS, n);
    hipLaunchKernelGGL(k_fc_out, dim3((n * 2 + T - 1) / T), dim3(T), 0, stream, a1 + A1 + A2 + A3, SS, P, v, var, n,
                       (const int32_t*)nullptr);
    return (int)hipGetLastError();
}

static int vn_forward_impl(const float* P, const float* prepared, const int8_t* states, const uint32_t* obs_key,
                           const ReqList& rq, int max_nodes, int n, float* v, float* var,
                           float* scratch, hipStream_t stream) {
    if (n <= 0) return 0;
    constexpr int SS = TM_VALUENET_SCRATCH_MFMA;   // a3 (1792) + hidden (256) + 16 pad words (word 0 of a tile's first row: its arrival counter)
    static_assert(SS >= A3 + HID + 1 && SS % 4 == 0, "scratch row");
    const int lds = 4 * WAVE_LDS * (int)sizeof(float);
    static std::once_flag attr_once;
    static int attr_err = 0;
    std::call_once(attr_once, [&] {
        attr_err = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_vn_conv),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    });
    if (attr_err) return attr_err;
    int blocks = (n + 3) / 4;
    if (blocks > 256 * CONV_WG_PER_CU) blocks = 256 * CONV_WG_PER_CU;   // resident workgroups, waves stride over the states
    hipLaunchKernelGGL(k_vn_conv, dim3(blocks), dim3(256), lds, stream, P, prepared, states, obs_key, rq,
                       max_nodes, n, scratch, SS, reinterpret_cast<int32_t*>(scratch + A3 + HID), 32 * SS);
    if (n >= 8192)      // (request slots: the leaf-parallel kinds' seven per game)
        hipLaunchKernelGGL((k_vn_fc1<4, 4, 128, 3, 2>), dim3((n + 63) / 64, 4), dim3(512), 0, stream, P, prepared, scratch, SS, n,
                           scratch + A3, SS, rq, reinterpret_cast<int32_t*>(scratch + A3 + HID), 64 * SS, v, var);
    else
        hipLaunchKernelGGL((k_vn_fc1<2, 4, 256, 6, 2>), dim3((n + 31) / 32, 4), dim3(512), 0, stream, P, prepared, scratch, SS, n,
                           scratch + A3, SS, rq, reinterpret_cast<int32_t*>(scratch + A3 + HID), 32 * SS, v, var);
    return (int)hipGetLastError();
}

// matrix-core path; prepared: tm_valuenet_prepare output; scratch: n x TM_VALUENET_SCRATCH_MFMA floats
int tm_valuenet_forward(const float* P, const float* prepared, const int8_t* states, int n, float* v, float* var,
                        float* scratch, void* stream_) {
    return vn_forward_impl(P, prepared, states, nullptr, ReqList{nullptr, nullptr, 0, 0, 1}, 0, n, v, var, scratch, (hipStream_t)stream_);
}

// the tree engine's evaluation requests, rendered inside the first kernel: v/var -> s->eval_v / s->eval_var
int tm_valuenet_forward_requests(const float* P, const float* prepared, const tm_store* s, float* scratch, void* stream_) {
    // the requests of the last tm_sim_step launch, drawn from its dense list (s->eval_parity = that launch's)
    const ReqList rq{reinterpret_cast<const int2*>(s->eval_list), s->eval_cnt, s->eval_parity, TM_EVAL_SEGS(s->n_games), s->eval_slots};
    return vn_forward_impl(P, prepared, nullptr, s->obs_key, rq, s->max_nodes,
                           s->n_games * s->eval_slots, s->eval_v, s->eval_var, scratch, (hipStream_t)stream_);
}

}  // extern "C"
