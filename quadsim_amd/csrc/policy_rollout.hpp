// policy_rollout.hpp -- policy-in-the-loop roll-out in ONE launch:
//   for t in 0..T-1:  a_t = clip(MLP(obs_t), -1, 1);  obs_{t+1}, r_t, done_t = env.step(a_t)
// i.e. the loop of run_trained_docking_ppo2.py:37-60 (model.predict(obs, deterministic=True); env.step(action))
// for N envs with the actor of the shipped PPO2 model (shared_fc0 12->128, pi_fc0 128->128, pi 128->4, ReLU).
// The MLP runs on the matrix cores with exact-f32 MFMA (v_mfma_f32_16x16x4_f32: a k-ordered fmaf chain, so the
// network output is an ordinary float32 evaluation), the env step is the same device code as k_env.
//
// Orientation: every layer is computed TRANSPOSED, H^T[hidden][env] = W^T[hidden][k] . X^T[k][env], with the
// weights as the A operand and the activations as the B operand.  An accumulator tile D (16 hidden x 16 envs) then
// holds, in register i of lane l, hidden row 4*(l>>4)+i of env column l&15 -- exactly the (k-slot = l>>4,
// column = l&15) placement the NEXT layer's B operand needs if its k-step "i of tile rt" is defined to sum the
// hidden indices {16 rt + 4 g + i : g = 0..3}.  So accumulators feed the next MFMA directly: no LDS round trip, no
// lane shuffles between layers; only the weight fetch address carries the permutation (weights sit in LDS as
// W^T rows with a 132-float stride, one ds_read_b128 per 16 MFMAs).  Layer 3 is folded into the layer-2 loop
// (each finished 16-row tile of H2 is immediately reduced into the 4 action rows), so H2 is never held whole.
// Per wave and step: 96 + 1024 + 128 MFMAs = 40 k SIMD cycles -- the roll-out is bound by the f32 matrix rate.
#pragma once
#include <hip/hip_runtime.h>
#include "quadsim_device.hpp"

namespace qs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kHid = 128;      // hidden width of both layers
constexpr int kLdW = 132;      // LDS row stride of W2^T / W3^T (floats): 132 = 128 + 4 keeps ds_read_b128 conflict-free
constexpr int kLdW1 = 13;      // LDS row stride of W1^T (12 inputs + 1)

struct MlpArgs {
    const float *wt1;   // [128][12]  = shared_fc0 weight transposed (out, in)
    const float *b1;    // [128]
    const float *wt2;   // [128][128] = pi_fc0 weight transposed
    const float *b2;    // [128]
    const float *wt3;   // [4][128]   = pi weight transposed
    const float *b3;    // [4]
};

constexpr size_t policy_lds_floats()
{
    return (size_t)kHid * kLdW + 16 * kLdW + kHid * kLdW1 + kHid + kHid + 16 + 4 * (12 * 64) + 4 * (64 * 4);
}

__device__ __forceinline__ f32x4 relu4(f32x4 v)
{
    return f32x4{fmaxf(v.x, 0.0f), fmaxf(v.y, 0.0f), fmaxf(v.z, 0.0f), fmaxf(v.w, 0.0f)};
}

// MLP for the 64 envs of one wave: obs (per owning lane) -> 4 actions (per owning lane), clipped to [-1,1]
__device__ __forceinline__ void mlp_actor(const float obs[12], float act[4], const float *sW1, const float *sB1,
                                          const float *sW2, const float *sB2, const float *sW3, const float *sB3,
                                          float *sObs, float *sAct, int lane)
{
    const int c = lane & 15, g = lane >> 4;
    // stage obs^T [12][64] so that each lane can fetch the (k, env column) element its B operand needs
#pragma unroll
    for (int k = 0; k < 12; ++k) sObs[k * 64 + lane] = obs[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- layer 1: H1^T [128][64] = W1^T [128][12] . obs^T [12][64]  (+ b1)
    f32x4 h1[8][4];
    float xb[3][4];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int et = 0; et < 4; ++et) xb[s][et] = sObs[(4 * s + g) * 64 + 16 * et + c];
#pragma unroll
    for (int rt = 0; rt < 8; ++rt) {
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(sB1 + 16 * rt + 4 * g);
        float a[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) a[s] = sW1[(16 * rt + c) * kLdW1 + 4 * s + g];
        f32x4 acc[4] = {bias, bias, bias, bias};              // k-step outermost: consecutive MFMAs on different accumulators
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int et = 0; et < 4; ++et) acc[et] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], xb[s][et], acc[et], 0, 0, 0);
#pragma unroll
        for (int et = 0; et < 4; ++et) {
            h1[rt][et] = relu4(acc[et]);
        }
    }
    // ---- layers 2 + 3: for each 16-row tile of H2^T: 32 k-steps over H1^T, bias, ReLU, then straight into the
    //      action accumulators (rows 0..3 of a 16-row tile; W3^T rows 4..15 are zero)
    f32x4 a3[4];
    {
        const f32x4 bias3 = *reinterpret_cast<const f32x4 *>(sB3 + 4 * g);
#pragma unroll
        for (int et = 0; et < 4; ++et) a3[et] = bias3;
    }
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(sB2 + 16 * nt + 4 * g);
        f32x4 h2[4] = {bias, bias, bias, bias};
#pragma unroll
        for (int rt = 0; rt < 8; ++rt) {
            const f32x4 w = *reinterpret_cast<const f32x4 *>(sW2 + (16 * nt + c) * kLdW + 16 * rt + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int et = 0; et < 4; ++et)
                    h2[et] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i], h1[rt][et][i], h2[et], 0, 0, 0);
        }
        const f32x4 w3 = *reinterpret_cast<const f32x4 *>(sW3 + c * kLdW + 16 * nt + 4 * g);
        f32x4 r[4];
#pragma unroll
        for (int et = 0; et < 4; ++et) r[et] = relu4(h2[et]);
#pragma unroll
        for (int i = 0; i < 4; ++i)                     // k-step outermost: consecutive MFMAs on different accumulators
#pragma unroll
            for (int et = 0; et < 4; ++et) a3[et] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3[i], r[et][i], a3[et], 0, 0, 0);
        // issue order of this tile: every weight read one 16-MFMA group ahead of its use (left alone, hipcc emits
        // read -> wait -> 16 MFMAs and exposes the LDS latency 8 times per tile): 24.9 -> 24.3 us/step.  The same
        // directives change nothing in mlp_actor_critic (43.7 vs 43.8 us/step) and are not applied there.
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x8, 16, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x8, 16, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    // rows 0..3 of the action tile live in lanes 0..15 (g == 0): hand them to the lane that owns the env
    if (g == 0) {
#pragma unroll
        for (int et = 0; et < 4; ++et) *reinterpret_cast<f32x4 *>(sAct + (16 * et + c) * 4) = a3[et];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const f32x4 av = *reinterpret_cast<const f32x4 *>(sAct + lane * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) act[i] = fminf(fmaxf(av[i], -1.0f), 1.0f);
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------------------
// Actor-critic heads of the PPO2 MlpPolicy the reference trains (rl_baselines/common/policies.py:35-92 mlp_extractor
// with net_arch [128, dict(vf=[128], pi=[128])], :583-588): shared_fc0 -> {pi_fc0 -> pi (mean), vf_fc0 -> vf (value)},
// i.e. what model.step evaluates per Runner step (rl_baselines/ppo2/ppo2.py:475).  Same exact-f32 MFMA chaining as
// mlp_actor; the value branch re-uses H1 and lands in row 4 of the 16-row output tile (rows 0..3 = action mean).
// LDS image (floats): W2pi^T 128x132 | W2vf^T 128x132 | W3pi^T 4x132 | W3vf 132 | W1^T 128x13 | b1 128 | b2pi 128 |
// b2vf 128 | b3 16 (mean biases, value bias at [4]) | per-wave stage 4 x 768  = 158 352 B of the CU's 160 KiB.
struct AcArgs {
    const float *wt1, *b1;      // shared_fc0: [128][12] (out, in), [128]
    const float *wt2, *b2;      // pi_fc0:     [128][128], [128]
    const float *wt3, *b3;      // pi:         [4][128], [4]
    const float *wtv2, *bv2;    // vf_fc0:     [128][128], [128]
    const float *wtv3, *bv3;    // vf:         [1][128], [1]
};

struct AcLds {
    const float *W1, *B1, *W2p, *B2p, *W2v, *B2v, *W3p, *W3v, *B3;
};

constexpr size_t ac_lds_floats()
{
    return (size_t)2 * kHid * kLdW + 4 * kLdW + kLdW + kHid * kLdW1 + 3 * kHid + 16 + 4 * (12 * 64);
}

// obs (per owning lane) -> out[0..3] = action mean, out[4] = value (per owning lane).  `stage` = 768 floats per wave.
__device__ __forceinline__ void mlp_actor_critic(const float obs[12], float out[5], const AcLds &L, float *stage, int lane)
{
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int k = 0; k < 12; ++k) stage[k * 64 + lane] = obs[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x4 h1[8][4];
    float xb[3][4];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int et = 0; et < 4; ++et) xb[s][et] = stage[(4 * s + g) * 64 + 16 * et + c];
#pragma unroll
    for (int rt = 0; rt < 8; ++rt) {
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(L.B1 + 16 * rt + 4 * g);
        float a[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) a[s] = L.W1[(16 * rt + c) * kLdW1 + 4 * s + g];
        f32x4 acc[4] = {bias, bias, bias, bias};              // k-step outermost: consecutive MFMAs on different accumulators
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int et = 0; et < 4; ++et) acc[et] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], xb[s][et], acc[et], 0, 0, 0);
#pragma unroll
        for (int et = 0; et < 4; ++et) {
            h1[rt][et] = relu4(acc[et]);
        }
    }
    f32x4 a3[4];
    {
        const f32x4 bias3 = *reinterpret_cast<const f32x4 *>(L.B3 + 4 * g);
#pragma unroll
        for (int et = 0; et < 4; ++et) a3[et] = bias3;
    }
    // Source order = issue order (scheduling barriers): the weights of k-group G + 1 (16 MFMAs = 512 matrix cycles later) are
    // requested before the MFMAs of group G, across tile and branch boundaries too; left alone, hipcc emits read -> wait ->
    // MFMAs and exposes the LDS latency 4-5 times per tile.  The dead rows of the output-layer tile read 16 zero bytes
    // (slots 8..11 of the padded b3): an address select, no branch in the MFMA stream.
    auto w2_group = [&](int G) {                           // G = 64 br + 8 nt + rt
        return *reinterpret_cast<const f32x4 *>(((G >> 6) ? L.W2v : L.W2p) + (16 * ((G >> 3) & 7) + c) * kLdW + 16 * (G & 7) + 4 * g);
    };
    f32x4 wbuf[2];
    wbuf[0] = w2_group(0);
#pragma unroll
    for (int br = 0; br < 2; ++br) {                       // 0: policy branch -> rows 0..3, 1: value branch -> row 4
        const float *B2 = br ? L.B2v : L.B2p;
        const bool row_live = br == 0 ? c < 4 : c == 4;
        const float *w3row = br == 0 ? L.W3p + (c & 3) * kLdW + 4 * g : L.W3v + 4 * g;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            const f32x4 bias = *reinterpret_cast<const f32x4 *>(B2 + 16 * nt + 4 * g);
            const f32x4 w3 = *reinterpret_cast<const f32x4 *>(row_live ? w3row + 16 * nt : L.B3 + 8);
            f32x4 h2[4] = {bias, bias, bias, bias};
#pragma unroll
            for (int rt = 0; rt < 8; ++rt) {
                const int G = 64 * br + 8 * nt + rt;
                if (G + 1 < 128) wbuf[(G + 1) & 1] = w2_group(G + 1);
                const f32x4 w = wbuf[G & 1];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int et = 0; et < 4; ++et)
                        h2[et] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i], h1[rt][et][i], h2[et], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            f32x4 r[4];
#pragma unroll
            for (int et = 0; et < 4; ++et) r[et] = relu4(h2[et]);
#pragma unroll
            for (int i = 0; i < 4; ++i)                     // k-step outermost: consecutive MFMAs on different accumulators
#pragma unroll
                for (int et = 0; et < 4; ++et) a3[et] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3[i], r[et][i], a3[et], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // rows 0..3 sit in lanes g == 0, row 4 in register 0 of lanes g == 1: hand them to the lane that owns the env
    // (the stage is free again: every lane of this wave read its layer-1 operands long ago)
    if (g == 0) {
#pragma unroll
        for (int et = 0; et < 4; ++et) *reinterpret_cast<f32x4 *>(stage + (16 * et + c) * 8) = a3[et];
    } else if (g == 1) {
#pragma unroll
        for (int et = 0; et < 4; ++et) stage[(16 * et + c) * 8 + 4] = a3[et][0];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const f32x4 av = *reinterpret_cast<const f32x4 *>(stage + lane * 8);
    out[0] = av[0]; out[1] = av[1]; out[2] = av[2]; out[3] = av[3];
    out[4] = stage[lane * 8 + 4];
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------------------
// Fast actor: the same MLP on the bf16 matrix rate (16x the f32 MFMA rate) with SPLIT operands.  Every f32 value v
// is carried as two bf16, hi = bf16(v) and lo = bf16(v - hi), and a product x*w is evaluated as
// hi*hi + hi*lo + lo*hi with f32 accumulation (three v_mfma_f32_16x16x32_bf16); the dropped lo*lo term and the
// 16-bit representation leave a relative error of about 2^-17 per product (~1e-5 on an action) instead of f32's
// 2^-24.  NOT bit-compatible with the float32 policy: an opt-in mode (qs_policy_rollout_fast).
// Layout: the accumulator-feeds-next-B-operand trick of the f32 path carries over.  A 16x16x32 B operand holds, in
// element j of lane (g = l>>4, c = l&15), k-slot 8g+j of column c; two accumulator tiles (2p, 2p+1) of the previous
// layer give that lane exactly 8 values -- hidden index h(p,g,j) = 16(2p + (j>>2)) + 4g + (j&3) -- so k-step p of the
// next layer sums those 32 hidden units, and the host packs the weights in that order: A fragments are stored
// ready-made, [tile][k-step][lane][8 bf16], one linear ds_read_b128 per fragment.
// Packed blob (bytes): A2hi 32768 | A2lo 32768 | A1hi 8192 | A1lo 8192 | A3hi 4096 | A3lo 4096 | b1 512 | b2 512 | b3 64.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kFastBlobBytes = 32768 * 2 + 8192 * 2 + 4096 * 2 + 512 + 512 + 64;

__device__ __forceinline__ void split_bf16(float v, __bf16 &hi, __bf16 &lo)
{
    hi = (__bf16)v;
    lo = (__bf16)(v - (float)hi);
}

// ReLU + hi/lo split of TWO accumulator values into one packed register each: 7 VALU instructions per pair (integer max
// against 0 = ReLU without a canonicalising v_max; v_cvt_pk_bf16_f32 rounds both at once; the hi halves are widened back with
// a shift and a mask; one packed subtract; one more v_cvt_pk) where the value-at-a-time form takes 14.  Same roundings.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void relu_split_pair(float v0, float v1, uint32_t &hi, uint32_t &lo)
{
    v0 = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v0), 0));
    v1 = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v1), 0));
    const f32x2 v = {v0, v1};
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    const f32x2 r = {v0 - __builtin_bit_cast(float, hi << 16), v1 - __builtin_bit_cast(float, hi & 0xffff0000u)};
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
}
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 w) { return __builtin_bit_cast(bf16x8, w); }

#ifndef QS_WEAVE
#define QS_WEAVE 1      // 0: layer-2 tiles one after the other (A/B reference), 1: hand-woven software pipeline
#endif

// Layer 2 of the split-bf16 heads: NT row tiles of 16 hidden units (one 128 x 128 layer = 8 tiles; the actor-critic runs its
// two branches as one sequence of 16), each 48 MFMAs over the four env tiles of the wave, followed by ReLU + hi/lo split into
// the B operands of the output layer (`out_step(pair, ch, cl)` after every second tile = one k-step of 32 hidden units).
// The tiles run as a software pipeline: the MFMAs of tile s + 1 are issued around the split of tile s, which depends only on
// tile s's accumulators -- the split's VALU work then rides in the issue shadow of the MFMAs (16 matrix cycles each, 8 of
// which hold the SIMD's vector-issue port) instead of standing between two MFMA runs.  hipcc does not weave the two streams by itself (at this register
// pressure its scheduler falls back to source order, sched_group_barrier patterns included), so the source order IS the woven
// order, pinned by scheduling barriers: MFMA m of tile s + 1 (order k-step | term | env tile, so that consecutive MFMAs use
// different accumulators), then ONE instruction-sized piece of the split of tile s (pair m / 6, piece m % 6); the A fragments
// of k-step p + 1 are requested while k-step p multiplies.  Runner, 65 536 envs: 17.1 -> 15.2 us per step.
struct L2Tile { const bf16x8 *hi, *lo; const float *bias; };     // fragments [k-step][lane] of the tile; the lane's 4 bias rows

// NET / E0: the pass covers env tiles E0 .. E0 + NET - 1 of the wave's four (a 2-tile pass halves the working set of
// accumulators and output-layer operands; the role-split runner's matrix waves run two such passes per branch).
template <int NT, int NET, int E0, class TileOf, class OutStep>
__device__ __forceinline__ void mlp_layer2_split(const u32x4 (&bh)[4][4], const u32x4 (&bl)[4][4], int lane, int g, TileOf &&tile_of,
                                                 OutStep &&out_step)
{
    auto h2_tile = [&](int s2, f32x4 (&h2)[NET]) {
        const L2Tile t = tile_of(s2);
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(t.bias);
#pragma unroll
        for (int et = 0; et < NET; ++et) h2[et] = bias;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const bf16x8 ah = t.hi[p * 64 + lane], al = t.lo[p * 64 + lane];
#pragma unroll
            for (int et = 0; et < NET; ++et) {
                h2[et] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, as_bf16x8(bh[p][E0 + et]), h2[et], 0, 0, 0);
                h2[et] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, as_bf16x8(bl[p][E0 + et]), h2[et], 0, 0, 0);
                h2[et] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, as_bf16x8(bh[p][E0 + et]), h2[et], 0, 0, 0);
            }
        }
    };
    auto split_tile = [&](const f32x4 (&h2)[NET], int half, u32x4 (&ch)[NET], u32x4 (&cl)[NET]) {
#pragma unroll
        for (int et = 0; et < NET; ++et)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                uint32_t h, l;
                relu_split_pair(h2[et][2 * pr], h2[et][2 * pr + 1], h, l);
                ch[et][2 * half + pr] = h;
                cl[et][2 * half + pr] = l;
            }
    };
    u32x4 ch[NET], cl[NET];                                // split ReLU(H2) of a tile pair = one k-step of the output layer
#if QS_WEAVE == 1
    auto woven_stage = [&](int s2n, f32x4 (&hn)[NET], const f32x4 (&hc)[NET], int half) {
        const L2Tile t = tile_of(s2n);
        const bf16x8 *A2hi = t.hi + lane, *A2lo = t.lo + lane;
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(t.bias);
        bf16x8 ah[2], al[2];
        ah[0] = A2hi[0]; al[0] = A2lo[0];
        float v0[2 * NET], v1[2 * NET], h0[2 * NET], h1[2 * NET];
        uint32_t hu[2 * NET];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 12 * NET; ++m) {
            const int p = m / (3 * NET), term = (m % (3 * NET)) / NET, et = m % NET;
            if (m % (3 * NET) == 0 && p < 3) { ah[(p + 1) & 1] = A2hi[(p + 1) * 64]; al[(p + 1) & 1] = A2lo[(p + 1) * 64]; }
            const f32x4 cin = (p == 0 && term == 0) ? bias : hn[et];
            hn[et] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(term == 0 ? al[p & 1] : ah[p & 1],
                                                             as_bf16x8(term == 1 ? bl[p][E0 + et] : bh[p][E0 + et]), cin, 0, 0, 0);
            const int pair = m / 6, piece = m % 6, se = pair >> 1, pr = pair & 1;
            // (the element goes through a float temporary: __builtin_bit_cast applied to a vector-element lvalue reads element 0)
            if (piece == 0) { const float x = hc[se][2 * pr]; v0[pair] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0)); }
            if (piece == 1) { const float x = hc[se][2 * pr + 1]; v1[pair] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0)); }
            if (piece == 2) { const f32x2 v = {v0[pair], v1[pair]}; hu[pair] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2)); }
            if (piece == 3) h0[pair] = __builtin_bit_cast(float, hu[pair] << 16);
            if (piece == 4) h1[pair] = __builtin_bit_cast(float, hu[pair] & 0xffff0000u);
            if (piece == 5) {
                const f32x2 r = {v0[pair] - h0[pair], v1[pair] - h1[pair]};
                ch[se][2 * half + pr] = hu[pair];
                cl[se][2 * half + pr] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    f32x4 hcur[NET], hnext[NET];
    h2_tile(0, hcur);
#pragma unroll
    for (int s2 = 0; s2 < NT; ++s2) {
        if (s2 + 1 < NT) woven_stage(s2 + 1, hnext, hcur, s2 & 1);
        else split_tile(hcur, s2 & 1, ch, cl);
        if (s2 & 1) out_step(s2 >> 1, ch, cl);
#pragma unroll
        for (int et = 0; et < NET; ++et) hcur[et] = hnext[et];
    }
#else
#pragma unroll
    for (int s2 = 0; s2 < NT; ++s2) {
        f32x4 h2[NET];
        h2_tile(s2, h2);
        split_tile(h2, s2 & 1, ch, cl);
        if (s2 & 1) out_step(s2 >> 1, ch, cl);
    }
#endif
}

__device__ __forceinline__ void mlp_actor_fast(const float obs[12], float act[4], const char *blob, float *sObs, float *sAct,
                                               int lane)
{
    const bf16x8 *A2hi = reinterpret_cast<const bf16x8 *>(blob);
    const bf16x8 *A2lo = reinterpret_cast<const bf16x8 *>(blob + 32768);
    const bf16x8 *A1hi = reinterpret_cast<const bf16x8 *>(blob + 65536);
    const bf16x8 *A1lo = reinterpret_cast<const bf16x8 *>(blob + 65536 + 8192);
    const bf16x8 *A3hi = reinterpret_cast<const bf16x8 *>(blob + 65536 + 16384);
    const bf16x8 *A3lo = reinterpret_cast<const bf16x8 *>(blob + 65536 + 16384 + 4096);
    const float *sB1 = reinterpret_cast<const float *>(blob + 65536 + 16384 + 8192);
    const float *sB2 = sB1 + 128;
    const float *sB3 = sB2 + 128;
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int k = 0; k < 12; ++k) sObs[k * 64 + lane] = obs[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- layer 1 B operands: k-slot 8g+j = input 8g+j (12 inputs, the rest zero)
    bf16x8 xh[4], xl[4];
#pragma unroll
    for (int et = 0; et < 4; ++et)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * g + j;                       // g is per lane: guard by value, keep the read in range
            const float v = (k < 12) ? sObs[(k < 12 ? k : 0) * 64 + 16 * et + c] : 0.0f;
            __bf16 h, l;
            split_bf16(v, h, l);
            xh[et][j] = h; xl[et][j] = l;
        }
    // ---- layer 1: 8 row tiles, ReLU, straight into split B operands of layer 2 (tiles 2p, 2p+1 -> k-step p)
    u32x4 bh[4][4], bl[4][4];       // [p][et]: element j of the B operand = half (j & 1) of word j >> 1
#pragma unroll
    for (int rt = 0; rt < 8; ++rt) {
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(sB1 + 16 * rt + 4 * g);
        const bf16x8 ah = A1hi[rt * 64 + lane], al = A1lo[rt * 64 + lane];
        f32x4 acc[4] = {bias, bias, bias, bias};              // term outermost: consecutive MFMAs on different accumulators
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int et = 0; et < 4; ++et)
                acc[et] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(term == 0 ? al : ah, term == 1 ? xl[et] : xh[et], acc[et], 0, 0, 0);
#pragma unroll
        for (int et = 0; et < 4; ++et) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                uint32_t h, l;
                relu_split_pair(acc[et][2 * pr], acc[et][2 * pr + 1], h, l);
                bh[rt >> 1][et][2 * (rt & 1) + pr] = h;
                bl[rt >> 1][et][2 * (rt & 1) + pr] = l;
            }
        }
    }
    // ---- layer 2 in row-tile pairs, each pair folded into the action accumulators (layer 3 k-step = the pair)
    f32x4 a3[4];
    {
        const f32x4 bias3 = *reinterpret_cast<const f32x4 *>(sB3 + 4 * g);
#pragma unroll
        for (int et = 0; et < 4; ++et) a3[et] = bias3;
    }
    mlp_layer2_split<8, 4, 0>(bh, bl, lane, g,
        [&](int nt) { return L2Tile{A2hi + nt * 4 * 64, A2lo + nt * 4 * 64, sB2 + 16 * nt + 4 * g}; },
        [&](int q, const u32x4 (&ch)[4], const u32x4 (&cl)[4]) {
            const bf16x8 wh = A3hi[q * 64 + lane], wl = A3lo[q * 64 + lane];
#pragma unroll
            for (int term = 0; term < 3; ++term)            // term outermost: consecutive MFMAs on different accumulators
#pragma unroll
                for (int et = 0; et < 4; ++et)
                    a3[et] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(term == 0 ? wl : wh, as_bf16x8(term == 1 ? cl[et] : ch[et]), a3[et], 0, 0, 0);
        });
    if (g == 0) {
#pragma unroll
        for (int et = 0; et < 4; ++et) *reinterpret_cast<f32x4 *>(sAct + (16 * et + c) * 4) = a3[et];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const f32x4 av = *reinterpret_cast<const f32x4 *>(sAct + lane * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) act[i] = fminf(fmaxf(av[i], -1.0f), 1.0f);
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------------------
// Fast actor-critic heads (qs_runner_rollout_fast): mlp_actor_critic with the two 128x128 layers and the output layer
// on the bf16 matrix rate with split (hi + lo) operands, exactly as mlp_actor_fast; the 12-input first layer stays on
// the f32 MFMA (96 MFMAs, and its f32 weights are a third of the size of split fragments padded to k = 32).
// LDS image (bytes): A2pi hi 32768 | lo 32768 | A2vf hi 32768 | lo 32768 | A3pi hi 1024 | lo 1024 | A3vf hi 256 | lo 256 |
// W1^T f32 128x13 6656 | b1 512 | b2pi 512 | b2vf 512 | b3 64 | per-wave stage 4 x 3072  = 154 176 B.
// A3pi keeps only output rows 0..3, A3vf only row 4 of the 16-row tile: [k-step q][row][g][8 bf16] / [q][g][8 bf16].
constexpr int kAcFastA2 = 0;
constexpr int kAcFastA3p = 4 * 32768;
constexpr int kAcFastA3v = kAcFastA3p + 2 * 1024;
constexpr int kAcFastW1 = kAcFastA3v + 2 * 256;
constexpr int kAcFastB = kAcFastW1 + kHid * kLdW1 * 4;
constexpr int kAcFastBlobBytes = kAcFastB + (3 * kHid + 16) * 4;
constexpr int kAcFastLdsBytes = kAcFastBlobBytes + 4 * (12 * 64) * 4;

__device__ __forceinline__ void mlp_actor_critic_fast(const float obs[12], float out[5], const char *blob, float *stage,
                                                      int lane)
{
    const float *sW1 = reinterpret_cast<const float *>(blob + kAcFastW1);
    const float *sB1 = reinterpret_cast<const float *>(blob + kAcFastB);
    const float *sB2p = sB1 + kHid, *sB2v = sB2p + kHid, *sB3 = sB2v + kHid;
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int k = 0; k < 12; ++k) stage[k * 64 + lane] = obs[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- layer 1 on the f32 MFMA, ReLU, straight into split B operands of layer 2 (tiles 2p, 2p+1 -> k-step p)
    float xb[3][4];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int et = 0; et < 4; ++et) xb[s][et] = stage[(4 * s + g) * 64 + 16 * et + c];
    u32x4 bh[4][4], bl[4][4];       // [p][et]: element j of the B operand = word j >> 1
#pragma unroll
    for (int rt = 0; rt < 8; ++rt) {
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(sB1 + 16 * rt + 4 * g);
        float a[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) a[s] = sW1[(16 * rt + c) * kLdW1 + 4 * s + g];
        f32x4 acc[4] = {bias, bias, bias, bias};              // k-step outermost: consecutive MFMAs on different accumulators
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int et = 0; et < 4; ++et) acc[et] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], xb[s][et], acc[et], 0, 0, 0);
#pragma unroll
        for (int et = 0; et < 4; ++et) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                uint32_t h, l;
                relu_split_pair(acc[et][2 * pr], acc[et][2 * pr + 1], h, l);
                bh[rt >> 1][et][2 * (rt & 1) + pr] = h;
                bl[rt >> 1][et][2 * (rt & 1) + pr] = l;
            }
        }
    }
    f32x4 a3[4];
    {
        const f32x4 bias3 = *reinterpret_cast<const f32x4 *>(sB3 + 4 * g);
#pragma unroll
        for (int et = 0; et < 4; ++et) a3[et] = bias3;
    }
    // the 16 layer-2 row tiles: s2 = 8 br + nt; br 0: policy branch -> output rows 0..3, br 1: value branch -> row 4
    mlp_layer2_split<16, 4, 0>(bh, bl, lane, g,
        [&](int s2) {
            const int br = s2 >> 3, nt = s2 & 7;
            return L2Tile{reinterpret_cast<const bf16x8 *>(blob + kAcFastA2 + br * 65536) + nt * 4 * 64,
                          reinterpret_cast<const bf16x8 *>(blob + kAcFastA2 + br * 65536 + 32768) + nt * 4 * 64,
                          (br ? sB2v : sB2p) + 16 * nt + 4 * g};
        },
        [&](int pair, const u32x4 (&ch)[4], const u32x4 (&cl)[4]) {
            // output-layer A fragments: rows 0..3 (policy) / row 4 (value) hold weights, every other row of the 16-row tile
            // reads 16 zero bytes (slots 8..11 of the padded b3) -- an address select, no branch in the MFMA stream
            const int br = pair >> 2, q = pair & 3;
            const bf16x8 *ph = br == 0 ? reinterpret_cast<const bf16x8 *>(blob + kAcFastA3p) + (q * 4 + (c & 3)) * 4 + g
                                       : reinterpret_cast<const bf16x8 *>(blob + kAcFastA3v) + q * 4 + g;
            const bf16x8 *pl = br == 0 ? ph + 1024 / 16 : ph + 256 / 16;
            const bool row_live = br == 0 ? c < 4 : c == 4;
            const bf16x8 *pz = reinterpret_cast<const bf16x8 *>(sB3 + 8);
            const bf16x8 wh = *(row_live ? ph : pz), wl = *(row_live ? pl : pz);
#pragma unroll
            for (int term = 0; term < 3; ++term)            // term outermost: consecutive MFMAs on different accumulators
#pragma unroll
                for (int et = 0; et < 4; ++et)
                    a3[et] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(term == 0 ? wl : wh, as_bf16x8(term == 1 ? cl[et] : ch[et]), a3[et], 0, 0, 0);
        });
    if (g == 0) {
#pragma unroll
        for (int et = 0; et < 4; ++et) *reinterpret_cast<f32x4 *>(stage + (16 * et + c) * 8) = a3[et];
    } else if (g == 1) {
#pragma unroll
        for (int et = 0; et < 4; ++et) stage[(16 * et + c) * 8 + 4] = a3[et][0];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const f32x4 av = *reinterpret_cast<const f32x4 *>(stage + lane * 8);
    out[0] = av[0]; out[1] = av[1]; out[2] = av[2]; out[3] = av[3];
    out[4] = stage[lane * 8 + 4];
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------------------
// Role-split runner (k_runner_split): the heads cut into the pieces a MATRIX wave runs while an ENV wave of the same tile
// steps the environments.  ac_fast_layer1: shared layer for the wave's 64 envs (observations read from the tile's LDS
// stage, [k][env]), result = the B operands of BOTH 128 x 128 branches, kept in registers.  ac_fast_branch<BR>: one branch
// (0 policy -> rows 0..3 of a3, 1 value -> row 4) over QS_SPLIT_NET env tiles per pass (256 registers per wave at two waves
// per SIMD: the four-tile pass fits next to the 128 operand registers once the LDS addresses share base registers, lds_opaque).
// An LDS byte offset the optimiser cannot see through: everything addressed as `blob + opaque + constant` then shares ONE
// address register with the constants in the instructions' offset fields, instead of one materialised address per constant
// (the matrix wave has 256 registers for 128 operand registers plus its working set; such addresses went to scratch).
__device__ __forceinline__ int lds_opaque(int off)
{
    asm volatile("" : "+v"(off));
    return off;
}

__device__ __forceinline__ void ac_fast_layer1(const char *blob, const float *stage, int lane, u32x4 (&bh)[4][4], u32x4 (&bl)[4][4])
{
    lane = lds_opaque(lane);       // lane-derived addresses are re-derived here every step instead of living across the step loop
    const int c = lane & 15, g = lane >> 4;
    // one base register per array, the (row tile, k-step, env tile) part in the offset fields (see lds_opaque)
    const float *w1 = reinterpret_cast<const float *>(blob + lds_opaque(kAcFastW1 + (c * kLdW1 + g) * 4));   // + 16 rt kLdW1 + 4 s
    const float *b1 = reinterpret_cast<const float *>(blob + lds_opaque(kAcFastB + 16 * g));                  // + 16 rt
    const float *xs = stage + lds_opaque(g * 64 + c);                                                         // + 256 s + 16 et
    float xb[3][4];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int et = 0; et < 4; ++et) xb[s][et] = xs[256 * s + 16 * et];
#pragma unroll
    for (int rt = 0; rt < 8; ++rt) {
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(b1 + 16 * rt);
        float a[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) a[s] = w1[16 * rt * kLdW1 + 4 * s];
        f32x4 acc[4] = {bias, bias, bias, bias};              // k-step outermost: consecutive MFMAs on different accumulators
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int et = 0; et < 4; ++et) acc[et] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], xb[s][et], acc[et], 0, 0, 0);
#pragma unroll
        for (int et = 0; et < 4; ++et) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                uint32_t h, l;
                relu_split_pair(acc[et][2 * pr], acc[et][2 * pr + 1], h, l);
                bh[rt >> 1][et][2 * (rt & 1) + pr] = h;
                bl[rt >> 1][et][2 * (rt & 1) + pr] = l;
            }
        }
    }
}

#ifndef QS_SPLIT_NET
#define QS_SPLIT_NET 4      // env tiles per layer-2 pass of a matrix wave: 4 = one pass; 2 = two passes with half the working set,
                            // but only two accumulators in flight -- a dependent v_mfma_f32_16x16x32_bf16 two instructions behind
                            // its producer waits for it (measured 22 cycles per MFMA against 18.5 with four)
#endif
// zeros: byte offset (from blob) of 2 KiB of zeros -- what the dead rows of the output-layer A tiles read
template <int BR, int E0>
__device__ __forceinline__ void ac_fast_branch_pass(const char *blob, int zeros, const u32x4 (&bh)[4][4], const u32x4 (&bl)[4][4], int lane,
                                                    f32x4 (&a3)[4])
{
    lane = lds_opaque(lane);       // as in ac_fast_layer1
    const int c = lane & 15, g = lane >> 4;
    const char *bias_base = blob + lds_opaque(kAcFastB + 16 * g);                       // + 512 (1 + BR) + 64 nt
    // output-layer A fragments: rows 0..3 (policy, [q][row][g]) / row 4 (value, [q][g]) hold weights, every other row of the
    // 16-row tile reads zeros: ONE address select per pass, the k-step q and the hi / lo half in the offset field
    const bool row_live = BR == 0 ? c < 4 : c == 4;
    const int a3_off = BR == 0 ? kAcFastA3p + ((c & 3) * 4 + g) * 16 : kAcFastA3v + g * 16;
    const char *a3_base = blob + lds_opaque(row_live ? a3_off : zeros);
    mlp_layer2_split<8, QS_SPLIT_NET, E0>(bh, bl, lane, g,
        [&](int nt) {
            return L2Tile{reinterpret_cast<const bf16x8 *>(blob + kAcFastA2 + BR * 65536) + nt * 4 * 64,
                          reinterpret_cast<const bf16x8 *>(blob + kAcFastA2 + BR * 65536 + 32768) + nt * 4 * 64,
                          reinterpret_cast<const float *>(bias_base + 512 * (1 + BR) + 64 * nt)};
        },
        [&](int q, const u32x4 (&ch)[QS_SPLIT_NET], const u32x4 (&cl)[QS_SPLIT_NET]) {
            const bf16x8 wh = *reinterpret_cast<const bf16x8 *>(a3_base + q * (BR == 0 ? 256 : 64));
            const bf16x8 wl = *reinterpret_cast<const bf16x8 *>(a3_base + q * (BR == 0 ? 256 : 64) + (BR == 0 ? 1024 : 256));
#pragma unroll
            for (int term = 0; term < 3; ++term)            // term outermost: consecutive MFMAs on different accumulators
#pragma unroll
                for (int et = 0; et < QS_SPLIT_NET; ++et)
                    a3[E0 + et] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(term == 0 ? wl : wh, as_bf16x8(term == 1 ? cl[et] : ch[et]), a3[E0 + et], 0, 0, 0);
        });
}

template <int BR>
__device__ __forceinline__ void ac_fast_branch(const char *blob, int zeros, const u32x4 (&bh)[4][4], const u32x4 (&bl)[4][4], int lane,
                                               f32x4 (&a3)[4])
{
    const float *sB3 = reinterpret_cast<const float *>(blob + kAcFastB) + 3 * kHid;
    const f32x4 bias3 = *reinterpret_cast<const f32x4 *>(sB3 + 4 * (lane >> 4));
#pragma unroll
    for (int et = 0; et < 4; ++et) a3[et] = bias3;
    ac_fast_branch_pass<BR, 0>(blob, zeros, bh, bl, lane, a3);
    if (QS_SPLIT_NET == 2) ac_fast_branch_pass<BR, QS_SPLIT_NET == 2 ? 2 : 0>(blob, zeros, bh, bl, lane, a3);
}

// The exact-float32 heads in the same two pieces: identical MFMA sequences per accumulator as mlp_actor_critic -- the one
// difference is that each branch starts its own accumulators from the output bias instead of sharing one set, which adds the
// other branch's exact zeros in a different place (the same sums bit for bit).
__device__ __forceinline__ void ac_exact_layer1(const AcLds &L, const float *stage, int lane, f32x4 (&h1)[8][4])
{
    const int c = lane & 15, g = lane >> 4;
    float xb[3][4];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int et = 0; et < 4; ++et) xb[s][et] = stage[(4 * s + g) * 64 + 16 * et + c];
#pragma unroll
    for (int rt = 0; rt < 8; ++rt) {
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(L.B1 + 16 * rt + 4 * g);
        float a[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) a[s] = L.W1[(16 * rt + c) * kLdW1 + 4 * s + g];
        f32x4 acc[4] = {bias, bias, bias, bias};              // k-step outermost: consecutive MFMAs on different accumulators
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int et = 0; et < 4; ++et) acc[et] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], xb[s][et], acc[et], 0, 0, 0);
#pragma unroll
        for (int et = 0; et < 4; ++et) {
            h1[rt][et] = relu4(acc[et]);
        }
    }
}

template <int BR>
__device__ __forceinline__ void ac_exact_branch(const AcLds &L, const f32x4 (&h1)[8][4], int lane, f32x4 (&a3)[4])
{
    const int c = lane & 15, g = lane >> 4;
    const f32x4 bias3 = *reinterpret_cast<const f32x4 *>(L.B3 + 4 * g);
#pragma unroll
    for (int et = 0; et < 4; ++et) a3[et] = bias3;
    const float *W2 = (BR ? L.W2v : L.W2p) + c * kLdW + 4 * g;          // + 16 nt kLdW + 16 rt
    const float *B2 = (BR ? L.B2v : L.B2p) + 4 * g;                     // + 16 nt
    // output-layer A operand: row c of the 16-row tile; only rows 0..3 (policy) / row 4 (value) are non-zero -- every other
    // row reads 16 zero bytes (slots 8..11 of the padded b3): an address select, no branch in the MFMA stream
    const bool row_live = BR == 0 ? c < 4 : c == 4;
    const float *w3row = BR == 0 ? L.W3p + (c & 3) * kLdW + 4 * g : L.W3v + 4 * g;
    // Source order = issue order (scheduling barriers): the weights of k-group G + 1 (16 MFMAs = 512 matrix cycles later) are
    // requested before the MFMAs of group G, across tile boundaries too; left alone, hipcc emits read -> wait -> MFMAs and
    // exposes the LDS latency 4-5 times per tile (36.5 cycles per MFMA instead of the pipe's 32).
    f32x4 wbuf[2];
    wbuf[0] = *reinterpret_cast<const f32x4 *>(W2);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(B2 + 16 * nt);
        const f32x4 w3 = *reinterpret_cast<const f32x4 *>(row_live ? w3row + 16 * nt : L.B3 + 8);
        f32x4 h2[4] = {bias, bias, bias, bias};
#pragma unroll
        for (int rt = 0; rt < 8; ++rt) {
            const int G = 8 * nt + rt;
            if (G + 1 < 64) wbuf[(G + 1) & 1] = *reinterpret_cast<const f32x4 *>(W2 + 16 * ((G + 1) >> 3) * kLdW + 16 * ((G + 1) & 7));
            const f32x4 w = wbuf[G & 1];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int et = 0; et < 4; ++et)
                    h2[et] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i], h1[rt][et][i], h2[et], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 r[4];
#pragma unroll
        for (int et = 0; et < 4; ++et) r[et] = relu4(h2[et]);
#pragma unroll
        for (int i = 0; i < 4; ++i)                     // k-step outermost: consecutive MFMAs on different accumulators
#pragma unroll
            for (int et = 0; et < 4; ++et) a3[et] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3[i], r[et][i], a3[et], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace qs
