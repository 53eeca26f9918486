// quadsim_hip.hip -- kernels + C ABI of libquadsim_hip.so (gfx950 only).
//
// HBM layout of the persistent env state: AoSoA tiles of 64 envs (one
// wavefront).  Tile k holds kRecWords (40) field rows of 64 floats:
//     st[(k*40 + f)*64 + lane]
// so every field access of a wave is one fully coalesced 256-B segment and a
// tile is one contiguous 10 KiB block.  Per-env params (mass, Ixx, Iyy, Izz)
// live in a parallel [tile][4][64] array that is only read when the handle
// uses per-env params.  API-facing buffers keep the reference's row-major
// shapes ([N,4] actions, [N,12] obs, ...).
//
// See include/quadsim.h for the contract of each entry point and the reference
// file:line it replaces.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <hsa/amd_hsa_signal.h>

#include <elf.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

// -DQS_DEBUG: in-kernel bound checks of every global index (a failed check prints file:line and traps the wave); the
// release build compiles them away.  tools/build_debug.sh builds libquadsim_hip_dbg.so for QUADSIM_HIP_LIB=...
#ifdef QS_DEBUG
#include <cassert>
#define QS_ASSERT(c) assert(c)
#else
#define QS_ASSERT(c) ((void)0)
#endif

#include "../../include/quadsim.h"
#include "quadsim_device.hpp"
#include "rollout_ops.hpp"
#include "policy_rollout.hpp"

using namespace qs;

// -DQS_STAMP (diagnostic build, tools/build_stamp.sh): every workgroup of the role-split step kernel records the
// 100 MHz real-time counter at its phase boundaries into a caller-provided buffer (qs_debug_set_stamps), keyed by
// (step counter, tile) -- the in-kernel timeline of consecutive launches of the real chain.  Never in the product build.
#ifdef QS_STAMP
// The stamp buffer travels in StepArgs (not in a __device__ global): the private-queue launches run a second copy of the code
// object, loaded through HSA, whose globals HIP's hipMemcpyToSymbol never reaches.
// stamps stay in registers until the wave's last instruction: a store next to a barrier would be waited for by it
#define QS_STAMP_DECL unsigned long long stamp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; (void)stamp_
// -DQS_STAMP=2 ("light"): only the first and the last stamp of a wave.  The first stays in a register; the last -- taken when
// all of the wave's stores have been ISSUED, not drained -- is stored together with it by lane 0 (two 8-byte stores, never waited
// for).  The full build's eight scalar-memory round trips, its load-landed waits and, above all, its flush of eight stores
// BEHIND the drained wave end (a store-acknowledge latency on every workgroup's tail) cost ~0.7-0.9 us per step: too much for a
// timeline whose PERIOD is to be compared with the unstamped chain.  Even two stamps per wave cost ~0.4 us per step when every
// workgroup takes them (each is a scalar-memory round trip at the wave's head / tail), so only one workgroup in 64 does.
#if QS_STAMP + 0 >= 2
#define QS_STAMP_AT(slot)                                                                                       \
    do {                                                                                                        \
        /* no control flow near the loads: the first stamp is taken unconditionally (a scalar-memory read nobody waits */ \
        /* for until the wave's end); a branch here changes where the compiler waits for the state rows (+0.8 us)       */ \
        /* ... and it is taken at stamp site 1, BEHIND the issue of the wave's state loads (~50 ns after the wave's start): a  */ \
        /* scalar-memory read in front of them delays every later s_waitcnt lgkmcnt(0), i.e. the loads' addresses              */ \
        if ((slot) == 1) stamp_[0] = __builtin_amdgcn_s_memrealtime();                                          \
        else if ((slot) == (role == 0 ? 7 : 6)) {                                                               \
            if (lane == 0 && A.stamps && (tile & 63) == 0) {     /* one workgroup in 64 records */               \
                const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();                               \
                const unsigned long long ix_ = ((k0 % 64ull) * (unsigned long long)(A.stamp_tiles) + (unsigned long long)tile) * 16ull + 8 * role; \
                if (ix_ + 8 <= A.stamp_cap) { A.stamps[ix_] = stamp_[0]; A.stamps[ix_ + (slot)] = now_; }       \
            }                                                                                                   \
        }                                                                                                       \
    } while (0)
#else
#define QS_STAMP_AT(slot) (stamp_[slot] = __builtin_amdgcn_s_memrealtime())
#endif
#if QS_STAMP + 0 >= 2
#define QS_STAMP_FLUSH() ((void)0)
#else
#define QS_STAMP_FLUSH()                                                                                        \
    do {                                                                                                        \
        if (lane == 0 && A.stamps) {                                                                            \
            const unsigned long long ix_ = ((k0 % 64ull) * (unsigned long long)(A.stamp_tiles) + (unsigned long long)tile) * 16ull + 8 * role; \
            if (ix_ + 8 <= A.stamp_cap) for (int j_ = 0; j_ < 8; ++j_) A.stamps[ix_ + j_] = stamp_[j_];          \
        }                                                                                                       \
    } while (0)
#endif
// runner kernels: phase durations summed over the T steps of one launch, [tile][role][8] words
#define QS_PHASE_DECL unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_t_ = __builtin_amdgcn_s_memrealtime(); \
    const unsigned long long ph_c0_ = __builtin_amdgcn_s_memtime(), ph_r0_ = ph_t_
#define QS_PHASE(slot) do { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); ph_[slot] += n_ - ph_t_; ph_t_ = n_; } while (0)
#define QS_PHASE_FLUSH(role_)                                                                                   \
    do {                                                                                                        \
        ph_[6] = __builtin_amdgcn_s_memtime() - ph_c0_;          /* shader clocks ... */                         \
        ph_[7] = __builtin_amdgcn_s_memrealtime() - ph_r0_;      /* ... per 10 ns ticks = the clock frequency */ \
        if (lane == 0 && A.stamps) {                                                                            \
            const unsigned long long ix_ = (unsigned long long)tile * 16ull + 8 * (role_);                      \
            if (ix_ + 8 <= A.stamp_cap) for (int j_ = 0; j_ < 8; ++j_) A.stamps[ix_ + j_] = ph_[j_];             \
        }                                                                                                       \
    } while (0)
#else
#define QS_STAMP_DECL ((void)0)
#define QS_STAMP_AT(slot) ((void)0)
#define QS_STAMP_FLUSH() ((void)0)
#define QS_PHASE_DECL ((void)0)
#define QS_PHASE(slot) ((void)0)
#define QS_PHASE_FLUSH(role_) ((void)0)
#endif

// Store flavour of the step kernels' state rows and outputs: non-temporal (`nt`).  Every byte a step writes is consumed by a
// LATER launch (the next step, the policy), never by this one, and each launch ends with the write-back of the L2s' dirty lines:
// streaming stores leave that write-back less to do (65 536 envs: 6.92 -> 6.60 us per step; 131 072: 9.21 -> 8.87 us; plain
// stores with -DQS_PLAIN_STORES for A/B).  `sc1` write-through stores, in contrast, evict the lines and cost more than they save.
#if defined(QS_PLAIN_STORES)
#define QS_ST(p, v) (*(p) = (v))
#define QS_SO(p, v) QS_ST(p, v)
#else
#define QS_ST(p, v) __builtin_nontemporal_store((v), (p))
#define QS_SO(p, v) QS_ST(p, v)
#endif

namespace {

#ifndef QS_BLOCK
#define QS_BLOCK 256
#endif
constexpr int kBlock = QS_BLOCK;  // 4 wavefronts = 4 tiles per workgroup
// Where the rocRAND reset of a step is prepared in the role-split step kernel -- template parameter PREP of k_env_split, chosen
// per launch by the host (bit-identical either way):
//   0  the target wave hands over the Philox words, the chaser wave expands them inside its reset branch (0.6 us of its critical
//      path in 4 workgroups of 5: >= 1 of 64 lanes resets);
//   2  a THIRD wave per workgroup does the draw and the preparation (state, its observation, per-episode params -> LDS), every
//      step, and touches nothing else: the chaser wave's branch is a 25-word LDS copy and the target wave draws nothing.  Every
//      SIMD hosts one wave of each role (tools/wave_map3.hip) at less than half of its issue rate, so the third wave's ~370
//      instructions run beside the others: 65 536 envs 5.05 -> 4.71 us per step (one private queue), 4.6 -> 4.05 us (two),
//      HIP stream 6.6 -> 6.25 us; 32 768 envs 4.31 -> 3.76 us.
//   (1, the TARGET wave preparing it, was measured slower -- it becomes the long pole at barrier #2 -- and is gone.)
// The third wave needs residency: 112 VGPRs allow 4 waves per SIMD, i.e. 4 096 waves on the chip; a launch whose tiles x 3 waves
// exceed that runs its workgroups in two rounds (131 072 envs in ONE launch: 8.9 -> 10.7 us), so PREP = 2 is used up to
// kPrepMaxTiles tiles per launch (QS_RESET_PREP=0/2 forces one; profiles/r03/ab_experiments.txt section J).
#ifndef QS_PREP_MAX_TILES
#define QS_PREP_MAX_TILES 1365
#endif
constexpr int64_t kPrepMaxTiles = QS_PREP_MAX_TILES;
#ifndef QS_SPLIT_MAX_ENVS
#define QS_SPLIT_MAX_ENVS 131072
#endif
constexpr int64_t kSplitMaxEnvs = QS_SPLIT_MAX_ENVS;

struct StepArgs {
    float *st;             // [tiles][40][64]
    float *par;            // [tiles][4][64]
    const float *actions;  // [N,4] (step) / [T,N,4] (rollout) / nullptr (in-kernel random)
    float *obs;            // [N,12] / [T,N,12]
    float *reward;         // [N] / [T,N]
    uint8_t *done;
    uint8_t *flags;        // nullable
    float *term_obs;       // nullable, [N,12] (T == 1 only)
    float *term_state;     // nullable, [N,26] (T == 1 only): chaser 13 | target 13 of the terminal step (docking_env.py:226-229)
    float *slab;           // nullable: packed roll-out slab [T,N,14] = obs 12, reward, done (as 0/1); replaces obs/reward/done
    int64_t n;
    int64_t tile0, tile_end;   // tiles [tile0, tile_end) are stepped by this launch (an env group; the whole handle by default)
    int64_t dbg_shift;         // diagnostic (tests): workgroup b steps tile (b + dbg_shift) % tiles of its launch, i.e. on ANOTHER XCD
    int64_t io_env0, io_n;     // the I/O arrays start at env io_env0 and hold io_n envs per step (0, n: full-batch arrays)
    int64_t T;             // rollout length (1 for step)
    uint64_t step_idx;     // explicit step index (k_fill_actions); the env kernels read the device counter below
    unsigned long long *ctr;   // device: global step counter k, one copy per tile [tiles]
    uint64_t gid0;         // global id of env 0
    EnvConst C;
    RandCfg rc;
    Par par_nom;
    int auto_reset;
    int randomise;
    float nominal_obs[12]; // state2rel of the nominal reset states (what a non-randomised reset returns)
    const float *init;     // stored per-env initial states: [N][26] (docking: chaser, target) / [N][13] (hovering)
    // private-queue launches (qs_set_queue_mode): the launch carries no release fence, so a tile's state stays dirty in the L2 of
    // the XCD that stepped it; `owner` [tiles] records that XCD and every workgroup checks that it runs where its tile lives
    unsigned *owner;       // nullptr: ordinary (fenced) launch, no check.  One 32-bit word per tile, 0xffffffff = unowned;
                           // written and read with agent-scope atomics ONLY: like the state it guards, a plainly stored owner
                           // would stay dirty in the writing XCD's L2 and a misplaced workgroup would never see it
    unsigned *err;         // device word: bit 0 set when a workgroup found its tile owned by another XCD (it then touches nothing)
    unsigned long long *stamps;        // QS_STAMP builds: in-kernel timeline buffer (qs_debug_set_stamps), else nullptr
    unsigned long long stamp_cap, stamp_tiles;
};

__device__ __forceinline__ void load_env(const float *__restrict__ st, int64_t tile, int lane, Env &e)
{
    const float *b = st + tile * (int64_t)(kRecWords * kTile) + lane;
#pragma unroll
    for (int i = 0; i < 13; ++i) e.sc[i] = b[(F_SC + i) * kTile];
#pragma unroll
    for (int i = 0; i < 13; ++i) e.st[i] = b[(F_ST + i) * kTile];
#pragma unroll
    for (int i = 0; i < 4; ++i) e.uc[i] = b[(F_UC + i) * kTile];
#pragma unroll
    for (int i = 0; i < 4; ++i) e.ut[i] = b[(F_UT + i) * kTile];
#pragma unroll
    for (int i = 0; i < 4; ++i) e.qd[i] = b[(F_QD + i) * kTile];
    e.ls = b[F_LS * kTile];
    e.t = b[F_T * kTile];
}

__device__ __forceinline__ void store_env(float *__restrict__ st, int64_t tile, int lane, const Env &e)
{
    float *b = st + tile * (int64_t)(kRecWords * kTile) + lane;
#pragma unroll
    for (int i = 0; i < 13; ++i) QS_ST(&b[(F_SC + i) * kTile], e.sc[i]);
#pragma unroll
    for (int i = 0; i < 13; ++i) QS_ST(&b[(F_ST + i) * kTile], e.st[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) QS_ST(&b[(F_UC + i) * kTile], e.uc[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) QS_ST(&b[(F_UT + i) * kTile], e.ut[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) QS_ST(&b[(F_QD + i) * kTile], e.qd[i]);
    QS_ST(&b[F_LS * kTile], e.ls);
    QS_ST(&b[F_T * kTile], e.t);
}

__device__ __forceinline__ Par load_par(const float *__restrict__ par, int64_t tile, int lane)
{
    const float *b = par + tile * (int64_t)(kParWords * kTile) + lane;
    Par P;
    P.m = b[0]; P.Ixx = b[kTile]; P.Iyy = b[2 * kTile]; P.Izz = b[3 * kTile];
    return P;
}

__device__ __forceinline__ void store_par(float *__restrict__ par, int64_t tile, int lane, const Par &P)
{
    float *b = par + tile * (int64_t)(kParWords * kTile) + lane;
    b[0] = P.m; b[kTile] = P.Ixx; b[2 * kTile] = P.Iyy; b[3 * kTile] = P.Izz;
}

__device__ __forceinline__ void store_obs(float *__restrict__ obs, int64_t env, const float o[12])
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 *p = reinterpret_cast<f4 *>(obs + env * 12);
    QS_SO(&p[0], (f4{o[0], o[1], o[2], o[3]}));
    QS_SO(&p[1], (f4{o[4], o[5], o[6], o[7]}));
    QS_SO(&p[2], (f4{o[8], o[9], o[10], o[11]}));
}

// plain (cached) flavour: rows that are completed by LATER stores of the same lane (the env-major roll-out arrays, where a
// lane's consecutive steps fill consecutive slots of one line) should stay in the L2 until they are whole
__device__ __forceinline__ void store_obs_cached(float *__restrict__ obs, int64_t env, const float o[12])
{
    float4 *p = reinterpret_cast<float4 *>(obs + env * 12);
    p[0] = make_float4(o[0], o[1], o[2], o[3]);
    p[1] = make_float4(o[4], o[5], o[6], o[7]);
    p[2] = make_float4(o[8], o[9], o[10], o[11]);
}

// one env.step for the lane's env + VecEnv auto-reset; shared by step and rollout kernels
// The global step counter k lives in device memory so that a captured launch (hipGraph / torch.cuda.graphs) advances
// it on every replay.  It is kept PER TILE (one 64-bit word per wavefront's tile; all tiles hold the same value):
// a wave reads its own word at the start and writes k + T back at the end, so no workgroup ever waits for or
// races with another one.  (A single shared word updated through a per-workgroup ticket cost 2 us per launch.)
__device__ __forceinline__ uint64_t step_counter_begin(const StepArgs &A, int64_t tile) { return A.ctr[tile]; }
// The single-step kernels request the word through the vector memory path (the zero below hides the wave-uniform address from
// the compiler): as a scalar load it shared one counter -- and one wait -- with the kernel-argument fetch in front of the state
// loads, i.e. an L2 round trip on every wave's critical path; as a vector load it is one more load beside the state's
// (nominal-reset kernel 4.87 -> 4.74 us per step, two queues 4.41 -> 4.16; section J15)
__device__ __forceinline__ uint64_t step_counter_begin_vmem(const StepArgs &A, int64_t tile)
{
    int zero;
    asm("v_mov_b32 %0, 0" : "=v"(zero));
    return A.ctr[tile + zero];
}
__device__ __forceinline__ void step_counter_end(const StepArgs &A, int64_t tile, int lane, uint64_t k)
{
    if (lane == 0) A.ctr[tile] = k + (uint64_t)A.T;
}

// private-queue launches only: true when this wave must not touch its tile (the tile's latest state is in another XCD's L2).
// Blocks are dealt to the XCDs round-robin from a start that is constant for a queue (measured: tools/xcc_map.hip), so this
// never fires; it turns a change of that hardware behaviour into a loud error instead of stale state.
// The owner word is requested with an agent-scope atomic load (`sc1`: never served from a stale line of this XCD's L2 or this
// CU's L1) and claimed with an agent-scope compare-and-swap executed at the memory side: every XCD sees the same word.
constexpr unsigned kUnowned = 0xffffffffu;
__device__ __forceinline__ unsigned chain_owner_request(const StepArgs &A, int64_t tile)
{
    // no control flow around the load (an ordinary launch reads a word of its own step counter instead and ignores it): a
    // load inside a branch is issued late and waited for at the branch's end (section J15)
    const unsigned *p = A.owner ? A.owner + tile : reinterpret_cast<const unsigned *>(A.ctr + tile);
    const unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return A.owner ? v : kUnowned;
}

// the check for a caller that requested the owner word earlier (no load latency on its critical path)
__device__ __forceinline__ bool chain_owner_mismatch(const StepArgs &A, int64_t tile, int lane, unsigned own)
{
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;   // HW_REG_XCC_ID
    if (own == kUnowned) {
        // first private step of this tile since the handle's last HIP-side call: claim it (both role waves may try; the
        // second one finds its own XCD).  A claim lost to ANOTHER XCD is a misplacement like any other.
        unsigned seen = kUnowned;
        if (lane == 0) {
            unsigned expect = kUnowned;
            __hip_atomic_compare_exchange_strong(&A.owner[tile], &expect, xcc, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            seen = expect;
        }
        own = __builtin_amdgcn_readfirstlane(seen);
        if (own == kUnowned) return false;
    }
    if (own != xcc) {
        if (lane == 0) __hip_atomic_fetch_or(A.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // host memory
        return true;
    }
    return false;
}

__device__ __forceinline__ bool chain_tile_misplaced(const StepArgs &A, int64_t tile, int lane)
{
    if (!A.owner) return false;
    return chain_owner_mismatch(A, tile, lane, chain_owner_request(A, tile));
}

// tile of workgroup-local index b (0 <= b < the launch's tile count); dbg_shift != 0 only in the placement-guard test.
// Branch-free on purpose, with dbg_shift next to tile0 / tile_end in StepArgs: a branch on a kernel argument at the very top of
// the kernel made the compiler fetch that argument, wait, and only then fetch the rest -- a second scalar-memory round trip in
// front of every wave's loads (+0.3 us per step; profiles/r03/ab_experiments.txt section J15).
__device__ __forceinline__ int64_t launch_tile(const StepArgs &A, int64_t b)
{
    const int64_t nt = A.tile_end - A.tile0;
    b += A.dbg_shift;
    b -= (b >= nt) ? nt : 0;
    return A.tile0 + b;
}

template <bool PARAMS, int RMODE>
__device__ __forceinline__ void maybe_reset(Env &e, Par &P, const StepArgs &A, int64_t env, uint64_t k, float obs[12], unsigned flags,
                                            bool &done, bool write_term);

// RMODE (compile time) = the handle's `randomise`: 0 nominal reset, 1 rocRAND init state, 2 + params.
template <int INTEG, bool PARAMS, int RMODE>
__device__ __forceinline__ void step_and_maybe_reset(Env &e, Par &P, const float a[4], const StepArgs &A, int64_t env,
                                                     uint64_t k, float obs[12], float &reward, unsigned &flags,
                                                     bool &done, bool write_term)
{
    env_step<INTEG>(e, a, P, A.C, obs, reward, flags);
    maybe_reset<PARAMS, RMODE>(e, P, A, env, k, obs, flags, done, write_term);
}

template <bool PARAMS, int RMODE>
__device__ __forceinline__ void maybe_reset(Env &e, Par &P, const StepArgs &A, int64_t env, uint64_t k, float obs[12], unsigned flags,
                                            bool &done, bool write_term)
{
    done = (flags & (FLAG_OVERLIMIT | FLAG_OVERTIME)) != 0;
    if (done && A.auto_reset) {
        if (write_term && A.term_obs) store_obs(A.term_obs, env - A.io_env0, obs);
        if (write_term && A.term_state) {
            float *ts = A.term_state + (env - A.io_env0) * 26;
#pragma unroll
            for (int i = 0; i < 13; ++i) { ts[i] = e.sc[i]; ts[13 + i] = e.st[i]; }
        }
        if (RMODE == 0) {
            // nominal states are constants: no need to re-derive their observation per lane
            nominal_init(e.sc, e.st);
#pragma unroll
            for (int i = 0; i < 4; ++i) { e.uc[i] = 0.0f; e.ut[i] = 0.0f; }
            e.ls = 0.0f;
            e.t = 0.0f;
#pragma unroll
            for (int i = 0; i < 12; ++i) obs[i] = A.nominal_obs[i];
        } else if (RMODE == 3) {
            // stored per-env initial state (docking-v1; script-set chaser_ini_state)
            float ic[13], it[13];
            const float *src = A.init + env * 26;
#pragma unroll
            for (int i = 0; i < 13; ++i) { ic[i] = src[i]; it[i] = src[13 + i]; }
            env_reset<false>(e, ic, it, obs);
        } else {
            float ic[13], it[13];
            Par Pn;
            random_init<RMODE == 2>(A.rc, STREAM_AUTORESET, A.gid0 + (uint64_t)env, k + 1, ic, it, Pn);
            if (PARAMS && RMODE == 2) P = Pn;
            env_reset<true>(e, ic, it, obs);   // randomised reset states always have a level target
        }
    }
}

// K1/K4: T fused env.steps for N envs in one launch, env state in registers between the tile load
// and the tile store.  T == 1 is DockingEnv.step (docking_env.py:104-231); T > 1 is the trainer's
// Runner loop (rl_baselines/ppo2/ppo2.py:472-499) with the actions pre-staged or drawn in-kernel.
// One kernel serves both so that a roll-out is bit-identical to T single steps (same machine code).
template <int INTEG, bool PARAMS, int RMODE>
__global__ __launch_bounds__(kBlock) void k_env(StepArgs A)
{
    const int lane = threadIdx.x & (kTile - 1);
    const int64_t wg_tile = (int64_t)blockIdx.x * (kBlock / kTile) + (threadIdx.x >> 6);
    if (wg_tile >= A.tile_end - A.tile0) return;
    const int64_t tile = launch_tile(A, wg_tile);
    const int64_t env = tile * kTile + lane;
    if (env >= A.n) return;
    if (chain_tile_misplaced(A, tile, lane)) return;
    const int64_t io = env - A.io_env0;          // row of this env in the I/O arrays
    QS_ASSERT(io >= 0 && io < A.io_n);
    const uint64_t k0 = step_counter_begin_vmem(A, tile);
    // the first action is requested together with the tile (one exposed memory latency per launch, not two) and
    // every later one a whole step ahead of its use
    float4 av_next = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (A.actions) av_next = reinterpret_cast<const float4 *>(A.actions)[io];
    Env e;
    load_env(A.st, tile, lane, e);
    Par P = A.par_nom;
    if (PARAMS) P = load_par(A.par, tile, lane);
#pragma clang loop unroll(disable)
    for (int64_t t = 0; t < A.T; ++t) {
        const uint64_t k = k0 + (uint64_t)t;
        const int64_t o = t * A.io_n + io;
        float a[4];
        if (A.actions) {
            const float4 av = av_next;
            if (t + 1 < A.T) av_next = reinterpret_cast<const float4 *>(A.actions)[o + A.io_n];
            a[0] = av.x; a[1] = av.y; a[2] = av.z; a[3] = av.w;
        } else {
            random_action(A.rc.seed, A.gid0 + (uint64_t)env, k, a);
        }
        float obs[12], reward;
        unsigned flags;
        bool done;
        step_and_maybe_reset<INTEG, PARAMS, RMODE>(e, P, a, A, env, k, obs, reward, flags, done, true);
        if (A.slab) {
            // one 56-byte row per env-step (the unit the multi-GPU all-gather moves): seven 8-byte stores
            float2 *row = reinterpret_cast<float2 *>(A.slab + o * 14);
#pragma unroll
            for (int i = 0; i < 6; ++i) row[i] = make_float2(obs[2 * i], obs[2 * i + 1]);
            row[6] = make_float2(reward, done ? 1.0f : 0.0f);
        } else {
            store_obs(A.obs, o, obs);
            QS_SO(&A.reward[o], reward);
            QS_SO(&A.done[o], (uint8_t)(done ? 1 : 0));
        }
        if (A.flags) QS_SO(&A.flags[o], (uint8_t)flags);
    }
    store_env(A.st, tile, lane, e);
    if (PARAMS && RMODE == 2) store_par(A.par, tile, lane, P);
    step_counter_end(A, tile, lane, k0);
}

// Role-split variant of k_env: one workgroup = one tile = TWO waves.  Wave 0 carries the chaser side of the 64 envs (action
// mix, chaser drone step, state2rel, reward, done, the chaser's reset), wave 1 the target side (target drone step, the
// target's PID, the rocRAND draw a reset of this step would consume).  A lone wave issues a vector instruction every 4
// cycles and two waves on a SIMD every 2 each (MI355X_MICROARCH.md), so at one tile per SIMD the two half-length
// instruction streams run in the time of one.  Hand-overs through 7.5 KiB of LDS, two workgroup barriers per step:
//   target wave:  advance target | draw Philox words  -> #1 ->  PID, limit new control            -> #2 -> apply reset
//   chaser wave:  mix, advance chaser                 -> #1 ->  state2rel, reward, done -> flag  -> #2 -> reset, stores
// waves per workgroup of k_env_split: with PREP == 2 the rocRAND reset modes get a third wave
constexpr int split_waves(int rmode, int prep) { return (prep == 2 && (rmode == 1 || rmode == 2)) ? 3 : 2; }

template <int INTEG, bool PARAMS, int RMODE, int PREP>
__global__ __launch_bounds__(3 * kTile) void k_env_split(StepArgs A)
{
    __shared__ float s_tgt[13][kTile];
    // PREP == 2: [chaser reset state 13 | its observation 12 | per-episode params 4][lane]: what a reset of THIS step would
    // install, prepared every step off the chaser wave's critical path.  One buffer suffices in a roll-out too: it is written
    // between barriers #1 and #2 of a step and read behind #2; the next write is behind the NEXT step's #1, which the readers
    // have passed.  PREP == 0: [step parity][block]: the chaser wave reads step t's Philox words while t+1's are drawn.
    constexpr bool kPrep = PREP == 2 && (RMODE == 1 || RMODE == 2);
    __shared__ float s_rst[kPrep ? 29 : 1][kTile];
    __shared__ uint4 s_phx[kPrep ? 1 : 2][kPrep ? 1 : 2][kTile];
    __shared__ unsigned char s_done[kTile], s_limt[kTile];
    const int lane = threadIdx.x & (kTile - 1);
    const int role = threadIdx.x >> 6;
    const int64_t tile = launch_tile(A, blockIdx.x);   // grid = the tiles of this launch's env group
    const int64_t env = tile * kTile + lane;
    bool active = env < A.n;                     // idle lanes of the tail tile compute on zeros and store nothing
    const int64_t io = env - A.io_env0;          // row of this env in the I/O arrays
    QS_ASSERT(tile < A.tile_end && (!active || (io >= 0 && io < A.io_n)));
    // private-queue launches: the tile's owning XCD is requested here and examined only after the first compute phase (below),
    // so that the check costs no memory latency; a misplaced workgroup computes on whatever it loaded and stores nothing
    const unsigned owner_xcc = chain_owner_request(A, tile);
    const uint64_t k0 = step_counter_begin_vmem(A, tile);
    QS_STAMP_DECL;
    QS_STAMP_AT(0);
    const float *b = A.st + tile * (int64_t)(kRecWords * kTile) + lane;
    float *bw = A.st + tile * (int64_t)(kRecWords * kTile) + lane;
    Par P = A.par_nom;
    if (PARAMS) P = load_par(A.par, tile, lane);
    if (role == 0) {
        // in a roll-out the chaser wave is the long pole of every step while target waves on the same SIMD run ahead with
        // speculative draws: give it the issue slots first (roll-out 2.28 -> 2.13 us/step; no help for a single step)
        if (A.T > 1) __builtin_amdgcn_s_setprio(3);
        float sc[13], uc[4];
#pragma unroll
        for (int i = 0; i < 13; ++i) sc[i] = b[(F_SC + i) * kTile];
#pragma unroll
        for (int i = 0; i < 4; ++i) uc[i] = b[(F_UC + i) * kTile];
        float ls = b[F_LS * kTile], tt = b[F_T * kTile];
        // the action is requested LAST: loads return in issue order, and the action -- fresh from the caller, the one
        // operand that is not cache-resident -- is not needed before the integration (which uses the PREVIOUS limited
        // control, quadrotor.py:126-144) is done; its miss latency hides under drone_advance
        float4 av_next = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (A.actions && active) av_next = reinterpret_cast<const float4 *>(A.actions)[io];
#if defined(QS_STAMP) && QS_STAMP + 0 < 2
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
#endif
        QS_STAMP_AT(1);
#pragma clang loop unroll(disable)
        for (int64_t t = 0; t < A.T; ++t) {
            const uint64_t k = k0 + (uint64_t)t;
            const int64_t o = t * A.io_n + io;
            tt += 1.0f;
            const bool lim_c = drone_advance<INTEG>(sc, uc, P, A.C.dt);   // Drone.step's integration: previous control only
            if (A.owner && chain_owner_mismatch(A, tile, lane, owner_xcc)) active = false;
            float a[4];
            if (A.actions) {
                const float4 av = av_next;
                if (t + 1 < A.T && active) av_next = reinterpret_cast<const float4 *>(A.actions)[o + A.io_n];
                a[0] = av.x; a[1] = av.y; a[2] = av.z; a[3] = av.w;
            } else {
                random_action(A.rc.seed, A.gid0 + (uint64_t)env, k, a);
            }
            float u_c[4];
            chaser_command(a, P.m, u_c);
            u_limit(u_c, P.m * kG, uc);                                   // ... and the hand-over of the new limited control
            QS_STAMP_AT(2);
            __syncthreads();                                              // #1: the target's new state is in LDS
            QS_STAMP_AT(3);
            if (A.T == 1) __builtin_amdgcn_s_setprio(3);                 // single step: from here on this wave is the long pole
            float st[13];
#pragma unroll
            for (int i = 0; i < 13; ++i) st[i] = s_tgt[i][lane];
            const bool lim_t = s_limt[lane] != 0;
            float obs[12], reward;
            unsigned flags;
            rel_obs(sc, st, obs);
            score_step(obs, a, sc[2], tt, ls, A.C, lim_c, lim_t, reward, flags);
            const bool done = (flags & (FLAG_OVERLIMIT | FLAG_OVERTIME)) != 0;
            const bool rs = done && A.auto_reset;
            s_done[lane] = rs ? 1 : 0;
            QS_STAMP_AT(4);
            __syncthreads();                                              // #2: reset flags out, this step's Philox words in
            QS_STAMP_AT(5);
            if (rs) {
                if (A.term_obs && active) store_obs(A.term_obs, io, obs);
                if (A.term_state && active) {
                    float *ts = A.term_state + io * 26;
#pragma unroll
                    for (int i = 0; i < 13; ++i) ts[i] = sc[i];
                }
                float ic[13], it[13];
                if (RMODE == 0) {
                    nominal_init(ic, it);
#pragma unroll
                    for (int i = 0; i < 12; ++i) obs[i] = A.nominal_obs[i];
                } else if (RMODE == 3) {
                    const float *src = A.init + (active ? env : 0) * 26;
#pragma unroll
                    for (int i = 0; i < 13; ++i) { ic[i] = src[i]; it[i] = src[13 + i]; }
                    rel_obs<false>(ic, it, obs);
                } else {
                    if (kPrep) {
                        // the reset state, its observation and the episode's parameters were prepared by the third wave: a copy
#pragma unroll
                        for (int i = 0; i < 13; ++i) ic[i] = s_rst[i][lane];
#pragma unroll
                        for (int i = 0; i < 12; ++i) obs[i] = s_rst[13 + i][lane];
                        if (PARAMS && RMODE == 2) P = Par{s_rst[25][lane], s_rst[26][lane], s_rst[27][lane], s_rst[28][lane]};
                    } else {
                        const uint4 w0 = s_phx[t & 1][0][lane], w1 = RMODE == 2 ? s_phx[t & 1][1][lane] : make_uint4(0, 0, 0, 0);
                        Par Pn;
                        random_init_apply<RMODE == 2>(A.rc, w0, w1, ic, it, Pn);
                        if (PARAMS && RMODE == 2) P = Pn;
                        rel_obs<true>(ic, it, obs);
                    }
                }
#pragma unroll
                for (int i = 0; i < 13; ++i) sc[i] = ic[i];
#pragma unroll
                for (int i = 0; i < 4; ++i) uc[i] = 0.0f;
                ls = 0.0f;
                tt = 0.0f;
            }
            QS_STAMP_AT(6);
            if (active) {
                if (A.slab) {
                    float2 *row = reinterpret_cast<float2 *>(A.slab + o * 14);
#pragma unroll
                    for (int i = 0; i < 6; ++i) row[i] = make_float2(obs[2 * i], obs[2 * i + 1]);
                    row[6] = make_float2(reward, done ? 1.0f : 0.0f);
                } else {
                    store_obs(A.obs, o, obs);
                    QS_SO(&A.reward[o], reward);
                    QS_SO(&A.done[o], (uint8_t)(done ? 1 : 0));
                }
                if (A.flags) QS_SO(&A.flags[o], (uint8_t)flags);
            }
        }
        if (active) {
#pragma unroll
            for (int i = 0; i < 13; ++i) QS_ST(&bw[(F_SC + i) * kTile], sc[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) QS_ST(&bw[(F_UC + i) * kTile], uc[i]);
            QS_ST(&bw[F_LS * kTile], ls);
            QS_ST(&bw[F_T * kTile], tt);
            if (PARAMS && RMODE == 2) store_par(A.par, tile, lane, P);
        }
        if (!A.owner || active || env >= A.n) step_counter_end(A, tile, lane, k0);   // a misplaced tile's counter stays put, too
#if defined(QS_STAMP) && QS_STAMP + 0 < 2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        QS_STAMP_AT(7);
        QS_STAMP_FLUSH();
    } else if (role == 1) {
        float st[13], ut[4], qd[4];
#pragma unroll
        for (int i = 0; i < 13; ++i) st[i] = b[(F_ST + i) * kTile];
#pragma unroll
        for (int i = 0; i < 4; ++i) ut[i] = b[(F_UT + i) * kTile];
#pragma unroll
        for (int i = 0; i < 4; ++i) qd[i] = b[(F_QD + i) * kTile];
        const float pdes[3] = {10.0f, -50.0f, 5.0f};              // docking_env.py:60
        const float vdes[3] = {A.C.vdes_x, 0.0f, 0.0f};
        const float dv[3] = {0.0f, 0.0f, 0.0f};
        if (A.T == 1) __builtin_amdgcn_s_setprio(3);   // single step: the chaser wave waits at #1 for this wave's step + draw
#if defined(QS_STAMP) && QS_STAMP + 0 < 2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        QS_STAMP_AT(1);
#pragma clang loop unroll(disable)
        for (int64_t t = 0; t < A.T; ++t) {
            const uint64_t k = k0 + (uint64_t)t;
            float pre[13];
#pragma unroll
            for (int i = 0; i < 13; ++i) pre[i] = st[i];
            const bool lim_t = drone_advance<INTEG>(st, ut, P, A.C.dt);   // with the previous limited control
            if (A.owner && chain_owner_mismatch(A, tile, lane, owner_xcc)) active = false;
#pragma unroll
            for (int i = 0; i < 13; ++i) s_tgt[i][lane] = st[i];
            s_limt[lane] = lim_t ? 1 : 0;
            uint4 w0 = make_uint4(0, 0, 0, 0), w1 = w0;
            Par Pn = P;
            if ((RMODE == 1 || RMODE == 2) && !kPrep) {
                random_init_words<RMODE == 2>(A.rc, STREAM_AUTORESET, A.gid0 + (uint64_t)env, k + 1, w0, w1);
                s_phx[t & 1][0][lane] = w0;
                if (RMODE == 2) s_phx[t & 1][1][lane] = w1;     // the params block: only drawn with per-episode params
            }
            QS_STAMP_AT(2);
            __syncthreads();                                              // #1
            QS_STAMP_AT(3);
            if (A.T == 1) __builtin_amdgcn_s_setprio(0);
            float u_t[4];
            target_control(A.C.kind, pdes, vdes, qd, 0.0f, pre, dv, P.m, u_t);   // from the state BEFORE stepping
            u_limit(u_t, P.m * kG, ut);
            QS_STAMP_AT(4);
            __syncthreads();                                              // #2
            QS_STAMP_AT(5);
            if (s_done[lane]) {
                if (A.term_state && active) {
                    float *ts = A.term_state + io * 26 + 13;
#pragma unroll
                    for (int i = 0; i < 13; ++i) ts[i] = st[i];
                }
                float ic[13], it[13];
                if (RMODE == 3) {
                    const float *src = A.init + (active ? env : 0) * 26;
#pragma unroll
                    for (int i = 0; i < 13; ++i) it[i] = src[13 + i];
                } else {
                    nominal_init(ic, it);
                    if (PARAMS && RMODE == 2) {
                        if (kPrep) {
                            Pn = Par{s_rst[25][lane], s_rst[26][lane], s_rst[27][lane], s_rst[28][lane]};
                        } else {
                            random_init_apply<true>(A.rc, w0, w1, ic, it, Pn);
                            nominal_init(ic, it);
                        }
                        P = Pn;
                    }
                }
#pragma unroll
                for (int i = 0; i < 13; ++i) st[i] = it[i];
#pragma unroll
                for (int i = 0; i < 4; ++i) ut[i] = 0.0f;
            }
        }
        if (active) {
#pragma unroll
            for (int i = 0; i < 13; ++i) QS_ST(&bw[(F_ST + i) * kTile], st[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) QS_ST(&bw[(F_UT + i) * kTile], ut[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) QS_ST(&bw[(F_QD + i) * kTile], qd[i]);
        }
        QS_STAMP_AT(6);
        QS_STAMP_FLUSH();
    }
    else if (kPrep) {
        // third wave (rocRAND reset modes only): what a reset of each step would install -- the draw, random_init_apply and the
        // state2rel of the result: the same device functions the serial kernel runs inside its reset branch, so the same bits --
        // for EVERY lane, into LDS; it touches no global memory but the step counter and joins both barriers of every step.  The
        // chaser wave's reset branch is a 25-word copy, the target wave draws nothing.
#pragma clang loop unroll(disable)
        for (int64_t t = 0; t < A.T; ++t) {
            const uint64_t k = k0 + (uint64_t)t;
            uint4 w0, w1 = make_uint4(0, 0, 0, 0);
            random_init_words<RMODE == 2>(A.rc, STREAM_AUTORESET, A.gid0 + (uint64_t)env, k + 1, w0, w1);
            __syncthreads();                                              // #1
            float ic[13], it_[13], robs[12];
            Par Pn;
            random_init_apply<RMODE == 2>(A.rc, w0, w1, ic, it_, Pn);
            rel_obs<true>(ic, it_, robs);
#pragma unroll
            for (int i = 0; i < 13; ++i) s_rst[i][lane] = ic[i];
#pragma unroll
            for (int i = 0; i < 12; ++i) s_rst[13 + i][lane] = robs[i];
            if (PARAMS && RMODE == 2) {
                s_rst[25][lane] = Pn.m; s_rst[26][lane] = Pn.Ixx; s_rst[27][lane] = Pn.Iyy; s_rst[28][lane] = Pn.Izz;
            }
            __syncthreads();                                              // #2
        }
    }
}

// Policy-in-the-loop roll-out: T steps of  a = clip(MLP(obs));  obs, r, done = env.step(a)  in one launch
// (run_trained_docking_ppo2.py:37-60 for N envs).  MLP on exact-f32 MFMA (policy_rollout.hpp), env step = the
// device code of k_env.  obs_0 is derived from the stored state (an observation is always state2rel of the state).
template <int INTEG, int RMODE>
__global__ __launch_bounds__(kBlock, 1) void k_policy_rollout(StepArgs A, MlpArgs M, float *__restrict__ actions_out)
{
    __shared__ __attribute__((aligned(16))) float lds[policy_lds_floats()];
    float *sW2 = lds;
    float *sW3 = sW2 + kHid * kLdW;
    float *sW1 = sW3 + 16 * kLdW;
    float *sB1 = sW1 + kHid * kLdW1;
    float *sB2 = sB1 + kHid;
    float *sB3 = sB2 + kHid;
    float *sObsAll = sB3 + 16;
    float *sActAll = sObsAll + 4 * (12 * 64);
    // weights -> LDS (W3^T rows 4..15 and b3[4..15] are zero padding of the 16-row MFMA tile)
    for (int i = threadIdx.x; i < kHid * kHid; i += kBlock) sW2[(i >> 7) * kLdW + (i & 127)] = M.wt2[i];
    for (int i = threadIdx.x; i < 16 * kHid; i += kBlock) sW3[(i >> 7) * kLdW + (i & 127)] = (i >> 7) < 4 ? M.wt3[i] : 0.0f;
    for (int i = threadIdx.x; i < kHid * 12; i += kBlock) sW1[(i / 12) * kLdW1 + (i % 12)] = M.wt1[i];
    for (int i = threadIdx.x; i < kHid; i += kBlock) { sB1[i] = M.b1[i]; sB2[i] = M.b2[i]; }
    if (threadIdx.x < 16) sB3[threadIdx.x] = threadIdx.x < 4 ? M.b3[threadIdx.x] : 0.0f;
    __syncthreads();

    const int lane = threadIdx.x & (kTile - 1);
    const int w = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * (kBlock / kTile) + w;
    const int64_t env = tile * kTile + lane;
    const bool active = env < A.n;             // MFMA needs the whole wave: idle lanes carry a nominal env, store nothing
    float *sObs = sObsAll + w * (12 * 64), *sAct = sActAll + w * (64 * 4);
    Env e;
    if (active) load_env(A.st, tile, lane, e);
    else { nominal_init(e.sc, e.st); for (int i = 0; i < 4; ++i) { e.uc[i] = 0.0f; e.ut[i] = 0.0f; e.qd[i] = i == 0; } e.ls = 0.0f; e.t = 0.0f; }
    Par P = A.par_nom;
    const uint64_t k0 = active ? step_counter_begin(A, tile) : 0;
    float obs[12];
    rel_obs(e.sc, e.st, obs);
#pragma clang loop unroll(disable)
    for (int64_t t = 0; t < A.T; ++t) {
        float a[4];
        mlp_actor(obs, a, sW1, sB1, sW2, sB2, sW3, sB3, sObs, sAct, lane);
        float reward;
        unsigned flags;
        bool done;
        step_and_maybe_reset<INTEG, false, RMODE>(e, P, a, A, active ? env : 0, k0 + (uint64_t)t, obs, reward, flags, done, false);
        if (active) {
            const int64_t o = t * A.n + env;
            store_obs(A.obs, o, obs);
            A.reward[o] = reward;
            A.done[o] = done ? 1 : 0;
            if (A.flags) A.flags[o] = (uint8_t)flags;
            if (actions_out) reinterpret_cast<float4 *>(actions_out)[o] = make_float4(a[0], a[1], a[2], a[3]);
        }
    }
    if (active) { store_env(A.st, tile, lane, e); step_counter_end(A, tile, lane, k0); }
}

// The same roll-out with the actor on the bf16 matrix rate and split (hi + lo) operands: policy_rollout.hpp,
// "Fast actor".  `blob` = the host-packed weight image (kFastBlobBytes), copied verbatim into LDS.
template <int INTEG, int RMODE>
__global__ __launch_bounds__(kBlock, 1) void k_policy_rollout_fast(StepArgs A, const uint4 *__restrict__ blob, float *__restrict__ actions_out)
{
    __shared__ __attribute__((aligned(16))) char lds[kFastBlobBytes + 4 * (12 * 64 + 64 * 4) * 4];
    for (int i = threadIdx.x; i < kFastBlobBytes / 16; i += kBlock) reinterpret_cast<uint4 *>(lds)[i] = blob[i];
    __syncthreads();
    const int lane = threadIdx.x & (kTile - 1);
    const int w = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * (kBlock / kTile) + w;
    const int64_t env = tile * kTile + lane;
    const bool active = env < A.n;
    float *stage = reinterpret_cast<float *>(lds + kFastBlobBytes);
    float *sObs = stage + w * (12 * 64), *sAct = stage + 4 * (12 * 64) + w * (64 * 4);
    Env e;
    if (active) load_env(A.st, tile, lane, e);
    else { nominal_init(e.sc, e.st); for (int i = 0; i < 4; ++i) { e.uc[i] = 0.0f; e.ut[i] = 0.0f; e.qd[i] = i == 0; } e.ls = 0.0f; e.t = 0.0f; }
    Par P = A.par_nom;
    const uint64_t k0 = active ? step_counter_begin(A, tile) : 0;
    float obs[12];
    rel_obs(e.sc, e.st, obs);
#pragma clang loop unroll(disable)
    for (int64_t t = 0; t < A.T; ++t) {
        float a[4];
        mlp_actor_fast(obs, a, lds, sObs, sAct, lane);
        float reward;
        unsigned flags;
        bool done;
        step_and_maybe_reset<INTEG, false, RMODE>(e, P, a, A, active ? env : 0, k0 + (uint64_t)t, obs, reward, flags, done, false);
        if (active) {
            const int64_t o = t * A.n + env;
            store_obs(A.obs, o, obs);
            A.reward[o] = reward;
            A.done[o] = done ? 1 : 0;
            if (A.flags) A.flags[o] = (uint8_t)flags;
            if (actions_out) reinterpret_cast<float4 *>(actions_out)[o] = make_float4(a[0], a[1], a[2], a[3]);
        }
    }
    if (active) { store_env(A.st, tile, lane, e); step_counter_end(A, tile, lane, k0); }
}

// PPO2 data collection in one launch: the Runner loop of rl_baselines/ppo2/ppo2.py:472-499 (+ last_values, :506) for
// N envs and T = n_steps.  Per step: mb_obs <- obs; (mean, value) <- MLP heads on exact-f32 MFMA; action = mean +
// std * N(0,1) (rocRAND Philox + Box-Muller, or caller-supplied noise); neglogp of the diagonal Gaussian
// (common/distributions.py:406-410); env.step(clip(action, -1, 1)); mb_dones holds the done flags BEFORE the step
// (ppo2.py:479), rewards / the new done after it.  squash: the fork's tanh variant (common/policies.py:238-242,
// distributions.py:412-415): env gets tanh(u), neglogp += sum log(1 - tanh(u)^2 + 1e-6), mb_actions keeps u.
struct RunnerArgs {
    AcArgs net;
    float std[4], inv_std[4];
    float nl_const;            // 0.5 log(2 pi) * 4 + sum(logstd)
    int squash;
    const float *noise;        // nullable [T,N,4]
    const uint8_t *dones_in;   // nullable [N]: done flags carried over from the previous run
    const uint4 *blob;         // FAST only: packed split-bf16 weight image (kAcFastBlobBytes)
    float *actions;            // [T,N,4]
    float *values;             // [T,N]
    float *neglogp;            // [T,N]
    float *last_obs;           // nullable [N,12]
    float *last_values;        // [N]
    uint8_t *last_dones;       // [N]
    int env_major;             // mb_obs / mb_actions rows at env*T + t (already swap_and_flatten-ed) instead of t*N + env
};

// FAST: the networks on the bf16 matrix rate with split operands (mlp_actor_critic_fast; R.blob = host-packed image)
// PARAMS: per-env mass / inertia (domain randomisation; RMODE 2 redraws them at every episode start)
template <int INTEG, int RMODE, bool PARAMS, bool FAST>
__global__ __launch_bounds__(kBlock, 1) void k_runner_rollout(StepArgs A, RunnerArgs R)
{
    __shared__ __attribute__((aligned(16))) char lds_raw[FAST ? kAcFastLdsBytes : (int)(ac_lds_floats() * sizeof(float))];
    float *lds = reinterpret_cast<float *>(lds_raw);
    AcLds L{};
    float *sStage;
    if (FAST) {
        for (int i = threadIdx.x; i < kAcFastBlobBytes / 16; i += kBlock) reinterpret_cast<uint4 *>(lds_raw)[i] = R.blob[i];
        sStage = reinterpret_cast<float *>(lds_raw + kAcFastBlobBytes);
    } else {
        float *sW2p = lds;
        float *sW2v = sW2p + kHid * kLdW;
        float *sW3p = sW2v + kHid * kLdW;
        float *sW3v = sW3p + 4 * kLdW;
        float *sW1 = sW3v + kLdW;
        float *sB1 = sW1 + kHid * kLdW1;
        float *sB2p = sB1 + kHid;
        float *sB2v = sB2p + kHid;
        float *sB3 = sB2v + kHid;
        sStage = sB3 + 16;
        for (int i = threadIdx.x; i < kHid * kHid; i += kBlock) {
            sW2p[(i >> 7) * kLdW + (i & 127)] = R.net.wt2[i];
            sW2v[(i >> 7) * kLdW + (i & 127)] = R.net.wtv2[i];
        }
        for (int i = threadIdx.x; i < 4 * kHid; i += kBlock) sW3p[(i >> 7) * kLdW + (i & 127)] = R.net.wt3[i];
        for (int i = threadIdx.x; i < kHid; i += kBlock) sW3v[i] = R.net.wtv3[i];
        for (int i = threadIdx.x; i < kHid * 12; i += kBlock) sW1[(i / 12) * kLdW1 + (i % 12)] = R.net.wt1[i];
        for (int i = threadIdx.x; i < kHid; i += kBlock) { sB1[i] = R.net.b1[i]; sB2p[i] = R.net.b2[i]; sB2v[i] = R.net.bv2[i]; }
        if (threadIdx.x < 16) sB3[threadIdx.x] = threadIdx.x < 4 ? R.net.b3[threadIdx.x] : (threadIdx.x == 4 ? R.net.bv3[0] : 0.0f);
        L = AcLds{sW1, sB1, sW2p, sB2p, sW2v, sB2v, sW3p, sW3v, sB3};
    }
    __syncthreads();

    const int lane = threadIdx.x & (kTile - 1);
    const int w = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * (kBlock / kTile) + w;
    const int64_t env = tile * kTile + lane;
    const bool active = env < A.n;             // MFMA needs the whole wave: idle lanes carry a nominal env, store nothing
    float *stage = sStage + w * (12 * 64);
    QS_ASSERT((char *)(stage + 12 * 64) <= lds_raw + sizeof lds_raw);
    Env e;
    if (active) load_env(A.st, tile, lane, e);
    else { nominal_init(e.sc, e.st); for (int i = 0; i < 4; ++i) { e.uc[i] = 0.0f; e.ut[i] = 0.0f; e.qd[i] = i == 0; } e.ls = 0.0f; e.t = 0.0f; }
    Par P = A.par_nom;
    if (PARAMS && active) P = load_par(A.par, tile, lane);
    const uint64_t k0 = active ? step_counter_begin(A, tile) : 0;
    bool done_prev = (active && R.dones_in) ? R.dones_in[env] != 0 : false;
    float obs[12];
    rel_obs(e.sc, e.st, obs);
#pragma clang loop unroll(disable)
    for (int64_t t = 0; t < A.T; ++t) {
        const int64_t o = t * A.n + env;
        QS_ASSERT(!active || (o >= 0 && o < A.T * A.n));
        // the two wide arrays can be written env-major right away (ppo2.py:522-523 flattens them afterwards anyway): a
        // lane's consecutive steps then fill consecutive 48- / 16-byte slots of its own row, which the XCD's L2 merges
        const int64_t ow = R.env_major ? env * A.T + t : o;
        if (active) { if (R.env_major) store_obs_cached(A.obs, ow, obs); else store_obs(A.obs, ow, obs); }   // mb_obs: the observation the policy acts on
        float head[5];
        if (FAST) mlp_actor_critic_fast(obs, head, lds_raw, stage, lane);
        else mlp_actor_critic(obs, head, L, stage, lane);
        float eps[4];
        if (R.noise) {
            const float4 nv = active ? reinterpret_cast<const float4 *>(R.noise)[o] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            eps[0] = nv.x; eps[1] = nv.y; eps[2] = nv.z; eps[3] = nv.w;
        } else {
            random_normal4(A.rc.seed, A.gid0 + (uint64_t)(active ? env : 0), k0 + (uint64_t)t, eps);
        }
        float u[4], a[4];
        float nl = R.nl_const;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u[i] = fmaf(R.std[i], eps[i], head[i]);                   // distributions.py:429
            const float d = (u[i] - head[i]) * R.inv_std[i];          // :407
            nl = fmaf(0.5f * d, d, nl);
        }
        if (R.squash) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float sech2;
                a[i] = q_tanh(u[i], sech2);                           // policies.py:238
                nl += q_ln(sech2 + 1e-6f);                            // distributions.py:414, 1 - tanh(u)^2 + 1e-6
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = fminf(fmaxf(u[i], -1.0f), 1.0f);   // ppo2.py:483
        }
        if (active) {
            reinterpret_cast<float4 *>(R.actions)[ow] = make_float4(u[0], u[1], u[2], u[3]);
            R.values[o] = head[4];
            R.neglogp[o] = nl;
            A.done[o] = done_prev ? 1 : 0;                            // mb_dones: flags before the step (ppo2.py:479)
        }
        float reward;
        unsigned flags;
        bool done;
        step_and_maybe_reset<INTEG, PARAMS, RMODE>(e, P, a, A, active ? env : 0, k0 + (uint64_t)t, obs, reward, flags, done, false);
        done_prev = done;
        if (active) {
            A.reward[o] = reward;
            if (A.flags) A.flags[o] = (uint8_t)flags;
        }
    }
    // last_values = model.value(obs) on the observation after the last step (ppo2.py:506)
    float head[5];
    if (FAST) mlp_actor_critic_fast(obs, head, lds_raw, stage, lane);
    else mlp_actor_critic(obs, head, L, stage, lane);
    if (active) {
        R.last_values[env] = head[4];
        R.last_dones[env] = done_prev ? 1 : 0;
        if (R.last_obs) store_obs(R.last_obs, env, obs);
        store_env(A.st, tile, lane, e);
        if (PARAMS && RMODE == 2) store_par(A.par, tile, lane, P);
        step_counter_end(A, tile, lane, k0);
    }
}

// Role-split variant of the Runner kernel: one workgroup = four tiles = EIGHT waves.  Waves 0..3 ("matrix" role,
// one per SIMD) only evaluate the networks, waves 4..7 ("env" role, wave 4 + i next to wave i) own the environment state of
// the same four tiles: sampling, neglogp, env.step, every mb_* store except the values.  Per step and tile
//   env wave:     obs -> LDS | draw N(0,1), target's half of env.step -> #b -> sample, neglogp, stores, chaser's half, new obs -> LDS -> #a
//   matrix wave:  -> #a -> layer 1, policy branch, means -> LDS       -> #b -> value branch, store value
// so the value branch (almost half of a step's MFMAs) and the env step (VALU) run at the same time on the same SIMD, and
// the matrix wave keeps no environment registers: both roles fit 256 registers, two waves per SIMD.  The means travel
// through the tile's obs stage (the matrix wave has its observations in registers by then), the values through a
// buffer private to the matrix wave.  Every wave passes the same 2 T + 1 workgroup barriers.  FAST as in k_runner_rollout;
// the heads are the same instruction sequences on the same operands as there, so the two kernels agree bit for bit.
template <int INTEG, int RMODE, bool PARAMS, bool FAST>
__global__ __launch_bounds__(2 * kBlock, 1) void k_runner_split(StepArgs A, RunnerArgs R)
{
    constexpr int kHeadBytes = FAST ? kAcFastLdsBytes : (int)(ac_lds_floats() * sizeof(float));     // weights + 4 obs stages
    constexpr int kZeros = kHeadBytes + 4 * kTile * 4;                                              // FAST: 2 KiB of zeros
    __shared__ __attribute__((aligned(16))) char lds_raw[kZeros + (FAST ? 2048 : 0)];
    AcLds L{};
    float *sStage;
    if (FAST) {
        for (int i = threadIdx.x; i < kAcFastBlobBytes / 16; i += 2 * kBlock) reinterpret_cast<uint4 *>(lds_raw)[i] = R.blob[i];
        if (threadIdx.x < 128) reinterpret_cast<uint4 *>(lds_raw + kZeros)[threadIdx.x] = make_uint4(0, 0, 0, 0);
        sStage = reinterpret_cast<float *>(lds_raw + kAcFastBlobBytes);
    } else {
        float *sW2p = reinterpret_cast<float *>(lds_raw);
        float *sW2v = sW2p + kHid * kLdW;
        float *sW3p = sW2v + kHid * kLdW;
        float *sW3v = sW3p + 4 * kLdW;
        float *sW1 = sW3v + kLdW;
        float *sB1 = sW1 + kHid * kLdW1;
        float *sB2p = sB1 + kHid;
        float *sB2v = sB2p + kHid;
        float *sB3 = sB2v + kHid;
        sStage = sB3 + 16;
        for (int i = threadIdx.x; i < kHid * kHid; i += 2 * kBlock) {
            sW2p[(i >> 7) * kLdW + (i & 127)] = R.net.wt2[i];
            sW2v[(i >> 7) * kLdW + (i & 127)] = R.net.wtv2[i];
        }
        for (int i = threadIdx.x; i < 4 * kHid; i += 2 * kBlock) sW3p[(i >> 7) * kLdW + (i & 127)] = R.net.wt3[i];
        for (int i = threadIdx.x; i < kHid; i += 2 * kBlock) sW3v[i] = R.net.wtv3[i];
        for (int i = threadIdx.x; i < kHid * 12; i += 2 * kBlock) sW1[(i / 12) * kLdW1 + (i % 12)] = R.net.wt1[i];
        for (int i = threadIdx.x; i < kHid; i += 2 * kBlock) { sB1[i] = R.net.b1[i]; sB2p[i] = R.net.b2[i]; sB2v[i] = R.net.bv2[i]; }
        if (threadIdx.x < 16) sB3[threadIdx.x] = threadIdx.x < 4 ? R.net.b3[threadIdx.x] : (threadIdx.x == 4 ? R.net.bv3[0] : 0.0f);
        L = AcLds{sW1, sB1, sW2p, sB2p, sW2v, sB2v, sW3p, sW3v, sB3};
    }
    __syncthreads();
    const int lane = threadIdx.x & (kTile - 1);
    const int w = (threadIdx.x >> 6) & 3;
    const bool matrix_role = threadIdx.x < kBlock;
    const int64_t tile = (int64_t)blockIdx.x * (kBlock / kTile) + w;
    const int64_t env = tile * kTile + lane;
    const bool active = env < A.n;             // MFMA needs the whole wave: idle lanes carry a nominal env, store nothing
    float *stage = sStage + w * (12 * 64);
    float *sval = reinterpret_cast<float *>(lds_raw + kHeadBytes) + w * kTile;
    QS_ASSERT((char *)(stage + 12 * 64) <= lds_raw + kHeadBytes);
    if (matrix_role) {
        const int c = lane & 15, g = lane >> 4;
        // layer-1 result = the B operands of both 128 x 128 branches, 128 registers either way
        u32x4 bh[FAST ? 4 : 1][4], bl[FAST ? 4 : 1][4];
        f32x4 h1[FAST ? 1 : 8][4];
        f32x4 a3[4];
        QS_PHASE_DECL;
#pragma clang loop unroll(disable)
        for (int64_t t = 0; t <= A.T; ++t) {
            __syncthreads();                                                  // #a: this step's observations are in LDS
            QS_PHASE(0);
            if constexpr (FAST) ac_fast_layer1(lds_raw, stage, lane, bh, bl);
            else ac_exact_layer1(L, stage, lane, h1);
            QS_PHASE(1);
            if (t < A.T) {
                if constexpr (FAST) ac_fast_branch<0>(lds_raw, kZeros, bh, bl, lane, a3);
                else ac_exact_branch<0>(L, h1, lane, a3);
                if (g == 0) {
#pragma unroll
                    for (int et = 0; et < 4; ++et) *reinterpret_cast<f32x4 *>(stage + (16 * et + c) * 8) = a3[et];
                }
                QS_PHASE(2);
                __syncthreads();                                              // #b: the means are in LDS
                QS_PHASE(3);
            }
            if constexpr (FAST) ac_fast_branch<1>(lds_raw, kZeros, bh, bl, lane, a3);
            else ac_exact_branch<1>(L, h1, lane, a3);
            QS_PHASE(4);
            if (g == 1) {
#pragma unroll
                for (int et = 0; et < 4; ++et) sval[16 * et + c] = a3[et][0];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const float v = sval[lane];
            __builtin_amdgcn_wave_barrier();
            if (active) {
                float *vout = t < A.T ? R.values + t * A.n : R.last_values;   // last: model.value(obs) after the last step (ppo2.py:506)
                vout[env] = v;
            }
            QS_PHASE(5);
        }
        QS_PHASE_FLUSH(0);
    } else {
        Env e;
        if (active) load_env(A.st, tile, lane, e);
        else { nominal_init(e.sc, e.st); for (int i = 0; i < 4; ++i) { e.uc[i] = 0.0f; e.ut[i] = 0.0f; e.qd[i] = i == 0; } e.ls = 0.0f; e.t = 0.0f; }
        Par P = A.par_nom;
        if (PARAMS && active) P = load_par(A.par, tile, lane);
        const uint64_t k0 = active ? step_counter_begin(A, tile) : 0;
        bool done_prev = (active && R.dones_in) ? R.dones_in[env] != 0 : false;
        float obs[12];
        rel_obs(e.sc, e.st, obs);
#pragma unroll
        for (int k = 0; k < 12; ++k) stage[k * 64 + lane] = obs[k];
        QS_PHASE_DECL;
#pragma clang loop unroll(disable)
        for (int64_t t = 0; t < A.T; ++t) {
            const int64_t o = t * A.n + env;
            QS_ASSERT(!active || (o >= 0 && o < A.T * A.n));
            const int64_t ow = R.env_major ? env * A.T + t : o;
            if (active) { if (R.env_major) store_obs_cached(A.obs, ow, obs); else store_obs(A.obs, ow, obs); }
            QS_PHASE(0);
            __syncthreads();                                                  // #a
            QS_PHASE(1);
            float eps[4];
            if (R.noise) {
                const float4 nv = active ? reinterpret_cast<const float4 *>(R.noise)[o] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                eps[0] = nv.x; eps[1] = nv.y; eps[2] = nv.z; eps[3] = nv.w;
            } else {
                random_normal4(A.rc.seed, A.gid0 + (uint64_t)(active ? env : 0), k0 + (uint64_t)t, eps);
            }
            // the target's half of env.step does not need the action: it runs here, next to the policy branch
            const bool lim_t = env_step_target<INTEG>(e, P, A.C);
            QS_PHASE(2);
            __syncthreads();                                                  // #b
            QS_PHASE(3);
            const f32x4 mean = *reinterpret_cast<const f32x4 *>(stage + lane * 8);
            float u[4], a[4];
            float nl = R.nl_const;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u[i] = fmaf(R.std[i], eps[i], mean[i]);                       // distributions.py:429
                const float d = (u[i] - mean[i]) * R.inv_std[i];              // :407
                nl = fmaf(0.5f * d, d, nl);
            }
            if (R.squash) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float sech2;
                    a[i] = q_tanh(u[i], sech2);                               // policies.py:238
                    nl += q_ln(sech2 + 1e-6f);                                // distributions.py:414
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = fminf(fmaxf(u[i], -1.0f), 1.0f);   // ppo2.py:483
            }
            if (active) {
                reinterpret_cast<float4 *>(R.actions)[ow] = make_float4(u[0], u[1], u[2], u[3]);
                R.neglogp[o] = nl;
                A.done[o] = done_prev ? 1 : 0;                                // mb_dones: flags before the step (ppo2.py:479)
            }
            QS_PHASE(4);
            float reward;
            unsigned flags;
            bool done;
            env_step_chaser<INTEG>(e, a, P, A.C, lim_t, obs, reward, flags);
            maybe_reset<PARAMS, RMODE>(e, P, A, active ? env : 0, k0 + (uint64_t)t, obs, flags, done, false);
            done_prev = done;
#pragma unroll
            for (int k = 0; k < 12; ++k) stage[k * 64 + lane] = obs[k];
            if (active) {
                A.reward[o] = reward;
                if (A.flags) A.flags[o] = (uint8_t)flags;
            }
            QS_PHASE(5);
        }
        QS_PHASE_FLUSH(1);
        __syncthreads();                                                      // #a of the value-only pass
        if (active) {
            R.last_dones[env] = done_prev ? 1 : 0;
            if (R.last_obs) store_obs(R.last_obs, env, obs);
            store_env(A.st, tile, lane, e);
            if (PARAMS && RMODE == 2) store_par(A.par, tile, lane, P);
            step_counter_end(A, tile, lane, k0);
        }
    }
}

// hovering-v0 (HoveringEnv.step, hovering_env.py:47-78): T fused steps, one drone per lane.  Uses rows F_SC..
// (state) and F_UC.. (last limited control) of the tile; obs [T,N,13] = state after the step (or the stored
// ini_state after an auto-reset, hovering_env.py:80-82).
template <int INTEG, bool PARAMS>
__global__ __launch_bounds__(kBlock) void k_hover(StepArgs A)
{
    const int lane = threadIdx.x & (kTile - 1);
    const int64_t tile = A.tile0 + (int64_t)blockIdx.x * (kBlock / kTile) + (threadIdx.x >> 6);
    const int64_t env = tile * kTile + lane;
    if (tile >= A.tile_end || env >= A.n) return;
    const int64_t io = env - A.io_env0;
    QS_ASSERT(io >= 0 && io < A.io_n);
    const uint64_t k0 = step_counter_begin(A, tile);
    float *b = A.st + tile * (int64_t)(kRecWords * kTile) + lane;
    float s[13], up[4];
#pragma unroll
    for (int i = 0; i < 13; ++i) s[i] = b[(F_SC + i) * kTile];
#pragma unroll
    for (int i = 0; i < 4; ++i) up[i] = b[(F_UC + i) * kTile];
    Par P = A.par_nom;
    if (PARAMS) P = load_par(A.par, tile, lane);
#pragma clang loop unroll(disable)
    for (int64_t t = 0; t < A.T; ++t) {
        const int64_t o = t * A.io_n + io;
        float a[4];
        if (A.actions) {
            const float4 av = reinterpret_cast<const float4 *>(A.actions)[o];
            a[0] = av.x; a[1] = av.y; a[2] = av.z; a[3] = av.w;
        } else {
            random_action(A.rc.seed, A.gid0 + (uint64_t)env, k0 + (uint64_t)t, a);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = 0.5f * a[i] + 0.5f;    // hovering actions live in [0,1]
        }
        float reward;
        unsigned flags;
        hover_step<INTEG>(s, up, a, P, A.C.dt, reward, flags);
        const bool done = (flags & FLAG_OVERLIMIT) != 0;
        if (done && A.auto_reset) {
            if (A.term_obs) for (int i = 0; i < 13; ++i) A.term_obs[io * 13 + i] = s[i];
            const float *src = A.init + env * 13;
#pragma unroll
            for (int i = 0; i < 13; ++i) s[i] = src[i];
#pragma unroll
            for (int i = 0; i < 4; ++i) up[i] = 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 13; ++i) A.obs[o * 13 + i] = s[i];
        A.reward[o] = reward;
        A.done[o] = done ? 1 : 0;
        if (A.flags) A.flags[o] = (uint8_t)flags;
    }
#pragma unroll
    for (int i = 0; i < 13; ++i) b[(F_SC + i) * kTile] = s[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[(F_UC + i) * kTile] = up[i];
    step_counter_end(A, tile, lane, k0);
}

// construction-time jitter of docking-v1 (imitating_docking_env.py:34: chaser pos += U(-0.3,0.3)^3) and
// hovering-v0 (hovering_env.py:23-24: pos = (0,0,5)+U(-1,1)^3, att = euler2quat(U(-0.2,0.2)^3)), drawn from
// the rocRAND INIT stream (ctr 0) instead of numpy's global RNG; same 16-bit lattice as random_init.
__global__ __launch_bounds__(kBlock) void k_ctor_init(float *init, int64_t n, int hover, uint64_t seed, uint64_t gid0)
{
    const int64_t env = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (env >= n) return;
    uint4 w = philox_block(seed, STREAM_CTOR, gid0 + (uint64_t)env, 0);
    if (!hover) {
        float *d = init + env * 26;
        for (int i = 0; i < 26; ++i) d[i] = 0.0f;
        d[0] = __fmaf_rn(sym(u16lo(w.x)), 0.3f, 8.0f);
        d[1] = __fmaf_rn(sym(u16hi(w.x)), 0.3f, -50.0f);
        d[2] = __fmaf_rn(sym(u16lo(w.y)), 0.3f, 5.0f);
        d[6] = 1.0f;
        d[13] = 10.0f; d[14] = -50.0f; d[15] = 5.0f; d[19] = 1.0f;
    } else {
        float *d = init + env * 13;
        for (int i = 0; i < 13; ++i) d[i] = 0.0f;
        d[0] = sym(u16lo(w.x));
        d[1] = sym(u16hi(w.x));
        d[2] = __fmaf_rn(sym(u16lo(w.y)), 1.0f, 5.0f);
        float e0 = sym(u16hi(w.y)) * 0.2f, e1 = sym(u16lo(w.z)) * 0.2f, e2 = sym(u16hi(w.z)) * 0.2f;
        float sr, cr, sp, cp, sy, cy;
        q_sincos_small(0.5f * e0, sr, cr);
        q_sincos_small(0.5f * e1, sp, cp);
        q_sincos_small(0.5f * e2, sy, cy);
        euler2quat_trig(sr, cr, sp, cp, sy, cy, d + 6);
    }
}

__global__ __launch_bounds__(kBlock) void k_fill_init_nominal(float *init, int64_t n)
{
    const int64_t env = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (env >= n) return;
    float sc[13], st[13];
    nominal_init(sc, st);
    for (int i = 0; i < 13; ++i) { init[env * 26 + i] = sc[i]; init[env * 26 + 13 + i] = st[i]; }
}

// K2: masked reset (DockingEnv.reset, docking_env.py:233-244); init_all also rewrites q_des, like __init__
__global__ __launch_bounds__(kBlock) void k_reset(StepArgs A, const uint8_t *__restrict__ mask, int init_all)
{
    const int lane = threadIdx.x & (kTile - 1);
    const int64_t tile = (int64_t)blockIdx.x * (kBlock / kTile) + (threadIdx.x >> 6);
    const int64_t env = tile * kTile + lane;
    if (env >= A.n) return;
    if (mask && !mask[env]) return;
    Env e;
    load_env(A.st, tile, lane, e);
    float ic[13], it[13], obs[12];
    if (A.init) {
        const float *src = A.init + env * 26;
        for (int i = 0; i < 13; ++i) { ic[i] = src[i]; it[i] = src[13 + i]; }
    } else if (A.randomise) {
        Par Pn;
        random_init<true>(A.rc, STREAM_RESET, A.gid0 + (uint64_t)env, A.ctr[tile], ic, it, Pn);
        if (A.randomise >= 2) store_par(A.par, tile, lane, Pn);
    } else {
        nominal_init(ic, it);
    }
    if (init_all) { e.qd[0] = 1.0f; e.qd[1] = 0.0f; e.qd[2] = 0.0f; e.qd[3] = 0.0f; }
    env_reset(e, ic, it, obs);
    store_env(A.st, tile, lane, e);
    if (A.obs) store_obs(A.obs, env, obs);
}

// HoveringEnv.reset (hovering_env.py:80-82): state <- stored ini_state, last control <- 0; obs = the state
__global__ __launch_bounds__(kBlock) void k_hover_reset(StepArgs A, const uint8_t *__restrict__ mask)
{
    const int lane = threadIdx.x & (kTile - 1);
    const int64_t tile = (int64_t)blockIdx.x * (kBlock / kTile) + (threadIdx.x >> 6);
    const int64_t env = tile * kTile + lane;
    if (env >= A.n) return;
    if (mask && !mask[env]) return;
    float *b = A.st + tile * (int64_t)(kRecWords * kTile) + lane;
    const float *src = A.init + env * 13;
    for (int i = 0; i < 13; ++i) { b[(F_SC + i) * kTile] = src[i]; if (A.obs) A.obs[env * 13 + i] = src[i]; }
    for (int i = 0; i < 4; ++i) b[(F_UC + i) * kTile] = 0.0f;
}

__global__ __launch_bounds__(kBlock) void k_fill_ctr(unsigned long long *ctr, int64_t tiles, unsigned long long k)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < tiles) ctr[i] = k;
}

__global__ void k_nominal_obs(float *out)
{
    float sc[13], st[13], o[12];
    nominal_init(sc, st);
    rel_obs(sc, st, o);
    for (int i = 0; i < 12; ++i) out[i] = o[i];
}

__global__ __launch_bounds__(kBlock) void k_fill_par(float *par, int64_t n, Par P)
{
    const int lane = threadIdx.x & (kTile - 1);
    const int64_t tile = (int64_t)blockIdx.x * (kBlock / kTile) + (threadIdx.x >> 6);
    if (tile * kTile + lane >= n) return;
    store_par(par, tile, lane, P);
}

__global__ __launch_bounds__(kBlock) void k_fill_actions(float *__restrict__ actions, int64_t n, int64_t T, uint64_t seed,
                                                         uint64_t gid0, uint64_t step0)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n * T) return;
    const int64_t t = i / n, env = i - t * n;
    float a[4];
    random_action(seed, gid0 + (uint64_t)env, step0 + (uint64_t)t, a);
    reinterpret_cast<float4 *>(actions)[i] = make_float4(a[0], a[1], a[2], a[3]);
}

// AoS <-> AoSoA conversion for qs_get_state / qs_set_state / params
struct StateIO {
    float *chaser, *target, *u_prev, *qdes, *ls, *t;
};
template <bool TO_USER>
__global__ __launch_bounds__(kBlock) void k_state_io(float *st, int64_t n, StateIO io)
{
    const int lane = threadIdx.x & (kTile - 1);
    const int64_t tile = (int64_t)blockIdx.x * (kBlock / kTile) + (threadIdx.x >> 6);
    const int64_t env = tile * kTile + lane;
    if (env >= n) return;
    float *b = st + tile * (int64_t)(kRecWords * kTile) + lane;
    auto mv = [&](float *user, int f) {
        if (!user) return;
        if (TO_USER) *user = b[f * kTile];
        else b[f * kTile] = *user;
    };
    for (int i = 0; i < 13; ++i) mv(io.chaser ? io.chaser + env * 13 + i : nullptr, F_SC + i);
    for (int i = 0; i < 13; ++i) mv(io.target ? io.target + env * 13 + i : nullptr, F_ST + i);
    for (int i = 0; i < 8; ++i) mv(io.u_prev ? io.u_prev + env * 8 + i : nullptr, F_UC + i);
    for (int i = 0; i < 4; ++i) mv(io.qdes ? io.qdes + env * 4 + i : nullptr, F_QD + i);
    mv(io.ls ? io.ls + env : nullptr, F_LS);
    mv(io.t ? io.t + env : nullptr, F_T);
}

template <bool TO_USER>
__global__ __launch_bounds__(kBlock) void k_par_io(float *par, int64_t n, float *mass, float *inertia)
{
    const int lane = threadIdx.x & (kTile - 1);
    const int64_t tile = (int64_t)blockIdx.x * (kBlock / kTile) + (threadIdx.x >> 6);
    const int64_t env = tile * kTile + lane;
    if (env >= n) return;
    float *b = par + tile * (int64_t)(kParWords * kTile) + lane;
    if (TO_USER) {
        if (mass) mass[env] = b[0];
        if (inertia) for (int i = 0; i < 3; ++i) inertia[env * 3 + i] = b[(1 + i) * kTile];
    } else {
        if (mass) b[0] = mass[env];
        if (inertia) for (int i = 0; i < 3; ++i) b[(1 + i) * kTile] = inertia[env * 3 + i];
    }
}

// ---- layer-1 kernels on row-major user arrays ------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_drone_step(int64_t n, float *state, float *u_prev, const float *u,
                                                       const float *par, uint8_t *limited, Par par_nom, float dt,
                                                       int integ)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float s[13], up[4], uu[4];
    for (int j = 0; j < 13; ++j) s[j] = state[i * 13 + j];
    for (int j = 0; j < 4; ++j) { up[j] = u_prev[i * 4 + j]; uu[j] = u[i * 4 + j]; }
    Par P = par_nom;
    if (par) { P.m = par[i * 4]; P.Ixx = par[i * 4 + 1]; P.Iyy = par[i * 4 + 2]; P.Izz = par[i * 4 + 3]; }
    bool over = integ == 0 ? drone_step<0>(s, up, uu, P, dt) : drone_step<1>(s, up, uu, P, dt);
    for (int j = 0; j < 13; ++j) state[i * 13 + j] = s[j];
    for (int j = 0; j < 4; ++j) u_prev[i * 4 + j] = up[j];
    if (limited) limited[i] = over ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void k_ctrl(int64_t n, int mode, float *state_des, const float *state,
                                                 const float *state_last, float mass, float *u_out)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float sd[13], s[13], dv[3] = {0.0f, 0.0f, 0.0f}, u[4];
    for (int j = 0; j < 13; ++j) { sd[j] = state_des[i * 13 + j]; s[j] = state[i * 13 + j]; }
    if (mode == 1 && state_last) for (int j = 0; j < 3; ++j) dv[j] = s[3 + j] - state_last[i * 13 + 3 + j];
    target_control(mode, sd, sd + 3, sd + 6, sd[12], s, dv, mass, u);
    for (int j = 0; j < 4; ++j) { state_des[i * 13 + 6 + j] = sd[6 + j]; u_out[i * 4 + j] = u[j]; }
    state_des[i * 13 + 10] = 0.0f;   // roll_rate_des,  PIDController.py:101
    state_des[i * 13 + 11] = 0.0f;   // pitch_rate_des, PIDController.py:102
}

// PID expert (run_expert_policy.py:49-69, run_expert_record.py:121-136): vel_controller on the chaser towards
// 0.2 m behind the target, inverse action map (inv(rotor2control) u - mean)/std, not clipped.  Reads the envs'
// current chaser / target state straight from the tiles; state_des [N][13] is the expert's persistent desired
// state (pos = chaser start, vel = des_vel, [6:12] rewritten by the controller).  First step of an episode
// (t == 0) keeps the previous des_vel (:58-59).
template <bool PARAMS>
__global__ __launch_bounds__(kBlock) void k_expert_action(const float *__restrict__ st, const float *__restrict__ par, int64_t n,
                                                          float *__restrict__ state_des, float kp, float kd, Par par_nom,
                                                          float *__restrict__ actions)
{
    const int lane = threadIdx.x & (kTile - 1);
    const int64_t tile = (int64_t)blockIdx.x * (kBlock / kTile) + (threadIdx.x >> 6);
    const int64_t env = tile * kTile + lane;
    if (env >= n) return;
    const float *b = st + tile * (int64_t)(kRecWords * kTile) + lane;
    float sc[13], tp[3], sd[13];
    for (int i = 0; i < 13; ++i) sc[i] = b[(F_SC + i) * kTile];
    for (int i = 0; i < 3; ++i) tp[i] = b[(F_ST + i) * kTile];
    const float t = b[F_T * kTile];
    for (int i = 0; i < 13; ++i) sd[i] = state_des[env * 13 + i];
    Par P = par_nom;
    if (PARAMS) P = load_par(par, tile, lane);
    if (t != 0.0f) {
        sd[3] = kp * (tp[0] - 0.2f - sc[0]) + kd * (-sc[3]);
        sd[4] = kp * (tp[1] - sc[1]) + kd * (-sc[4]);
        sd[5] = kp * (tp[2] - sc[2]) + kd * (-sc[5]);
    }
    const float dv[3] = {0.0f, 0.0f, 0.0f};        // state_last aliases the current state
    float u[4];
    target_control(1, sd, sd + 3, sd + 6, sd[12], sc, dv, P.m, u);
    sd[10] = 0.0f; sd[11] = 0.0f;
    constexpr float a = 1.0f / (2.0f * kL), bq = 1.0f / (4.0f * kLambda);
    const float f4 = 0.25f * u[0];
    const float f0 = f4 - a * u[2] + bq * u[3], f1 = f4 + a * u[1] - bq * u[3];
    const float f2 = f4 + a * u[2] + bq * u[3], f3 = f4 - a * u[1] - bq * u[3];
    const float inv_mean = q_rcp(0.5f * P.m * kG);
    reinterpret_cast<float4 *>(actions)[env] = make_float4(f0 * inv_mean - 1.0f, f1 * inv_mean - 1.0f, f2 * inv_mean - 1.0f,
                                                           f3 * inv_mean - 1.0f);
    for (int i = 3; i < 12; ++i) state_des[env * 13 + i] = sd[i];
}

// layer 0: utils/transform.py as batch functions.  op 0 quat2euler [n,4]->[n,3] (:94-120), 1 euler2quat [n,3]->[n,4]
// (:123-136), 2 quat2rot [n,4]->[n,9] (:4-20), 3 rot2euler [n,9]->[n,3] (:23-46)
__global__ __launch_bounds__(kBlock) void k_transform(int op, int64_t n, const float *in, float *out)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    if (op == 0) {
        float q[4] = {in[i * 4], in[i * 4 + 1], in[i * 4 + 2], in[i * 4 + 3]};
        quat2euler(q, out[i * 3], out[i * 3 + 1], out[i * 3 + 2]);
    } else if (op == 1) {
        float q[4];
        euler2quat(in[i * 3], in[i * 3 + 1], in[i * 3 + 2], q);
        for (int j = 0; j < 4; ++j) out[i * 4 + j] = q[j];
    } else if (op == 2) {
        float q[4] = {in[i * 4], in[i * 4 + 1], in[i * 4 + 2], in[i * 4 + 3]};
        Rot R = quat2rot(q);
        const float r[9] = {1.0f, R.r01, R.r02, R.r10, 1.0f, R.r12, R.r20, R.r21, 1.0f};
        for (int j = 0; j < 9; ++j) out[i * 9 + j] = r[j];
    } else {
        const float *R = in + i * 9;
        const float r12 = R[5];
        const bool sat = (r12 >= 1.0f) || (r12 < -1.0f);
        out[i * 3] = q_asin(fminf(fmaxf(r12, -1.0f), 1.0f));
        out[i * 3 + 1] = sat ? 0.0f : q_atan2(-R[2], R[8]);
        out[i * 3 + 2] = q_atan2(-R[3], R[4]);
    }
}

__global__ __launch_bounds__(kBlock) void k_rel_obs(int64_t n, const float *chaser, const float *target, float *obs)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float sc[13], st[13], o[12];
    for (int j = 0; j < 13; ++j) { sc[j] = chaser[i * 13 + j]; st[j] = target[i * 13 + j]; }
    rel_obs(sc, st, o);
    for (int j = 0; j < 12; ++j) obs[i * 12 + j] = o[j];
}

// ---------------------------------------------------------------------------------------------
thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return fail(QS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// roctx ranges around the hot entry points (trace readability under rocprofv3 --marker-trace): resolved at run time and
// only when QS_ROCTX=1, so the library carries no link-time dependency on a profiler library
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    bool on = false;
};
inline Roctx &roctx()
{
    static Roctx r = [] {
        Roctx x;
        const char *en = getenv("QS_ROCTX");
        if (en && atoi(en)) {
            void *h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
            if (h) {
                x.push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
                x.pop = (int (*)())dlsym(h, "roctxRangePop");
                x.on = x.push && x.pop;
            }
        }
        return x;
    }();
    return r;
}
struct Range {
    bool on;
    explicit Range(const char *name) : on(roctx().on) { if (on) roctx().push(name); }
    ~Range() { if (on) roctx().pop(); }
};

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

inline int64_t tiles_of(int64_t n) { return (n + kTile - 1) / kTile; }
inline unsigned grid_tiles(int64_t n) { return (unsigned)((tiles_of(n) + (kBlock / kTile) - 1) / (kBlock / kTile)); }
inline unsigned grid_flat(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

}  // namespace

#ifdef QS_STAMP
static unsigned long long *g_host_stamps = nullptr;     // qs_debug_set_stamps: handed to every launch through StepArgs
static unsigned long long g_host_stamp_cap = 0;
#endif

struct QsEnv {
    QsConfig cfg;
    int64_t n = 0, tiles = 0;
    float *st = nullptr, *par = nullptr;
    bool per_env_params = false;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    unsigned long long *d_ctr = nullptr;   // device: step counter, one copy per tile
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float nominal_obs[12] = {0};
    float *gae_ws = nullptr;    // workspace of the chunked GAE scan
    size_t gae_ws_floats = 0;
    float *init = nullptr;      // stored per-env initial states (docking-v1, hovering-v0, qs_set_init_state)
    int obs_dim = 12;
    // staging for QS_IO_HOST: a device buffer plus a pinned, device-mapped host mirror of the same size
    void *stage = nullptr;
    size_t stage_bytes = 0;
    char *hpin = nullptr;       // host address of the mirror
    char *hpin_dev = nullptr;   // its device address (kernels of small batches read / write it in place)
    // env groups (qs_set_groups): contiguous tile ranges stepped on their own streams, optionally by their own launcher threads
    std::vector<struct QsGroup *> groups;
    hipEvent_t fork_ev = nullptr;
    bool main_dirty = true;     // the handle enqueued work on its main stream that the group streams have not been ordered behind
    bool groups_dirty = false;  // group streams hold work the main stream has not been ordered behind
    bool runner_env_major = false;   // qs_set_rollout_layout
    struct QsChain *chain = nullptr; // qs_set_queue_mode: private AQL queue for the step launches
};

namespace {

StepArgs make_args(const QsEnv *e)
{
    StepArgs A;
    memset(&A, 0, sizeof A);
    A.st = e->st;
    A.par = e->par;
    A.n = e->n;
    A.tile0 = 0; A.tile_end = e->tiles;
    A.io_env0 = 0; A.io_n = e->n;
    A.T = 1;
    A.step_idx = 0;
    A.ctr = e->d_ctr;
    A.gid0 = e->cfg.env_id_offset;
    A.C.kind = e->cfg.kind == QS_KIND_DOCKING_V2 ? 1 : 0;
    A.C.dt = e->cfg.dt;
    A.C.rmax = e->cfg.kind == QS_KIND_DOCKING_V2 ? 10.0f : 3.0f;
    A.C.vdes_x = e->cfg.kind == QS_KIND_DOCKING_V2 ? 0.2f : 0.0f;
    A.init = e->init;
    A.rc.seed = e->cfg.seed;
    for (int i = 0; i < 4; ++i) A.rc.rr[i] = e->cfg.init_range[i];
    A.rc.rr[4] = e->cfg.mass_scale[0]; A.rc.rr[5] = e->cfg.mass_scale[1];
    A.rc.rr[6] = e->cfg.inertia_scale[0]; A.rc.rr[7] = e->cfg.inertia_scale[1];
    A.rc.par_nom[0] = e->cfg.mass;
    for (int i = 0; i < 3; ++i) A.rc.par_nom[1 + i] = e->cfg.inertia[i];
    A.par_nom = Par{e->cfg.mass, e->cfg.inertia[0], e->cfg.inertia[1], e->cfg.inertia[2]};
    A.auto_reset = e->cfg.auto_reset;
    A.randomise = e->cfg.randomise;

    for (int i = 0; i < 12; ++i) A.nominal_obs[i] = e->nominal_obs[i];
#ifdef QS_STAMP
    A.stamps = g_host_stamps; A.stamp_cap = g_host_stamp_cap; A.stamp_tiles = (unsigned long long)e->tiles;
#endif
    return A;
}

int ensure_stage(QsEnv *e, size_t bytes)
{
    if (e->stage_bytes >= bytes) return QS_OK;
    if (e->stage) { HIP_TRY(hipStreamSynchronize(e->stream)); HIP_TRY(hipFree(e->stage)); e->stage = nullptr; e->stage_bytes = 0; }
    if (e->hpin) { HIP_TRY(hipHostFree(e->hpin)); e->hpin = nullptr; e->hpin_dev = nullptr; }
    HIP_TRY(hipMalloc(&e->stage, bytes));
    e->stage_bytes = bytes;
    if (e->cfg.io_space == QS_IO_HOST) {
        HIP_TRY(hipHostMalloc((void **)&e->hpin, bytes, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void **)&e->hpin_dev, e->hpin, 0));
    }
    return QS_OK;
}

// bump allocator over the staging buffer (256-B aligned slices)
struct Stage {
    char *base;
    size_t off = 0;
    template <typename T> T *take(size_t count)
    {
        T *p = reinterpret_cast<T *>(base + off);
        off += (count * sizeof(T) + 255) & ~size_t(255);
        return p;
    }
};

// Host-buffer calls (QS_IO_HOST: the single-env gym shims, SB2-style numpy VecEnvs) bounce through the pinned mirror.
// Up to kDirectBytes the kernels read and write the mapped host memory in place (a step of a few envs costs one
// launch and one stream sync, no copy engine); larger blocks take ONE DMA in and ONE DMA out of the device buffer.
// Slices are taken in the same order on both sides, so a device pointer maps to its host twin by offset.
constexpr size_t kDirectBytes = 64u << 10;
struct Bounce {
    QsEnv *e;
    bool direct;
    char *dbase;
    Stage S;
    size_t in_end = 0;      // inputs occupy [0, in_end), outputs [in_end, S.off)
    Bounce(QsEnv *env, size_t bytes)
        : e(env), direct(bytes <= kDirectBytes), dbase(direct ? env->hpin_dev : (char *)env->stage), S{dbase} {}
    template <typename T> T *take(size_t count) { return S.take<T>(count); }
    template <typename T> T *host(T *dptr) const { return reinterpret_cast<T *>(e->hpin + ((char *)dptr - dbase)); }
    void inputs_done() { in_end = S.off; }
    int push()              // after the caller filled host(...) of every input slice
    {
        if (!direct && in_end) HIP_TRY(hipMemcpyAsync(dbase, e->hpin, in_end, hipMemcpyHostToDevice, e->stream));
        return QS_OK;
    }
    int pull()              // after the kernels were enqueued: outputs land in host(...) of every output slice
    {
        if (!direct && S.off > in_end)
            HIP_TRY(hipMemcpyAsync(e->hpin + in_end, dbase + in_end, S.off - in_end, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        return QS_OK;
    }
};

// reset-preparation variant of the role-split kernel for a launch of `tiles` tiles (see kPrepMaxTiles)
std::atomic<int> &prep_forced()
{
    static std::atomic<int> f{getenv("QS_RESET_PREP") ? atoi(getenv("QS_RESET_PREP")) : -1};
    return f;
}

int prep_for(int rmode, int64_t tiles)
{
    const int forced = prep_forced().load(std::memory_order_relaxed);
    if (rmode != 1 && rmode != 2) return 0;
    if (forced == 0 || forced == 2) return forced;
    return tiles <= kPrepMaxTiles ? 2 : 0;
}

template <int INTEG, bool PARAMS, int RMODE>
void launch_one(hipStream_t s, const StepArgs &A)
{
    // role-split kernel up to kSplitMaxEnvs envs (few waves per SIMD: the two half-length streams of a tile overlap), the
    // serial kernel above (SIMDs already saturated: the hand-overs only cost).  Both inline the same device functions and
    // the library is built with -ffp-contract=on, so they compute the same bits.  QS_SPLIT=0/1 forces one (A/B runs).
    // The choice follows the handle's env count, not the launch's: the groups of a handle are in flight together.
    static const int forced = getenv("QS_SPLIT") ? atoi(getenv("QS_SPLIT")) : -1;
    const bool split = forced >= 0 ? forced != 0 : A.n <= kSplitMaxEnvs;
    const int64_t tiles = A.tile_end - A.tile0;
    if (split && prep_for(RMODE, tiles) == 2 && (RMODE == 1 || RMODE == 2))
        hipLaunchKernelGGL((k_env_split<INTEG, PARAMS, RMODE, (RMODE == 1 || RMODE == 2) ? 2 : 0>), dim3((unsigned)tiles), dim3(3 * kTile), 0, s, A);
    else if (split) hipLaunchKernelGGL((k_env_split<INTEG, PARAMS, RMODE, 0>), dim3((unsigned)tiles), dim3(2 * kTile), 0, s, A);
    else hipLaunchKernelGGL((k_env<INTEG, PARAMS, RMODE>), dim3((unsigned)((tiles + kBlock / kTile - 1) / (kBlock / kTile))), dim3(kBlock), 0, s, A);
}

template <int INTEG>
void launch_integ(hipStream_t s, const StepArgs &A, bool params, int rmode)
{
    if (rmode == 3) { if (params) launch_one<INTEG, true, 3>(s, A); else launch_one<INTEG, false, 3>(s, A); }
    else if (rmode == 2) launch_one<INTEG, true, 2>(s, A);      // per-episode params imply per-env params
    else if (rmode == 1) { if (params) launch_one<INTEG, true, 1>(s, A); else launch_one<INTEG, false, 1>(s, A); }
    else { if (params) launch_one<INTEG, true, 0>(s, A); else launch_one<INTEG, false, 0>(s, A); }
}

// the env kernels of tiles [A.tile0, A.tile_end) on stream s
int launch_env_on(QsEnv *e, const StepArgs &A, hipStream_t s)
{
    if (e->cfg.kind == QS_KIND_HOVERING_V0) {
        const unsigned grid = (unsigned)((A.tile_end - A.tile0 + kBlock / kTile - 1) / (kBlock / kTile));
        const bool fr = e->cfg.integrator == QS_INTEG_FROZEN, pp = e->per_env_params;
        if (fr && !pp) hipLaunchKernelGGL((k_hover<0, false>), dim3(grid), dim3(kBlock), 0, s, A);
        else if (fr) hipLaunchKernelGGL((k_hover<0, true>), dim3(grid), dim3(kBlock), 0, s, A);
        else if (!pp) hipLaunchKernelGGL((k_hover<1, false>), dim3(grid), dim3(kBlock), 0, s, A);
        else hipLaunchKernelGGL((k_hover<1, true>), dim3(grid), dim3(kBlock), 0, s, A);
        HIP_TRY(hipGetLastError());
        return QS_OK;
    }
    const int rmode = e->init ? 3 : e->cfg.randomise;   // stored initial states take precedence over `randomise`
    if (e->cfg.integrator == QS_INTEG_FROZEN) launch_integ<0>(s, A, e->per_env_params, rmode);
    else launch_integ<1>(s, A, e->per_env_params, rmode);
    HIP_TRY(hipGetLastError());
    return QS_OK;
}

int launch_env(QsEnv *e, StepArgs &A) { return launch_env_on(e, A, e->stream); }

}  // namespace

// ---- env groups -----------------------------------------------------------------------------------------------------
// A handle's tiles can be partitioned into G contiguous groups, each stepped on its OWN stream (EnvPool-style: a trainer
// runs the policy of one group while the others step).  Envs never interact, so a group launch is the ordinary step kernel
// over a tile range: results are bit-identical to the single launch.  What the groups buy is overlap: at 65 536 envs one
// step is ~5 us of kernel plus a ~1.8 us dependent-kernel boundary (MI355X_MICROARCH.md, price list, "boundary"); with two
// chains in flight one group's boundary and wave ramp hide under the other group's compute.  Two launches per step would
// make ONE host thread the bottleneck (~2.5-3 us per launch), so each group may get its own launcher thread: the API
// thread posts a launch record into a single-producer ring and returns; the group's thread issues it on the group's stream.
struct QsGroup {
    enum { kRing = 64 };
    enum ReqType { REQ_LAUNCH = 0, REQ_WAIT_EVENT = 1 };
    struct Req {
        int type;
        StepArgs A;
        hipEvent_t ev;
    };
    QsEnv *env = nullptr;
    int index = 0;
    int64_t tile0 = 0, tile_end = 0, env0 = 0, env_end = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t done_ev = nullptr;
    // launcher thread (optional)
    bool threaded = false;
    std::thread th;
    Req ring[kRing];
    std::atomic<uint64_t> head{0}, tail{0};     // posted / issued
    std::atomic<int> stop{0}, sleeping{0}, err{0};
    std::mutex m;
    std::condition_variable cv;
};

namespace {

int group_execute(QsGroup *g, const QsGroup::Req &r)
{
    if (r.type == QsGroup::REQ_LAUNCH) return launch_env_on(g->env, r.A, g->stream);
    HIP_TRY(hipStreamWaitEvent(g->stream, r.ev, 0));
    return QS_OK;
}

void group_worker(QsGroup *g)
{
    (void)hipSetDevice(g->env->cfg.device);
    for (;;) {
        const uint64_t t = g->tail.load(std::memory_order_relaxed);
        int spins = 0;
        while (g->head.load(std::memory_order_acquire) == t) {
            if (g->stop.load(std::memory_order_acquire)) return;
            if (++spins < 40000) { __builtin_ia32_pause(); continue; }
            // idle for ~100 us: sleep until the API thread posts again (it checks `sleeping` after publishing)
            std::unique_lock<std::mutex> lk(g->m);
            g->sleeping.store(1, std::memory_order_seq_cst);
            if (g->head.load(std::memory_order_seq_cst) == t && !g->stop.load())
                g->cv.wait_for(lk, std::chrono::milliseconds(50));
            g->sleeping.store(0, std::memory_order_seq_cst);
            spins = 0;
        }
        const int rc = group_execute(g, g->ring[t % QsGroup::kRing]);
        if (rc != QS_OK) { int z = 0; g->err.compare_exchange_strong(z, rc); }
        g->tail.store(t + 1, std::memory_order_release);
    }
}

int group_post(QsGroup *g, const QsGroup::Req &r)
{
    if (!g->threaded) return group_execute(g, r);
    const uint64_t h = g->head.load(std::memory_order_relaxed);
    while (h - g->tail.load(std::memory_order_acquire) >= QsGroup::kRing) __builtin_ia32_pause();
    g->ring[h % QsGroup::kRing] = r;
    g->head.store(h + 1, std::memory_order_seq_cst);
    if (g->sleeping.load(std::memory_order_seq_cst)) { std::lock_guard<std::mutex> lk(g->m); g->cv.notify_one(); }
    return QS_OK;
}

// every record posted to group g has been issued to its stream
void group_wait_issued(QsGroup *g)
{
    if (!g->threaded) return;
    const uint64_t h = g->head.load(std::memory_order_relaxed);
    while (g->tail.load(std::memory_order_acquire) != h) __builtin_ia32_pause();
}

// every posted record has been issued to its stream
int groups_drain(QsEnv *e)
{
    int rc = QS_OK;
    for (QsGroup *g : e->groups) {
        if (g->threaded) {
            const uint64_t h = g->head.load(std::memory_order_relaxed);
            while (g->tail.load(std::memory_order_acquire) != h) __builtin_ia32_pause();
        }
        const int ge = g->err.exchange(0);
        if (ge != QS_OK && rc == QS_OK) rc = fail(ge, "a group launcher thread reported error %d (group %d)", ge, g->index);
    }
    return rc;
}

// group streams wait for everything enqueued so far on the main stream
int groups_fork(QsEnv *e)
{
    if (e->groups.empty()) return QS_OK;
    HIP_TRY(hipEventRecord(e->fork_ev, e->stream));
    for (QsGroup *g : e->groups) {
        QsGroup::Req r;
        r.type = QsGroup::REQ_WAIT_EVENT;
        r.ev = e->fork_ev;
        int rc = group_post(g, r);
        if (rc) return rc;
    }
    // a re-record of fork_ev must not overtake a wait that has not been issued yet
    int rc = groups_drain(e);
    e->main_dirty = false;
    return rc;
}

// the main stream waits for everything enqueued so far on the group streams
int groups_join(QsEnv *e)
{
    if (e->groups.empty()) return QS_OK;
    int rc = groups_drain(e);
    if (rc) return rc;
    for (QsGroup *g : e->groups) {
        HIP_TRY(hipEventRecord(g->done_ev, g->stream));
        HIP_TRY(hipStreamWaitEvent(e->stream, g->done_ev, 0));
    }
    e->groups_dirty = false;
    return QS_OK;
}

void groups_destroy(QsEnv *e)
{
    for (QsGroup *g : e->groups) {
        if (g->threaded) {
            g->stop.store(1, std::memory_order_release);
            { std::lock_guard<std::mutex> lk(g->m); g->cv.notify_one(); }
            if (g->th.joinable()) g->th.join();
        }
        if (g->stream) (void)hipStreamSynchronize(g->stream);
        if (g->done_ev) (void)hipEventDestroy(g->done_ev);
        if (g->own_stream && g->stream) (void)hipStreamDestroy(g->stream);
        delete g;
    }
    e->groups.clear();
    if (e->fork_ev) { (void)hipEventDestroy(e->fork_ev); e->fork_ev = nullptr; }
    e->groups_dirty = false;
    e->main_dirty = true;
}

}  // namespace

// ---- private AQL queue for the step launches (qs_set_queue_mode) -------------------------------------------------------
// Every kernel HIP launches ends with an agent-scope release: the eight XCD L2s are not coherent with each other, so their
// dirty lines are written back before the next packet may start.  For a chain of dependent step launches that write-back is
// 1.6 of 6.5 us per step at 65 536 envs (profiles/r02/ab_experiments.txt, section E) -- and it is not needed: tile b is
// stepped by workgroup b of every launch, workgroup b always lands on the same XCD, so the tile's state can stay dirty in
// that XCD's L2 from one step to the next.  HIP has no launch without the fence; an AQL packet written by hand has:
//     header = KERNEL_DISPATCH | BARRIER (ordered behind the previous packet) | ACQUIRE agent (fresh kernargs / actions;
//              0.16 us, does not touch dirty lines) | RELEASE none.
// The handle therefore owns an HSA queue, loads its own copy of the library's code object into it, and writes one packet per
// qs_step.  Everything else stays on HIP: any other entry point first DRAINS the queue with a release packet (host wait).
//
// Ordering against the CALLER's work (the handle's stream S), round 3 -- QS_ORDER_STREAM, the default where the device has
// stream memory operations: a step behaves like a launch on S although it runs elsewhere.  Per submission (one qs_step, or the
// T steps of qs_rollout_stepwise):
//     S:      hipStreamWriteValue64(fwd, n)              -- executes when everything enqueued on S so far has finished
//     queue:  barrier-value packet (fwd >= n) | step packet(s), the last one with a completion signal `rev`
//     S:      hipStreamWaitValue64(rev == V0 - n)        -- whatever is enqueued on S afterwards runs after the step(s)
// and that last packet carries an agent-scope RELEASE, so that the outputs of all T steps are in memory -- not dirty in one XCD's
// L2 -- when `rev` fires (the state lines are written back with them and stay valid in their L2).  No host synchronisation
// anywhere.  A per-step loop (T = 1) thereby pays what a HIP launch pays -- per-step consumable outputs ARE the write-back:
// write-through output stores instead of the release were measured no faster (the kernel then ends when memory, not the L2,
// acknowledges its stores) -- plus the hand-shake; a T-step roll-out pays both once.  QS_ORDER_HOST is round 2's contract:
// inputs complete at the call, outputs valid after qs_sync (no hand-shake packets).
struct QsChainLane {                  // one private queue and the contiguous tile range it steps
    hsa_queue_t *queue = nullptr;
    char *kernargs = nullptr;
    std::vector<uint64_t> slot_qidx;  // queue index of the packet that last used each kernarg slot
    uint64_t issued = 0;              // step packets written so far
    int64_t tile0 = 0, tile_end = 0;
    hsa_signal_t done{};
    // QS_ORDER_STREAM: completion signal of this lane's submissions.  Allocated by HIP as "signal memory" (the only memory
    // hipStreamWaitValue64 accepts); HIP hands out the address of the signal's VALUE, the handle is the amd_signal_t around it.
    void *rev_ptr = nullptr;
    hsa_signal_t rev{};
    int64_t rev_value = 0;            // value of `rev` once every submission so far has completed (counts DOWN: AQL decrements)
};

struct QsChain {
    hsa_agent_t gpu{}, cpu{};
    hsa_amd_memory_pool_t kernarg_pool{};
    std::vector<QsChainLane> lanes;
    int requested = 0;                // the queue count qs_set_queue_mode was called with (lanes.size() may be smaller)
    hsa_executable_t exe{};
    hsa_code_object_reader_t reader{};
    bool have_exe = false, have_reader = false;
    std::vector<char> image;          // the gfx950 code object (kept alive for the executable)
    uint64_t kernel_object = 0;
    int v_integ = -1, v_params = -1, v_rmode = -1, v_split = -1, v_prep = -1;   // the instantiation kernel_object belongs to
    uint32_t kernarg_size = 0, group_size = 0, private_size = 0;
    unsigned block = 0;
    size_t stride = 0, slots = 0;
    unsigned *d_owner = nullptr;      // [tiles] 32-bit words, agent-scope atomics only
    unsigned *d_err = nullptr;        // the placement guard's error word: pinned, coherent HOST memory (d_err = its device address) -- a
    volatile unsigned *h_err = nullptr;   // misplaced workgroup sets it with a system-scope atomic, and every later submission sees it
                                          // without a synchronisation: a stream-ordered loop that never drains still fails within a few steps
    bool kernargs_on_device = false;  // kernarg ring in BAR-mapped device memory (else: host memory, correct but slow)
    bool dirty = false;               // packets enqueued since the last drain
    bool hip_dirty = true;            // the handle did HIP-side work since the last packet
    // QS_ORDER_STREAM
    bool stream_ordered = false;
    void *fwd_ptr = nullptr;          // value word of the forward signal (HIP signal memory)
    hsa_signal_t fwd{};
    uint64_t fwd_seq = 0;             // submissions so far
    int dbg_shift = 0;                // one-shot: the next step packet runs with StepArgs::dbg_shift (placement-guard test)
};

namespace {

#define HSA_TRY(expr)                                                                                   \
    do {                                                                                                \
        hsa_status_t s_ = (expr);                                                                       \
        if (s_ != HSA_STATUS_SUCCESS) {                                                                 \
            const char *m_ = nullptr;                                                                   \
            hsa_status_string(s_, &m_);                                                                 \
            return fail(QS_ERR_HIP, "%s failed: %s", #expr, m_ ? m_ : "unknown HSA status");           \
        }                                                                                               \
    } while (0)

constexpr int64_t kRevStart = (int64_t)1 << 40;

struct AgentPick {
    uint32_t want_bdf, want_domain;
    int want_index, seen;
    hsa_agent_t gpu, cpu;
    bool have_gpu, have_cpu;
};

hsa_status_t chain_agent_cb(hsa_agent_t a, void *data)
{
    AgentPick *p = (AgentPick *)data;
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (t == HSA_DEVICE_TYPE_CPU && !p->have_cpu) { p->cpu = a; p->have_cpu = true; }
    if (t == HSA_DEVICE_TYPE_GPU) {
        uint32_t bdf = 0, dom = 0;
        const bool ok = hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) == HSA_STATUS_SUCCESS &&
                        hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom) == HSA_STATUS_SUCCESS;
        // the agent must be THE device of the handle: matched by PCI domain + bus/device/function (BDFID = bus << 8 | device << 3 |
        // function); by ordinal only when HIP cannot name the device's PCI address
        if (p->want_bdf != 0xffffffffu ? (ok && bdf == p->want_bdf && dom == p->want_domain) : p->seen == p->want_index) {
            p->gpu = a;
            p->have_gpu = true;
        }
        ++p->seen;
    }
    return HSA_STATUS_SUCCESS;
}

// device-local memory the host may write through the PCIe BAR: where HIP itself keeps kernel arguments on this platform
// (kernargs in host memory would cost every workgroup a PCIe read: 44 us per step instead of 5)
struct DevPoolPick {
    hsa_agent_t cpu;
    hsa_amd_memory_pool_t pool;
    bool found;
};
hsa_status_t chain_device_pool_cb(hsa_amd_memory_pool_t pool, void *data)
{
    DevPoolPick *p = (DevPoolPick *)data;
    hsa_amd_segment_t seg;
    if (hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS || seg != HSA_AMD_SEGMENT_GLOBAL)
        return HSA_STATUS_SUCCESS;
    uint32_t flags = 0;
    bool alloc = false;
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if (!alloc || !(flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED)) return HSA_STATUS_SUCCESS;
    hsa_amd_memory_pool_access_t acc = HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED;
    hsa_amd_agent_memory_pool_get_info(p->cpu, pool, HSA_AMD_AGENT_MEMORY_POOL_INFO_ACCESS, &acc);
    if (acc == HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED) return HSA_STATUS_SUCCESS;
    p->pool = pool;
    p->found = true;
    return HSA_STATUS_INFO_BREAK;
}

hsa_status_t chain_kernarg_pool_cb(hsa_amd_memory_pool_t pool, void *data)
{
    hsa_amd_segment_t seg;
    if (hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS || seg != HSA_AMD_SEGMENT_GLOBAL)
        return HSA_STATUS_SUCCESS;
    uint32_t flags = 0;
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    if (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) { *(hsa_amd_memory_pool_t *)data = pool; return HSA_STATUS_INFO_BREAK; }
    return HSA_STATUS_SUCCESS;
}

// the gfx950 code object of THIS library: the .hip_fatbin section of the shared object the code runs from holds a clang
// offload bundle; its amdgcn entry is the ELF that HIP itself loads
int chain_read_code_object(std::vector<char> &out)
{
    Dl_info di;
    if (!dladdr((void *)&chain_read_code_object, &di) || !di.dli_fname) return fail(QS_ERR_HIP, "queue mode: cannot locate the library file");
    const int fd = open(di.dli_fname, O_RDONLY);
    if (fd < 0) return fail(QS_ERR_HIP, "queue mode: cannot open %s", di.dli_fname);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return fail(QS_ERR_HIP, "queue mode: fstat failed"); }
    const char *base = (const char *)mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (base == MAP_FAILED) return fail(QS_ERR_HIP, "queue mode: mmap failed");
    int rc = fail(QS_ERR_HIP, "queue mode: no gfx950 code object in %s", di.dli_fname);
    const Elf64_Ehdr *eh = (const Elf64_Ehdr *)base;
    const Elf64_Shdr *sh = (const Elf64_Shdr *)(base + eh->e_shoff);
    const char *names = base + sh[eh->e_shstrndx].sh_offset;
    for (int i = 0; i < eh->e_shnum; ++i) {
        if (strcmp(names + sh[i].sh_name, ".hip_fatbin") != 0) continue;
        const char *fb = base + sh[i].sh_offset;
        const char magic[] = "__CLANG_OFFLOAD_BUNDLE__";
        if (sh[i].sh_size < 32 || memcmp(fb, magic, 24) != 0) break;
        uint64_t n;
        memcpy(&n, fb + 24, 8);
        const char *q = fb + 32;
        for (uint64_t k = 0; k < n; ++k) {
            uint64_t off, size, tl;
            memcpy(&off, q, 8); memcpy(&size, q + 8, 8); memcpy(&tl, q + 16, 8);
            const char *triple = q + 24;
            q += 24 + tl;
            if (tl >= 6 && memmem(triple, tl, "amdgcn", 6) && memmem(triple, tl, "gfx950", 6) && off + size <= sh[i].sh_size) {
                out.assign(fb + off, fb + off + size);
                rc = QS_OK;
            }
        }
        break;
    }
    munmap((void *)base, (size_t)st.st_size);
    return rc;
}

void chain_close(QsEnv *e)
{
    QsChain *c = e->chain;
    if (!c) return;
    // nothing may be left waiting on a hand-shake value that will never come
    if (c->fwd.handle) hsa_signal_store_screlease(c->fwd, INT64_MAX);
    for (QsChainLane &L : c->lanes) {
        if (L.queue) hsa_queue_destroy(L.queue);
        if (L.kernargs) hsa_amd_memory_pool_free(L.kernargs);
        if (L.done.handle) hsa_signal_destroy(L.done);
        if (L.rev_ptr) (void)hipFree(L.rev_ptr);
    }
    if (c->fwd_ptr) (void)hipFree(c->fwd_ptr);
    if (c->have_exe) hsa_executable_destroy(c->exe);
    if (c->have_reader) hsa_code_object_reader_destroy(c->reader);
    if (c->d_owner) (void)hipFree(c->d_owner);
    if (c->h_err) (void)hipHostFree((void *)c->h_err);
    delete c;
    e->chain = nullptr;
}

// the step-kernel instantiation launch_env_on would pick for the handle AS IT IS NOW (qs_set_params / qs_set_init_state after
// qs_set_queue_mode change it): (re-)resolved against the loaded executable whenever it differs from the one in use
int chain_resolve_kernel(QsEnv *e)
{
    QsChain *c = e->chain;
    const int integ = e->cfg.integrator == QS_INTEG_FROZEN ? 0 : 1;
    const int rmode = e->init ? 3 : e->cfg.randomise;
    const int params = (rmode == 2 || e->per_env_params) ? 1 : 0;
    static const int forced = getenv("QS_SPLIT") ? atoi(getenv("QS_SPLIT")) : -1;
    const int split = (forced >= 0 ? forced != 0 : e->n <= kSplitMaxEnvs) ? 1 : 0;
    int64_t lane_tiles = 0;
    for (const QsChainLane &L : c->lanes) lane_tiles = std::max<int64_t>(lane_tiles, L.tile_end - L.tile0);
    if (c->lanes.empty()) lane_tiles = e->tiles;          // chain_open resolves once before the lanes exist: re-resolved at the first step
    const int prep = split ? prep_for(rmode, lane_tiles) : 0;
    if (c->kernel_object && integ == c->v_integ && rmode == c->v_rmode && params == c->v_params && split == c->v_split && prep == c->v_prep) return QS_OK;
    char sym[160];
    if (split) snprintf(sym, sizeof sym, "_ZN12_GLOBAL__N_111k_env_splitILi%dELb%dELi%dELi%dEEEvNS_8StepArgsE.kd", integ, params, rmode, prep);
    else snprintf(sym, sizeof sym, "_ZN12_GLOBAL__N_15k_envILi%dELb%dELi%dEEEvNS_8StepArgsE.kd", integ, params, rmode);
    hsa_executable_symbol_t ks;
    uint64_t ko = 0;
    uint32_t ka = 0, gs = 0, ps = 0;
    HSA_TRY(hsa_executable_get_symbol_by_name(c->exe, sym, &c->gpu, &ks));
    HSA_TRY(hsa_executable_symbol_get_info(ks, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &ko));
    HSA_TRY(hsa_executable_symbol_get_info(ks, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &ka));
    HSA_TRY(hsa_executable_symbol_get_info(ks, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &gs));
    HSA_TRY(hsa_executable_symbol_get_info(ks, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &ps));
    if (ka < sizeof(StepArgs)) return fail(QS_ERR_HIP, "qs_set_queue_mode: kernel argument block is %u B, StepArgs %zu B", ka, sizeof(StepArgs));
    if (c->stride && (((size_t)ka + 255) & ~size_t(255)) > c->stride)
        return fail(QS_ERR_HIP, "queue mode: kernel argument block of %s (%u B) exceeds the ring's slot size", sym, ka);
    c->kernel_object = ko; c->kernarg_size = ka; c->group_size = gs; c->private_size = ps;
    c->block = split ? split_waves(rmode, prep) * kTile : kBlock;
    c->v_integ = integ; c->v_rmode = rmode; c->v_params = params; c->v_split = split; c->v_prep = prep;
    return QS_OK;
}

// HIP "signal memory": an HSA signal created by HIP, of which it hands out the address of the value word.  The handle our
// own AQL packets need is the amd_signal_t around that word (amd_hsa_signal.h: value at offset 8, 64-byte aligned).
int chain_alloc_hip_signal(void **value_ptr, hsa_signal_t *handle, int64_t initial)
{
    HIP_TRY(hipExtMallocWithFlags(value_ptr, 8, hipMallocSignalMemory));
    const uintptr_t h = (uintptr_t)*value_ptr - offsetof(amd_signal_t, value);
    if (h & (AMD_SIGNAL_ALIGN_BYTES - 1)) return fail(QS_ERR_HIP, "queue mode: HIP signal memory is not the value word of an amd_signal_t");
    handle->handle = (uint64_t)h;
    if (((amd_signal_t *)h)->kind != AMD_SIGNAL_KIND_USER) return fail(QS_ERR_HIP, "queue mode: HIP signal memory is not a user signal");
    hsa_signal_store_screlease(*handle, initial);
    if (hsa_signal_load_scacquire(*handle) != initial) return fail(QS_ERR_HIP, "queue mode: HIP signal memory does not behave like an HSA signal");
    return QS_OK;
}

bool chain_can_stream_order(QsEnv *e)
{
    int can = 0;
    return hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, e->cfg.device) == hipSuccess && can != 0;
}

int chain_enable_stream_order(QsEnv *e)
{
    QsChain *c = e->chain;
    if (!c->fwd_ptr) {
        int r = chain_alloc_hip_signal(&c->fwd_ptr, &c->fwd, 0);
        if (r) return r;
        c->fwd_seq = 0;
    }
    for (QsChainLane &L : c->lanes) {
        if (L.rev_ptr) continue;
        int r = chain_alloc_hip_signal(&L.rev_ptr, &L.rev, kRevStart);
        if (r) return r;
        L.rev_value = kRevStart;
    }
    c->stream_ordered = true;
    return QS_OK;
}

int chain_open(QsEnv *e, int nq)
{
    if (e->cfg.kind == QS_KIND_HOVERING_V0) return fail(QS_ERR_INVALID, "qs_set_queue_mode: docking envs only");
    if (e->cfg.io_space != QS_IO_DEVICE) return fail(QS_ERR_INVALID, "qs_set_queue_mode: device buffers only");
    QsChain *c = new (std::nothrow) QsChain();
    if (!c) return fail(QS_ERR_NOMEM, "qs_set_queue_mode: out of host memory");
    e->chain = c;
    auto body = [&]() -> int {
        HSA_TRY(hsa_init());
        char bus[32] = "";
        AgentPick pick{0xffffffffu, 0, e->cfg.device, 0, {}, {}, false, false};
        unsigned dom = 0, b = 0, d = 0, f = 0;
        if (hipDeviceGetPCIBusId(bus, sizeof bus, e->cfg.device) == hipSuccess && sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &f) == 4) {
            pick.want_bdf = (b << 8) | (d << 3) | f;
            pick.want_domain = dom;
        }
        HSA_TRY(hsa_iterate_agents(chain_agent_cb, &pick));
        if (!pick.have_gpu || !pick.have_cpu) return fail(QS_ERR_HIP, "qs_set_queue_mode: no HSA agent for device %d (%s)", e->cfg.device, bus);
        c->gpu = pick.gpu; c->cpu = pick.cpu;
        hsa_status_t ps = hsa_amd_agent_iterate_memory_pools(c->cpu, chain_kernarg_pool_cb, &c->kernarg_pool);
        if (ps != HSA_STATUS_INFO_BREAK) return fail(QS_ERR_HIP, "qs_set_queue_mode: no kernarg memory pool");
        int r = chain_read_code_object(c->image);
        if (r) return r;
        HSA_TRY(hsa_code_object_reader_create_from_memory(c->image.data(), c->image.size(), &c->reader));
        c->have_reader = true;
        HSA_TRY(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &c->exe));
        c->have_exe = true;
        HSA_TRY(hsa_executable_load_agent_code_object(c->exe, c->gpu, c->reader, nullptr, nullptr));
        HSA_TRY(hsa_executable_freeze(c->exe, nullptr));
        r = chain_resolve_kernel(e);
        if (r) return r;
        c->stride = (((size_t)c->kernarg_size + 255) & ~size_t(255)) + 256;   // room for any instantiation's hidden arguments
        // a small ring: a slot is rewritten only after its packet ran, and recently used kernarg lines are still in the caches
        // (4 096 slots: 5.43 us per step, 256: 5.27, 64 and 16: 5.24, 4: host-bound; profiles/r02/ab_experiments.txt, section E)
        c->slots = getenv("QS_CHAIN_SLOTS") ? (size_t)atoi(getenv("QS_CHAIN_SLOTS")) : 64;
        if (c->slots < 2 || c->slots > 4096) c->slots = 64;
        DevPoolPick dp{c->cpu, {}, false};
        (void)hsa_amd_agent_iterate_memory_pools(c->gpu, chain_device_pool_cb, &dp);
        c->kernargs_on_device = dp.found;
        // nq queues, each stepping a contiguous range of tiles (whole multiples of 8 tiles where possible: one per XCD)
        std::vector<int64_t> cut{0};
        for (int q = 1; q <= nq; ++q) {
            int64_t t1 = q == nq ? e->tiles : ((e->tiles * q / nq) + 7) / 8 * 8;
            if (t1 > e->tiles) t1 = e->tiles;
            if (t1 > cut.back()) cut.push_back(t1);        // small handles: fewer, non-empty lanes
        }
        c->lanes.resize(cut.size() - 1);
        for (size_t q = 0; q + 1 < cut.size(); ++q) {
            QsChainLane &L = c->lanes[q];
            L.tile0 = cut[q]; L.tile_end = cut[q + 1];
            HSA_TRY(hsa_queue_create(c->gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &L.queue));
            if (dp.found) {
                HSA_TRY(hsa_amd_memory_pool_allocate(dp.pool, c->stride * c->slots, 0, (void **)&L.kernargs));
                HSA_TRY(hsa_amd_agents_allow_access(1, &c->cpu, nullptr, L.kernargs));
            } else {
                HSA_TRY(hsa_amd_memory_pool_allocate(c->kernarg_pool, c->stride * c->slots, 0, (void **)&L.kernargs));
                HSA_TRY(hsa_amd_agents_allow_access(1, &c->gpu, nullptr, L.kernargs));
            }
            memset(L.kernargs, 0, c->stride * c->slots);   // the hidden arguments behind StepArgs are never read: zeros
            L.slot_qidx.assign(c->slots, 0);
            HSA_TRY(hsa_signal_create(0, 0, nullptr, &L.done));
        }
        HIP_TRY(hipMalloc((void **)&c->d_owner, (size_t)e->tiles * sizeof(unsigned)));
        HIP_TRY(hipHostMalloc((void **)&c->h_err, 64, hipHostMallocMapped | hipHostMallocCoherent));
        *c->h_err = 0;
        HIP_TRY(hipHostGetDevicePointer((void **)&c->d_err, (void *)c->h_err, 0));
        // stream-ordered hand-shake: needs HIP's stream memory operations and its signal memory; without them the mode
        // stays host-ordered (round 2's contract)
        const char *ord = getenv("QS_CHAIN_ORDER");            // "host": start with round 2's contract (A/B runs)
        if (!(ord && ord[0] == 'h') && chain_can_stream_order(e)) {
            r = chain_enable_stream_order(e);
            if (r) return r;
        }
        return QS_OK;
    };
    const int rc = body();
    if (rc != QS_OK) chain_close(e);
    return rc;
}

// one AQL packet behind everything enqueued before it on lane L (barrier bit): kind 0 the step kernel, 1 a barrier-AND packet
// (drain), 2 an AMD barrier-value packet that holds the lane until `wait_sig` >= wait_value (the caller's stream is ready)
enum { PKT_STEP = 0, PKT_BARRIER = 1, PKT_WAIT_VALUE = 2 };
uint64_t chain_write_packet(QsChain *c, QsChainLane &L, int kind, const void *kernarg, unsigned grid, int acquire, int release,
                            hsa_signal_t completion, hsa_signal_t wait_sig = hsa_signal_t{0}, int64_t wait_value = 0)
{
    const uint64_t idx = hsa_queue_add_write_index_relaxed(L.queue, 1);
    while (idx - hsa_queue_load_read_index_scacquire(L.queue) >= L.queue->size) __builtin_ia32_pause();
    void *slot = (char *)L.queue->base_address + (idx & (L.queue->size - 1)) * 64;
    uint32_t word0;                                 // header (16 bits) + the 16 bits behind it, published by ONE 32-bit store
    const uint16_t fences = (1 << HSA_PACKET_HEADER_BARRIER) | (acquire << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                            (release << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    if (kind == PKT_BARRIER) {
        hsa_barrier_and_packet_t *p = (hsa_barrier_and_packet_t *)slot;
        memset((char *)p + 4, 0, 60);
        p->completion_signal = completion;
        word0 = (uint16_t)((HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | fences);
    } else if (kind == PKT_WAIT_VALUE) {
        hsa_amd_barrier_value_packet_t *p = (hsa_amd_barrier_value_packet_t *)slot;
        memset((char *)p + 4, 0, 60);
        p->signal = wait_sig;
        p->value = wait_value;
        p->mask = -1;
        p->cond = HSA_SIGNAL_CONDITION_GTE;
        p->completion_signal = completion;
        word0 = (uint16_t)((HSA_PACKET_TYPE_VENDOR_SPECIFIC << HSA_PACKET_HEADER_TYPE) | fences) |
                ((uint32_t)HSA_AMD_PACKET_TYPE_BARRIER_VALUE << 16);
    } else {
        hsa_kernel_dispatch_packet_t *p = (hsa_kernel_dispatch_packet_t *)slot;
        p->workgroup_size_x = (uint16_t)c->block; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
        p->reserved0 = 0;
        p->grid_size_x = grid; p->grid_size_y = 1; p->grid_size_z = 1;
        p->private_segment_size = c->private_size;
        p->group_segment_size = c->group_size;
        p->kernel_object = c->kernel_object;
        p->kernarg_address = (void *)kernarg;
        p->reserved2 = 0;
        p->completion_signal = completion;
        word0 = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | fences) |
                ((uint32_t)(1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS) << 16);
    }
    __atomic_store_n((uint32_t *)slot, word0, __ATOMIC_RELEASE);
    hsa_signal_store_screlease(L.queue->doorbell_signal, (hsa_signal_value_t)idx);
    return idx;
}

// every packet has run and what it wrote is visible to the whole system (host wait); reports a misplaced tile
int chain_drain(QsEnv *e)
{
    QsChain *c = e->chain;
    if (!c || !c->dirty) return QS_OK;
    for (QsChainLane &L : c->lanes) {
        hsa_signal_store_relaxed(L.done, 1);
        chain_write_packet(c, L, PKT_BARRIER, nullptr, 0, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_SYSTEM, L.done);
    }
    for (QsChainLane &L : c->lanes)
        while (hsa_signal_wait_scacquire(L.done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) != 0) {}
    c->dirty = false;
    if (*c->h_err) {
        *c->h_err = 0;
        return fail(QS_ERR_HIP, "queue mode: a workgroup ran on another XCD than the one holding its tile; the steps since the last "
                                "synchronisation are invalid (this placement is not promised by HIP: use qs_set_queue_mode(env, 0))");
    }
    return QS_OK;
}

// T consecutive steps (T kernarg blocks: steps[t] differ in their I/O pointers only) on every lane, behind ONE hand-shake with
// the handle's stream when the chain is stream-ordered
int chain_submit(QsEnv *e, const StepArgs *steps, int64_t T)
{
    QsChain *c = e->chain;
    if (*c->h_err) {
        // a workgroup of an EARLIER step found its tile on another XCD (the word is host memory: no synchronisation needed to see
        // it).  Reported here as well as at the next drain, so that a loop of nothing but steps cannot run on unnoticed; the flag
        // stays set until a draining call has reported it and re-armed the handle.
        return fail(QS_ERR_HIP, "queue mode: a workgroup ran on another XCD than the one holding its tile; the steps since the last "
                                "synchronisation are invalid (call qs_sync, then continue or use qs_set_queue_mode(env, 0))");
    }
    int r = chain_resolve_kernel(e);               // qs_set_params / qs_set_init_state since the last step?
    if (r) return r;
    if (c->stream_ordered) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(e->stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
            return fail(QS_ERR_INVALID, "queue mode: a private-queue step cannot be captured into a hipGraph (use qs_set_queue_mode(env, 0))");
    }
    if (c->hip_dirty) {
        // HIP-side work of the handle (reset, set_state, ...) must have finished, and no tile has an owning XCD yet
        HIP_TRY(hipMemsetAsync(c->d_owner, 0xff, (size_t)e->tiles * sizeof(unsigned), e->stream));
        if (!c->stream_ordered) HIP_TRY(hipStreamSynchronize(e->stream));    // stream-ordered: the write-value below is behind it
        c->hip_dirty = false;
    }
    const size_t nl = c->lanes.size();
    if (c->stream_ordered) {
        ++c->fwd_seq;
        HIP_TRY(hipStreamWriteValue64(e->stream, c->fwd_ptr, c->fwd_seq, 0));
        for (QsChainLane &L : c->lanes)
            chain_write_packet(c, L, PKT_WAIT_VALUE, nullptr, 0, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, hsa_signal_t{0}, c->fwd,
                               (int64_t)c->fwd_seq);
    }
    for (int64_t t = 0; t < T; ++t) {
        StepArgs A = steps[t];
        A.owner = c->d_owner;
        A.err = c->d_err;
        A.dbg_shift = c->dbg_shift;
        c->dbg_shift = 0;
        // all lanes' kernarg blocks first, ONE read-back behind them, then the packets: the read-back is a PCIe round trip
        char *ka[8];
        size_t slot[8];
        for (size_t q = 0; q < nl; ++q) {
            QsChainLane &L = c->lanes[q];
            A.tile0 = L.tile0; A.tile_end = L.tile_end;
            // the slot about to be rewritten belongs to step `issued - slots`, queue packet p: that kernel has FINISHED once the
            // packet behind it has been taken off the queue (every packet carries the barrier bit): read index past p + 1
            slot[q] = L.issued % c->slots;
            if (L.issued >= c->slots)
                while (hsa_queue_load_read_index_scacquire(L.queue) < L.slot_qidx[slot[q]] + 2) __builtin_ia32_pause();
            ka[q] = L.kernargs + slot[q] * c->stride;
            memcpy(ka[q], &A, sizeof A);
        }
        if (c->kernargs_on_device) {
            // posted writes through the BAR: reading the last word back makes sure they have landed before a doorbell rings
            __builtin_ia32_sfence();
            (void)*(volatile uint32_t *)(ka[nl - 1] + sizeof A - sizeof(uint32_t));
        }
        const bool last = t + 1 == T;
        for (size_t q = 0; q < nl; ++q) {
            QsChainLane &L = c->lanes[q];
            const int64_t tiles = L.tile_end - L.tile0;
            const unsigned grid = c->v_split ? (unsigned)(tiles * c->block)
                                                        : (unsigned)(((tiles + kBlock / kTile - 1) / (kBlock / kTile)) * kBlock);
            // stream-ordered: the LAST packet of the submission publishes -- agent-scope release (the outputs of all T steps leave
            // the L2s; the state lines are written back too but stay valid where they are) and the completion signal the
            // caller's stream waits for.  (Write-through `sc1` output stores on every step instead of this release were built
            // and measured: 6.90 against 5.74 us per step in a 600-step roll-out, and they cost the ordinary launches 0.9 us
            // through the store code they displaced; profiles/r03/ab_experiments.txt section I.)
            const bool sig = c->stream_ordered && last;
            L.slot_qidx[slot[q]] = chain_write_packet(c, L, PKT_STEP, ka[q], grid, HSA_FENCE_SCOPE_AGENT,
                                                      sig ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_NONE, sig ? L.rev : hsa_signal_t{0});
            ++L.issued;
            if (sig) --L.rev_value;
        }
    }
    c->dirty = true;
    if (c->stream_ordered)
        for (QsChainLane &L : c->lanes)
            HIP_TRY(hipStreamWaitValue64(e->stream, L.rev_ptr, (uint64_t)L.rev_value, hipStreamWaitValueEq, ~0ull));
    return QS_OK;
}

int chain_step(QsEnv *e, const StepArgs &A) { return chain_submit(e, &A, 1); }

// entry points that use the main stream: order it behind pending group work / the private queue first
int main_stream_entry(QsEnv *e)
{
    if (e->groups_dirty) { int rc = groups_join(e); if (rc) return rc; }
    e->main_dirty = true;
    if (e->chain) {
        int rc = chain_drain(e);
        e->chain->hip_dirty = true;
        if (rc) return rc;
    }
    return QS_OK;
}

int fill_params(QsEnv *e)
{
    Par P{e->cfg.mass, e->cfg.inertia[0], e->cfg.inertia[1], e->cfg.inertia[2]};
    hipLaunchKernelGGL(k_fill_par, dim3(grid_tiles(e->n)), dim3(kBlock), 0, e->stream, e->par, e->n, P);
    HIP_TRY(hipGetLastError());
    return QS_OK;
}

int do_reset(QsEnv *e, const uint8_t *d_mask, float *d_obs, int init_all)
{
    StepArgs A = make_args(e);
    A.obs = d_obs;
    if (e->cfg.kind == QS_KIND_HOVERING_V0)
        hipLaunchKernelGGL(k_hover_reset, dim3(grid_tiles(e->n)), dim3(kBlock), 0, e->stream, A, d_mask);
    else
    hipLaunchKernelGGL(k_reset, dim3(grid_tiles(e->n)), dim3(kBlock), 0, e->stream, A, d_mask, init_all);
    HIP_TRY(hipGetLastError());
    return QS_OK;
}

// CHECK_ENV_RAW: handle + device; CHECK_ENV: + this call uses the main stream (joins pending group work first)
#define CHECK_ENV_RAW(e)                                               \
    if (!(e)) return fail(QS_ERR_INVALID, "%s: null handle", __func__); \
    DeviceGuard guard_((e)->cfg.device);                               \
    if (!guard_.ok) return fail(QS_ERR_HIP, "%s: hipSetDevice(%d) failed", __func__, (e)->cfg.device)
#define CHECK_ENV(e)                                                   \
    CHECK_ENV_RAW(e);                                                  \
    if (!(e)->groups.empty() || (e)->chain) { int rcj_ = main_stream_entry(e); if (rcj_) return rcj_; }

}  // namespace

extern "C" {

int qs_version(void) { return QS_VERSION; }

#ifdef QS_STAMP
int qs_debug_set_stamps(void *dev_ptr, uint64_t capacity_words)
{
    g_host_stamps = (unsigned long long *)dev_ptr;
    g_host_stamp_cap = capacity_words;
    return QS_OK;
}
#endif
const char *qs_last_error(void) { return g_err; }

// Diagnostic (not in quadsim.h; used by tools/hsa_chain_exp.py only): the kernel-argument block qs_step would pass to the
// step kernel for these buffers, so that an experiment can dispatch the very same kernel through a queue of its own.
int qs_debug_step_kernargs(QsEnv *e, const float *actions, float *obs, float *reward, uint8_t *done, uint8_t *flags,
                           float *terminal_obs, void *out, uint64_t cap, uint64_t *size, int32_t *split, int64_t *tiles,
                           int64_t tile0, int64_t tile_end)
{
    if (!e || !out || !size) return fail(QS_ERR_INVALID, "qs_debug_step_kernargs: null argument");
    StepArgs A = make_args(e);
    if (tile_end > tile0) { A.tile0 = tile0; A.tile_end = tile_end; }
    A.actions = actions; A.obs = obs; A.reward = reward; A.done = done; A.flags = flags; A.term_obs = terminal_obs;
    if (cap < sizeof A) return fail(QS_ERR_INVALID, "qs_debug_step_kernargs: buffer too small (%zu needed)", sizeof A);
    memcpy(out, &A, sizeof A);
    *size = sizeof A;
    if (split) *split = e->n <= kSplitMaxEnvs ? 1 : 0;
    if (tiles) *tiles = e->tiles;
    return QS_OK;
}

int qs_config_default(QsConfig *cfg)
{
    if (!cfg) return fail(QS_ERR_INVALID, "qs_config_default: null cfg");
    memset(cfg, 0, sizeof *cfg);
    cfg->struct_size = (int32_t)sizeof(QsConfig);
    cfg->kind = QS_KIND_DOCKING_V0;
    cfg->num_envs = 1;
    cfg->device = 0;
    cfg->integrator = QS_INTEG_FROZEN;
    cfg->dt = 0.02f;
    cfg->auto_reset = 0;
    cfg->randomise = QS_RANDOMISE_NONE;
    cfg->io_space = QS_IO_DEVICE;
    cfg->seed = 0;
    cfg->env_id_offset = 0;
    cfg->mass_scale[0] = cfg->mass_scale[1] = 1.0f;
    cfg->inertia_scale[0] = cfg->inertia_scale[1] = 1.0f;
    cfg->mass = 0.18f;
    cfg->inertia[0] = 0.00025f; cfg->inertia[1] = 0.000232f; cfg->inertia[2] = 0.0003738f;
    cfg->stream = nullptr;
    cfg->external_stream = 0;
    return QS_OK;
}

int qs_create(const QsConfig *cfg, QsEnv **out)
{
    if (!cfg || !out) return fail(QS_ERR_INVALID, "qs_create: null argument");
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(QsConfig))
        return fail(QS_ERR_INVALID, "qs_create: QsConfig size mismatch (got %d, want %zu)", cfg->struct_size, sizeof(QsConfig));
    if (cfg->num_envs < 1) return fail(QS_ERR_INVALID, "qs_create: num_envs must be >= 1");
    if (cfg->num_envs > ((int64_t)1 << 31)) return fail(QS_ERR_INVALID, "qs_create: num_envs too large");
    if (cfg->kind < QS_KIND_DOCKING_V0 || cfg->kind > QS_KIND_HOVERING_V0) return fail(QS_ERR_INVALID, "qs_create: unknown env kind %d", cfg->kind);
    if ((cfg->kind == QS_KIND_DOCKING_V1 || cfg->kind == QS_KIND_HOVERING_V0) && cfg->randomise == QS_RANDOMISE_INIT)
        return fail(QS_ERR_INVALID, "qs_create: docking-v1 / hovering-v0 reset to their stored initial state; randomise must be 0");
    if (cfg->kind == QS_KIND_HOVERING_V0 && cfg->randomise != 0) return fail(QS_ERR_INVALID, "qs_create: hovering-v0 has no randomised resets");
    if (cfg->integrator != QS_INTEG_FROZEN && cfg->integrator != QS_INTEG_RK4) return fail(QS_ERR_INVALID, "qs_create: unknown integrator %d", cfg->integrator);
    if (cfg->randomise < 0 || cfg->randomise > 2) return fail(QS_ERR_INVALID, "qs_create: randomise must be 0..2");
    if (cfg->io_space != QS_IO_DEVICE && cfg->io_space != QS_IO_HOST) return fail(QS_ERR_INVALID, "qs_create: bad io_space");
    if (cfg->randomise && !(cfg->init_range[2] >= 0.0f && cfg->init_range[2] <= 1.5707964f))
        return fail(QS_ERR_INVALID, "qs_create: init_range[2] (euler half-range) must be within [0, pi/2]");
    if (!(cfg->dt > 0.0f) || !(cfg->mass > 0.0f) || !(cfg->inertia[0] > 0.0f) || !(cfg->inertia[1] > 0.0f) || !(cfg->inertia[2] > 0.0f))
        return fail(QS_ERR_INVALID, "qs_create: dt, mass and inertia must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(QS_ERR_NO_DEVICE, "qs_create: no HIP device visible (this library has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(QS_ERR_INVALID, "qs_create: device %d out of range (%d visible)", cfg->device, ndev);
    DeviceGuard guard(cfg->device);
    if (!guard.ok) return fail(QS_ERR_HIP, "qs_create: hipSetDevice(%d) failed", cfg->device);

    QsEnv *e = new (std::nothrow) QsEnv();
    if (!e) return fail(QS_ERR_NOMEM, "qs_create: out of host memory");
    e->cfg = *cfg;
    e->n = cfg->num_envs;
    e->tiles = tiles_of(e->n);
    e->per_env_params = cfg->randomise >= QS_RANDOMISE_PARAMS;
    e->obs_dim = cfg->kind == QS_KIND_HOVERING_V0 ? 13 : 12;
    int rc = QS_OK;
    auto body = [&]() -> int {
        if (cfg->external_stream) e->stream = (hipStream_t)cfg->stream;
        else { HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking)); e->own_stream = true; }
        HIP_TRY(hipEventCreate(&e->ev0));
        HIP_TRY(hipEventCreate(&e->ev1));
        const size_t st_bytes = (size_t)e->tiles * kRecWords * kTile * sizeof(float);
        const size_t par_bytes = (size_t)e->tiles * kParWords * kTile * sizeof(float) + 64;  // + scratch for nominal_obs
        HIP_TRY(hipMalloc((void **)&e->st, st_bytes));
        HIP_TRY(hipMalloc((void **)&e->par, par_bytes));
        HIP_TRY(hipMalloc((void **)&e->d_ctr, (size_t)e->tiles * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(e->d_ctr, 0, (size_t)e->tiles * sizeof(unsigned long long), e->stream));
        HIP_TRY(hipMemsetAsync(e->st, 0, st_bytes, e->stream));
        HIP_TRY(hipMemsetAsync(e->par, 0, par_bytes, e->stream));
        int r = fill_params(e);
        if (r) return r;
        // observation of the nominal reset, evaluated once by the same device code the kernels use
        hipLaunchKernelGGL(k_nominal_obs, dim3(1), dim3(1), 0, e->stream, e->par + (size_t)e->tiles * kParWords * kTile);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(e->nominal_obs, e->par + (size_t)e->tiles * kParWords * kTile, 12 * sizeof(float),
                               hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        if (cfg->kind == QS_KIND_DOCKING_V1 || cfg->kind == QS_KIND_HOVERING_V0) {
            const bool hover = cfg->kind == QS_KIND_HOVERING_V0;
            HIP_TRY(hipMalloc((void **)&e->init, (size_t)e->n * (hover ? 13 : 26) * sizeof(float)));
            hipLaunchKernelGGL(k_ctor_init, dim3(grid_flat(e->n)), dim3(kBlock), 0, e->stream, e->init, e->n, hover ? 1 : 0,
                               cfg->seed, cfg->env_id_offset);
            HIP_TRY(hipGetLastError());
        }
        // __init__: initial states (nominal / stored), q_des = identity; per-episode randomisation starts at the first reset
        StepArgs A = make_args(e);
        A.randomise = 0;
        if (cfg->kind == QS_KIND_HOVERING_V0)
            hipLaunchKernelGGL(k_hover_reset, dim3(grid_tiles(e->n)), dim3(kBlock), 0, e->stream, A, (const uint8_t *)nullptr);
        else
            hipLaunchKernelGGL(k_reset, dim3(grid_tiles(e->n)), dim3(kBlock), 0, e->stream, A, (const uint8_t *)nullptr, 1);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(e->stream));
        return QS_OK;
    };
    rc = body();
    if (rc != QS_OK) { qs_destroy(e); return rc; }
    *out = e;
    return QS_OK;
}

int qs_destroy(QsEnv *e)
{
    if (!e) return QS_OK;
    DeviceGuard guard(e->cfg.device);
    groups_destroy(e);
    if (e->chain) { (void)chain_drain(e); chain_close(e); }
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->st) (void)hipFree(e->st);
    if (e->par) (void)hipFree(e->par);
    if (e->init) (void)hipFree(e->init);
    if (e->d_ctr) (void)hipFree(e->d_ctr);
    if (e->gae_ws) (void)hipFree(e->gae_ws);
    if (e->stage) (void)hipFree(e->stage);
    if (e->hpin) (void)hipHostFree(e->hpin);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return QS_OK;
}

int qs_set_stream(QsEnv *e, void *hip_stream, int32_t external)
{
    CHECK_ENV(e);                                      // drains the private queues
    // a stream-ordered chain leaves `wait until rev == n` commands on the stream it was stepped from; they must have passed
    // before a step is issued from another stream (which would move `rev` past n)
    if (e->chain && e->chain->stream_ordered && e->chain->fwd_seq) HIP_TRY(hipStreamSynchronize(e->stream));
    // an owned stream is drained and destroyed; switching between caller-owned streams is the caller's ordering
    // problem (torch does it for stream capture) and must not synchronise
    if (e->own_stream) { HIP_TRY(hipStreamSynchronize(e->stream)); HIP_TRY(hipStreamDestroy(e->stream)); e->own_stream = false; }
    if (external) e->stream = (hipStream_t)hip_stream;
    else { HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking)); e->own_stream = true; }
    return QS_OK;
}

int qs_sync(QsEnv *e)
{
    CHECK_ENV(e);
    HIP_TRY(hipStreamSynchronize(e->stream));
    return QS_OK;
}

int qs_timer_start(QsEnv *e)
{
    CHECK_ENV(e);
    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    return QS_OK;
}

int qs_timer_stop(QsEnv *e, float *ms)
{
    CHECK_ENV(e);
    if (!ms) return fail(QS_ERR_INVALID, "qs_timer_stop: null output");
    HIP_TRY(hipEventRecord(e->ev1, e->stream));
    HIP_TRY(hipEventSynchronize(e->ev1));
    HIP_TRY(hipEventElapsedTime(ms, e->ev0, e->ev1));
    return QS_OK;
}

int qs_get_step_counter(QsEnv *e, uint64_t *k)
{
    if (!k) return fail(QS_ERR_INVALID, "qs_get_step_counter: null argument");
    CHECK_ENV(e);
    unsigned long long v = 0;
    HIP_TRY(hipMemcpyAsync(&v, e->d_ctr, sizeof v, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    *k = v;
    return QS_OK;
}

int qs_set_step_counter(QsEnv *e, uint64_t k)
{
    CHECK_ENV(e);
    hipLaunchKernelGGL(k_fill_ctr, dim3(grid_flat(e->tiles)), dim3(kBlock), 0, e->stream, e->d_ctr, e->tiles, (unsigned long long)k);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
    return QS_OK;
}

int qs_reset(QsEnv *e, const uint8_t *mask, float *obs_out)
{
    Range rg_("qs_reset");
    CHECK_ENV(e);
    const int64_t n = e->n;
    if (e->cfg.io_space == QS_IO_DEVICE) return do_reset(e, mask, obs_out, 0);
    const int64_t od = e->obs_dim;
    const size_t need = (size_t)n * (od * 4 + 1) + 1024;
    int r = ensure_stage(e, need);
    if (r) return r;
    Bounce B(e, need);
    uint8_t *d_mask = B.take<uint8_t>(n);
    B.inputs_done();
    float *d_obs = B.take<float>(n * od);
    if (mask) memcpy(B.host(d_mask), mask, n);
    if (obs_out && mask) {
        // rows of envs that are not reset keep the caller's values: seed the output slice with them
        memcpy(B.host(d_obs), obs_out, n * od * sizeof(float));
        if (!B.direct) HIP_TRY(hipMemcpyAsync(d_obs, B.host(d_obs), n * od * sizeof(float), hipMemcpyHostToDevice, e->stream));
    }
    if (mask && (r = B.push())) return r;
    r = do_reset(e, mask ? d_mask : nullptr, d_obs, 0);
    if (r) return r;
    if ((r = B.pull())) return r;
    if (obs_out) memcpy(obs_out, B.host(d_obs), n * od * sizeof(float));
    return QS_OK;
}

int qs_step_ex(QsEnv *e, const float *actions, float *obs, float *reward, uint8_t *done, uint8_t *flags, float *terminal_obs,
               float *terminal_state)
{
    Range rg_("qs_step_ex");
    CHECK_ENV_RAW(e);
    if (!actions || !obs || !reward || !done) return fail(QS_ERR_INVALID, "qs_step: actions, obs, reward and done are required");
    if (terminal_state && e->cfg.kind == QS_KIND_HOVERING_V0)
        return fail(QS_ERR_INVALID, "qs_step_ex: hovering-v0 has no terminal_state (its terminal observation IS the state)");
    if (e->chain) {
        // private-queue mode: one hand-written AQL packet behind the previous step's; nothing of the HIP stream is touched
        if (e->groups_dirty) { int rj = groups_join(e); if (rj) return rj; HIP_TRY(hipStreamSynchronize(e->stream)); }
        StepArgs A = make_args(e);
        A.actions = actions; A.obs = obs; A.reward = reward; A.done = done; A.flags = flags; A.term_obs = terminal_obs;
        A.term_state = terminal_state;
        return chain_step(e, A);
    }
    if (!e->groups.empty()) { int rcj = main_stream_entry(e); if (rcj) return rcj; }
    const int64_t n = e->n;
    StepArgs A = make_args(e);
    int r;
    if (e->cfg.io_space == QS_IO_DEVICE) {
        A.actions = actions; A.obs = obs; A.reward = reward; A.done = done; A.flags = flags; A.term_obs = terminal_obs;
        A.term_state = terminal_state;
        r = launch_env(e, A);
        if (r) return r;
    } else {
        const int64_t od = e->obs_dim;
        const size_t need = (size_t)n * (4 * 4 + od * 4 + 4 + 1 + 1 + od * 4 + 26 * 4) + 4096;
        r = ensure_stage(e, need);
        if (r) return r;
        Bounce B(e, need);
        float *d_act = B.take<float>(n * 4);
        B.inputs_done();
        float *d_obs = B.take<float>(n * od), *d_rew = B.take<float>(n);
        uint8_t *d_done = B.take<uint8_t>(n), *d_flags = B.take<uint8_t>(n);
        float *d_term = B.take<float>(n * od);
        float *d_tst = B.take<float>(n * 26);
        memcpy(B.host(d_act), actions, n * 4 * sizeof(float));
        if ((r = B.push())) return r;
        if (terminal_obs) {
            // rows of envs that did not finish keep the caller's values
            memcpy(B.host(d_term), terminal_obs, n * od * sizeof(float));
            if (!B.direct) HIP_TRY(hipMemcpyAsync(d_term, B.host(d_term), n * od * sizeof(float), hipMemcpyHostToDevice, e->stream));
        }
        if (terminal_state) {
            memcpy(B.host(d_tst), terminal_state, n * 26 * sizeof(float));
            if (!B.direct) HIP_TRY(hipMemcpyAsync(d_tst, B.host(d_tst), n * 26 * sizeof(float), hipMemcpyHostToDevice, e->stream));
        }
        A.actions = d_act; A.obs = d_obs; A.reward = d_rew; A.done = d_done; A.flags = d_flags;
        A.term_obs = terminal_obs ? d_term : nullptr;
        A.term_state = terminal_state ? d_tst : nullptr;
        r = launch_env(e, A);
        if (r) return r;
        if ((r = B.pull())) return r;
        memcpy(obs, B.host(d_obs), n * od * sizeof(float));
        memcpy(reward, B.host(d_rew), n * sizeof(float));
        memcpy(done, B.host(d_done), n);
        if (flags) memcpy(flags, B.host(d_flags), n);
        if (terminal_obs) memcpy(terminal_obs, B.host(d_term), n * od * sizeof(float));
        if (terminal_state) memcpy(terminal_state, B.host(d_tst), n * 26 * sizeof(float));
    }
    return QS_OK;
}

int qs_step(QsEnv *e, const float *actions, float *obs, float *reward, uint8_t *done, uint8_t *flags, float *terminal_obs)
{
    return qs_step_ex(e, actions, obs, reward, done, flags, terminal_obs, nullptr);
}

// ---- env groups: see the QsGroup comment above ---------------------------------------------------------------------
int qs_set_groups(QsEnv *e, int32_t groups, int32_t launcher_threads)
{
    CHECK_ENV(e);                                     // joins and thereby retires any previous grouping's work
    if (groups < 0 || groups > 64) return fail(QS_ERR_INVALID, "qs_set_groups: groups must be 0..64");
    if (e->cfg.io_space != QS_IO_DEVICE && groups > 1) return fail(QS_ERR_INVALID, "qs_set_groups: device buffers only");
    HIP_TRY(hipStreamSynchronize(e->stream));
    groups_destroy(e);
    if (groups <= 1) return QS_OK;
    if (groups > e->tiles) groups = (int32_t)e->tiles;
    HIP_TRY(hipEventCreateWithFlags(&e->fork_ev, hipEventDisableTiming));
    const int64_t base = e->tiles / groups, rem = e->tiles % groups;
    int64_t t0 = 0;
    for (int32_t i = 0; i < groups; ++i) {
        QsGroup *g = new (std::nothrow) QsGroup();
        if (!g) return fail(QS_ERR_NOMEM, "qs_set_groups: out of host memory");
        e->groups.push_back(g);
        g->env = e;
        g->index = i;
        g->tile0 = t0;
        g->tile_end = t0 + base + (i < rem ? 1 : 0);
        t0 = g->tile_end;
        g->env0 = g->tile0 * kTile;
        g->env_end = g->tile_end * kTile < e->n ? g->tile_end * kTile : e->n;
        hipError_t he = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
        if (he == hipSuccess) { g->own_stream = true; he = hipEventCreateWithFlags(&g->done_ev, hipEventDisableTiming); }
        if (he != hipSuccess) { groups_destroy(e); return fail(QS_ERR_HIP, "qs_set_groups: %s", hipGetErrorString(he)); }
    }
    for (QsGroup *g : e->groups) {
        g->threaded = launcher_threads != 0;
        if (g->threaded) g->th = std::thread(group_worker, g);
    }
    e->main_dirty = true;
    return QS_OK;
}

int qs_group_count(QsEnv *e, int32_t *groups)
{
    if (!e || !groups) return fail(QS_ERR_INVALID, "qs_group_count: null argument");
    *groups = e->groups.empty() ? 1 : (int32_t)e->groups.size();
    return QS_OK;
}

int qs_group_range(QsEnv *e, int32_t g, int64_t *env_begin, int64_t *env_end)
{
    if (!e || !env_begin || !env_end) return fail(QS_ERR_INVALID, "qs_group_range: null argument");
    if (e->groups.empty()) {
        if (g != 0) return fail(QS_ERR_INVALID, "qs_group_range: group %d out of range", g);
        *env_begin = 0; *env_end = e->n;
        return QS_OK;
    }
    if (g < 0 || g >= (int32_t)e->groups.size()) return fail(QS_ERR_INVALID, "qs_group_range: group %d out of range", g);
    *env_begin = e->groups[g]->env0; *env_end = e->groups[g]->env_end;
    return QS_OK;
}

int qs_group_stream(QsEnv *e, int32_t g, void **hip_stream)
{
    if (!e || !hip_stream) return fail(QS_ERR_INVALID, "qs_group_stream: null argument");
    if (g < 0 || g >= (int32_t)e->groups.size()) return fail(QS_ERR_INVALID, "qs_group_stream: group %d out of range", g);
    *hip_stream = (void *)e->groups[g]->stream;
    return QS_OK;
}

int qs_group_set_stream(QsEnv *e, int32_t g, void *hip_stream)
{
    CHECK_ENV(e);                                     // joins: nothing of this group is pending on its old stream afterwards
    if (g < 0 || g >= (int32_t)e->groups.size()) return fail(QS_ERR_INVALID, "qs_group_set_stream: group %d out of range", g);
    QsGroup *G = e->groups[g];
    HIP_TRY(hipStreamSynchronize(G->stream));
    if (G->own_stream) { HIP_TRY(hipStreamDestroy(G->stream)); G->own_stream = false; }
    G->stream = (hipStream_t)hip_stream;
    return QS_OK;
}

int qs_groups_fork(QsEnv *e)
{
    CHECK_ENV_RAW(e);
    return groups_fork(e);
}

int qs_groups_join(QsEnv *e)
{
    CHECK_ENV_RAW(e);
    return groups_join(e);
}

static int step_group_post(QsEnv *e, QsGroup *G, const float *actions, float *obs, float *reward, uint8_t *done, uint8_t *flags,
                           float *terminal_obs, float *terminal_state, bool group_local)
{
    QsGroup::Req r;
    r.type = QsGroup::REQ_LAUNCH;
    r.ev = nullptr;
    r.A = make_args(e);
    r.A.tile0 = G->tile0; r.A.tile_end = G->tile_end;
    if (group_local) { r.A.io_env0 = G->env0; r.A.io_n = G->env_end - G->env0; }
    r.A.actions = actions; r.A.obs = obs; r.A.reward = reward; r.A.done = done; r.A.flags = flags;
    r.A.term_obs = terminal_obs; r.A.term_state = terminal_state;
    return group_post(G, r);
}

int qs_step_group(QsEnv *e, int32_t g, const float *actions, float *obs, float *reward, uint8_t *done, uint8_t *flags,
                  float *terminal_obs, float *terminal_state)
{
    Range rg_("qs_step_group");
    CHECK_ENV_RAW(e);
    if (g < 0 || g >= (int32_t)e->groups.size()) return fail(QS_ERR_INVALID, "qs_step_group: group %d out of range (qs_set_groups first)", g);
    if (!actions || !obs || !reward || !done) return fail(QS_ERR_INVALID, "qs_step_group: actions, obs, reward and done are required");
    if (terminal_state && e->cfg.kind == QS_KIND_HOVERING_V0) return fail(QS_ERR_INVALID, "qs_step_group: hovering-v0 has no terminal_state");
    if (e->main_dirty) { int rc = groups_fork(e); if (rc) return rc; }
    e->groups_dirty = true;
    int rc = step_group_post(e, e->groups[g], actions, obs, reward, done, flags, terminal_obs, terminal_state, true);
    // with a launcher thread the record is only POSTED here; the contract (quadsim.h) lets the caller enqueue the group's
    // policy on the group's stream right after this call, so the launch has to be on that stream before the call returns
    group_wait_issued(e->groups[g]);
    return rc;
}

int qs_step_groups(QsEnv *e, const float *actions, float *obs, float *reward, uint8_t *done, uint8_t *flags, float *terminal_obs,
                   float *terminal_state)
{
    Range rg_("qs_step_groups");
    CHECK_ENV_RAW(e);
    if (e->groups.empty() || e->chain) return qs_step_ex(e, actions, obs, reward, done, flags, terminal_obs, terminal_state);
    if (!actions || !obs || !reward || !done) return fail(QS_ERR_INVALID, "qs_step_groups: actions, obs, reward and done are required");
    if (terminal_state && e->cfg.kind == QS_KIND_HOVERING_V0) return fail(QS_ERR_INVALID, "qs_step_groups: hovering-v0 has no terminal_state");
    if (e->main_dirty) { int rc = groups_fork(e); if (rc) return rc; }
    e->groups_dirty = true;
    for (QsGroup *G : e->groups) {
        int rc = step_group_post(e, G, actions, obs, reward, done, flags, terminal_obs, terminal_state, false);
        if (rc) return rc;
    }
    // the launcher threads issue the G launches concurrently; all of them are on their streams when the call returns
    for (QsGroup *G : e->groups) group_wait_issued(G);
    return QS_OK;
}

int qs_rollout(QsEnv *e, int64_t T, const float *actions, float *obs, float *reward, uint8_t *done, uint8_t *flags)
{
    Range rg_("qs_rollout");
    CHECK_ENV(e);
    if (T < 1) return fail(QS_ERR_INVALID, "qs_rollout: T must be >= 1");
    if (!obs || !reward || !done) return fail(QS_ERR_INVALID, "qs_rollout: obs, reward and done are required");
    if (!e->cfg.auto_reset) return fail(QS_ERR_INVALID, "qs_rollout: requires auto_reset (a roll-out runs through episode ends)");
    const int64_t n = e->n, tn = T * n;
    StepArgs A = make_args(e);
    A.T = T;
    int r;
    if (e->cfg.io_space == QS_IO_DEVICE) {
        A.actions = actions; A.obs = obs; A.reward = reward; A.done = done; A.flags = flags;
        r = launch_env(e, A);
        if (r) return r;
    } else {
        const int64_t od = e->obs_dim;
        r = ensure_stage(e, (size_t)tn * (4 * 4 + od * 4 + 4 + 1 + 1) + 4096);
        if (r) return r;
        Stage S{(char *)e->stage};
        float *d_act = S.take<float>(tn * 4), *d_obs = S.take<float>(tn * od), *d_rew = S.take<float>(tn);
        uint8_t *d_done = S.take<uint8_t>(tn), *d_flags = S.take<uint8_t>(tn);
        if (actions) HIP_TRY(hipMemcpyAsync(d_act, actions, tn * 4 * sizeof(float), hipMemcpyHostToDevice, e->stream));
        A.actions = actions ? d_act : nullptr; A.obs = d_obs; A.reward = d_rew; A.done = d_done; A.flags = d_flags;
        r = launch_env(e, A);
        if (r) return r;
        HIP_TRY(hipMemcpyAsync(obs, d_obs, tn * od * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(reward, d_rew, tn * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(done, d_done, tn, hipMemcpyDeviceToHost, e->stream));
        if (flags) HIP_TRY(hipMemcpyAsync(flags, d_flags, tn, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    return QS_OK;
}

int qs_rollout_slab(QsEnv *e, int64_t T, const float *actions, float *slab, uint8_t *flags)
{
    Range rg_("qs_rollout_slab");
    CHECK_ENV(e);
    if (T < 1 || !slab) return fail(QS_ERR_INVALID, "qs_rollout_slab: bad arguments");
    if (!e->cfg.auto_reset) return fail(QS_ERR_INVALID, "qs_rollout_slab: requires auto_reset");
    if (e->cfg.io_space != QS_IO_DEVICE) return fail(QS_ERR_INVALID, "qs_rollout_slab: device buffers only");
    if (e->cfg.kind == QS_KIND_HOVERING_V0) return fail(QS_ERR_INVALID, "qs_rollout_slab: docking envs only");
    StepArgs A = make_args(e);
    A.T = T; A.actions = actions; A.slab = slab; A.flags = flags;
    return launch_env(e, A);
}

int qs_rollout_stepwise(QsEnv *e, int64_t T, const float *actions, float *obs, float *reward, uint8_t *done, uint8_t *flags)
{
    Range rg_("qs_rollout_stepwise");
    CHECK_ENV_RAW(e);
    if (e->chain) {                                   // as qs_step_ex: packets behind the previous ones, no drain
        if (e->groups_dirty) { int rj = groups_join(e); if (rj) return rj; HIP_TRY(hipStreamSynchronize(e->stream)); }
    } else if (!e->groups.empty()) { int rcj = main_stream_entry(e); if (rcj) return rcj; }
    if (T < 1 || !actions || !obs || !reward || !done) return fail(QS_ERR_INVALID, "qs_rollout_stepwise: bad arguments");
    if (!e->cfg.auto_reset) return fail(QS_ERR_INVALID, "qs_rollout_stepwise: requires auto_reset");
    if (e->cfg.io_space != QS_IO_DEVICE) return fail(QS_ERR_INVALID, "qs_rollout_stepwise: device buffers only");
    const int64_t n = e->n;
    StepArgs A = make_args(e);
    std::vector<StepArgs> steps;
    if (e->chain) steps.reserve((size_t)T);
    for (int64_t t = 0; t < T; ++t) {
        A.actions = actions + t * n * 4;
        A.obs = obs + t * n * e->obs_dim;
        A.reward = reward + t * n;
        A.done = done + t * n;
        A.flags = flags ? flags + t * n : nullptr;
        if (e->chain) { steps.push_back(A); continue; }
        int r = launch_env(e, A);
        if (r) return r;
    }
    // queue mode: T packets per queue behind ONE hand-shake with the handle's stream (stream-ordered chains), or drained by the
    // next entry point (host-ordered chains)
    return e->chain ? chain_submit(e, steps.data(), T) : QS_OK;
}

int qs_fill_random_actions(QsEnv *e, int64_t T, uint64_t step0, float *actions)
{
    CHECK_ENV(e);
    if (T < 1 || !actions) return fail(QS_ERR_INVALID, "qs_fill_random_actions: bad arguments");
    const int64_t tn = T * e->n;
    float *d = actions;
    if (e->cfg.io_space == QS_IO_HOST) {
        int r = ensure_stage(e, (size_t)tn * 16 + 1024);
        if (r) return r;
        d = (float *)e->stage;
    }
    hipLaunchKernelGGL(k_fill_actions, dim3(grid_flat(tn)), dim3(kBlock), 0, e->stream, d, e->n, T, e->cfg.seed,
                       e->cfg.env_id_offset, step0);
    HIP_TRY(hipGetLastError());
    if (e->cfg.io_space == QS_IO_HOST) {
        HIP_TRY(hipMemcpyAsync(actions, d, tn * 16, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    return QS_OK;
}

static int state_io(QsEnv *e, bool to_user, float *chaser, float *target, float *u_prev, float *qdes, float *ls, float *t)
{
    const int64_t n = e->n;
    const int64_t words[6] = {13, 13, 8, 4, 1, 1};
    float *user[6] = {chaser, target, u_prev, qdes, ls, t};
    StateIO io{chaser, target, u_prev, qdes, ls, t};
    if (e->cfg.io_space == QS_IO_DEVICE) {
        if (to_user) hipLaunchKernelGGL(k_state_io<true>, dim3(grid_tiles(n)), dim3(kBlock), 0, e->stream, e->st, n, io);
        else hipLaunchKernelGGL(k_state_io<false>, dim3(grid_tiles(n)), dim3(kBlock), 0, e->stream, e->st, n, io);
        HIP_TRY(hipGetLastError());
        return QS_OK;
    }
    const size_t need = (size_t)n * 40 * 4 + 4096;
    int r = ensure_stage(e, need);
    if (r) return r;
    Bounce B(e, need);
    float *dev[6];
    for (int i = 0; i < 6; ++i) {
        dev[i] = B.take<float>(n * words[i]);
        if (user[i] && !to_user) memcpy(B.host(dev[i]), user[i], n * words[i] * 4);
    }
    if (!to_user) { B.inputs_done(); if ((r = B.push())) return r; }
    io = StateIO{chaser ? dev[0] : nullptr, target ? dev[1] : nullptr, u_prev ? dev[2] : nullptr,
                 qdes ? dev[3] : nullptr, ls ? dev[4] : nullptr, t ? dev[5] : nullptr};
    if (to_user) hipLaunchKernelGGL(k_state_io<true>, dim3(grid_tiles(n)), dim3(kBlock), 0, e->stream, e->st, n, io);
    else hipLaunchKernelGGL(k_state_io<false>, dim3(grid_tiles(n)), dim3(kBlock), 0, e->stream, e->st, n, io);
    HIP_TRY(hipGetLastError());
    if ((r = B.pull())) return r;
    if (to_user)
        for (int i = 0; i < 6; ++i)
            if (user[i]) memcpy(user[i], B.host(dev[i]), n * words[i] * 4);
    return QS_OK;
}

int qs_get_state(QsEnv *e, float *chaser, float *target, float *u_prev, float *qdes, float *last_shaping, float *t)
{
    CHECK_ENV(e);
    return state_io(e, true, chaser, target, u_prev, qdes, last_shaping, t);
}

int qs_set_state(QsEnv *e, const float *chaser, const float *target, const float *u_prev, const float *qdes,
                 const float *last_shaping, const float *t)
{
    CHECK_ENV(e);
    return state_io(e, false, (float *)chaser, (float *)target, (float *)u_prev, (float *)qdes, (float *)last_shaping, (float *)t);
}

static int par_io(QsEnv *e, bool to_user, float *mass, float *inertia)
{
    const int64_t n = e->n;
    float *dm = mass, *di = inertia;
    if (e->cfg.io_space == QS_IO_HOST) {
        int r = ensure_stage(e, (size_t)n * 16 + 1024);
        if (r) return r;
        Stage S{(char *)e->stage};
        dm = S.take<float>(n); di = S.take<float>(n * 3);
        if (!to_user) {
            if (mass) HIP_TRY(hipMemcpyAsync(dm, mass, n * 4, hipMemcpyHostToDevice, e->stream));
            if (inertia) HIP_TRY(hipMemcpyAsync(di, inertia, n * 12, hipMemcpyHostToDevice, e->stream));
        }
        if (!mass) dm = nullptr;
        if (!inertia) di = nullptr;
    }
    if (to_user) hipLaunchKernelGGL(k_par_io<true>, dim3(grid_tiles(n)), dim3(kBlock), 0, e->stream, e->par, n, dm, di);
    else hipLaunchKernelGGL(k_par_io<false>, dim3(grid_tiles(n)), dim3(kBlock), 0, e->stream, e->par, n, dm, di);
    HIP_TRY(hipGetLastError());
    if (e->cfg.io_space == QS_IO_HOST) {
        if (to_user) {
            if (mass) HIP_TRY(hipMemcpyAsync(mass, dm, n * 4, hipMemcpyDeviceToHost, e->stream));
            if (inertia) HIP_TRY(hipMemcpyAsync(inertia, di, n * 12, hipMemcpyDeviceToHost, e->stream));
        }
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    return QS_OK;
}

int qs_set_params(QsEnv *e, const float *mass, const float *inertia)
{
    CHECK_ENV(e);
    if (!mass && !inertia) return fail(QS_ERR_INVALID, "qs_set_params: nothing to set");
    int r = par_io(e, false, (float *)mass, (float *)inertia);
    if (r) return r;
    e->per_env_params = true;
    return QS_OK;
}

int qs_get_params(QsEnv *e, float *mass, float *inertia)
{
    CHECK_ENV(e);
    return par_io(e, true, mass, inertia);
}

int qs_obs_dim(QsEnv *e, int32_t *dim)
{
    if (!e || !dim) return fail(QS_ERR_INVALID, "qs_obs_dim: null argument");
    *dim = e->obs_dim;
    return QS_OK;
}

static int init_io(QsEnv *e, bool to_user, float *chaser, float *target)
{
    const int64_t n = e->n;
    const bool hover = e->cfg.kind == QS_KIND_HOVERING_V0;
    const int64_t w = hover ? 13 : 26;
    // the store is row-major [n][w]; user arrays are [n][13] each: strided 2-D copies
    const hipMemcpyKind kd = e->cfg.io_space == QS_IO_HOST ? (to_user ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice)
                                                             : hipMemcpyDeviceToDevice;
    if (chaser) {
        if (to_user) HIP_TRY(hipMemcpy2DAsync(chaser, 52, e->init, w * 4, 52, n, kd, e->stream));
        else HIP_TRY(hipMemcpy2DAsync(e->init, w * 4, chaser, 52, 52, n, kd, e->stream));
    }
    if (target && !hover) {
        if (to_user) HIP_TRY(hipMemcpy2DAsync(target, 52, e->init + 13, w * 4, 52, n, kd, e->stream));
        else HIP_TRY(hipMemcpy2DAsync(e->init + 13, w * 4, target, 52, 52, n, kd, e->stream));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    return QS_OK;
}

int qs_set_init_state(QsEnv *e, const float *chaser_init, const float *target_init)
{
    CHECK_ENV(e);
    if (!chaser_init) return fail(QS_ERR_INVALID, "qs_set_init_state: chaser_init is required");
    const bool hover = e->cfg.kind == QS_KIND_HOVERING_V0;
    if (!e->init) {
        // first use on a docking-v0/v2 handle: start from the nominal pair for every env
        HIP_TRY(hipStreamSynchronize(e->stream));
        HIP_TRY(hipMalloc((void **)&e->init, (size_t)e->n * 26 * sizeof(float)));
        hipLaunchKernelGGL(k_fill_init_nominal, dim3(grid_flat(e->n)), dim3(kBlock), 0, e->stream, e->init, e->n);
        HIP_TRY(hipGetLastError());
    }
    (void)hover;
    return init_io(e, false, (float *)chaser_init, (float *)target_init);
}

int qs_get_init_state(QsEnv *e, float *chaser_init, float *target_init)
{
    CHECK_ENV(e);
    if (!e->init) return fail(QS_ERR_INVALID, "qs_get_init_state: this handle resets to the nominal / randomised state (no stored initial states)");
    return init_io(e, true, chaser_init, target_init);
}

// ---- roll-out post-processing -------------------------------------------
int qs_gae(QsEnv *e, int64_t T, int64_t n, const float *rewards, const float *values, const uint8_t *dones,
           const float *last_values, const uint8_t *last_dones, float gamma, float lam, float *advs, float *returns)
{
    Range rg_("qs_gae");
    CHECK_ENV(e);
    if (T < 1 || n < 1 || !rewards || !values || !dones || !last_values || !last_dones || !advs || !returns)
        return fail(QS_ERR_INVALID, "qs_gae: bad arguments");
    if (e->cfg.io_space != QS_IO_DEVICE) return fail(QS_ERR_INVALID, "qs_gae: device buffers only");
    GaeArgs G;
    G.rewards = rewards; G.values = values; G.last_values = last_values; G.dones = dones; G.last_dones = last_dones;
    G.advs = advs; G.returns = returns;
    G.T = T; G.N = n; G.C = (T + kGaeChunk - 1) / kGaeChunk;
    G.gamma = gamma; G.lam = lam;
    if (n >= 16384) {
        // wide batch: enough lanes to cover the latency of a T-long serial walk; read everything once
        G.ws = nullptr;
        hipLaunchKernelGGL(k_gae_serial, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, G);
        HIP_TRY(hipGetLastError());
        return QS_OK;
    }
    const size_t need = (size_t)2 * G.C * n;
    if (e->gae_ws_floats < need) {
        HIP_TRY(hipStreamSynchronize(e->stream));
        if (e->gae_ws) HIP_TRY(hipFree(e->gae_ws));
        e->gae_ws = nullptr; e->gae_ws_floats = 0;
        HIP_TRY(hipMalloc((void **)&e->gae_ws, need * sizeof(float)));
        e->gae_ws_floats = need;
    }
    G.ws = e->gae_ws;
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)G.C);
    hipLaunchKernelGGL(k_gae_reduce, grid, dim3(256), 0, e->stream, G);
    hipLaunchKernelGGL(k_gae_apply, grid, dim3(256), 0, e->stream, G);
    HIP_TRY(hipGetLastError());
    return QS_OK;
}

int qs_swap_and_flatten(QsEnv *e, int64_t T, int64_t n, int64_t d, const float *in, float *out)
{
    Range rg_("qs_swap_and_flatten");
    CHECK_ENV(e);
    if (T < 1 || n < 1 || !in || !out) return fail(QS_ERR_INVALID, "qs_swap_and_flatten: bad arguments");
    if (e->cfg.io_space != QS_IO_DEVICE) return fail(QS_ERR_INVALID, "qs_swap_and_flatten: device buffers only");
    dim3 grid((unsigned)((n + 31) / 32), (unsigned)((T + 31) / 32));
    switch (d) {
        case 1: hipLaunchKernelGGL(k_swap_flatten<1>, grid, dim3(256), 0, e->stream, in, out, T, n); break;
        case 4: hipLaunchKernelGGL(k_swap_flatten_v4<1>, grid, dim3(256), 0, e->stream, (const float4 *)in, (float4 *)out, T, n); break;
        case 12: hipLaunchKernelGGL(k_swap_flatten_v4<3>, grid, dim3(256), 0, e->stream, (const float4 *)in, (float4 *)out, T, n); break;
        case 13: hipLaunchKernelGGL(k_swap_flatten<13>, grid, dim3(256), 0, e->stream, in, out, T, n); break;
        default: return fail(QS_ERR_INVALID, "qs_swap_and_flatten: row width %lld not supported (1, 4, 12, 13)", (long long)d);
    }
    HIP_TRY(hipGetLastError());
    return QS_OK;
}

int qs_swap_and_flatten_u8(QsEnv *e, int64_t T, int64_t n, const uint8_t *in, uint8_t *out)
{
    CHECK_ENV(e);
    if (T < 1 || n < 1 || !in || !out) return fail(QS_ERR_INVALID, "qs_swap_and_flatten_u8: bad arguments");
    if (e->cfg.io_space != QS_IO_DEVICE) return fail(QS_ERR_INVALID, "qs_swap_and_flatten_u8: device buffers only");
    dim3 grid((unsigned)((n + 31) / 32), (unsigned)((T + 31) / 32));
    hipLaunchKernelGGL((k_swap_flatten<1, uint8_t>), grid, dim3(256), 0, e->stream, in, out, T, n);
    HIP_TRY(hipGetLastError());
    return QS_OK;
}

int qs_gae_flatten(QsEnv *e, int64_t T, int64_t n, const float *rewards, const float *values, const float *neglogp,
                   const uint8_t *dones, const float *last_values, const uint8_t *last_dones, float gamma, float lam,
                   float *flat_returns, float *flat_values, float *flat_neglogp, float *flat_rewards, uint8_t *flat_masks,
                   float *advs, float *returns)
{
    Range rg_("qs_gae_flatten");
    CHECK_ENV(e);
    if (T < 1 || n < 1 || !rewards || !values || !dones || !last_values || !last_dones || !flat_returns || !flat_values ||
        !flat_rewards || !flat_masks)
        return fail(QS_ERR_INVALID, "qs_gae_flatten: bad arguments");
    if ((neglogp == nullptr) != (flat_neglogp == nullptr)) return fail(QS_ERR_INVALID, "qs_gae_flatten: neglogp and flat_neglogp go together");
    if ((advs == nullptr) != (returns == nullptr)) return fail(QS_ERR_INVALID, "qs_gae_flatten: advs and returns go together");
    if (e->cfg.io_space != QS_IO_DEVICE) return fail(QS_ERR_INVALID, "qs_gae_flatten: device buffers only");
    GaeFlatArgs G;
    G.rewards = rewards; G.values = values; G.neglogp = neglogp; G.last_values = last_values;
    G.dones = dones; G.last_dones = last_dones;
    G.f_returns = flat_returns; G.f_values = flat_values; G.f_neglogp = flat_neglogp; G.f_rewards = flat_rewards;
    G.f_masks = flat_masks; G.advs = advs; G.returns = returns;
    G.T = T; G.N = n; G.gamma = gamma; G.lam = lam;
    hipLaunchKernelGGL(k_gae_flatten, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, G);
    HIP_TRY(hipGetLastError());
    return QS_OK;
}

int qs_episode_stats(QsEnv *e, int64_t T, int64_t n, const float *rewards, const uint8_t *dones, const uint8_t *last_dones,
                     float *ep_ret, int32_t *ep_len, uint64_t *count, int64_t cap, int64_t *out_key, float *out_ret, int32_t *out_len)
{
    Range rg_("qs_episode_stats");
    CHECK_ENV(e);
    if (T < 1 || n < 1 || cap < 0 || !rewards || !dones || !last_dones || !ep_ret || !ep_len || !count || (cap > 0 && (!out_key || !out_ret || !out_len)))
        return fail(QS_ERR_INVALID, "qs_episode_stats: bad arguments");
    if (e->cfg.io_space != QS_IO_DEVICE) return fail(QS_ERR_INVALID, "qs_episode_stats: device buffers only");
    EpisodeArgs E;
    E.rewards = rewards; E.dones = dones; E.last_dones = last_dones; E.ep_ret = ep_ret; E.ep_len = ep_len;
    E.count = (unsigned long long *)count; E.out_key = out_key; E.out_ret = out_ret; E.out_len = out_len;
    E.T = T; E.N = n; E.cap = cap;
    HIP_TRY(hipMemsetAsync(count, 0, sizeof(uint64_t), e->stream));
    hipLaunchKernelGGL(k_episode_stats, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, E);
    HIP_TRY(hipGetLastError());
    return QS_OK;
}

int qs_policy_rollout(QsEnv *e, int64_t T, const float *wt1, const float *b1, const float *wt2, const float *b2,
                      const float *wt3, const float *b3, float *obs, float *reward, uint8_t *done, uint8_t *flags, float *actions)
{
    Range rg_("qs_policy_rollout");
    CHECK_ENV(e);
    if (T < 1 || !wt1 || !b1 || !wt2 || !b2 || !wt3 || !b3 || !obs || !reward || !done)
        return fail(QS_ERR_INVALID, "qs_policy_rollout: bad arguments");
    if (e->cfg.io_space != QS_IO_DEVICE) return fail(QS_ERR_INVALID, "qs_policy_rollout: device buffers only");
    if (!e->cfg.auto_reset) return fail(QS_ERR_INVALID, "qs_policy_rollout: requires auto_reset");
    if (e->cfg.kind == QS_KIND_HOVERING_V0 || e->per_env_params || e->init || e->cfg.randomise > 1)
        return fail(QS_ERR_INVALID, "qs_policy_rollout: docking-v0/v2 with nominal or rocRAND-initialised resets only");
    StepArgs A = make_args(e);
    A.T = T; A.obs = obs; A.reward = reward; A.done = done; A.flags = flags;
    MlpArgs M{wt1, b1, wt2, b2, wt3, b3};
    const unsigned grid = grid_tiles(e->n);
    const bool fr = e->cfg.integrator == QS_INTEG_FROZEN;
    const int rm = e->cfg.randomise;
    if (fr && rm == 0) hipLaunchKernelGGL((k_policy_rollout<0, 0>), dim3(grid), dim3(kBlock), 0, e->stream, A, M, actions);
    else if (fr) hipLaunchKernelGGL((k_policy_rollout<0, 1>), dim3(grid), dim3(kBlock), 0, e->stream, A, M, actions);
    else if (rm == 0) hipLaunchKernelGGL((k_policy_rollout<1, 0>), dim3(grid), dim3(kBlock), 0, e->stream, A, M, actions);
    else hipLaunchKernelGGL((k_policy_rollout<1, 1>), dim3(grid), dim3(kBlock), 0, e->stream, A, M, actions);
    HIP_TRY(hipGetLastError());
    return QS_OK;
}

int qs_policy_rollout_fast(QsEnv *e, int64_t T, const void *packed_weights, float *obs, float *reward, uint8_t *done,
                           uint8_t *flags, float *actions)
{
    Range rg_("qs_policy_rollout_fast");
    CHECK_ENV(e);
    if (T < 1 || !packed_weights || !obs || !reward || !done) return fail(QS_ERR_INVALID, "qs_policy_rollout_fast: bad arguments");
    if (e->cfg.io_space != QS_IO_DEVICE) return fail(QS_ERR_INVALID, "qs_policy_rollout_fast: device buffers only");
    if (!e->cfg.auto_reset) return fail(QS_ERR_INVALID, "qs_policy_rollout_fast: requires auto_reset");
    if (e->cfg.kind == QS_KIND_HOVERING_V0 || e->per_env_params || e->init || e->cfg.randomise > 1)
        return fail(QS_ERR_INVALID, "qs_policy_rollout_fast: docking-v0/v2 with nominal or rocRAND-initialised resets only");
    if (((uintptr_t)packed_weights & 15) != 0) return fail(QS_ERR_INVALID, "qs_policy_rollout_fast: packed weights must be 16-byte aligned");
    StepArgs A = make_args(e);
    A.T = T; A.obs = obs; A.reward = reward; A.done = done; A.flags = flags;
    const uint4 *blob = (const uint4 *)packed_weights;
    const unsigned grid = grid_tiles(e->n);
    const bool fr = e->cfg.integrator == QS_INTEG_FROZEN;
    const int rm = e->cfg.randomise;
    if (fr && rm == 0) hipLaunchKernelGGL((k_policy_rollout_fast<0, 0>), dim3(grid), dim3(kBlock), 0, e->stream, A, blob, actions);
    else if (fr) hipLaunchKernelGGL((k_policy_rollout_fast<0, 1>), dim3(grid), dim3(kBlock), 0, e->stream, A, blob, actions);
    else if (rm == 0) hipLaunchKernelGGL((k_policy_rollout_fast<1, 0>), dim3(grid), dim3(kBlock), 0, e->stream, A, blob, actions);
    else hipLaunchKernelGGL((k_policy_rollout_fast<1, 1>), dim3(grid), dim3(kBlock), 0, e->stream, A, blob, actions);
    HIP_TRY(hipGetLastError());
    return QS_OK;
}

int qs_policy_rollout_fast_blob_bytes(void) { return kFastBlobBytes; }

// 1: the one-wave-per-tile Runner kernels, 0: the role-split ones (default; QUADSIM_RUNNER_SERIAL=1 or the diagnostic entry
// below select the former for A/B runs and for the bit-identity test)
static std::atomic<int> &runner_serial_flag()
{
    static std::atomic<int> flag{[] { const char *v = getenv("QUADSIM_RUNNER_SERIAL"); return (v && v[0] == '1') ? 1 : 0; }()};
    return flag;
}

static int runner_launch(QsEnv *e, const char *who, int64_t T, const float logstd[4], int squash, const AcArgs *net,
                         const void *blob, const float *noise, const uint8_t *dones_in, float *mb_obs, float *mb_actions,
                         float *mb_values, float *mb_neglogp, uint8_t *mb_dones, float *mb_rewards, uint8_t *mb_flags,
                         float *last_obs, float *last_values, uint8_t *last_dones)
{
    Range rg_(who);
    if (T < 1 || !mb_obs || !mb_actions || !mb_values || !mb_neglogp || !mb_dones || !mb_rewards || !last_values || !last_dones)
        return fail(QS_ERR_INVALID, "%s: bad arguments", who);
    if (e->cfg.io_space != QS_IO_DEVICE) return fail(QS_ERR_INVALID, "%s: device buffers only", who);
    if (!e->cfg.auto_reset) return fail(QS_ERR_INVALID, "%s: requires auto_reset", who);
    if (e->cfg.kind == QS_KIND_HOVERING_V0 || e->init)
        return fail(QS_ERR_INVALID, "%s: docking-v0/v2 with nominal or rocRAND resets only (no stored initial states)", who);
    StepArgs A = make_args(e);
    A.T = T; A.obs = mb_obs; A.reward = mb_rewards; A.done = mb_dones; A.flags = mb_flags;
    RunnerArgs R{};
    if (net) R.net = *net;
    R.blob = (const uint4 *)blob;
    double ls = 0.0;
    for (int i = 0; i < 4; ++i) {
        if (!(logstd[i] == logstd[i])) return fail(QS_ERR_INVALID, "%s: logstd is NaN", who);
        R.std[i] = expf(logstd[i]);
        R.inv_std[i] = 1.0f / R.std[i];
        ls += (double)logstd[i];
    }
    R.nl_const = (float)(0.5 * 1.8378770664093453 * 4.0 + ls);      // 0.5 log(2 pi) d + sum logstd
    R.squash = squash;
    R.noise = noise; R.dones_in = dones_in;
    R.actions = mb_actions; R.values = mb_values; R.neglogp = mb_neglogp;
    R.last_obs = last_obs; R.last_values = last_values; R.last_dones = last_dones;
    R.env_major = e->runner_env_major ? 1 : 0;
    const unsigned grid = grid_tiles(e->n);
    const bool fr = e->cfg.integrator == QS_INTEG_FROZEN;
    const int rm = e->cfg.randomise;
    const bool params = e->per_env_params || rm == 2;
#define QS_RUNNER_GO(I, RM, PA, FAST) hipLaunchKernelGGL((k_runner_rollout<I, RM, PA, FAST>), dim3(grid), dim3(kBlock), 0, e->stream, A, R)
#define QS_RUNNER_INTEG(I, FAST)                                  \
    do {                                                          \
        if (rm == 2) QS_RUNNER_GO(I, 2, true, FAST);              \
        else if (rm == 1 && params) QS_RUNNER_GO(I, 1, true, FAST);  \
        else if (rm == 1) QS_RUNNER_GO(I, 1, false, FAST);        \
        else if (params) QS_RUNNER_GO(I, 0, true, FAST);          \
        else QS_RUNNER_GO(I, 0, false, FAST);                     \
    } while (0)
    // the role-split kernel (matrix waves + env waves); QUADSIM_RUNNER_SERIAL=1 keeps the one-wave-per-tile kernel for A/B
    // runs (same results bit for bit: the same instruction sequences on the same operands)
    const bool serial_fast = runner_serial_flag().load(std::memory_order_relaxed) != 0;
#define QS_RUNNER_SPLIT_GO(I, RM, PA, FAST) hipLaunchKernelGGL((k_runner_split<I, RM, PA, FAST>), dim3(grid), dim3(2 * kBlock), 0, e->stream, A, R)
#define QS_RUNNER_SPLIT(I, FAST)                                        \
    do {                                                                \
        if (rm == 2) QS_RUNNER_SPLIT_GO(I, 2, true, FAST);              \
        else if (rm == 1 && params) QS_RUNNER_SPLIT_GO(I, 1, true, FAST);  \
        else if (rm == 1) QS_RUNNER_SPLIT_GO(I, 1, false, FAST);        \
        else if (params) QS_RUNNER_SPLIT_GO(I, 0, true, FAST);          \
        else QS_RUNNER_SPLIT_GO(I, 0, false, FAST);                     \
    } while (0)
    if (!serial_fast) {
        if (blob) { if (fr) QS_RUNNER_SPLIT(0, true); else QS_RUNNER_SPLIT(1, true); }
        else { if (fr) QS_RUNNER_SPLIT(0, false); else QS_RUNNER_SPLIT(1, false); }
    }
    else if (blob) { if (fr) QS_RUNNER_INTEG(0, true); else QS_RUNNER_INTEG(1, true); }
    else { if (fr) QS_RUNNER_INTEG(0, false); else QS_RUNNER_INTEG(1, false); }
#undef QS_RUNNER_SPLIT
#undef QS_RUNNER_SPLIT_GO
#undef QS_RUNNER_INTEG
#undef QS_RUNNER_GO
    HIP_TRY(hipGetLastError());
    return QS_OK;
}

int qs_runner_rollout(QsEnv *e, int64_t T, const QsActorCritic *pol, const float *noise, const uint8_t *dones_in,
                      float *mb_obs, float *mb_actions, float *mb_values, float *mb_neglogp, uint8_t *mb_dones,
                      float *mb_rewards, uint8_t *mb_flags, float *last_obs, float *last_values, uint8_t *last_dones)
{
    CHECK_ENV(e);
    if (!pol) return fail(QS_ERR_INVALID, "qs_runner_rollout: bad arguments");
    if (pol->struct_size != sizeof(QsActorCritic)) return fail(QS_ERR_INVALID, "qs_runner_rollout: QsActorCritic.struct_size mismatch");
    if (!pol->wt1 || !pol->b1 || !pol->wt2 || !pol->b2 || !pol->wt3 || !pol->b3 || !pol->wtv2 || !pol->bv2 || !pol->wtv3 || !pol->bv3)
        return fail(QS_ERR_INVALID, "qs_runner_rollout: null weight pointer");
    const AcArgs net{pol->wt1, pol->b1, pol->wt2, pol->b2, pol->wt3, pol->b3, pol->wtv2, pol->bv2, pol->wtv3, pol->bv3};
    return runner_launch(e, "qs_runner_rollout", T, pol->logstd, pol->squash, &net, nullptr, noise, dones_in, mb_obs, mb_actions,
                         mb_values, mb_neglogp, mb_dones, mb_rewards, mb_flags, last_obs, last_values, last_dones);
}

int qs_runner_rollout_fast(QsEnv *e, int64_t T, const void *packed_weights, const float *logstd, int squash, const float *noise,
                           const uint8_t *dones_in, float *mb_obs, float *mb_actions, float *mb_values, float *mb_neglogp,
                           uint8_t *mb_dones, float *mb_rewards, uint8_t *mb_flags, float *last_obs, float *last_values,
                           uint8_t *last_dones)
{
    CHECK_ENV(e);
    if (!packed_weights || !logstd) return fail(QS_ERR_INVALID, "qs_runner_rollout_fast: bad arguments");
    if (((uintptr_t)packed_weights & 15) != 0) return fail(QS_ERR_INVALID, "qs_runner_rollout_fast: packed weights must be 16-byte aligned");
    return runner_launch(e, "qs_runner_rollout_fast", T, logstd, squash, nullptr, packed_weights, noise, dones_in, mb_obs,
                         mb_actions, mb_values, mb_neglogp, mb_dones, mb_rewards, mb_flags, last_obs, last_values, last_dones);
}

int qs_runner_rollout_fast_blob_bytes(void) { return kAcFastBlobBytes; }

// Diagnostic (not in quadsim.h; tests and A/B tools only): Runner kernel flavour for every later qs_runner_rollout* call of
// the process -- 1 one wave per tile, 0 role-split (matrix waves + env waves).  Returns the previous setting.
int qs_debug_set_runner_serial(int on) { return runner_serial_flag().exchange(on ? 1 : 0); }

int qs_set_queue_mode(QsEnv *e, int32_t mode)
{
    CHECK_ENV(e);                                     // drains a queue that is being switched off
    if (mode < QS_QUEUE_HIP_STREAM || mode > 4) return fail(QS_ERR_INVALID, "qs_set_queue_mode: mode must be 0 (HIP stream) or 1..4 private queues");
    if (e->chain && e->chain->requested == mode) return QS_OK;
    chain_close(e);
    if (mode == QS_QUEUE_HIP_STREAM) return QS_OK;
    HIP_TRY(hipStreamSynchronize(e->stream));
    int rc = chain_open(e, mode);
    if (rc == QS_OK) e->chain->requested = mode;
    return rc;
}

// Diagnostic (not in quadsim.h; tests only): pretend every tile is held by an XCD that does not exist, so that the placement
// check of the next private-queue step fires in every workgroup
int qs_debug_chain_poison_owner(QsEnv *e)
{
    if (!e || !e->chain) return fail(QS_ERR_INVALID, "qs_debug_chain_poison_owner: not in private-queue mode");
    DeviceGuard guard(e->cfg.device);
    int rc = chain_drain(e);
    if (rc) return rc;
    HIP_TRY(hipMemset(e->chain->d_owner, 9, (size_t)e->tiles * sizeof(unsigned)));
    e->chain->hip_dirty = false;          // keep the poisoned owners: the next step must not reset them
    return QS_OK;
}

// Diagnostic (not in quadsim.h; tests and A/B tools only): force the reset-preparation variant of the role-split step kernel for
// every later launch of the process: 0 two waves, 2 three waves, -1 the default choice by tiles per launch.  Returns the previous
// setting.  (Same results bit for bit: tests/test_gpu_groups_and_rollout.py::test_reset_preparation_wave_is_bit_identical.)
int qs_debug_set_reset_prep(int mode) { return prep_forced().exchange((mode == 0 || mode == 2) ? mode : -1); }

// Diagnostic (not in quadsim.h; tests only): the NEXT private-queue step runs with workgroup b stepping tile b + shift of its
// launch -- every tile on another XCD than the one that holds its state -- without any synchronisation in between: the
// placement guard must see the owner words the previous step wrote from the other XCDs.
int qs_debug_chain_shift_once(QsEnv *e, int32_t shift)
{
    if (!e || !e->chain) return fail(QS_ERR_INVALID, "qs_debug_chain_shift_once: not in private-queue mode");
    e->chain->dbg_shift = shift;
    return QS_OK;
}

int qs_get_queue_ordering(QsEnv *e, int32_t *ordering)
{
    if (!e || !ordering) return fail(QS_ERR_INVALID, "qs_get_queue_ordering: null argument");
    *ordering = (e->chain && e->chain->stream_ordered) ? QS_ORDER_STREAM : QS_ORDER_HOST;
    return QS_OK;
}

int qs_set_queue_ordering(QsEnv *e, int32_t ordering)
{
    CHECK_ENV(e);                                     // drains the queues
    if (!e->chain) return fail(QS_ERR_INVALID, "qs_set_queue_ordering: qs_set_queue_mode first");
    if (ordering != QS_ORDER_HOST && ordering != QS_ORDER_STREAM) return fail(QS_ERR_INVALID, "qs_set_queue_ordering: unknown ordering %d", ordering);
    QsChain *c = e->chain;
    if (c->stream_ordered && c->fwd_seq) HIP_TRY(hipStreamSynchronize(e->stream));   // pending hand-shake waits
    if (ordering == QS_ORDER_HOST) { c->stream_ordered = false; return QS_OK; }
    if (!chain_can_stream_order(e))
        return fail(QS_ERR_INVALID, "qs_set_queue_ordering: this device / HIP runtime has no stream memory operations (hipStreamWaitValue64)");
    return chain_enable_stream_order(e);
}

int qs_get_queue_mode(QsEnv *e, int32_t *mode)
{
    if (!e || !mode) return fail(QS_ERR_INVALID, "qs_get_queue_mode: null argument");
    *mode = e->chain ? (int32_t)e->chain->lanes.size() : QS_QUEUE_HIP_STREAM;
    return QS_OK;
}

int qs_set_rollout_layout(QsEnv *e, int32_t layout)
{
    CHECK_ENV(e);
    if (layout != QS_LAYOUT_TIME_MAJOR && layout != QS_LAYOUT_ENV_MAJOR) return fail(QS_ERR_INVALID, "qs_set_rollout_layout: unknown layout %d", layout);
    e->runner_env_major = layout == QS_LAYOUT_ENV_MAJOR;
    return QS_OK;
}

int qs_expert_action(QsEnv *e, float *state_des, float kp, float kd, float *actions)
{
    CHECK_ENV(e);
    if (!state_des || !actions) return fail(QS_ERR_INVALID, "qs_expert_action: null argument");
    if (e->cfg.io_space != QS_IO_DEVICE) return fail(QS_ERR_INVALID, "qs_expert_action: device buffers only");
    if (e->cfg.kind == QS_KIND_HOVERING_V0) return fail(QS_ERR_INVALID, "qs_expert_action: docking envs only");
    Par pn{e->cfg.mass, e->cfg.inertia[0], e->cfg.inertia[1], e->cfg.inertia[2]};
    if (e->per_env_params)
        hipLaunchKernelGGL(k_expert_action<true>, dim3(grid_tiles(e->n)), dim3(kBlock), 0, e->stream, e->st, e->par, e->n, state_des, kp, kd, pn, actions);
    else
        hipLaunchKernelGGL(k_expert_action<false>, dim3(grid_tiles(e->n)), dim3(kBlock), 0, e->stream, e->st, e->par, e->n, state_des, kp, kd, pn, actions);
    HIP_TRY(hipGetLastError());
    return QS_OK;
}

// ---- layer 1 ---------------------------------------------------------------------------------
int qs_drone_step(QsEnv *e, int64_t n, float *state, float *u_prev, const float *u, const float *par, uint8_t *limited)
{
    CHECK_ENV(e);
    if (n < 1 || !state || !u_prev || !u) return fail(QS_ERR_INVALID, "qs_drone_step: bad arguments");
    Par pn{e->cfg.mass, e->cfg.inertia[0], e->cfg.inertia[1], e->cfg.inertia[2]};
    float *ds = state, *dup = u_prev;
    const float *du = u, *dp = par;
    uint8_t *dl = limited;
    if (e->cfg.io_space == QS_IO_HOST) {
        int r = ensure_stage(e, (size_t)n * (13 + 4 + 4 + 4) * 4 + n + 4096);
        if (r) return r;
        Stage S{(char *)e->stage};
        float *a = S.take<float>(n * 13), *b = S.take<float>(n * 4), *c = S.take<float>(n * 4), *d = S.take<float>(n * 4);
        uint8_t *l = S.take<uint8_t>(n);
        HIP_TRY(hipMemcpyAsync(a, state, n * 52, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(b, u_prev, n * 16, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(c, u, n * 16, hipMemcpyHostToDevice, e->stream));
        if (par) HIP_TRY(hipMemcpyAsync(d, par, n * 16, hipMemcpyHostToDevice, e->stream));
        ds = a; dup = b; du = c; dp = par ? d : nullptr; dl = limited ? l : nullptr;
    }
    hipLaunchKernelGGL(k_drone_step, dim3(grid_flat(n)), dim3(kBlock), 0, e->stream, n, ds, dup, du, dp, dl, pn, e->cfg.dt,
                       e->cfg.integrator);
    HIP_TRY(hipGetLastError());
    if (e->cfg.io_space == QS_IO_HOST) {
        HIP_TRY(hipMemcpyAsync(state, ds, n * 52, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(u_prev, dup, n * 16, hipMemcpyDeviceToHost, e->stream));
        if (limited) HIP_TRY(hipMemcpyAsync(limited, dl, n, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    return QS_OK;
}

int qs_ctrl(QsEnv *e, int64_t n, int32_t mode, float *state_des, const float *state, const float *state_last, float mass,
            float *u_out)
{
    CHECK_ENV(e);
    if (n < 1 || !state_des || !state || !u_out || (mode != 0 && mode != 1)) return fail(QS_ERR_INVALID, "qs_ctrl: bad arguments");
    if (mode == 1 && !state_last) return fail(QS_ERR_INVALID, "qs_ctrl: vel_controller needs state_last");
    float *dsd = state_des, *duo = u_out;
    const float *dsn = state, *dsl = state_last;
    if (e->cfg.io_space == QS_IO_HOST) {
        int r = ensure_stage(e, (size_t)n * (13 * 3 + 4) * 4 + 4096);
        if (r) return r;
        Stage S{(char *)e->stage};
        float *a = S.take<float>(n * 13), *b = S.take<float>(n * 13), *c = S.take<float>(n * 13), *d = S.take<float>(n * 4);
        HIP_TRY(hipMemcpyAsync(a, state_des, n * 52, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(b, state, n * 52, hipMemcpyHostToDevice, e->stream));
        if (state_last) HIP_TRY(hipMemcpyAsync(c, state_last, n * 52, hipMemcpyHostToDevice, e->stream));
        dsd = a; dsn = b; dsl = state_last ? c : nullptr; duo = d;
    }
    hipLaunchKernelGGL(k_ctrl, dim3(grid_flat(n)), dim3(kBlock), 0, e->stream, n, (int)mode, dsd, dsn, dsl, mass, duo);
    HIP_TRY(hipGetLastError());
    if (e->cfg.io_space == QS_IO_HOST) {
        HIP_TRY(hipMemcpyAsync(state_des, dsd, n * 52, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(u_out, duo, n * 16, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    return QS_OK;
}

int qs_transform(QsEnv *e, int32_t op, int64_t n, const float *in, float *out)
{
    CHECK_ENV(e);
    if (n < 1 || !in || !out || op < 0 || op > 3) return fail(QS_ERR_INVALID, "qs_transform: bad arguments");
    const int wi[4] = {4, 3, 4, 9}, wo[4] = {3, 4, 9, 3};
    const float *di = in;
    float *dout = out;
    if (e->cfg.io_space == QS_IO_HOST) {
        int r = ensure_stage(e, (size_t)n * (wi[op] + wo[op]) * 4 + 1024);
        if (r) return r;
        Stage S{(char *)e->stage};
        float *a = S.take<float>(n * wi[op]), *b = S.take<float>(n * wo[op]);
        HIP_TRY(hipMemcpyAsync(a, in, n * wi[op] * 4, hipMemcpyHostToDevice, e->stream));
        di = a; dout = b;
    }
    hipLaunchKernelGGL(k_transform, dim3(grid_flat(n)), dim3(kBlock), 0, e->stream, (int)op, n, di, dout);
    HIP_TRY(hipGetLastError());
    if (e->cfg.io_space == QS_IO_HOST) {
        HIP_TRY(hipMemcpyAsync(out, dout, n * wo[op] * 4, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    return QS_OK;
}

int qs_rel_obs(QsEnv *e, int64_t n, const float *chaser, const float *target, float *obs)
{
    CHECK_ENV(e);
    if (n < 1 || !chaser || !target || !obs) return fail(QS_ERR_INVALID, "qs_rel_obs: bad arguments");
    const float *dc = chaser, *dt = target;
    float *dob = obs;
    if (e->cfg.io_space == QS_IO_HOST) {
        int r = ensure_stage(e, (size_t)n * (13 * 2 + 12) * 4 + 4096);
        if (r) return r;
        Stage S{(char *)e->stage};
        float *a = S.take<float>(n * 13), *b = S.take<float>(n * 13), *c = S.take<float>(n * 12);
        HIP_TRY(hipMemcpyAsync(a, chaser, n * 52, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(b, target, n * 52, hipMemcpyHostToDevice, e->stream));
        dc = a; dt = b; dob = c;
    }
    hipLaunchKernelGGL(k_rel_obs, dim3(grid_flat(n)), dim3(kBlock), 0, e->stream, n, dc, dt, dob);
    HIP_TRY(hipGetLastError());
    if (e->cfg.io_space == QS_IO_HOST) {
        HIP_TRY(hipMemcpyAsync(obs, dob, n * 48, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    return QS_OK;
}

}  // extern "C"
