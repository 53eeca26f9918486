// step_kernels.hpp -- StepArgs, the step / roll-out / policy / Runner kernels and their launch helpers
// A fragment of quadsim_hip.hip (ONE translation unit: the kernels' mangled names, which the private-queue code resolves
// in the code object, live in that unit's anonymous namespace); included there at a fixed position, nowhere else.
#pragma once

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

// LDS image of the f32 actor (policy_lds_floats()): W2^T | W3^T (16-row tile) | W1^T | b1 | b2 | b3 (16) | per-wave obs / action staging
struct MlpLds {
    float *W2, *W3, *W1, *B1, *B2, *B3, *ObsAll, *ActAll;
};
__device__ __forceinline__ MlpLds mlp_lds_layout(float *lds)
{
    MlpLds L;
    L.W2 = lds;
    L.W3 = L.W2 + kHid * kLdW;
    L.W1 = L.W3 + 16 * kLdW;
    L.B1 = L.W1 + kHid * kLdW1;
    L.B2 = L.B1 + kHid;
    L.B3 = L.B2 + kHid;
    L.ObsAll = L.B3 + 16;
    L.ActAll = L.ObsAll + 4 * (12 * 64);
    return L;
}
// weights -> LDS (W3^T rows 4..15 and b3[4..15] are zero padding of the 16-row MFMA tile), by the 256 threads of a workgroup.
// Every request of a thread goes out before the first LDS write waits for one: a copy loop of load / wait / write pairs costs a
// launch with T = 1 (one step per launch, VecDockingEnv.step_policy) one L2 round trip per iteration
__device__ __forceinline__ void mlp_stage_weights(const MlpArgs &M, const MlpLds &L)
{
    static_assert(kHid == 128 && kBlock == 256 && kLdW % 4 == 0, "staging layout");
    const float4 *w2v = reinterpret_cast<const float4 *>(M.wt2);
    float4 v2[16], v3[2];
    float v1[6];
#pragma unroll
    for (int j = 0; j < 16; ++j) v2[j] = w2v[j * kBlock + threadIdx.x];                       // 4 096 float4: row i4 >> 5
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i4 = j * kBlock + threadIdx.x;                                              // 512 float4 of the padded tile
        v3[j] = (i4 >> 5) < 4 ? reinterpret_cast<const float4 *>(M.wt3)[i4] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) v1[j] = M.wt1[j * kBlock + threadIdx.x];                      // 1 536 floats
    const float vb1 = threadIdx.x < kHid ? M.b1[threadIdx.x] : 0.0f, vb2 = threadIdx.x < kHid ? M.b2[threadIdx.x] : 0.0f;
    const float vb3 = threadIdx.x < 4 ? M.b3[threadIdx.x] : 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int i4 = j * kBlock + threadIdx.x;
        *reinterpret_cast<float4 *>(L.W2 + (i4 >> 5) * kLdW + (i4 & 31) * 4) = v2[j];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i4 = j * kBlock + threadIdx.x;
        *reinterpret_cast<float4 *>(L.W3 + (i4 >> 5) * kLdW + (i4 & 31) * 4) = v3[j];
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int i = j * kBlock + threadIdx.x;
        L.W1[(i / 12) * kLdW1 + (i % 12)] = v1[j];
    }
    if (threadIdx.x < kHid) { L.B1[threadIdx.x] = vb1; L.B2[threadIdx.x] = vb2; }
    if (threadIdx.x < 16) L.B3[threadIdx.x] = vb3;
}
// the split-bf16 weight image (host-packed, kFastBlobBytes) verbatim into LDS: all requests first, then the LDS writes
__device__ __forceinline__ void mlp_stage_blob(const uint4 *__restrict__ blob, char *lds)
{
    constexpr int kN16 = kFastBlobBytes / 16, kPer = (kN16 + kBlock - 1) / kBlock;
    uint4 v[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int i = j * kBlock + threadIdx.x;
        v[j] = i < kN16 ? blob[i] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int i = j * kBlock + threadIdx.x;
        if (i < kN16) reinterpret_cast<uint4 *>(lds)[i] = v[j];
    }
}

// The actor alone: actions [n,4] = clip(MLP(obs [n,12])) on the matrix cores -- model.predict(obs, deterministic=True) of
// run_trained_docking_ppo2.py:41 for n rows, for loops that need the env step as a call of its own (terminal observations,
// infos).  One wave per 64 rows; the same mlp_actor as the fused kernels, hence the same bits for the same observations.
__global__ __launch_bounds__(kBlock, 1) void k_policy_forward(MlpArgs M, const float *__restrict__ obs, float *__restrict__ actions, int64_t n)
{
    __shared__ __attribute__((aligned(16))) float lds[policy_lds_floats()];
    const MlpLds L = mlp_lds_layout(lds);
    mlp_stage_weights(M, L);
    __syncthreads();
    const int lane = threadIdx.x & (kTile - 1), w = threadIdx.x >> 6;
    const int64_t row = ((int64_t)blockIdx.x * (kBlock / kTile) + w) * kTile + lane;
    float o[12], a[4];
    if (row < n) {
        const float4 *p = reinterpret_cast<const float4 *>(obs + row * 12);
        const float4 v0 = p[0], v1 = p[1], v2 = p[2];
        o[0] = v0.x; o[1] = v0.y; o[2] = v0.z; o[3] = v0.w; o[4] = v1.x; o[5] = v1.y; o[6] = v1.z; o[7] = v1.w;
        o[8] = v2.x; o[9] = v2.y; o[10] = v2.z; o[11] = v2.w;
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) o[i] = 0.0f;          // MFMA needs the whole wave
    }
    mlp_actor(o, a, L.W1, L.B1, L.W2, L.B2, L.W3, L.B3, L.ObsAll + w * (12 * 64), L.ActAll + w * (64 * 4), lane);
    if (row < n) reinterpret_cast<float4 *>(actions)[row] = make_float4(a[0], a[1], a[2], a[3]);
}

__global__ __launch_bounds__(kBlock, 1) void k_policy_forward_fast(const uint4 *__restrict__ blob, const float *__restrict__ obs,
                                                                   float *__restrict__ actions, int64_t n)
{
    __shared__ __attribute__((aligned(16))) char lds[kFastBlobBytes + 4 * (12 * 64 + 64 * 4) * 4];
    mlp_stage_blob(blob, lds);
    __syncthreads();
    const int lane = threadIdx.x & (kTile - 1), w = threadIdx.x >> 6;
    const int64_t row = ((int64_t)blockIdx.x * (kBlock / kTile) + w) * kTile + lane;
    float *stage = reinterpret_cast<float *>(lds + kFastBlobBytes);
    float o[12], a[4];
    if (row < n) {
        const float4 *p = reinterpret_cast<const float4 *>(obs + row * 12);
        const float4 v0 = p[0], v1 = p[1], v2 = p[2];
        o[0] = v0.x; o[1] = v0.y; o[2] = v0.z; o[3] = v0.w; o[4] = v1.x; o[5] = v1.y; o[6] = v1.z; o[7] = v1.w;
        o[8] = v2.x; o[9] = v2.y; o[10] = v2.z; o[11] = v2.w;
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) o[i] = 0.0f;
    }
    mlp_actor_fast(o, a, lds, stage + w * (12 * 64), stage + 4 * (12 * 64) + w * (64 * 4), lane);
    if (row < n) reinterpret_cast<float4 *>(actions)[row] = make_float4(a[0], a[1], a[2], a[3]);
}

// Policy-in-the-loop roll-out: T steps of  a = clip(MLP(obs));  obs, r, done = env.step(a)  in one launch
// (run_trained_docking_ppo2.py:37-60 for N envs).  MLP on exact-f32 MFMA (policy_rollout.hpp), env step = the
// device code of k_env.  obs_0 is derived from the stored state (an observation is always state2rel of the state).
template <int INTEG, int RMODE>
__global__ __launch_bounds__(kBlock, 1) void k_policy_rollout(StepArgs A, MlpArgs M, float *__restrict__ actions_out)
{
    __shared__ __attribute__((aligned(16))) float lds[policy_lds_floats()];
    const MlpLds L = mlp_lds_layout(lds);
    float *const sW1 = L.W1, *const sB1 = L.B1, *const sW2 = L.W2, *const sB2 = L.B2, *const sW3 = L.W3, *const sB3 = L.B3;
    float *const sObsAll = L.ObsAll, *const sActAll = L.ActAll;
    mlp_stage_weights(M, L);
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
    mlp_stage_blob(blob, lds);
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
