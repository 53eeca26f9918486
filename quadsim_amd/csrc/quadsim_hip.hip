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
//
// ONE translation unit (the private-queue code looks the step kernels up by their mangled names in the code object of this
// very library), kept in pieces that are included below at fixed positions:
//     quadsim_device.hpp   the per-env device functions (drone step, controller, state2rel, reward, rocRAND draws)
//     step_kernels.hpp     StepArgs, the step / roll-out / policy / Runner kernels and their launch helpers
//     rollout_ops.hpp      GAE, flatten, episode statistics            policy_rollout.hpp   the MLP on the matrix cores
//     env_groups.hpp       env groups (qs_set_groups)                  private_queue.hpp    private AQL queues (qs_set_queue_mode)
// and here: the handle (QsEnv), its launch / reset / bounce-buffer helpers, and the C ABI.
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

#include "step_kernels.hpp"

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

#include "env_groups.hpp"

#include "private_queue.hpp"

namespace {

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
    try { e->groups.reserve((size_t)groups); }           // no C++ exception may cross the C ABI (push_back below cannot throw now)
    catch (...) { return fail(QS_ERR_NOMEM, "qs_set_groups: out of host memory"); }
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
    if (e->chain) {
        try { steps.reserve((size_t)T); }                // no C++ exception may cross the C ABI
        catch (...) { return fail(QS_ERR_NOMEM, "qs_rollout_stepwise: out of host memory for %lld kernel-argument blocks", (long long)T); }
    }
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
    if ((((uintptr_t)wt2) | ((uintptr_t)wt3)) & 15u) return fail(QS_ERR_INVALID, "qs_policy_rollout: wt2 and wt3 must be 16-byte aligned");
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

int qs_policy_forward(QsEnv *e, int64_t n, const float *wt1, const float *b1, const float *wt2, const float *b2, const float *wt3,
                      const float *b3, const float *obs, float *actions)
{
    Range rg_("qs_policy_forward");
    CHECK_ENV(e);
    if (n < 1 || !wt1 || !b1 || !wt2 || !b2 || !wt3 || !b3 || !obs || !actions) return fail(QS_ERR_INVALID, "qs_policy_forward: bad arguments");
    if ((((uintptr_t)wt2) | ((uintptr_t)wt3) | ((uintptr_t)obs) | ((uintptr_t)actions)) & 15u)
        return fail(QS_ERR_INVALID, "qs_policy_forward: wt2, wt3, obs and actions must be 16-byte aligned");
    if (e->cfg.io_space != QS_IO_DEVICE) return fail(QS_ERR_INVALID, "qs_policy_forward: device buffers only");
    MlpArgs M{wt1, b1, wt2, b2, wt3, b3};
    hipLaunchKernelGGL(k_policy_forward, dim3(grid_tiles(n)), dim3(kBlock), 0, e->stream, M, obs, actions, n);
    HIP_TRY(hipGetLastError());
    return QS_OK;
}

int qs_policy_forward_fast(QsEnv *e, int64_t n, const void *packed_weights, const float *obs, float *actions)
{
    Range rg_("qs_policy_forward_fast");
    CHECK_ENV(e);
    if (n < 1 || !packed_weights || !obs || !actions) return fail(QS_ERR_INVALID, "qs_policy_forward_fast: bad arguments");
    if ((((uintptr_t)packed_weights) | ((uintptr_t)obs) | ((uintptr_t)actions)) & 15u)
        return fail(QS_ERR_INVALID, "qs_policy_forward_fast: packed_weights, obs and actions must be 16-byte aligned");
    if (e->cfg.io_space != QS_IO_DEVICE) return fail(QS_ERR_INVALID, "qs_policy_forward_fast: device buffers only");
    hipLaunchKernelGGL(k_policy_forward_fast, dim3(grid_tiles(n)), dim3(kBlock), 0, e->stream, (const uint4 *)packed_weights, obs, actions, n);
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
