/*
 * quadsim.h -- C ABI of libquadsim_hip.so: the MI355X (gfx950) drop-in for
 * QuadSim's env.step() hot path (docking-v0 / docking-v2).
 *
 * The reference has no FFI of its own -- the path is plain Python classes
 * (SURVEY.md section 8b).  Each entry point below therefore cites the Python
 * method it replaces; INTEGRATION.md shows the ctypes binding a maintainer adds
 * to gym_docking to switch over.  Plain pointers and sizes only; no torch, no
 * HIP types in the signatures (a hipStream_t travels as void*).
 *
 * Conventions
 *   - every function returns 0 (QS_OK) or a negative QS_ERR_*; the message is
 *     available from qs_last_error() (thread-local).
 *   - all data buffers are caller-owned.  With io_space == QS_IO_DEVICE they
 *     are device pointers on cfg.device (e.g. torch.Tensor.data_ptr()); with
 *     QS_IO_HOST they are host pointers and the call stages them through the
 *     GPU and returns after the result is back (single-env gym shim).
 *   - device calls are asynchronous on the handle's stream; qs_sync() waits.
 *   - a handle is not thread-safe (call it from one thread at a time; the launcher
 *     threads of qs_set_groups are internal); distinct handles are independent
 *     (one per GPU / per process).
 *   - there is no CPU fallback: creation fails if no HIP device is present.
 *
 * Layouts (row-major, float32 unless noted)
 *   state vector [13] = pos(3) vel(3) quat w,x,y,z(4) body-rates(3)   (dynamics/quadrotor.py:25)
 *   actions [N,4] in [-1,1]                                             (docking_env.py:93)
 *   obs [N,12] = rel_pos rel_vel rel_euler rel_euler_rates              (docking_env.py:287-293)
 *   (hovering-v0: obs [N,13] = the drone's state, hovering_env.py:78; read `12` as qs_obs_dim below)
 *   done [N] uint8; flags [N] uint8 (QS_FLAG_*)
 */
#ifndef QUADSIM_H
#define QUADSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QS_VERSION 131 /* 0.1.3.1: + qs_policy_forward; 0.1.3: + stream-ordered private queues (qs_set_queue_ordering); 0.1.2: + private-queue mode; 0.1.1: qs_step_ex, env groups, qs_gae_flatten, qs_episode_stats, ... */

enum {
    QS_OK = 0,
    QS_ERR_INVALID = -1, /* bad argument / shape */
    QS_ERR_HIP = -2,     /* a HIP runtime call failed */
    QS_ERR_NO_DEVICE = -3,
    QS_ERR_NOMEM = -4
};

/* env kinds: gym ids registered at gym-docking/gym_docking/__init__.py:3-17 */
enum {
    QS_KIND_DOCKING_V0 = 0, /* DockingEnv,       envs/docking_env.py        */
    QS_KIND_DOCKING_V2 = 1, /* MovingDockingEnv, envs/moving_docking_env.py */
    QS_KIND_DOCKING_V1 = 2, /* ImitatingDockingEnv, envs/imitating_docking_env.py: v0 whose chaser start is jittered
                               once at construction (:34) and restored by every reset */
    QS_KIND_HOVERING_V0 = 3 /* HoveringEnv, envs/hovering_env.py: one drone, obs = raw state [13], action in [0,1]^4 */
};

enum {
    QS_INTEG_FROZEN = 0, /* what Drone.step computes: RK45 over a frozen RHS == s + dt*df(s,u_prev)
                            (dynamics/quadrotor.py:115-134).  The parity mode. */
    QS_INTEG_RK4 = 1     /* classic RK4 re-evaluating df at the stage states (the intended physics;
                            no counterpart in the reference) */
};

enum { QS_IO_DEVICE = 0, QS_IO_HOST = 1 };

/* bits of flags[] -- info['flag_docking'], info['done_overlimit'] (docking_env.py:226-229) */
enum {
    QS_FLAG_DOCKED = 1,
    QS_FLAG_OVERLIMIT = 2,
    QS_FLAG_OVERTIME = 4,       /* t >= 600, docking_env.py:152 */
    QS_FLAG_CHASER_LIMITED = 8, /* Drone.attitude_limit fired (quadrotor.py:135-138) */
    QS_FLAG_TARGET_LIMITED = 16
};

enum {
    QS_RANDOMISE_NONE = 0,  /* nominal reset states (docking_env.py:34-57) -- the reference's behaviour */
    QS_RANDOMISE_INIT = 1,  /* + rocRAND jitter of the chaser's initial state (the commented ranges at docking_env.py:34-37) */
    QS_RANDOMISE_PARAMS = 2 /* + per-episode mass / inertia scale (BASELINE config 5) */
};

typedef struct QsEnv QsEnv;

typedef struct QsConfig {
    int32_t struct_size; /* sizeof(QsConfig), set by qs_config_default */
    int32_t kind;        /* QS_KIND_* */
    int64_t num_envs;    /* N parallel envs on this device */
    int32_t device;      /* HIP device ordinal */
    int32_t integrator;  /* QS_INTEG_* */
    float dt;            /* 0.02 (dynamics/quadrotor.py:10) */
    int32_t auto_reset;  /* 1: SB2 VecEnv semantics (reset on done inside step, terminal obs kept) ;
                            0: gym.Env semantics (step never resets, docking_env.py:104-231) */
    int32_t randomise;   /* QS_RANDOMISE_* */
    int32_t io_space;    /* QS_IO_* */
    uint64_t seed;          /* Philox key */
    uint64_t env_id_offset; /* global id of env 0 of this handle (shard offset; RNG is keyed by global id) */
    float init_range[4];    /* half-ranges: chaser pos [m], vel [m/s], euler [rad], body rates [rad/s] */
    float mass_scale[2];    /* [lo,hi] multiplier of mass    (QS_RANDOMISE_PARAMS) */
    float inertia_scale[2]; /* [lo,hi] multiplier of Ixx,Iyy,Izz */
    float mass;             /* nominal 0.18 (quadrotor.py:16) */
    float inertia[3];       /* nominal diag (2.5e-4, 2.32e-4, 3.738e-4) (quadrotor.py:17-19) */
    void *stream;           /* hipStream_t to launch on when external_stream != 0 (NULL = the device's default stream) */
    int32_t external_stream; /* 0: the handle creates and owns a stream ; 1: launch on `stream` (e.g. torch's current stream) */
    int32_t reserved;
} QsConfig;

/* fills *cfg with the reference's constants: v0, N=1, frozen, dt 0.02, auto_reset 0, no randomisation */
int qs_config_default(QsConfig *cfg);

int qs_version(void);
const char *qs_last_error(void);

/* DockingEnv.__init__ / MovingDockingEnv.__init__ (docking_env.py:15-102) for N envs.
 * All envs start in the nominal state with target_state_des attitude (1,0,0,0). */
int qs_create(const QsConfig *cfg, QsEnv **out);
int qs_destroy(QsEnv *env);

/* DockingEnv.reset (docking_env.py:233-244) for the envs with mask[i] != 0 (mask NULL = all).
 * obs_out [N,12] nullable; rows of unmasked envs are left untouched.
 * Does not touch the target's desired attitude (the reference never resets it). */
int qs_reset(QsEnv *env, const uint8_t *mask, float *obs_out);

/* DockingEnv.step (docking_env.py:104-231) / MovingDockingEnv.step (moving_docking_env.py:111-192)
 * for all N envs in one fused kernel.  actions [N,4]; obs [N,12]; reward [N]; done [N];
 * flags [N] nullable; terminal_obs [N,12] nullable (rows written only where done && auto_reset:
 * SB2's infos[i]['terminal_observation']). */
int qs_step(QsEnv *env, const float *actions, float *obs, float *reward, uint8_t *done, uint8_t *flags,
            float *terminal_obs);

/* qs_step that also hands out the TERMINAL state of every env that finished: terminal_state [N,26] nullable = chaser
 * [13] | target [13] as they stood when the step returned done (rows written only where done && auto_reset).  This is
 * what info['chaser'] / info['target'] hold on a done step in the reference (docking_env.py:226-229 return
 * self.state_chaser / self.state_target of the terminal step; SB2's SubprocVecEnv worker resets only afterwards). */
int qs_step_ex(QsEnv *env, const float *actions, float *obs, float *reward, uint8_t *done, uint8_t *flags,
               float *terminal_obs, float *terminal_state);

/* ---- env groups: several step chains of ONE handle in flight at a time -------------------------------------------
 * The reference's trainers step 10 worker processes that run concurrently (run_docking_ppo2.py:65-67
 * SubprocVecEnv([...]*10): step_async sends to every worker, step_wait collects).  The device analogue: the N envs
 * of a handle are partitioned into G contiguous groups (whole 64-env tiles), each with its own stream, so that the
 * step of one group overlaps the kernel boundary / the policy of another (EnvPool-style send / recv).  Envs never
 * interact and RNG is keyed by global env id, so any grouping computes bit-identical results to qs_step.
 *   qs_set_groups(env, G, threads): G <= 1 removes the grouping.  threads != 0: every group gets a launcher thread --
 *     the calling thread only posts a launch record (a lone host thread issues ~0.35 launches / us, less than two
 *     groups consume).
 *   qs_group_range: [env_begin, env_end) of group g.   qs_group_stream: its hipStream_t (run that group's policy on
 *     it and no cross-stream ordering is needed at all: qs_step_group / qs_step_groups return only after the launch is
 *     ON that stream, launcher thread or not);  qs_group_set_stream: use a caller-owned stream instead.
 *   qs_step_group(env, g, ...): one step of group g on its stream; every pointer addresses the GROUP's rows
 *     (actions [n_g,4], obs [n_g,12], ... of envs env_begin..env_end-1).
 *   qs_step_groups(env, ...): one step of ALL groups, full-batch [N,...] pointers as qs_step_ex; one call, G launches.
 * Ordering: work of the handle itself is ordered automatically (a group step waits for earlier main-stream calls such as
 * qs_reset; every other entry point first waits for pending group steps).  Data the CALLER produces or consumes on
 * another stream is the caller's to order: qs_groups_fork makes the group streams wait for the main stream's work so
 * far, qs_groups_join makes the main stream wait for the group streams (qs_sync, the timers and every non-group call do
 * it implicitly). */
int qs_set_groups(QsEnv *env, int32_t groups, int32_t launcher_threads);
int qs_group_count(QsEnv *env, int32_t *groups);
int qs_group_range(QsEnv *env, int32_t g, int64_t *env_begin, int64_t *env_end);
int qs_group_stream(QsEnv *env, int32_t g, void **hip_stream);
int qs_group_set_stream(QsEnv *env, int32_t g, void *hip_stream);
int qs_step_group(QsEnv *env, int32_t g, const float *actions, float *obs, float *reward, uint8_t *done, uint8_t *flags,
                  float *terminal_obs, float *terminal_state);
int qs_step_groups(QsEnv *env, const float *actions, float *obs, float *reward, uint8_t *done, uint8_t *flags,
                   float *terminal_obs, float *terminal_state);
int qs_groups_fork(QsEnv *env);
int qs_groups_join(QsEnv *env);

/* ---- private-queue mode: step launches without the end-of-kernel cache write-back --------------------------------
 * Every kernel HIP launches ends with an agent-scope release (the chip's eight L2s are not coherent with each other,
 * so their dirty lines are written back before the next packet starts).  In a chain of step launches that is 1.6 of
 * 6.5 us per step at 65 536 envs, and not needed: tile b is stepped by workgroup b of every launch on the same XCD, so
 * its state can stay dirty in that XCD's L2 from one step to the next.  QS_QUEUE_PRIVATE gives the handle an AQL queue of
 * its own; qs_step / qs_step_ex / qs_rollout_stepwise then write one packet per step (ordered behind the previous one,
 * acquire at agent scope, NO release) -- results bit-identical to the default mode.  Contract:
 *   - the handle's own calls stay ordered: any other entry point first drains the queue with a system-scope release
 *     (host wait);
 *   - ordering against the CALLER's work, QS_ORDER_STREAM (default wherever the device has stream memory operations,
 *     hipDeviceAttributeCanUseStreamWaitValue): a step behaves like a launch on the handle's stream although it runs on
 *     the private queue -- the call enqueues a write-value on the stream (inputs produced on that stream before the call
 *     are complete when the step starts), the queue's packets wait for it, the last packet of the call releases at agent
 *     scope and signals, and the call ends by enqueuing a wait-value on the stream (work enqueued on that stream after
 *     the call sees the outputs).  No host synchronisation: `obs -> policy kernels -> qs_step` loops run exactly as in
 *     the default mode (rl_baselines/ppo2/ppo2.py:472-499) -- but no faster: a per-step loop pays the release every
 *     step, as a HIP launch does, plus the hand-shake; the mode pays off for qs_rollout_stepwise, which hand-shakes and
 *     releases ONCE for its T steps.  Not capturable into a hipGraph; switching the handle's stream (qs_set_stream)
 *     synchronises the old one.
 *   - QS_ORDER_HOST (round 2's contract, qs_set_queue_ordering; the fallback without stream memory operations): no
 *     hand-shake -- the buffers passed to a step must be complete when it is called, and its outputs may be read after
 *     qs_sync() (or any other entry point);
 *   - every workgroup checks that it runs on the XCD that holds its tile (the hardware deals blocks to XCDs round-robin
 *     from a fixed start; HIP does not promise it): the owning XCD of a tile is kept in a word that is only accessed by
 *     agent-scope atomics, so every XCD sees it; if the check ever fails the workgroup touches nothing, raises an error
 *     word in host memory, and both the next step call and the next synchronising call (qs_sync, qs_get_state, ...)
 *     return QS_ERR_HIP -- a loop of nothing but steps cannot run on unnoticed; the synchronising call re-arms the handle.
 * mode = the number of private queues, 1..4: with more than one, the tiles are split into that many contiguous ranges and
 * every step writes one packet per queue -- the chains then overlap each other's kernel boundary (65 536 envs: 5.2 us per
 * step with one queue, 4.6 us with two; without the release there is no chip-wide write-back for them to collide on).
 * qs_set_params / qs_set_init_state after qs_set_queue_mode are honoured (the step-kernel variant is re-resolved).
 * Docking envs, device buffers. */
enum { QS_QUEUE_HIP_STREAM = 0, QS_QUEUE_PRIVATE = 1 /* 2, 3, 4: that many private queues */ };
enum { QS_ORDER_HOST = 0, QS_ORDER_STREAM = 1 };
int qs_set_queue_mode(QsEnv *env, int32_t mode);
int qs_get_queue_mode(QsEnv *env, int32_t *mode);
int qs_set_queue_ordering(QsEnv *env, int32_t ordering); /* QS_ORDER_*; QS_ERR_INVALID outside private-queue mode */
int qs_get_queue_ordering(QsEnv *env, int32_t *ordering);

/* T consecutive steps in ONE launch, env state held in registers (the loop body of the
 * trainer's Runner, rl_baselines/ppo2/ppo2.py:472-499, with the policy's actions pre-staged).
 * Requires auto_reset.  actions [T,N,4], or NULL: U(-1,1) actions drawn in-kernel from the
 * action stream (synthetic roll-outs).  obs [T,N,12], reward [T,N], done [T,N], flags [T,N] nullable.
 * Results are identical to T calls of qs_step. */
int qs_rollout(QsEnv *env, int64_t T, const float *actions, float *obs, float *reward, uint8_t *done,
               uint8_t *flags);

/* qs_rollout writing ONE packed slab [T,N,14] float32 = obs 12, reward, done (0.0 / 1.0) per env-step: the 56-byte
 * unit BASELINE configs 4/5 all-gather over xGMI (one large collective per roll-out instead of three).  actions
 * [T,N,4] or NULL (in-kernel U(-1,1)); flags [T,N] nullable.  Same values as qs_rollout.  Docking envs. */
int qs_rollout_slab(QsEnv *env, int64_t T, const float *actions, float *slab, uint8_t *flags);

/* Same contract and bit-identical results as qs_rollout, but issued as T single-step launches from
 * native code (what a per-step trainer loop costs the GPU, without the interpreter between launches).
 * actions [T,N,4] required. */
int qs_rollout_stepwise(QsEnv *env, int64_t T, const float *actions, float *obs, float *reward, uint8_t *done,
                        uint8_t *flags);

/* U(-1,1) synthetic actions [T,N,4] for steps step0 .. step0+T-1 (rocRAND Philox4x32-10 action stream;
 * identical to what qs_rollout draws in-kernel when actions == NULL). */
int qs_fill_random_actions(QsEnv *env, int64_t T, uint64_t step0, float *actions);

/* Internal state in the reference's own terms (any pointer may be NULL = skip):
 * chaser [N,13] env.state_chaser; target [N,13] env.state_target; u_prev [N,8] = chaser.u, target.u
 * (last LIMITED controls, quadrotor.py:140); qdes [N,4] env.target_state_des[6:10];
 * last_shaping [N]; t [N] (env.t as float). */
int qs_get_state(QsEnv *env, float *chaser, float *target, float *u_prev, float *qdes, float *last_shaping,
                 float *t);
int qs_set_state(QsEnv *env, const float *chaser, const float *target, const float *u_prev, const float *qdes,
                 const float *last_shaping, const float *t);

/* Per-env initial states that reset() (and the auto-reset) return to: env.chaser_ini_state / env.target_ini_state
 * (docking_env.py:39-57; scripts mutate them, run_expert_policy.py:44,63-64) or HoveringEnv.ini_state
 * (hovering_env.py:26-29).  chaser_init [N,13]; target_init [N,13] nullable = nominal (ignored for hovering).
 * docking-v1 and hovering-v0 handles are created with rocRAND-drawn per-env initial states (the reference draws
 * them from numpy's global RNG at construction); this call overrides them.  Switches the handle to stored-init
 * resets (takes precedence over `randomise`) and re-initialises nothing by itself: call qs_reset afterwards. */
int qs_set_init_state(QsEnv *env, const float *chaser_init, const float *target_init);
int qs_get_init_state(QsEnv *env, float *chaser_init, float *target_init);

/* width of an observation row: 12 for the docking envs, 13 for hovering-v0 */
int qs_obs_dim(QsEnv *env, int32_t *dim);

/* per-env mass [N] and diagonal inertia [N,3] (Drone.mass / Drone.Inertia; also sets F_max = 4 m g,
 * controller.mass and action_mean/std = m g / 2 consistently).  Switches the handle to per-env params. */
int qs_set_params(QsEnv *env, const float *mass, const float *inertia);
int qs_get_params(QsEnv *env, float *mass, float *inertia);

/* global step counter k (number of env steps executed so far); keys the RNG streams.  It lives in device memory
 * and is advanced by the step kernels themselves, so a launch captured in a hipGraph advances it on every replay;
 * reading it synchronises the handle's stream. */
int qs_get_step_counter(QsEnv *env, uint64_t *k);
int qs_set_step_counter(QsEnv *env, uint64_t k);

/* external != 0: launch on hip_stream from now on (NULL = default stream); external == 0: back to an owned stream */
int qs_set_stream(QsEnv *env, void *hip_stream, int32_t external);
int qs_sync(QsEnv *env);

/* HIP-event stopwatch on the handle's stream (bench.py: kernel time on the launching stream) */
int qs_timer_start(QsEnv *env);
int qs_timer_stop(QsEnv *env, float *elapsed_ms); /* synchronises on the stop event */

/* ---- roll-out post-processing: the steps either side of env.step in the trainer's Runner --------- */

/* GAE(lambda) reverse scan, rl_baselines/ppo2/ppo2.py:507-520.  rewards, values [T,n] float32; dones [T,n] u8 =
 * mb_dones (the done flag BEFORE step t, :474); last_values [n] = model.value(last obs); last_dones [n] = self.dones
 * after the last step.  Out: advs [T,n] (mb_advs), returns [T,n] (mb_returns = advs + values).  Device buffers. */
int qs_gae(QsEnv *env, int64_t T, int64_t n, const float *rewards, const float *values, const uint8_t *dones,
           const float *last_values, const uint8_t *last_dones, float gamma, float lam, float *advs, float *returns);

/* swap_and_flatten, rl_baselines/ppo2/ppo2.py:531-539: in [T,n,d] -> out [n*T,d] (env-major).  d in {1,4,12,13}. */
int qs_swap_and_flatten(QsEnv *env, int64_t T, int64_t n, int64_t d, const float *in, float *out);

/* swap_and_flatten of a uint8 [T,n] array (mb_dones) without the float round trip */
int qs_swap_and_flatten_u8(QsEnv *env, int64_t T, int64_t n, const uint8_t *in, uint8_t *out);

/* GAE and the env-major flatten of every per-(t, env) scalar of Runner.run in ONE pass over the roll-out
 * (rl_baselines/ppo2/ppo2.py:507-523: mb_returns = GAE; then swap_and_flatten of mb_returns, mb_dones, mb_values,
 * mb_neglogpacs, true_reward).  Inputs as qs_gae plus neglogp [T,n] (nullable together with flat_neglogp).  Outputs,
 * env-major [n*T]: flat_returns, flat_values, flat_neglogp, flat_rewards (float32) and flat_masks (uint8 0/1 = mb_dones);
 * advs / returns [T,n] time-major are optional (both or neither) and equal qs_gae's.  Device buffers. */
int qs_gae_flatten(QsEnv *env, int64_t T, int64_t n, const float *rewards, const float *values, const float *neglogp,
                   const uint8_t *dones, const float *last_values, const uint8_t *last_dones, float gamma, float lam,
                   float *flat_returns, float *flat_values, float *flat_neglogp, float *flat_rewards, uint8_t *flat_masks,
                   float *advs, float *returns);

/* Episode accounting of one roll-out: return and length of every episode that ENDS inside it -- what the Monitor
 * wrapper (run_docking_ppo2.py:19-35) reports as info['episode'] = {'r', 'l'} and Runner._run collects into ep_infos
 * (ppo2.py:486-489).  rewards [T,n]; dones [T,n] = flags BEFORE each step (mb_dones); last_dones [n] = flags after the
 * last step.  ep_ret [n] float32 / ep_len [n] int32 carry the unfinished episode of every env from one roll-out to the
 * next (in/out; zero them once).  Out: *count (device uint64) = episodes found; the first min(count, cap) of them as
 * (out_key = t*n + env, out_ret, out_len) in unspecified order -- sort by key for the reference's (step, env) order.
 * cap = T*n can never overflow.  Device buffers. */
int qs_episode_stats(QsEnv *env, int64_t T, int64_t n, const float *rewards, const uint8_t *dones,
                     const uint8_t *last_dones, float *ep_ret, int32_t *ep_len, uint64_t *count, int64_t cap,
                     int64_t *out_key, float *out_ret, int32_t *out_len);

/* Policy-in-the-loop roll-out in one launch: for t < T:  a_t = clip(MLP(obs_t), -1, 1);  obs_{t+1}, r_t, done_t =
 * env.step(a_t) -- the loop of run_trained_docking_ppo2.py:37-60 for N envs, with the deterministic actor of the
 * shipped PPO2 MlpPolicy (shared_fc0 12->128, pi_fc0 128->128, pi 128->4, ReLU).  Weights are passed TRANSPOSED
 * (out, in), row-major float32 device arrays: wt1 [128,12], b1 [128], wt2 [128,128], b2 [128], wt3 [4,128], b3 [4]
 * (wt2 and wt3 16-byte aligned: they are read as float4).
 * obs [T,N,12] = obs_{t+1}; reward, done, flags [T,N]; actions [T,N,4] nullable = a_t.  The MLP runs on the matrix
 * cores in exact float32 (v_mfma_f32_16x16x4_f32).  docking-v0/v2, auto_reset, randomise 0/1. */
int qs_policy_rollout(QsEnv *env, int64_t T, const float *wt1, const float *b1, const float *wt2, const float *b2,
                      const float *wt3, const float *b3, float *obs, float *reward, uint8_t *done, uint8_t *flags,
                      float *actions);

/* The actor alone: actions [n,4] = clip(MLP(obs [n,12]), -1, 1) -- `model.predict(obs, deterministic=True)` of
 * run_trained_docking_ppo2.py:41 for n rows, on the matrix cores, launched on the handle's stream (the handle lends its
 * device and stream; n need not be its env count).  The same evaluation as inside qs_policy_rollout, hence the same
 * bits for the same observations: `qs_policy_forward; qs_step` == `qs_policy_rollout(T = 1)` where the latter's
 * observation is the stored one.  For loops that need the env step as a call of its own (terminal observations,
 * infos).  Weights as for qs_policy_rollout; obs and actions device arrays, 16-byte aligned.  _fast: split-bf16
 * operands, packed_weights as for qs_policy_rollout_fast. */
int qs_policy_forward(QsEnv *env, int64_t n, const float *wt1, const float *b1, const float *wt2, const float *b2,
                      const float *wt3, const float *b3, const float *obs, float *actions);
int qs_policy_forward_fast(QsEnv *env, int64_t n, const void *packed_weights, const float *obs, float *actions);

/* The same roll-out with the actor on the bf16 matrix rate (16x the f32 MFMA rate) and split operands: each f32
 * value is carried as bf16 hi + lo and x*w is evaluated as hi*hi + hi*lo + lo*hi with f32 accumulation -- about 1e-5
 * error on an action instead of float32's 1e-7; opt-in, not bit-compatible with the float32 policy.
 * packed_weights: device image of qs_policy_rollout_fast_blob_bytes() bytes, 16-byte aligned (layout:
 * quadsim_amd/csrc/policy_rollout.hpp "Fast actor"; quadsim_amd.policy.pack_fast_weights builds it). */
int qs_policy_rollout_fast(QsEnv *env, int64_t T, const void *packed_weights, float *obs, float *reward, uint8_t *done,
                           uint8_t *flags, float *actions);
int qs_policy_rollout_fast_blob_bytes(void);

/* PPO2 data collection in ONE launch: Runner._run's loop, rl_baselines/ppo2/ppo2.py:472-499, plus last_values (:506),
 * for the N envs of the handle and T = n_steps, with the actor-critic MlpPolicy the reference trains and ships
 * (trained_model/best_model_v0.zip; rl_baselines/common/policies.py:35-92,:583-588): shared_fc0 12->128, then
 * pi_fc0 128->128 -> pi 128->4 (action mean) and vf_fc0 128->128 -> vf 128->1 (value), ReLU, state-independent logstd.
 * Per step t (model.step, policies.py:592-603 without the fork's tanh unless `squash`):
 *   mb_obs[t] = obs_t;  mean, value = net(obs_t);  u = mean + exp(logstd) * eps_t   (distributions.py:426-430)
 *   mb_neglogp[t] = 0.5 sum(((u - mean) / std)^2) + 0.5 log(2 pi) 4 + sum(logstd)    (distributions.py:406-410)
 *   mb_actions[t] = u;  mb_values[t] = value;  mb_dones[t] = done flags BEFORE the step (ppo2.py:479)
 *   obs_{t+1}, mb_rewards[t], done = env.step(clip(u, -1, 1))                         (ppo2.py:480-484)
 * and after the loop last_obs = obs_T, last_values = value(obs_T), last_dones = the final done flags -- exactly the
 * inputs of qs_gae.  eps_t: `noise` [T,N,4] when given (what tf.random_normal would have drawn), else four standard
 * normals per env-step from rocRAND Philox4x32-10 (stream 4, subsequence = global env id, block = global step index)
 * through Box-Muller.  squash != 0 selects the fork's tanh variant (policies.py:238-242, distributions.py:412-415):
 * the env receives tanh(u), mb_neglogp adds sum(log(1 - tanh(u)^2 + 1e-6)), mb_actions keeps u.
 * Weights are TRANSPOSED (out, in) row-major float32 DEVICE arrays; logstd is read on the HOST.  The networks run on
 * the matrix cores in exact float32 (v_mfma_f32_16x16x4_f32).  docking-v0/v2, auto_reset, any randomise mode (per-env
 * mass / inertia and their per-episode redraw included: domain-randomised collection, BASELINE config 5), device I/O. */
typedef struct QsActorCritic {
    uint32_t struct_size;       /* sizeof(QsActorCritic) */
    int32_t squash;
    const float *wt1, *b1;      /* shared_fc0 [128,12], [128] */
    const float *wt2, *b2;      /* pi_fc0     [128,128], [128] */
    const float *wt3, *b3;      /* pi         [4,128], [4] */
    const float *wtv2, *bv2;    /* vf_fc0     [128,128], [128] */
    const float *wtv3, *bv3;    /* vf         [1,128], [1] */
    float logstd[4];            /* pi/logstd (host values) */
} QsActorCritic;
int qs_runner_rollout(QsEnv *env, int64_t T, const QsActorCritic *policy, const float *noise /* nullable [T,N,4] */,
                      const uint8_t *dones_in /* nullable [N] */, float *mb_obs /* [T,N,12] */,
                      float *mb_actions /* [T,N,4] */, float *mb_values /* [T,N] */, float *mb_neglogp /* [T,N] */,
                      uint8_t *mb_dones /* [T,N] */, float *mb_rewards /* [T,N] */, uint8_t *mb_flags /* nullable [T,N] */,
                      float *last_obs /* nullable [N,12] */, float *last_values /* [N] */, uint8_t *last_dones /* [N] */);

/* Layout of the two WIDE roll-out arrays of qs_runner_rollout(_fast), mb_obs and mb_actions:
 *   QS_LAYOUT_TIME_MAJOR (default): [T,N,12] / [T,N,4], as the Runner loop fills them (ppo2.py:473-478);
 *   QS_LAYOUT_ENV_MAJOR: [N,T,12] / [N,T,4] = what swap_and_flatten (ppo2.py:531-539) turns them into before the
 *   trainer sees them -- written that way directly, so Runner.run() needs no transpose pass over them.
 * The per-(t, env) scalars (values, neglogp, dones, rewards) stay [T,N]: qs_gae_flatten reads and flattens them. */
enum { QS_LAYOUT_TIME_MAJOR = 0, QS_LAYOUT_ENV_MAJOR = 1 };
int qs_set_rollout_layout(QsEnv *env, int32_t layout);

/* qs_runner_rollout with the networks on the bf16 matrix rate and split (hi + lo) operands, as qs_policy_rollout_fast:
 * about 1e-5 error on means and values instead of float32's 1e-7; opt-in.  packed_weights: device image of
 * qs_runner_rollout_fast_blob_bytes() bytes, 16-byte aligned (layout: quadsim_amd/csrc/policy_rollout.hpp "Fast
 * actor-critic heads"; quadsim_amd.runner.pack_fast_actor_critic builds it); logstd [4] on the HOST. */
int qs_runner_rollout_fast(QsEnv *env, int64_t T, const void *packed_weights, const float *logstd, int squash,
                           const float *noise, const uint8_t *dones_in, float *mb_obs, float *mb_actions, float *mb_values,
                           float *mb_neglogp, uint8_t *mb_dones, float *mb_rewards, uint8_t *mb_flags, float *last_obs,
                           float *last_values, uint8_t *last_dones);
int qs_runner_rollout_fast_blob_bytes(void);

/* PID expert of run_expert_policy.py:49-69 / run_expert_record.py:121-136 for all N docking envs: from the handle's
 * current chaser / target states, des_vel = kp (p_target + (-0.2,0,0) - p_chaser) + kd (-v_chaser), vel_controller on
 * the chaser, action = (inv(rotor2control) u - action_mean) / action_std (not clipped).  state_des [N,13] in/out is
 * the expert's persistent desired state (initialise to env.chaser_ini_state); actions [N,4] out.  Device buffers. */
int qs_expert_action(QsEnv *env, float *state_des, float kp, float kd, float *actions);

/* ---- layer-1 entry points: n independent drones / controllers (n need not equal N) ----------- */

/* Drone.step (dynamics/quadrotor.py:126-144): state [n,13] in/out, u_prev [n,4] in/out (Drone.u),
 * u [n,4] commanded control, par [n,4] = mass,Ixx,Iyy,Izz or NULL (nominal), limited [n] u8 nullable. */
int qs_drone_step(QsEnv *env, int64_t n, float *state, float *u_prev, const float *u, const float *par,
                  uint8_t *limited);

/* controller.PID (mode 0, controller/PIDController.py:179-185) or controller.vel_controller
 * (mode 1, :106-141): state_des [n,13] in/out (the reference mutates [6:12]), state [n,13],
 * state_last [n,13] (mode 1 only, else NULL), mass scalar, u_out [n,4]. */
int qs_ctrl(QsEnv *env, int64_t n, int32_t mode, float *state_des, const float *state, const float *state_last,
            float mass, float *u_out);

/* state2rel over dock-port states (docking_env.py:257-295 with quadrotor.py:213-224):
 * chaser [n,13], target [n,13] -> obs [n,12] */
int qs_rel_obs(QsEnv *env, int64_t n, const float *chaser, const float *target, float *obs);

/* ---- layer 0: utils/transform.py for n inputs -------------------------------------------------
 * op 0 quat2euler [n,4] -> [n,3] (transform.py:94-120); 1 euler2quat [n,3] -> [n,4] (:123-136);
 * 2 quat2rot [n,4] -> [n,9] row-major (:4-20; the reference's element-wise form); 3 rot2euler [n,9] -> [n,3] (:23-46) */
int qs_transform(QsEnv *env, int32_t op, int64_t n, const float *in, float *out);

#ifdef __cplusplus
}
#endif
#endif /* QUADSIM_H */
