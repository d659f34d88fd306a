/*
 * spacegym.h -- C ABI of the MI355X batched Space-Gym step engine (libspacegym_hip.so).
 *
 * The reference (MIMUW-RL/space-gym) has no FFI; the seam this library sits behind is its gym.Env
 * protocol.  Each entry point names the reference interface it replaces (paths under the reference
 * repo root).  One handle advances `num_envs` independent (ship, planets, goal/orbit) instances in
 * lock-step on one GPU; envs never interact (gym_space/dynamic_model.py:145-165 sums only an env's
 * own planets), so a multi-GPU job is one handle per device with disjoint `env_index_base`.
 *
 * Conventions
 *   - every function returns 0 on success or a negative SG_ERR_* code; sg_last_error() gives the text;
 *   - plain pointers and sizes only; "host" pointers are ordinary memory, "device" pointers are HIP
 *     device memory on the handle's GPU, owned by the caller;
 *   - no allocation, no host synchronisation and no host<->device copy inside the *_device calls:
 *     they only enqueue kernels on the given stream (hipGraph-capturable);
 *   - the host-buffer calls (sg_reset, sg_step, sg_get_state, sg_set_state, sg_vector_field, sg_save_state, sg_load_state,
 *     sg_seed, sg_set_auto_reset) run on the handle's own stream, wait for whatever the *_device calls have enqueued on
 *     the caller's streams before, and return when they are complete -- no manual synchronisation between the two kinds;
 *   - a handle is not thread-safe; independent handles are.
 *   - observations/rewards are float32 (the reference returns float64; parity tolerance in DESIGN.md).
 */
#ifndef SPACEGYM_H
#define SPACEGYM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SG_OK 0
#define SG_ERR_INVALID -1   /* bad argument / unknown env id */
#define SG_ERR_HIP -2       /* HIP runtime error (message has the HIP text) */
#define SG_ERR_NO_DEVICE -3 /* no usable GPU: the engine has no CPU path */

typedef struct sg_env sg_env; /* opaque handle */

typedef struct sg_config {
    /* Registered id, as in gym_space/__init__.py:26-146: GoalContinuous{2,3,4}P-v0,
     * Kepler{CircleOrbit,EllipseEasy,EllipseHard,RandomOrbits}-v0; and the discrete-action ids of keyboard_agent.py:10-74:
     * GoalDiscrete{2,3,4}-v0, KeplerDiscrete-v0. */
    char env_id[64];
    int64_t num_envs;          /* batch on this device */
    uint64_t seed;             /* SpaceshipEnv.seed, spaceship_env.py:92-94 / goal.py:74-77 */
    uint32_t env_index_base;   /* global index of local env 0; the RNG is keyed by the global index */
    int32_t max_episode_steps; /* 0 -> the id's registered value (500): gym TimeLimit, __init__.py:29 */
    int32_t auto_reset;        /* 1: finished envs restart inside step (VectorEnv semantics); 0: they keep
                                  their terminal state, like a bare reference env */
    int32_t steering;          /* 0: Steering.velocity, what every registered id passes (ship_steering=1, __init__.py:32);
                                  1: Steering.acceleration (ship_steering=0, the constructor default of GoalEnv / KeplerEnv:
                                  goal.py:27, kepler.py:198): omega is integrated, the thruster is a torque
                                  (dynamic_model.py:138-141,160-161) and the angular-velocity event (:210-212) is live */
} sg_config;

/* The keyword arguments of the reference's constructors, GoalEnv.__init__ (gym_space/envs/goal.py:18-31) and
 * KeplerEnv.__init__ (gym_space/envs/kepler.py:189-203), on top of what gym_space/__init__.py:26-146 registers for the id --
 * what gym.make(id, **kwargs) does.  sg_params_init() sets every field to "keep the id's registered value" (NaN; the integers
 * -1); set the ones to override.  ship_steering is sg_config.steering; fixed_position, reward_value and renderer_kwargs are
 * read by nothing on the step path of the reference and have no field.  Keywords of the other family are refused, as Python
 * refuses an unexpected keyword argument. */
typedef struct sg_params {
    uint32_t struct_size;          /* sizeof(sg_params), set by sg_params_init */
    int32_t n_planets;             /* GoalEnv: 2, 3 or 4 (goal.py:25; 1 is a different sampler, :78-107, not served) */
    int32_t randomize;             /* KeplerEnv: 0 / 1 (kepler.py:191,257-259: a new reference orbit every episode) */
    int32_t reserved;
    /* GoalEnv._reward, goal.py:147-158 */
    double goal_vel_reward_scale;  /* goal.py:20,49 (x _distance_fctr = 100, :16,163) */
    double safety_reward_scale;    /* goal.py:21,51 (x 100, :226) */
    double goal_sparse_reward;     /* goal.py:22,50,155 */
    double survival_reward_scale;  /* goal.py:25,48,150 */
    double danger_zone;            /* goal.py:24,32,221: the safety term acts within this distance of the nearest planet's surface */
    /* KeplerEnv._dense_reward5, kepler.py:111-150 */
    double ref_orbit_a, ref_orbit_eccentricity, ref_orbit_angle; /* kepler.py:192-194 */
    double numerator_C, rad_penalty_C, act_penalty_C;             /* kepler.py:196-198,138-150 */
    double step_size;              /* kepler.py:199 (the constructor default is 0.1, every registered id passes 0.07); GoalEnv's is
                                      fixed at 0.07 (goal.py:66).  Up to 0.072 the env-step is one Nystrom step with the thrust in
                                      closed form; longer ones go through the Dormand-Prince kernels; refused where the heading
                                      could advance by more than pi / 4 within one env-step (Steering.velocity: above 0.157) */
    /* ShipParams (goal.py:45-47, kepler.py:207-209) */
    double ship_moi;               /* moment of inertia (Steering.acceleration only: dynamic_model.py:160-161) */
    double max_engine_force;       /* dynamic_model.py:171; 0 .. 4 */
} sg_params;
void sg_params_init(sg_params *params);

/* GoalContinuousEnv(**kwargs) / KeplerContinuousEnv(**kwargs) construction (goal.py:18-72,
 * kepler.py:189-231) for num_envs instances on GPU `device`: sg_create with the kwargs the id was registered with
 * (gym_space/__init__.py:26-146), sg_create_ex with `params` on top of them (NULL: none).
 * num_envs is at most 4 194 304 per handle (the episode queue of the rollout kernels is addressed by 32-bit offsets); larger
 * batches are several handles with disjoint env_index_base.  Besides its device columns (~100-490 B per env) a handle owns two
 * page-locked result blocks for sg_step_begin / sg_step_end of (2 obs_dim + 1.5) * 4 B per env each. */
int sg_create(const sg_config *cfg, int device, sg_env **out);
int sg_create_ex(const sg_config *cfg, const sg_params *params, int device, sg_env **out);
/* The parameters a handle was built with, every field filled in (the effective values). */
int sg_get_params(const sg_env *env, sg_params *out);
int sg_destroy(sg_env *env);
/* The same for a batch of cfg->num_envs envs cut into contiguous blocks over n_devices GPUs (one handle per device, the
 * remainder spread over the first ones; env_index_base of block k = cfg->env_index_base + its first env): envs never
 * interact (gym_space/dynamic_model.py:145-165) and the RNG is keyed by the global env index, so the blocks together are the
 * same envs as one handle of the whole batch.  handles_out has n_devices entries; on failure none is left allocated.  The
 * exchange a single-process VectorEnv view needs on top (a rooted gather of obs | reward | done per step) is
 * space_gym_amd/sharded.py's, over torch.distributed (RCCL). */
int sg_create_sharded(const sg_config *cfg, int n_devices, const int *devices, sg_env **handles_out);
int sg_create_sharded_ex(const sg_config *cfg, const sg_params *params, int n_devices, const int *devices, sg_env **handles_out);
const char *sg_last_error(const sg_env *env); /* env may be NULL: last error of a failed sg_create */

int64_t sg_num_envs(const sg_env *env);
int32_t sg_obs_dim(const sg_env *env);      /* 7 + 2N + 2 (Goal, spaceship_env.py:102-111,124-131); 10 (Kepler, kepler.py:158-187) */
int32_t sg_num_planets(const sg_env *env);  /* planets with a per-env position: N (Goal), 0 (Kepler) */
int32_t sg_discrete_actions(const sg_env *env); /* 1 for the ids with Discrete(6) actions (spaceship_env.py:184-187) */

/* SpaceshipEnv.seed (spaceship_env.py:92-94): takes effect at the next reset. */
int sg_seed(sg_env *env, uint64_t seed);
int sg_set_auto_reset(sg_env *env, int32_t on);

/* SpaceshipEnv.reset (spaceship_env.py:59-66) for every env; obs is float32 [num_envs, obs_dim]. */
int sg_reset(sg_env *env, float *obs_host);
int sg_reset_device(sg_env *env, float *obs_dev, void *hip_stream);

/* SpaceshipEnv.step (spaceship_env.py:68-78) for every env, plus what gym.wrappers.TimeLimit and a
 * VectorEnv add around it (elapsed-step counter, truncation, auto-reset).
 *   actions     continuous ids: float32 [num_envs, 2] raw policy output in [-1, 1]^2 (clamped into range on the device;
 *               the reference asserts, spaceship_env.py:71); discrete ids: int32 [num_envs] indices 0..5
 *               (spaceship_env.py:183-202; out-of-range indices act as 0, the reference raises ValueError)
 *   obs         float32 [num_envs, obs_dim]; for a finished env (auto_reset on) the first observation of
 *               its next episode
 *   reward      float32 [num_envs]
 *   done        uint8   [num_envs]  terminal event or truncation
 *   truncated   uint8   [num_envs]  elapsed == max_episode_steps without a terminal event ("TimeLimit.truncated")
 *   terminal_obs  optional float32 [num_envs, obs_dim]; rows of finished envs receive the last observation of
 *               the episode that ended, other rows are left untouched.  May be NULL. */
int sg_step(sg_env *env, const void *actions_host, float *obs_host, float *reward_host, uint8_t *done_host,
            uint8_t *truncated_host, float *terminal_obs_host);
int sg_step_device(sg_env *env, const void *actions_dev, float *obs_dev, float *reward_dev, uint8_t *done_dev,
                   uint8_t *truncated_dev, float *terminal_obs_dev, void *hip_stream);

/* The same step in two halves, for callers that have something else to do while it runs -- gym.vector's
 * step_async() / step_wait() around SpaceshipEnv.step (spaceship_env.py:68-78).
 *   sg_step_begin  enqueues, on the handle's stream, the step kernel with a page-locked result block owned by the handle as
 *                  its output (the kernel stores across PCIe itself: no copy commands behind it), and returns without
 *                  waiting.  `actions_host` must stay unchanged until sg_step_end: page-locked memory (sg_host_alloc) is
 *                  read by the kernel where it is, anything else is copied to the device first.  want_terminal_obs != 0 adds the terminal observations to the block (rows of envs that did
 *                  not finish read NaN).
 *   sg_step_end    waits for that step and returns pointers into its result block: obs [num_envs, obs_dim], reward, done,
 *                  truncated as in sg_step, terminal_obs or NULL (any out pointer may be NULL).  The handle alternates
 *                  between two blocks: the pointers stay valid until the sg_step_begin after next -- the results of step t can
 *                  be read while step t + 1 is in flight, and the kernel enqueued by the sg_step_begin of step t + 2 writes
 *                  them again (it stores into the block while it runs).
 * One step may be in flight per handle; any other call on the handle between the two is ordered behind the step. */
/* (The two result blocks are allocated by the first sg_step_begin -- callers of the device-pointer entry points never pay for
 *  them; where device-mapped page-locked memory of that size cannot be had the handle falls back to a device block and copies.
 *  Whether `actions_host` is page-locked is asked on every call.  If sg_step_end fails the step is no longer in flight: the
 *  handle accepts the next sg_step_begin.) */
int sg_step_begin(sg_env *env, const void *actions_host, int32_t want_terminal_obs);
int sg_step_end(sg_env *env, const float **obs, const float **reward, const uint8_t **done, const uint8_t **truncated,
                const float **terminal_obs);

/* `n_steps` consecutive steps for pre-supplied actions (open-loop rollout, e.g. random-action benchmarking or replaying an
 * action tape): actions [n_steps, num_envs, 2] (discrete ids: int32 [n_steps, num_envs]), obs [n_steps, num_envs, obs_dim], reward/done/truncated [n_steps, num_envs].
 * Bit-identical to n_steps calls of sg_step_device.  All steps run in ONE kernel launch with the env state held in
 * registers; sg_set_unfused_rollout(env, 1) switches to n_steps launches of the step kernel. */
int sg_rollout_device(sg_env *env, int32_t n_steps, const void *actions_dev, float *obs_dev, float *reward_dev,
                      uint8_t *done_dev, uint8_t *truncated_dev, void *hip_stream);
int sg_set_unfused_rollout(sg_env *env, int32_t on);

/* The same rollout, also returning what a finished env's LAST observation was: in obs[t] a finished env already shows the
 * first observation of its next episode (VectorEnv convention), so the observation SpaceshipEnv.step returned with
 * done=True (spaceship_env.py:75-78) -- needed to bootstrap from truncated episodes -- would otherwise be lost.  One record
 * per finished env-step is appended to the list, in no particular order (device memory, owned by the caller):
 *   count     uint32 [1]            records appended by this call (set to 0 first); it keeps counting past `capacity`, the
 *                                   excess records are dropped -- size the list for n_steps * num_envs * (finish rate ~2 %)
 *   step_env  int32  [capacity, 2]  (step within this call, env index) of each record
 *   obs       float32 [capacity, obs_dim]
 * With auto_reset off nothing is appended (obs[t] is the terminal observation itself). */
typedef struct sg_terminal_list {
    uint32_t *count;
    int32_t *step_env;
    float *obs;
    uint32_t capacity;
} sg_terminal_list;
int sg_rollout_device_terminal(sg_env *env, int32_t n_steps, const void *actions_dev, float *obs_dev, float *reward_dev,
                               uint8_t *done_dev, uint8_t *truncated_dev, const sg_terminal_list *list, void *hip_stream);

/* The wave-pair rollout kernels bound every wait between their waves; a wait that runs out (never, on working hardware)
 * invalidates that rollout and is recorded on the handle: every later call on the handle then fails with SG_ERR_HIP until
 * sg_check_status -- which waits for the work enqueued so far, reports the condition and clears it -- has been called. */
int sg_check_status(sg_env *env);

/* Event counters of a handle (SURVEY section 5 "metrics"; the reference itself only keeps KeplerEnv's last penalties as
 * attributes, kepler.py:146-149): env-steps taken, episodes finished (terminal event or truncation), truncations
 * (gym.wrappers.TimeLimit), goals reached (GoalEnv._reward's `goal_pos` test, goal.py:154-157; 0 for the Kepler ids) since
 * counting was switched on or last reset.  Off by default -- the step and rollout kernels then do nothing for it; on, every
 * step / rollout call is followed by a pass over the done / truncated flags it wrote, and the reward code adds its goal hits. */
typedef struct sg_counters {
    uint64_t env_steps, episodes_finished, truncations, goal_hits;
} sg_counters;
/* (All four are counted on the device by the work the calls enqueue, so a replayed hipGraph of *_device calls counts as
 *  well; switching the counters on or off after a graph was captured does not change that graph.) */
int sg_set_counters(sg_env *env, int32_t on);                              /* switching (on or off) zeroes the counters */
int sg_get_counters(sg_env *env, sg_counters *out, int32_t reset);         /* waits for the enqueued work first */

/* On-device action source for sg_rollout_device: the uniformly random policy (what the reference's README loop and the
 * benchmark use: env.action_space.sample(), gym spaces Box / Discrete).  Fills actions_dev [n_steps, num_envs, 2] float32
 * with i.i.d. U(-1, 1) values (discrete ids: int32 [n_steps, num_envs] uniform in 0..5).  Entry (t, i) is a function of
 * (seed, env_index_base + i, first_step + t) only (Philox4x32-10), so it does not depend on how a job is sharded over
 * GPUs or cut into calls.  Enqueues one kernel on the stream. */
int sg_random_actions_device(sg_env *env, int32_t n_steps, uint64_t seed, uint64_t first_step, void *actions_dev, void *hip_stream);

/* State access (the reference exposes env._ship_state._state_vec, planet.center_pos, env.goal_pos as plain
 * attributes; golden-vector injection needs the same).  Host arrays, any may be NULL to skip:
 *   ship    float32 [num_envs, 6]   x, y, theta, vx, vy, omega   (dynamic_model.py:40)
 *   planets float32 [num_envs, N, 2]                              (Goal only)
 *   goal    float32 [num_envs, 2]   Goal: goal position; KeplerRandomOrbits: (ref_orbit_angle, ref_orbit_eccentricity)
 *   elapsed int32   [num_envs]      steps taken in the current episode */
int sg_get_state(sg_env *env, float *ship, float *planets, float *goal, int32_t *elapsed);
int sg_set_state(sg_env *env, const float *ship, const float *planets, const float *goal, const int32_t *elapsed);

/* Complete snapshot of a handle (checkpoint / restore), as an opaque blob of sg_state_bytes(env) bytes of host memory: every
 * per-env column -- ship, planets, goal / orbit, step and episode counters, and the tiling state HexagonalTiling keeps
 * between goal hits (hexagonal_tiling.py:99-128: free-tile list, ship / goal tile, column shifts) -- plus the RNG key.
 * Loading it into a handle of the same env id and batch size makes the following steps bit-identical to those that
 * followed the save. */
size_t sg_state_bytes(const sg_env *env);
int sg_save_state(sg_env *env, void *blob_host, size_t bytes);
int sg_load_state(sg_env *env, const void *blob_host, size_t bytes);

/* SpaceshipEnv.vector_field(raw_action, state_vec=None) (spaceship_env.py:96-100): the RHS of the ODE,
 * out float32 [num_envs, 6] = (vx, vy, omega', ax, ay, angular acceleration) at each env's current planets and either its
 * current ship state (ship == NULL) or the given one (float32 [num_envs, 6]).  Host arrays; actions as in sg_step. */
int sg_vector_field(sg_env *env, const void *actions_host, const float *ship_host, float *out_host);

/* Page-locked host memory for the arrays passed to sg_reset / sg_step (optional: any host memory works, pinned memory
 * makes the per-step copies plain DMA).  sg_host_alloc returns NULL on failure. */
void *sg_host_alloc(size_t bytes);
void sg_host_free(void *ptr);

/* Measurement aid (no reference counterpart): with profiling on, each step-kernel launch carries start/stop events
 * that timestamp the dispatch itself; sg_get_profile returns and clears the durations recorded so far (milliseconds).
 * The caller synchronises the stream(s) first. */
int sg_set_profiling(sg_env *env, int32_t on);
int sg_get_profile(sg_env *env, int64_t *launches, double *total_ms, double *min_ms, double *max_ms);

/* Measurement aid: name of the kernel (as rocprofv3 --kernel-trace prints it) that sg_rollout_device launches for n_steps
 * steps on this handle; valid until the next call on the handle. */
const char *sg_rollout_kernel(sg_env *env, int32_t n_steps);

/* The HIP stream the host-buffer calls run on (hipStream_t), for callers that want to order work after it.  Created by
 * sg_create, destroyed by sg_destroy: the value is valid for the lifetime of the handle and must not be destroyed by the caller. */
void *sg_stream(const sg_env *env);

const char *sg_version(void);

#ifdef __cplusplus
}
#endif
#endif
