/*
 * spacegym_oracle.h -- CPU restatement (fp64, scalar) of the Space-Gym step path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load this library; the product path (space_gym_amd/) never does.
 *
 * Each function cites the reference file:line (under /root/reference) it restates.  The
 * integrator + event root finder live in scipy (unpinned by the reference; 1.15.3 in the
 * build image): scipy/integrate/_ivp/{rk.py,ivp.py,common.py,base.py} and
 * scipy.optimize.brentq -- restated here from their published algorithms
 * (Dormand-Prince 5(4) with Shampine's 4th-order dense output; Hairer's initial-step rule;
 * Brent's root finder) and pinned against golden vectors captured from the reference
 * itself in the build container (tests/golden/step_*.npz, tools/gen_golden.py).
 */
#ifndef SPACEGYM_ORACLE_H
#define SPACEGYM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGO_MAX_PLANETS 4
#define SGO_FAMILY_GOAL 0
#define SGO_FAMILY_KEPLER 1

typedef struct sgo_params {
    int32_t family;      /* SGO_FAMILY_* */
    int32_t n_planets;   /* bodies in `planets` (Kepler: central planet + zero-mass border) */
    double step_size;    /* goal.py:66, gym_space/__init__.py:76 -> 0.07 */
    double world_size;   /* goal.py:10 (3.0); kepler.py:216 (6.0) */
    double max_abs_vel_angle;
    double planet_mass[SGO_MAX_PLANETS];
    double planet_radius[SGO_MAX_PLANETS];
    double max_engine_force;
    double ship_mass;
    /* GoalEnv (goal.py:18-72, gym_space/__init__.py:26-70) */
    double goal_radius, danger_zone, survival_reward_scale, goal_vel_reward_scale;
    double safety_reward_scale, goal_sparse_reward, distance_fctr;
    /* KeplerEnv (kepler.py:189-231, gym_space/__init__.py:72-146) */
    double ref_orbit_a, ref_orbit_eccentricity, ref_orbit_angle;
    double numerator_C, rad_penalty_C, act_penalty_C;
    /* reset sampler (hexagonal_tiling.py:15-48) */
    int32_t tiling_rows, tiling_cols;
    double tiling_a;
    int32_t max_episode_steps; /* gym TimeLimit, gym_space/__init__.py:29,45,61,82 */
    int32_t randomize_orbit;   /* KeplerRandomOrbits-v0 */
    int32_t discrete_actions;  /* DiscreteSpaceshipEnv (spaceship_env.py:183-202): actions are int32 indices 0..5 */
    int32_t steering_acceleration; /* 1: Steering.acceleration (ship_steering=0, the classes' constructor default; no registered
                                      id): omega is integrated, the thruster is a torque (dynamic_model.py:138-141,160-161) */
    double moi, max_thruster_force; /* ship_moi 0.01 (gym_space/__init__.py:33), 0.05 (goal.py:46, kepler.py:208) */
} sgo_params;

typedef struct sgo_diag {
    int32_t n_rk_steps;   /* accepted RK45 steps */
    int32_t nfev;         /* RHS evaluations */
    int32_t event_index;  /* -1 or index of the terminal event (planets.., world_max, world_min, ang_vel) */
    double t_event;
} sgo_diag;

/* Fill params for a registered id; returns 0, or -1 for an unknown id. */
int sgo_params_for_id(const char *env_id, sgo_params *out);
int sgo_obs_dim(const sgo_params *p);

/* dynamic_model.py:94-125 make_step.  `action` is the TRANSLATED action (engine in [0,1], thruster)
 * as float32, like the reference passes it.  state is updated in place. Returns done (0/1). */
int sgo_make_step(const sgo_params *p, const double *planets_xy, double *state, const float *action,
                  sgo_diag *diag);
/* Same with the two numbers the RHS derives from the action given directly (float32-rounded for the continuous envs,
 * float64 for the discrete ones): engine_force_scalar (dynamic_model.py:171) and omega (dynamic_model.py:140). */
int sgo_make_step_forces(const sgo_params *p, const double *planets_xy, double *state, double engine_force_scalar,
                         double omega_cmd, sgo_diag *diag);
/* General form: `torque` = thruster_action * max_thruster_force (dynamic_model.py:175), used when steering_acceleration. */
int sgo_make_step_full(const sgo_params *p, const double *planets_xy, double *state, double engine_force_scalar,
                       double omega_cmd, double torque, sgo_diag *diag);

/* spaceship_env.py:113-131 (+ kepler.py:172-187). */
void sgo_make_observation(const sgo_params *p, const double *state, const double *planets_xy,
                          const double *goal_xy, double *obs);

/* goal.py:147-158,160-164,204-227.  *hit = 1 when the goal was reached (caller resamples). */
double sgo_goal_reward(const sgo_params *p, const double *state1, const double *last_xy,
                       const double *planets_xy, const double *goal_xy, int *hit);

/* kepler.py:111-156 with per-call orbit (a, ecc, angle) so RandomOrbits can vary them. */
double sgo_kepler_reward(const sgo_params *p, const double *state1, const float *action,
                         double ref_a, double ref_ecc, double ref_angle);
/* act_term = act_penalty_C * ||last_action|| already evaluated (float32 arithmetic for continuous, float64 for discrete) */
double sgo_kepler_reward_act(const sgo_params *p, const double *state1, double act_term, double ref_a, double ref_ecc,
                             double ref_angle);

/* spaceship_env.py:68-78: one env.step() on injected inputs, no reset, no goal resample.
 * raw_action is the policy output: float32[2] in [-1,1]^2, or for the discrete ids one int32 index 0..5.
 * Outputs: state (in place), obs[D], reward, done, goal_hit. */
void sgo_env_step(const sgo_params *p, const double *planets_xy, const double *goal_xy, double *state,
                  const void *raw_action, double *obs, double *reward, uint8_t *done, uint8_t *goal_hit,
                  sgo_diag *diag);

/* SpaceshipEnv.vector_field (spaceship_env.py:96-100): RHS of the ODE at `state` for the raw action: [vx, vy, omega', ax, ay, alpha]
 * where omega' is the commanded omega under Steering.velocity (the RHS overwrites it, dynamic_model.py:138-141). */
void sgo_vector_field(const sgo_params *p, const double *planets_xy, const double *state, const void *raw_action, double *field);

/* Batched form of sgo_env_step over m independent transitions (row-major arrays). threads<=1: serial. */
void sgo_env_step_batch(const sgo_params *p, int64_t m, const double *planets_xy, const double *goal_xy,
                        double *state, const void *raw_action, double *obs, double *reward,
                        uint8_t *done, uint8_t *goal_hit, sgo_diag *diag, int threads);

/* ---------------------------------------------------------------- reset sampler + vector env
 * Counter-based RNG shared with the HIP engine: Philox4x32-10, key=(seed lo, seed hi),
 * counter=(env_id, episode, block, stream).  The sampling ALGORITHM restates
 * hexagonal_tiling.py:53-134, goal.py:133-145, kepler.py:233-267; the reference's MT19937 stream is not
 * reproducible on a GPU, so parity with the reference is distributional (tests/golden/reset_*.npz). */
void sgo_philox4x32_10(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]);

typedef struct sgo_env_state {
    double state[6];
    double planets_xy[2 * SGO_MAX_PLANETS];
    double goal_xy[2];
    double orbit[3];        /* Kepler: ref_angle, ref_ecc, ref_a (per env for RandomOrbits) */
    int32_t elapsed;        /* steps in the current episode */
    uint32_t episode;       /* episode counter, part of the RNG counter */
    uint32_t goal_draws;    /* goal resamples done in this episode, part of the RNG counter */
    int32_t ship_tile, goal_tile;
    int32_t n_free;
    int32_t free_tiles[64]; /* ordered list, may hold duplicates (hexagonal_tiling.py:101-106) */
} sgo_env_state;

void sgo_env_reset(const sgo_params *p, uint64_t seed, uint32_t env_id, sgo_env_state *e);
/* Goal: hexagonal_tiling.py:95-134 find_new_goal (called on a hit, goal.py:157). */
void sgo_env_resample_goal(const sgo_params *p, uint64_t seed, uint32_t env_id, sgo_env_state *e);
/* One vector-env step with TimeLimit + auto-reset, mirroring the engine's sg_step semantics
 * (DESIGN.md §step semantics).  obs gets the post-reset observation for finished envs. */
void sgo_vec_step(const sgo_params *p, uint64_t seed, int64_t b, uint32_t env_id0, sgo_env_state *envs,
                  const void *raw_actions, double *obs, double *reward, uint8_t *done, uint8_t *truncated,
                  double *terminal_obs, int threads);
void sgo_vec_reset(const sgo_params *p, uint64_t seed, int64_t b, uint32_t env_id0, sgo_env_state *envs,
                   double *obs, int threads);

#ifdef __cplusplus
}
#endif
#endif
