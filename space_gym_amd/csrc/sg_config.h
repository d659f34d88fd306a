/*
 * sg_config.h -- POD parameter block handed by value to every kernel (lands in SGPRs), filled on the
 * host by sg_fill_config() from the registered env id.  Values follow the reference constants cited
 * next to each field (paths under /root/reference).
 */
#ifndef SG_CONFIG_H
#define SG_CONFIG_H

#include <stdint.h>

#define SG_FAMILY_GOAL 0
#define SG_FAMILY_KEPLER 1
#define SG_MAX_PLANETS 4

typedef struct SgDev {
    int32_t family;             /* SG_FAMILY_* */
    int32_t n_planets;          /* Goal: 2..4 gravitating planets; Kepler: 1 (+ zero-mass border circle) */
    int32_t max_episode_steps;  /* gym TimeLimit, gym_space/__init__.py:29,45,61,82 */
    int32_t auto_reset;         /* 1: finished envs are re-initialised inside the step kernel */
    int32_t num_envs;
    uint32_t env_index_base;    /* global index of local env 0 (RNG is keyed by the global index) */
    uint32_t seed_lo, seed_hi;
    int32_t randomize_orbit;    /* KeplerRandomOrbits-v0, kepler.py:257-259 */
    int32_t discrete_actions;   /* DiscreteSpaceshipEnv (spaceship_env.py:183-202): actions are int32 indices 0..5 */
    int32_t steering_acceleration; /* 1: Steering.acceleration (ship_steering=0: constructor default of the reference classes, used
                                      by no registered id): omega is a state, thruster = torque (dynamic_model.py:138-141,160-161) */

    float h;                    /* step_size 0.07: goal.py:66, gym_space/__init__.py:76 */
    float half_world;           /* world_size / 2: goal.py:10 (3.0), kepler.py:216 (6.0) */
    float two_over_world;       /* lidar scale, spaceship_env.py:139 */
    float max_engine_force;     /* 0.4, gym_space/__init__.py:38 */
    float max_thruster_force;   /* 0.05: goal.py:46, kepler.py:208 */
    float inv_moi;              /* 1 / ship_moi = 100: gym_space/__init__.py:33 */
    float omega_limit;          /* max_abs_vel_angle = 6: goal.py:67, kepler.py:217 (event dynamic_model.py:210-212) */
    float gm;                   /* G * m_ship * m_planet: helpers.py:19,34; goal.py:14,43; kepler.py:204 */
    float planet_r;             /* hexagonal_tiling.py:45-47 / kepler.py:17 */
    float border_r;             /* kepler.py:18 */

    double planet_r_d;          /* planet radius in fp64 for the event-root polish */

    /* GoalEnv._reward, goal.py:147-158,160-164,204-227; scales gym_space/__init__.py:34-37 */
    double goal_r2;             /* goal_radius^2 */
    double danger_r2;           /* (planet_r + danger_zone)^2, goal.py:24 */
    double survival;            /* survival_reward_scale */
    double goal_scale;          /* goal_vel_reward_scale * _distance_fctr = 5 * 100 */
    double safety_scale;        /* safety_reward_scale * _distance_fctr = 10 * 100 */
    double sparse;              /* goal_sparse_reward */

    /* KeplerEnv._dense_reward5, kepler.py:111-150; constants gym_space/__init__.py:84-146 */
    double k_a, k_ecc, k_phi;   /* fixed reference orbit (a, eccentricity, angle) */
    double k_b, k_c;            /* semi-minor axis, focal distance: kepler.py:43-49 */
    double k_cos, k_sin;        /* cos / sin of the orbit angle (kepler.py:51-58) */
    double k_gm;                /* G * planets[0].mass, kepler.py:60 */
    double k_C, k_Cr;           /* numerator_C, rad_penalty_C */
    float k_Ca;                 /* act_penalty_C (float32 arithmetic in the reference, kepler.py:138,143) */

    /* HexagonalTiling, hexagonal_tiling.py:15-48,136-158 */
    int32_t t_rows, t_cols, t_tiles;
    uint32_t t_cols_rcp16;      /* ceil(65536 / t_cols): tile / t_cols == (tile * rcp) >> 16 for tile < 64 (no runtime division) */
    float t_a, t_hex_h;
    float t_x0, t_y0;           /* centre of tile 0 before column shifts (case A) */
    float t_free_x;             /* world_size - tiling_width */
    float noise_ship, noise_planet, noise_goal; /* hex_h/2 - object radius, hexagonal_tiling.py:132 */

    /* reset kinematics: goal.py:140-145, kepler.py:233-267 */
    float vel_std;              /* 0.07 (Goal) / 0.05 (Kepler) */
    float omega_std;            /* 0.7*6/3 (Goal) / 0.7*6/5 (Kepler) */
    float omega_max;            /* 0.7 * max_abs_vel_angle = 4.2 */
    float kep_rmin, kep_rmax;   /* planet_r + 0.5, border_r - 0.5 */
} SgDev;

#endif
