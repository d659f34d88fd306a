/*
 * sg_host_config.hpp -- fills the SgDev parameter block for a registered env id (host side).
 * Constants restate gym_space/__init__.py:26-146 (ids, kwargs, max_episode_steps), goal.py:10-72,
 * kepler.py:17-18,189-231, hexagonal_tiling.py:8-48,161-174 and helpers.py:19 of the reference.
 */
#ifndef SG_HOST_CONFIG_HPP
#define SG_HOST_CONFIG_HPP

#include <cmath>
#include <cstring>

#include "sg_config.h"

namespace sg {

// hexagonal_tiling.py:161-174: smallest rows x cols hexagon grid with >= min_tiles tiles that fits the world
inline void tiling_grid(int min_tiles, double world, int &rows, int &cols, double &a) {
    const double s3 = std::sqrt(3.0);
    int r = (int)std::ceil(std::sqrt(72.0 * s3 * min_tiles - 6.0 * s3 + 12.0) / 12.0 - 0.25 + s3 / 12.0);
    int c;
    for (;; ++r) {
        c = (int)std::floor(2.0 * s3 * r / 3.0 - 1.0 / 3.0 + s3 / 3.0);
        if (r * c >= min_tiles) break;
    }
    rows = r; cols = c;
    a = 2.0 * s3 * world / (3.0 * (2.0 * r + 1.0));
}

inline void fill_common(SgDev &d) {
    std::memset(&d, 0, sizeof(d));
    d.max_episode_steps = 500;  // gym_space/__init__.py:29,45,61,82
    d.auto_reset = 1;
    d.h = 0.07f;                // goal.py:66; gym_space/__init__.py:76
    d.max_engine_force = 0.4f;  // gym_space/__init__.py:38; kepler.py:199
    d.omega_max = (float)(0.7 * 6.0);  // goal.py:142, kepler.py:263 with max_abs_vel_angle = 6
    d.max_thruster_force = 0.05f; d.inv_moi = 100.0f; d.omega_limit = 6.0f;  // goal.py:46,67; __init__.py:33
}

inline void fill_goal(SgDev &d, int n_planets) {
    fill_common(d);
    const double G = 6.6743e-11, world = 3.0, s3 = std::sqrt(3.0);
    d.family = SG_FAMILY_GOAL;
    d.n_planets = n_planets;
    d.half_world = (float)(world / 2);
    d.two_over_world = (float)(2.0 / world);
    d.gm = (float)(G * 1.0 * (1e9 / n_planets));  // goal.py:14,43,46
    int n_objects = n_planets + 2, rows, cols;
    int min_tiles = (n_planets == 2) ? n_objects : (int)std::ceil(n_objects / 0.6);  // hexagonal_tiling.py:8,26-29
    double a;
    tiling_grid(min_tiles, world, rows, cols, a);
    const double hex_h = a * s3, planet_r = 0.75 * hex_h / 2, small_r = planet_r / 2;  // :37,45-48
    d.planet_r = (float)planet_r; d.planet_r_d = planet_r;
    d.goal_r2 = small_r * small_r;
    d.danger_r2 = (planet_r + 0.25) * (planet_r + 0.25);  // goal.py:24
    d.survival = 0.2; d.goal_scale = 5.0 * 100.0; d.safety_scale = 10.0 * 100.0; d.sparse = 5.0;  // __init__.py:34-37, goal.py:16
    d.t_rows = rows; d.t_cols = cols; d.t_tiles = rows * cols;
    d.t_cols_rcp16 = (65536u + (uint32_t)cols - 1u) / (uint32_t)cols;
    d.t_a = (float)a; d.t_hex_h = (float)hex_h;
    d.t_x0 = (float)(-world / 2 + a);          // hexagonal_tiling.py:145 (hex_width / 2 = a)
    d.t_y0 = (float)(world / 2 - hex_h / 2);   // :146
    d.t_free_x = (float)(world - (3 * a * (cols - 1) / 2 + 2 * a));  // :39,71
    d.noise_ship = (float)(hex_h / 2 - small_r);
    d.noise_planet = (float)(hex_h / 2 - planet_r);
    d.noise_goal = (float)(hex_h / 2 - small_r);
    d.vel_std = 0.07f;                        // goal.py:141
    d.omega_std = (float)(0.7 * 6.0 / 3.0);   // goal.py:142-143
}

inline void fill_kepler(SgDev &d, double a, double ecc, double phi, int randomize) {
    fill_common(d);
    const double G = 6.6743e-11;
    d.family = SG_FAMILY_KEPLER;
    d.n_planets = 1;
    d.half_world = 3.0f; d.two_over_world = (float)(2.0 / 6.0);  // kepler.py:216
    d.gm = (float)(G * 1.0 * 6e8);  // kepler.py:204
    d.planet_r = 0.2f; d.planet_r_d = 0.2; d.border_r = 3.0f;  // kepler.py:17-18
    d.k_a = a; d.k_ecc = ecc; d.k_phi = phi;
    d.k_b = std::sqrt(a * a * (1 - ecc * ecc));   // kepler.py:43-45
    // kepler.py:47-49.  Separate statements: with -ffp-contract=on a fused a*a - b*b can come out slightly negative for
    // the circular orbit (the reference gets +2.2e-16 there, i.e. c = 1.5e-8).
    const double aa = a * a;
    const double bb = d.k_b * d.k_b;
    const double c2 = aa - bb;
    d.k_c = c2 > 0 ? std::sqrt(c2) : 0.0;
    d.k_cos = std::cos(phi); d.k_sin = std::sin(phi);
    d.k_gm = G * 6e8;
    d.k_C = 0.01; d.k_Cr = 2.0; d.k_Ca = 0.5f;  // gym_space/__init__.py:86-88
    d.randomize_orbit = randomize;
    d.vel_std = 0.05f;                        // kepler.py:261
    d.omega_std = (float)(0.7 * 6.0 / 5.0);   // kepler.py:263-264
    d.kep_rmin = 0.2f + 0.5f; d.kep_rmax = 3.0f - 0.5f;  // kepler.py:235-237
}

// returns 0 on success, -1 for an id the engine does not serve
inline int fill_config(const char *id, SgDev &d) {
    if (!std::strcmp(id, "GoalContinuous2P-v0")) { fill_goal(d, 2); return 0; }
    if (!std::strcmp(id, "GoalContinuous3P-v0")) { fill_goal(d, 3); return 0; }
    if (!std::strcmp(id, "GoalContinuous4P-v0")) { fill_goal(d, 4); return 0; }
    if (!std::strcmp(id, "KeplerCircleOrbit-v0")) { fill_kepler(d, 1.2, 0.0, 0.0, 0); return 0; }
    if (!std::strcmp(id, "KeplerEllipseEasy-v0")) { fill_kepler(d, 1.2, 0.5, 0.8, 0); return 0; }
    if (!std::strcmp(id, "KeplerEllipseHard-v0")) { fill_kepler(d, 1.2, 0.725, 3.925, 0); return 0; }
    if (!std::strcmp(id, "KeplerRandomOrbits-v0")) { fill_kepler(d, 1.2, 0.5, 3.75, 1); return 0; }  // kepler.py:193-195
    // discrete-action ids, registered by keyboard_agent.py:10-74 (same classes, Steering.velocity): Goal with
    // max_engine_force = 1; KeplerDiscrete-v0 without max_episode_steps, i.e. no TimeLimit
    if (!std::strncmp(id, "GoalDiscrete", 12) && id[12] >= '2' && id[12] <= '4' && !std::strcmp(id + 13, "-v0")) {
        fill_goal(d, id[12] - '0');
        d.max_engine_force = 1.0f; d.discrete_actions = 1;
        return 0;
    }
    if (!std::strcmp(id, "KeplerDiscrete-v0")) {
        fill_kepler(d, 1.2, 0.0, 0.0, 0);
        d.discrete_actions = 1; d.max_episode_steps = 2147483647;
        return 0;
    }
    return -1;
}

// The constructor kwargs of the reference classes on top of an id's registered ones (sg_params, include/spacegym.h):
// GoalEnv.__init__ goal.py:18-72, KeplerEnv.__init__ kepler.py:189-231; gym.make(id, **kwargs) overrides what
// gym_space/__init__.py:26-146 registered the same way.  A field that is NaN (ints: negative) keeps the id's value.
// Returns NULL, or what is wrong with the parameters.  P: any struct with sg_params' fields.
template <typename P>
inline const char *apply_params(const P &p, SgDev &d) {
    auto given = [](double v) { return !std::isnan(v); };
    const int discrete = d.discrete_actions, max_steps = d.max_episode_steps;
    float engine = d.max_engine_force, inv_moi = d.inv_moi;
    if (given(p.max_engine_force)) {
        if (!(p.max_engine_force >= 0.0 && p.max_engine_force <= 4.0)) return "max_engine_force must be in [0, 4]";
        engine = (float)p.max_engine_force;
    }
    if (given(p.ship_moi)) {
        if (!(p.ship_moi > 0.0)) return "ship_moi must be positive";
        inv_moi = (float)(1.0 / p.ship_moi);
    }
    if (d.family == SG_FAMILY_GOAL) {
        if (given(p.ref_orbit_a) || given(p.ref_orbit_eccentricity) || given(p.ref_orbit_angle) || given(p.numerator_C) ||
            given(p.rad_penalty_C) || given(p.act_penalty_C) || given(p.step_size) || p.randomize >= 0)
            return "a KeplerEnv keyword (ref_orbit_*, *_C, step_size, randomize) for a Goal id: GoalEnv.__init__ has no such argument "
                   "(its step_size is fixed at 0.07, goal.py:66)";
        if (p.n_planets >= 0 && p.n_planets != d.n_planets) {
            // (n_planets = 1 is another sampler in the reference, goal.py:78-107, used by no registered id: not served)
            if (p.n_planets < 2 || p.n_planets > SG_MAX_PLANETS) return "n_planets must be 2, 3 or 4";
            fill_goal(d, p.n_planets);
        }
        // GoalEnv._reward (goal.py:147-158) with the scales of the constructor (goal.py:48-51) and _distance_fctr = 100 (:16)
        if (given(p.survival_reward_scale)) d.survival = p.survival_reward_scale;
        if (given(p.goal_vel_reward_scale)) d.goal_scale = p.goal_vel_reward_scale * 100.0;
        if (given(p.safety_reward_scale)) d.safety_scale = p.safety_reward_scale * 100.0;
        if (given(p.goal_sparse_reward)) d.sparse = p.goal_sparse_reward;
        if (given(p.danger_zone)) {  // goal.py:32,221
            if (!(p.danger_zone >= 0.0)) return "danger_zone must not be negative";
            d.danger_r2 = (d.planet_r_d + p.danger_zone) * (d.planet_r_d + p.danger_zone);
        }
    } else {
        if (given(p.survival_reward_scale) || given(p.goal_vel_reward_scale) || given(p.safety_reward_scale) || given(p.goal_sparse_reward) ||
            given(p.danger_zone) || p.n_planets >= 0)
            return "a GoalEnv keyword (*_reward_scale, goal_sparse_reward, danger_zone, n_planets) for a Kepler id: KeplerEnv.__init__ "
                   "has no such argument";
        const double a = given(p.ref_orbit_a) ? p.ref_orbit_a : d.k_a, ecc = given(p.ref_orbit_eccentricity) ? p.ref_orbit_eccentricity : d.k_ecc;
        const double phi = given(p.ref_orbit_angle) ? p.ref_orbit_angle : d.k_phi;
        if (!(a > 0.0)) return "ref_orbit_a must be positive";
        if (!(ecc >= 0.0 && ecc < 1.0)) return "ref_orbit_eccentricity must be in [0, 1)";
        const double C = given(p.numerator_C) ? p.numerator_C : d.k_C, Cr = given(p.rad_penalty_C) ? p.rad_penalty_C : d.k_Cr;
        const float Ca = given(p.act_penalty_C) ? (float)p.act_penalty_C : d.k_Ca, h = given(p.step_size) ? (float)p.step_size : d.h;
        if (!(h > 0.0f)) return "step_size must be positive";
        const int rnd = p.randomize >= 0 ? (p.randomize ? 1 : 0) : d.randomize_orbit;
        fill_kepler(d, a, ecc, phi, rnd);  // (derives b, c, cos / sin of the angle: kepler.py:43-58)
        d.k_C = C; d.k_Cr = Cr; d.k_Ca = Ca; d.h = h;
    }
    d.discrete_actions = discrete; d.max_episode_steps = max_steps;
    d.max_engine_force = engine; d.inv_moi = inv_moi;
    return nullptr;
}

// Which integrator serves a parameter block.  The fast step's closed-form thrust integrals (sg_device.hpp, Integrator::fast_step)
// are series for a heading advance of at most 0.36 rad per env-step: |omega| = 5 |a1| <= 5 with Steering.velocity, so
// step_size <= 0.072 -- every registered id (0.07).  A longer env-step (KeplerEnv's constructor default is 0.1, kepler.py:197) is
// integrated by the kernels written for Steering.acceleration (one Dormand-Prince step tried first, heading by sincos_poly),
// which serve Steering.velocity as the special case alpha = 0, omega = 5 a1.  Their heading advance within an env-step is at most
// omega_limit * h + alpha_max h^2 / 2 and must stay within sincos_poly's pi / 4.
inline bool needs_general_kernels(const SgDev &d) { return d.steering_acceleration || !(5.0f * d.h <= 0.36f); }
inline bool step_size_supported(const SgDev &d) {
    const double alpha_max = (double)d.max_thruster_force * d.inv_moi, h = d.h;
    const double adv = d.steering_acceleration ? d.omega_limit * h + 0.5 * alpha_max * h * h : 5.0 * h;
    return adv <= 0.785;
}

inline int obs_dim(const SgDev &d) { return d.family == SG_FAMILY_GOAL ? 7 + 2 * d.n_planets + 2 : 10; }

}  // namespace sg
#endif
