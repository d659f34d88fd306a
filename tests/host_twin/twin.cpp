// Host twin of the device math (TEST-ONLY): space_gym_amd/csrc/sg_device.hpp compiled by g++ so the fp32
// numerics of the engine can be checked against the oracle and the golden vectors without a GPU.
// It is built by tests/host_twin/build.py into tests/host_twin/_build and is never loaded by the product.
#include <cmath>
#include <cstdint>
#include <cstring>

#include "sg_device.hpp"
#include "sg_host_config.hpp"

using namespace sg;

// Steering.acceleration for the next twin_step calls (the reference's ship_steering=0; no registered id uses it)
static int g_steering_acceleration = 0;
// optional diagnostics of the next twin_step calls: how each env-step was integrated (sg::kPath*) and the probe step's
// squared error norm
static int32_t *g_path = nullptr;
static float *g_probe_err = nullptr, *g_probe_abs = nullptr;
extern "C" void twin_set_diag(int32_t *path, float *probe_err, float *probe_abs) { g_path = path; g_probe_err = probe_err; g_probe_abs = probe_abs; }
extern "C" void twin_set_steering_acceleration(int on) { g_steering_acceleration = on ? 1 : 0; }

template <int N, bool ACCEL>
static void goal_steps(const SgDev &c, int64_t m, const float *state, const float *planets, const float *goal,
                       const void *action, float *state1, float *obs, float *reward, uint8_t *done, uint8_t *hit,
                       float *t_adv, int32_t *n_rk, int32_t *event) {
    constexpr int D = 7 + 2 * N + 2;
    for (int64_t i = 0; i < m; i++) {
        GoalEnv<N> e;
        const float *s = state + 6 * i;
        e.x = s[0]; e.y = s[1]; e.th = s[2]; e.vx = s[3]; e.vy = s[4]; e.om = s[5];
        e.gx = goal[2 * i]; e.gy = goal[2 * i + 1];
        for (int j = 0; j < N; j++) { e.px[j] = planets[(i * N + j) * 2]; e.py[j] = planets[(i * N + j) * 2 + 1]; }
        float o[D], r;
        int dn, ht;
        StepResult sr;
        float a0, a1;
        load_action(c.discrete_actions != 0, action, i, a0, a1);
        goal_env_step<N, ACCEL>(c, e, a0, a1, o, r, dn, ht, sr);
        float *s1 = state1 + 6 * i;
        s1[0] = e.x; s1[1] = e.y; s1[2] = e.th; s1[3] = e.vx; s1[4] = e.vy; s1[5] = e.om;
        std::memcpy(obs + D * i, o, sizeof(o));
        reward[i] = r; done[i] = (uint8_t)dn; hit[i] = (uint8_t)ht;
        t_adv[i] = sr.t; n_rk[i] = sr.n_rk; event[i] = sr.event;
        if (g_path) g_path[i] = sr.path;
        if (g_probe_err) g_probe_err[i] = sr.probe_err;
        if (g_probe_abs) { g_probe_abs[2 * i] = sr.probe_ep; g_probe_abs[2 * i + 1] = sr.probe_ev; }
    }
}

extern "C" int twin_obs_dim(const char *env_id) {
    SgDev c;
    if (fill_config(env_id, c)) return -1;
    return obs_dim(c);
}

// Kepler: `goal` may carry a per-env orbit (phi, ecc) as for KeplerRandomOrbits-v0 (the engine stores cos/sin phi in fp64)
extern "C" int twin_step(const char *env_id, int64_t m, const float *state, const float *planets, const float *goal,
                         const void *action, float *state1, float *obs, float *reward, uint8_t *done, uint8_t *hit,
                         float *t_adv, int32_t *n_rk, int32_t *event) {
    SgDev c;
    if (fill_config(env_id, c)) return -1;
    c.steering_acceleration = g_steering_acceleration;
    if (c.family == SG_FAMILY_GOAL) {
#define TWIN_GOAL(NP)                                                                                                         \
    if (g_steering_acceleration)                                                                                              \
        goal_steps<NP, true>(c, m, state, planets, goal, action, state1, obs, reward, done, hit, t_adv, n_rk, event);          \
    else                                                                                                                      \
        goal_steps<NP, false>(c, m, state, planets, goal, action, state1, obs, reward, done, hit, t_adv, n_rk, event)
        if (c.n_planets == 2) { TWIN_GOAL(2); }
        else if (c.n_planets == 3) { TWIN_GOAL(3); }
        else { TWIN_GOAL(4); }
#undef TWIN_GOAL
        return 0;
    }
    for (int64_t i = 0; i < m; i++) {
        KeplerEnv e;
        const float *s = state + 6 * i;
        e.x = s[0]; e.y = s[1]; e.th = s[2]; e.vx = s[3]; e.vy = s[4]; e.om = s[5];
        e.phi = (float)c.k_phi; e.ecc = (float)c.k_ecc;
        Orbit ob = fixed_orbit(c);
        if (goal) {
            c.randomize_orbit = 1;
            e.phi = goal[2 * i]; e.ecc = goal[2 * i + 1];
            ob = make_orbit(c.k_a, (double)e.ecc, std::cos((double)e.phi), std::sin((double)e.phi));
        }
        float o[10], r;
        int dn;
        StepResult sr;
        float a0, a1;
        load_action(c.discrete_actions != 0, action, i, a0, a1);
        if (g_steering_acceleration) kepler_env_step<true>(c, ob, e, a0, a1, o, r, dn, sr);
        else kepler_env_step<false>(c, ob, e, a0, a1, o, r, dn, sr);
        float *s1 = state1 + 6 * i;
        s1[0] = e.x; s1[1] = e.y; s1[2] = e.th; s1[3] = e.vx; s1[4] = e.vy; s1[5] = e.om;
        std::memcpy(obs + 10 * i, o, sizeof(o));
        reward[i] = r; done[i] = (uint8_t)dn; hit[i] = 0;
        t_adv[i] = sr.t; n_rk[i] = sr.n_rk; event[i] = sr.event;
        if (g_path) g_path[i] = sr.path;
        if (g_probe_err) g_probe_err[i] = sr.probe_err;
        if (g_probe_abs) { g_probe_abs[2 * i] = sr.probe_ep; g_probe_abs[2 * i + 1] = sr.probe_ev; }
    }
    return 0;
}

// Reset sampler twin: state[6], planets[N*2], goal[2], tiles = ship | goal << 8 | case_b << 16 | flip << 17,
// free_counts, col_shift[4]; followed by `n_hits` goal resamples whose goals/tiles are appended.
template <int N>
static void goal_resets(const SgDev &c, int64_t m, uint32_t env0, uint32_t episode, int n_hits, float *state,
                        float *planets, float *goals, uint32_t *tiles, uint64_t *free_counts, float *col_shift) {
    for (int64_t i = 0; i < m; i++) {
        Tiling T;
        T.episode = episode;
        ShipInit s;
        float px[N], py[N], gx, gy;
        goal_reset<N>(c, env0 + (uint32_t)i, T, s, px, py, gx, gy);
        float *st = state + 6 * i;
        st[0] = s.x; st[1] = s.y; st[2] = s.th; st[3] = s.vx; st[4] = s.vy; st[5] = s.om;
        for (int j = 0; j < N; j++) { planets[(i * N + j) * 2] = px[j]; planets[(i * N + j) * 2 + 1] = py[j]; }
        for (int k = 0; k <= n_hits; k++) {
            if (k > 0) goal_resample(c, env0 + (uint32_t)i, T, gx, gy);
            goals[(i * (n_hits + 1) + k) * 2] = gx; goals[(i * (n_hits + 1) + k) * 2 + 1] = gy;
            tiles[i * (n_hits + 1) + k] = T.ship_tile | (T.goal_tile << 8) | (T.case_b << 16) | (T.flip << 17);
            free_counts[i * (n_hits + 1) + k] = T.free_counts;
        }
        col_shift[4 * i] = T.cs0; col_shift[4 * i + 1] = T.cs1; col_shift[4 * i + 2] = T.cs2; col_shift[4 * i + 3] = T.cs3;
    }
}

extern "C" int twin_reset(const char *env_id, uint64_t seed, int64_t m, uint32_t env0, uint32_t episode, int n_hits,
                          float *state, float *planets, float *goals, uint32_t *tiles, uint64_t *free_counts,
                          float *col_shift, float *orbit) {
    SgDev c;
    if (fill_config(env_id, c)) return -1;
    c.seed_lo = (uint32_t)seed; c.seed_hi = (uint32_t)(seed >> 32);
    if (c.family == SG_FAMILY_GOAL) {
        if (c.n_planets == 2) goal_resets<2>(c, m, env0, episode, n_hits, state, planets, goals, tiles, free_counts, col_shift);
        else if (c.n_planets == 3) goal_resets<3>(c, m, env0, episode, n_hits, state, planets, goals, tiles, free_counts, col_shift);
        else goal_resets<4>(c, m, env0, episode, n_hits, state, planets, goals, tiles, free_counts, col_shift);
        return 0;
    }
    for (int64_t i = 0; i < m; i++) {
        ShipInit s;
        float phi = (float)c.k_phi, ecc = (float)c.k_ecc;
        kepler_reset(c, env0 + (uint32_t)i, episode, s, phi, ecc);
        float *st = state + 6 * i;
        st[0] = s.x; st[1] = s.y; st[2] = s.th; st[3] = s.vx; st[4] = s.vy; st[5] = s.om;
        orbit[2 * i] = phi; orbit[2 * i + 1] = ecc;
    }
    return 0;
}

// tile at position `pos` of a free-tile multiset (sixteen 4-bit counters), as goal resampling looks it up; and its size
extern "C" uint32_t twin_free_at(uint64_t f, uint32_t pos) { return free_at(free_prefix(f), pos); }
extern "C" uint32_t twin_free_total(uint64_t f) { return free_total(f); }

extern "C" void twin_philox(uint32_t k0, uint32_t k1, const uint32_t *ctr, uint32_t *out) {
    uint32_t o[4];
    philox4x32_10(k0, k1, ctr[0], ctr[1], ctr[2], ctr[3], o);
    std::memcpy(out, o, sizeof(o));
}

// the integrator's short sine / cosine (single and two at a time) and the general one, for the accuracy test
extern "C" void twin_sincos(int which, int64_t n, const float *r, float *s, float *c) {
    for (int64_t i = 0; i < n; i++) {
        if (which == 0) sg::sincos_small(r[i], s[i], c[i]);
        else if (which == 1) {
            sg::f2 ss, cc;
            sg::sincos_small2(sg::mk2(r[i], -r[i]), ss, cc);
            s[i] = ss.x; c[i] = cc.y;
        } else sg::sincos_acc(r[i], s[i], c[i]);
    }
}
