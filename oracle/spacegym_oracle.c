/*
 * spacegym_oracle.c -- CPU restatement (fp64, scalar C) of the Space-Gym step path.
 *
 * TEST INFRASTRUCTURE ONLY: see spacegym_oracle.h.  Never linked into or loaded by the product.
 *
 * Pinned against tests/golden/step_*.npz (outputs of the unmodified reference captured in the
 * build container by tools/gen_golden.py); see tests/test_oracle_golden.py.
 *
 * Build with -ffp-contract=off: the reference is NumPy/Python, which never fuses a*b+c.
 */
#include "spacegym_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NEQ 6
static const double SGO_G = 6.6743e-11; /* helpers.py:19 */
#define TWO_PI 6.283185307179586     /* 2 * np.pi */

/* ======================================================================================
 * Parameters per registered id (gym_space/__init__.py:26-146, goal.py:18-72, kepler.py:189-231)
 * ====================================================================================== */

/* hexagonal_tiling.py:161-174 compute_tiling_rows_cols_a */
static void tiling_rows_cols_a(int min_tiles, double world_size, int *r_out, int *c_out, double *a_out) {
    double m = (double)min_tiles;
    double r_ = sqrt(72.0 * sqrt(3.0) * m - 6.0 * sqrt(3.0) + 12.0) / 12.0 - 1.0 / 4.0 + sqrt(3.0) / 12.0;
    int r = (int)ceil(r_), c;
    for (;;) {
        c = (int)floor(2.0 * sqrt(3.0) * r / 3.0 - 1.0 / 3.0 + sqrt(3.0) / 3.0);
        if (r * c >= min_tiles) break;
        r += 1;
    }
    *r_out = r;
    *c_out = c;
    *a_out = 2.0 * sqrt(3.0) * world_size / (3.0 * (2.0 * r + 1.0));
}

static void goal_params(int n_planets, sgo_params *p) {
    memset(p, 0, sizeof(*p));
    p->family = SGO_FAMILY_GOAL;
    p->n_planets = n_planets;
    p->step_size = 0.07;          /* goal.py:66 */
    p->world_size = 3.0;          /* goal.py:10 */
    p->max_abs_vel_angle = 6.0;   /* goal.py:67 */
    p->max_engine_force = 0.4;    /* __init__.py:38 */
    p->ship_mass = 1.0;           /* goal.py:46 */
    /* hexagonal_tiling.py:25-48 */
    int n_objects = n_planets + 2;
    int min_tiles = (n_planets == 2) ? n_objects : (int)ceil(n_objects / 0.6);
    tiling_rows_cols_a(min_tiles, p->world_size, &p->tiling_rows, &p->tiling_cols, &p->tiling_a);
    double hex_height = p->tiling_a * sqrt(3.0);
    double planets_radius = hex_height / 2.0;
    planets_radius *= 0.75; /* PLANET_TILE_RATIO */
    for (int i = 0; i < n_planets; i++) {
        p->planet_mass[i] = 1e9 / n_planets; /* goal.py:14,43 */
        p->planet_radius[i] = planets_radius;
    }
    p->goal_radius = planets_radius / 2.0; /* hexagonal_tiling.py:48, goal.py:88 */
    p->danger_zone = 0.25;                 /* goal.py:24 */
    p->survival_reward_scale = 0.2;        /* __init__.py:34-37 */
    p->goal_vel_reward_scale = 5.0;
    p->safety_reward_scale = 10.0;
    p->goal_sparse_reward = 5.0;
    p->distance_fctr = 100.0;              /* goal.py:16 */
    p->max_episode_steps = 500;            /* __init__.py:29 */
    p->moi = 0.01; p->max_thruster_force = 0.05; /* __init__.py:33, goal.py:46 */
}

static void kepler_params(double a, double ecc, double angle, int randomize, sgo_params *p) {
    memset(p, 0, sizeof(*p));
    p->family = SGO_FAMILY_KEPLER;
    p->n_planets = 2;            /* kepler.py:204-206: planet + zero-mass border */
    p->step_size = 0.07;         /* __init__.py:76 */
    p->world_size = 6.0;         /* kepler.py:216: 2 * _border_radius */
    p->max_abs_vel_angle = 6.0;
    p->max_engine_force = 0.4;
    p->ship_mass = 1.0;
    p->planet_mass[0] = 6e8;   p->planet_radius[0] = 0.2; /* kepler.py:17,204 */
    p->planet_mass[1] = 0.0;   p->planet_radius[1] = 3.0; /* kepler.py:18,206 */
    p->ref_orbit_a = a; p->ref_orbit_eccentricity = ecc; p->ref_orbit_angle = angle;
    p->numerator_C = 0.01; p->rad_penalty_C = 2.0; p->act_penalty_C = 0.5; /* __init__.py:86-88 */
    p->max_episode_steps = 500;
    p->randomize_orbit = randomize;
    p->moi = 0.01; p->max_thruster_force = 0.05; /* __init__.py:74, kepler.py:208 */
}

int sgo_params_for_id(const char *id, sgo_params *out) {
    if (!strcmp(id, "GoalContinuous2P-v0")) { goal_params(2, out); return 0; }
    if (!strcmp(id, "GoalContinuous3P-v0")) { goal_params(3, out); return 0; }
    if (!strcmp(id, "GoalContinuous4P-v0")) { goal_params(4, out); return 0; }
    if (!strcmp(id, "KeplerCircleOrbit-v0")) { kepler_params(1.2, 0.0, 0.0, 0, out); return 0; }
    if (!strcmp(id, "KeplerEllipseEasy-v0")) { kepler_params(1.2, 0.5, 0.8, 0, out); return 0; }
    if (!strcmp(id, "KeplerEllipseHard-v0")) { kepler_params(1.2, 0.725, 3.925, 0, out); return 0; }
    /* constructor defaults kepler.py:193-195 hold until the first reset draws (kepler.py:257-259) */
    if (!strcmp(id, "KeplerRandomOrbits-v0")) { kepler_params(1.2, 0.5, 3.75, 1, out); return 0; }
    /* discrete-action ids registered by keyboard_agent.py:10-74: max_engine_force = 1 for Goal; KeplerDiscrete-v0 has no
     * max_episode_steps, i.e. no TimeLimit */
    if (!strncmp(id, "GoalDiscrete", 12) && id[12] >= '2' && id[12] <= '4' && !strcmp(id + 13, "-v0")) {
        goal_params(id[12] - '0', out);
        out->max_engine_force = 1.0; out->discrete_actions = 1;
        return 0;
    }
    if (!strcmp(id, "KeplerDiscrete-v0")) {
        kepler_params(1.2, 0.0, 0.0, 0, out);
        out->discrete_actions = 1; out->max_episode_steps = 2147483647;
        return 0;
    }
    return -1;
}

int sgo_obs_dim(const sgo_params *p) {
    return p->family == SGO_FAMILY_GOAL ? 7 + 2 * p->n_planets + 2 : 10;
}

/* ======================================================================================
 * RHS: dynamic_model.py:129-176 + helpers.py:22-35
 * ====================================================================================== */
typedef struct {
    const sgo_params *p;
    const double *planets; /* [n][2] */
    double engine_force_scalar; /* float32(engine * max_engine_force) widened (numpy>=2, NEP 50) */
    double omega_cmd;           /* float32(thruster * 5.0) widened: dynamic_model.py:140 */
    double torque;              /* thruster * max_thruster_force: dynamic_model.py:175 */
    int nfev;
} rhs_ctx;

static double norm2(double x, double y) { return sqrt(x * x + y * y); } /* np.linalg.norm on a 2-vector */

/* ship_vector_field (dynamic_model.py:129-142): note it OVERWRITES y[5] in place after computing the
 * acceleration and before reading the velocities. */
static void rhs(rhs_ctx *c, double t, double *y, double *f) {
    (void)t;
    const sgo_params *p = c->p;
    c->nfev++;
    /* ship_external_force (dynamic_model.py:168-176) */
    double angle = y[2];
    double fx = -cos(angle) * c->engine_force_scalar;
    double fy = -sin(angle) * c->engine_force_scalar;
    /* ship_acceleration (dynamic_model.py:145-165) */
    for (int j = 0; j < p->n_planets; j++) {
        double dx = c->planets[2 * j] - y[0], dy = c->planets[2 * j + 1] - y[1]; /* helpers.py:31 */
        double d = norm2(dx, dy);
        double dirx = dx / d, diry = dy / d;
        double scalar = SGO_G * p->ship_mass * p->planet_mass[j] / (d * d); /* helpers.py:34 */
        fx += dirx * scalar;
        fy += diry * scalar;
    }
    double ax = fx / p->ship_mass, ay = fy / p->ship_mass;
    if (!p->steering_acceleration) y[5] = c->omega_cmd; /* Steering.velocity, dynamic_model.py:138-141 */
    f[0] = y[3]; f[1] = y[4]; f[2] = y[5];
    f[3] = ax;   f[4] = ay;
    f[5] = p->steering_acceleration ? c->torque / p->moi : 0.0; /* dynamic_model.py:160-163 */
}

/* ======================================================================================
 * Termination events: dynamic_model.py:183-217.  Order: planets..., world_max, world_min, ang_vel
 * ====================================================================================== */
static int n_events(const sgo_params *p) { return p->n_planets + 3; }

static double event_fn(const sgo_params *p, const double *planets, int k, const double *y) {
    int n = p->n_planets;
    if (k < n) return norm2(planets[2 * k] - y[0], planets[2 * k + 1] - y[1]) - p->planet_radius[k];
    double half = p->world_size / 2.0;
    if (k == n) return fmin(half - y[0], half - y[1]);
    if (k == n + 1) return fmin(half + y[0], half + y[1]);
    return p->max_abs_vel_angle - fabs(y[5]);
}

/* ======================================================================================
 * scipy RK45 (scipy/integrate/_ivp/rk.py): Dormand-Prince 5(4), rtol=1e-3, atol=1e-6 defaults
 * ====================================================================================== */
static const double RK_C[6] = {0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1};
static const double RK_A[6][5] = {
    {0, 0, 0, 0, 0},
    {1.0 / 5, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
static const double RK_B[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
static const double RK_E[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};
static const double RK_P[7][4] = {
    {1, -8048581381.0 / 2820520608, 8663915743.0 / 2820520608, -12715105075.0 / 11282082432},
    {0, 0, 0, 0},
    {0, 131558114200.0 / 32700410799, -68118460800.0 / 10900136933, 87487479700.0 / 32700410799},
    {0, -1754552775.0 / 470086768, 14199869525.0 / 1410260304, -10690763975.0 / 1880347072},
    {0, 127303824393.0 / 49829197408, -318862633887.0 / 49829197408, 701980252875.0 / 199316789632},
    {0, -282668133.0 / 205662961, 2019193451.0 / 616988883, -1453857185.0 / 822651844},
    {0, 40617522.0 / 29380423, -110615467.0 / 29380423, 69997945.0 / 29380423}};
#define RK_SAFETY 0.9
#define RK_MIN_FACTOR 0.2
#define RK_MAX_FACTOR 10.0
#define RK_RTOL 1e-3
#define RK_ATOL 1e-6

static double rms_norm(const double *x) { /* common.py norm(): ||x||_2 / sqrt(n) */
    double s = 0;
    for (int i = 0; i < NEQ; i++) s += x[i] * x[i];
    return sqrt(s) / sqrt((double)NEQ);
}

/* common.py select_initial_step (Hairer, Norsett, Wanner, Sec. II.4) */
static double select_initial_step(rhs_ctx *c, double t0, const double *y0, double t_bound, const double *f0) {
    double interval = fabs(t_bound - t0);
    if (interval == 0.0) return 0.0;
    double scale[NEQ], tmp[NEQ], y1[NEQ], f1[NEQ];
    for (int i = 0; i < NEQ; i++) scale[i] = RK_ATOL + fabs(y0[i]) * RK_RTOL;
    for (int i = 0; i < NEQ; i++) tmp[i] = y0[i] / scale[i];
    double d0 = rms_norm(tmp);
    for (int i = 0; i < NEQ; i++) tmp[i] = f0[i] / scale[i];
    double d1 = rms_norm(tmp);
    double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    h0 = fmin(h0, interval);
    for (int i = 0; i < NEQ; i++) y1[i] = y0[i] + h0 * 1.0 * f0[i];
    rhs(c, t0 + h0 * 1.0, y1, f1);
    for (int i = 0; i < NEQ; i++) tmp[i] = (f1[i] - f0[i]) / scale[i];
    double d2 = rms_norm(tmp) / h0;
    double h1;
    if (d1 <= 1e-15 && d2 <= 1e-15) h1 = fmax(1e-6, h0 * 1e-3);
    else h1 = pow(0.01 / fmax(d1, d2), 1.0 / (4 + 1)); /* error_estimator_order = 4 */
    return fmin(fmin(100 * h0, h1), interval); /* max_step = inf */
}

typedef struct {
    double t_old, t, h;
    double y_old[NEQ];
    double Q[NEQ][4];
} dense_out;

static void dense_eval(const dense_out *d, double t, double *y) { /* rk.py RkDenseOutput._call_impl */
    double x = (t - d->t_old) / d->h;
    double pw[4];
    pw[0] = x;
    for (int j = 1; j < 4; j++) pw[j] = pw[j - 1] * x; /* np.cumprod */
    for (int i = 0; i < NEQ; i++) {
        double s = 0;
        for (int j = 0; j < 4; j++) s += d->Q[i][j] * pw[j];
        y[i] = d->h * s + d->y_old[i];
    }
}

typedef struct {
    const sgo_params *p;
    const double *planets;
    const dense_out *sol;
    int k;
} event_eq;

static double event_eq_eval(const event_eq *e, double t) {
    double y[NEQ];
    dense_eval(e->sol, t, y);
    return event_fn(e->p, e->planets, e->k, y);
}

/* scipy.optimize.brentq (Brent 1973, as in scipy/optimize/Zeros/brentq.c), xtol = rtol = 4*EPS
 * (ivp.py:51-76 solve_event_equation). */
static double brentq(const event_eq *e, double xa, double xb) {
    const double xtol = 4 * 2.220446049250313e-16, rtol = 4 * 2.220446049250313e-16;
    double xpre = xa, xcur = xb, xblk = 0, fpre, fcur, fblk = 0, spre = 0, scur = 0, sbis, delta, stry, dpre, dblk;
    fpre = event_eq_eval(e, xpre);
    fcur = event_eq_eval(e, xcur);
    if (fpre == 0) return xpre;
    if (fcur == 0) return xcur;
    if ((fpre < 0) == (fcur < 0)) return xcur; /* scipy raises ValueError here; unreachable for a detected sign change */
    for (int i = 0; i < 100; i++) {
        if (fpre != 0 && fcur != 0 && ((fpre < 0) != (fcur < 0))) {
            xblk = xpre; fblk = fpre;
            spre = scur = xcur - xpre;
        }
        if (fabs(fblk) < fabs(fcur)) {
            xpre = xcur; xcur = xblk; xblk = xpre;
            fpre = fcur; fcur = fblk; fblk = fpre;
        }
        delta = (xtol + rtol * fabs(xcur)) / 2;
        sbis = (xblk - xcur) / 2;
        if (fcur == 0 || fabs(sbis) < delta) return xcur;
        if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
            if (xpre == xblk) {
                stry = -fcur * (xcur - xpre) / (fcur - fpre); /* secant */
            } else {
                dpre = (fpre - fcur) / (xpre - xcur); /* inverse quadratic */
                dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
            }
            if (2 * fabs(stry) < fmin(fabs(spre), 3 * fabs(sbis) - delta)) { spre = scur; scur = stry; }
            else { spre = sbis; scur = sbis; }
        } else { spre = sbis; scur = sbis; }
        xpre = xcur; fpre = fcur;
        if (fabs(scur) > delta) xcur += scur;
        else xcur += (sbis > 0 ? delta : -delta);
        fcur = event_eq_eval(e, xcur);
    }
    return xcur;
}

/* dynamic_model.py:94-125 make_step = solve_ivp(RK45, (0, h), y0, events) + wrap angle */
int sgo_make_step(const sgo_params *p, const double *planets, double *state, const float *action, sgo_diag *diag) {
    /* dynamic_model.py:170-171: engine_action * max_engine_force in float32 (numpy>=2 keeps f32 * pyfloat in f32) */
    return sgo_make_step_forces(p, planets, state, (double)(float)(action[0] * (float)p->max_engine_force),
                                (double)(float)(action[1] * 5.0f) /* dynamic_model.py:140 */, diag);
}

int sgo_make_step_forces(const sgo_params *p, const double *planets, double *state, double engine_force_scalar,
                         double omega_cmd, sgo_diag *diag) {
    return sgo_make_step_full(p, planets, state, engine_force_scalar, omega_cmd, 0.0, diag);
}

int sgo_make_step_full(const sgo_params *p, const double *planets, double *state, double engine_force_scalar,
                       double omega_cmd, double torque, sgo_diag *diag) {
    rhs_ctx c;
    c.p = p; c.planets = planets; c.nfev = 0;
    c.engine_force_scalar = engine_force_scalar;
    c.omega_cmd = omega_cmd;
    c.torque = torque;

    const double t0 = 0.0, t_bound = p->step_size;
    double t = t0, y[NEQ], f[NEQ];
    memcpy(y, state, sizeof(y));
    /* RungeKutta.__init__ (rk.py:85-105) */
    rhs(&c, t, y, f); /* mutates y[5] */
    double h_abs = select_initial_step(&c, t, y, t_bound, f);
    /* solve_ivp (ivp.py:636-646): g at (t0, y0) AFTER the solver was constructed */
    int ne = n_events(p);
    double g[SGO_MAX_PLANETS + 3], g_new[SGO_MAX_PLANETS + 3];
    for (int k = 0; k < ne; k++) g[k] = event_fn(p, planets, k, y);

    int status = -2, n_steps = 0, ev_index = -1;
    double t_event = NAN;
    double K[7][NEQ];
    while (status == -2) {
        /* OdeSolver.step (base.py) + RungeKutta._step_impl (rk.py:111-176) */
        double t_old = t, y_old[NEQ], h = 0, y_new[NEQ], f_new[NEQ], t_new = t;
        memcpy(y_old, y, sizeof(y));
        if (t == t_bound) {
            status = 0; /* corner case in OdeSolver.step; unreachable with step_size > 0 */
        } else {
            double min_step = 10 * fabs(nextafter(t, INFINITY) - t);
            if (h_abs < min_step) h_abs = min_step;
            int accepted = 0, rejected = 0;
            while (!accepted) {
                if (h_abs < min_step) { status = -1; break; }
                h = h_abs;
                t_new = t + h;
                if (t_new - t_bound > 0) t_new = t_bound;
                h = t_new - t;
                h_abs = fabs(h);
                /* rk_step (rk.py:14-71) */
                memcpy(K[0], f, sizeof(f));
                for (int s = 1; s < 6; s++) {
                    double ys[NEQ];
                    for (int i = 0; i < NEQ; i++) {
                        double dy = 0;
                        for (int j = 0; j < s; j++) dy += K[j][i] * RK_A[s][j];
                        ys[i] = y[i] + dy * h;
                    }
                    rhs(&c, t + RK_C[s] * h, ys, K[s]);
                }
                for (int i = 0; i < NEQ; i++) {
                    double s_ = 0;
                    for (int j = 0; j < 6; j++) s_ += K[j][i] * RK_B[j];
                    y_new[i] = y[i] + h * s_;
                }
                rhs(&c, t + h, y_new, f_new);
                memcpy(K[6], f_new, sizeof(f_new));
                double err[NEQ];
                for (int i = 0; i < NEQ; i++) {
                    double scale = RK_ATOL + fmax(fabs(y[i]), fabs(y_new[i])) * RK_RTOL;
                    double e_ = 0;
                    for (int j = 0; j < 7; j++) e_ += K[j][i] * RK_E[j];
                    err[i] = e_ * h / scale;
                }
                double error_norm = rms_norm(err);
                if (error_norm < 1) {
                    double factor = (error_norm == 0) ? RK_MAX_FACTOR
                                                      : fmin(RK_MAX_FACTOR, RK_SAFETY * pow(error_norm, -0.2));
                    if (rejected) factor = fmin(1.0, factor);
                    h_abs *= factor;
                    accepted = 1;
                } else {
                    h_abs *= fmax(RK_MIN_FACTOR, RK_SAFETY * pow(error_norm, -0.2));
                    rejected = 1;
                }
            }
            if (status == -1) break;
            n_steps++;
            t = t_new;
            memcpy(y, y_new, sizeof(y));
            memcpy(f, f_new, sizeof(f));
            if (t - t_bound >= 0) status = 0; /* 'finished' */
        }
        /* events (ivp.py:673-694) */
        for (int k = 0; k < ne; k++) g_new[k] = event_fn(p, planets, k, y);
        int active[SGO_MAX_PLANETS + 3], n_active = 0;
        for (int k = 0; k < ne; k++) { /* find_active_events, direction == 0 */
            int up = (g[k] <= 0) && (g_new[k] >= 0), down = (g[k] >= 0) && (g_new[k] <= 0);
            if (up || down) active[n_active++] = k;
        }
        if (n_active > 0) {
            dense_out sol; /* _dense_output_impl: Q = K.T.dot(P) */
            sol.t_old = t_old; sol.t = t; sol.h = t - t_old;
            memcpy(sol.y_old, y_old, sizeof(y_old));
            for (int i = 0; i < NEQ; i++)
                for (int j = 0; j < 4; j++) {
                    double s_ = 0;
                    for (int s = 0; s < 7; s++) s_ += K[s][i] * RK_P[s][j];
                    sol.Q[i][j] = s_;
                }
            /* handle_events: all events are terminal -> earliest root wins (stable order on ties) */
            double best = INFINITY;
            for (int a = 0; a < n_active; a++) {
                event_eq e = {p, planets, &sol, active[a]};
                double root = brentq(&e, t_old, t);
                if (root < best) { best = root; ev_index = active[a]; }
            }
            status = 1;
            t = best; t_event = best;
            dense_eval(&sol, t, y);
        }
        memcpy(g, g_new, sizeof(g));
    }
    /* dynamic_model.py:121-124 */
    memcpy(state, y, sizeof(y));
    double th = fmod(state[2], TWO_PI); /* wrap_ship_angle: Python float % */
    if (th != 0 && th < 0) th += TWO_PI;
    state[2] = th;
    if (diag) { diag->n_rk_steps = n_steps; diag->nfev = c.nfev; diag->event_index = ev_index; diag->t_event = t_event; }
    return status == 1;
}

/* ======================================================================================
 * Observation: spaceship_env.py:113-140, kepler.py:172-187
 * ====================================================================================== */
static void lidar(const sgo_params *p, const double *ship_xy, const double *obj_xy, double obj_radius, double *out) {
    double vx = obj_xy[0] - ship_xy[0], vy = obj_xy[1] - ship_xy[1];
    double ang = atan2(vy, vx);            /* helpers.py:8-9 */
    ang = fmod(ang, TWO_PI);               /* spaceship_env.py:138 */
    if (ang != 0 && ang < 0) ang += TWO_PI;
    double scale = (norm2(vx, vy) - obj_radius) * 2 / p->world_size;
    out[0] = cos(ang) * scale;
    out[1] = sin(ang) * scale;
}

void sgo_make_observation(const sgo_params *p, const double *s, const double *planets, const double *goal, double *obs) {
    int k = 0;
    obs[k++] = s[0]; obs[k++] = s[1];
    obs[k++] = cos(s[2]); obs[k++] = sin(s[2]);
    obs[k++] = s[3]; obs[k++] = s[4]; obs[k++] = s[5];
    if (p->family == SGO_FAMILY_GOAL) { /* with_lidar and with_goal (goal.py:69-70) */
        for (int j = 0; j < p->n_planets; j++, k += 2) lidar(p, s, planets + 2 * j, p->planet_radius[j], obs + k);
        lidar(p, s, goal, 0.0, obs + k);
    } else { /* kepler.py:172-187; goal slots hold the per-env orbit when RandomOrbits */
        obs[k++] = goal ? goal[0] : p->ref_orbit_angle;
        obs[k++] = goal ? goal[1] : p->ref_orbit_eccentricity;
        obs[k++] = goal ? goal[2] : p->ref_orbit_a;
    }
}

/* ======================================================================================
 * Goal reward: goal.py:147-158 (_reward), :160-164 (_goal_vel_reward2), :204-227 (_safety_reward_simple2)
 * ====================================================================================== */
double sgo_goal_reward(const sgo_params *p, const double *s1, const double *last_xy, const double *planets,
                       const double *goal, int *hit) {
    double current_dist = norm2(goal[0] - s1[0], goal[1] - s1[1]);
    double last_dist = norm2(goal[0] - last_xy[0], goal[1] - last_xy[1]);
    double goal_vel = (last_dist - current_dist) * p->distance_fctr;
    /* _safety_reward_simple2 */
    double sum_safety = 0, mindist = INFINITY;
    int closest = -1;
    for (int j = 0; j < p->n_planets; j++) {
        double ddx = s1[0] - planets[2 * j], ddy = s1[1] - planets[2 * j + 1];
        double dist = sqrt(ddx * ddx + ddy * ddy);
        if (dist < mindist) { closest = j; mindist = dist; }
    }
    double r = p->planet_radius[closest];
    if ((mindist - r) < p->danger_zone) {
        double px = last_xy[0] - planets[2 * closest], py = last_xy[1] - planets[2 * closest + 1];
        double prev_dist = sqrt(px * px + py * py);
        if (prev_dist > mindist) sum_safety -= p->distance_fctr * (prev_dist - mindist);
    }
    double reward = p->survival_reward_scale + p->goal_vel_reward_scale * goal_vel + p->safety_reward_scale * sum_safety;
    *hit = 0;
    if (norm2(goal[0] - s1[0], goal[1] - s1[1]) < p->goal_radius) { /* goal.py:154-157 */
        reward += p->goal_sparse_reward;
        *hit = 1;
    }
    return reward;
}

/* ======================================================================================
 * Kepler reward: kepler.py:43-156
 * ====================================================================================== */
static void rotate(const double *xy, double alpha, double *out) { /* kepler.py:51-58 */
    double c = cos(alpha), s = sin(alpha);
    out[0] = c * xy[0] + s * xy[1];
    out[1] = -s * xy[0] + c * xy[1];
}

double sgo_kepler_reward(const sgo_params *p, const double *s1, const float *action, double a, double ecc,
                         double ref_angle) {
    /* np.linalg.norm(last_action) on a float32 array stays float32; act_penalty_C * f32 stays f32 */
    float act_penalty = sqrtf(action[0] * action[0] + action[1] * action[1]);
    float act_term = (float)p->act_penalty_C * act_penalty;
    return sgo_kepler_reward_act(p, s1, (double)act_term, a, ecc, ref_angle);
}

double sgo_kepler_reward_act(const sgo_params *p, const double *s1, double act_term, double a, double ecc,
                             double ref_angle) {
    const double *pos = s1, *vel = s1 + 3;
    double b = sqrt(a * a * (1 - ecc * ecc)); /* _b */
    double c = sqrt(a * a - b * b);           /* _c */
    /* _orbit_cur_rad (kepler.py:90-96) */
    double w[2];
    rotate(pos, ref_angle, w);
    w[0] = w[0] - c;
    double cur_rad = norm2(w[0], w[1]);
    /* _orbit_target_vel (kepler.py:64-88) */
    double theta = atan2(w[1], w[0]);
    double ct = ecc * cos(theta);
    double target_rad = b / sqrt(1 - ct * ct);
    double nw = norm2(w[0], w[1]);
    double pw[2] = {w[0] * target_rad / nw, w[1] * target_rad / nw};
    double Vt[2] = {-1.0 * a / b * pw[1], 1.0 * b / a * pw[0]};
    double r = norm2(pw[0] + c, pw[1] + 0.0);
    double alpha = SGO_G * p->planet_mass[0];
    double orbit_vel = sqrt(alpha * (2 / r - 1 / a)); /* _orbit_vel */
    double nv = norm2(Vt[0], Vt[1]);
    Vt[0] = Vt[0] * orbit_vel / nv; Vt[1] = Vt[1] * orbit_vel / nv;
    double V[2];
    rotate(Vt, -ref_angle, V);
    /* _dense_reward5 (kepler.py:111-150) */
    double rad_penalty = fabs(cur_rad - target_rad);
    double vel_x_penalty = fabs(V[0] - vel[0]);
    double vel_y_penalty = fabs(V[1] - vel[1]);
    double C = p->numerator_C;
    return C / (p->rad_penalty_C * rad_penalty + vel_x_penalty + vel_y_penalty + act_term + C);
}

/* ======================================================================================
 * One env.step(): spaceship_env.py:68-78
 * ====================================================================================== */
static void translate_raw(const sgo_params *p, const void *raw_action, double *efs_o, double *omega_o, double *act_term_o,
                          double *torque_o);

void sgo_vector_field(const sgo_params *p, const double *planets, const double *state, const void *raw_action, double *field) {
    double efs, omega, act_term, torque, y[NEQ];
    translate_raw(p, raw_action, &efs, &omega, &act_term, &torque);
    static const double origin[2 * SGO_MAX_PLANETS] = {0};
    rhs_ctx c;
    c.p = p; c.planets = (p->family == SGO_FAMILY_GOAL) ? planets : origin; c.nfev = 0;
    c.engine_force_scalar = efs; c.omega_cmd = omega; c.torque = torque;
    memcpy(y, state, sizeof(y));
    rhs(&c, 0.0, y, field);
}

void sgo_env_step(const sgo_params *p, const double *planets, const double *goal, double *state,
                  const void *raw_action, double *obs, double *reward, uint8_t *done, uint8_t *goal_hit, sgo_diag *diag) {
    double efs, omega, act_term, torque;
    translate_raw(p, raw_action, &efs, &omega, &act_term, &torque);
    double last_xy[2] = {state[0], state[1]}; /* spaceship_env.py:74 */
    static const double origin[2 * SGO_MAX_PLANETS] = {0};
    const double *pl = (p->family == SGO_FAMILY_GOAL) ? planets : origin; /* kepler.py:204-206: both at (0,0) */
    *done = (uint8_t)sgo_make_step_full(p, pl, state, efs, omega, torque, diag);
    int hit = 0;
    if (p->family == SGO_FAMILY_GOAL) {
        sgo_make_observation(p, state, pl, goal, obs);
        *reward = sgo_goal_reward(p, state, last_xy, pl, goal, &hit);
    } else {
        /* goal (if given) carries the per-env orbit [angle, ecc, a] */
        double ang = goal ? goal[0] : p->ref_orbit_angle, ecc = goal ? goal[1] : p->ref_orbit_eccentricity;
        double a = goal ? goal[2] : p->ref_orbit_a;
        sgo_make_observation(p, state, pl, goal, obs);
        *reward = sgo_kepler_reward_act(p, state, act_term, a, ecc, ang);
    }
    *goal_hit = (uint8_t)hit;
}

static void translate_raw(const sgo_params *p, const void *raw_action, double *efs_o, double *omega_o, double *act_term_o,
                          double *torque_o) {
    double efs, omega, act_term, torque;
    if (p->discrete_actions) {
        /* DiscreteSpaceshipEnv._translate_raw_action (spaceship_env.py:189-202): python floats -> float64 arithmetic */
        static const double table[6][2] = {{0, 0}, {1, 0}, {0, -1}, {0, 1}, {1, -1}, {1, 1}};
        int k = *(const int32_t *)raw_action;
        if (k < 0 || k > 5) k = 0; /* the reference raises ValueError */
        const double engine = table[k][0], thruster = table[k][1];
        efs = engine * p->max_engine_force;   /* dynamic_model.py:171 */
        omega = thruster * 5.0;               /* dynamic_model.py:140 */
        torque = thruster * p->max_thruster_force; /* dynamic_model.py:175 */
        act_term = p->act_penalty_C * sqrt(engine * engine + thruster * thruster); /* kepler.py:138,143 in float64 */
    } else {
        /* ContinuousSpaceshipEnv._translate_raw_action (spaceship_env.py:210-214), float32 arithmetic */
        const float *raw = (const float *)raw_action;
        float action[2] = {(raw[0] + 1.0f) / 2.0f, raw[1]};
        efs = (double)(float)(action[0] * (float)p->max_engine_force);
        omega = (double)(float)(action[1] * 5.0f);
        torque = (double)(float)(action[1] * (float)p->max_thruster_force); /* float32 product, like engine_force_scalar */
        act_term = (double)((float)p->act_penalty_C * sqrtf(action[0] * action[0] + action[1] * action[1]));
    }
    *efs_o = efs; *omega_o = omega; *act_term_o = act_term; *torque_o = torque;
}

void sgo_env_step_batch(const sgo_params *p, int64_t m, const double *planets, const double *goal, double *state,
                        const void *raw, double *obs, double *reward, uint8_t *done, uint8_t *goal_hit,
                        sgo_diag *diag, int threads) {
    const int D = sgo_obs_dim(p), n = p->n_planets;
    const int gstride = (p->family == SGO_FAMILY_GOAL) ? 2 : 3;
    const size_t astride = p->discrete_actions ? sizeof(int32_t) : 2 * sizeof(float);
    (void)threads;
#pragma omp parallel for schedule(static) num_threads(threads > 1 ? threads : 1)
    for (int64_t i = 0; i < m; i++)
        sgo_env_step(p, planets ? planets + i * 2 * n : NULL, goal ? goal + i * gstride : NULL, state + i * NEQ,
                     (const char *)raw + i * astride, obs + i * D, reward + i, done + i, goal_hit + i, diag ? diag + i : NULL);
}

/* ======================================================================================
 * Counter-based RNG shared with the HIP engine
 * ====================================================================================== */
void sgo_philox4x32_10(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]) {
    /* Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3" (SC'11) */
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

#define SGO_STREAM_RESET 0u
#define SGO_STREAM_GOAL 1u
static void stream_words(uint64_t seed, uint32_t env_id, uint32_t episode, uint32_t block, uint32_t stream, uint32_t out[4]) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, ctr[4] = {env_id, episode, block, stream};
    sgo_philox4x32_10(key, ctr, out);
}
/* uniform in (0,1): exact in fp32, identical on host and device */
static float u23(uint32_t w) { return ((float)(w >> 9) + 0.5f) * (1.0f / 8388608.0f); }
static float u16(uint32_t h) { return ((float)(h & 0xffffu) + 0.5f) * (1.0f / 65536.0f); }
static uint32_t below16(uint32_t h, uint32_t n) { return ((h & 0xffffu) * n) >> 16; }

/* ----- hexagonal tiling geometry (hexagonal_tiling.py:33-48,136-158) */
typedef struct { int case_b, flip; double col_shift[4]; } tiling_layout;

static void tile_center(const sgo_params *p, const tiling_layout *L, int tile, double *xy) {
    double a = p->tiling_a, hex_h = a * sqrt(3.0), hex_w = 2 * a, W = p->world_size;
    int row = tile / p->tiling_cols, col = tile % p->tiling_cols;
    double x0 = -W / 2 + hex_w / 2, y0 = W / 2 - hex_h / 2;
    if (L->case_b) y0 -= hex_h / 2;
    double x_shift = col * 1.5 * a + L->col_shift[col];
    double y_rows = -row * hex_h, y_cols = -(col % 2) * hex_h / 2;
    if (L->case_b) y_cols *= -1;
    double x = x0 + x_shift, y = y0 + y_rows + y_cols;
    if (L->flip) { xy[0] = y; xy[1] = x; } else { xy[0] = x; xy[1] = y; }
}

/* helpers.py:48-53 uniform_disk_distribution + hexagonal_tiling.py:130-134: one word = (angle:hi16, radius:lo16) */
static void disc_in_tile(const sgo_params *p, const tiling_layout *L, int tile, double obj_radius, uint32_t w, double *xy) {
    double hex_h = p->tiling_a * sqrt(3.0), noise_radius = hex_h / 2 - obj_radius;
    double angle = TWO_PI * (double)u16(w >> 16), r = sqrt((double)u16(w) * noise_radius * noise_radius);
    tile_center(p, L, tile, xy);
    xy[0] += r * cos(angle);
    xy[1] += r * sin(angle);
}

/* Reset words (RNG): word k of an episode = component k%4 of Philox block k/4 of the reset stream.
 *   Goal   block 0: flags, col01, col23, tiles01      block 1: tiles23, tiles45, goal_c01, goal_c2
 *          block 2: disc_ship, disc_p0, disc_p1, disc_p2   block 3: disc_p3, disc_goal, theta, -
 *          block 4: bm1_u1, bm1_u2, bm2_u1, bm2_u2
 *   Kepler block 0: angle, dist, theta, ecc   block 1: orbit_angle, -, -, -   block 2: bm1_u1, bm1_u2, bm2_u1, bm2_u2 */
static void reset_words(uint64_t seed, uint32_t env_id, uint32_t episode, int n_blocks, uint32_t *w) {
    for (int b = 0; b < n_blocks; b++) stream_words(seed, env_id, episode, (uint32_t)b, SGO_STREAM_RESET, w + 4 * b);
}

/* The engine stores the per-episode layout; the oracle regenerates it from the reset stream (same words). */
static void episode_layout(const sgo_params *p, const uint32_t *w, tiling_layout *L) {
    uint32_t f = w[0];
    L->case_b = f & 1u; L->flip = (f >> 1) & 1u;                 /* hexagonal_tiling.py:69 */
    uint32_t c01 = w[1], c23 = w[2];
    double u[4] = {u16(c01 >> 16), u16(c01), u16(c23 >> 16), u16(c23)}, cum = 0;
    int cols = p->tiling_cols;
    double tiling_width = 3 * p->tiling_a * (cols - 1) / 2 + 2 * p->tiling_a; /* hexagonal_tiling.py:39 */
    double free_x = p->world_size - tiling_width;
    for (int j = 0; j < 4; j++) L->col_shift[j] = 0;
    for (int j = 0; j < cols; j++) { cum += u[j]; L->col_shift[j] = cum; }   /* :70 cumsum */
    for (int j = 0; j < cols; j++) L->col_shift[j] *= free_x / cum;            /* :71-72 */
}

static void free_insert_sorted(sgo_env_state *e, int tile) {
    /* hexagonal_tiling.py:104 appends; the engine keeps the multiset sorted (distribution-equivalent: candidates are
     * drawn uniformly without replacement by POSITION and ties go to the first DRAWN, so order is immaterial). */
    if (e->n_free >= 63) return;
    int i = e->n_free++;
    while (i > 0 && e->free_tiles[i - 1] > tile) { e->free_tiles[i] = e->free_tiles[i - 1]; i--; }
    e->free_tiles[i] = tile;
}

/* hexagonal_tiling.py:99-128 _reset_goal_tile_nr + :95-97 find_new_goal.  words: [gate|.., cand0:cand1, cand2:.., disc] */
static void choose_goal(const sgo_params *p, const tiling_layout *L, sgo_env_state *e, int first, const uint32_t w[4]) {
    if (!first) { /* :101-106 */
        free_insert_sorted(e, e->ship_tile);
        e->ship_tile = e->goal_tile;
    }
    if ((w[0] & 0xffu) < 64u) { /* uniform() < 0.25, :108-110 */
        e->goal_tile = e->ship_tile;
    } else {
        int n = e->n_free, n_cand = n < 3 ? n : 3, idx[64];
        uint32_t draws[3] = {w[1] >> 16, w[1], w[2] >> 16};
        for (int i = 0; i < n; i++) idx[i] = i;
        int best = -1, best_dist = -1;
        int cols = p->tiling_cols, sr = e->ship_tile / cols, sc = e->ship_tile % cols;
        for (int i = 0; i < n_cand; i++) { /* choice(len(free), size=n_cand, replace=False): partial Fisher-Yates */
            int j = i + (int)below16(draws[i], (uint32_t)(n - i));
            int tmp = idx[i]; idx[i] = idx[j]; idx[j] = tmp;
            int tile = e->free_tiles[idx[i]];
            int dist = abs(tile / cols - sr) + abs(tile % cols - sc); /* :119-121 */
            if (dist > best_dist) { best_dist = dist; best = idx[i]; }   /* first max wins, :122-124 */
        }
        e->goal_tile = e->free_tiles[best];
        for (int i = best; i + 1 < e->n_free; i++) e->free_tiles[i] = e->free_tiles[i + 1]; /* pop, :126 */
        e->n_free--;
    }
    disc_in_tile(p, L, e->goal_tile, p->goal_radius, w[3], e->goal_xy);
}

static void box_muller(uint32_t w1, uint32_t w2, double *z0, double *z1) {
    double r = sqrt(-2.0 * log((double)u23(w1))), a = TWO_PI * (double)u23(w2);
    *z0 = r * cos(a);
    *z1 = r * sin(a);
}

void sgo_env_reset(const sgo_params *p, uint64_t seed, uint32_t env_id, sgo_env_state *e) {
    uint32_t w[20];
    e->elapsed = 0;
    e->goal_draws = 0;
    if (p->family == SGO_FAMILY_GOAL) {
        tiling_layout L;
        reset_words(seed, env_id, e->episode, 5, w);
        const uint32_t flags = w[0];
        episode_layout(p, w, &L);
        int N = p->n_planets, T = p->tiling_rows * p->tiling_cols, tiles[SGO_MAX_PLANETS + 1];
        uint32_t t01 = w[3], t23 = w[4], t45 = w[5];
        uint32_t draws[6] = {t01 >> 16, t01, t23 >> 16, t23, t45 >> 16, t45};
        if (N == 2 && ((flags >> 8) & 0xffu) < 64u) { /* hexagonal_tiling.py:75-87 */
            static const int diag[4][3] = {{1, 0, 3}, {2, 0, 3}, {0, 1, 2}, {3, 1, 2}};
            memcpy(tiles, diag[(flags >> 16) & 3u], sizeof(diag[0]));
        } else { /* :89 choice(n_tiles, size=n_objects-1, replace=False) */
            int perm[16];
            for (int i = 0; i < T; i++) perm[i] = i;
            for (int i = 0; i <= N; i++) {
                int j = i + (int)below16(draws[i], (uint32_t)(T - i));
                int tmp = perm[i]; perm[i] = perm[j]; perm[j] = tmp;
                tiles[i] = perm[i];
            }
        }
        e->ship_tile = tiles[0]; /* :90 */
        e->n_free = 0;           /* :91 */
        for (int t = 0; t < T; t++) {
            int used = 0;
            for (int i = 0; i <= N; i++) used |= (tiles[i] == t);
            if (!used) e->free_tiles[e->n_free++] = t;
        }
        double ship_radius = p->planet_radius[0] / 2; /* hexagonal_tiling.py:48 */
        disc_in_tile(p, &L, tiles[0], ship_radius, w[8], e->state); /* :92-93 */
        for (int j = 0; j < N; j++) disc_in_tile(p, &L, tiles[j + 1], p->planet_radius[j], w[9 + j], e->planets_xy + 2 * j);
        uint32_t gw[4];
        gw[0] = flags >> 24; gw[1] = w[6]; gw[2] = w[7]; gw[3] = w[13];
        e->goal_tile = -1;
        choose_goal(p, &L, e, 1, gw); /* goal.py:138 */
        /* goal.py:140-145 */
        e->state[2] = TWO_PI * (double)u23(w[14]);
        double z0, z1, z2, z3;
        box_muller(w[16], w[17], &z0, &z1);
        box_muller(w[18], w[19], &z2, &z3);
        e->state[3] = z0 * 0.07; e->state[4] = z1 * 0.07;
        double max_w = 0.7 * p->max_abs_vel_angle, om = z2 * max_w / 3;
        e->state[5] = fmin(fmax(om, -max_w), max_w);
        e->orbit[0] = e->orbit[1] = e->orbit[2] = 0;
    } else { /* kepler.py:233-267 */
        reset_words(seed, env_id, e->episode, 3, w);
        double planet_angle = TWO_PI * (double)u23(w[0]);
        double lo = p->planet_radius[0] + 0.5, hi = p->planet_radius[1] - 0.5;
        double dist = lo + (hi - lo) * (double)u23(w[1]);
        e->state[0] = cos(planet_angle) * dist; e->state[1] = sin(planet_angle) * dist;
        e->state[2] = TWO_PI * (double)u23(w[2]);
        uint32_t we = w[3], wa = w[4];
        if (p->randomize_orbit) { /* kepler.py:257-259 (global np.random in the reference) */
            e->orbit[1] = (double)u23(we) * 0.7;
            e->orbit[0] = (double)u23(wa) * 2 * 3.141592653589793;
        } else {
            e->orbit[0] = p->ref_orbit_angle; e->orbit[1] = p->ref_orbit_eccentricity;
        }
        e->orbit[2] = p->ref_orbit_a;
        double z0, z1, z2, z3;
        box_muller(w[8], w[9], &z0, &z1);
        box_muller(w[10], w[11], &z2, &z3);
        e->state[3] = z0 * 0.05; e->state[4] = z1 * 0.05;
        double max_w = 0.7 * p->max_abs_vel_angle, om = z2 * max_w / 5;
        e->state[5] = fmin(fmax(om, -max_w), max_w);
        memset(e->planets_xy, 0, sizeof(e->planets_xy));
        e->goal_xy[0] = e->goal_xy[1] = 0;
        e->n_free = 0; e->ship_tile = e->goal_tile = -1;
    }
}

void sgo_env_resample_goal(const sgo_params *p, uint64_t seed, uint32_t env_id, sgo_env_state *e) {
    tiling_layout L;
    uint32_t w[4], w0[4];
    stream_words(seed, env_id, e->episode, 0, SGO_STREAM_RESET, w0);
    episode_layout(p, w0, &L);
    e->goal_draws += 1;
    stream_words(seed, env_id, e->episode, e->goal_draws, SGO_STREAM_GOAL, w);
    choose_goal(p, &L, e, 0, w);
}

static void env_observe(const sgo_params *p, const sgo_env_state *e, double *obs) {
    sgo_make_observation(p, e->state, e->planets_xy, p->family == SGO_FAMILY_GOAL ? e->goal_xy : e->orbit, obs);
}

void sgo_vec_reset(const sgo_params *p, uint64_t seed, int64_t b, uint32_t env_id0, sgo_env_state *envs, double *obs,
                   int threads) {
    const int D = sgo_obs_dim(p);
    (void)threads;
#pragma omp parallel for schedule(static) num_threads(threads > 1 ? threads : 1)
    for (int64_t i = 0; i < b; i++) {
        envs[i].episode = 0;
        sgo_env_reset(p, seed, env_id0 + (uint32_t)i, &envs[i]);
        if (obs) env_observe(p, &envs[i], obs + i * D);
    }
}

void sgo_vec_step(const sgo_params *p, uint64_t seed, int64_t b, uint32_t env_id0, sgo_env_state *envs,
                  const void *raw_actions, double *obs, double *reward, uint8_t *done, uint8_t *truncated,
                  double *terminal_obs, int threads) {
    const int D = sgo_obs_dim(p);
    const size_t astride = p->discrete_actions ? sizeof(int32_t) : 2 * sizeof(float);
    (void)threads;
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads > 1 ? threads : 1)
    for (int64_t i = 0; i < b; i++) {
        sgo_env_state *e = &envs[i];
        uint8_t dn, hit;
        sgo_env_step(p, e->planets_xy, p->family == SGO_FAMILY_GOAL ? e->goal_xy : e->orbit, e->state,
                     (const char *)raw_actions + i * astride, obs + i * D, reward + i, &dn, &hit, NULL);
        if (hit) sgo_env_resample_goal(p, seed, env_id0 + (uint32_t)i, e); /* goal.py:157 */
        e->elapsed += 1;
        int trunc = !dn && e->elapsed >= p->max_episode_steps; /* gym TimeLimit */
        done[i] = (uint8_t)(dn || trunc);
        truncated[i] = (uint8_t)trunc;
        if (done[i]) {
            if (terminal_obs) memcpy(terminal_obs + i * D, obs + i * D, sizeof(double) * D);
            e->episode += 1;
            sgo_env_reset(p, seed, env_id0 + (uint32_t)i, e);
            env_observe(p, e, obs + i * D);
        }
    }
}
