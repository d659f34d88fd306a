/*
 * sg_device.hpp -- per-environment device math of the Space-Gym engine (fp32, one env per lane).
 *
 * Everything here is straight-line per-lane code with no memory traffic: the kernels in
 * sg_kernels.hip load one env into registers, call these functions, and store the result.
 *
 * What is restated, and from where (paths under /root/reference):
 *   - RHS            gym_space/dynamic_model.py:129-176, gym_space/helpers.py:22-35
 *   - integrator     dynamic_model.py:112-118 -> scipy.integrate.solve_ivp(method="RK45"): the SAME
 *                    Dormand-Prince 5(4) adaptive controller (initial-step rule, error norm,
 *                    accept/reject, step factor) in fp32, so that the engine follows the reference
 *                    also where RK45's own truncation error is visible (fast grazes near a planet).
 *   - events         dynamic_model.py:183-217 with scipy's sign-change detection per accepted RK
 *                    step (ivp.py:133-156) and a root on the same 4th-order dense output
 *                    (rk.py RkDenseOutput) -- located with a bracketed Newton iteration.
 *   - observation    gym_space/envs/spaceship_env.py:113-140, kepler.py:172-187
 *   - rewards        goal.py:147-158,160-164,204-227; kepler.py:43-156 (fp64 epilogue)
 *   - reset sampler  hexagonal_tiling.py:53-158, goal.py:133-145, kepler.py:233-267 on a counter-based
 *                    RNG (Philox4x32-10 -> xoshiro128++), see DESIGN.md.
 *
 * Heading: with Steering.velocity (every registered id) the RHS pins omega = 5*a1 for the whole
 * step (dynamic_model.py:138-141), so theta(t) = theta0 + omega*t exactly and only (x, y, vx, vy)
 * are integrated; the theta/omega components still enter RK45's norms exactly as in scipy.
 *
 * The same source is compiled by g++ into a test-only host twin (tests/host_twin) so the fp32
 * numerics can be checked against the oracle without a GPU.  It is never used by the product.
 */
#ifndef SG_DEVICE_HPP
#define SG_DEVICE_HPP

#include <math.h>
#include <stdint.h>

#include "sg_config.h"

#if defined(__HIPCC__)
#define SG_FN __device__ __forceinline__
#define SG_MFN __device__ __forceinline__
#else
#define SG_FN static inline
#define SG_MFN inline
#endif

#ifndef SG_STAMP
#define SG_STAMP(slot)
#endif

namespace sg {

// ------------------------------------------------------------------------------------------------
// scalar helpers
// ------------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
SG_FN float rsq(float x) { return __builtin_amdgcn_rsqf(x); }      // v_rsq_f32, 1 ulp
SG_FN float rcp(float x) { return __builtin_amdgcn_rcpf(x); }      // v_rcp_f32, 1 ulp
SG_FN float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }   // v_sqrt_f32, 1 ulp
SG_FN float flog2(float x) { return __builtin_amdgcn_logf(x); }    // v_log_f32
SG_FN float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
#else
SG_FN float rsq(float x) { return 1.0f / sqrtf(x); }
SG_FN float rcp(float x) { return 1.0f / x; }
SG_FN float fsqrt(float x) { return sqrtf(x); }
SG_FN float flog2(float x) { return log2f(x); }
SG_FN float fexp2(float x) { return exp2f(x); }
#endif

constexpr float kTwoPi = 6.283185307179586f;

// sin/cos for |r| <= pi/4 (minimax polynomials, ~1 ulp)
SG_FN void sincos_poly(float r, float &s, float &c) {
    float z = r * r;
    float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    s = fmaf(ps * z, r, r);
    float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    c = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
}

// sin/cos for |r| <= 0.36 (weighted minimax fits; 2e-8 / 4e-8 absolute in fp32 arithmetic, i.e. within rounding): the heading
// advance inside one env-step with Steering.velocity, |5 a1| * step_size <= 0.35.  Beyond the interval the error grows
// smoothly (2e-7 at 0.45).  7 instructions against 11 for sincos_poly.
SG_FN void sincos_small(float r, float &s, float &c) {
    float z = r * r;
    s = fmaf(r * z, fmaf(z, 8.295134641230106e-3f, -1.666649580001831e-1f), r);
    c = fmaf(z, fmaf(z, fmaf(z, -1.3838880180093369e-3f, 4.166642666449376e-2f), -0.5f), 1.0f);
}

// ------------------------------------------------------------------------------------------------
// Pairs of floats.  On the device a pair is a 64-bit register pair and +, -, *, fma2 are ONE v_pk_* instruction for both
// halves (measured: the integrator with the compiler's own packing switched off is 8.5 % slower); the integrator below keeps
// (x, y) pairs -- and pairs of independent scalars where it has them -- in this form.  The host twin (tests) does the same
// arithmetic element by element: fma2 is a fused multiply-add per half on both sides.
// ------------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
typedef float f2 __attribute__((ext_vector_type(2)));
SG_FN f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
SG_FN f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
#else
struct f2 { float x, y; };
SG_FN f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
SG_FN f2 operator+(f2 a, f2 b) { return mk2(a.x + b.x, a.y + b.y); }
SG_FN f2 operator-(f2 a, f2 b) { return mk2(a.x - b.x, a.y - b.y); }
SG_FN f2 operator*(f2 a, f2 b) { return mk2(a.x * b.x, a.y * b.y); }
SG_FN f2 operator*(f2 a, float b) { return mk2(a.x * b, a.y * b); }
SG_FN f2 operator*(float a, f2 b) { return mk2(a * b.x, a * b.y); }
SG_FN f2 fma2(f2 a, f2 b, f2 c) { return mk2(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)); }
#endif
SG_FN f2 sp2(float a) { return mk2(a, a); }
SG_FN f2 fma2(float a, f2 b, f2 c) { return fma2(sp2(a), b, c); }
SG_FN f2 fma2(f2 a, float b, f2 c) { return fma2(a, sp2(b), c); }
SG_FN f2 fma2(f2 a, float b, float c) { return fma2(a, sp2(b), sp2(c)); }
SG_FN f2 fma2(f2 a, f2 b, float c) { return fma2(a, b, sp2(c)); }

// sincos_small for two angles at once
SG_FN void sincos_small2(f2 r, f2 &s, f2 &c) {
    f2 z = r * r;
    s = fma2(r * z, fma2(z, 8.295134641230106e-3f, -1.666649580001831e-1f), r);
    c = fma2(z, fma2(z, fma2(z, -1.3838880180093369e-3f, 4.166642666449376e-2f), -0.5f), 1.0f);
}

// sin/cos for |a| < ~1e3 with a two-term Cody-Waite reduction to [-pi/4, pi/4]
SG_FN void sincos_acc(float a, float &s, float &c) {
    float k = rintf(a * 0.6366197723675814f);
    float r = fmaf(-k, 1.57079637050628662109375f, a);
    r = fmaf(-k, -4.37113900018624283e-8f, r);
    float sr, cr;
    sincos_poly(r, sr, cr);
    int q = (int)k & 3;
    float s1 = (q & 1) ? cr : sr, c1 = (q & 1) ? sr : cr;
    s = (q & 2) ? -s1 : s1;
    c = ((q + 1) & 2) ? -c1 : c1;
}

SG_FN float wrap_two_pi(float th) {  // dynamic_model.py:179-180 (Python %: result in [0, 2pi))
    th = th - kTwoPi * floorf(th * (1.0f / kTwoPi));
    th = (th < 0.0f) ? th + kTwoPi : th;
    return (th >= kTwoPi) ? th - kTwoPi : th;
}

SG_FN double rsqrt_f64(double x) {  // fp32 seed + one Newton step in fp64: ~2e-14 relative
    double y = (double)rsq((float)x);
    return y * (1.5 - 0.5 * x * y * y);
}

// ------------------------------------------------------------------------------------------------
// Dormand-Prince 5(4) tableau and Shampine's dense output: scipy/integrate/_ivp/rk.py class RK45
// ------------------------------------------------------------------------------------------------
constexpr float C2 = 1.0f / 5, C3 = 3.0f / 10, C4 = 4.0f / 5, C5 = 8.0f / 9;
constexpr float A21 = 1.0f / 5;
constexpr float A31 = 3.0f / 40, A32 = 9.0f / 40;
constexpr float A41 = 44.0f / 45, A42 = -56.0f / 15, A43 = 32.0f / 9;
constexpr float A51 = 19372.0f / 6561, A52 = -25360.0f / 2187, A53 = 64448.0f / 6561, A54 = -212.0f / 729;
constexpr float A61 = 9017.0f / 3168, A62 = -355.0f / 33, A63 = 46732.0f / 5247, A64 = 49.0f / 176,
                A65 = -5103.0f / 18656;
constexpr float B1 = 35.0f / 384, B3 = 500.0f / 1113, B4 = 125.0f / 192, B5 = -2187.0f / 6784, B6 = 11.0f / 84;
// Position increment in Nystrom form: sum_j B_j (v + h sum_l A_jl a_l) = v + h sum_l beta_l a_l, beta_l = sum_j B_j A_jl
// (sum beta = 1/2).  Algebraically the same DP5 step; h*v is then formed once, in fp64, for the reward path.
constexpr float BETA1 = 35.0f / 384, BETA3 = 50.0f / 159, BETA4 = 25.0f / 192, BETA5 = -243.0f / 6784;
// E sums to zero, so sum_j E_j K_j == sum_{j>=1} E_j (K_j - K_0): evaluated on differences in fp32
constexpr float E3 = 71.0f / 16695, E4 = -71.0f / 1920, E5 = 17253.0f / 339200, E6 = -22.0f / 525, E7 = 1.0f / 40;
// P[:, 1..3]; each column sums to zero over the stages, so Q_j = sum_{s>=2} P[s][j] (K_s - K_0)
constexpr float P31 = (float)(131558114200.0 / 32700410799.0), P32 = (float)(-68118460800.0 / 10900136933.0),
                P33 = (float)(87487479700.0 / 32700410799.0);
constexpr float P41 = (float)(-1754552775.0 / 470086768.0), P42 = (float)(14199869525.0 / 1410260304.0),
                P43 = (float)(-10690763975.0 / 1880347072.0);
constexpr float P51 = (float)(127303824393.0 / 49829197408.0), P52 = (float)(-318862633887.0 / 49829197408.0),
                P53 = (float)(701980252875.0 / 199316789632.0);
constexpr float P61 = (float)(-282668133.0 / 205662961.0), P62 = (float)(2019193451.0 / 616988883.0),
                P63 = (float)(-1453857185.0 / 822651844.0);
constexpr float P71 = (float)(40617522.0 / 29380423.0), P72 = (float)(-110615467.0 / 29380423.0),
                P73 = (float)(69997945.0 / 29380423.0);
constexpr float kRtol = 1e-3f, kAtol = 1e-6f;  // solve_ivp defaults (dynamic_model.py:112-118 passes none)
constexpr float kSafety = 0.9f, kMinFactor = 0.2f, kMaxFactor = 10.0f;  // rk.py:8-11
// Probe step (see Integrator::attempt; Steering.acceleration only -- with Steering.velocity the fast step below takes its
// place).  Every env-step is first tried as ONE Dormand-Prince step over the whole of it, before
// anything of scipy's step-size machinery is evaluated.  The step is kept if its error norm is below kProbeNorm (scipy's
// tolerance is norm < 1) and no event can have happened; a step that ends beyond a terminal surface with so small an error is
// terminal, and only its terminal state -- which feeds the x1000 reward -- is worked out on scipy's own step sequence; everything
// else follows scipy's sequence from t = 0 (select_initial_step and so on).  79 % of the env-steps are one full step for scipy
// too (the probe step IS its step); where scipy splits, the kept step ends within 3e-9 of scipy's two-step result (fp64, 221 771
// such steps of random rollouts: all had an error norm below 1e-3).  -DSG_PROBE_NORM=0.0f switches the probe off (A/B builds).
#ifndef SG_PROBE_NORM
#define SG_PROBE_NORM 1e-3f
#endif
constexpr float kProbeNorm = SG_PROBE_NORM;
// ... and the step's estimate of its own position error (the 4th- against the 5th-order position, absolute) below kProbePosErr.
// That is the quantity the Goal reward amplifies x1000: on 360 000 adversarial env-steps (|v| up to 2.5 within 7 cm of a surface)
// kept steps with an estimate below 3e-8 are within 3.6e-6 (relative) of the reference's reward, those above it up to 1.6e-5;
// in random rollouts 1e-5 of the env-steps exceed 2e-8 (99.99 % are below 1.3e-8).
#ifndef SG_PROBE_POS_ERR
#define SG_PROBE_POS_ERR 2e-8f
#endif
constexpr float kProbePosErr = SG_PROBE_POS_ERR;
// Kepler (no x1000 in its reward: where the reward is steep -- on the reference orbit, engine off -- the estimate is below 1e-9
// and kept steps are within fp32 rounding of the reference, 2e-6 relative; the estimate grows to 6e-7 only next to the planet,
// where the reward is flat): bounded for the state's sake (tolerance 1e-5).
#ifndef SG_PROBE_POS_ERR_KEPLER
#define SG_PROBE_POS_ERR_KEPLER 1e-6f
#endif
constexpr float kProbePosErrKepler = SG_PROBE_POS_ERR_KEPLER;
// Fast step (Steering.velocity; see Integrator::fast_step).  With omega constant over the env-step the heading is linear in t,
// so the thrust -(cos, sin)(theta0 + omega t) F and its first and second time integrals are known in closed form, and what is
// left to integrate numerically, x'' = gravity(x), has no velocity in it: Nystrom's fifth-order method (Hairer, Norsett,
// Wanner, Solving ODEs I, II.14: nodes 0, 1/5, 2/3, 1) needs FOUR evaluations of the gravity sum where the Dormand-Prince step
// needs seven of the whole right-hand side.  Over the 0.07 s of an env-step its result is closer to the exact flow than RK45's
// own (median 6e-13 against 2e-11 on the reference's transitions, tests/golden); it is kept where its own error indicator --
// the distance between the fourth stage's position and the fifth-order end position, third order -- is below kFastInd: on the
// reference's non-terminal 3P transitions those steps (97 %) end within 4e-9 (position) and 1.4e-8 (velocity) of the reference;
// on 60 000 adversarial env-steps per family (|v| up to 2.5 within 7 cm of a surface) kept steps are within 4.8e-6 (relative) of
// the reference's reward.  In random rollouts 0.08 % (3P) to 0.5 % (4P) of the env-steps exceed the bound: those are taken in
// two halves; 0.005 % end on scipy's sequence for their accuracy, 0.005-0.02 % because a graze cannot be excluded.
constexpr float N5_C2 = 1.0f / 5, N5_C3 = 2.0f / 3;
constexpr float N5_A21 = 1.0f / 50, N5_A31 = -1.0f / 27, N5_A32 = 7.0f / 27;
constexpr float N5_BB1 = 14.0f / 336, N5_BB2 = 100.0f / 336, N5_BB3 = 54.0f / 336;            // position weights (the fourth is 0)
constexpr float N5_B1 = 14.0f / 336, N5_B2 = 125.0f / 336, N5_B3 = 162.0f / 336, N5_B4 = 35.0f / 336;  // velocity weights
// fourth stage position - end position = h^2 sum_j D_j k_j,  D_j = abar_4j - bbar_j  (abar_4 = 3/10, -2/35, 9/35; sum D = 0)
constexpr float N5_D1 = (float)(3.0 / 10 - 14.0 / 336), N5_D2 = (float)(-2.0 / 35 - 100.0 / 336), N5_D3 = (float)(9.0 / 35 - 54.0 / 336);
#ifndef SG_FAST_IND
#define SG_FAST_IND 1.5e-6f
#endif
constexpr float kFastInd = SG_FAST_IND;
// (Kepler: no x1000 in its reward, as for kProbePosErrKepler)
#ifndef SG_FAST_IND_KEPLER
#define SG_FAST_IND_KEPLER 1e-5f
#endif
constexpr float kFastIndKepler = SG_FAST_IND_KEPLER;
// A fast step that ends beyond a terminal surface only has to decide that the env-step IS terminal (its terminal state is
// worked out on scipy's own sequence): an indicator below kFastIndCross (position error below 1e-6) and an end point at least
// kFastCrossDepth behind every surface it crossed (in the units of the event functions: |p - c|^2 - R^2, or a wall distance).
#ifndef SG_FAST_IND_CROSS
#define SG_FAST_IND_CROSS 3e-5f
#endif
constexpr float kFastIndCross = SG_FAST_IND_CROSS;
constexpr float kFastCrossDepth = 1e-5f;
constexpr int kMaxRkAttempts = 12;  // bound on accepted+rejected RK steps per env-step (reference mean: 1.19)
constexpr int kRootIters = 2;      // minimum fp32 safeguarded-Newton iterations before the fp64 Newton polish ...
constexpr int kRootMaxIters = 28;  // ... and the cap for lanes that have not settled by then (near-tangent grazes)

struct StepResult {
    double dXd, dYd; // displacement from the start position, fp64 accumulation of h v + h^2 sum_l beta_l a_l
    float dX, dY;    // the same rounded to fp32
    float vx, vy;
    float t;         // time actually advanced: step_size, or the event time
    float dth, om;   // heading advance theta(t) - theta0 and angular velocity at t
    int done;        // a terminal event fired (dynamic_model.py:124)
    int event;       // circle index, NC = world_max, NC + 1 = world_min
    int n_rk;        // accepted RK45 steps (diagnostics)
    int path;        // diagnostics: how the env-step was integrated (kPath*)
    float probe_err; // diagnostics: squared error norm of the probe step
    float probe_ep, probe_ev;  // diagnostics: largest absolute error estimate of a position / velocity component
};
// kept probe step | probe step beyond a terminal surface (terminal state from scipy's sequence) | scipy's sequence because the
// probe step's error norm was not small | ... because a graze between the probe step's ends could not be excluded | probe off
enum : int { kPathProbe = 0, kPathProbeTerminal = 1, kPathScipyErr = 2, kPathScipyClear = 3, kPathScipy = 4 };

// RHS acceleration at displacement p = (X, Y) from the start position and heading advance delta since the step started:
// thrust -(cos, sin)(theta0 + delta) * F  (dynamic_model.py:168-176) + sum of planet pulls (helpers.py:22-35), as
//   thrust_at(T, Tp, delta) + G m * pull(cq, p)   with  T = -F (cos, sin) theta0,  Tp = T rotated by +90 degrees.
// cq are the circle centres relative to the start position, as (x, y) pairs; the first NG circles gravitate.
template <bool SMALL>
SG_FN f2 thrust_at(f2 T, f2 Tp, float delta) {
    float sd, cd;
    // |delta| <= 5 * 0.07 with Steering.velocity (SMALL), <= 6 * 0.07 + 2.5 * 0.07^2 with Steering.acceleration
    if (SMALL) sincos_small(delta, sd, cd); else sincos_poly(delta, sd, cd);
    return fma2(T, sp2(cd), Tp * sd);
}
// sum_j d_j / |d_j|^3 over the gravitating circles, d_j = c_j - p (every planet has the same G m: one factor for the sum);
// r2[j] = |d_j|^2 for the caller's event functions.  The cubes of two planets are one packed pair.
template <int NG>
SG_FN f2 pull(f2 c0, f2 c1, f2 c2, f2 c3, f2 p, float (&r2)[NG > 0 ? NG : 1]) {
    static_assert(NG <= 4, "at most four gravitating circles");
    // (the centres come as named pairs, not as an array: an array of pairs is not split into registers by the compiler)
    f2 s = mk2(0.0f, 0.0f);
    if (NG >= 2) {
        const f2 da = c0 - p, db = c1 - p;
        const float ra = fmaf(da.x, da.x, da.y * da.y), rb = fmaf(db.x, db.x, db.y * db.y);
        r2[0] = ra; r2[NG >= 2 ? 1 : 0] = rb;
        const f2 i = mk2(rsq(ra), rsq(rb));
        const f2 w = i * i * i;
        s = fma2(db, sp2(w.y), da * w.x);
    }
    if (NG == 4) {
        const f2 da = c2 - p, db = c3 - p;
        const float ra = fmaf(da.x, da.x, da.y * da.y), rb = fmaf(db.x, db.x, db.y * db.y);
        r2[NG == 4 ? 2 : 0] = ra; r2[NG == 4 ? 3 : 0] = rb;
        const f2 i = mk2(rsq(ra), rsq(rb));
        const f2 w = i * i * i;
        s = fma2(db, sp2(w.y), fma2(da, sp2(w.x), s));
    }
    if (NG & 1) {
        const f2 dc = (NG == 1 ? c0 : c2) - p;
        const float rc = fmaf(dc.x, dc.x, dc.y * dc.y), ic = rsq(rc);
        r2[NG - 1] = rc;
        const float w = ic * ic * ic;
        s = (NG == 1) ? dc * w : fma2(dc, sp2(w), s);
    }
    return s;
}

// dynamic_model.make_step (dynamic_model.py:94-125) in fp32, as a resumable integrator: begin() does what
// RungeKutta.__init__ + the first event evaluation do, attempt() is one pass of RungeKutta._step_impl's loop body (one
// accepted or rejected RK step) followed by solve_ivp's event handling; run() loops over the attempts of one env-step (one
// pass per attempt for the whole wave: lanes that are done sit out) and takes the result.
//   NC circles with radii cR (Goal: the planets; Kepler: planet + border, both centred on the origin),
//   the first NG of them gravitate; WALLS adds the world_max / world_min events (dynamic_model.py:196-208).
//   The angular-velocity event (:210-212, limit 6) cannot fire: |omega| = |5 a1| <= 5 for actions in [-1, 1].
enum : int { kRkContinue = 0, kRkFinished = 1, kRkEvent = 2, kRkEventDeferred = 3 };

// ACCEL selects Steering.acceleration at compile time: the velocity-steering kernels (every registered id) carry none of it.
template <int NC, int NG, bool WALLS, bool ACCEL = false>
struct Integrator {
    // constants of the env-step
    float t_end, half_world, gm, F, om, alpha, w_limit, x0, y0, nCF, nSF;  // om: omega at t = 0; alpha: d omega / dt; nCF, nSF: -F (cos, sin) theta0 (thrust_at)
    float cax[NC], cay[NC], cR[NC], cR2[NC];
    f2 cq0, cq1, cq2, cq3;  // centres relative to the start position, (x, y) pairs (named: see pull())
    SG_MFN f2 cq_at(int k) const { return k == 0 ? cq0 : k == 1 ? cq1 : k == 2 ? cq2 : cq3; }
    double cRd[NC];
    float wxp, wyp, wxm, wym;
    // running state
    float t, X, Y, vx, vy, h_abs;
    double Xd, Yd;
    float k0[4], g[NC + 2];  // k0: stage 1 of the next attempt (v, a) -- FSAL; g: event functions at the start of the env-step
    f2 gp0;                  // the gravity sum (pull) at the start position
    bool rejected, probe;  // probe: the next attempt is the probe step (one step over the whole env-step, before select_initial_step)
    bool fast;             // ... and before that, the fast step (Steering.velocity)
    bool probe_crossing;   // the probe step ended beyond a terminal surface with a small error: the env-step is terminal
    float th0;             // heading at t = 0 (select_initial_step's scale)
    int n_rk, attempts, path;
    float probe_err, probe_ep, probe_ev;  // diagnostics

    // An accepted RK step over which at least one event function changed sign: everything solve_event() needs besides the
    // env-step constants (set_constants).  The rollout kernel's pilot wave hands such cases to its finisher wave.
    struct EventCase {
        float h, t, X, Y, vx, vy, Xn, Yn;  // the step [t, t + h]: displacement and velocity at its start, displacement at its end
        double Xd, Yd;                     // fp64 displacement at the start
        float k0[4], k2[4], k3[4], k4[4], k5[4], k6[4];  // stages (vx, vy, ax, ay); stage 2 has zero dense-output weight
        unsigned mask;                     // event functions with a sign change (bit NC + 2: angular velocity)
        float s_w;                         // root of the angular-velocity event in [0, 1] (Steering.acceleration), else 2
        int n_rk;
        static constexpr int kWords = 8 + 4 + 24 + 3;
    };

    // heading advance since t = 0: omega is constant (Steering.velocity) or linear in t (Steering.acceleration), so theta(t)
    // is known in closed form; RK45 integrates such polynomials exactly and its error estimate for them is 0.
    SG_MFN float phase(float tt) const { return ACCEL ? fmaf(om, tt, 0.5f * alpha * tt * tt) : om * tt; }

    // Steering.velocity: om_ = 5 a1 (the RHS overwrites omega, dynamic_model.py:138-141), alpha_ = 0.
    // Steering.acceleration: om_ = the state's omega, alpha_ = a1 * max_thruster_force / moi (dynamic_model.py:160-161,175).
    // The constants of the env-step that the event solver needs (all of begin() except the RK state).
    SG_MFN void set_constants(float h_total, float half_world_, float gm_, float F_, float om_, float alpha_, float w_limit_,
                              float x0_, float y0_, const float (&cax_)[NC], const float (&cay_)[NC], const float (&cR_)[NC],
                              const double (&cRd_)[NC]) {
        half_world = half_world_; gm = gm_; F = F_; om = om_; alpha = ACCEL ? alpha_ : 0.0f; w_limit = w_limit_;
        x0 = x0_; y0 = y0_;
#pragma unroll
        for (int k = 0; k < NC; k++) { cax[k] = cax_[k]; cay[k] = cay_[k]; cR[k] = cR_[k]; cR2[k] = cR_[k] * cR_[k]; cRd[k] = cRd_[k]; }
        t_end = h_total;
        // circle centres relative to the start position (fp32 working copy; the fp64 root polish uses cax/cay)
        {
            float rx[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ry[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < NC; k++) {
                rx[k] = cax_[k]; ry[k] = cay_[k];
#if defined(__HIP_DEVICE_COMPILE__)
                // (opaque scalars: left alone, the compiler pairs the centres' x's of two planets straight out of the caller's
                //  arrays and then transposes to (x, y) pairs through memory)
                asm volatile("" : "+v"(rx[k]), "+v"(ry[k]));
#endif
            }
            const f2 o = mk2(x0, y0);
            cq0 = mk2(rx[0], ry[0]) - o; cq1 = mk2(rx[1], ry[1]) - o; cq2 = mk2(rx[2], ry[2]) - o; cq3 = mk2(rx[3], ry[3]) - o;
        }
        wxp = half_world - x0; wyp = half_world - y0; wxm = half_world + x0; wym = half_world + y0;
    }

    SG_MFN void begin(float h_total, float half_world_, float gm_, float F_, float om_, float alpha_, float w_limit_,
                      float x0_, float y0_, float th0_, float vx0, float vy0, const float (&cax_)[NC],
                      const float (&cay_)[NC], const float (&cR_)[NC], const double (&cRd_)[NC], bool use_probe = true) {
        SG_STAMP(8);
        set_constants(h_total, half_world_, gm_, F_, om_, alpha_, w_limit_, x0_, y0_, cax_, cay_, cR_, cRd_);
        {
            float S0, C0;
            sincos_acc(th0_, S0, C0);
            nCF = -(C0 * F); nSF = -(S0 * F);
        }
        t = 0.0f; X = 0.0f; Y = 0.0f; vx = vx0; vy = vy0;
        Xd = 0.0; Yd = 0.0;
        const f2 T = mk2(nCF, nSF);

        // RungeKutta.__init__ (rk.py:85-105): f0
        float r2s[NG > 0 ? NG : 1];  // |p - c_j|^2 at the start
        f2 origin = mk2(0.0f, 0.0f);
#if defined(__HIP_DEVICE_COMPILE__)
        // (opaque: with a literal 0 the compiler pairs the squared distances of two planets instead of (x, y) of one and
        //  transposes the centres through memory for it)
        asm volatile("" : "+v"(origin));
#endif
        gp0 = pull<NG>(cq0, cq1, cq2, cq3, origin, r2s);
        const f2 a0 = fma2(gm, gp0, T);  // (heading advance 0: the thrust is T itself)
        k0[0] = vx; k0[1] = vy; k0[2] = a0.x; k0[3] = a0.y;
        th0 = th0_;
        // the probe step comes first; select_initial_step only for the lanes whose probe step is not kept (attempt())
        // (use_probe = false: scipy's sequence straight away -- the replay of an env-step that is known to be terminal)
        probe = kProbeNorm > 0.0f && use_probe;
        fast = !ACCEL && probe;  // (its series are for |omega h| <= 0.36: 5 x 0.07 -- sg_create refuses other step sizes)
        probe_crossing = false;
        path = kPathScipy; probe_err = probe_ep = probe_ev = 0.0f;
        if (probe) h_abs = t_end; else initial_step();

        SG_STAMP(9);
        // event functions at (t0, y0), ivp.py:646.  attempt() only needs the SIGN of a circle event |p - c| - R (which events
        // are active over an accepted step), so it carries |p - c|^2 - R^2: no square root, and |p - c|^2 is what the gravity
        // term of the stage at that point has formed already.  solve_event() works on the distances themselves.
    #pragma unroll
        for (int k = 0; k < NG; k++) g[k] = r2s[k] - cR2[k];
    #pragma unroll
        for (int k = NG; k < NC; k++) { const f2 c = cq_at(k); g[k] = fmaf(c.x, c.x, c.y * c.y) - cR2[k]; }
        if (WALLS) { g[NC] = fminf(wxp, wyp); g[NC + 1] = fminf(wxm, wym); }

        rejected = false; n_rk = 0; attempts = 0;
    }

    // common.py select_initial_step (called from RungeKutta.__init__, rk.py:85-105) at (t0, y0): sets h_abs.  Needs begin().
    SG_MFN void initial_step() {
        const f2 T = mk2(nCF, nSF), Tp = mk2(-nSF, nCF), V = mk2(k0[0], k0[1]), a0 = mk2(k0[2], k0[3]);
        // scale = atol + |y0| rtol over all SIX components (x, y, theta, vx, vy, omega), as the pairs (x, y), (vx, vy)
        // and the two scalars theta, omega
        const f2 is_p = mk2(rcp(fmaf(fabsf(x0), kRtol, kAtol)), rcp(fmaf(fabsf(y0), kRtol, kAtol)));
        const f2 is_v = mk2(rcp(fmaf(fabsf(V.x), kRtol, kAtol)), rcp(fmaf(fabsf(V.y), kRtol, kAtol)));
        const float isth = rcp(fmaf(fabsf(th0), kRtol, kAtol)), isom = rcp(fmaf(fabsf(om), kRtol, kAtol));
        const f2 y_p = mk2(x0, y0) * is_p, y_v = V * is_v;
        const float y_th = th0 * isth, y_om = om * isom;
        const f2 n0 = fma2(y_v, y_v, y_p * y_p);
        const float d0 = fsqrt(fmaf(y_om, y_om, fmaf(y_th, y_th, n0.x + n0.y)) * (1.0f / 6));
        // f0 = (vx, vy, omega, ax, ay, alpha)
        const f2 f_p = V * is_p, f_v = a0 * is_v;
        const float f_th = om * isth, f_om = ACCEL ? alpha * isom : 0.0f;
        const f2 n1 = fma2(f_v, f_v, f_p * f_p);
        const float d1 = fsqrt(fmaf(f_om, f_om, fmaf(f_th, f_th, n1.x + n1.y)) * (1.0f / 6));
        float h0 = (fminf(d0, d1) < 1e-5f) ? 1e-6f : 0.01f * d0 * rcp(d1);  // d0 < 1e-5 or d1 < 1e-5 (common.py:96-99)
        h0 = fminf(h0, t_end);
        // y1 = y0 + h0 f0 ; f1 = fun(t0 + h0, y1)   (Euler probe: theta1 = theta0 + h0 omega, omega1 = omega + h0 alpha)
        float r2p[NG > 0 ? NG : 1];
        const f2 a1 = fma2(gm, pull<NG>(cq0, cq1, cq2, cq3, V * h0, r2p), thrust_at<!ACCEL>(T, Tp, h0 * om));
        const f2 e_p = (a0 * h0) * is_p, e_v = (a1 - a0) * is_v;
        const float e_th = ACCEL ? h0 * alpha * isth : 0.0f;
        const f2 n2 = fma2(e_v, e_v, e_p * e_p);
        const float d2 = fsqrt(fmaf(e_th, e_th, n2.x + n2.y) * (1.0f / 6)) * rcp(h0);
        const float dm = fmaxf(d1, d2);
        const float h1 = (dm <= 1e-15f) ? fmaxf(1e-6f, h0 * 1e-3f)
                                        : fexp2(0.2f * flog2(0.01f * rcp(dm)));  // (0.01/dm)^(1/5)
        h_abs = fminf(fminf(100.0f * h0, h1), t_end);
    }

    // scipy looks at the event functions at the end of each of ITS steps, so it can see a graze that dips below a surface and
    // comes out again within the env-step; a single step over the whole env-step only has the two ends.  It is kept only where
    // no such dip is possible.  With a = F + 1.1 bounding the acceleration (this step's engine force plus the gravity at a
    // surface, <= 1.0 + the other planets): a path of length L <= h |v| whose ends are both outside a circle of radius R by d
    // stays outside if (R + d)^2 - R^2 >= L^2 / 4 + R a h^2 / 4 (chord, plus the bend the acceleration gives it); inside a
    // circle (Kepler's border) the chord is harmless and R a h^2 / 4 is enough; a wall needs a h^2 / 8.  (20 % on top.)
    // A path that does come that close to a surface still cannot cross it and come back unless its distance to the surface has
    // an extremum between the two ends.  Circle: the radial velocity changes by at most (a + v_t^2 / r) per unit time
    // (v_t <= |v|), so it keeps its sign if both ends move the same way and |d0 + d1| > h (a r + |v|^2), d = (c - p).v
    // (r = R: the path is next to the surface).  Wall: the same for the velocity component across it, |v0 + v1| > a h.
    // gn: the event functions at the end, as g in begin(); Pn: the end position relative to the start.
    SG_MFN bool no_graze(f2 V, f2 Vn, f2 Pn, float h, const float (&gn)[NC + 2]) const {
        const f2 vv0 = V * V, vv1 = Vn * Vn;
        const float v2max = fmaxf(vv0.x + vv0.y, vv1.x + vv1.y);
        const float ah = 1.2f * h * (F + 1.1f), chord = 0.3f * h * h * v2max;
        bool keep = true;
#pragma unroll
        for (int k = 0; k < NC; k++) {
            const f2 c = cq_at(k), cn = c - Pn;
            const float d0 = fmaf(c.x, V.x, c.y * V.y), d1 = fmaf(cn.x, Vn.x, cn.y * Vn.y);
            const bool mono = d0 * d1 > 0.0f && fabsf(d0 + d1) > fmaf(ah, cR[k], 1.2f * h * v2max);
            const float clear = fmaf(0.25f * h * ah, cR[k], g[k] > 0.0f ? chord : 0.0f);
            const bool near = fminf(fabsf(g[k]), fabsf(gn[k])) <= clear;
            keep = keep && !(near && !mono);
        }
        if (WALLS) {
            const f2 s = V + Vn;
            const float clear = 0.125f * h * ah;
            const bool mono_x = V.x * Vn.x > 0.0f && fabsf(s.x) > ah, mono_y = V.y * Vn.y > 0.0f && fabsf(s.y) > ah;
            const bool near_x = fminf(fminf(wxp, wxp - Pn.x), fminf(wxm, wxm + Pn.x)) <= clear;
            const bool near_y = fminf(fminf(wyp, wyp - Pn.y), fminf(wym, wym + Pn.y)) <= clear;
            keep = keep && !(near_x && !mono_x) && !(near_y && !mono_y);
        }
        return keep;
    }

    // One step of Nystrom's method over h from the displacement P0 (relative to the start position of the env-step) with
    // velocity V, thrust T (Tp: T turned by +90 degrees) and gravity sum g1 there:  rest = displacement over the step - h V,
    // Vn = the velocity at its end, w = fourth stage position - end position (the error indicator).
    SG_MFN void nystrom5(float h, f2 P0, f2 V, f2 T, f2 Tp, f2 g1, f2 &rest, f2 &Vn, f2 &w) const {
        const float hh = h * h, hg = hh * gm;
        // A = (1 - cos phi) / phi^2, B = (phi - sin phi) / phi^2 at phi = omega c h for the nodes c = 1/5, 2/3 (packed) and 1,
        // S = sin phi / phi at the end: series in phi^2, |phi| <= 0.35 (truncation below 1e-7 relative)
        const float ph = om * h, z4 = ph * ph;
        const f2 ph23 = mk2(N5_C2, N5_C3) * ph, z23 = ph23 * ph23;
        const f2 A23 = fma2(z23, fma2(z23, 1.0f / 720, -1.0f / 24), 0.5f);
        const f2 B23 = ph23 * fma2(z23, fma2(z23, 1.0f / 5040, -1.0f / 120), 1.0f / 6);
        const float A4 = fmaf(z4, fmaf(z4, 1.0f / 720, -1.0f / 24), 0.5f);
        const float B4 = ph * fmaf(z4, fmaf(z4, 1.0f / 5040, -1.0f / 120), 1.0f / 6);
        const float S4 = fmaf(z4, fmaf(z4, fmaf(z4, -1.0f / 5040, 1.0f / 120), -1.0f / 6), 1.0f);
        const f2 p2 = fma2(T, sp2(A23.x), Tp * B23.x), p3 = fma2(T, sp2(A23.y), Tp * B23.y), p4 = fma2(T, sp2(A4), Tp * B4);
        float r2x[NG > 0 ? NG : 1];
        // stage positions: P0 + c h V + (c h)^2 p_c + h^2 G m sum_j abar_cj pull_j
        const f2 g2 = pull<NG>(cq0, cq1, cq2, cq3, fma2(hg * N5_A21, g1, fma2((N5_C2 * N5_C2) * hh, p2, fma2(N5_C2 * h, V, P0))), r2x);
        const f2 g3 = pull<NG>(cq0, cq1, cq2, cq3,
                               fma2(hg, fma2(N5_A32, g2, g1 * N5_A31), fma2((N5_C3 * N5_C3) * hh, p3, fma2(N5_C3 * h, V, P0))), r2x);
        // (the end position needs no fourth evaluation: its weight is 0)
        rest = fma2(hg, fma2(N5_BB3, g3, fma2(N5_BB2, g2, g1 * N5_BB1)), p4 * hh);
        w = fma2(N5_D3, g3, fma2(N5_D2, g2, g1 * N5_D1)) * hg;
        const f2 g4 = pull<NG>(cq0, cq1, cq2, cq3, fma2(h, V, P0) + (rest + w), r2x);
        Vn = fma2(h * gm, fma2(N5_B4, g4, fma2(N5_B3, g3, fma2(N5_B2, g2, g1 * N5_B1))),
                  fma2(h, fma2(T, sp2(S4), Tp * (ph * A4)), V));
    }

    template <typename SINK>
    SG_MFN int fast_step(StepResult &o, SINK &&sink) {
        const float h = t_end;
        const f2 V = mk2(vx, vy), T = mk2(nCF, nSF), Tp = mk2(-nSF, nCF);
        f2 rest, Vn, w;
        f2 origin = mk2(0.0f, 0.0f);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(origin));  // (as in begin())
#endif
        nystrom5(h, origin, V, T, Tp, gp0, rest, Vn, w);
        // h v0 in fp64 (the Goal reward multiplies position differences by up to 1000), the O(h^2) rest in fp32
        double Xdn = (double)h * (double)vx + (double)rest.x, Ydn = (double)h * (double)vy + (double)rest.y;
        float Xn = (float)Xdn, Yn = (float)Ydn;
        float ind2 = fmaf(w.x, w.x, w.y * w.y);

        // event functions at the end (as end_events in attempt())
        float gn[NC + 2], ggmin, shallow;  // shallow: how far behind the surfaces it crossed the step ends, at least
        auto end_events = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < NC; k++) {
                const f2 c = cq_at(k);
                const float ex = c.x - Xn, ey = c.y - Yn;
                gn[k] = fmaf(ex, ex, ey * ey) - cR2[k];
            }
            gn[NC] = gn[NC + 1] = 1.0f;
            if (WALLS) { gn[NC] = fminf(wxp - Xn, wyp - Yn); gn[NC + 1] = fminf(wxm + Xn, wym + Yn); }
            ggmin = 1.0f; shallow = 3.0e38f;
#pragma unroll
            for (int k = 0; k < NC + (WALLS ? 2 : 0); k++) {
                const float gg = g[k] * gn[k];
                ggmin = fminf(ggmin, gg);
                shallow = fminf(shallow, gg <= 0.0f ? fabsf(gn[k]) : 3.0e38f);
            }
        };
        end_events();
        const float lim = WALLS ? kFastInd : kFastIndKepler;
        probe_ev = 1.0f;
        if (!(ggmin <= 0.0f) && !(ind2 <= lim * lim)) {
            probe_ev = 2.0f;
            // Not accurate enough for the reward (a fast pass close to a planet: 0.08 % of the 3P env-steps, 0.5 % of the 4P
            // ones): the same env-step as two steps of half the length -- 1/32 of the error, 1/16 of the indicator -- each
            // held to a quarter of the bound (the velocity between the two is rounded to fp32, 3e-9 of position by itself).
            const float hs = 0.5f * h;
            f2 rA, VA, wA, rB, wB;
            nystrom5(hs, origin, V, T, Tp, gp0, rA, VA, wA);
            const double XdA = (double)hs * (double)vx + (double)rA.x, YdA = (double)hs * (double)vy + (double)rA.y;
            const f2 PA = mk2((float)XdA, (float)YdA);
            float sd, cd;
            sincos_small(om * hs, sd, cd);
            const f2 T1 = fma2(T, sp2(cd), Tp * sd), Tp1 = mk2(-T1.y, T1.x);
            float r2a[NG > 0 ? NG : 1];
            const f2 gA = pull<NG>(cq0, cq1, cq2, cq3, PA, r2a);
            nystrom5(hs, PA, VA, T1, Tp1, gA, rB, Vn, wB);
            Xdn = XdA + ((double)hs * (double)VA.x + (double)rB.x); Ydn = YdA + ((double)hs * (double)VA.y + (double)rB.y);
            Xn = (float)Xdn; Yn = (float)Ydn;
            ind2 = 16.0f * fmaxf(fmaf(wA.x, wA.x, wA.y * wA.y), fmaf(wB.x, wB.x, wB.y * wB.y));
            w = mk2(fmaxf(fabsf(wA.x), fabsf(wB.x)), fmaxf(fabsf(wA.y), fabsf(wB.y))) * 4.0f;
            end_events();
        }
        probe_err = ind2; probe_ep = fmaxf(fabsf(w.x), fabsf(w.y));  // (diagnostics; probe_ev: 1 = one step, 2 = two halves)

        const bool crossing = ggmin <= 0.0f;
        probe = false;
        if (!crossing) {
            if (!(ind2 <= lim * lim)) {  // (not even in two halves)
                path = kPathScipyErr;
                initial_step();
                return kRkContinue;
            }
            if (no_graze(V, Vn, mk2(Xn, Yn), h, gn)) {
                n_rk = 1; path = kPathProbe;
                t = t_end; X = Xn; Y = Yn; Xd = Xdn; Yd = Ydn; vx = Vn.x; vy = Vn.y;
                return kRkFinished;
            }
            path = kPathScipyClear;
        } else if (ind2 <= kFastIndCross * kFastIndCross && shallow >= kFastCrossDepth) {
            probe_crossing = true; path = kPathProbeTerminal;
            if (sink()) {
                o.dXd = 0.0; o.dYd = 0.0; o.dX = 0.0f; o.dY = 0.0f; o.vx = 0.0f; o.vy = 0.0f; o.t = t_end; o.dth = 0.0f; o.om = om;
                o.done = 1; o.event = -1; o.n_rk = 0; o.path = path;
                return kRkEventDeferred;
            }
        } else {
            path = kPathScipyErr;
        }
        initial_step();
        return kRkContinue;
    }

    // One RK attempt.  Returns kRkContinue, kRkFinished (the caller then takes the result with finish(): run() below), or
    // fills `o` and returns kRkEvent / kRkEventDeferred.
    // `sink()`: asked when the env-step is known to be terminal -- the probe step ended beyond a terminal surface, or an
    // accepted step of scipy's sequence has an event -- before anything is solved; if it returns true the caller takes the
    // terminal state from elsewhere (o.done = 1, return value kRkEventDeferred: the wave-pair rollout kernels replay such
    // env-steps in batches), else it is worked out here.
    struct NoSink { SG_MFN bool operator()() const { return false; } };
    template <typename SINK = NoSink>
    SG_MFN int attempt(StepResult &o, SINK &&sink = NoSink()) {
        // (t < t_end holds on entry: begin() starts at t = 0 and the exits below leave the loop once t_end is reached or the
        //  attempt budget is spent)
        const bool probing = probe;  // (only ever set for the first attempt after begin())
        probe = false;
        attempts++;
        // RungeKutta._step_impl (rk.py:111-176)
        h_abs = fmaxf(h_abs, 1e-9f);
        float t_new = t + h_abs;
        if (t_new - t_end > 0.0f) t_new = t_end;
        const float h = t_new - t;
        h_abs = h;
        const f2 V = mk2(vx, vy), P = mk2(X, Y), a0 = mk2(k0[2], k0[3]);  // K_1 = (V, a0): velocity and acceleration at t (FSAL)
        // Thrust at the stage times t + c h (stage 6 and the FSAL stage share t + h).  With Steering.velocity the heading
        // advance is omega (t + c h) = omega t + c (omega h) and small: the five sines and cosines are two packed pairs and one.
        f2 th1, th2, th3, th4, th5;
        {
            const f2 T = mk2(nCF, nSF), Tp = mk2(-nSF, nCF);
            if (ACCEL) {
                th1 = thrust_at<false>(T, Tp, phase(fmaf(C2, h, t))); th2 = thrust_at<false>(T, Tp, phase(fmaf(C3, h, t)));
                th3 = thrust_at<false>(T, Tp, phase(fmaf(C4, h, t))); th4 = thrust_at<false>(T, Tp, phase(fmaf(C5, h, t)));
                th5 = thrust_at<false>(T, Tp, phase(t + h));
            } else {
                const float ph0 = om * t, phh = om * h;
                f2 s12, c12, s34, c34;
                sincos_small2(fma2(mk2(C2, C3), phh, ph0), s12, c12);
                sincos_small2(fma2(mk2(C4, C5), phh, ph0), s34, c34);
                th1 = fma2(T, sp2(c12.x), Tp * s12.x); th2 = fma2(T, sp2(c12.y), Tp * s12.y);
                th3 = fma2(T, sp2(c34.x), Tp * s34.x); th4 = fma2(T, sp2(c34.y), Tp * s34.y);
                th5 = thrust_at<true>(T, Tp, ph0 + phh);
            }
        }
        // rk_step (rk.py:14-71); K_s = (v_s, a_s) as two pairs.  sa_s = sum_l A_sl a_l, i.e. (v_s - V) / h: kept for the
        // error estimate below.
        float r2x[NG > 0 ? NG : 1], r2n[NG > 0 ? NG : 1];
        // The stage position needs the velocities of the earlier stages, and those the accelerations up to the stage before
        // last: the accelerations of stages 2 and 3 depend on stage 1 only, those of 4 and 5 on 2 and 3, those of 6 and the
        // FSAL stage on 4 and 5 -- two interleaved chains of three, written pair by pair so that the scheduler sees them.
        const f2 sa1 = a0 * A21;
        const f2 v1 = fma2(h, sa1, V);
        const f2 g1 = pull<NG>(cq0, cq1, cq2, cq3, fma2(h, V * A21, P), r2x);
        const f2 g2 = pull<NG>(cq0, cq1, cq2, cq3, fma2(h, fma2(A32, v1, V * A31), P), r2x);
        const f2 a1 = fma2(gm, g1, th1), a2 = fma2(gm, g2, th2);
        const f2 sa2 = fma2(A32, a1, a0 * A31);
        const f2 v2 = fma2(h, sa2, V);
        const f2 sa3 = fma2(A43, a2, fma2(A42, a1, a0 * A41));
        const f2 v3 = fma2(h, sa3, V);
        const f2 g3 = pull<NG>(cq0, cq1, cq2, cq3, fma2(h, fma2(A43, v2, fma2(A42, v1, V * A41)), P), r2x);
        const f2 g4 = pull<NG>(cq0, cq1, cq2, cq3, fma2(h, fma2(A54, v3, fma2(A53, v2, fma2(A52, v1, V * A51))), P), r2x);
        const f2 a3 = fma2(gm, g3, th3), a4 = fma2(gm, g4, th4);
        const f2 sa4 = fma2(A54, a3, fma2(A53, a2, fma2(A52, a1, a0 * A51)));
        const f2 v4 = fma2(h, sa4, V);
        const f2 sa5 = fma2(A65, a4, fma2(A64, a3, fma2(A63, a2, fma2(A62, a1, a0 * A61))));
        const f2 v5 = fma2(h, sa5, V);
        // the position increment in Nystrom form, h v + h^2 sum_l beta_l a_l, accumulated in fp64
        const f2 nys = fma2(BETA5, a4, fma2(BETA4, a3, fma2(BETA3, a2, a0 * BETA1))) * (h * h);
        const double Xdn = Xd + ((double)h * (double)vx + (double)nys.x);
        const double Ydn = Yd + ((double)h * (double)vy + (double)nys.y);
        const float Xn = (float)Xdn, Yn = (float)Ydn;
        const f2 Pn = mk2(Xn, Yn);
        const f2 g5 = pull<NG>(cq0, cq1, cq2, cq3, fma2(h, fma2(A65, v4, fma2(A64, v3, fma2(A63, v2, fma2(A62, v1, V * A61)))), P), r2x);
        const f2 g6 = pull<NG>(cq0, cq1, cq2, cq3, Pn, r2n);
        const f2 a5 = fma2(gm, g5, th5), a6 = fma2(gm, g6, th5);  // a6: f_new (FSAL)
        // v_new - v = h * sum_j B_j a_j
        const f2 sa6 = fma2(B6, a5, fma2(B5, a4, fma2(B4, a3, fma2(B3, a2, a0 * B1))));
        const f2 v6 = fma2(h, sa6, V);
        const float vxn = v6.x, vyn = v6.y;

        // error norm over six components; theta and omega contribute exactly zero (sum E = 0, d omega/dt = 0).
        // sum_j E_j K_j on differences to stage 1 (sum E = 0); for the position components K_j = v_j and v_j - V = h sa_j:
        // the stage sums themselves, without the cancellation of a subtraction.
        float err2, abs_ep, abs_ev;
        {
            const f2 e_p = fma2(E7, sa6, fma2(E6, sa5, fma2(E5, sa4, fma2(E4, sa3, sa2 * E3)))) * h;
            const f2 e_v = fma2(E7, a6 - a0, fma2(E6, a5 - a0, fma2(E5, a4 - a0, fma2(E4, a3 - a0, (a2 - a0) * E3))));
            const f2 p0 = mk2(x0, y0) + P, p1 = mk2(x0, y0) + Pn;
            const f2 sc_p = fma2(mk2(fmaxf(fabsf(p0.x), fabsf(p1.x)), fmaxf(fabsf(p0.y), fabsf(p1.y))), kRtol, kAtol);
            const f2 sc_v = fma2(mk2(fmaxf(fabsf(vx), fabsf(vxn)), fmaxf(fabsf(vy), fabsf(vyn))), kRtol, kAtol);
            const f2 q_p = (e_p * h) * mk2(rcp(sc_p.x), rcp(sc_p.y)), q_v = (e_v * h) * mk2(rcp(sc_v.x), rcp(sc_v.y));
            const f2 n = fma2(q_v, q_v, q_p * q_p);
            err2 = n.x + n.y;
            abs_ep = fmaxf(fabsf(e_p.x), fabsf(e_p.y)) * h; abs_ev = fmaxf(fabsf(e_v.x), fabsf(e_v.y)) * h;
        }
        // err = sqrt(err2 / 6) is only compared with 1 and raised to -1/5 (rk.py:155-168): both from its square
        const float err = err2 * (1.0f / 6);

        // event functions at the end of the attempted step (ivp.py:673-694, find_active_events with direction 0); evaluated
        // after the step has been accepted -- or before that decision when the probe-step option needs them for it
        float gn[NC + 2], gg[NC + 2], ggmin = 1.0f;
        auto end_events = [&]() __attribute__((always_inline)) {
            // sign of |p - c| - R (see begin()); the gravitating circles' |p - c|^2 is the FSAL stage's
#pragma unroll
            for (int k = 0; k < NG; k++) gn[k] = r2n[k] - cR2[k];
#pragma unroll
            for (int k = NG; k < NC; k++) {
                const f2 c = cq_at(k);
                float ex = c.x - Xn, ey = c.y - Yn;
                gn[k] = fmaf(ex, ex, ey * ey) - cR2[k];
            }
            if (WALLS) { gn[NC] = fminf(wxp - Xn, wyp - Yn); gn[NC + 1] = fminf(wxm + Xn, wym + Yn); }
            // sign change or a zero at either end (find_active_events, ivp.py:128-156): g gn <= 0, which is
            // (g <= 0 && gn >= 0) || (g >= 0 && gn <= 0) as |g| is 0 or >= 1e-8 (no underflow).  The smallest product decides
            // whether there is any event; which ones is only worked out when there is.
#pragma unroll
            for (int k = 0; k < NC + (WALLS ? 2 : 0); k++) { gg[k] = g[k] * gn[k]; ggmin = fminf(ggmin, gg[k]); }
        };
        end_events();
        // angular-velocity event max_abs_vel_angle - |omega| (dynamic_model.py:210-212): only live with Steering.acceleration
        // (|5 a1| <= 5 otherwise); omega is linear in t, so its root is closed-form
        float s_w = 2.0f;
        bool w_event = false;
        if constexpr (ACCEL) {
            const float w0 = fmaf(alpha, t, om), w1 = fmaf(alpha, t_new, om);
            const float gw0 = w_limit - fabsf(w0), gw1 = w_limit - fabsf(w1);
            if ((gw0 <= 0.0f && gw1 >= 0.0f) || (gw0 >= 0.0f && gw1 <= 0.0f)) {
                // crossing of +-limit between w0 and w1: the limit with the sign of whichever end is outside / larger
                const float lim = (fabsf(w1) >= fabsf(w0) ? w1 : w0) >= 0.0f ? w_limit : -w_limit;
                const float dw = w1 - w0;
                s_w = (dw != 0.0f) ? fminf(fmaxf((lim - w0) * rcp(dw), 0.0f), 1.0f) : 0.0f;
                w_event = true;
            }
        }

        // Probe step: the first attempt of every env-step covers the whole of it (h = t_end), before select_initial_step has
        // been evaluated at all.
        //   kept      error norm below kProbeNorm and position error estimate below kProbePosErr, no event function changes sign, and no graze between the two ends possible
        //             (below): the env-step is this one step.  For 79 % of the env-steps that IS scipy's sequence; where scipy
        //             splits the step (a state component near zero makes its scale, and so the first step, small) its
        //             two-step result is within 3e-9 of this one.
        //   terminal  error norm below kProbeNorm and the step ends beyond a terminal surface: scipy's sequence ends at or
        //             beyond it too, so the env-step is terminal.  The terminal state -- the reward multiplies it by up to 1000
        //             -- has to be the root on scipy's own dense output: `sink` may take the env-step over (the wave-pair
        //             kernels replay it later, many at a time); otherwise scipy's sequence is run here from t = 0.
        //   else      scipy's sequence from t = 0: select_initial_step, then attempts until t_end.  If it starts with the
        //             whole env-step, this attempt was its first one and goes on into the accept / reject logic below.
        if (probing) {
            probe_err = err; probe_ep = abs_ep; probe_ev = abs_ev;
            // (WALLS: the Goal family)
            const bool small = err <= kProbeNorm * kProbeNorm && abs_ep <= (WALLS ? kProbePosErr : kProbePosErrKepler);
            const bool crossing = ggmin <= 0.0f || w_event;
            if (small && !crossing) {
                if (no_graze(V, v6, Pn, h, gn)) {  // (the probe step only has the two ends of the env-step)
                    n_rk = 1; path = kPathProbe;
                    t = t_new; X = Xn; Y = Yn; Xd = Xdn; Yd = Ydn; vx = vxn; vy = vyn;
                    return kRkFinished;
                }
                path = kPathScipyClear;
            } else if (small) {
                probe_crossing = true; path = kPathProbeTerminal;
                if (sink()) {
                    o.dXd = 0.0; o.dYd = 0.0; o.dX = 0.0f; o.dY = 0.0f; o.vx = 0.0f; o.vy = 0.0f; o.t = t_end; o.dth = 0.0f; o.om = om;
                    o.done = 1; o.event = -1; o.n_rk = 0; o.path = path;
                    return kRkEventDeferred;
                }
            } else {
                path = kPathScipyErr;
            }
            initial_step();
            if (h_abs < t_end) return kRkContinue;  // scipy starts with a shorter step: this attempt was one too many
        }

        // safety * err_norm^(-1/5), once for both outcomes (a wave usually has lanes of either kind); err = 0 gives +inf
        const float shrink = kSafety * fexp2(-0.1f * flog2(err));
        if (!(err < 1.0f)) {  // rejected (also for NaN): shrink and retry
            h_abs = h * fmaxf(kMinFactor, shrink);
            rejected = true;
            if (attempts >= kMaxRkAttempts) return kRkFinished;
            return kRkContinue;
        }
        float factor = fminf(kMaxFactor, shrink);  // (rk.py:166-168: MAX_FACTOR when the error norm is 0)
        if (rejected) factor = fminf(1.0f, factor);
        rejected = false;
        h_abs = h * factor;
        n_rk++;

        // events over this accepted step
        unsigned mask = 0;
        if (ggmin <= 0.0f) {
#pragma unroll
            for (int k = 0; k < NC + (WALLS ? 2 : 0); k++)
                if (gg[k] <= 0.0f) mask |= 1u << k;
        }
        if (w_event) mask |= 1u << (NC + 2);

        if (mask) {
            if (sink()) {
                o.dXd = 0.0; o.dYd = 0.0; o.dX = 0.0f; o.dY = 0.0f; o.vx = 0.0f; o.vy = 0.0f; o.t = t; o.dth = 0.0f; o.om = om;
                o.done = 1; o.event = -1; o.n_rk = n_rk; o.path = path;
                return kRkEventDeferred;
            }
            EventCase ev;
            ev.h = h; ev.t = t; ev.X = X; ev.Y = Y; ev.vx = vx; ev.vy = vy; ev.Xn = Xn; ev.Yn = Yn; ev.Xd = Xd; ev.Yd = Yd;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                ev.k0[i] = i ? V.y : V.x; ev.k2[i] = i ? v2.y : v2.x; ev.k3[i] = i ? v3.y : v3.x; ev.k4[i] = i ? v4.y : v4.x;
                ev.k5[i] = i ? v5.y : v5.x; ev.k6[i] = i ? v6.y : v6.x;
                ev.k0[i + 2] = i ? a0.y : a0.x; ev.k2[i + 2] = i ? a2.y : a2.x; ev.k3[i + 2] = i ? a3.y : a3.x;
                ev.k4[i + 2] = i ? a4.y : a4.x; ev.k5[i + 2] = i ? a5.y : a5.x; ev.k6[i + 2] = i ? a6.y : a6.x;
            }
            ev.mask = mask; ev.s_w = s_w; ev.n_rk = n_rk;
            solve_event(ev, o);
            o.path = path; o.probe_err = probe_err; o.probe_ep = probe_ep; o.probe_ev = probe_ev;
            return kRkEvent;
        }
        // (g keeps its values from the start of the env-step: a lane that goes on has seen no sign change and no zero, so every
        //  g_k still has the sign it started with, and only that sign enters the test above)
        if (attempts == 1) { SG_STAMP(10); }
        t = t_new; X = Xn; Y = Yn; Xd = Xdn; Yd = Ydn; vx = vxn; vy = vyn;
        k0[0] = vxn; k0[1] = vyn; k0[2] = a6.x; k0[3] = a6.y;
        if (!(t < t_end) || attempts >= kMaxRkAttempts) return kRkFinished;
        return kRkContinue;
    }

    // All attempts of one env-step.  The result of a lane that reaches the end of the step is read out of the integrator once,
    // after the loop, not in whichever pass of the loop the lane happens to finish.
    template <typename SINK = NoSink>
    SG_MFN int run(StepResult &o, SINK &&sink = NoSink()) {
        int status = kRkContinue;
        if constexpr (!ACCEL) {  // Steering.velocity: the step tried first is the fast one (outside the loop over the attempts:
            if (fast) {          //  what that loop carries from one attempt to the next is still what begin() has set)
                fast = false;
                status = fast_step(o, sink);
            }
        }
        if (status == kRkContinue)
            while ((status = attempt(o, sink)) == kRkContinue) {}
        if (status == kRkFinished) {
            finish(o);
            // (a probe step that ended beyond a terminal surface makes the env-step terminal -- the kernels that hand such steps
            //  over have restarted the env by the time the terminal state is worked out.  Should scipy's sequence end a hair on
            //  the other side of the surface, its end state is the terminal state: no root to look for.)
            if (probe_crossing) o.done = 1;
        }
        return status;
    }

    // Only the step that is tried first (the fast step; Steering.acceleration: the probe step), for callers that hand every
    // env-step it does not settle to somebody else (the one-launch-per-step kernels: lanes whose env-step is not a kept first
    // step are compacted across the workgroup and replayed there by begin(use_probe = false) + run(), which is the same
    // arithmetic as going on from here).  kRkFinished: kept, `o` is the result; kRkEventDeferred: the env-step is terminal
    // (o.done = 1; the terminal state is on scipy's sequence); kRkContinue: scipy's sequence decides.  `sink` must return true.
    template <typename SINK>
    SG_MFN int first(StepResult &o, SINK &&sink) {
        int status;
        if (!ACCEL && fast) { fast = false; status = fast_step(o, sink); }
        else status = attempt(o, sink);
        if (status == kRkFinished) finish(o);
        return status;
    }

    // solve_ivp's event handling for an accepted step with sign changes (ivp.py:673-694): the earliest root over the
    // step's 4th-order dense output (rk.py:178-192) becomes the end of the env-step.  Needs set_constants() only.
    SG_MFN void solve_event(const EventCase &ev, StepResult &o) const {
        const float h = ev.h, t = ev.t, X = ev.X, Y = ev.Y, vx = ev.vx, vy = ev.vy, Xn = ev.Xn, Yn = ev.Yn, s_w = ev.s_w;
        const double Xd = ev.Xd, Yd = ev.Yd;
        const unsigned mask = ev.mask;
        const float(&k0)[4] = ev.k0, (&k2)[4] = ev.k2, (&k3)[4] = ev.k3, (&k4)[4] = ev.k4, (&k5)[4] = ev.k5, (&k6)[4] = ev.k6;
        // event functions at both ends of the step (X = Y = 0 at t = 0); attempt() carries the circles' in squared form
        float g[NC + 2], gn[NC + 2];
#pragma unroll
        for (int k = 0; k < NC; k++) {
            const f2 c = cq_at(k);
            const float ax_ = c.x - X, ay_ = c.y - Y, bx_ = c.x - Xn, by_ = c.y - Yn;
            g[k] = fsqrt(fmaf(ax_, ax_, ay_ * ay_)) - cR[k];
            gn[k] = fsqrt(fmaf(bx_, bx_, by_ * by_)) - cR[k];
        }
        g[NC] = g[NC + 1] = gn[NC] = gn[NC + 1] = 1.0f;
        if (WALLS) {
            g[NC] = fminf(wxp - X, wyp - Y); g[NC + 1] = fminf(wxm + X, wym + Y);
            gn[NC] = fminf(wxp - Xn, wyp - Yn); gn[NC + 1] = fminf(wxm + Xn, wym + Yn);
        }
        // dense output over [t, t_new]: y(s) = y_old + h s (K0 + s (Q1 + s (Q2 + s Q3))), s in [0, 1]
        float q1[4], q2[4], q3[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float d2_ = k2[i] - k0[i], d3_ = k3[i] - k0[i], d4_ = k4[i] - k0[i], d5_ = k5[i] - k0[i], d6_ = k6[i] - k0[i];
            q1[i] = fmaf(P71, d6_, fmaf(P61, d5_, fmaf(P51, d4_, fmaf(P41, d3_, P31 * d2_))));
            q2[i] = fmaf(P72, d6_, fmaf(P62, d5_, fmaf(P52, d4_, fmaf(P42, d3_, P32 * d2_))));
            q3[i] = fmaf(P73, d6_, fmaf(P63, d5_, fmaf(P53, d4_, fmaf(P43, d3_, P33 * d2_))));
        }
        auto disp = [&](int i, float s) __attribute__((always_inline)) { return h * s * fmaf(s, fmaf(s, fmaf(s, q3[i], q2[i]), q1[i]), k0[i]); };
        auto dispd = [&](int i, float s) __attribute__((always_inline)) {  // d disp / ds
            return h * fmaf(s, fmaf(s, fmaf(4.0f * s, q3[i], 3.0f * q2[i]), 2.0f * q1[i]), k0[i]);
        };
        // Every event function is solved through SMOOTH components: a circle is one component; a wall event
        // min(W/2 -+ x, W/2 -+ y) (dynamic_model.py:196-205) has a kink, so its x and y parts are solved separately
        // and combined by the min semantics: leaving the world (g: + -> -) the first component to cross wins,
        // entering it (g: - -> +, injected states only) the last one does.
        float best = 2.0f;
        int best_k = -1, best_comp = 0;
        float bax = 0.0f, bay = 0.0f;  // absolute centre of the winning circle event
        double bRd = 0.0;
        unsigned m = mask & ((1u << (NC + 2)) - 1u);  // circle and wall events; the omega event is handled after them
        while (m) {  // usually one active event; all are terminal -> the earliest root wins (ivp.py:115-126)
            const int k = __builtin_ctz(m);
            m &= m - 1;
            // per-lane event description (selects, no dynamic register indexing)
            float ecx = 0.0f, ecy = 0.0f, eR = 0.0f, eax = 0.0f, eay = 0.0f, g0 = 0.0f, g1 = 0.0f;
            double eRd = 0.0;
#pragma unroll
            for (int j = 0; j < NC + (WALLS ? 2 : 0); j++)
                if (j == k) {
                    g0 = g[j]; g1 = gn[j];
                    if (j < NC) { const f2 c = cq_at(j); ecx = c.x; ecy = c.y; eR = cR[j]; eax = cax[j]; eay = cay[j]; eRd = cRd[j]; }
                }
            const bool circle = k < NC, leaving = g0 > 0.0f;
            const float sgn = (k == NC) ? 1.0f : -1.0f;                      // world_max : world_min
            const float wx = (k == NC) ? wxp : wxm, wy = (k == NC) ? wyp : wym;
            // Two slots solved side by side (independent chains -> ILP for the lone wave):
            //   slot A: the circle, or the x part of a wall event;   slot B: the y part of a wall event.
            // Safeguarded Newton from the regula-falsi point; the parts are nearly linear in s over one RK step.
            float gaA, gbA, gaB = 1.0f, gbB = 1.0f;
            if (circle) { gaA = g0; gbA = g1; }
            else { gaA = wx - sgn * X; gbA = wx - sgn * Xn; gaB = wy - sgn * Y; gbB = wy - sgn * Yn; }
            const bool crossA = gaA * gbA <= 0.0f;  // sign change or a zero at either end (as for `mask`)
            const bool crossB = !circle && gaB * gbB <= 0.0f;
            float loA = 0.0f, hiA = 1.0f, loB = 0.0f, hiB = 1.0f;
            float denA = gaA - gbA, denB = gaB - gbB;
            float sA = (denA != 0.0f) ? fminf(fmaxf(gaA * rcp(denA), 0.0f), 1.0f) : 0.0f;
            float sB = (denB != 0.0f) ? fminf(fmaxf(gaB * rcp(denB), 0.0f), 1.0f) : 0.0f;
            for (int it = 0; it < kRootMaxIters; it++) {
                const float pA = sA, pB = sB;
                if (crossA) {  // slot A: circle or x part (branches are skipped wave-wide when no lane needs them)
                    float gA, gpA;
                    const float dxA = X + disp(0, sA), uxA = dispd(0, sA);
                    if (circle) {
                        const float dyA = Y + disp(1, sA), uyA = dispd(1, sA);
                        const float exA = ecx - dxA, eyA = ecy - dyA;
                        const float r2A = fmaf(exA, exA, eyA * eyA), irA = rsq(r2A);
                        gA = r2A * irA - eR;
                        gpA = -(exA * uxA + eyA * uyA) * irA;
                    } else {
                        gA = wx - sgn * dxA;
                        gpA = -sgn * uxA;
                    }
                    if ((gA > 0.0f) == (gaA > 0.0f)) loA = sA; else hiA = sA;   // g(lo) keeps the sign of g(0)
                    const float nA = sA - gA * rcp(gpA);
                    sA = (nA > loA && nA < hiA) ? nA : ((gA == 0.0f) ? sA : 0.5f * (loA + hiA));
                }
                if (crossB) {  // slot B: y part of a wall event
                    const float dyB = Y + disp(1, sB), uyB = dispd(1, sB);
                    const float gB = wy - sgn * dyB, gpB = -sgn * uyB;
                    if ((gB > 0.0f) == (gaB > 0.0f)) loB = sB; else hiB = sB;
                    const float nB = sB - gB * rcp(gpB);
                    sB = (nB > loB && nB < hiB) ? nB : ((gB == 0.0f) ? sB : 0.5f * (loB + hiB));
                }
                // well-conditioned roots settle in 2-3 passes; near-tangent grazes (slope ~ 0) keep going, bisecting
                const bool convA = !crossA || fabsf(sA - pA) <= 1e-6f, convB = !crossB || fabsf(sB - pB) <= 1e-6f;
                if (it >= kRootIters - 1 && convA && convB) break;
            }
            // the event function is the circle, or min(x part, y part): leaving the world the first part to cross
            // wins, entering it (injected states only) the last one does
            float root;
            int root_comp = 0;
            if (circle) root = sA;
            else if (crossA && crossB) { const bool pickB = leaving ? (sB < sA) : (sB > sA); root = pickB ? sB : sA; root_comp = pickB; }
            else if (crossB) { root = sB; root_comp = 1; }
            else root = crossA ? sA : (leaving ? 2.0f : -1.0f);
            if (root < best && root >= 0.0f) { best = root; best_k = k; best_comp = root_comp; bax = eax; bay = eay; bRd = eRd; }
        }
        // One Newton step on the winning component with g evaluated in fp64 from the unrounded inputs: the Goal
        // reward multiplies the terminal position by up to 1000 (goal.py:147-152), so fp32 noise in g (~1e-7)
        // would show.  The slope only needs a few digits.
        if (best_k >= 0) {
            const float s = best;
            const float dx = X + disp(0, s), dy = Y + disp(1, s), ux = dispd(0, s), uy = dispd(1, s);
            double gd;
            float gp;
            if (best_k < NC) {
                const double ex = ((double)bax - (double)x0) - (double)dx, ey = ((double)bay - (double)y0) - (double)dy;
                const double r2 = ex * ex + ey * ey, ir = rsqrt_f64(r2);
                gd = r2 * ir - bRd;
                gp = -((float)ex * ux + (float)ey * uy) * (float)ir;
            } else {
                const double sg = (best_k == NC) ? 1.0 : -1.0, hw = (double)half_world;
                gd = best_comp ? hw - sg * ((double)y0 + (double)dy) : hw - sg * ((double)x0 + (double)dx);
                gp = (float)(-sg) * (best_comp ? uy : ux);
            }
            if (fabsf(gp) > 1e-12f) best = fminf(fmaxf(s - (float)gd * rcp(gp), 0.0f), 1.0f);
        }
        if (s_w < best) { best = s_w; best_k = NC + 2; }  // the omega event comes last in the reference's event order
        const float s = best;
        // leading term h s v in fp64, the O(h^2) remainder in fp32
        const double hs = (double)h * (double)s;
        o.dXd = Xd + (hs * (double)vx + (double)(h * s * s * fmaf(s, fmaf(s, q3[0], q2[0]), q1[0])));
        o.dYd = Yd + (hs * (double)vy + (double)(h * s * s * fmaf(s, fmaf(s, q3[1], q2[1]), q1[1])));
        o.dX = (float)o.dXd; o.dY = (float)o.dYd;
        o.vx = vx + disp(2, s); o.vy = vy + disp(3, s);
        o.t = fmaf(h, s, t);
        o.dth = phase(o.t); o.om = ACCEL ? fmaf(alpha, o.t, om) : om;
        o.done = 1; o.event = best_k; o.n_rk = ev.n_rk;
    }

    SG_MFN void finish(StepResult &o) const {
        o.dXd = Xd; o.dYd = Yd; o.dX = X; o.dY = Y; o.vx = vx; o.vy = vy; o.t = t;
        o.dth = phase(t); o.om = ACCEL ? fmaf(alpha, t, om) : om;
        o.done = 0; o.event = -1; o.n_rk = n_rk; o.path = path; o.probe_err = probe_err; o.probe_ep = probe_ep; o.probe_ev = probe_ev;
    }
};

template <int NC, int NG, bool WALLS, bool ACCEL = false>
SG_FN void make_step(float h_total, float half_world, float gm, float F, float om, float alpha, float w_limit, float x0,
                     float y0, float th0, float vx0, float vy0, const float (&cax)[NC], const float (&cay)[NC],
                     const float (&cR)[NC], const double (&cRd)[NC], StepResult &o, bool use_probe = true) {
    Integrator<NC, NG, WALLS, ACCEL> I;
    I.begin(h_total, half_world, gm, F, om, alpha, w_limit, x0, y0, th0, vx0, vy0, cax, cay, cR, cRd, use_probe);
    I.run(o);
}

// ------------------------------------------------------------------------------------------------
// Observation: spaceship_env.py:113-140.  The reference builds the lidar as
// unit(atan2(v)) * (|v| - r) * 2 / W, which equals v * (1 - r/|v|) * 2/W without any trigonometry.
// ------------------------------------------------------------------------------------------------
SG_FN void lidar(float vx, float vy, float r, float two_over_w, float &ox, float &oy) {
    float r2 = fmaf(vx, vx, vy * vy);
    float k = (r2 > 0.0f) ? (1.0f - r * rsq(r2)) * two_over_w : 0.0f;
    ox = vx * k;
    oy = vy * k;
}

// ------------------------------------------------------------------------------------------------
// Goal reward (goal.py:147-158).  The reference amplifies position differences x500 / x1000, so the
// distances are formed in fp64 from the UNROUNDED end position x0 + dX:
//   |last - c| - |cur - c| = (|last - c|^2 - |cur - c|^2) / (|last - c| + |cur - c|)
// with the numerator in fp64 and the denominator from fp32 square roots refined by one Newton step.
// ------------------------------------------------------------------------------------------------
SG_FN double dist_drop(double lx, double ly, double cx, double cy, double &cur2) {
    double l2 = lx * lx + ly * ly;
    cur2 = cx * cx + cy * cy;
    double s = l2 * rsqrt_f64(l2 > 0 ? l2 : 1.0) + cur2 * rsqrt_f64(cur2 > 0 ? cur2 : 1.0);  // |last| + |cur|
    if (!(s > 0)) return 0.0;
    double inv = (double)rcp((float)s);  // fp32 seed + one Newton step instead of an fp64 division (~2e-14 relative)
    inv = inv * (2.0 - s * inv);
    return (l2 - cur2) * inv;
}

// (CFG: the parameter block, or a copy of the fields used here that a K-step loop keeps in registers: GoalStepConsts)
template <int N, typename CFG>
SG_FN float goal_reward(const CFG &c, float x0, float y0, double dX, double dY, const float (&px)[N],
                        const float (&py)[N], float gx, float gy, int &hit) {
    const double lx = x0, ly = y0, cx = lx + dX, cy = ly + dY;
    double gcur2;
    double goal_drop = dist_drop((double)gx - lx, (double)gy - ly, (double)gx - cx, (double)gy - cy, gcur2);
    // _safety_reward_simple2 (goal.py:204-227): planet whose CENTRE is nearest to the new position, first wins ties
    int closest = 0;
    float best = 3.0e38f;
    const float fx = (float)cx, fy = (float)cy;
#pragma unroll
    for (int j = 0; j < N; j++) {
        float ex = fx - px[j], ey = fy - py[j];
        float d2 = fmaf(ex, ex, ey * ey);
        if (d2 < best) { best = d2; closest = j; }
    }
    float pcx = px[0], pcy = py[0];
#pragma unroll
    for (int j = 1; j < N; j++)
        if (j == closest) { pcx = px[j]; pcy = py[j]; }
    double pcur2;
    double planet_drop = dist_drop(lx - (double)pcx, ly - (double)pcy, cx - (double)pcx, cy - (double)pcy, pcur2);
    double safety = 0.0;
    if (pcur2 < c.danger_r2 && planet_drop > 0.0) safety = -planet_drop;  // goal.py:221-225
    double reward = c.survival + c.goal_scale * goal_drop + c.safety_scale * safety;
    hit = gcur2 < c.goal_r2;  // goal.py:154
    if (hit) reward += c.sparse;
    return (float)reward;
}

// ------------------------------------------------------------------------------------------------
// Kepler reward (kepler.py:111-156): fp64 on the unrounded end state; orbit (a, ecc, phi) may be per env.
// reward = C / (Cr |cur_rad - target_rad| + |Vx - vx| + |Vy - vy| + Ca ||action|| + C) has slope -r^2/C,
// i.e. up to 100 at r ~ 1, which is why fp32 state rounding is kept out of it.
// ------------------------------------------------------------------------------------------------
struct Orbit { double a, b, c, ecc, cosphi, sinphi, a_over_b, b_over_a, inv_a; };

// (CFG: the parameter block, or KeplerStepConsts -- the fields used here and in kepler_observe, kept in registers by a K-step loop)
template <typename CFG>
SG_FN float kepler_reward(const CFG &c, const Orbit &ob, float x0, float y0, double dX, double dY, float vx, float vy,
                          float engine, float thruster) {
    const double x = (double)x0 + dX, y = (double)y0 + dY;
    // _rotate(pos, phi) (kepler.py:51-58), then shift by the focal distance (kepler.py:68-73)
    double wx = ob.cosphi * x + ob.sinphi * y - ob.c, wy = -ob.sinphi * x + ob.cosphi * y;
    double w2 = wx * wx + wy * wy;
    double iw = rsqrt_f64(w2);
    double cur_rad = w2 * iw;                       // _orbit_cur_rad (kepler.py:90-96)
    double ect = ob.ecc * wx * iw;                  // ecc * cos(theta), theta = atan2(wy, wx)
    double target_rad = ob.b * rsqrt_f64(1.0 - ect * ect);  // kepler.py:75,109
    double sc = target_rad * iw;
    double px_ = wx * sc, py_ = wy * sc;            // projection onto the orbit
    double tx = -ob.a_over_b * py_, ty = ob.b_over_a * px_;  // kepler.py:77-78 (curl = 1)
    double rx = px_ + ob.c;
    double ir = rsqrt_f64(rx * rx + py_ * py_);     // 1 / r
    double v2 = c.k_gm * (2.0 * ir - ob.inv_a);   // _orbit_vel^2 (kepler.py:60-62)
    double speed = v2 > 0 ? v2 * rsqrt_f64(v2) : 0.0;
    double it = rsqrt_f64(tx * tx + ty * ty);
    double ux = tx * it * speed, uy = ty * it * speed;
    double Vx = ob.cosphi * ux - ob.sinphi * uy, Vy = ob.sinphi * ux + ob.cosphi * uy;  // _rotate(Vt, -phi)
    double rad_pen = fabs(cur_rad - target_rad);
    double vxp = fabs(Vx - (double)vx), vyp = fabs(Vy - (double)vy);
    // np.linalg.norm(last_action) on the float32 translated action, times act_penalty_C, in float32
    float act = c.k_Ca * fsqrt(fmaf(engine, engine, thruster * thruster));
    return (float)(c.k_C / (c.k_Cr * rad_pen + vxp + vyp + (double)act + c.k_C));
}

SG_FN Orbit make_orbit(double a, double ecc, double cosphi, double sinphi) {
    Orbit ob;
    ob.a = a; ob.ecc = ecc; ob.cosphi = cosphi; ob.sinphi = sinphi;
    double b2 = a * a * (1.0 - ecc * ecc);
    ob.b = b2 * rsqrt_f64(b2);                      // _b (kepler.py:43-45)
    double c2 = a * a - b2;
    ob.c = c2 > 0 ? c2 * rsqrt_f64(c2) : 0.0;       // _c (kepler.py:47-49)
    ob.a_over_b = a / ob.b; ob.b_over_a = ob.b / a; ob.inv_a = 1.0 / a;
    return ob;
}

// ------------------------------------------------------------------------------------------------
// Counter-based RNG: Philox4x32-10 (Salmon et al., SC'11) keyed by the seed, counter =
// (global env index, episode, block, stream); a reset expands its four words with xoshiro128++.
// ------------------------------------------------------------------------------------------------
SG_FN uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

SG_FN void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t h0 = mulhi32(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t h1 = mulhi32(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        c0 = h1 ^ c1 ^ k0; c1 = l1;
        c2 = h0 ^ c3 ^ k1; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

constexpr uint32_t kStreamReset = 0u, kStreamGoal = 1u, kStreamAction = 2u;

// Reset words (RNG): word k of an episode = component k%4 of Philox block k/4 of the reset stream, so the
// blocks can be generated by different lanes at once (cooperative restart in sg_engine.hip) or one after the other.
//   Goal   block 0: flags, col01, col23, tiles01            block 1: tiles23, tiles45, goal_c01, goal_c2
//          block 2: disc_ship, disc_p0, disc_p1, disc_p2    block 3: disc_p3, disc_goal, theta, -
//          block 4: bm1_u1, bm1_u2, bm2_u1, bm2_u2
//   Kepler block 0: angle, dist, theta, ecc   block 1: orbit_angle, -, -, -   block 2: bm1_u1, bm1_u2, bm2_u1, bm2_u2
constexpr int kGoalResetBlocks = 5, kKeplerResetBlocks = 3;

SG_FN float u23(uint32_t w) { return ((float)(w >> 9) + 0.5f) * (1.0f / 8388608.0f); }       // (0,1), exact
SG_FN float u16(uint32_t h) { return ((float)(h & 0xffffu) + 0.5f) * (1.0f / 65536.0f); }    // (0,1), exact
SG_FN uint32_t below16(uint32_t h, uint32_t n) { return ((h & 0xffffu) * n) >> 16; }

// Per-episode tiling layout + the free-tile multiset (hexagonal_tiling.py:42-43,62-72,91).
// The free list of the reference is an ordered list that can hold duplicates (:101-106); candidates are
// drawn by POSITION without replacement and ties go to the first DRAWN, so only the multiset matters.
// It is kept as sixteen 4-bit counters (saturating at 15) indexed by tile number, positions taken in
// ascending tile order.
struct Tiling {
    uint32_t episode, goal_draws;
    uint32_t ship_tile, goal_tile, case_b, flip;
    uint64_t free_counts;
    float cs0, cs1, cs2, cs3;  // column shifts; scalars, not an array: a runtime-indexed member would push the struct to scratch
};

SG_FN uint32_t free_total(uint64_t f) {
    uint64_t s = (f & 0x0f0f0f0f0f0f0f0full) + ((f >> 4) & 0x0f0f0f0f0f0f0f0full);
    return (uint32_t)((s * 0x0101010101010101ull) >> 56);
}
// The multiset as running totals, for looking up the tile at a position without loops or branches (a goal resample looks up
// three; with a loop over the set counters and a 16-step scan for multisets with duplicates -- both unrolled three times --
// these look-ups were more than half of the resampling code, which a whole wave runs for the one lane that reached its goal):
//   even[k] (byte k) = number of entries in tiles 0 .. 2k,   odd[k] = number of entries in tiles 0 .. 2k + 1   (each <= 240)
struct FreePrefix { uint64_t even, odd; };
SG_FN FreePrefix free_prefix(uint64_t f) {
    const uint64_t e = f & 0x0f0f0f0f0f0f0f0full, o = (f >> 4) & 0x0f0f0f0f0f0f0f0full;
    const uint64_t pe = e * 0x0101010101010101ull, po = o * 0x0101010101010101ull;  // byte k = sum of bytes 0..k (<= 120: no carries)
    FreePrefix p;
    p.even = pe + (po << 8);
    p.odd = pe + po;
    return p;
}
// number of bytes of x that are > pos (pos <= 254; bytes up to 255), by SWAR: a byte is > pos if its high bit says so, or the
// high bits agree and its low seven bits are > pos's
SG_FN uint32_t bytes_above(uint64_t x, uint32_t pos) {
    const uint64_t H = 0x8080808080808080ull, L = 0x0101010101010101ull;
    const uint64_t low_gt = (((x & ~H) + (uint64_t)(127u - (pos & 127u)) * L) & H);  // low seven bits > pos & 127 (no carries: <= 254)
    const uint64_t hi = x & H;
    const uint64_t gt = (pos < 128u) ? (hi | low_gt) : (hi & low_gt);
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popcll(gt);
#else
    return (uint32_t)__builtin_popcountll(gt);
#endif
}
// tile at position `pos` of the sorted multiset (pos < total): the number of tiles whose running total is <= pos
SG_FN uint32_t free_at(const FreePrefix &p, uint32_t pos) { return 16u - bytes_above(p.even, pos) - bytes_above(p.odd, pos); }
SG_FN uint64_t free_add(uint64_t f, uint32_t tile) {
    uint32_t cnt = (uint32_t)(f >> (4 * tile)) & 15u;
    return cnt < 15u ? f + (1ull << (4 * tile)) : f;
}
SG_FN uint64_t free_remove(uint64_t f, uint32_t tile) { return f - (1ull << (4 * tile)); }

// hexagonal_tiling.py:136-158 _tile_center_pos
SG_FN void tile_center(const SgDev &c, const Tiling &T, uint32_t tile, float &x, float &y) {
    uint32_t row = (tile * c.t_cols_rcp16) >> 16, col = tile - row * (uint32_t)c.t_cols;
    const float shift = (col == 1u) ? T.cs1 : (col == 2u) ? T.cs2 : (col == 3u) ? T.cs3 : T.cs0;
    float tx = c.t_x0 + (float)col * 1.5f * c.t_a + shift;
    float y0 = T.case_b ? c.t_y0 - 0.5f * c.t_hex_h : c.t_y0;
    float ycol = (col & 1u) ? 0.5f * c.t_hex_h : 0.0f;
    float ty = y0 - (float)row * c.t_hex_h + (T.case_b ? ycol : -ycol);
    x = T.flip ? ty : tx;
    y = T.flip ? tx : ty;
}

// helpers.py:48-53 uniform_disk_distribution inside a tile (hexagonal_tiling.py:130-134); word = angle:hi16 | r:lo16
SG_FN void disc_in_tile(const SgDev &c, const Tiling &T, uint32_t tile, float noise_radius, uint32_t w, float &x, float &y) {
    float s, co;
    sincos_acc(kTwoPi * u16(w >> 16), s, co);
    float r = noise_radius * fsqrt(u16(w));
    tile_center(c, T, tile, x, y);
    x = fmaf(r, co, x);
    y = fmaf(r, s, y);
}

// hexagonal_tiling.py:99-128 _reset_goal_tile_nr.  w = [gate, cand0:cand1, cand2:-, disc]
SG_FN void choose_goal_tile(const SgDev &c, Tiling &T, bool first, const uint32_t (&w)[4]) {
    if (!first) {  // :101-106 the ship now sits on the tile of the goal it reached
        T.free_counts = free_add(T.free_counts, T.ship_tile);
        T.ship_tile = T.goal_tile;
    }
    if ((w[0] & 0xffu) < 64u) {  // uniform() < 0.25, :108-110
        T.goal_tile = T.ship_tile;
    } else {
        const uint32_t n = free_total(T.free_counts), n_cand = n < 3u ? n : 3u;
        const uint32_t cols = (uint32_t)c.t_cols, sr = (T.ship_tile * c.t_cols_rcp16) >> 16, sc = T.ship_tile - sr * cols;
        const uint32_t draws[3] = {w[1] >> 16, w[1], w[2] >> 16};
        // choice(n, size=n_cand, replace=False) as a partial Fisher-Yates over positions 0..n-1 (slot i swaps with
        // slot j_i >= i).  With at most three draws the touched slots are tracked by value, not in an array:
        //   after swap 0: slot 0 holds j0 and slot j0 holds 0; after swap 1: slot j1 holds what slot 1 held.
        uint32_t chosen[3] = {0u, 0u, 0u};
        const uint32_t j0 = below16(draws[0], n);
        chosen[0] = j0;
        const uint32_t v1_old = (j0 == 1u) ? 0u : 1u;  // content of slot 1 before swap 1
        uint32_t j1 = 0xffffffffu;
        if (n_cand > 1) {
            j1 = 1u + below16(draws[1], n - 1u);
            chosen[1] = (j1 == j0) ? 0u : j1;
        }
        if (n_cand > 2) {
            const uint32_t j2 = 2u + below16(draws[2], n - 2u);
            chosen[2] = (j2 == j1) ? v1_old : (j2 == j0) ? 0u : j2;
        }
        uint32_t best_tile = 0;
        int best_dist = -1;
        const FreePrefix fp = free_prefix(T.free_counts);
#pragma unroll
        for (uint32_t i = 0; i < 3; i++) {  // fully unrolled: keeps chosen[] in registers
            if (i < n_cand) {
                uint32_t tile = free_at(fp, chosen[i]);
                uint32_t r = (tile * c.t_cols_rcp16) >> 16, cc = tile - r * cols;
                int dist = abs((int)r - (int)sr) + abs((int)cc - (int)sc);  // :119-121
                if (dist > best_dist) { best_dist = dist; best_tile = tile; }  // first max, :122-124
            }
        }
        T.goal_tile = best_tile;
        T.free_counts = free_remove(T.free_counts, best_tile);  // pop, :126
    }
}

// hexagonal_tiling.py:95-97 find_new_goal: goal tile, then a uniform disc inside it (:130-134)
SG_FN void choose_goal(const SgDev &c, Tiling &T, bool first, const uint32_t (&w)[4], float &gx, float &gy) {
    choose_goal_tile(c, T, first, w);
    disc_in_tile(c, T, T.goal_tile, c.noise_goal, w[3], gx, gy);
}

SG_FN void box_muller(uint32_t w1, uint32_t w2, float &z0, float &z1) {
    float r = fsqrt(-2.0f * 0.6931471805599453f * flog2(u23(w1)));
    float s, co;
    sincos_acc(kTwoPi * u23(w2), s, co);
    z0 = r * co;
    z1 = r * s;
}

struct ShipInit { float x, y, th, vx, vy, om; };

// GoalEnv._reset (goal.py:133-145) + HexagonalTiling.reset (hexagonal_tiling.py:53-93)
// Pieces of GoalEnv._reset (goal.py:133-145) + HexagonalTiling.reset (hexagonal_tiling.py:53-93), shared by the serial
// reset below and the lane-cooperative restart in sg_engine.hip.
SG_FN void layout_from_words(const SgDev &c, uint32_t flags, uint32_t c01, uint32_t c23, Tiling &T) {
    T.case_b = flags & 1u; T.flip = (flags >> 1) & 1u;  // hexagonal_tiling.py:69
    // :70 cumsum over the first t_cols (2..4) columns
    const float a0 = u16(c01 >> 16), a1 = a0 + u16(c01);
    const float a2 = (c.t_cols > 2) ? a1 + u16(c23 >> 16) : a1, a3 = (c.t_cols > 3) ? a2 + u16(c23) : a2;
    const float k = c.t_free_x * rcp(a3);  // :71-72 (a3 is the last real column's cumulative sum)
    T.cs0 = a0 * k; T.cs1 = a1 * k;
    T.cs2 = (c.t_cols > 2) ? a2 * k : 0.0f; T.cs3 = (c.t_cols > 3) ? a3 * k : 0.0f;
}

// tiles[0] = ship, tiles[1..N] = planets; also the free-tile multiset and T.ship_tile
template <int N>
SG_FN void tiles_from_words(const SgDev &c, uint32_t flags, uint32_t t01, uint32_t t23, uint32_t t45, Tiling &T,
                            uint32_t (&tiles)[N + 1]) {
    const uint32_t draws[6] = {t01 >> 16, t01, t23 >> 16, t23, t45 >> 16, t45};
    // choice(n_tiles, size=N+1, replace=False) (:89): Fisher-Yates on a 16-nibble permutation word
    uint64_t perm = 0xfedcba9876543210ull;
    const uint32_t nt = (uint32_t)c.t_tiles;
#pragma unroll
    for (uint32_t i = 0; i <= (uint32_t)N; i++) {
        uint32_t j = i + below16(draws[i], nt - i);
        uint64_t vi = (perm >> (4 * i)) & 15ull, vj = (perm >> (4 * j)) & 15ull;
        perm = (perm & ~((15ull << (4 * i)) | (15ull << (4 * j)))) | (vj << (4 * i)) | (vi << (4 * j));
        tiles[i] = (uint32_t)vj;
    }
    if (N == 2 && ((flags >> 8) & 0xffu) < 64u) {  // :75-87 the four diagonal layouts, w.p. 0.25
        const uint32_t d = (flags >> 16) & 3u;
        tiles[0] = (d == 0) ? 1u : (d == 1) ? 2u : (d == 2) ? 0u : 3u;
        tiles[1] = (d < 2) ? 0u : 1u;
        tiles[2] = (d < 2) ? 3u : 2u;
    }
    uint64_t used = 0;
#pragma unroll
    for (int i = 0; i <= N; i++) used |= 1ull << (4 * tiles[i]);
    const uint64_t all = (nt >= 16u) ? 0x1111111111111111ull : (0x1111111111111111ull & ((1ull << (4 * nt)) - 1ull));
    T.free_counts = all & ~used;  // :91
    T.ship_tile = tiles[0];       // :90
}

SG_FN void kinematics_from_words(const SgDev &c, uint32_t wth, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, ShipInit &s) {
    s.th = kTwoPi * u23(wth);  // goal.py:140
    float z0, z1, z2, z3;
    box_muller(b0, b1, z0, z1);
    box_muller(b2, b3, z2, z3);
    s.vx = z0 * c.vel_std; s.vy = z1 * c.vel_std;                               // goal.py:141 / kepler.py:261
    s.om = fminf(fmaxf(z2 * c.omega_std, -c.omega_max), c.omega_max);           // goal.py:142-144 / kepler.py:263-265
}

// GoalEnv._reset, one lane does everything (reset kernel, host twin)
template <int N>
SG_FN void goal_reset(const SgDev &c, uint32_t env_global, Tiling &T, ShipInit &s, float (&px)[N], float (&py)[N],
                      float &gx, float &gy) {
    uint32_t w[4 * kGoalResetBlocks];
#pragma unroll
    for (uint32_t b = 0; b < (uint32_t)kGoalResetBlocks; b++) {
        uint32_t o[4];
        philox4x32_10(c.seed_lo, c.seed_hi, env_global, T.episode, b, kStreamReset, o);
        w[4 * b] = o[0]; w[4 * b + 1] = o[1]; w[4 * b + 2] = o[2]; w[4 * b + 3] = o[3];
    }
    T.goal_draws = 0;
    const uint32_t flags = w[0];
    layout_from_words(c, flags, w[1], w[2], T);
    uint32_t tiles[N + 1];
    tiles_from_words<N>(c, flags, w[3], w[4], w[5], T, tiles);
    disc_in_tile(c, T, tiles[0], c.noise_ship, w[8], s.x, s.y);  // :92-93
#pragma unroll
    for (int j = 0; j < N; j++) disc_in_tile(c, T, tiles[j + 1], c.noise_planet, w[9 + j], px[j], py[j]);
    const uint32_t gw[4] = {flags >> 24, w[6], w[7], w[13]};
    T.goal_tile = 0xffu;
    choose_goal(c, T, true, gw, gx, gy);  // goal.py:138
    kinematics_from_words(c, w[14], w[16], w[17], w[18], w[19], s);
}

// GoalEnv._resample_goal on a hit (goal.py:154-157 -> hexagonal_tiling.py:95-134)
SG_FN void goal_resample(const SgDev &c, uint32_t env_global, Tiling &T, float &gx, float &gy) {
    uint32_t w[4];
    T.goal_draws += 1;
    philox4x32_10(c.seed_lo, c.seed_hi, env_global, T.episode, T.goal_draws, kStreamGoal, w);
    choose_goal(c, T, false, w, gx, gy);
}

// KeplerEnv._reset (kepler.py:233-267)
SG_FN void kepler_reset(const SgDev &c, uint32_t env_global, uint32_t episode, ShipInit &s, float &phi, float &ecc) {
    uint32_t w[4 * kKeplerResetBlocks];
#pragma unroll
    for (uint32_t b = 0; b < (uint32_t)kKeplerResetBlocks; b++) {
        uint32_t o[4];
        philox4x32_10(c.seed_lo, c.seed_hi, env_global, episode, b, kStreamReset, o);
        w[4 * b] = o[0]; w[4 * b + 1] = o[1]; w[4 * b + 2] = o[2]; w[4 * b + 3] = o[3];
    }
    float sa, ca;
    sincos_acc(kTwoPi * u23(w[0]), sa, ca);
    float dist = fmaf(c.kep_rmax - c.kep_rmin, u23(w[1]), c.kep_rmin);
    s.x = ca * dist; s.y = sa * dist;
    if (c.randomize_orbit) { ecc = u23(w[3]) * 0.7f; phi = u23(w[4]) * kTwoPi; }    // kepler.py:257-259
    kinematics_from_words(c, w[2], w[8], w[9], w[10], w[11], s);
}

// Random policy on the device (sg_random_actions_device): the actions of env `env_global` at steps 2p and 2p + 1 come from one
// Philox block keyed by the caller's seed, counter (env, p lo, p hi, stream): a0, a1 = 2 u23(w) - 1 in (-1, 1) (exact in
// fp32), or for the discrete ids floor(6 w / 2^32).  A function of (seed, global env index, step) only.
SG_FN void random_action_words(uint32_t seed_lo, uint32_t seed_hi, uint32_t env_global, uint64_t step, uint32_t &w0, uint32_t &w1) {
    uint32_t o[4];
    const uint64_t p = step >> 1;
    philox4x32_10(seed_lo, seed_hi, env_global, (uint32_t)p, (uint32_t)(p >> 32), kStreamAction, o);
    w0 = (step & 1u) ? o[2] : o[0];
    w1 = (step & 1u) ? o[3] : o[1];
}

// ------------------------------------------------------------------------------------------------
// One env in registers, and SpaceshipEnv.step (spaceship_env.py:68-78) on it.
// ------------------------------------------------------------------------------------------------
template <int N>
struct GoalEnv {
    float x, y, th, vx, vy, om;
    float gx, gy;
    float px[N], py[N];
};
struct KeplerEnv {
    float x, y, th, vx, vy, om;
    float phi, ecc;  // per-env reference orbit (KeplerRandomOrbits); ignored for the fixed-orbit ids
};

// ContinuousSpaceshipEnv._translate_raw_action (spaceship_env.py:210-214), float32 like the reference.
// The reference asserts the action is inside [-1, 1]^2 (spaceship_env.py:71); the kernel clamps instead of
// trapping (the Python wrapper can validate on the host).  NaN actions clamp to -1.
SG_FN void translate_action(float &a0, float &a1, float max_engine_force, float &engine, float &F, float &om) {
    a0 = fminf(fmaxf(a0, -1.0f), 1.0f);
    a1 = fminf(fmaxf(a1, -1.0f), 1.0f);
    engine = (a0 + 1.0f) * 0.5f;
    F = engine * max_engine_force;  // dynamic_model.py:171 (float32 product)
    om = a1 * 5.0f;                 // dynamic_model.py:140
}

// omega at t = 0 and its rate for the env-step: Steering.velocity pins omega = 5 a1 (dynamic_model.py:138-141);
// Steering.acceleration keeps the state's omega and applies alpha = (a1 * max_thruster_force) / moi (:160-161,175)
// (the ACCEL kernels also serve Steering.velocity with an env-step too long for the fast step -- sg_host_config.hpp,
//  needs_general_kernels -- as the special case alpha = 0, omega = the commanded one: hence the run-time test in them)
template <bool ACCEL>
SG_FN void steering(const SgDev &c, float a1, float om_cmd, float om_state, float &om0, float &alpha) {
    if (ACCEL && c.steering_acceleration) { om0 = om_state; alpha = (a1 * c.max_thruster_force) * c.inv_moi; }
    else { om0 = om_cmd; alpha = 0.0f; }
}

// Raw action of env `idx` from the caller's action buffer: float32 [.., 2] for the continuous ids, int32 [..] for the
// discrete ones.  DiscreteSpaceshipEnv._translate_raw_action (spaceship_env.py:189-202) maps the index to
// (engine, thruster) in {0,1} x {-1,0,1}; it is returned as the raw pair (2 engine - 1, thruster) that
// translate_action() turns back into exactly that (engine, thruster).  Out-of-range indices act as 0 (the reference raises).
SG_FN void load_action(bool discrete, const void *actions, int64_t idx, float &a0, float &a1) {
    if (discrete) {
        const int k = static_cast<const int32_t *>(actions)[idx];
        const bool engine = (k == 1) | (k == 4) | (k == 5);
        a0 = engine ? 1.0f : -1.0f;
        a1 = (k == 2 || k == 4) ? -1.0f : (k == 3 || k == 5) ? 1.0f : 0.0f;
    } else {
        const float *p = static_cast<const float *>(actions) + 2 * idx;
        a0 = p[0]; a1 = p[1];
    }
}

template <int N, typename CFG>
SG_FN void goal_observe(const CFG &c, const GoalEnv<N> &e, float (&obs)[7 + 2 * N + 2]) {
    float s, co;
    sincos_acc(e.th, s, co);
    obs[0] = e.x; obs[1] = e.y; obs[2] = co; obs[3] = s; obs[4] = e.vx; obs[5] = e.vy; obs[6] = e.om;
#pragma unroll
    for (int j = 0; j < N; j++) lidar(e.px[j] - e.x, e.py[j] - e.y, c.planet_r, c.two_over_world, obs[7 + 2 * j], obs[8 + 2 * j]);
    obs[7 + 2 * N] = (e.gx - e.x) * c.two_over_world;  // goal lidar, radius 0 (spaceship_env.py:129)
    obs[8 + 2 * N] = (e.gy - e.y) * c.two_over_world;
}

template <typename CFG>
SG_FN void kepler_observe(const CFG &c, const KeplerEnv &e, float (&obs)[10]) {
    float s, co;
    sincos_acc(e.th, s, co);
    obs[0] = e.x; obs[1] = e.y; obs[2] = co; obs[3] = s; obs[4] = e.vx; obs[5] = e.vy; obs[6] = e.om;
    obs[7] = c.randomize_orbit ? e.phi : (float)c.k_phi;  // kepler.py:172-187
    obs[8] = c.randomize_orbit ? e.ecc : (float)c.k_ecc;
    obs[9] = (float)c.k_a;
}

// Goal: integrate -> observation (old goal) -> reward; the caller resamples the goal on a hit (goal.py:154-157).
// Split into begin / finish around the resumable integrator so that the rollout kernel can interleave envs.
template <int N, bool ACCEL = false>
SG_FN void goal_env_begin(const SgDev &c, const GoalEnv<N> &e, float a0, float a1, Integrator<N, N, true, ACCEL> &I,
                          bool use_probe = true) {
    float engine, F, om, om0, alpha;
    translate_action(a0, a1, c.max_engine_force, engine, F, om);
    steering<ACCEL>(c, a1, om, e.om, om0, alpha);
    float cR[N];
    double cRd[N];
#pragma unroll
    for (int j = 0; j < N; j++) { cR[j] = c.planet_r; cRd[j] = c.planet_r_d; }
    I.begin(c.h, c.half_world, c.gm, F, om0, alpha, c.omega_limit, e.x, e.y, e.th, e.vx, e.vy, e.px, e.py, cR, cRd, use_probe);
}

// What the reward and the observation of a Goal env-step read of the parameter block (goal_reward, goal_observe), by value: a
// K-step loop keeps them in registers instead of re-reading the parameter block -- scalar loads with their latency -- every step.
struct GoalStepConsts {
    double goal_r2, danger_r2, survival, goal_scale, safety_scale, sparse;
    float planet_r, two_over_world;
};
SG_FN GoalStepConsts goal_step_consts(const SgDev &c) {
    GoalStepConsts k;
    k.goal_r2 = c.goal_r2; k.danger_r2 = c.danger_r2; k.survival = c.survival; k.goal_scale = c.goal_scale;
    k.safety_scale = c.safety_scale; k.sparse = c.sparse; k.planet_r = c.planet_r; k.two_over_world = c.two_over_world;
    return k;
}

// The same for Kepler (kepler_reward, kepler_observe)
struct KeplerStepConsts {
    double k_gm, k_C, k_Cr, k_phi, k_ecc, k_a;
    float k_Ca;
    int32_t randomize_orbit;
};
SG_FN KeplerStepConsts kepler_step_consts(const SgDev &c) {
    KeplerStepConsts k;
    k.k_gm = c.k_gm; k.k_C = c.k_C; k.k_Cr = c.k_Cr; k.k_phi = c.k_phi; k.k_ecc = c.k_ecc; k.k_a = c.k_a; k.k_Ca = c.k_Ca;
    k.randomize_orbit = c.randomize_orbit;
    return k;
}

// The same from a by-value copy of the few parameters it needs: a K-step loop keeps them in registers instead of re-reading
// the parameter block every step.
struct StepConsts {
    float max_engine_force, h, half_world, gm, omega_limit, planet_r, max_thruster_force, inv_moi;
    double planet_r_d;
    int32_t steering_acceleration;  // (read by the ACCEL kernels only: see steering())
};
SG_FN StepConsts step_consts(const SgDev &c) {
    StepConsts k;
    k.max_engine_force = c.max_engine_force; k.h = c.h; k.half_world = c.half_world; k.gm = c.gm; k.omega_limit = c.omega_limit;
    k.planet_r = c.planet_r; k.max_thruster_force = c.max_thruster_force; k.inv_moi = c.inv_moi; k.planet_r_d = c.planet_r_d;
    k.steering_acceleration = c.steering_acceleration;
    return k;
}
template <int N, bool ACCEL = false>
SG_FN void goal_env_begin(const StepConsts &k, const GoalEnv<N> &e, float a0, float a1, Integrator<N, N, true, ACCEL> &I,
                          bool use_probe = true) {
    float engine, F, om, om0, alpha;
    translate_action(a0, a1, k.max_engine_force, engine, F, om);
    if (ACCEL && k.steering_acceleration) { om0 = e.om; alpha = (a1 * k.max_thruster_force) * k.inv_moi; }  // steering<ACCEL>
    else { om0 = om; alpha = 0.0f; }
    float cR[N];
    double cRd[N];
#pragma unroll
    for (int j = 0; j < N; j++) { cR[j] = k.planet_r; cRd[j] = k.planet_r_d; }
    I.begin(k.h, k.half_world, k.gm, F, om0, alpha, k.omega_limit, e.x, e.y, e.th, e.vx, e.vy, e.px, e.py, cR, cRd, use_probe);
}

template <int N>
SG_FN void goal_env_finish(const SgDev &c, GoalEnv<N> &e, const StepResult &r, float (&obs)[7 + 2 * N + 2],
                           float &reward, int &done, int &hit) {
    SG_STAMP(11);
    reward = goal_reward<N>(c, e.x, e.y, r.dXd, r.dYd, e.px, e.py, e.gx, e.gy, hit);
    SG_STAMP(12);
    e.x = (float)((double)e.x + r.dXd); e.y = (float)((double)e.y + r.dYd); e.vx = r.vx; e.vy = r.vy; e.om = r.om;
    e.th = wrap_two_pi(e.th + r.dth);
    done = r.done;
    goal_observe<N>(c, e, obs);
}

template <int N, bool ACCEL = false>
SG_FN void goal_env_step(const SgDev &c, GoalEnv<N> &e, float a0, float a1, float (&obs)[7 + 2 * N + 2], float &reward,
                         int &done, int &hit, StepResult &r) {
    Integrator<N, N, true, ACCEL> I;
    goal_env_begin<N, ACCEL>(c, e, a0, a1, I);
    I.run(r);
    goal_env_finish<N>(c, e, r, obs, reward, done, hit);
}

// SpaceshipEnv.vector_field (spaceship_env.py:96-100): RHS of the ODE at a state, [vx, vy, omega', ax, ay, alpha]
template <int NC, int NG, bool ACCEL>
SG_FN void vector_field(const SgDev &c, float x, float y, float th, float vx, float vy, float om_state, float a0, float a1,
                        const float (&cax)[NC], const float (&cay)[NC], float (&f)[6]) {
    float engine, F, om, om0, alpha, S0, C0;
    static_assert(NC <= 4, "at most four circles");
    translate_action(a0, a1, c.max_engine_force, engine, F, om);
    steering<ACCEL>(c, a1, om, om_state, om0, alpha);
    sincos_acc(th, S0, C0);
    const f2 o = mk2(x, y);
    const f2 cq0 = mk2(cax[0], cay[0]) - o, cq1 = mk2(cax[NC > 1 ? 1 : 0], cay[NC > 1 ? 1 : 0]) - o,
             cq2 = mk2(cax[NC > 2 ? 2 : 0], cay[NC > 2 ? 2 : 0]) - o, cq3 = mk2(cax[NC > 3 ? 3 : 0], cay[NC > 3 ? 3 : 0]) - o;
    f[0] = vx; f[1] = vy; f[2] = om0; f[5] = alpha;
    float r2[NG > 0 ? NG : 1];
    const f2 a = fma2(c.gm, pull<NG>(cq0, cq1, cq2, cq3, mk2(0.0f, 0.0f), r2), mk2(-(C0 * F), -(S0 * F)));
    f[3] = a.x; f[4] = a.y;
}

SG_FN Orbit fixed_orbit(const SgDev &c) {
    Orbit ob;
    ob.a = c.k_a; ob.b = c.k_b; ob.c = c.k_c; ob.ecc = c.k_ecc; ob.cosphi = c.k_cos; ob.sinphi = c.k_sin;
    ob.a_over_b = c.k_a / c.k_b; ob.b_over_a = c.k_b / c.k_a; ob.inv_a = 1.0 / c.k_a;
    return ob;
}

template <bool ACCEL = false>
SG_FN void kepler_env_step(const SgDev &c, const Orbit &ob, KeplerEnv &e, float a0, float a1, float (&obs)[10],
                           float &reward, int &done, StepResult &r, bool use_probe = true) {
    float engine, F, om, om0, alpha;
    translate_action(a0, a1, c.max_engine_force, engine, F, om);
    steering<ACCEL>(c, a1, om, e.om, om0, alpha);
    // planet (R = 0.2, gravitating) and the zero-mass border circle (R = 3, crossed from inside), both at the origin
    // (kepler.py:204-206).  The world_max/min walls at +-3 enclose the border circle and can never fire first.
    const float cax[2] = {0.0f, 0.0f}, cay[2] = {0.0f, 0.0f}, cR[2] = {c.planet_r, c.border_r};
    const double cRd[2] = {c.planet_r_d, (double)c.border_r};
    make_step<2, 1, false, ACCEL>(c.h, c.half_world, c.gm, F, om0, alpha, c.omega_limit, e.x, e.y, e.th, e.vx, e.vy, cax, cay,
                                  cR, cRd, r, use_probe);
    reward = kepler_reward(c, ob, e.x, e.y, r.dXd, r.dYd, r.vx, r.vy, engine, a1);
    e.x = (float)((double)e.x + r.dXd); e.y = (float)((double)e.y + r.dYd); e.vx = r.vx; e.vy = r.vy; e.om = r.om;
    e.th = wrap_two_pi(e.th + r.dth);
    done = r.done;
    kepler_observe(c, e, obs);
}

}  // namespace sg
#endif
