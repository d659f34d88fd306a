/*
 * sg_engine.hip -- HIP kernels (gfx950) and the C ABI of include/spacegym.h.
 *
 * HBM layout (per handle, struct of 16-byte columns, one element per env, so every wave-level access is a
 * fully coalesced 1 KiB global_load/store_dwordx4):
 *     q0[B]      float4  (x, y, theta, vx)
 *     q1[B]      float4  Goal: (vy, omega, goal_x, goal_y)        Kepler: (vy, omega, orbit_angle, orbit_ecc)
 *     pl[k][B]   float4  Goal: planets 2k and 2k+1 as (px, py, px, py), k < ceil(N/2)
 *     elapsed[B] uint32  steps taken in the current episode (gym TimeLimit)
 *   touched only on reset / goal hit (about 2% of env-steps):
 *     aux[B]     uint4   (episode, goal_draws, ship_tile | goal_tile<<8 | case_b<<16 | flip<<17, -)
 *     freec[B]   uint2   free-tile multiset, sixteen 4-bit counters (hexagonal_tiling.py:91,101-106,126)
 *     cshift[B]  float4  per-episode column shifts of the tiling (hexagonal_tiling.py:70-72)
 *     orbd[B]    double2 KeplerRandomOrbits: (cos, sin) of the per-env orbit angle
 *
 * One env per lane, 256-lane workgroups, no inter-lane communication: envs are independent
 * (gym_space/dynamic_model.py:145-165).  The step kernel fuses action translation, the RK45 step with
 * thrust + N-body gravity, termination events, heading update, observation, reward, goal resample,
 * TimeLimit and auto-reset in one launch.
 */
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/spacegym.h"
#include "sg_device.hpp"
#include "sg_host_config.hpp"

using namespace sg;

struct SgBuffers {
    float4 *q0, *q1, *pl[2];
    uint32_t *elapsed;
    uint4 *aux;
    uint2 *freec;
    float4 *cshift;
    double2 *orbd;
};

constexpr int kBlock = 256;

// ------------------------------------------------------------------------------------------------ device helpers
__device__ __forceinline__ Tiling load_tiling(const SgBuffers &b, int i) {
    Tiling T;
    const uint4 a = b.aux[i];
    const uint2 f = b.freec[i];
    const float4 cs = b.cshift[i];
    T.episode = a.x; T.goal_draws = a.y;
    T.ship_tile = a.z & 0xffu; T.goal_tile = (a.z >> 8) & 0xffu; T.case_b = (a.z >> 16) & 1u; T.flip = (a.z >> 17) & 1u;
    T.free_counts = (uint64_t)f.x | ((uint64_t)f.y << 32);
    T.col_shift[0] = cs.x; T.col_shift[1] = cs.y; T.col_shift[2] = cs.z; T.col_shift[3] = cs.w;
    return T;
}
__device__ __forceinline__ void store_tiling(const SgBuffers &b, int i, const Tiling &T) {
    b.aux[i] = make_uint4(T.episode, T.goal_draws, T.ship_tile | (T.goal_tile << 8) | (T.case_b << 16) | (T.flip << 17), 0u);
    b.freec[i] = make_uint2((uint32_t)T.free_counts, (uint32_t)(T.free_counts >> 32));
    b.cshift[i] = make_float4(T.col_shift[0], T.col_shift[1], T.col_shift[2], T.col_shift[3]);
}

template <int N>
__device__ __forceinline__ void load_goal_env(const SgBuffers &b, int i, GoalEnv<N> &e) {
    const float4 a = b.q0[i], c = b.q1[i];
    e.x = a.x; e.y = a.y; e.th = a.z; e.vx = a.w; e.vy = c.x; e.om = c.y; e.gx = c.z; e.gy = c.w;
    const float4 p0 = b.pl[0][i];
    e.px[0] = p0.x; e.py[0] = p0.y; e.px[1] = p0.z; e.py[1] = p0.w;
    if constexpr (N > 2) {
        const float4 p1 = b.pl[1][i];
        e.px[2] = p1.x; e.py[2] = p1.y;
        if constexpr (N > 3) { e.px[3] = p1.z; e.py[3] = p1.w; }
    }
}
template <int N>
__device__ __forceinline__ void store_goal_ship(const SgBuffers &b, int i, const GoalEnv<N> &e) {
    b.q0[i] = make_float4(e.x, e.y, e.th, e.vx);
    b.q1[i] = make_float4(e.vy, e.om, e.gx, e.gy);
}
template <int N>
__device__ __forceinline__ void store_goal_planets(const SgBuffers &b, int i, const GoalEnv<N> &e) {
    b.pl[0][i] = make_float4(e.px[0], e.py[0], e.px[1], e.py[1]);
    if constexpr (N == 3) b.pl[1][i] = make_float4(e.px[2], e.py[2], 0.0f, 0.0f);
    if constexpr (N == 4) b.pl[1][i] = make_float4(e.px[2], e.py[2], e.px[3], e.py[3]);
}

template <int D>
__device__ __forceinline__ void store_row(float *dst, int64_t i, const float (&v)[D]) {
    float *p = dst + i * D;
#pragma unroll
    for (int k = 0; k < D; k++) p[k] = v[k];
}

// ------------------------------------------------------------------------------------------------ Goal kernels
template <int N>
__global__ __launch_bounds__(kBlock) void goal_reset_kernel(SgDev c, SgBuffers b, int fresh, float *obs) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= c.num_envs) return;
    Tiling T;
    T.episode = fresh ? 0u : b.aux[i].x + 1u;
    GoalEnv<N> e;
    ShipInit s;
    goal_reset<N>(c, c.env_index_base + (uint32_t)i, T, s, e.px, e.py, e.gx, e.gy);
    e.x = s.x; e.y = s.y; e.th = s.th; e.vx = s.vx; e.vy = s.vy; e.om = s.om;
    store_goal_ship<N>(b, i, e);
    store_goal_planets<N>(b, i, e);
    store_tiling(b, i, T);
    b.elapsed[i] = 0u;
    if (obs) {
        float o[7 + 2 * N + 2];
        goal_observe<N>(c, e, o);
        store_row(obs, i, o);
    }
}

template <int N>
__global__ __launch_bounds__(kBlock) void goal_step_kernel(SgDev c, SgBuffers b, const float2 *__restrict__ actions,
                                                          float *__restrict__ obs, float *__restrict__ reward,
                                                          uint8_t *__restrict__ done, uint8_t *__restrict__ truncated,
                                                          float *__restrict__ terminal_obs) {
    constexpr int D = 7 + 2 * N + 2;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= c.num_envs) return;
    const float2 a = actions[i];
    GoalEnv<N> e;
    load_goal_env<N>(b, i, e);
    uint32_t el = b.elapsed[i];

    float o[D], r;
    int dn, hit;
    StepResult sr;
    goal_env_step<N>(c, e, a.x, a.y, o, r, dn, hit, sr);

    el += 1u;
    const int trunc = !dn && (int)el >= c.max_episode_steps;  // gym.wrappers.TimeLimit
    const int fin = dn | trunc;
    const bool restart = fin && c.auto_reset;
    if (hit | restart) {  // ~2% of lanes: goal resample (goal.py:154-157) or a new episode
        Tiling T = load_tiling(b, i);
        if (restart) {
            if (terminal_obs) store_row(terminal_obs, i, o);
            T.episode += 1u;
            ShipInit s;
            goal_reset<N>(c, c.env_index_base + (uint32_t)i, T, s, e.px, e.py, e.gx, e.gy);
            e.x = s.x; e.y = s.y; e.th = s.th; e.vx = s.vx; e.vy = s.vy; e.om = s.om;
            store_goal_planets<N>(b, i, e);
            goal_observe<N>(c, e, o);
            el = 0u;
        } else {
            goal_resample(c, c.env_index_base + (uint32_t)i, T, e.gx, e.gy);
        }
        store_tiling(b, i, T);
    }
    store_goal_ship<N>(b, i, e);
    b.elapsed[i] = el;
    store_row(obs, i, o);
    reward[i] = r;
    done[i] = (uint8_t)fin;
    truncated[i] = (uint8_t)trunc;
}

// ------------------------------------------------------------------------------------------------ Kepler kernels
__device__ __forceinline__ void load_kepler_env(const SgBuffers &b, int i, KeplerEnv &e) {
    const float4 a = b.q0[i], c = b.q1[i];
    e.x = a.x; e.y = a.y; e.th = a.z; e.vx = a.w; e.vy = c.x; e.om = c.y; e.phi = c.z; e.ecc = c.w;
}
__device__ __forceinline__ void store_kepler_env(const SgBuffers &b, int i, const KeplerEnv &e) {
    b.q0[i] = make_float4(e.x, e.y, e.th, e.vx);
    b.q1[i] = make_float4(e.vy, e.om, e.phi, e.ecc);
}
__device__ __forceinline__ void kepler_new_episode(const SgDev &c, const SgBuffers &b, int i, uint32_t episode, KeplerEnv &e) {
    ShipInit s;
    kepler_reset(c, c.env_index_base + (uint32_t)i, episode, s, e.phi, e.ecc);
    e.x = s.x; e.y = s.y; e.th = s.th; e.vx = s.vx; e.vy = s.vy; e.om = s.om;
    if (c.randomize_orbit) b.orbd[i] = make_double2(cos((double)e.phi), sin((double)e.phi));
    b.aux[i] = make_uint4(episode, 0u, 0u, 0u);
}

__global__ __launch_bounds__(kBlock) void kepler_reset_kernel(SgDev c, SgBuffers b, int fresh, float *obs) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= c.num_envs) return;
    KeplerEnv e;
    e.phi = (float)c.k_phi; e.ecc = (float)c.k_ecc;
    kepler_new_episode(c, b, i, fresh ? 0u : b.aux[i].x + 1u, e);
    store_kepler_env(b, i, e);
    b.elapsed[i] = 0u;
    if (obs) {
        float o[10];
        kepler_observe(c, e, o);
        store_row(obs, i, o);
    }
}

__global__ __launch_bounds__(kBlock) void kepler_step_kernel(SgDev c, SgBuffers b, const float2 *__restrict__ actions,
                                                            float *__restrict__ obs, float *__restrict__ reward,
                                                            uint8_t *__restrict__ done, uint8_t *__restrict__ truncated,
                                                            float *__restrict__ terminal_obs) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= c.num_envs) return;
    const float2 a = actions[i];
    KeplerEnv e;
    load_kepler_env(b, i, e);
    uint32_t el = b.elapsed[i];
    Orbit ob = fixed_orbit(c);
    if (c.randomize_orbit) {
        const double2 cs = b.orbd[i];
        ob = make_orbit(c.k_a, (double)e.ecc, cs.x, cs.y);
    }
    float o[10], r;
    int dn;
    StepResult sr;
    kepler_env_step(c, ob, e, a.x, a.y, o, r, dn, sr);
    el += 1u;
    const int trunc = !dn && (int)el >= c.max_episode_steps;
    const int fin = dn | trunc;
    if (fin && c.auto_reset) {
        if (terminal_obs) store_row(terminal_obs, i, o);
        kepler_new_episode(c, b, i, b.aux[i].x + 1u, e);
        kepler_observe(c, e, o);
        el = 0u;
    }
    store_kepler_env(b, i, e);
    b.elapsed[i] = el;
    store_row(obs, i, o);
    reward[i] = r;
    done[i] = (uint8_t)fin;
    truncated[i] = (uint8_t)trunc;
}

// ------------------------------------------------------------------------------------------------ state get/set
// AoS staging <-> the column layout.  mode 0: columns -> staging, 1: staging -> columns (NULL staging = skip).
__global__ __launch_bounds__(kBlock) void state_io_kernel(SgDev c, SgBuffers b, int mode, float *ship, float *planets,
                                                         float *goal, int32_t *elapsed) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= c.num_envs) return;
    const int N = c.family == SG_FAMILY_GOAL ? c.n_planets : 0;
    float4 a = b.q0[i], q = b.q1[i];
    if (mode == 0) {
        if (ship) { float *s = ship + 6 * i; s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = q.x; s[5] = q.y; }
        if (goal) { goal[2 * i] = q.z; goal[2 * i + 1] = q.w; }
        if (planets)
            for (int j = 0; j < N; j++) {
                const float4 p = b.pl[j >> 1][i];
                planets[(i * N + j) * 2] = (j & 1) ? p.z : p.x;
                planets[(i * N + j) * 2 + 1] = (j & 1) ? p.w : p.y;
            }
        if (elapsed) elapsed[i] = (int32_t)b.elapsed[i];
    } else {
        if (ship) { const float *s = ship + 6 * i; a = make_float4(s[0], s[1], s[2], s[3]); q.x = s[4]; q.y = s[5]; }
        if (goal) { q.z = goal[2 * i]; q.w = goal[2 * i + 1]; }
        b.q0[i] = a; b.q1[i] = q;
        if (planets)
            for (int k = 0; k < (N + 1) / 2; k++) {
                float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
                p.x = planets[(i * N + 2 * k) * 2]; p.y = planets[(i * N + 2 * k) * 2 + 1];
                if (2 * k + 1 < N) { p.z = planets[(i * N + 2 * k + 1) * 2]; p.w = planets[(i * N + 2 * k + 1) * 2 + 1]; }
                b.pl[k][i] = p;
            }
        if (elapsed) b.elapsed[i] = (uint32_t)elapsed[i];
        if (goal && c.family == SG_FAMILY_KEPLER && c.randomize_orbit) b.orbd[i] = make_double2(cos((double)q.z), sin((double)q.z));
    }
}

// ================================================================================================ host side
struct sg_env {
    SgDev dev;
    SgBuffers buf;
    int device;
    int fresh;  // next reset starts the episode counters at 0 (after create / seed)
    hipStream_t stream;
    // device staging for the host-buffer entry points
    float *d_actions, *d_obs, *d_reward, *d_tobs, *d_ship, *d_planets, *d_goal;
    uint8_t *d_done, *d_trunc;
    int32_t *d_elapsed;
    std::string err;
    char id[64];
    // optional per-launch timing of the step kernel (sg_set_profiling)
    int profiling;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
};

static thread_local std::string g_create_error;

static int fail(sg_env *e, int code, const char *fmt, ...) {
    char msg[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(msg, sizeof(msg), fmt, ap);
    va_end(ap);
    if (e) e->err = msg; else g_create_error = msg;
    return code;
}

#define SG_HIP(e, call)                                                                        \
    do {                                                                                       \
        hipError_t err_ = (call);                                                              \
        if (err_ != hipSuccess) return fail(e, SG_ERR_HIP, "%s: %s", #call, hipGetErrorString(err_)); \
    } while (0)

static inline int grid_for(const sg_env *e) { return (int)((e->dev.num_envs + kBlock - 1) / kBlock); }

template <typename T>
static hipError_t dmalloc(T **p, size_t n) { return hipMalloc((void **)p, n * sizeof(T)); }

extern "C" const char *sg_version(void) { return "spacegym-mi355x 0.1 (gfx950)"; }

extern "C" const char *sg_last_error(const sg_env *env) { return env ? env->err.c_str() : g_create_error.c_str(); }

extern "C" int sg_create(const sg_config *cfg, int device, sg_env **out) {
    if (!cfg || !out) return fail(nullptr, SG_ERR_INVALID, "sg_create: null argument");
    *out = nullptr;
    if (cfg->num_envs <= 0 || cfg->num_envs > (int64_t)1 << 30) return fail(nullptr, SG_ERR_INVALID, "sg_create: num_envs out of range");
    char id[64];
    std::memcpy(id, cfg->env_id, sizeof(id));
    id[63] = 0;
    SgDev d;
    if (fill_config(id, d)) return fail(nullptr, SG_ERR_INVALID, "sg_create: unknown env id '%s'", id);
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(nullptr, SG_ERR_NO_DEVICE, "sg_create: no HIP device visible; the engine has no CPU path");
    if (device < 0 || device >= n_dev) return fail(nullptr, SG_ERR_INVALID, "sg_create: device %d out of range (%d visible)", device, n_dev);
    SG_HIP(nullptr, hipSetDevice(device));
    sg_env *e = new (std::nothrow) sg_env();
    if (!e) return fail(nullptr, SG_ERR_INVALID, "sg_create: out of host memory");
    std::memset(&e->buf, 0, sizeof(e->buf));
    e->d_actions = e->d_obs = e->d_reward = e->d_tobs = e->d_ship = e->d_planets = e->d_goal = nullptr;
    e->d_done = e->d_trunc = nullptr; e->d_elapsed = nullptr; e->stream = nullptr;
    std::memcpy(e->id, id, sizeof(id));
    d.num_envs = (int32_t)cfg->num_envs;
    d.env_index_base = cfg->env_index_base;
    d.seed_lo = (uint32_t)cfg->seed; d.seed_hi = (uint32_t)(cfg->seed >> 32);
    if (cfg->max_episode_steps > 0) d.max_episode_steps = cfg->max_episode_steps;
    d.auto_reset = cfg->auto_reset ? 1 : 0;
    e->dev = d; e->device = device; e->fresh = 1; e->profiling = 0;
    const size_t B = (size_t)d.num_envs, D = (size_t)obs_dim(d);
    const size_t NP = d.family == SG_FAMILY_GOAL ? (size_t)d.n_planets : 0;
    hipError_t st = hipSuccess;
    auto ok = [&](hipError_t r) { if (st == hipSuccess) st = r; };
    ok(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    ok(dmalloc(&e->buf.q0, B)); ok(dmalloc(&e->buf.q1, B));
    ok(dmalloc(&e->buf.elapsed, B)); ok(dmalloc(&e->buf.aux, B));
    if (d.family == SG_FAMILY_GOAL) {
        ok(dmalloc(&e->buf.pl[0], B));
        if (d.n_planets > 2) ok(dmalloc(&e->buf.pl[1], B));
        ok(dmalloc(&e->buf.freec, B)); ok(dmalloc(&e->buf.cshift, B));
    } else if (d.randomize_orbit) {
        ok(dmalloc(&e->buf.orbd, B));
    }
    ok(dmalloc(&e->d_actions, 2 * B)); ok(dmalloc(&e->d_obs, D * B)); ok(dmalloc(&e->d_tobs, D * B));
    ok(dmalloc(&e->d_reward, B)); ok(dmalloc(&e->d_done, B)); ok(dmalloc(&e->d_trunc, B));
    ok(dmalloc(&e->d_ship, 6 * B)); ok(dmalloc(&e->d_goal, 2 * B)); ok(dmalloc(&e->d_elapsed, B));
    if (NP) ok(dmalloc(&e->d_planets, 2 * NP * B));
    if (st == hipSuccess) {  // defined contents before the first reset
        ok(hipMemsetAsync(e->buf.q0, 0, B * sizeof(float4), e->stream));
        ok(hipMemsetAsync(e->buf.q1, 0, B * sizeof(float4), e->stream));
        ok(hipMemsetAsync(e->buf.elapsed, 0, B * sizeof(uint32_t), e->stream));
        ok(hipMemsetAsync(e->buf.aux, 0, B * sizeof(uint4), e->stream));
        ok(hipStreamSynchronize(e->stream));
    }
    if (st != hipSuccess) {
        int rc = fail(nullptr, SG_ERR_HIP, "sg_create: %s", hipGetErrorString(st));
        sg_destroy(e);
        return rc;
    }
    *out = e;
    return SG_OK;
}

extern "C" int sg_destroy(sg_env *e) {
    if (!e) return SG_OK;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (auto &p : e->prof_events) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    void *ptrs[] = {e->buf.q0, e->buf.q1, e->buf.pl[0], e->buf.pl[1], e->buf.elapsed, e->buf.aux, e->buf.freec,
                    e->buf.cshift, e->buf.orbd, e->d_actions, e->d_obs, e->d_reward, e->d_tobs, e->d_ship,
                    e->d_planets, e->d_goal, e->d_done, e->d_trunc, e->d_elapsed};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return SG_OK;
}

extern "C" int64_t sg_num_envs(const sg_env *e) { return e ? e->dev.num_envs : 0; }
extern "C" int32_t sg_obs_dim(const sg_env *e) { return e ? obs_dim(e->dev) : 0; }
extern "C" int32_t sg_num_planets(const sg_env *e) { return e && e->dev.family == SG_FAMILY_GOAL ? e->dev.n_planets : 0; }
extern "C" void *sg_stream(const sg_env *e) { return e ? (void *)e->stream : nullptr; }

extern "C" int sg_seed(sg_env *e, uint64_t seed) {
    if (!e) return SG_ERR_INVALID;
    e->dev.seed_lo = (uint32_t)seed; e->dev.seed_hi = (uint32_t)(seed >> 32);
    e->fresh = 1;
    return SG_OK;
}

extern "C" int sg_set_auto_reset(sg_env *e, int32_t on) {
    if (!e) return SG_ERR_INVALID;
    e->dev.auto_reset = on ? 1 : 0;
    return SG_OK;
}

extern "C" int sg_reset_device(sg_env *e, float *obs_dev, void *hip_stream) {
    if (!e) return SG_ERR_INVALID;
    hipStream_t s = (hipStream_t)hip_stream;
    const int grid = grid_for(e);
    if (e->dev.family == SG_FAMILY_GOAL) {
        switch (e->dev.n_planets) {
            case 2: goal_reset_kernel<2><<<grid, kBlock, 0, s>>>(e->dev, e->buf, e->fresh, obs_dev); break;
            case 3: goal_reset_kernel<3><<<grid, kBlock, 0, s>>>(e->dev, e->buf, e->fresh, obs_dev); break;
            default: goal_reset_kernel<4><<<grid, kBlock, 0, s>>>(e->dev, e->buf, e->fresh, obs_dev); break;
        }
    } else {
        kepler_reset_kernel<<<grid, kBlock, 0, s>>>(e->dev, e->buf, e->fresh, obs_dev);
    }
    e->fresh = 0;
    SG_HIP(e, hipGetLastError());
    return SG_OK;
}

// With profiling on, the step kernel is dispatched through hipExtLaunchKernelGGL, whose start/stop events take the
// begin/end timestamps of that dispatch itself (what rocprofv3 --kernel-trace reports), not of the stream around it.
template <typename K, typename... Args>
static void launch_maybe_timed(sg_env *e, K kernel, int grid, hipStream_t s, Args... args) {
    if (!e->profiling) {
        kernel<<<grid, kBlock, 0, s>>>(args...);
        return;
    }
    hipEvent_t a = nullptr, b = nullptr;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        kernel<<<grid, kBlock, 0, s>>>(args...);
        return;
    }
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, s, a, b, 0, args...);
    e->prof_events.emplace_back(a, b);
}

static int launch_step(sg_env *e, const float *actions, float *obs, float *reward, uint8_t *done, uint8_t *trunc,
                       float *tobs, hipStream_t s) {
    const int grid = grid_for(e);
    const float2 *a2 = reinterpret_cast<const float2 *>(actions);
    if (e->dev.family == SG_FAMILY_GOAL) {
        switch (e->dev.n_planets) {
            case 2: launch_maybe_timed(e, goal_step_kernel<2>, grid, s, e->dev, e->buf, a2, obs, reward, done, trunc, tobs); break;
            case 3: launch_maybe_timed(e, goal_step_kernel<3>, grid, s, e->dev, e->buf, a2, obs, reward, done, trunc, tobs); break;
            default: launch_maybe_timed(e, goal_step_kernel<4>, grid, s, e->dev, e->buf, a2, obs, reward, done, trunc, tobs); break;
        }
    } else {
        launch_maybe_timed(e, kepler_step_kernel, grid, s, e->dev, e->buf, a2, obs, reward, done, trunc, tobs);
    }
    return SG_OK;
}

extern "C" int sg_set_profiling(sg_env *e, int32_t on) {
    if (!e) return SG_ERR_INVALID;
    e->profiling = on ? 1 : 0;
    return SG_OK;
}

// Durations (ms) of the step-kernel launches recorded since profiling was switched on or last read.  The caller must
// have synchronised the stream(s) those launches went to.  Any output pointer may be NULL.
extern "C" int sg_get_profile(sg_env *e, int64_t *launches, double *total_ms, double *min_ms, double *max_ms) {
    if (!e) return SG_ERR_INVALID;
    double tot = 0, mn = 1e300, mx = 0;
    int64_t n = 0;
    for (auto &p : e->prof_events) {
        float ms = 0.f;
        if (hipEventSynchronize(p.second) == hipSuccess && hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
            tot += ms; n++;
            if (ms < mn) mn = ms;
            if (ms > mx) mx = ms;
        }
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    e->prof_events.clear();
    if (launches) *launches = n;
    if (total_ms) *total_ms = tot;
    if (min_ms) *min_ms = n ? mn : 0;
    if (max_ms) *max_ms = mx;
    return SG_OK;
}

extern "C" int sg_step_device(sg_env *e, const float *actions, float *obs, float *reward, uint8_t *done, uint8_t *trunc,
                              float *tobs, void *hip_stream) {
    if (!e || !actions || !obs || !reward || !done || !trunc) return e ? fail(e, SG_ERR_INVALID, "sg_step_device: null buffer") : SG_ERR_INVALID;
    launch_step(e, actions, obs, reward, done, trunc, tobs, (hipStream_t)hip_stream);
    SG_HIP(e, hipGetLastError());
    return SG_OK;
}

extern "C" int sg_rollout_device(sg_env *e, int32_t n_steps, const float *actions, float *obs, float *reward,
                                 uint8_t *done, uint8_t *trunc, void *hip_stream) {
    if (!e || n_steps < 0 || !actions || !obs || !reward || !done || !trunc) return e ? fail(e, SG_ERR_INVALID, "sg_rollout_device: bad argument") : SG_ERR_INVALID;
    const size_t B = (size_t)e->dev.num_envs, D = (size_t)obs_dim(e->dev);
    for (int32_t t = 0; t < n_steps; t++)
        launch_step(e, actions + t * 2 * B, obs + t * D * B, reward + t * B, done + t * B, trunc + t * B, nullptr,
                    (hipStream_t)hip_stream);
    SG_HIP(e, hipGetLastError());
    return SG_OK;
}

extern "C" int sg_reset(sg_env *e, float *obs_host) {
    if (!e) return SG_ERR_INVALID;
    SG_HIP(e, hipSetDevice(e->device));
    int rc = sg_reset_device(e, obs_host ? e->d_obs : nullptr, e->stream);
    if (rc) return rc;
    if (obs_host)
        SG_HIP(e, hipMemcpyAsync(obs_host, e->d_obs, sizeof(float) * obs_dim(e->dev) * (size_t)e->dev.num_envs, hipMemcpyDeviceToHost, e->stream));
    SG_HIP(e, hipStreamSynchronize(e->stream));
    return SG_OK;
}

extern "C" int sg_step(sg_env *e, const float *actions, float *obs, float *reward, uint8_t *done, uint8_t *trunc, float *tobs) {
    if (!e || !actions || !obs || !reward || !done || !trunc) return e ? fail(e, SG_ERR_INVALID, "sg_step: null buffer") : SG_ERR_INVALID;
    SG_HIP(e, hipSetDevice(e->device));
    const size_t B = (size_t)e->dev.num_envs, D = (size_t)obs_dim(e->dev);
    SG_HIP(e, hipMemcpyAsync(e->d_actions, actions, sizeof(float) * 2 * B, hipMemcpyHostToDevice, e->stream));
    if (tobs) SG_HIP(e, hipMemcpyAsync(e->d_tobs, tobs, sizeof(float) * D * B, hipMemcpyHostToDevice, e->stream));
    launch_step(e, e->d_actions, e->d_obs, e->d_reward, e->d_done, e->d_trunc, tobs ? e->d_tobs : nullptr, e->stream);
    SG_HIP(e, hipGetLastError());
    SG_HIP(e, hipMemcpyAsync(obs, e->d_obs, sizeof(float) * D * B, hipMemcpyDeviceToHost, e->stream));
    SG_HIP(e, hipMemcpyAsync(reward, e->d_reward, sizeof(float) * B, hipMemcpyDeviceToHost, e->stream));
    SG_HIP(e, hipMemcpyAsync(done, e->d_done, B, hipMemcpyDeviceToHost, e->stream));
    SG_HIP(e, hipMemcpyAsync(trunc, e->d_trunc, B, hipMemcpyDeviceToHost, e->stream));
    if (tobs) SG_HIP(e, hipMemcpyAsync(tobs, e->d_tobs, sizeof(float) * D * B, hipMemcpyDeviceToHost, e->stream));
    SG_HIP(e, hipStreamSynchronize(e->stream));
    return SG_OK;
}

extern "C" int sg_get_state(sg_env *e, float *ship, float *planets, float *goal, int32_t *elapsed) {
    if (!e) return SG_ERR_INVALID;
    SG_HIP(e, hipSetDevice(e->device));
    const size_t B = (size_t)e->dev.num_envs, NP = e->dev.family == SG_FAMILY_GOAL ? (size_t)e->dev.n_planets : 0;
    if (planets && !NP) return fail(e, SG_ERR_INVALID, "sg_get_state: this env id has no per-env planets");
    state_io_kernel<<<grid_for(e), kBlock, 0, e->stream>>>(e->dev, e->buf, 0, ship ? e->d_ship : nullptr,
                                                          planets ? e->d_planets : nullptr, goal ? e->d_goal : nullptr,
                                                          elapsed ? e->d_elapsed : nullptr);
    SG_HIP(e, hipGetLastError());
    if (ship) SG_HIP(e, hipMemcpyAsync(ship, e->d_ship, sizeof(float) * 6 * B, hipMemcpyDeviceToHost, e->stream));
    if (planets) SG_HIP(e, hipMemcpyAsync(planets, e->d_planets, sizeof(float) * 2 * NP * B, hipMemcpyDeviceToHost, e->stream));
    if (goal) SG_HIP(e, hipMemcpyAsync(goal, e->d_goal, sizeof(float) * 2 * B, hipMemcpyDeviceToHost, e->stream));
    if (elapsed) SG_HIP(e, hipMemcpyAsync(elapsed, e->d_elapsed, sizeof(int32_t) * B, hipMemcpyDeviceToHost, e->stream));
    SG_HIP(e, hipStreamSynchronize(e->stream));
    return SG_OK;
}

extern "C" int sg_set_state(sg_env *e, const float *ship, const float *planets, const float *goal, const int32_t *elapsed) {
    if (!e) return SG_ERR_INVALID;
    SG_HIP(e, hipSetDevice(e->device));
    const size_t B = (size_t)e->dev.num_envs, NP = e->dev.family == SG_FAMILY_GOAL ? (size_t)e->dev.n_planets : 0;
    if (planets && !NP) return fail(e, SG_ERR_INVALID, "sg_set_state: this env id has no per-env planets");
    if (ship) SG_HIP(e, hipMemcpyAsync(e->d_ship, ship, sizeof(float) * 6 * B, hipMemcpyHostToDevice, e->stream));
    if (planets) SG_HIP(e, hipMemcpyAsync(e->d_planets, planets, sizeof(float) * 2 * NP * B, hipMemcpyHostToDevice, e->stream));
    if (goal) SG_HIP(e, hipMemcpyAsync(e->d_goal, goal, sizeof(float) * 2 * B, hipMemcpyHostToDevice, e->stream));
    if (elapsed) SG_HIP(e, hipMemcpyAsync(e->d_elapsed, elapsed, sizeof(int32_t) * B, hipMemcpyHostToDevice, e->stream));
    state_io_kernel<<<grid_for(e), kBlock, 0, e->stream>>>(e->dev, e->buf, 1, ship ? e->d_ship : nullptr,
                                                          planets ? e->d_planets : nullptr, goal ? e->d_goal : nullptr,
                                                          const_cast<int32_t *>(elapsed ? e->d_elapsed : nullptr));
    SG_HIP(e, hipGetLastError());
    SG_HIP(e, hipStreamSynchronize(e->stream));
    return SG_OK;
}
