"""Environment ids served by the engine, with the kwargs the reference registers them with
(gym_space/__init__.py:26-146).  The native library holds the same constants (csrc/sg_host_config.hpp);
this table is what the Python surface exposes (spec lookup, spaces, optional gym registration)."""
import numpy as np

from .spaces import Box, Discrete

_GOAL_KW = dict(ship_steering=1, ship_moi=0.01, survival_reward_scale=0.2, goal_vel_reward_scale=5.0,
                safety_reward_scale=10.0, goal_sparse_reward=5.0, max_engine_force=0.4)
_KEPLER_KW = dict(ship_steering=1, ship_moi=0.01, rad_penalty_C=2, numerator_C=0.01, act_penalty_C=0.5, step_size=0.07)

ENV_SPECS = {
    "GoalContinuous2P-v0": dict(family="goal", n_planets=2, max_episode_steps=500, kwargs=dict(n_planets=2, **_GOAL_KW)),
    "GoalContinuous3P-v0": dict(family="goal", n_planets=3, max_episode_steps=500, kwargs=dict(n_planets=3, **_GOAL_KW)),
    "GoalContinuous4P-v0": dict(family="goal", n_planets=4, max_episode_steps=500, kwargs=dict(n_planets=4, **_GOAL_KW)),
    "KeplerCircleOrbit-v0": dict(family="kepler", n_planets=0, max_episode_steps=500,
                                 kwargs=dict(randomize=False, ref_orbit_a=1.2, ref_orbit_eccentricity=0, ref_orbit_angle=0, **_KEPLER_KW)),
    "KeplerEllipseEasy-v0": dict(family="kepler", n_planets=0, max_episode_steps=500,
                                 kwargs=dict(randomize=False, ref_orbit_a=1.2, ref_orbit_eccentricity=0.5, ref_orbit_angle=0.8, **_KEPLER_KW)),
    "KeplerEllipseHard-v0": dict(family="kepler", n_planets=0, max_episode_steps=500,
                                 kwargs=dict(randomize=False, ref_orbit_a=1.2, ref_orbit_eccentricity=0.725, ref_orbit_angle=3.925, **_KEPLER_KW)),
    "KeplerRandomOrbits-v0": dict(family="kepler", n_planets=0, max_episode_steps=500, kwargs=dict(randomize=True, **_KEPLER_KW)),
    # discrete-action ids (DiscreteSpaceshipEnv, spaceship_env.py:183-202), registered only by keyboard_agent.py:10-74
    "GoalDiscrete2-v0": dict(family="goal", n_planets=2, max_episode_steps=500, discrete=True,
                             kwargs=dict(n_planets=2, **{**_GOAL_KW, "max_engine_force": 1})),
    "GoalDiscrete3-v0": dict(family="goal", n_planets=3, max_episode_steps=500, discrete=True,
                             kwargs=dict(n_planets=3, **{**_GOAL_KW, "max_engine_force": 1})),
    "GoalDiscrete4-v0": dict(family="goal", n_planets=4, max_episode_steps=500, discrete=True,
                             kwargs=dict(n_planets=4, **{**_GOAL_KW, "max_engine_force": 1})),
    "KeplerDiscrete-v0": dict(family="kepler", n_planets=0, max_episode_steps=None, discrete=True,  # no TimeLimit registered
                              kwargs=dict(randomize=False, ref_orbit_a=1.2, ref_orbit_eccentricity=0, ref_orbit_angle=0,
                                          max_engine_force=0.4, reward_value=0, **_KEPLER_KW)),
}


# Constructor signatures of the reference classes: keyword -> default (REQUIRED: no default).  GoalEnv.__init__ goal.py:18-31,
# KeplerEnv.__init__ kepler.py:189-203.  `fixed_position`, `reward_value` and `renderer_kwargs` are accepted and ignored: nothing
# on the reference's step path reads them.
REQUIRED = object()
CLASS_KWARGS = {
    "goal": dict(goal_vel_reward_scale=REQUIRED, safety_reward_scale=REQUIRED, goal_sparse_reward=REQUIRED, fixed_position=False,
                 danger_zone=0.25, survival_reward_scale=0.0, n_planets=2, ship_steering=0, ship_moi=0.01, max_engine_force=0.4,
                 renderer_kwargs=None),
    "kepler": dict(randomize=False, ref_orbit_a=1.2, ref_orbit_eccentricity=0.5, ref_orbit_angle=3.75, reward_value=0,
                   numerator_C=0.01, rad_penalty_C=2.0, act_penalty_C=0.5, step_size=0.1, ship_steering=0, ship_moi=0.01,
                   max_engine_force=0.4),
}
# The reference's classes (gym_space/envs/goal.py:286-291, kepler.py:270-275), for make_vec_from_class: the id whose family and
# action space they share; every keyword then comes from the class defaults above and the caller.
ENV_CLASSES = {"GoalContinuousEnv": "GoalContinuous2P-v0", "GoalDiscreteEnv": "GoalDiscrete2-v0",
               "KeplerContinuousEnv": "KeplerCircleOrbit-v0", "KeplerDiscreteEnv": "KeplerDiscrete-v0"}
_IGNORED = ("fixed_position", "reward_value", "renderer_kwargs")


def constructor_kwargs(env_id, overrides=None, from_class=False):
    """The keyword arguments the reference would construct the env with: the class defaults, the kwargs the id was registered
    with (gym_space/__init__.py:26-146; not with from_class) and the caller's overrides, as gym.make(id, **overrides) merges
    them.  An unknown keyword or a missing required one raises TypeError like the reference's constructor."""
    fam = ENV_SPECS[env_id]["family"]
    kw = dict(CLASS_KWARGS[fam])
    if not from_class:
        kw.update(ENV_SPECS[env_id]["kwargs"])
    for k, v in (overrides or {}).items():
        if k not in CLASS_KWARGS[fam]:
            raise TypeError(f"__init__() got an unexpected keyword argument {k!r} ({'GoalEnv' if fam == 'goal' else 'KeplerEnv'})")
        kw[k] = v
    missing = [k for k, v in kw.items() if v is REQUIRED]
    if missing:
        raise TypeError(f"__init__() missing required arguments: {missing}")
    if fam == "goal" and kw.get("fixed_position"):
        raise ValueError("fixed_position=True is not served (the reference never reads it either: goal.py:23)")
    return kw


def obs_dim(env_id, n_planets=None):
    s = ENV_SPECS[env_id]
    return 7 + 2 * (n_planets or s["n_planets"]) + 2 if s["family"] == "goal" else 10


def single_observation_space(env_id, n_planets=None):
    s = ENV_SPECS[env_id]
    if s["family"] == "goal":  # spaceship_env.py:102-111
        high = [1.0, 1.0, 1.0, 1.0, np.inf, np.inf, 1.0] + (2 * (n_planets or s["n_planets"]) + 2) * [2 * np.sqrt(2)]
    else:  # kepler.py:158-170 (low = -high including the three orbit slots)
        high = [1.0, 1.0, 1.0, 1.0, np.inf, np.inf, 1.0, 2 * np.pi, 0.7, 2]
    high = np.array(high, dtype=np.float32)
    return Box(-high, high)


def is_discrete(env_id):
    return bool(ENV_SPECS[env_id].get("discrete"))


def single_action_space(env_id):
    if is_discrete(env_id):
        return Discrete(2 * 3)  # engine on/off x thruster cw/none/ccw, spaceship_env.py:184-187
    ones = np.ones(2, dtype=np.float32)  # spaceship_env.py:206-208
    return Box(-ones, ones)


def register_with_gym():
    """Register vector entry points with gym / gymnasium when one of them is importable (neither is required)."""
    registered = []
    for modname in ("gymnasium", "gym"):
        try:
            mod = __import__(modname)
        except ImportError:
            continue
        for env_id, spec in ENV_SPECS.items():
            new_id = env_id.replace("-v0", "Vec-v0")
            try:
                mod.register(id=new_id, entry_point="space_gym_amd.vector_env:make_vec", kwargs=dict(env_id=env_id),
                             max_episode_steps=None)
                registered.append((modname, new_id))
            except Exception:
                pass
    return registered
