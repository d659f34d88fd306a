"""CPU-only tests: the C ABI loads and exports every declared symbol, the host surface mirrors the reference's,
and the product fails loudly without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, has_gpu


def test_library_exports_every_declared_symbol():
    from space_gym_amd import _native
    lib = _native.load()
    header = open(os.path.join(ROOT, "include", "spacegym.h")).read()
    declared = set(re.findall(r"\b(sg_[a-z_]+)\s*\(", header))
    assert declared == set(_native.SYMBOLS), declared ^ set(_native.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)
    assert b"gfx950" in lib.sg_version()


def test_fails_loudly_without_gpu():
    if has_gpu():
        pytest.skip("a GPU is visible")
    import space_gym_amd as sg
    from space_gym_amd._native import NativeError
    with pytest.raises(NativeError, match="no CPU path"):
        sg.make_vec("GoalContinuous3P-v0", 16)


def test_unknown_id_is_rejected():
    import space_gym_amd as sg
    with pytest.raises(ValueError):
        sg.make_vec("DoNotCrashContinuous-v0", 4)  # broken in the reference at this revision, not served


def test_spaces_mirror_the_reference():
    """spaceship_env.py:102-111,206-208 and kepler.py:158-170."""
    from space_gym_amd.registration import ENV_SPECS, obs_dim, single_action_space, single_observation_space
    assert {k: obs_dim(k) for k in ENV_SPECS} == {
        "GoalContinuous2P-v0": 13, "GoalContinuous3P-v0": 15, "GoalContinuous4P-v0": 17, "KeplerCircleOrbit-v0": 10,
        "KeplerEllipseEasy-v0": 10, "KeplerEllipseHard-v0": 10, "KeplerRandomOrbits-v0": 10,
        "GoalDiscrete2-v0": 13, "GoalDiscrete3-v0": 15, "GoalDiscrete4-v0": 17, "KeplerDiscrete-v0": 10}
    assert single_action_space("GoalDiscrete3-v0").n == 6 and single_action_space("GoalDiscrete3-v0").contains(5)
    sp = single_observation_space("GoalContinuous3P-v0")
    assert sp.shape == (15,) and sp.dtype == np.float32
    assert np.isinf(sp.high[4:6]).all() and np.allclose(sp.high[7:], 2 * np.sqrt(2)) and np.array_equal(sp.low, -sp.high)
    k = single_observation_space("KeplerCircleOrbit-v0")
    assert np.allclose(k.high[7:], [2 * np.pi, 0.7, 2]) and np.array_equal(k.low, -k.high)
    a = single_action_space("GoalContinuous3P-v0")
    assert a.contains(np.array([1.0, -1.0], np.float32)) and not a.contains(np.array([1.5, 0.0], np.float32))


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under space_gym_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "space_gym_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "spacegym_oracle" not in src, f


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` as a plain command (no torchrun around it): the parent starts two fresh ranks through
    torch.distributed.run before touching any GPU, relays their output and leaves with their exit code.  Without a GPU
    each rank stops at the engine's "no CPU path" check -- which is what this CPU test sees, twice."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["SG_BENCH_REHEARSE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                          "--batch", "512", "--preroll", "4", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300,
                         env=env, cwd=ROOT)
    if has_gpu():
        assert out.returncode == 0, out.stderr[-2000:]
        import json
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(line) == 1 and json.loads(line[0])["n_gpus"] == 2
    else:
        assert out.returncode != 0
        assert (out.stdout + out.stderr).count("bench.py needs an MI355X") == 2, (out.stdout + out.stderr)[-2000:]


def test_no_vector_register_spills_in_the_default_kernels():
    """The wave-pair kernels are compiled for two waves per SIMD (256 vector registers) and sit close to that limit: a few
    registers more and the compiler spills to scratch inside the step loops (measured: +40 % per step).  The compiler's
    own resource report (hipcc -Rpass-analysis=kernel-resource-usage, no GPU needed) must show no vector spill for any
    kernel a default plan launches (the three-waves-per-SIMD variant of the one-wave Goal rollout kernel, which only
    SPACEGYM_SPARE_DEPTH=2 selects, is exempt)."""
    import re, shutil, subprocess
    from space_gym_amd import build
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    cmd = [build.hipcc(), *build.flags(("-Rpass-analysis=kernel-resource-usage",)), "-o", os.devnull, os.path.join(build.CSRC, "sg_engine.hip")]
    err = subprocess.run(cmd, capture_output=True, text=True, timeout=900).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: \s*(.+?)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        k, _, v = m.group(1).partition(":")
        if k.strip() == "Function Name":
            cur = {"name": v.strip()}
            rows.append(cur)
        elif cur is not None:
            cur[k.strip()] = v.strip()
    assert len(rows) >= 40  # every kernel of the library reported
    spilled = {r["name"]: int(r["VGPRs Spill"]) for r in rows if int(r.get("VGPRs Spill", "0")) > 0}
    exempt = {n for n in spilled if re.search(r"goal_rollout_kernelILi\dELb[01]ELi2E", n)}
    assert not (set(spilled) - exempt), {n: spilled[n] for n in set(spilled) - exempt}
    pair = [r for r in rows if "goal_pair_rollout_kernel" in r["name"]]
    assert len(pair) == 12 and all(int(r["VGPRs"]) <= 256 and int(r["ScratchSize [bytes/lane]"]) == 0 for r in pair)


def test_step_kernels_wait_for_memory_only_at_the_top_of_a_pass():
    """goal_step_kernel / kepler_step_kernel load the next subtile's inputs a whole pass ahead, straight into LDS
    (global_load_lds), and store the previous subtile's outputs in front of those loads: the pass that follows must not wait for
    memory (with the inputs loaded into registers the compiler had put a wait for everything behind the stores and waits and
    copies right behind the loads: 59.9 instead of 53.8 us per launch at 1 048 576 envs).  Checked on the device assembly (no GPU
    needed): behind the loop's group of LDS-DMA loads at least 300 instructions follow without an `s_waitcnt vmcnt`."""
    import re, shutil, subprocess, tempfile
    from space_gym_amd import build
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "sg.s")
        flags = [f for f in build.flags() if f not in ("-shared", "-fPIC")]
        subprocess.run([build.hipcc(), *flags, "-S", "--cuda-device-only", "-o", asm, os.path.join(build.CSRC, "sg_engine.hip")],
                       check=True, capture_output=True, timeout=900)
        lines = open(asm).read().splitlines()
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\d+(goal|kepler)_step_kernelI\w*:", l)]
    assert len(starts) == 8  # 2P / 3P / 4P x two steerings, Kepler x two steerings
    for st in starts:
        end = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))
        ins = [l.split(";")[0].strip() for l in lines[st + 1:end]]
        ins = [l for l in ins if l and not l.startswith(".") and not l.endswith(":")]
        dma = [k for k, l in enumerate(ins) if l.startswith("global_load_lds")]
        assert len(dma) >= 14, lines[st]  # the prologue's group and the loop's
        last = dma[-1]
        nxt = next((k for k in range(last + 1, len(ins)) if ins[k].startswith("s_waitcnt") and "vmcnt" in ins[k]), len(ins))
        assert nxt - last >= 300, (lines[st], nxt - last)
        # and nothing but address arithmetic between the loads of the group
        first = next(k for k in dma if last - k < 120)
        assert not any(l.startswith("s_waitcnt") and "vmcnt" in l for l in ins[first:last]), lines[st]


def test_fast_step_coefficients_satisfy_the_order_conditions():
    """The constants of Integrator::fast_step (sg_device.hpp, N5_*) are Nystrom's fifth-order method for x'' = f(t, x)
    (Hairer, Norsett, Wanner I, II.14): read out of the header, they satisfy the order conditions up to order 5 -- the
    quadrature conditions b.c^k = 1/(k+1), the row sums sum_j abar_ij = c_i^2/2, bbar_i = b_i (1 - c_i), and the three
    conditions that involve the stage matrix (b.Abar c = 1/24, b.(c Abar c) = 1/30, b.Abar c^2 = 1/60) -- the indicator weights
    D_j = abar_4j - bbar_j sum to zero (a constant force has no error), and a step of the method converges with order 5
    (local error ratio 64 per halving) on a forced two-body problem."""
    from fractions import Fraction as Fr
    from conftest import ROOT
    src = open(os.path.join(ROOT, "space_gym_amd", "csrc", "sg_device.hpp")).read()
    vals = {}
    for name, expr in re.findall(r"\b(N5_[A-Z0-9]+) = ([^,;]+)", src):
        expr = expr.replace("(float)", "").replace("f", "").strip()
        vals[name] = float(eval(re.sub(r"(\d+\.?\d*)", r"Fr('\1')", expr), {"Fr": Fr}))
    c = np.array([0.0, vals["N5_C2"], vals["N5_C3"], 1.0])
    A = np.zeros((4, 4))
    A[1, 0] = vals["N5_A21"]; A[2, :2] = [vals["N5_A31"], vals["N5_A32"]]; A[3, :3] = [3 / 10, -2 / 35, 9 / 35]  # (row 4 enters as D)
    b = np.array([vals["N5_B1"], vals["N5_B2"], vals["N5_B3"], vals["N5_B4"]])
    bb = np.array([vals["N5_BB1"], vals["N5_BB2"], vals["N5_BB3"], 0.0])
    D = np.array([vals["N5_D1"], vals["N5_D2"], vals["N5_D3"]])
    tol = 1e-15
    for k in range(5):
        assert abs(b @ c ** k - 1 / (k + 1)) < tol
    assert np.abs(bb - b * (1 - c)).max() < tol and np.abs(A.sum(1) - c ** 2 / 2).max() < tol
    assert abs(b @ (A @ c) - 1 / 24) < tol and abs(b @ (c * (A @ c)) - 1 / 30) < tol and abs(b @ (A @ c ** 2) - 1 / 60) < tol
    assert np.abs(D - (A[3, :3] - bb[:3])).max() < 1e-8 and abs(D.sum()) < 1e-7  # (D is rounded to float in the header)

    def f(t, x):
        r2 = x @ x
        return -x / (r2 * np.sqrt(r2)) + 0.3 * np.array([np.cos(1 + 3 * t), np.sin(1 + 3 * t)])

    def step(x, v, h):
        k = []
        for i in range(4):
            k.append(f(c[i] * h, x + c[i] * h * v + h * h * sum(A[i, j] * k[j] for j in range(i))))
        return x + h * v + h * h * sum(bb[j] * k[j] for j in range(4)), v + h * sum(b[j] * k[j] for j in range(4))

    from scipy.integrate import solve_ivp
    x0, v0 = np.array([1.0, 0.2]), np.array([0.1, 0.9])
    err = []
    for h in (0.2, 0.1, 0.05):
        ref = solve_ivp(lambda t, y: np.r_[y[2:], f(t, y[:2])], (0, h), np.r_[x0, v0], method="DOP853", rtol=3e-14, atol=1e-16).y[:, -1]
        x1, v1 = step(x0, v0, h)
        err.append(max(np.abs(x1 - ref[:2]).max(), np.abs(v1 - ref[2:]).max()))
    assert 50 < err[0] / err[1] < 80 and 50 < err[1] / err[2] < 80


def test_constructor_kwargs_merge_like_gym_make():
    """make_vec(env_id, **kwargs) merges the reference's constructor kwargs like gym.make(id, **kwargs): class defaults
    (goal.py:18-31, kepler.py:189-203) < registered kwargs (gym_space/__init__.py:26-146) < the caller's; an unknown keyword or a
    missing required one is a TypeError like the reference constructor's."""
    import pytest
    from space_gym_amd.registration import constructor_kwargs
    kw = constructor_kwargs("GoalContinuous3P-v0")
    assert kw["n_planets"] == 3 and kw["ship_steering"] == 1 and kw["danger_zone"] == 0.25 and kw["survival_reward_scale"] == 0.2
    kw = constructor_kwargs("GoalContinuous3P-v0", dict(max_engine_force=0.7, danger_zone=0.4))
    assert kw["max_engine_force"] == 0.7 and kw["danger_zone"] == 0.4 and kw["goal_vel_reward_scale"] == 5.0
    kw = constructor_kwargs("KeplerCircleOrbit-v0")
    assert kw["step_size"] == 0.07 and kw["ref_orbit_eccentricity"] == 0 and kw["ship_steering"] == 1
    kw = constructor_kwargs("KeplerCircleOrbit-v0", from_class=True)  # KeplerContinuousEnv(): the class's own defaults
    assert kw["step_size"] == 0.1 and kw["ref_orbit_eccentricity"] == 0.5 and kw["ship_steering"] == 0 and kw["ref_orbit_angle"] == 3.75
    with pytest.raises(TypeError):
        constructor_kwargs("GoalContinuous3P-v0", dict(step_size=0.1))  # GoalEnv.__init__ has no step_size (fixed: goal.py:66)
    with pytest.raises(TypeError):
        constructor_kwargs("KeplerCircleOrbit-v0", dict(danger_zone=0.1))
    with pytest.raises(TypeError):
        constructor_kwargs("GoalContinuous2P-v0", from_class=True)  # the three reward scales have no default


def test_sg_params_layout_matches_the_header():
    """the ctypes mirror of sg_params has the fields of include/spacegym.h in the same order"""
    import re
    from space_gym_amd import _native
    header = open(os.path.join(ROOT, "include", "spacegym.h")).read()
    body = header[header.index("typedef struct sg_params {"):header.index("} sg_params;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.replace("typedef struct sg_params {", "").strip()
        if not decl:
            continue
        ctype, rest = decl.split(None, 1)
        names += [n.strip() for n in rest.split(",")]
    assert names == [f for f, _ in _native.SgParams._fields_], names


def test_register_with_gym_against_a_stub_module(monkeypatch):
    """register_with_gym() against a stand-in `gym` module (no gym / gymnasium in the image): every served id is registered once
    under <name>Vec-v0 with the vector entry point and the id as kwarg, without a TimeLimit of gym's own (the engine counts
    the steps itself); a failing register() call does not stop the others; without either module nothing happens."""
    import sys
    import types
    from space_gym_amd import registration
    calls = []

    def register(id, entry_point, kwargs=None, max_episode_steps=None, **rest):
        if id == "KeplerDiscreteVec-v0":
            raise RuntimeError("already registered")  # e.g. a second call
        calls.append((id, entry_point, kwargs, max_episode_steps))
    monkeypatch.setitem(sys.modules, "gym", types.SimpleNamespace(register=register))
    monkeypatch.setitem(sys.modules, "gymnasium", None)  # import gymnasium -> ImportError
    done = registration.register_with_gym()
    ids = [c[0] for c in calls]
    assert sorted(ids) == sorted(k.replace("-v0", "Vec-v0") for k in registration.ENV_SPECS if k != "KeplerDiscrete-v0")
    assert all(c[1] == "space_gym_amd.vector_env:make_vec" and c[3] is None for c in calls)
    assert all(c[2] == dict(env_id=c[0].replace("Vec-v0", "-v0")) for c in calls)
    assert done == [("gym", i) for i in ids]
    # the entry point is what gym.make would call: make_vec(env_id=..., **user kwargs)
    import importlib
    mod, fn = calls[0][1].split(":")
    assert callable(getattr(importlib.import_module(mod), fn))
    monkeypatch.setitem(sys.modules, "gym", None)
    assert registration.register_with_gym() == []
