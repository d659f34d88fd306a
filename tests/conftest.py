import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

FAMILIES = {
    "goal2p": "GoalContinuous2P-v0",
    "goal3p": "GoalContinuous3P-v0",
    "goal4p": "GoalContinuous4P-v0",
    "kepler_circle": "KeplerCircleOrbit-v0",
    "kepler_easy": "KeplerEllipseEasy-v0",
    "kepler_hard": "KeplerEllipseHard-v0",
    # discrete-action ids of keyboard_agent.py:10-74
    "goal_discrete2": "GoalDiscrete2-v0",
    "goal_discrete3": "GoalDiscrete3-v0",
    "goal_discrete4": "GoalDiscrete4-v0",
    "kepler_discrete": "KeplerDiscrete-v0",
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def golden_steps():
    return {fam: load_golden("step_" + fam) for fam in FAMILIES}


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
