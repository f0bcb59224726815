import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (the product path has no CPU fallback)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def ticks():
    return np.load(os.path.join(GOLDEN, "ticks.npz"))


@pytest.fixture(scope="session")
def gjk_golden():
    return np.load(os.path.join(GOLDEN, "gjk_vectors.npz"))


def golden_mission(ticks, name):
    import lsc_planner_amd as L
    g = lambda k: ticks[f"{name}/{k}"]
    return L.Mission(g("start"), g("goal"), g("world_min"), g("world_max"), g("radius"), g("downwash"), g("max_vel"),
                     g("max_acc"), g("nominal_velocity"), name=name)


def oracle_swarm(O, ms):
    prm = O.make_params(world_min=ms.world_min, world_max=ms.world_max, obs_f32=True)
    return O.Swarm(prm, ms.radius, ms.downwash, ms.max_vel, ms.max_acc, ms.nominal_velocity)
