import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """The package directory is named `pigeon.jl_amd` (not an importable identifier): load it under the alias pigeon_jl_amd."""
    if "pigeon_jl_amd" in sys.modules:
        return sys.modules["pigeon_jl_amd"]
    d = os.path.join(ROOT, "pigeon.jl_amd")
    spec = importlib.util.spec_from_file_location("pigeon_jl_amd", os.path.join(d, "__init__.py"), submodule_search_locations=[d])
    m = importlib.util.module_from_spec(spec)
    sys.modules["pigeon_jl_amd"] = m
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as o
    o.build()
    return o


def make_oracle(oracle_mod, traj, **kw):
    o = oracle_mod.Oracle(**kw)
    o.set_trajectory(traj.data)
    return o


@pytest.fixture(scope="session")
def skidpad(pkg):
    return pkg.load_path_fixture("skidpadoval")
