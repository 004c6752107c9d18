"""Shared test plumbing.

* registers the `gpu` marker (tests that need a real MI355X);
* builds and loads the CPU oracle (oracle/, test infrastructure only);
* loads the product package `deciphon-old_amd/` (hyphenated directory, so it is
  imported by path under the module name `deciphon_old_amd`).
"""
import ctypes as C
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_product():
    name = "deciphon_old_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "deciphon-old_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(
        name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_py import Oracle  # noqa: E402


@pytest.fixture(scope="session")
def oracle32():
    return Oracle(32)


@pytest.fixture(scope="session")
def oracle64():
    return Oracle(64)


@pytest.fixture(scope="session")
def dcp():
    return load_product()


@pytest.fixture(scope="session")
def bench_mod():
    """bench.py as a module: the workload generators (core sizes, queries) of BASELINE.json's configs."""
    import bench
    return bench


@pytest.fixture(scope="session")
def c3_profiles(dcp, bench_mod):
    """The C3 / C4 database: 20 000 sampled profiles (host objects, built once per session)."""
    from concurrent.futures import ThreadPoolExecutor

    sizes = bench_mod.core_sizes_for("c3", 20000)
    cfg = dcp.ProteinCfg(2, 0.01)  # ENTRY_DIST_OCCUPANCY
    with ThreadPoolExecutor(16) as ex:
        profiles = list(ex.map(lambda p: dcp.ProteinProfile.sample(0xDEC1F0 + p, int(sizes[p]), cfg, f"PF{p:05d}"),
                               range(len(sizes))))
    return sizes, profiles
