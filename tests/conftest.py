import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def standalone_flat():
    from rustracer_amd import flatten, presets
    return flatten(presets.standalone_scene())


@pytest.fixture(scope="session")
def default_camera(oracle_mod):
    return oracle_mod.camera_lookat(-1.8, 1.0, 2.0)  # Raytracer::new, lib.rs:252
