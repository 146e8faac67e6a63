import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def product_library():
    """The host helpers of the C ABI (Voronoi cells, neighbour rings, ...) live in libsurtr_hip.so: build it when a fresh
    checkout has none (hipcc cross-compiles gfx950 without a GPU).  Building is not a fallback: nothing is computed here."""
    from surtr_amd import engine
    if not os.path.exists(engine.lib_path()):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def emul_lib_path():
    """Single-lane CPU emulation of the kernels (tests/emul), built on demand."""
    d = os.path.join(ROOT, "tests", "emul")
    subprocess.check_call(["make", "-s", "-C", d, "all"])
    return os.path.join(d, "libsurtr_emul.so")


@pytest.fixture()
def emul_engine_small(emul_lib_path):
    """Emulation built with a tiny LDS topology so that solids overflow into the wide (global) variant."""
    from surtr_amd import engine
    engine._use_library_for_tests(os.path.join(os.path.dirname(emul_lib_path), "libsurtr_emul_small.so"))
    try:
        yield engine
    finally:
        engine._use_library_for_tests(None)


@pytest.fixture()
def emul_engine(emul_lib_path):
    """Engine bound to the CPU emulation of the kernels -- checks kernel logic without a GPU."""
    from surtr_amd import engine
    engine._use_library_for_tests(emul_lib_path)
    try:
        yield engine
    finally:
        engine._use_library_for_tests(None)


@pytest.fixture()
def gpu_engine():
    """Engine bound to libsurtr_hip.so; fails loudly if the HIP library or device is missing."""
    from surtr_amd import engine
    engine._use_library_for_tests(None)
    assert os.path.exists(engine.lib_path()), "libsurtr_hip.so missing: run __graft_entry__.build()"
    return engine
