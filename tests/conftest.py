"""pytest configuration: registers the `gpu` marker, puts the repo root on sys.path, and gives the tests that need fault
injection a view of the TEST build of the library (tests/native/libvgen_hip_hooks.so — the product's sources compiled a
second time with -DVGEN_TEST_HOOKS; the shipped vgen_amd/libvgen_hip.so has no such entry points)."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HOOKS_SO = os.path.join(ROOT, "tests", "native", "libvgen_hip_hooks.so")
HOOKS_CLI = os.path.join(ROOT, "tests", "native", "vgen-hip-hooks")


def locked_make(*args):
    """`make <args>` under a lock file: test modules of different pytest-xdist workers run the same Makefiles (the product's, the
    oracle's, tests/native) when the tree is stale, and two makes writing the same objects side by side fail."""
    import fcntl
    import subprocess
    with open(os.path.join(ROOT, "tests", "native", ".make.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.check_call(["make", *args])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_hooks = None


def hooks_api():
    """A second instance of vgen_amd/api.py bound to the test build of the library: the same ctypes view, plus
    GpuRunner.fail_after (vgen_debug_fail_after) and the VGEN_DEBUG_GTAB_FAIL switch.  Contexts of the two instances are
    independent libraries in one process and must not be mixed in one call."""
    global _hooks
    if _hooks is None:
        assert os.path.exists(HOOKS_SO), f"{HOOKS_SO} not built (make -C tests/native, or __graft_entry__.build())"
        spec = importlib.util.spec_from_file_location("vgen_amd_hooks_api", os.path.join(ROOT, "vgen_amd", "api.py"))
        mod = importlib.util.module_from_spec(spec)
        mod._SO_OVERRIDE = HOOKS_SO
        sys.modules["vgen_amd_hooks_api"] = mod     # (dataclasses look their module up there)
        spec.loader.exec_module(mod)
        assert mod.library_path() == HOOKS_SO
        _hooks = mod
    return _hooks


@pytest.fixture(scope="module")
def vgh():
    """The ctypes view over the test build (fault injection available)."""
    api = hooks_api()
    assert api.device_count() >= 1, "no HIP device: the gpu-marked tests need an MI355X"
    return api
