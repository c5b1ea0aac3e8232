import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_code_is_built():
    """Build libmugiq_hip.so / libmugiq_oracle.so in-tree if a fresh checkout has not done so yet (the .so files are
    git-ignored).  Building is not a fallback: the tests below still fail loudly if the library cannot be loaded."""
    lib = os.path.join(ROOT, "mugiq_amd", "libmugiq_hip.so")
    orc = os.path.join(ROOT, "oracle", "libmugiq_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def hip():
    """The product bindings, built in-tree. GPU tests fail loudly if the HIP library is missing."""
    import mugiq_amd
    return mugiq_amd


_MAXIMA = {}


@pytest.fixture(scope="session")
def record_max():
    """record_max(name, value): keep the largest `value` seen under `name` over the session; written to
    gpurun_out/parity_maxima.json at the end (the measured parity maxima quoted in DESIGN.md section 2)."""
    def rec(name, value):
        _MAXIMA[name] = max(float(value), _MAXIMA.get(name, 0.0))
    return rec


def pytest_sessionfinish(session, exitstatus):
    if _MAXIMA:
        import json
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_maxima.json"), "w") as f:
            json.dump(dict(sorted(_MAXIMA.items())), f, indent=1)
