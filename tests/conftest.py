import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "soak: long-running repeat of a case the suite already covers at a smaller size or outside "
                                       "SURVEY 8 (Float32 top-r at 2^23, a captured iteration at n = 1.5e7); opt-in: SPX_SOAK=1")


def pytest_collection_modifyitems(config, items):
    # The driver kills `pytest -m gpu` at a fixed limit: the soak cases stay out of it unless asked for (SPX_SOAK=1).  Every
    # coverage row of SURVEY 8 keeps its oracle comparison without them (profiles/r04_pytest_durations.txt).
    if os.environ.get("SPX_SOAK") == "1":
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("soak") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def kats():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.build()
    return oracle
