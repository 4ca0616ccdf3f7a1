import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_PARITY = []      # (test id, quantity, achieved error, bound) of every parity comparison made in this session


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def parity(request):
    """`parity(quantity, achieved, bound)` records an achieved error next to its bound and returns achieved <= bound.  The
    session's records are printed as a table at the end of the run (also under -q) and written to
    gpurun_out/parity_report.json, so the margin under every tolerance is on record."""
    def rec(quantity, achieved, bound):
        achieved, bound = float(achieved), float(bound)
        _PARITY.append((request.node.nodeid.split("::", 1)[-1], quantity, achieved, bound))
        return achieved <= bound
    return rec


def pytest_terminal_summary(terminalreporter):
    if not _PARITY:
        return
    worst = {}
    for test, q, a, b in _PARITY:
        key = (test, q.split("/")[0])
        if key not in worst or a / max(b, 1e-30) > worst[key][0] / max(worst[key][1], 1e-30):
            worst[key] = (a, b, q)
    terminalreporter.write_sep("-", f"parity: worst achieved error per test and quantity group ({len(_PARITY)} comparisons)")
    for (test, grp), (a, b, q) in sorted(worst.items()):
        terminalreporter.write_line(f"{test:70s} {q:44s} {a:10.3e}  (bound {b:.1e}, {100 * a / max(b, 1e-30):5.1f} %)")
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report.json"), "w") as f:
            json.dump([dict(test=t, quantity=q, achieved=a, bound=b) for t, q, a, b in _PARITY], f)
    except OSError:
        pass
