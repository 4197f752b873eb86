import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The round-end GPU run gives the whole `-m gpu` suite 1200 s.  Tests marked `heavy(est=seconds on a normal box)` run LAST,
# cheapest first.  Their estimates are summed when the session has collected its tests: a suite whose heavy tests alone are
# estimated above LH_TEST_EST_LIMIT_S (default 600 s - the ~220 ordinary GPU tests take another ~250 s) FAILS at collection -
# the suite must be cut or its limit argued, not silently outgrow the driver's lease.
# LH_TEST_BUDGET_S (default 0: off) is the old self-protection for a slow box: with a budget set, a heavy test skips itself
# - loudly - when the time already spent plus its own estimate, scaled by how much slower than estimated the heavy tests
# before it ran, would not fit.  Off by default: a skip must not be able to pass for coverage.
SESSION_T0 = time.time()
BUDGET_S = float(os.environ.get("LH_TEST_BUDGET_S", "0"))
EST_LIMIT_S = float(os.environ.get("LH_TEST_EST_LIMIT_S", "600"))
_ratios = []
_budget_skips = []  # heavy tests this session skipped for time: reported at the end, in capitals, and in gpurun_out/


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "heavy(est): a long test (est = seconds on a normal box): runs last, skips when "
                                       "the session's time budget (LH_TEST_BUDGET_S) would be exceeded")


def _est(item):
    m = item.get_closest_marker("heavy")
    return None if m is None else float(m.kwargs.get("est", m.args[0] if m.args else 60))


def pytest_collection_modifyitems(config, items):
    items.sort(key=lambda it: (0, 0.0) if _est(it) is None else (1, _est(it)))  # stable: everything else keeps its order


def pytest_collection_finish(session):
    total = sum(_est(it) or 0.0 for it in session.items)
    if EST_LIMIT_S > 0 and total > EST_LIMIT_S:
        raise pytest.UsageError("the heavy tests selected for this session are estimated at %.0f s together, above "
                                "LH_TEST_EST_LIMIT_S = %.0f s: the GPU suite would outgrow the 1200 s it is given" % (total, EST_LIMIT_S))


def slow_factor():
    """How much slower than estimated the last heavy tests ran (>= 1)."""
    return max([1.0] + _ratios[-4:])


@pytest.fixture(autouse=True)
def _time_budget(request):
    est = _est(request.node)
    if est is None:
        yield
        return
    elapsed = time.time() - SESSION_T0
    want = est * slow_factor() * 1.25
    if BUDGET_S > 0 and elapsed + want > BUDGET_S:
        _budget_skips.append({"test": request.node.nodeid, "elapsed_s": round(elapsed), "needs_s": round(want), "budget_s": BUDGET_S})
        pytest.skip("time budget: %.0f s into the session, this test needs ~%.0f s here (estimate %.0f s x %.1f), budget "
                    "%.0f s (LH_TEST_BUDGET_S)" % (elapsed, want, est, slow_factor(), BUDGET_S))
    t0 = time.time()
    yield
    _ratios.append(min(max((time.time() - t0) / est, 0.25), 20.0))


@pytest.fixture(scope="session")
def hl():
    import halo2_lasso_amd
    return halo2_lasso_amd


@pytest.fixture(scope="session")
def ctx(hl):
    """Device context; fails loudly (no fallback) when there is no GPU."""
    c = hl.Context(0)
    yield c
    c.close()


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """A green run must not pass for coverage when heavy tests dropped out for time: say so where nobody can miss it."""
    if not _budget_skips:
        return
    tr = terminalreporter
    tr.section("HEAVY TESTS SKIPPED FOR TIME (LH_TEST_BUDGET_S) - NOT EXERCISED IN THIS RUN", sep="!")
    for s in _budget_skips:
        tr.write_line("  BUDGET-SKIP %s (%d s into the session, needs ~%d s, budget %.0f s)"
                      % (s["test"], s["elapsed_s"], s["needs_s"], s["budget_s"]))
    tr.write_line("  %d heavy test(s) were NOT run; re-run with LH_TEST_BUDGET_S=0 to run everything" % len(_budget_skips))
    try:
        import json
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "heavy_budget_skips.json"), "w") as f:
            json.dump(_budget_skips, f, indent=1)
    except OSError:
        pass
