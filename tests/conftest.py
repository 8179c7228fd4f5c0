import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _cpu_share():
    """CPUs this process may use: the cgroup quota when there is one (the GPU box shows 128 hardware threads but grants a
    16-CPU share), else the affinity mask."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return max(1, int(round(int(quota) / int(period))))
    except Exception:
        pass
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracles (NumPy/OpenBLAS, C/OpenMP) must not run teams larger than the CPU share: they get throttled, which makes
    # the suite several times slower and stalls the GPU-driving thread of the same process as well.
    share = _cpu_share()
    os.environ.setdefault("OMP_NUM_THREADS", str(share))            # inherited by the tests' child processes
    os.environ.setdefault("OPENBLAS_NUM_THREADS", str(share))
    try:
        from threadpoolctl import threadpool_limits
        config._bh_blas_cap = threadpool_limits(limits=share)
    except Exception:
        pass


@pytest.fixture(scope="session")
def bh():
    """The product package with the HIP library initialised on cuda:0 (GPU tests only)."""
    import benlsip_jl_amd as pkg
    pkg.init(0)
    rc = pkg._lib.lib().bh_selftest()
    assert rc == 0, pkg._lib.lib().bh_last_error_detail()
    return pkg


@pytest.fixture(scope="session")
def ref():
    import benlsip_ref
    return benlsip_ref


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """How much of each tolerance the session used (tests/_util.py::note_tol): worst value / bound per tolerance kind."""
    try:
        from _util import TOL_USED
    except Exception:
        return
    if not TOL_USED:
        return
    tr = terminalreporter
    tr.write_sep("-", "tolerance use (worst value / bound per kind)")
    for label, (n, worst, detail, (value, bound)) in sorted(TOL_USED.items()):
        tr.write_line("  %5.1f %%  %s  [%d checks; worst %.3e of %.3e%s]" % (100.0 * worst, label, n, value, bound, (", " + detail) if detail else ""))
