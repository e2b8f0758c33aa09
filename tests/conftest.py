import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


import pytest  # noqa: E402


@pytest.fixture(autouse=True)
def no_split_clamps(request):
    """Every GPU test: no split-K launch of a plan outgrows the scratch the plan sized for it (skf_split_clamps; the
    stand-alone products on a caller's scratch are not counted).  The emulator modules assert the same at their end."""
    if request.node.get_closest_marker('gpu') is None:
        yield
        return
    import skfusion_amd._native as nat
    from skfusion_amd._engine import split_clamps
    rt = nat.get_runtime()
    before = split_clamps(rt)
    yield
    assert split_clamps(rt) == before


def pytest_sessionfinish(session, exitstatus):
    """Measured deviations of the parity tests next to their bounds (tests/helpers.py `within`)."""
    try:
        import helpers
    except Exception:
        return
    if not helpers.DEVIATIONS:
        return
    out = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'test_deviations.txt'), 'w') as fh:
            fh.write('%-90s %12s %12s %8s\n' % ('check', 'measured', 'bound', 'bound/m'))
            for what, v, b in helpers.DEVIATIONS:
                fh.write('%-90s %12.3e %12.3e %8.1f\n' % (what[:90], v, b, b / v if v > 0 else float('inf')))
    except OSError:
        pass
