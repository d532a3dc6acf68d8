import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """A HipLd context on cuda:0 -- fails loudly (no CPU fallback) when there is no device."""
    import tomahawk_amd as T
    if T.device_count() < 1:
        pytest.fail("no HIP device visible: GPU tests cannot fall back to the CPU")
    eng = T.HipLd(0)
    yield eng
    eng.close()


@pytest.fixture
def opt(hip):
    """Engine switches for one test (twk_hip_set_option through the binding); whatever a test set is back at its
    default when the test ends, pass or fail."""
    class Switches:
        def set(self, key, value):
            hip.set_option(key, int(value))

        def unset(self, key):
            hip.unset_option(key)
    yield Switches()
    for key in list(hip._defaults):
        hip.unset_option(key)


def pytest_sessionfinish(session, exitstatus):
    """Account for every exemption the parity checker granted in this session (tests/util.py): totals and rates go to
    gpurun_out/parity_exemptions.json (TWK_PARITY_EXEMPTIONS overrides the path; the round's copy is committed as
    profiles/rNN_parity_exemptions.json) and to the terminal, and a rate above its cap fails the session."""
    import json
    from tests import util
    if not util.COMPARED["records"]:
        return
    summary, bad = util.exemption_summary()
    path = os.environ.get("TWK_PARITY_EXEMPTIONS") or os.path.join(ROOT, "gpurun_out", "parity_exemptions.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(summary, fh, indent=1, sort_keys=True)
    except OSError:
        pass
    tr = session.config.pluginmanager.get_plugin("terminalreporter")
    line = (f"parity exemptions: {summary['compared'].get('records', 0)} records compared "
            f"({summary['compared'].get('cubic', 0)} from the cubic) in {summary['compared'].get('calls', 0)} calls; granted: "
            + (", ".join(f"{k}={v}" for k, v in sorted(summary["exemptions"].items()) if v) or "none"))
    if tr:
        tr.write_line(line)
        for b in bad:
            tr.write_line("PARITY EXEMPTION CAP EXCEEDED: " + b, red=True)
    if bad:
        session.exitstatus = 1
