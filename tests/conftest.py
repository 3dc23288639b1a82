import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun / the driver's GPU tier)")
    config.addinivalue_line("markers", "slow: long CPU test")


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests skip (not fail) on a box without a GPU — the CPU tier runs `-m "not gpu"`, but a bare `pytest tests`
    must stay green there too."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:      # noqa: BLE001
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="needs a GPU (MI355X)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


PARITY = []        # (test id, label, measured, bound): written to gpurun_out/parity_summary_<operand build>.json when a GPU session ends


def record_parity(label, value, bound=None):
    """Measured parity error of the running test, kept for the session summary (the driver runs `pytest -q`: prints are swallowed, the file is not)."""
    PARITY.append(dict(test=os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], label=label, value=float(value), bound=None if bound is None else float(bound)))


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def pytest_sessionfinish(session, exitstatus):
    if not PARITY:
        return
    import json
    try:
        from pixart_sigma_amd import lib
        operand = lib.OPERAND
    except Exception:      # noqa: BLE001
        operand = "unknown"
    out_dir = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, f"parity_summary_{operand}.json")
        old = json.load(open(path))["entries"] if os.path.exists(path) else []
        keep = [e for e in old if (e["test"], e["label"]) not in {(p["test"], p["label"]) for p in PARITY}]
        worst = max((e["value"] / e["bound"] for e in keep + PARITY if e.get("bound")), default=None)
        json.dump(dict(operand=operand, exitstatus=int(exitstatus), worst_value_over_bound=worst, entries=keep + PARITY), open(path, "w"), indent=1)
    except OSError:
        pass


@pytest.fixture(scope="session")
def golden():
    import torch

    def load(name):
        return torch.load(os.path.join(GOLDEN_DIR, name + ".pt"), weights_only=False)
    return load
