"""pytest configuration: `gpu` marker, package import under the name `foley_amd`, shared helpers."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test (opt-in with FOLEY_SLOW=1)")


def pytest_collection_modifyitems(config, items):
    have_gpu = torch.cuda.is_available()
    for it in items:
        if "gpu" in it.keywords and not have_gpu:
            it.add_marker(pytest.mark.skip(reason="no GPU visible"))
        if "slow" in it.keywords and not os.environ.get("FOLEY_SLOW"):
            it.add_marker(pytest.mark.skip(reason="set FOLEY_SLOW=1 to run"))


def golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiu" else z[k]) for k in z.files}


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def record_parity(name, **values):
    """Append one measured-parity record (errors, the reference's own noise floor d0, their ratio) to
    gpurun_out/parity_records.json (or $FOLEY_PARITY_JSON): the gates are inequalities, the records make movements INSIDE a
    gate visible from round to round (copied to profiles/rNN_parity.json)."""
    import json
    path = os.environ.get("FOLEY_PARITY_JSON") or os.path.join(ROOT, "gpurun_out", "parity_records.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    try:
        with open(path) as f:
            recs = json.load(f)
    except (OSError, ValueError):
        recs = {}
    recs[name] = {k: (float(v) if isinstance(v, (int, float)) else v) for k, v in values.items()}
    with open(path, "w") as f:
        json.dump(recs, f, indent=1, sort_keys=True)


@pytest.fixture(scope="session")
def dev():
    return torch.device("cuda:0")
