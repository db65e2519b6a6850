import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_PARITY = {}


def record_parity(case, **measured):
    """Keep the MEASURED errors of a parity test (not just pass / fail): printed, and written at the end of the session to
    gpurun_out/r04_parity.json (MPNN_PARITY_OUT overrides), which is committed as profiles/r04_parity.json."""
    _PARITY.setdefault(case, {}).update({k: (float(v) if isinstance(v, (int, float)) else v) for k, v in measured.items()})
    print("parity[%s] %s" % (case, " ".join("%s=%.3g" % (k, v) if isinstance(v, float) else "%s=%s" % (k, v)
                                              for k, v in _PARITY[case].items())))


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY:
        return
    import json
    path = os.environ.get("MPNN_PARITY_OUT", os.path.join(REPO, "gpurun_out", "r04_parity.json"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        old = {}
        if os.path.exists(path):
            with open(path) as f:
                old = json.load(f)
        old.update(_PARITY)
        old["_math"] = os.environ.get("MPNN_GRU_MATH", "split (default)")
        with open(path, "w") as f:
            json.dump(old, f, indent=1, sort_keys=True)
    except OSError:
        pass


class Fixture:
    """One golden .npz split into inputs / params / outputs / grads (see make_golden.py)."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.raw = {k: z[k] for k in z.files}
        self.inputs, self.params, self.out, self.gin, self.gp, self.pre = {}, {}, {}, {}, {}, {}
        for k, v in self.raw.items():
            tv = torch.from_numpy(np.asarray(v)) if v.dtype.kind in "fiu" else v
            if k.startswith("in."):
                self.inputs[k[3:]] = tv
            elif k.startswith("p."):
                self.params[k[2:]] = tv
            elif k.startswith("pre."):
                self.pre[k[4:]] = tv
            elif k.startswith("g.in."):
                self.gin[k[5:]] = tv
            elif k.startswith("g.p."):
                self.gp[k[4:]] = tv
            elif k == "out":
                self.out[""] = tv
            elif k.startswith("out."):
                self.out[k[4:]] = tv
        self.cot = torch.from_numpy(self.raw["cot"]) if "cot" in self.raw else None
        alias = str(self.raw["alias"]) if "alias" in self.raw else ""
        if alias:
            for item in alias.split(";"):
                k, first = item.split("=")
                self.params[k] = self.params[first]


@pytest.fixture
def golden():
    return Fixture


def max_err(a, b):
    return float((torch.as_tensor(a).detach().double() - torch.as_tensor(b).detach().double()).abs().max())
