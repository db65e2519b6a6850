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
