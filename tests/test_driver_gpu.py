"""The reference's lipophilicity driver assembly (test_lipo.py:100-152), trained for a few steps on
the HIP path and on the CPU oracle from the same initial state: losses must track each other."""
import copy
import os
import sys

import pytest
import torch
from torch import nn, optim

from conftest import REPO

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(REPO, "examples"))


class OracleLipo(nn.Module):
    """CPU twin of GraphWrapper(lipo BasicModel): same parameters, forward = oracle restatement."""

    def __init__(self, gpu_module, steps):
        super().__init__()
        self.steps = steps
        self.params = nn.ParameterDict()
        self.bufs = {}
        first = {}
        for k, v in gpu_module.state_dict(keep_vars=True).items():
            if isinstance(v, nn.Parameter):
                if id(v) not in first:            # the 50 tower aliases are one tensor: one CPU parameter
                    first[id(v)] = k
                    self.params[k.replace(".", "/")] = nn.Parameter(v.detach().cpu().clone())
            else:
                self.bufs[k] = v.detach().cpu().clone()
        self.src = gpu_module

    def state(self):
        sd = {}
        seen = {}
        for k, v in self.src.state_dict(keep_vars=True).items():
            if isinstance(v, nn.Parameter):
                first = seen.setdefault(id(v), k)
                sd[k] = self.params[first.replace(".", "/")]
            else:
                sd[k] = self.bufs[k]
        return sd

    def forward(self, batch):
        from oracle import dense_ref as O
        out, buf = O.lipo_model_forward(self.state(), batch, steps=self.steps, training=self.training,
                                        return_buffers=True)
        if self.training:
            self.bufs.update(buf)
        return out


def test_lipo_driver_trains_like_the_oracle():
    import train_lipo as T
    dev = torch.device("cuda:0")
    torch.manual_seed(317)
    steps = 3
    gpu = T.build_model(steps)
    cpu_tail = copy.deepcopy(nn.Sequential(gpu[1], gpu[2]))
    cpu = nn.Sequential(OracleLipo(gpu[0], steps), cpu_tail[0], cpu_tail[1])
    gpu = gpu.to(dev)
    batches_gpu = T.make_batches(64, 16, 317, dev)
    batches_cpu = [{k: v.cpu() for k, v in b.items()} for b in batches_gpu]
    crit = nn.MSELoss()
    og = optim.Adam(gpu.parameters(), lr=1e-2, weight_decay=1e-4)
    oc = optim.Adam(cpu.parameters(), lr=1e-2, weight_decay=1e-4)
    gpu.train()
    cpu.train()
    lg, lc = [], []
    for bg, bc in zip(batches_gpu, batches_cpu):
        og.zero_grad()
        loss = crit(gpu(bg), bg["labels"].unsqueeze(-1))
        loss.backward()
        og.step()
        lg.append(loss.item())
        oc.zero_grad()
        loss = crit(cpu(bc), bc["labels"].unsqueeze(-1))
        loss.backward()
        oc.step()
        lc.append(loss.item())
    assert len(lg) == 4
    # first step: same parameters, pure forward parity; later steps compound Adam on fp32-rounded grads
    assert abs(lg[0] - lc[0]) < 1e-4 * max(1.0, abs(lc[0]))
    for a, b in zip(lg, lc):
        assert abs(a - b) < 2e-2 * max(1.0, abs(b)), (lg, lc)
