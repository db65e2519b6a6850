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


def test_lipo_driver_every_step_is_a_parity_check():
    """Teacher-forced variant of the run above: before every step the HIP model takes over the oracle's parameters and
    running statistics, so EVERY step (not only the first) compares one forward + backward on identical state: loss to
    1e-4, every parameter gradient to 2e-3 of that gradient's largest entry (fp32 through the 52-layer bond tower)."""
    import train_lipo as T
    dev = torch.device("cuda:0")
    torch.manual_seed(317)
    steps = 3
    gpu = T.build_model(steps)
    cpu_tail = copy.deepcopy(nn.Sequential(gpu[1], gpu[2]))
    oracle = OracleLipo(gpu[0], steps)
    cpu = nn.Sequential(oracle, cpu_tail[0], cpu_tail[1])
    gpu = gpu.to(dev)
    batches_gpu = T.make_batches(64, 16, 317, dev)
    batches_cpu = [{k: v.cpu() for k, v in b.items()} for b in batches_gpu]
    crit = nn.MSELoss()
    oc = optim.Adam(cpu.parameters(), lr=1e-2, weight_decay=1e-4)
    gpu.train()
    cpu.train()

    def sync():
        sd = {k: v.detach().clone() for k, v in oracle.state().items()}
        gpu[0].load_state_dict(sd)
        gpu[1].load_state_dict(cpu[1].state_dict())
        gpu[2].load_state_dict(cpu[2].state_dict())

    for step, (bg, bc) in enumerate(zip(batches_gpu, batches_cpu)):
        sync()
        oc.zero_grad()                     # (the oracle twin holds the HIP module as `src`: this clears its grads too)
        gpu.zero_grad()
        lc = crit(cpu(bc), bc["labels"].unsqueeze(-1))
        lc.backward()
        lg = crit(gpu(bg), bg["labels"].unsqueeze(-1))
        lg.backward()
        assert abs(lg.item() - lc.item()) < 1e-4 * max(1.0, abs(lc.item())), (step, lg.item(), lc.item())
        cpu_grads = {k.replace("/", "."): p.grad for k, p in oracle.params.items()}
        # gradients that are zero in exact arithmetic (a bias in front of a batch norm) are rounding noise on both sides:
        # every comparison gets a floor of 1e-4 of the largest gradient entry of the model
        floor = 1e-4 * max(float(g.abs().max()) for g in cpu_grads.values() if g is not None)
        seen = set()
        for k, p in gpu[0].named_parameters():
            if p.data_ptr() in seen or k not in cpu_grads or cpu_grads[k] is None:
                continue
            seen.add(p.data_ptr())
            want = cpu_grads[k]
            scale = max(floor, float(want.abs().max()))
            got = p.grad.cpu() if p.grad is not None else torch.zeros_like(want)   # no gradient at all == exactly zero
            assert float((got - want).abs().max()) / scale < 2e-3, (step, k, p.grad is None)
        for mg, mc in ((gpu[1], cpu[1]), (gpu[2], cpu[2])):
            for (k, pg), (_, pc) in zip(mg.named_parameters(), mc.named_parameters()):
                scale = max(floor, float(pc.grad.abs().max()))
                assert float((pg.grad.cpu() - pc.grad).abs().max()) / scale < 2e-3, (step, k)
        oc.step()
