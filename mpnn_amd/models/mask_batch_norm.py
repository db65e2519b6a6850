"""Masked batch norms sitting between message and update in the lipo / attention models, on the HIP
kernels of csrc/masked_bn.hip (mpnn_masked_bn_fwd_f32 / _bwd_f32).

Reference: models/mask_batch_norm.py:5-38.  Quirks kept on purpose (parity):
* MaskBatchNorm1d divides by (sqrt(var) + eps) -- eps OUTSIDE the root -- uses the biased masked
  variance for the running estimate, and in eval mode divides by (running_var**0.5 + eps);
* MaskBatchNorm (no parameters) takes the mean of the UNMASKED sum over the masked count and
  normalises by sqrt(var + eps), eps = 1e-6.
Statistics are global reductions over all atoms of the batch: two masked column-reduction passes and a
normalise pass forward, one reduction pass and an elementwise pass backward.
"""
import torch
from torch import nn

from mpnn_amd import ops


def _flatten(tensor, mask):
    return tensor.reshape(-1, tensor.shape[-1]), mask.reshape(-1)


class MaskBatchNorm(nn.Module):
    sync_stats = False        # True: moments over all ranks of the default process group (parallel.synced_masked_batch_norm)

    def forward(self, tensor, mask, eps=1e-6):
        if self.sync_stats:
            from mpnn_amd import parallel
            return parallel.synced_masked_batch_norm(tensor, mask, None, None, eps, masked_mean=False, eps_inside=True)[0]
        y, mk = _flatten(tensor, mask)
        out, _, _ = ops.masked_batch_norm(y, mk, None, None, None, eps, ops.BN_EPS_INSIDE)
        return out.view(tensor.shape)


class MaskBatchNorm1d(nn.BatchNorm1d):
    sync_stats = False        # True (training mode): moments over all ranks of the default process group

    def forward(self, tensor, mask):
        y, mk = _flatten(tensor, mask)
        w, b = (self.weight, self.bias) if self.affine else (None, None)
        if self.sync_stats and (self.training or not self.track_running_stats):
            from mpnn_amd import parallel
            out, mean, var = parallel.synced_masked_batch_norm(tensor, mask, w, b, self.eps, masked_mean=True,
                                                               eps_inside=False)
            if self.track_running_stats:
                with torch.no_grad():
                    keep = 1 - self.momentum
                    self.running_mean = keep * self.running_mean + self.momentum * mean
                    self.running_var = keep * self.running_var + self.momentum * var
            return out
        if (not self.training) and self.track_running_stats:
            out, _, _ = ops.masked_batch_norm(y, mk, w, b, (self.running_mean, self.running_var), self.eps,
                                              ops.BN_MASKED_MEAN | ops.BN_USE_STATS)
        else:
            out, mean, var = ops.masked_batch_norm(y, mk, w, b, None, self.eps, ops.BN_MASKED_MEAN)
            if self.track_running_stats:
                with torch.no_grad():
                    keep = 1 - self.momentum
                    self.running_mean = keep * self.running_mean + self.momentum * mean
                    self.running_var = keep * self.running_var + self.momentum * var
        return out.view(tensor.shape)
