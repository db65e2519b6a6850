"""GraphWrapper variant of the lipophilicity driver: numeric atom features are batch-normalised
(masked) and appended to the categorical ones before the model runs.
Reference: models/graph_norm_wrapper.py:6-13."""
import torch
from torch import nn

from .mask_batch_norm import MaskBatchNorm1d


class GraphWrapper(nn.Module):
    def __init__(self, graph_model, norm_features):
        super().__init__()
        self.bn = MaskBatchNorm1d(norm_features)
        self.add_module('graph_model', graph_model)

    def forward(self, graph_batch):
        mask = graph_batch['mask']
        atoms = torch.cat([graph_batch['afm'], self.bn(graph_batch['nafm'], mask)], dim=-1)
        g = graph_batch.get('graph')
        if g is not None:
            return self.graph_model.forward(atoms, g, g, mask)
        return self.graph_model.forward(atoms, graph_batch['bfm'], graph_batch['adj'], mask)
