"""GraphWrapper: dict batch -> positional model call (reference: models/graph_model_wrapper.py:4-10).

A sparse-native batch carries a MolGraph under 'graph'; it is handed to the model in place of
both 'bfm' and 'adj' (atom arrays are then (V, .) instead of (B, N, .)).
"""
from torch import nn


class GraphWrapper(nn.Module):
    def __init__(self, graph_model):
        super().__init__()
        self.add_module('graph_model', graph_model)

    def forward(self, graph_batch):
        g = graph_batch.get('graph')
        if g is not None:
            return self.graph_model.forward(graph_batch['afm'], g, g, graph_batch['mask'])
        return self.graph_model.forward(graph_batch['afm'], graph_batch['bfm'], graph_batch['adj'],
                                        graph_batch['mask'])
