"""BiLiniearEdgeNetwork (sic): parameter-free bilinear message, bond features ARE the tensor.

Reference: mpnn_functions/message/bilinear_edge_network.py:6-38.  Requires ef == nf**3 and is only
wired into an unused model (models/basic_model_ecfp.py:8); it has no sparse structure to exploit
(every pair carries nf^3 numbers), so it stays a dense torch contraction on the device.
"""
import torch
from torch import nn


class BiLiniearEdgeNetwork(nn.Module):
    def __init__(self, node_features, edge_features, message_features, activation_fn=None, attn_act=None):
        super().__init__()
        self.nf = node_features
        self.ef = edge_features
        self.mf = message_features
        self.act_fn = activation_fn if activation_fn is not None else nn.ReLU()

    def forward(self, afm, bfm, reuse_graph_tensors=False):
        B, N, nf = afm.shape
        T = bfm.view(B, N, N, nf, nf, nf)
        v = torch.einsum("bja,bijakc->bijkc", afm, T)
        return torch.einsum("bijkc,bic->bijk", v, afm).squeeze()
