"""GRUUpdate: masked GRU node update, one fused HIP kernel (mpnn_gru_update_f32).

Reference: mpnn_functions/update/gru_update.py:5-68.  Parameter names, shapes ((in, 3H) weights,
gate order r,z,n) and initialisation follow GRUCell there so checkpoints interchange; the
wrapper's constructor-argument swap (`GRUCell(self.mf, self.nf)`, :53) is kept, which like the
reference makes the op well-formed only for message_features == node_features.
"""
import torch
from torch import nn

from mpnn_amd import ops


class GRUCell(nn.Module):
    def __init__(self, node_features, message_features):
        super().__init__()
        self.nf = node_features
        self.mf = message_features
        self.weight_ih = nn.Parameter(torch.empty(self.mf, 3 * self.nf))
        self.weight_hh = nn.Parameter(torch.empty(self.nf, 3 * self.nf))
        self.bias_ih = nn.Parameter(torch.empty(3 * self.mf))
        self.bias_hh = nn.Parameter(torch.empty(3 * self.nf))
        self.init_params()

    def init_params(self):
        gain = nn.init.calculate_gain('sigmoid')
        nn.init.xavier_uniform_(self.weight_ih, gain=gain)
        nn.init.xavier_uniform_(self.weight_hh, gain=gain)
        nn.init.zeros_(self.bias_ih)
        nn.init.zeros_(self.bias_hh)

    def forward(self, messages, node_states, mask):
        """(V,mf), (V,nf), (V,1) -> (V,nf); includes the wrapper's final `* mask`."""
        if self.mf != self.nf:
            raise RuntimeError("GRU update needs message_features == node_features (got %d, %d), as the "
                               "reference does (gru_update.py:53)" % (self.mf, self.nf))
        return ops.gru_update(messages, node_states, mask.reshape(-1), self.weight_ih, self.weight_hh,
                              self.bias_ih, self.bias_hh)


class GRUUpdate(nn.Module):
    def __init__(self, node_features, message_features):
        super().__init__()
        self.nf = node_features
        self.mf = message_features
        self.gru_cell = GRUCell(self.mf, self.nf)

    def forward(self, messages, node_states, mask):
        shape = node_states.shape
        out = self.gru_cell(messages.reshape(-1, self.mf), node_states.reshape(-1, self.nf), mask)
        return out.view(shape)
