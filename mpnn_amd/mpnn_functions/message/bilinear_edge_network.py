"""BiLiniearEdgeNetwork (sic): parameter-free bilinear message, bond features ARE the tensor.

Reference: mpnn_functions/message/bilinear_edge_network.py:6-38.  Requires ef == nf**3 and is only wired into an unused
model (models/basic_model_ecfp.py:8).  It has no sparse structure to exploit (every pair carries nf^3 numbers): the
forward is one HBM-bound HIP kernel over the dense padded batch (mpnn_bilinear_message_f32, nf <= 8: beyond that a
pair's tensor is > 2 KB and the batch does not fit anything); the gradients are the same contraction re-associated,
left to torch einsum on the device.
"""
import torch
from torch import nn

from mpnn_amd import _lib


class _Bilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, afm, bfm):
        lib = _lib.load()
        B, N, nf = (int(s) for s in afm.shape)
        afm_c, bfm_c = afm.contiguous().float(), bfm.contiguous().float()
        out = torch.empty(B, N, N, nf, dtype=torch.float32, device=afm.device)
        _lib.check(lib.mpnn_bilinear_message_f32(_lib.fptr(afm_c), _lib.fptr(bfm_c), _lib.fptr(out), B, N, nf,
                                                 _lib.stream()), "mpnn_bilinear_message_f32")
        ctx.save_for_backward(afm_c, bfm_c)
        return out

    @staticmethod
    def backward(ctx, dout):
        afm, bfm = ctx.saved_tensors
        B, N, nf = afm.shape
        T = bfm.view(B, N, N, nf, nf, nf)
        dafm = dbfm = None
        if ctx.needs_input_grad[0]:
            dafm = (torch.einsum("bijk,bijakc,bic->bja", dout, T, afm) + torch.einsum("bijk,bijakc,bja->bic", dout, T, afm))
        if ctx.needs_input_grad[1]:
            dbfm = torch.einsum("bja,bijk,bic->bijakc", afm, dout, afm).reshape(bfm.shape)
        return dafm, dbfm


class BiLiniearEdgeNetwork(nn.Module):
    def __init__(self, node_features, edge_features, message_features, activation_fn=None, attn_act=None):
        super().__init__()
        self.nf = node_features
        self.ef = edge_features
        self.mf = message_features
        self.act_fn = activation_fn if activation_fn is not None else nn.ReLU()

    def forward(self, afm, bfm, reuse_graph_tensors=False):
        nf = afm.shape[-1]
        if nf > 8:
            raise _lib.MpnnError("BiLiniearEdgeNetwork: nf = %d (> 8) is not supported" % nf)
        return _Bilinear.apply(afm, bfm).squeeze()
