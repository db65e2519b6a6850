"""Message functions (operator slot `message_func`): per-pair message m_ij from the source atom's features and the
bond features of the pair.  Backed by mpnn_edge_message_f32 and its gradient kernels.

    EdgeNetwork            m_ij = A(e_ij) h_j, A from the bond-feature tower
    AttEdgeNetwork         the same on a feature-gated source row (gate from h_i and e_ij)
    GGNNMsgPass            A looked up in a table by integer bond type
    BiLiniearEdgeNetwork   parameter-free bilinear form of h_i, h_j with a per-pair tensor
"""
from . import att_edge_network as _att
from . import bilinear_edge_network as _bil
from . import edge_network as _en
from . import ggnn_msg_pass as _ggnn

EdgeNetwork = _en.EdgeNetwork
AttEdgeNetwork = _att.AttEdgeNetwork
GGNNMsgPass = _ggnn.GGNNMsgPass
BiLiniearEdgeNetwork = _bil.BiLiniearEdgeNetwork        # the reference's spelling

__all__ = ["AttEdgeNetwork", "EdgeNetwork", "BiLiniearEdgeNetwork", "GGNNMsgPass"]
