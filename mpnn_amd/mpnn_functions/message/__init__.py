from .att_edge_network import AttEdgeNetwork
from .edge_network import EdgeNetwork
from .bilinear_edge_network import BiLiniearEdgeNetwork
from .ggnn_msg_pass import GGNNMsgPass

__all__ = ["AttEdgeNetwork", "EdgeNetwork", "BiLiniearEdgeNetwork", "GGNNMsgPass"]
