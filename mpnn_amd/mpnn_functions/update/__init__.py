"""Node update (operator slot `update_func`): GRUUpdate = masked GRU cell on mpnn_gru_update_f32 (+ its fused backward)."""
from . import gru_update as _gru

GRUUpdate = _gru.GRUUpdate
GRUCell = _gru.GRUCell

__all__ = ["GRUUpdate"]
