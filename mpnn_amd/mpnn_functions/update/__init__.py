from .gru_update import GRUUpdate, GRUCell

__all__ = ["GRUUpdate"]
