from .graph_level_output import GraphLevelOutput

__all__ = ["GraphLevelOutput"]
