from .adjacent_message_agg import AdjMsgAgg
from .weighted_adjacent_message_agg import WAdjMsgAgg
from .attention_message_agg import AttMsgAgg

__all__ = ["AdjMsgAgg", "WAdjMsgAgg", "AttMsgAgg"]
