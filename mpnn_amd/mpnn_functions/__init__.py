"""Operator API of the reference (mpnn_functions/__init__.py:1-4), backed by the HIP kernels."""
from .message import *            # noqa: F401,F403
from .update import *             # noqa: F401,F403
from .readout import *            # noqa: F401,F403
from .message_aggregators import *  # noqa: F401,F403
