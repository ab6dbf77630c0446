from .tal_assigner import TaskAlignedAssigner  # noqa: F401
from .atss_assigner import ATSSAssigner  # noqa: F401
from .anchor_generator import generate_anchors  # noqa: F401
