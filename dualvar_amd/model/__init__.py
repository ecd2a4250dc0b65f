from .simclr import SimCLR_Naked, SimCLR_TimeSeriesV4  # noqa: F401
from .moco import MoCo_Naked, MoCo_TimeSeriesV4  # noqa: F401
from .classifier import LinearClassifier  # noqa: F401
