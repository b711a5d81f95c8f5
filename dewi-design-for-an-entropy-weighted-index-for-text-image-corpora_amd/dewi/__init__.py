"""DEWI (entropy-weighted index) — MI355X-native scoring-and-retrieval hot path.

Drop-in for the reference's ``dewi.index`` + ``dewi.scorer`` + ``dewi.types`` (same
class names, signatures, defaults, errors and file formats), backed by hand-written HIP
kernels for gfx950 through the C ABI in ``include/dewi_hip.h``.
"""

__version__ = "0.1.0"

from .scorer import DewiScorer, RobustStats
from .types import Payload, Weights

__all__ = ["__version__", "DewiScorer", "RobustStats", "Weights", "Payload"]
