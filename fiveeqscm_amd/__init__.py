"""fiveeqscm_amd — MI355X-native ensemble engine for the five-equation FaIR model.

Hot path: Python host (torch-ROCm tensors) -> C ABI (include/fiveeq.h) -> one
hand-written HIP kernel per timestep on gfx950.  No CPU fallback for the engine.
"""
from .concentrations import calculate_hfc_conc  # noqa: F401  (reference-compatible, NumPy)

__all__ = ["calculate_hfc_conc"]
__version__ = "0.1.0"
