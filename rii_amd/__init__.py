"""rii_amd -- MI355X-native IVFPQ query engine behind the rii.Rii API (drop-in for the query hot path of
matsui528/rii).  Host code is Python over a C-ABI HIP library (include/rii_amd.h)."""
from .core import RiiGpu, RiiAmdError, build_library, library_path, host_simd_arch   # noqa: F401
from .api import Rii, estimate_best_threshold_function                               # noqa: F401
from . import codec                                                                    # noqa: F401

__version__ = "0.1.0"
