"""MI355X-native AD-Census matcher: host-side mirror of the reference interface over the C-ABI.

The directory name carries a hyphen, so import it with
    importlib.import_module("stereo-matching_amd")
(with the repository root on sys.path).
"""
from . import adcensus  # noqa: F401
from .adcensus import (ADCensusOption, ADCensusStereo, BUFFERS, EXPORTS, HARNESS_EXPORTS, LIB_PATH, STAGES,  # noqa: F401
                       lib, match_batch)
