"""MI355X-native AD-Census matcher: host-side mirror of the reference interface over the C-ABI.

The directory name carries a hyphen, so import it with
    importlib.import_module("stereo-matching_amd")
(with the repository root on sys.path).
"""
from . import adcensus  # noqa: F401
from .adcensus import (ADC_ERR_ARG, ADC_ERR_HIP, ADC_ERR_NOMEM, ADC_ERR_NOT_INIT, ADC_ERR_UNSUPPORTED, ADC_OK,  # noqa: F401
                       ADCensusOption, ADCensusStereo, BUFFERS, EXPORTS, HARNESS_EXPORTS, KERNELS, LIB_PATH, STAGES,
                       STATIC_SCRATCH_BYTES, lib, match_batch)
