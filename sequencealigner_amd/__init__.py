"""sequencealigner_amd -- MI355X-native Needleman-Wunsch engine (score + traceback).

Thin ctypes binding over the C ABI in include/sa_hip.h.  The compute path is
the HIP library only: importing works without a GPU (the library loads and its
symbols resolve), every compute call fails loudly when no device is present.
"""
from ._lib import (  # noqa: F401
    ADJACENT,
    TRIANGULAR,
    DEVICE_NONE,
    ALPHABET,
    Aligner,
    SaError,
    lib_path,
    load_library,
    similarity_bp,
    Settings,
    RunReport,
    default_settings,
    load_settings,
    validate_settings,
    run_csv,
)
