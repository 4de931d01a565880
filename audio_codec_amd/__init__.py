"""audio_codec_amd -- MI355X-native LC3plus encode engine (gfx950 HIP kernels behind the ETSI lc3_enc_* C ABI).

The product is the C-ABI shared library ``liblc3plus_hip.so`` (sources in ``csrc/``; public headers in
``/include``).  This Python package is only a thin ctypes binding used by the tests and the benchmark."""
from .api import (Batch, DecBatch, Decoder, Encoder, LC3Error, lib_path, load_library)  # noqa: F401
from .build import build  # noqa: F401
