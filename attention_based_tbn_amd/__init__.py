"""MI355X-native TBN hot path (BN-Inception backbones -> attention-weighted mid-fusion -> temporal
consensus) behind the reference's `core.models` API.  Heavy math: libtbn_hip.so (hand-written HIP
for gfx950); this package is the Python host side."""
from .config import load_config, get_modality  # noqa: F401

__version__ = "0.1.0"
