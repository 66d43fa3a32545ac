"""x265 lowres-lookahead hot path on MI355X: ctypes plumbing around libx265gpu.so."""
from . import api, synth  # noqa: F401
