"""MI355X-native NRD denoising hot path: C-ABI library (lib/libNRD_hip.so) + thin ctypes plumbing."""
from . import api  # noqa: F401
