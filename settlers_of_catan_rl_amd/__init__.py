"""settlers_of_catan_rl_amd - MI355X-native batched Catan self-play hot path (HIP kernels behind a C ABI)."""
from . import spec  # noqa: F401

__all__ = ["spec"]
