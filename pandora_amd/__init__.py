"""pandora_amd: MI355X-native ray-traversal hot path of mathijs727/pandora behind a C ABI.

`pandora_amd.capi` binds lib/libpandora_hip.so (hand-written gfx950 kernels + host builders);
`pandora_amd.accel` drives it for ray batches; `pandora_amd.scenes` generates synthetic scenes/rays.
"""
from . import capi, accel, scenes  # noqa: F401

__all__ = ["capi", "accel", "scenes"]
