"""MI355X-native D-FINE training hot path (package directory ``d-fine-seg_amd/``; import it
as ``dfine_seg_amd`` through the shim at the repository root).

HIP kernels + C ABI: ``csrc/`` / ``include/dfine_hip.h``; this package is the host-side
mirror of the reference's Python interfaces for that path. There is no CPU fallback.
"""
__all__ = ["ops"]
