"""chalkydri_amd — MI355X-native AprilTag detect + SQPnP pose hot path behind the reference's API.

Host-side mirror of the reference interface (crates/chalkydri-apriltags `Detector`, crates/chalkydri_sqpnp
`SqPnP`, crates/apriltags `AprilTags::process`) over the C ABI in include/chalkydri_hip.h.  All arithmetic on
the path runs in hand-written HIP kernels (csrc/*.hip, gfx950); there is no CPU fallback.
"""
from ._lib import ChalkydriError, default_config, family, lib  # noqa: F401

__all__ = ["ChalkydriError", "default_config", "family", "lib"]
