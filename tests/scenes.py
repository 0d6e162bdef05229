"""Scene helpers live in the package (chalkydri_amd/scenes.py); re-exported here for the tests."""
from chalkydri_amd.scenes import REF_CALIB, bench_stream, pinhole_calib, render_view, wall_layout  # noqa: F401
