"""The C-ABI library loads without a GPU, exports every symbol the header declares, keeps the reference's POD layouts and
fails loudly (no CPU fallback) when no HIP device is present."""
import ctypes as C
import os
import re

import pytest

from chalkydri_amd import _abi as A
from chalkydri_amd import _lib, default_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "chalkydri_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ck_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(built):
    L = _lib.lib()
    names = _declared()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_struct_layouts():
    assert C.sizeof(A.VisionMeasurement) == 64                      # crates/whacknet/src/lib.rs:92-95
    assert C.sizeof(A.ImageU8) == 24 and A.ImageU8.width.offset == 8 and A.ImageU8.stride.offset == 16   # image_u8_t
    assert C.sizeof(A.Detection) == 96 and C.sizeof(A.ClusterPoint) == 8 and C.sizeof(A.Cluster) == 16
    assert C.sizeof(A.Iso3) == 56 and C.sizeof(A.SqpnpResult) == 144
    L = _lib.lib()
    assert L.ck_abi_version() == 2


def test_defaults_mirror_the_reference(built):
    cfg = default_config(1280, 800)
    # DetectorBuilder::default + add_family_bits(tag36h11, 3) (crates/apriltags/src/lib.rs:45,230,258-262) on AprilTag-3 defaults
    assert (cfg.min_white_black_diff, cfg.max_nmaxima, cfg.refine_edges, cfg.max_hamming, cfg.n_families) == (5, 10, 1, 3, 1)
    assert abs(cfg.decode_sharpening - 0.25) < 1e-15 and abs(cfg.max_line_fit_mse - 10.0) < 1e-15
    assert cfg.families[0].contents.name == b"tag36h11"
    prm = A.SqpnpParams()
    _lib.lib().ck_sqpnp_params_default(C.byref(prm))
    assert prm.max_iter == 15 and prm.tol_sq == 1e-16               # chalkydri_sqpnp/src/lib.rs:203-204


def test_no_device_means_loud_failure(built):
    from chalkydri_amd.detector import AprilTagDetector, device_count
    if device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(_lib.ChalkydriError) as e:
        AprilTagDetector(640, 480)
    assert e.value.code == A.CK_ENODEVICE
    assert b"no CPU fallback" in _lib.lib().ck_strerror(A.CK_ENODEVICE)


def test_product_package_never_touches_the_oracle():
    """No import, link, dlopen or call of anything under oracle/ from the product (comments may cite oracle functions)."""
    pkg = os.path.join(ROOT, "chalkydri_amd")
    for dp, _, fs in os.walk(pkg):
        if os.sep + "build" in dp:
            continue
        for f in fs:
            path = os.path.join(dp, f)
            if f.endswith((".hip", ".c", ".h", ".cpp", ".hpp")):
                text = open(path, errors="ignore").read()
                code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
                code = re.sub(r"//[^\n]*", "", code)
                assert not re.search(r"\bora_[a-z0-9_]+\s*\(", code), path
                assert "ck_oracle" not in code and "oracle/" not in code, path
            elif f.endswith(".py") or f == "Makefile":
                text = open(path, errors="ignore").read()
                assert "pyoracle" not in text and "libck_oracle" not in text and "ck_oracle" not in text, path
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), path


def test_create_validates_before_touching_a_device(built):
    """Argument errors are reported without a GPU: the order in ck_create is validation first, device second."""
    import ctypes as C
    from chalkydri_amd import _abi as A, default_config
    from chalkydri_amd.detector import _bind
    from chalkydri_amd._lib import lib
    L = _bind(lib())
    h = C.c_void_p()

    def rc(**kw):
        size = kw.pop("size", (640, 480))
        cfg = default_config(size[0], size[1], **kw)
        return L.ck_create(C.byref(cfg), C.byref(h))
    assert rc(size=(8, 8)) == A.CK_EINVAL                       # smaller than the 16-pixel minimum
    assert rc(size=(5000, 480)) == A.CK_EINVAL                  # boundary points carry 13-bit half-pixel coordinates
    assert rc(quad_decimate=3) == A.CK_EUNSUPPORTED
    assert rc(min_component_px=0) == A.CK_EINVAL
    # any image_u8_t the reference accepts (crates/apriltags/src/lib.rs:204-209) is a valid geometry: widths that are not
    # multiples of 4, heights that leave 1..3 rows for the last tile row, large component thresholds
    for kw in ({"size": (642, 480)}, {"size": (641, 450)}, {"size": (640, 450)}, {"size": (130, 33)}, {"min_component_px": 200}):
        assert rc(**kw) in (A.CK_OK, A.CK_ENODEVICE), kw
        if h.value:
            L.ck_destroy(h)
            h.value = None
    assert rc() in (A.CK_OK, A.CK_ENODEVICE)                    # a valid config fails only for lack of a device
    if h.value:
        L.ck_destroy(h)
