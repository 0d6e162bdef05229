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
    assert L.ck_abi_version() == 3


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


def test_product_library_reads_no_diagnostic_knob(built):
    """The drop-in must not change what it returns with the environment: the measurement / path-forcing knobs (CK_*_STOP_AFTER,
    CK_FIT_SKIP, CK_FMERGE_CAP, CK_SEG_CHUNKS, ...) exist only in the -DCK_DIAG build (lib/diag/).  The product library names exactly
    two variables: CK_POISON (allocation fill / guard pages) and CK_STREAMS (post-segmentation stages on two streams: same bytes)."""
    import re
    from conftest import DIAG_LIB
    from chalkydri_amd import _lib
    prod = open(os.path.join(os.path.dirname(_lib.__file__), "lib", "libchalkydri_hip.so"), "rb").read()
    names = set(m.decode() for m in re.findall(rb"CK_[A-Z0-9_]{3,}(?=\x00)", prod))
    # (error-code names appear in ck_strerror's texts; anything else that looks like a knob is a failure)
    knobs = {n for n in names if not n.startswith(("CK_E", "CK_OK", "CK_FRAME_", "CK_HIP", "CK_ALLOC", "CK_ABI"))}
    assert knobs <= {"CK_POISON", "CK_STREAMS"}, knobs
    assert b"STOP_AFTER" not in prod and b"CK_FIT_" not in prod and b"CK_FMERGE" not in prod
    diag = open(DIAG_LIB, "rb").read()
    for k in (b"CK_TILE_STOP_AFTER", b"CK_FMERGE_CAP", b"CK_FIT_FLAT", b"CK_FIT_SKIP", b"CK_PARTS", b"CK_EMIT_STOP_AFTER"):
        assert k in diag, k


def test_fit_kernels_keep_their_register_budgets(built):
    """The split quad fit's kernels read other lanes' registers (v_readlane) around divergent code; a build of k_tail that spilled
    heavily once lost detections (the reads stood inside divergent blocks, where the spill code restores active lanes
    only).  The reads are all-lane now and the current build is verified by the full-size tests — this check makes a toolchain or
    source change that pushes a kernel over its budget fail HERE, on the CPU, instead of silently at 2448x2048: the code
    object's own metadata (llvm-readelf --notes) must show no spill in k_chunk and k_tile, at most the known 2 registers in
    k_tail, and no scratch at all in the segmentation kernels."""
    import shutil
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "llvm-objdump")):
        pytest.skip("no llvm-objdump in this image")
    build = os.path.join(ROOT, "chalkydri_amd", "csrc", "build")
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for obj in ("k_quads.o", "k_ccl.o"):
            shutil.copy(os.path.join(build, obj), td)
            subprocess.check_call([os.path.join(llvm, "llvm-objdump"), "--offloading", obj], cwd=td, stdout=subprocess.DEVNULL)
            co = [f for f in os.listdir(td) if f.startswith(obj) and "amdgcn" in f][0]
            notes = subprocess.check_output([os.path.join(llvm, "llvm-readelf"), "--notes", co], cwd=td, text=True)
            name = None
            for line in notes.splitlines():
                m = re.match(r"\s*\.name:\s+(\S+)", line)
                if m:
                    name = m.group(1)
                    res[name] = {}
                m = re.match(r"\s*\.(vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|vgpr_count):\s+(\d+)", line)
                if m and name:
                    res[name][m.group(1)] = int(m.group(2))
    pick = lambda frag: [(k, v) for k, v in res.items() if frag in k]
    assert pick("k_chunk") and pick("k_tail") and pick("k_tile") and pick("k_fmerge")
    for k, v in pick("k_chunk") + pick("k_tile") + pick("k_fmerge"):
        assert v["vgpr_spill_count"] == 0 and v["private_segment_fixed_size"] == 0, (k, v)
    for k, v in pick("k_tail"):
        assert v["vgpr_spill_count"] <= 2, (k, v)
    for k, v in pick("k_seq"):   # the sort kernels: budgets as measured; a jump means the class lost its occupancy step
        assert v["vgpr_spill_count"] <= 20, (k, v)
