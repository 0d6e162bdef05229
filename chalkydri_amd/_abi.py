"""ctypes mirror of include/chalkydri_hip.h and csrc/synth.h (POD structs only)."""
import ctypes as C

CK_OK = 0
CK_EINVAL, CK_ENOMEM, CK_EDEVICE, CK_ENODEVICE, CK_ECAPACITY, CK_EUNSUPPORTED = -1, -2, -3, -4, -5, -6
CK_MAX_FAMILIES = 4
CK_INVALID_LABEL = 0xFFFFFFFF


class ImageU8(C.Structure):
    _fields_ = [("buf", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("stride", C.c_int32)]


class Family(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("nbits", C.c_uint32), ("ncodes", C.c_uint32),
                ("codes", C.POINTER(C.c_uint64)), ("bit_x", C.POINTER(C.c_uint32)),
                ("bit_y", C.POINTER(C.c_uint32)), ("width_at_border", C.c_int32),
                ("total_width", C.c_int32), ("reversed_border", C.c_int32), ("min_hamming", C.c_uint32),
                ("n_upstream", C.c_uint32)]


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("max_batch", C.c_int32), ("device", C.c_int32),
                ("quad_decimate", C.c_int32), ("min_white_black_diff", C.c_int32),
                ("min_component_px", C.c_int32), ("min_cluster_pixels", C.c_int32),
                ("max_nmaxima", C.c_int32), ("cos_critical_rad", C.c_double),
                ("max_line_fit_mse", C.c_double), ("refine_edges", C.c_int32),
                ("decode_sharpening", C.c_double), ("max_hamming", C.c_int32), ("n_families", C.c_int32),
                ("families", C.POINTER(Family) * CK_MAX_FAMILIES), ("max_points_per_frame", C.c_int32),
                ("max_clusters_per_frame", C.c_int32), ("max_quads_per_frame", C.c_int32)]


class Detection(C.Structure):
    _fields_ = [("id", C.c_int32), ("hamming", C.c_int32), ("family", C.c_int32),
                ("decision_margin", C.c_float), ("c", C.c_double * 2), ("p", (C.c_double * 2) * 4)]


class ClusterPoint(C.Structure):
    _fields_ = [("x", C.c_uint16), ("y", C.c_uint16), ("gx", C.c_int8), ("gy", C.c_int8), ("pad", C.c_uint16)]


class Cluster(C.Structure):
    _fields_ = [("rep0", C.c_uint32), ("rep1", C.c_uint32), ("start", C.c_uint32), ("count", C.c_uint32)]


class Quad(C.Structure):
    _fields_ = [("p", (C.c_double * 2) * 4), ("reversed_border", C.c_int32), ("rep0", C.c_uint32),
                ("rep1", C.c_uint32)]


class StageMs(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("h2d", "threshold", "segment", "clusters", "quads", "decode", "d2h", "total")]


class Iso3(C.Structure):
    _fields_ = [("t", C.c_double * 3), ("q", C.c_double * 4)]


class SqpnpParams(C.Structure):
    _fields_ = [("max_iter", C.c_int32), ("tol_sq", C.c_double)]


class SqpnpProblem(C.Structure):
    _fields_ = [("n_tags", C.c_int32), ("n_bearings", C.c_int32), ("tag_offset", C.c_int32),
                ("bearing_offset", C.c_int32), ("robot_to_cam", Iso3), ("gyro", C.c_double),
                ("sign_change_error", C.c_double)]


class SqpnpResult(C.Structure):
    _fields_ = [("valid", C.c_int32), ("pad", C.c_int32), ("rot", C.c_double * 9), ("pos", C.c_double * 3),
                ("std_devs", C.c_double * 3), ("yaw", C.c_double), ("energy", C.c_double)]


class OpenCV5(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")]


class VisionMeasurement(C.Structure):
    _fields_ = [("pose_x", C.c_double), ("pose_y", C.c_double), ("pose_rot", C.c_double),
                ("std_x", C.c_double), ("std_y", C.c_double), ("std_rot", C.c_double), ("ts", C.c_uint64),
                ("camera_id", C.c_uint8), ("tag_count", C.c_uint8), ("reserved", C.c_uint8 * 6)]


class FieldTag(C.Structure):
    _fields_ = [("id", C.c_int32), ("pad", C.c_int32), ("pose", Iso3)]


class ProcessParams(C.Structure):
    _fields_ = [("cam", OpenCV5), ("robot_to_cam", Iso3), ("field", C.POINTER(FieldTag)),
                ("n_field", C.c_int32), ("camera_id", C.c_uint8), ("sign_change_error", C.c_double),
                ("sqpnp", SqpnpParams), ("allow_unverified_ids", C.c_int32)]


class SynthTag(C.Structure):
    _fields_ = [("family", C.c_int32), ("id", C.c_int32), ("H", C.c_double * 9),
                ("corners", (C.c_double * 2) * 4), ("center", C.c_double * 2)]


class SynthParams(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("width", "height", "n_tags", "min_side", "max_side", "max_tilt_t64",
                                         "noise_amp", "ramp_amp", "black", "white", "bg", "family_mode",
                                         "max_id")]


assert C.sizeof(VisionMeasurement) == 64  # crates/whacknet/src/lib.rs:92-95

# per-frame status bits (include/chalkydri_hip.h)
CK_FRAME_OK, CK_FRAME_POINTS_OVERFLOW, CK_FRAME_CLUSTERS_OVERFLOW, CK_FRAME_QUADS_OVERFLOW, CK_FRAME_DETS_OVERFLOW = 0, 1, 2, 4, 8
CK_FRAME_UNVERIFIED_ID = 16
