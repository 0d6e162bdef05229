"""Loads the C-ABI product library (chalkydri_amd/lib/libchalkydri_hip.so).

There is deliberately no fallback: if the HIP library has not been built, importing anything that needs it
raises.  Build with `python -c "import __graft_entry__ as g; g.build()"` or `make -C chalkydri_amd/csrc`.
"""
import ctypes as C
import os

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
# CHALKYDRI_HIP_LIB selects another build of the library (same-box A/B measurements, out-of-tree installs)
LIB_PATH = os.environ.get("CHALKYDRI_HIP_LIB") or os.path.join(_HERE, "lib", "libchalkydri_hip.so")
_lib = None


class ChalkydriError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        msg = lib().ck_strerror(code).decode()
        if code == A.CK_EDEVICE:
            msg += ": " + lib().ck_last_error().decode()
        super().__init__(f"{where}: {msg} ({code})" if where else f"{msg} ({code})")


def check(code, where=""):
    if code != A.CK_OK:
        raise ChalkydriError(code, where)


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64; if it and the system one (which our library
    is linked against) both get loaded — which happens when this library is loaded before torch — the second one finds no
    GPU.  When torch is installed, its runtime is therefore put in place first, exactly as when torch is imported first
    (bench.py's order).  Without torch the system runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build the HIP extension first (no CPU fallback exists)")
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    L.ck_strerror.restype = C.c_char_p
    L.ck_strerror.argtypes = [C.c_int]
    L.ck_family_builtin.restype = P(A.Family)
    L.ck_family_builtin.argtypes = [C.c_char_p]
    L.ck_config_default.restype = None
    L.ck_config_default.argtypes = [P(A.Config), C.c_int32, C.c_int32, C.c_int32]
    L.ck_sqpnp_params_default.restype = None
    L.ck_sqpnp_params_default.argtypes = [P(A.SqpnpParams)]
    L.ck_synth_params_default.restype = None
    L.ck_synth_params_default.argtypes = [P(A.SynthParams), C.c_int32, C.c_int32, C.c_int32]
    L.ck_synth_render.restype = C.c_int
    L.ck_synth_render.argtypes = [C.c_uint64, P(A.SynthParams), P(P(A.Family)), C.c_int32, C.c_void_p,
                                  C.c_int32, P(A.SynthTag), C.c_int32, P(C.c_int32)]
    L.ck_synth_background.restype = None
    L.ck_synth_background.argtypes = [C.c_uint64, P(A.SynthParams), C.c_void_p, C.c_int32]
    L.ck_synth_draw_tag.restype = C.c_int
    L.ck_synth_draw_tag.argtypes = [P(A.SynthParams), P(A.Family), P(A.SynthTag), C.c_void_p, C.c_int32]
    L.ck_synth_fill_truth.restype = None
    L.ck_synth_fill_truth.argtypes = [P(A.SynthTag)]
    _lib = L
    return L


def family(name):
    f = lib().ck_family_builtin(name.encode())
    if not f:
        raise KeyError(name)
    return f


def default_config(width, height, max_batch=1, families=("tag36h11",), **overrides):
    cfg = A.Config()
    lib().ck_config_default(C.byref(cfg), width, height, max_batch)
    cfg.n_families = len(families)
    for i, name in enumerate(families):
        cfg.families[i] = family(name) if isinstance(name, str) else name
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg
