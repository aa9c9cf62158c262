"""ctypes binding of ``libnerfrpn_hip.so`` (C ABI in ``include/nerfrpn.h``).

The product path has no CPU fallback: if the shared library is missing or a call fails, this module raises.
``build()`` compiles the library in-tree (``make -C nerf_rpn_amd/csrc``); hipcc cross-compiles gfx950 without a GPU.
"""
import ctypes
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# NRPN_LIBRARY: load another build of the same ABI (A/B timing of two kernel versions on one GPU box); default = the in-tree library
SO_PATH = os.environ.get("NRPN_LIBRARY") or os.path.join(_HERE, "libnerfrpn_hip.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "nerfrpn.h")
TOOLS_HEADER = os.path.join(os.path.dirname(_HERE), "include", "nerfrpn_tools.h")     # measurement switches (nrpn_set_*): not the boundary

F32, BF16 = 0, 1
CONV_BIAS, CONV_RELU, CONV_OUT_F32 = 1, 2, 4

_lib = None

TILE_AUTO, TILE_128, TILE_256X128_WS, TILE_256X256, TILE_256X256_W4, TILE_HALO = 0, 128, 256, 512, 1024, 2048


class ConvOpts(ctypes.Structure):
    """``nrpn_conv_opts`` (include/nerfrpn.h): per-call plan + fused-epilogue extras of ``nrpn_conv3d_fwd_ex``."""
    _fields_ = [("size", ctypes.c_int32), ("tile", ctypes.c_int32), ("lds_dma", ctypes.c_int32), ("kstep_bytes", ctypes.c_int32),
                ("stagger", ctypes.c_int32), ("big_split", ctypes.c_int32), ("debug", ctypes.c_int32), ("halo_pairing", ctypes.c_int32),
                ("scale", ctypes.c_void_p), ("relu_mask", ctypes.c_void_p), ("stats", ctypes.c_void_p)]

    def __init__(self, tile=0, lds_dma=-1, kstep_bytes=0, stagger=-1, big_split=-1, debug=0, scale=0, relu_mask=0, stats=0, halo_pairing=0):
        super().__init__(ctypes.sizeof(ConvOpts), tile, lds_dma, kstep_bytes, stagger, big_split, debug, halo_pairing, scale or None,
                         relu_mask or None, stats or None)

    def ptr(self):
        return ctypes.addressof(self)


class NrpnError(RuntimeError):
    pass


def build(force=False):
    """Compile every HIP translation unit for gfx950 and link the shared library (in-tree)."""
    if force:
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "clean"])
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"])
    return SO_PATH


def declared_symbols(tools=True):
    """Every ``nrpn_*`` function declared in include/nerfrpn.h (+ include/nerfrpn_tools.h unless ``tools=False``); used by the ABI test."""
    text = open(HEADER).read() + (open(TOOLS_HEADER).read() if tools else "")
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nrpn_[a-z0-9_]+)\s*\(", text)))


def _ctype(decl):
    decl = decl.strip()
    if "*" in decl or decl.startswith("nrpn_stream_t"):
        return ctypes.c_void_p
    base = decl.replace("const", "").split()
    t = " ".join(base[:-1]) if len(base) > 1 else base[0]
    if t in ("int64_t", "long long", "size_t"):
        return ctypes.c_int64
    if t == "int":
        return ctypes.c_int
    if t == "float":
        return ctypes.c_float
    raise NrpnError(f"unmapped C type in nerfrpn.h: {decl!r}")


def _prototypes():
    """Parse include/nerfrpn.h (+ the tools header) into {name: (restype, [argtypes])} so ctypes converts and checks every argument."""
    text = open(HEADER).read() + open(TOOLS_HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for ret, name, args in re.findall(r"\b(int|size_t|int64_t|const char \*)\s*(nrpn_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        args = args.strip()
        argtypes = [] if args in ("", "void") else [_ctype(a) for a in args.split(",")]
        restype = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "int64_t": ctypes.c_int64,
                   "const char *": ctypes.c_char_p}[ret]
        protos[name] = (restype, argtypes)
    return protos


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise NrpnError(f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(there is no CPU fallback for the HIP path)")
    lib = ctypes.CDLL(SO_PATH)
    for name, (restype, argtypes) in _prototypes().items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = restype, argtypes
    _lib = lib
    return lib


KNOB_EPOCH = [0]      # bumped by every nrpn_set_* call: memoised size / plan queries (ops.query) are functions of (arguments, process knobs)


def call(name, *args):
    """Call ``nrpn_<name>`` and raise NrpnError with the library's message on a non-zero status."""
    lib = load()
    if name.startswith("set_"):
        KNOB_EPOCH[0] += 1
    rc = getattr(lib, "nrpn_" + name)(*args)
    if rc != 0:
        raise NrpnError(f"nrpn_{name} failed ({rc}): {lib.nrpn_last_error().decode()}")
    return rc


def query(name, *args):
    """Call a size query (returns a number, not a status)."""
    return getattr(load(), "nrpn_" + name)(*args)
