"""ctypes binding of libdinounet_hip.so (the C ABI declared in include/dinounet_hip.h).

There is no fallback: if the library is missing or a call fails, a RuntimeError is raised (the reference
extension raises from AT_ASSERTM the same way, ops/src/cuda/ms_deform_attn_cuda.cu:33-57)."""
import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libdinounet_hip.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "dinounet_hip.h")

DU_F32, DU_BF16 = 0, 1
PLAIN_ROW, PLAIN_COL, IM2COL_ROW, IM2COL_COL = 0, 1, 2, 3
ACT_NONE, ACT_GELU, ACT_RELU, ACT_LEAKY, ACT_SWIGLU = 0, 1, 2, 3, 4
STORE_PLAIN, STORE_PIXEL_SHUFFLE2, STORE_QKV_ROPE, STORE_SLABS, STORE_QKV_HEADS, STORE_MSDA_PREP = 0, 1, 2, 4, 5, 6
ERRORS = {-1: "DU_ERR_BAD_ARG", -2: "DU_ERR_UNSUPPORTED", -3: "DU_ERR_LAUNCH"}


class ConvGeom(C.Structure):
    _fields_ = [("p2", C.c_void_p), ("ld2", C.c_int64), ("C1", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32),
                ("C", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("Ho", C.c_int32), ("Wo", C.c_int32), ("transposed", C.c_int32)]


class GemmArgs(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("out_dtype", C.c_int32), ("a_mode", C.c_int32), ("b_mode", C.c_int32),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("A", C.c_void_p), ("lda", C.c_int64), ("a_batch_stride", C.c_int64),
                ("B", C.c_void_p), ("ldb", C.c_int64), ("b_batch_stride", C.c_int64),
                ("C", C.c_void_p), ("ldc", C.c_int64), ("c_batch_stride", C.c_int64),
                ("batch", C.c_int32), ("split_k", C.c_int32), ("alpha", C.c_float),
                ("bias", C.c_void_p), ("act", C.c_int32), ("gamma", C.c_void_p),
                ("row_scale", C.c_void_p), ("rs_rows", C.c_int32),
                ("residual", C.c_void_p), ("ldr", C.c_int64),
                ("store_mode", C.c_int32), ("ps_H", C.c_int32), ("ps_W", C.c_int32), ("ps_C", C.c_int32),
                ("geom", ConvGeom), ("ws", C.c_void_p), ("ws_elems", C.c_int64),
                ("rope_sin", C.c_void_p), ("rope_cos", C.c_void_p), ("rope_prefix", C.c_int32), ("rope_qscale", C.c_float),
                ("a_colsum", C.c_void_p), ("b_colsum", C.c_void_p), ("C2", C.c_void_p), ("ks_ws", C.c_void_p), ("ks_ws_bytes", C.c_int64)]


class TnJob(C.Structure):
    """du_tn_job (include/dinounet_hip.h): one queued weight-gradient product of du_gemm_tn_group."""
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("B", C.c_void_p), ("ldb", C.c_int64), ("C", C.c_void_p), ("ldc", C.c_int64),
                ("a_colsum", C.c_void_p), ("alpha", C.c_void_p), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("accumulate", C.c_int32), ("b_colsum", C.c_void_p),
                ("gather", C.c_int32), ("Hs", C.c_int32), ("Ws", C.c_int32), ("Cb", C.c_int32),
                ("taps", C.c_int32), ("inner", C.c_int32), ("inner_total", C.c_int32), ("c_off", C.c_int32)]


_CTYPE = {"int": C.c_int, "int64_t": C.c_int64, "float": C.c_float, "int32_t": C.c_int32}


def header_prototypes(path=HEADER):
    """Parse `int du_xxx(...);` prototypes from the public header -> {name: [ctypes argtypes]}."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int64_t|int|const char\*)\s+(du_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    types.append(C.c_void_p)
                else:
                    types.append(_CTYPE[a.split()[0] if not a.startswith("const") else a.split()[1]])
        protos[name] = (C.c_char_p if "char" in ret else (C.c_int64 if ret == "int64_t" else C.c_int), types)
    return protos


_lib = None


def lib():
    """Load the library (once).  Fails loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU / eager fallback exists). "
                "Build it with `python -m dinounet_amd._build` or `__graft_entry__.build()`.")
        l = C.CDLL(LIB_PATH)
        for name, (ret, types) in header_prototypes().items():
            fn = getattr(l, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = ret
            fn.argtypes = types
        _lib = l
        # A-B aids (tools/ab_bench.py): library options from the environment, e.g. DINOUNET_LIB_OPTIONS="10=1,9=2" -> du_set_option(10, 1), (9, 2)
        for kv in os.environ.get("DINOUNET_LIB_OPTIONS", "").split(","):
            if "=" in kv:
                k, v = kv.split("=", 1)
                l.du_set_option(int(k), int(v))
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"dinounet_hip: {what} failed with {ERRORS.get(rc, rc)}")
