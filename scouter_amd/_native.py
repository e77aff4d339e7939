"""ctypes binding of libscouter_hip.so (the C ABI declared in include/scouter_hip.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised.  PyTorch is used
above this layer only for device memory, streams and torch.distributed."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCOUTER_HIP_LIB") or os.path.join(_HERE, "lib", "libscouter_hip.so")   # (env: dev builds)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "scouter_hip.h")

_lib = None

_CTYPES = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
    "size_t": ctypes.c_size_t, "void": None,
}


def declared_symbols():
    """[(name, restype, [argtypes])] parsed from include/scouter_hip.h -- the single source of truth."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = []
    for m in re.finditer(r"^\s*([a-z_ ]+?[\s\*]+)(scouter_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.M | re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3)

        def ctype(decl):
            decl = decl.strip()
            if "*" in decl:
                return ctypes.c_char_p if decl.startswith("const char") and name == "?" else ctypes.c_void_p
            base = decl.replace("const", "").replace("unsigned", "").split()
            return _CTYPES[base[0]]

        if ret.replace(" ", "") == "constchar*":
            restype = ctypes.c_char_p
        elif "*" in ret:
            restype = ctypes.c_void_p
        else:
            restype = _CTYPES[ret.replace("const", "").split()[0]]
        argtypes = [] if args.strip() in ("", "void") else [ctype(a) for a in args.split(",")]
        out.append((name, restype, argtypes))
    return out


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "scouter_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU/PyTorch fallback for the HIP path)" % LIB_PATH)
        import torch  # noqa: F401  -- first: the library must bind to the HIP runtime PyTorch loads (its bundled
        #                        libamdhip64), not bring a second runtime into the process ("no ROCm-capable device")
        _lib = ctypes.CDLL(LIB_PATH)
        for name, restype, argtypes in declared_symbols():
            fn = getattr(_lib, name)      # AttributeError here = header/library mismatch
            fn.restype = restype
            fn.argtypes = argtypes
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().scouter_last_error()
        raise RuntimeError("scouter_hip %s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))
