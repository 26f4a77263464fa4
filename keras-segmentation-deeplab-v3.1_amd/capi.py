"""ctypes binding of libdl3.so — the C-ABI drop-in boundary (include/dl3.h).

The prototypes are parsed from include/dl3.h itself, so the header is the single source of
truth: every declared symbol must resolve in the shared library or loading fails loudly.
There is NO fallback: if libdl3.so is missing (or a symbol is), ``lib()`` raises — the product
path never routes through a CPU implementation.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(_ROOT, "include", "dl3.h")
LIBPATH = os.environ.get("DL3_LIBPATH") or os.path.join(_HERE, "libdl3.so")  # DL3_LIBPATH: A/B builds (tuning aid)

ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2
IMPL_AUTO, IMPL_GATHER, IMPL_MARCH = 0, 1, 2
LABEL_U8, LABEL_I32 = 0, 1

_CTYPES = {
    "const float *": ctypes.c_void_p,
    "float *": ctypes.c_void_p,
    "int *": ctypes.c_void_p,
    "const int *": ctypes.c_void_p,
    "void *": ctypes.c_void_p,
    "const void *": ctypes.c_void_p,
    "void * *": ctypes.POINTER(ctypes.c_void_p),
    "const long long *": ctypes.c_void_p,
    "unsigned long long *": ctypes.c_void_p,
    "const unsigned long long *": ctypes.c_void_p,
    "const char *": ctypes.c_char_p,
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "size_t": ctypes.c_size_t,
    "unsigned long long": ctypes.c_ulonglong,
    "void": None,
}


def _norm_type(t):
    t = re.sub(r"\s+", " ", t.strip())
    t = t.replace(" *", "*").replace("*", " *")
    return t.strip()


def parse_header(path=HEADER):
    """Return {name: (restype_str, [(argtype_str, argname), ...])} for every prototype."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(dl3_\w+)\s*\(([^;{]*?)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        ret = _norm_type(ret)
        alist = []
        args = args.strip()
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)(\w+)$", a)
                alist.append((_norm_type(mm.group(1)), mm.group(2)))
        protos[name] = (ret, alist)
    return protos


class DL3Error(RuntimeError):
    pass


_lib = None
_protos = None


def lib():
    """Load libdl3.so (once) and attach argtypes/restype for every header prototype."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so); it must be loaded first so that libdl3.so
    # binds to the SAME runtime instance whose streams and device pointers it is handed
    import torch  # noqa: F401
    if torch.cuda.is_available():
        torch.cuda.init()
    if not os.path.exists(LIBPATH):
        raise DL3Error(
            "libdl3.so not found at %s — build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIBPATH)
    L = ctypes.CDLL(LIBPATH)
    _protos = parse_header()
    for name, (ret, args) in _protos.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            raise DL3Error("libdl3.so does not export %s declared in include/dl3.h" % name)
        fn.restype = _CTYPES[ret]
        fn.argtypes = [_CTYPES[t] for t, _ in args]
    _lib = L
    return L


def protos():
    lib()
    return _protos


MATH_MODES = {None: -1, "env": -1, "f32": 0, "split": 1}


def set_gemm_math(mode):
    """matrix math of the 1x1-conv GEMM launches issued from now on (include/dl3.h dl3_set_gemm_math): 'f32' (the f32
    MFMA), 'split' (exact 3-way bf16 split on the bf16 MFMA, fp32-roundoff-class error) or None (follow DL3_GEMM_MATH).
    Process-wide; an engine keeps the mode its hipGraph was captured with."""
    if mode not in MATH_MODES:
        raise ValueError("matrix math must be one of 'f32', 'split', None")
    check(lib().dl3_set_gemm_math(MATH_MODES[mode]), "dl3_set_gemm_math")


def get_gemm_math():
    return "split" if lib().dl3_get_gemm_math() == 1 else "f32"


def check(rc, name="dl3"):
    if rc != 0:
        msg = lib().dl3_last_error().decode()
        raise DL3Error("%s failed (rc=%d): %s" % (name, rc, msg))


def ptr(t, offset=0):
    """Device pointer of a torch tensor (+ element offset), or None."""
    if t is None:
        return None
    return t.data_ptr() + 4 * offset


def call(name, *args):
    """Call an int-returning op and raise on a non-zero status."""
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        check(rc, name)
