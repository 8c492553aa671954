"""ctypes binding of libmonkeynet_hip.so (the C-ABI declared in include/monkeynet_hip.h).

The prototypes are parsed from the header itself, so the Python side can never drift from the C-ABI.
There is NO fallback: if the shared library is missing or fails to load, importing any op raises.
"""
import ctypes
import os
import re

from . import knobs

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
HEADER = os.path.join(os.path.dirname(_PKG), "include", "monkeynet_hip.h")
DEFAULT_LIB = os.path.join(_PKG, "libmonkeynet_hip.so")

_CTYPES = {
    "int": ctypes.c_int, "long": ctypes.c_long, "size_t": ctypes.c_size_t, "double": ctypes.c_double,
    "float": ctypes.c_float, "void": None, "uint64_t": ctypes.c_uint64,
}


def parse_header(path=HEADER):
    """Return {name: (restype, [argtypes], [argnames])} for every function declared in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#.*", "", text)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int|size_t)\s+(mnk_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if "char" in ret else _CTYPES[ret]
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                nm = re.search(r"(\w+)$", a).group(1)
                ty = a[: a.rfind(nm)].strip()
                if "*" in ty:
                    base = ty.replace("const", "").replace("*", "").strip()
                    if base == "char":
                        argtypes.append(ctypes.c_char_p)
                    elif base in ("uint64_t", "double") and name.startswith("mnk_prof"):
                        argtypes.append(ctypes.POINTER(_CTYPES[base]))
                    else:
                        argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_CTYPES[ty.replace("const", "").strip()])
                argnames.append(nm)
        protos[name] = (restype, argtypes, argnames)
    return protos


class MnkError(RuntimeError):
    pass


class Library:
    def __init__(self, path, strict=True):
        if not os.path.exists(path):
            raise MnkError(
                "libmonkeynet_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or monkey-net_amd/csrc/build.sh).  There is no CPU fallback." % path)
        self.path = path
        # PyTorch-ROCm ships its own libamdhip64: load it first so that this library binds to the same HIP runtime as
        # the tensors it is handed (loading the system runtime first leaves two runtimes in the process, and launches
        # from the second one fail with "no ROCm-capable device is detected")
        try:
            import torch  # noqa: F401
        except ImportError:  # pragma: no cover - the C-ABI itself does not need torch
            pass
        self.cdll = ctypes.CDLL(path)
        self.protos = parse_header()
        for name, (restype, argtypes, _) in self.protos.items():
            if not strict and not hasattr(self.cdll, name):
                continue
            fn = getattr(self.cdll, name)      # AttributeError if a declared symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
        self.is_device_build = bool(self.cdll.mnk_is_device_build())
        # the generated CPython binding (csrc/gen_fastcall.py -> _mnkfast): the same entry points of THIS library, bound by
        # address, ~0.4 instead of ~3.5 us of host time per call.  A wrapper answers NotImplemented for an argument it does
        # not take (a ctypes object); such a call -- and every call when the module was not built -- goes through ctypes.
        self.fast = {}
        fast = _fast_module()
        if fast is not None:
            for name in fast.names():
                if name in self.protos and hasattr(self.cdll, name):
                    f = fast.bind(name, ctypes.cast(getattr(self.cdll, name), ctypes.c_void_p).value)
                    if f is not None:
                        self.fast[name] = f

    def call(self, name, *args):
        if name == "mnk_set_tuning":
            self.tuning_epoch = getattr(self, "tuning_epoch", 0) + 1     # cached tuning values (mnk.ops.subpixel) are re-read
        f = self.fast.get(name)
        rc = f(*args) if f is not None else NotImplemented
        if rc is NotImplemented:
            rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            raise MnkError("%s failed (%d): %s" % (name, rc, self.cdll.mnk_last_error().decode()))

    def query(self, name, *args):
        f = self.fast.get(name)
        r = f(*args) if f is not None else NotImplemented
        return getattr(self.cdll, name)(*args) if r is NotImplemented else r


def _fast_module():
    try:
        import _mnkfast                      # monkey-net_amd/_mnkfast*.so, next to libmonkeynet_hip.so (csrc/build.sh)
        return _mnkfast
    except ImportError:
        return None


_LIB = None


def lib():
    """The process-wide library handle (loaded on first use)."""
    global _LIB
    if _LIB is None:
        _LIB = Library(knobs.get("MNK_LIBRARY") or DEFAULT_LIB)
    return _LIB
