"""Optional roctx ranges around the phases of a training step (the reference has no tracing at all;
`rocprofv3 --marker-trace --kernel-trace` then attributes kernels to "denoise k=…", "frozen", "target fwd", "backward",
"all_reduce", "optimizer").  Enabled with ``LECO_ROCTX=1``; otherwise `push` / `pop` are two no-op calls per phase.

The ranges are host-side brackets around graph launches: they order against the GPU timeline through the launch API
calls inside them, which is what rocprofv3's marker domain records."""
import ctypes
import os

_lib = None
_on = os.environ.get("LECO_ROCTX", "0") not in ("", "0")


def _load():
    global _lib, _on
    if _lib is not None or not _on:
        return _lib
    for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
        for prefix in ("", "/opt/rocm/lib/"):
            try:
                lib = ctypes.CDLL(prefix + name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.restype = ctypes.c_int
                _lib = lib
                return _lib
            except (OSError, AttributeError):
                continue
    _on = False         # asked for, not available: stay silent and cheap
    return None


def enabled() -> bool:
    return _on and _load() is not None


def push(name: str) -> None:
    if _on and _load() is not None:
        _lib.roctxRangePushA(name.encode())


def pop() -> None:
    if _on and _lib is not None:
        _lib.roctxRangePop()
