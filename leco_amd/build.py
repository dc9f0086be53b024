"""In-tree build of ``libleco_hip.so`` (hipcc, gfx950 only).  Cross-compiles without a GPU."""
from __future__ import annotations

import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libleco_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-I", os.path.join(CSRC, "prims")]


def _sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".cpp"))]


def _digest(src: str) -> str:
    h = hashlib.sha1(" ".join(FLAGS).encode())
    deps = [src, os.path.join(ROOT, "include", "leco_hip.h"), os.path.join(CSRC, "common.h"),
            os.path.join(CSRC, "prims", "leco_prims.h")]
    for d in deps:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build(verbose: bool = False) -> str:
    """Incremental, content-hashed; serialised across processes with a file lock (torchrun starts one
    process per GPU and each of them imports the package)."""
    import fcntl
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose: bool) -> str:

    def one(src):
        obj = os.path.join(OBJ, os.path.basename(src) + "." + _digest(src) + ".o")
        if not os.path.exists(obj):
            cmd = [HIPCC, *FLAGS, "-x", "hip", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(8) as ex:
        objs = list(ex.map(one, _sources()))
    stamp = os.path.join(OBJ, "link.stamp")
    key = " ".join(objs)
    if not (os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == key):
        tmp = LIB + f".tmp{os.getpid()}"
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs], check=True)
        os.replace(tmp, LIB)      # atomic: a concurrently loading process never sees a half-written library
        with open(stamp, "w") as f:
            f.write(key)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
