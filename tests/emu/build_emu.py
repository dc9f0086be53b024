"""TEST INFRASTRUCTURE ONLY -- builds tests/emu/_build/libleco_emu.so: the *same* kernel
sources as libleco_hip.so (leco_amd/csrc/*.hip), compiled for the host with ROCm's clang++
against the fiber emulator in this directory, so kernel logic can be checked on a CPU."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "leco_amd", "csrc")
OUT = os.path.join(HERE, "_build")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def sources():
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))
            if f.endswith(".hip") and f != "runtime.hip"]
    srcs += [os.path.join(CSRC, "common.cpp"), os.path.join(HERE, "emu_runtime.cpp")]
    return srcs


def _digest(path):
    h = hashlib.sha1()
    deps = [path, os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(HERE, "leco_prims.h"),
            os.path.join(ROOT, "include", "leco_hip.h"), os.path.join(CSRC, "common.h")]
    for d in deps:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _runtime(name: str) -> str:
    """Path of a sanitizer runtime of THIS clang (whatever its version directory is called)."""
    try:
        out = subprocess.run([CLANG, f"-print-file-name=libclang_rt.{name}-x86_64.so"], capture_output=True, text=True,
                             check=True).stdout.strip()
        if os.path.isabs(out) and os.path.exists(out):
            return out
        out = subprocess.run([CLANG, f"-print-file-name=libclang_rt.{name}.so"], capture_output=True, text=True, check=True).stdout.strip()
        return out if os.path.isabs(out) and os.path.exists(out) else ""
    except (OSError, subprocess.CalledProcessError):
        return ""


ASAN_RT = _runtime("asan")
TSAN_RT = _runtime("tsan")


def build(verbose=False, asan=None, ubsan=None, tsan=None):
    """asan=True (or LECO_EMU_ASAN=1): the AddressSanitizer build, libleco_emu_asan.so -- the kernels' global-memory
    accesses are checked against the tensors' allocations.  The interpreter has to run with LD_PRELOAD=ASAN_RT
    (tools/emu_asan.py does that).  ubsan=True (or LECO_EMU_UBSAN=1): the UndefinedBehaviorSanitizer build,
    libleco_emu_ubsan.so (static-array bounds, shifts, signed overflow, float-to-int range; reports go to stderr and the
    run continues; no preload needed) -- tools/emu_asan.py --ubsan."""
    if asan is None:
        asan = os.environ.get("LECO_EMU_ASAN") == "1"
    if ubsan is None:
        ubsan = os.environ.get("LECO_EMU_UBSAN") == "1" and not asan
    if tsan is None:
        tsan = os.environ.get("LECO_EMU_TSAN") == "1" and not asan and not ubsan
    os.makedirs(OUT, exist_ok=True)
    flags = ["-std=c++17", "-O2", "-fPIC", "-pthread", "-Wno-unknown-attributes", "-Wno-unused-value",
             "-Wno-unknown-pragmas", "-Wno-pass-failed",
             "-I", HERE, "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    tag, link = "", []
    if tsan:
        link = ["-fsanitize=thread", "-shared-libsan"]
        flags += link + ["-fno-omit-frame-pointer", "-g1"]
        tag = ".tsan"
    elif asan:
        link = ["-fsanitize=address", "-shared-libasan"]
        flags += link + ["-fno-omit-frame-pointer", "-g1"]
        tag = ".asan"
    elif ubsan:
        # alignment: the kernels' vector loads of 2-byte-aligned bf16 rows are legal on the GPU (and memcpy-like on x86)
        link = ["-fsanitize=undefined,bounds,float-cast-overflow", "-fno-sanitize=alignment,vptr,function", "-shared-libsan"]
        flags += link + ["-fno-omit-frame-pointer", "-g1"]
        tag = ".ubsan"

    def one(src):
        obj = os.path.join(OUT, os.path.basename(src) + "." + _digest(src) + tag + ".o")
        if not os.path.exists(obj):
            cmd = [CLANG, "-x", "c++", *flags, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(8) as ex:
        objs = list(ex.map(one, sources()))
    lib = os.path.join(OUT, "libleco_emu%s.so" % tag.replace(".", "_"))
    stamp = os.path.join(OUT, "link" + tag + ".stamp")
    key = " ".join(objs)
    if not os.path.exists(lib) or not os.path.exists(stamp) or open(stamp).read() != key:
        rpath = ["-Wl,-rpath," + os.path.dirname(ASAN_RT)] if ubsan else []
        subprocess.run([CLANG, "-shared", "-pthread", *link, *rpath, "-o", lib, *objs], check=True)
        with open(stamp, "w") as f:
            f.write(key)
    return lib


def build_selftest(tsan=False):
    """tests/emu/selftest_sched.cpp + the fiber runtime as a stand-alone program (the work-item schedules' own test);
    tsan=True: under ThreadSanitizer (the cross-workgroup race probe)."""
    os.makedirs(OUT, exist_ok=True)
    srcs = [os.path.join(HERE, "selftest_sched.cpp"), os.path.join(HERE, "emu_runtime.cpp")]
    exe = os.path.join(OUT, "selftest_sched." + hashlib.sha1("".join(_digest(x) for x in srcs).encode()).hexdigest()[:16]
                       + (".tsan" if tsan else ""))
    if not os.path.exists(exe):
        san = ["-fsanitize=thread", "-fno-omit-frame-pointer", "-g1"] if tsan else []
        subprocess.run([CLANG, "-x", "c++", "-std=c++17", "-O2", "-pthread", "-Wno-unknown-attributes", "-Wno-unused-value", *san,
                        "-I", HERE, "-I", os.path.join(ROOT, "include"), "-I", CSRC, *srcs, "-o", exe], check=True)
    return exe


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
