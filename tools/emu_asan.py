#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- run CPU-tier tests with the kernel sources compiled under AddressSanitizer.

    python tools/emu_asan.py [pytest arguments]          (default: tests/test_kernels.py -m "not gpu" -x -q)
    python tools/emu_asan.py --ubsan [pytest arguments]  (UndefinedBehaviorSanitizer: static-array bounds, shifts, signed
                                                          overflow, float-to-int range; "runtime error:" lines on stderr)
    python tools/emu_asan.py --tsan [pytest arguments]   (ThreadSanitizer: unsynchronised accesses of two WORKGROUPS -- the emulator
                                                          runs workgroups on a pool of OS threads -- to the same global memory;
                                                          "WARNING: ThreadSanitizer" blocks on stderr)

The host emulator (tests/emu/) compiles the unmodified leco_amd/csrc/*.hip for the CPU; with -fsanitize=address every
global-memory access of a kernel is checked against the allocation of the tensor it belongs to (the interpreter runs
with the sanitizer runtime preloaded, so torch's CPU tensors carry red zones): a read or write past the end of an
operand -- silent on the GPU inside the caching allocator's blocks -- aborts with the kernel's source line.  LDS is a
static buffer and is not covered (LECO_EMU_LDS=poison covers never-written LDS).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))


def main():
    import build_emu
    argv = sys.argv[1:]
    if argv and argv[0] == "--ubsan":       # UndefinedBehaviorSanitizer instead: reports on stderr, the run continues
        build_emu.build(ubsan=True)
        args = argv[1:] or [os.path.join(ROOT, "tests", "test_kernels.py"), "-m", "not gpu", "-q"]
        env = dict(os.environ, LECO_EMU_UBSAN="1", UBSAN_OPTIONS=os.environ.get("UBSAN_OPTIONS", "print_stacktrace=0"))
        os.execve(sys.executable, [sys.executable, "-m", "pytest", "-p", "no:cacheprovider", "-s", *args], env)
    if argv and argv[0] == "--tsan":        # ThreadSanitizer: workgroup against workgroup (the pool's OS threads)
        if not os.path.exists(build_emu.TSAN_RT):
            sys.exit("emu_asan.py --tsan: this clang has no libclang_rt.tsan runtime to preload (build_emu.TSAN_RT is empty)")
        build_emu.build(tsan=True)
        args = argv[1:] or [os.path.join(ROOT, "tests", "test_kernels.py"), "-m", "not gpu", "-q"]
        # torch's OpenMP workers synchronise through an uninstrumented libgomp: every tensor they filled would be reported
        # against the kernel that reads it -- keep torch single-threaded under this sanitizer
        env = dict(os.environ, LECO_EMU_TSAN="1", LD_PRELOAD=build_emu.TSAN_RT, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1",
                   TSAN_OPTIONS=os.environ.get("TSAN_OPTIONS", "report_signal_unsafe=0:halt_on_error=0"))
        os.execve(sys.executable, [sys.executable, "-m", "pytest", "-p", "no:cacheprovider", "-s", *args], env)
    if not os.path.exists(build_emu.ASAN_RT):
        sys.exit("emu_asan.py: this clang has no libclang_rt.asan runtime to preload (build_emu.ASAN_RT is empty)")
    build_emu.build(asan=True)
    args = argv or [os.path.join(ROOT, "tests", "test_kernels.py"), "-m", "not gpu", "-x", "-q"]
    env = dict(os.environ, LECO_EMU_ASAN="1", LD_PRELOAD=build_emu.ASAN_RT,
               ASAN_OPTIONS=os.environ.get("ASAN_OPTIONS", "detect_leaks=0:detect_stack_use_after_return=0:"
                                                           "abort_on_error=1:allocator_may_return_null=1"))
    os.execve(sys.executable, [sys.executable, "-m", "pytest", "-p", "no:cacheprovider", *args], env)


if __name__ == "__main__":
    main()
