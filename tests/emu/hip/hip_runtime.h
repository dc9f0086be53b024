// TEST INFRASTRUCTURE ONLY -- host emulation of the small HIP subset the leco_amd kernels
// use, so that the *unmodified* kernel sources under leco_amd/csrc/ can be executed on a
// CPU (no GPU in the build container) and checked against the oracle at tiny shapes.
//
// Never shipped, never loaded by the product path: leco_amd/hip.py only ever dlopens
// libleco_hip.so (the gfx950 build).  tests/emu/build_emu.py compiles the kernel sources
// with host clang++ and `-I tests/emu`, which makes `#include <hip/hip_runtime.h>` and
// `#include <leco_prims.h>` resolve to this directory instead of ROCm / csrc/prims.
//
// Execution model: one workgroup at a time per OS thread; its work-items are fibers
// (a register-only context switch on x86-64, ucontext elsewhere: emu_runtime.cpp) scheduled round-robin.  __syncthreads() and the wave64 collectives (MFMA,
// shuffles) are rendezvous points; the last arriver of a wave collective performs it for
// the whole wave.  Lane<->matrix-element maps follow cdna_hip_programming.md section 3.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <algorithm>
using std::min;
using std::max;
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 { unsigned x, y, z; };

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
template <class F> static inline hipError_t hipFuncSetAttribute(F, int, int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }

namespace emu {
extern thread_local emu_uint3 t_idx, b_idx;
extern thread_local dim3 b_dim, g_dim;
void sync_block();
// wave collective: every lane of the calling wave deposits `bytes` from `in`; returns a
// pointer to the wave's 64 x slot table (slot stride `emu::kSlot` bytes) valid until the
// lane's next collective.
constexpr int kSlot = 64;
const unsigned char* wave_gather(const void* in, int bytes);
int lane();
unsigned long block_serial();      // number of the workgroup this OS thread is running (1, 2, ...)
bool lds_poison();                 // LECO_EMU_LDS=poison
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
}  // namespace emu

#define threadIdx (emu::t_idx)
#define blockIdx (emu::b_idx)
#define blockDim (emu::b_dim)
#define gridDim (emu::g_dim)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __syncthreads() emu::sync_block()
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), [=]() { kern(__VA_ARGS__); })

#define __expf(x) expf(x)
#define __logf(x) logf(x)
static inline unsigned atomicAdd(unsigned* addr, unsigned v) { return __atomic_fetch_add(addr, v, __ATOMIC_SEQ_CST); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline float atomicAdd(float* addr, float v) {
    unsigned* p = reinterpret_cast<unsigned*>(addr);
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    for (;;) {
        float f;
        memcpy(&f, &old, 4);
        f += v;
        unsigned nw;
        memcpy(&nw, &f, 4);
        if (__atomic_compare_exchange_n(p, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
            float r;
            memcpy(&r, &old, 4);
            return r;
        }
    }
}
