// Round 6 probe (verdict item 6): what does ONE hand-off between the workgroups of an N-split "cluster" cost on this pool's
// MI355X boxes?  The fused form of a C = 640 / 1280 transformer chain (to_out -> LayerNorm -> to_q) would split every Linear's
// columns over P workgroups per 64-row stripe and exchange the stripe's activations / LayerNorm row statistics through L2
// after each Linear: producer plain stores -> __syncthreads -> lane-0 agent release -> relaxed agent flag; consumer one relaxed
// poll loop -> agent acquire -> __syncthreads -> plain loads (MI355X_MICROARCH.md "inter-workgroup visibility").
// Here: G workgroups (one per CU) in clusters of P; every iteration each workgroup publishes PAYLOAD bytes and reads its P - 1
// partners' -- checked word by word -- T times inside one launch; reported: us per hand-off, against the same loop without
// the exchange.  Every spin is bounded; a timeout is counted and reported, never waited out.
//   hipcc --offload-arch=gfx950 -O3 -o cluster_probe cluster_probe.hip && ./cluster_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned word(int wg, int t, int i) { return (unsigned)(wg * 2654435761u) ^ (unsigned)(t * 40503u) ^ (unsigned)i; }

template <bool EXCHANGE>
__global__ __launch_bounds__(256) void probe(unsigned* payload, unsigned* flags, int P, int words, int T, unsigned* errors,
                                             unsigned* timeouts, unsigned* sink) {
    const int wg = (int)blockIdx.x, tid = (int)threadIdx.x;
    const int c0 = wg / P * P;                         // first workgroup of my cluster
    unsigned acc = 0;
    __shared__ int abort_flag;
    if (tid == 0) abort_flag = 0;
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        unsigned* mine = payload + ((size_t)(t & 1) * gridDim.x + wg) * words;
        for (int i = tid; i < words; i += 256) mine[i] = word(wg, t, i);
        if (EXCHANGE) {
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(flags + wg, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int p = 0; p < P; ++p) {
                    const int other = c0 + p;
                    if (other == wg) continue;
                    int spins = 0;
                    while (__hip_atomic_load(flags + other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(t + 1)) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1 << 20)) { atomicAdd(timeouts, 1u); abort_flag = 1; break; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            if (abort_flag) break;
            for (int p = 0; p < P; ++p) {
                const int other = c0 + p;
                if (other == wg) continue;
                const unsigned* theirs = payload + ((size_t)(t & 1) * gridDim.x + other) * words;
                for (int i = tid; i < words; i += 256) {
                    const unsigned v = theirs[i];
                    if (v != word(other, t, i)) atomicAdd(errors, 1u);
                    acc += v;
                }
            }
        } else {
            __syncthreads();
            for (int i = tid; i < words; i += 256) acc += mine[i];
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int G = prop.multiProcessorCount;            // one workgroup per CU: all co-resident
    const int T = 200;
    unsigned *payload, *flags, *errors, *timeouts, *sink;
    const int max_words = 16384 / 4;
    CHECK(hipMalloc(&payload, (size_t)2 * G * max_words * 4));
    CHECK(hipMalloc(&flags, G * 4));
    CHECK(hipMalloc(&errors, 4)); CHECK(hipMalloc(&timeouts, 4)); CHECK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("# %s, %d CUs, %d workgroups of 256 threads (one per CU), %d iterations per launch\n", prop.name, G, G, T);
    printf("# P = workgroups per cluster, payload = bytes each workgroup publishes (and reads from each partner) per hand-off\n");
    for (int P : {2, 4, 8}) {
        for (int bytes : {1024, 4096, 16384}) {
            const int words = bytes / 4;
            float ms[2] = {0.f, 0.f};
            unsigned herr = 0, hto = 0;
            for (int ex = 0; ex < 2; ++ex) {
                float best = 1e30f;
                for (int rep = 0; rep < 5; ++rep) {
                    CHECK(hipMemset(flags, 0, G * 4)); CHECK(hipMemset(errors, 0, 4)); CHECK(hipMemset(timeouts, 0, 4));
                    CHECK(hipDeviceSynchronize());
                    CHECK(hipEventRecord(e0));
                    if (ex) hipLaunchKernelGGL(probe<true>, dim3(G), dim3(256), 0, 0, payload, flags, P, words, T, errors, timeouts, sink);
                    else hipLaunchKernelGGL(probe<false>, dim3(G), dim3(256), 0, 0, payload, flags, P, words, T, errors, timeouts, sink);
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    float t;
                    CHECK(hipEventElapsedTime(&t, e0, e1));
                    if (t < best) best = t;
                    if (ex) {
                        unsigned a, b;
                        CHECK(hipMemcpy(&a, errors, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&b, timeouts, 4, hipMemcpyDeviceToHost));
                        herr += a; hto += b;
                    }
                }
                ms[ex] = best;
            }
            printf("P=%d payload=%5d B: %7.2f us per iteration with the hand-off, %6.2f us without -> %6.2f us per hand-off   "
                   "(wrong words %u, timeouts %u over 5 launches)\n", P, bytes, ms[1] * 1e3f / T, ms[0] * 1e3f / T, (ms[1] - ms[0]) * 1e3f / T,
                   herr, hto);
        }
    }
    return 0;
}
