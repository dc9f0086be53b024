// TEST INFRASTRUCTURE ONLY -- self-test of the emulator's work-item schedules (LECO_EMU_SCHED, emu_runtime.cpp).
// Built and run by tests/test_host.py::test_emulator_schedules_*.  Two kernels, four waves each:
//   order_kernel   records the order in which the waves pass three wave rendezvous: round-robin interleaves them
//                  (0 1 2 3 0 1 2 3 ...), greedy lets a wave run to its end first (0 0 0 1 1 1 ...), the reversed
//                  schedules mirror that;
//   reuse_kernel   reuses one LDS slot per wave across iterations and, with `fenced == 0`, leaves out the barrier
//                  between the last read of an iteration and the first write of the next one: a real race on the
//                  hardware.  Near-lockstep round-robin never sees it; the greedy schedules must.
//   lds_probe_kernel reads dynamic LDS it never wrote: 0x7fc0 (the NaN fill) under LECO_EMU_LDS=poison, otherwise whatever
//                  the previous workgroup on this OS thread left (0 / 1 here).
//   block_order_kernel records the workgroup dispatch order (LECO_EMU_BLOCKS, one OS thread).
// Prints "order: ...", "reuse fenced=F: ok|RACE", "lds: ..." and "blocks: ..." lines.
#include <hip/hip_runtime.h>
#include <leco_prims.h>

#include <cstdio>

using namespace leco;

__global__ void order_kernel(unsigned* cnt, int* order) {
    const int wave = threadIdx.x >> 6;
    for (int it = 0; it < 3; ++it) {
        float v = leco::shfl_xor((float)threadIdx.x, 1);          // a wave rendezvous
        if ((threadIdx.x & 63) == 0 && v >= 0.f) order[atomicAdd(cnt, 1u)] = wave;
    }
}

__global__ void reuse_kernel(int* out, int fenced) {
    __shared__ float slot[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int it = 0; it < 4; ++it) {
        float mine = leco::shfl_xor((float)(100 * it + wave), 1);  // every lane of the wave holds 100 it + wave
        if (lane == 0) slot[wave] = mine;
        __syncthreads();
        float nb = slot[(wave + 1) & 3];                          // the neighbour wave's value of THIS iteration
        acc += leco::shfl_xor(nb, 1);
        if (fenced) __syncthreads();                               // without it the neighbour may already hold it + 1
    }
    if (lane == 0) out[wave] = (int)acc;
}

__global__ void lds_probe_kernel(unsigned* out) {
    const unsigned short* lds = (const unsigned short*)dyn_lds();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[blockIdx.x + 7];     // never written
    __syncthreads();
    if (threadIdx.x == 0) ((unsigned short*)dyn_lds())[blockIdx.x + 8] = 1;   // what the NEXT workgroup would find
}

__global__ void block_order_kernel(unsigned* cnt, int* order) {
    if (threadIdx.x == 0) order[atomicAdd(cnt, 1u)] = (int)blockIdx.x;
}

// Sixty-four workgroups add 1 to the same word 20 000 times each: with `atomic` through atomicAdd, otherwise with a plain read-modify-write -- a
// race between workgroups, which only the ThreadSanitizer build (build_selftest(tsan=True)) can see.
__global__ void cross_block_kernel(unsigned* word, int atomic) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < 20000; ++i) {       // long enough that the pool's threads are certainly inside it together
        if (atomic) atomicAdd(word, 1u);
        else *(volatile unsigned*)word = *(volatile unsigned*)word + 1;
    }
}

// Waves 2 and 3 return before the second barrier (the way a loader wave leaves a kernel before its epilogue): on the hardware
// a terminated wave no longer counts at s_barrier, so waves 0 and 1 must get through -- under every schedule.
__global__ void early_exit_kernel(int* out) {
    __shared__ int slot[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) slot[wave] = 10 + wave;
    __syncthreads();
    if (wave >= 2) return;
    const int got = slot[wave + 2];
    __syncthreads();                                               // two waves only
    if (lane == 0) slot[wave] = got;
    __syncthreads();
    if (lane == 0) out[wave] = slot[wave ^ 1];
}

int main(int argc, char** argv) {
    if (argc > 2 && !strcmp(argv[1], "cross")) {
        unsigned word = 0;
        unsigned* pw = &word;
        const int atomic = atoi(argv[2]);
        hipLaunchKernelGGL(cross_block_kernel, dim3(64), dim3(64), 0, 0, pw, atomic);
        printf("cross atomic=%d: %u\n", atomic, word);
        return 0;
    }
    unsigned cnt = 0;
    int order[12];
    for (int& v : order) v = -1;
    unsigned* pc = &cnt;
    int* po = order;
    hipLaunchKernelGGL(order_kernel, dim3(1), dim3(256), 0, 0, pc, po);
    printf("order:");
    for (int v : order) printf(" %d", v);
    printf("\n");
    for (int fenced = 1; fenced >= 0; --fenced) {
        int out[4] = {0, 0, 0, 0};
        int* pout = out;
        hipLaunchKernelGGL(reuse_kernel, dim3(1), dim3(256), 0, 0, pout, fenced);
        bool ok = true;
        for (int w = 0; w < 4; ++w) ok = ok && out[w] == 600 + 4 * ((w + 1) & 3);      // sum over it of 100 it + (w + 1) % 4
        printf("reuse fenced=%d: %s\n", fenced, ok ? "ok" : "RACE");
    }
    unsigned probe[4] = {0, 0, 0, 0};
    unsigned* pp = probe;
    hipLaunchKernelGGL(lds_probe_kernel, dim3(4), dim3(64), 0, 0, pp);
    printf("lds: %x %x %x %x\n", probe[0], probe[1], probe[2], probe[3]);
    int early[2] = {0, 0};
    int* pe = early;
    hipLaunchKernelGGL(early_exit_kernel, dim3(1), dim3(256), 0, 0, pe);
    printf("early: %d %d\n", early[0], early[1]);
    unsigned bcnt = 0;
    int border[6] = {-1, -1, -1, -1, -1, -1};
    unsigned* pbc = &bcnt;
    int* pbo = border;
    hipLaunchKernelGGL(block_order_kernel, dim3(6), dim3(64), 0, 0, pbc, pbo);
    printf("blocks:");
    for (int v : border) printf(" %d", v);
    printf("\n");
    return 0;
}
