// Microbenchmark: what a device-wide barrier costs inside one persistent kernel (256 blocks x 256 threads, one per CU),
// against a kernel boundary (~5-6 us per dependent launch, r4_kernel_chain_microbench.txt): (a) cooperative groups'
// grid.sync(), (b) a hand-rolled counter barrier with agent-scope release / acquire fences.  Between barriers every
// block writes 16 KB and reads 16 KB another block wrote (the barrier has real cross-XCD traffic to publish).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/grid_sync.hip -o tools/microbench/grid_sync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ __launch_bounds__(256) void k_cg(float4 *buf, int rounds, float *sink) {
    cg::grid_group grid = cg::this_grid();
    const int nb = gridDim.x;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        float4 *mine = buf + ((size_t)(r & 1) * nb + blockIdx.x) * 1024;
        for (int k = 0; k < 4; ++k) mine[threadIdx.x + 256 * k] = make_float4(r, k, blockIdx.x, threadIdx.x);
        grid.sync();
        const float4 *other = buf + ((size_t)(r & 1) * nb + (blockIdx.x + 37) % nb) * 1024;
        for (int k = 0; k < 4; ++k) acc += other[threadIdx.x + 256 * k].x;
    }
    if (acc == -1.f) sink[0] = acc;
}

__device__ __forceinline__ void hand_barrier(unsigned int *ctr, unsigned int target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE);          // agent scope by default: publishes this block's writes
        while (__atomic_load_n(ctr, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // the other threads' view
}

__global__ __launch_bounds__(256) void k_hand(float4 *buf, int rounds, float *sink, unsigned int *ctr, int *bad) {
    const int nb = gridDim.x;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        float4 *mine = buf + ((size_t)(r & 1) * nb + blockIdx.x) * 1024;
        for (int k = 0; k < 4; ++k) mine[threadIdx.x + 256 * k] = make_float4(r, k, blockIdx.x, threadIdx.x);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        hand_barrier(ctr, (unsigned int)(r + 1) * nb);
        const float4 *other = buf + ((size_t)(r & 1) * nb + (blockIdx.x + 37) % nb) * 1024;
        for (int k = 0; k < 4; ++k) {
            const float4 v = other[threadIdx.x + 256 * k];
            if (v.x != (float)r) atomicAdd(bad, 1);              // stale data = the barrier did not publish
            acc += v.x;
        }
    }
    if (acc == -1.f) sink[0] = acc;
}

int main() {
    const int nb = 256, rounds = 200;
    float4 *buf; float *sink; unsigned int *ctr; int *bad;
    hipMalloc(&buf, (size_t)2 * nb * 1024 * 16); hipMalloc(&sink, 4); hipMalloc(&ctr, 4); hipMalloc(&bad, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 2; ++variant) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(ctr, 0, 4); hipMemset(bad, 0, 4);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            int r = rounds;
            void *args_cg[] = {&buf, &r, &sink};
            void *args_h[] = {&buf, &r, &sink, &ctr, &bad};
            hipError_t err = variant == 0 ? hipLaunchCooperativeKernel((const void *)k_cg, dim3(nb), dim3(256), args_cg, 0, 0)
                                          : hipLaunchCooperativeKernel((const void *)k_hand, dim3(nb), dim3(256), args_h, 0, 0);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            int hbad = 0; hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
            printf("%-28s %s  %d rounds in %.3f ms = %.2f us per round (write 16 KB, barrier, read 16 KB)  stale reads %d\n",
                   variant == 0 ? "cooperative grid.sync()" : "counter barrier + fences", hipGetErrorString(err), rounds, ms,
                   1e3 * ms / rounds, hbad);
        }
    }
    return 0;
}
