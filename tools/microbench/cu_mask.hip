// CU masks and stream priorities on gfx950, measured (self-play beside training on one GPU).
//   hipcc --offload-arch=gfx950 -O3 -o cu_mask cu_mask.hip && ./cu_mask
// Part A: which physical CUs (XCC, SE, CU of HW_ID / XCC_ID) a stream made by hipExtStreamCreateWithCUMask lands on, per
//         mask bit range -- the layout a mask must have to take the same number of CUs out of every XCD.
// Part B: a chain of 31 small dependent kernels (the training step's shape: 128 blocks x 256 threads, a few us each) timed
//         alone and beside a chip-filling kernel of ~200 us blocks with 70 KB LDS each (the tower's shape), with the chain on
//         a normal / high-priority stream and the filler on all CUs or on a mask that leaves R CUs per XCD free.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_where(unsigned *out, int spin) {
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        out[blockIdx.x] = (hw & 0xffff) | ((xc & 0xf) << 16);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}
}

// filler: `spin` ticks of s_memrealtime (100 MHz) per block, dynamic LDS to set the residency
__global__ __launch_bounds__(256) void k_fill(float *sink, int spin) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = (float)blockIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz (s_memtime counts shader clocks here)
    float a = lds[(threadIdx.x + 1) & 255];
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) a = a * 1.0001f + 0.5f;
    if (a == 12345.678f) sink[0] = a;
}

__global__ __launch_bounds__(256) void k_link(float *buf, int n, int work) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float a = buf[i % n];
    for (int j = 0; j < work; ++j) a = a * 1.0001f + 0.25f;
    buf[i % n] = a;
}

static hipStream_t masked_stream(const std::vector<uint32_t> &mask, int prio = 0) {
    hipStream_t s;
    (void)prio;
    CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    return s;
}

static void part_a(int ncu) {
    const int words = (ncu + 31) / 32, nblk = 8192;
    unsigned *out;
    CK(hipMalloc(&out, nblk * sizeof(unsigned)));
    std::vector<unsigned> h(nblk);
    struct Case { const char *name; int lo, hi; };
    const Case cases[] = {{"bits 0..7", 0, 8}, {"bits 0..15", 0, 16}, {"bits 0..31", 0, 32}, {"bits 32..63", 32, 64},
                          {"bits 8..15", 8, 16}, {"bits 64..71", 64, 72}, {"all", 0, ncu}};
    for (const Case &c : cases) {
        std::vector<uint32_t> mask(words, 0);
        for (int b = c.lo; b < c.hi && b < ncu; ++b) mask[b / 32] |= 1u << (b % 32);
        hipStream_t s = masked_stream(mask);
        hipLaunchKernelGGL(k_where, dim3(nblk), dim3(64), 0, s, out, 2000);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), out, nblk * sizeof(unsigned), hipMemcpyDeviceToHost));
        std::map<int, std::set<int>> per_xcc;
        for (unsigned v : h) per_xcc[(v >> 16) & 0xf].insert(((v >> 13) & 7) * 100 + ((v >> 12) & 1) * 50 + ((v >> 8) & 0xf));
        printf("A %-12s:", c.name);
        int total = 0;
        for (auto &kv : per_xcc) {
            printf(" xcc%d[%zu:", kv.first, kv.second.size());
            int shown = 0;
            for (int id : kv.second) if (shown++ < 4) printf(" se%d.cu%d", id / 100, id % 50);
            printf("%s]", kv.second.size() > 4 ? " .." : "");
            total += (int)kv.second.size();
        }
        printf("  = %d CUs\n", total);
        CK(hipStreamDestroy(s));
    }
    CK(hipFree(out));
}

static float run_chain(hipStream_t t, float *buf, int links, int blocks, int work, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, t));
    for (int r = 0; r < reps; ++r)
        for (int l = 0; l < links; ++l) hipLaunchKernelGGL(k_link, dim3(blocks), dim3(256), 0, t, buf, blocks * 256, work);
    CK(hipEventRecord(e1, t));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / reps;
}

static void part_b(int ncu, int interleave_xcc) {
    float *sink, *buf;
    CK(hipMalloc(&sink, 1024)); CK(hipMalloc(&buf, 1 << 22));
    CK(hipMemset(buf, 0, 1 << 22));
    const int words = (ncu + 31) / 32;
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("B stream priority range: least %d greatest %d\n", lo, hi);
    hipStream_t tn, th;
    CK(hipStreamCreateWithPriority(&tn, hipStreamNonBlocking, lo));
    CK(hipStreamCreateWithPriority(&th, hipStreamNonBlocking, hi));
    const int links = 31, blocks = 128, work = 200, reps = 20;
    const int fill_blocks = 512 * 40, fill_spin = 20000, fill_lds = 70 * 1024;   // 40 rounds of 200 us blocks, two per CU
    CK(hipFuncSetAttribute((const void *)k_fill, hipFuncAttributeMaxDynamicSharedMemorySize, fill_lds));
    run_chain(tn, buf, links, blocks, work, 3);
    printf("B chain alone (31 links x 128 blocks): normal %.3f ms, high %.3f ms\n", run_chain(tn, buf, links, blocks, work, reps),
           run_chain(th, buf, links, blocks, work, reps));
    for (int reserve : {0, 1, 2, 4}) {       // CUs left free per XCD
        std::vector<uint32_t> mask(words, 0);
        int kept = 0;
        for (int b = 0; b < ncu; ++b) {
            // bit b -> (xcc, cu within xcc): interleaved (b % 8, b / 8) or blocked (b / 32, b % 32), from part A
            const int cu_in_xcc = interleave_xcc ? b / 8 : b % (ncu / 8);
            if (cu_in_xcc >= reserve) { mask[b / 32] |= 1u << (b % 32); ++kept; }
        }
        hipStream_t f = masked_stream(mask);
        hipEvent_t f0, f1;
        CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
        // filler alone
        CK(hipEventRecord(f0, f));
        hipLaunchKernelGGL(k_fill, dim3(fill_blocks), dim3(256), fill_lds, f, sink, fill_spin);
        CK(hipEventRecord(f1, f));
        CK(hipEventSynchronize(f1));
        float alone;
        CK(hipEventElapsedTime(&alone, f0, f1));
        for (int pr = 0; pr < 2; ++pr) {
            hipStream_t t = pr ? th : tn;
            CK(hipEventRecord(f0, f));
            for (int i = 0; i < 6; ++i) hipLaunchKernelGGL(k_fill, dim3(fill_blocks), dim3(256), fill_lds, f, sink, fill_spin);
            CK(hipEventRecord(f1, f));
            // chains while the filler runs (6 x ~8 ms): as many whole steps as fit in ~30 ms
            int steps = 0;
            float chain_ms = 0.f;
            while (hipEventQuery(f1) == hipErrorNotReady && steps < 400) { chain_ms += run_chain(t, buf, links, blocks, work, 1); ++steps; }
            CK(hipEventSynchronize(f1));
            float beside;
            CK(hipEventElapsedTime(&beside, f0, f1));
            printf("B reserve %d CU/XCD (filler on %d CUs): filler alone %.2f ms/launch, beside %.2f ms/launch; chain (%s prio) %.3f ms "
                   "per 31 links over %d chains\n", reserve, kept, alone, beside / 6, pr ? "high" : "normal", chain_ms / (steps ? steps : 1), steps);
        }
        CK(hipEventDestroy(f0)); CK(hipEventDestroy(f1));
        CK(hipStreamDestroy(f));
    }
}

int main(int argc, char **argv) {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs\n", p.gcnArchName, p.multiProcessorCount);
    part_a(p.multiProcessorCount);
    part_b(p.multiProcessorCount, argc > 1 ? atoi(argv[1]) : 1);
    if (argc <= 1) part_b(p.multiProcessorCount, 0);
    return 0;
}
