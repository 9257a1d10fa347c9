// Grid-barrier microbenchmark for gfx950 (profiles/r04_grid_barrier_ubench.txt): what does one barrier over all resident
// workgroups of a cooperative launch cost on the 8-XCD part, flat (one counter) and two-level (a counter per group of
// workgroups whose last arriver signs in at the top, everyone polls its group's release word)?  Every spin is bounded: a
// workgroup that waits longer than ~0.2 s raises `abort` and every workgroup leaves (the grid always drains).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_gridbarrier.hip -o tools/ubench_gridbarrier.bin && tools/ubench_gridbarrier.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// flat: counter counts arrivals of all rounds; round r is over when it reaches r * G
__device__ bool barrier_flat(unsigned* counter, unsigned target, unsigned* abort) {
    __shared__ int ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int good = 1;
        unsigned spins = 0;
        while (ld_agent(counter) < target) {
            if (++spins > (1u << 22) || ld_agent(abort)) { __hip_atomic_store(abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); good = 0; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

// two-level: groups of `gsize` workgroups (group = blockIdx % ngroups: the dispatcher deals workgroups round-robin over the
// XCDs, so with ngroups = 8 a group is one XCD's workgroups).  grp[g] counts the group's arrivals; its last arriver adds 1 to
// top; the workgroup that completes top publishes the round in every group's release word; everyone polls its own group's.
__device__ bool barrier_two(unsigned* grp, unsigned* top, unsigned* rel, int ngroups, unsigned round, unsigned* abort) {
    __shared__ int ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int g = blockIdx.x % ngroups;
        const unsigned gsize = (gridDim.x - g + ngroups - 1) / ngroups;
        int good = 1;
        const unsigned a = __hip_atomic_fetch_add(grp + 32 * g, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (a + 1 == round * gsize) {
            const unsigned t = __hip_atomic_fetch_add(top, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (t + 1 == round * (unsigned)ngroups)
                for (int k = 0; k < ngroups; ++k) __hip_atomic_store(rel + 32 * k, round, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned spins = 0;
        while (ld_agent(rel + 32 * g) < round) {
            if (++spins > (1u << 22) || ld_agent(abort)) { __hip_atomic_store(abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); good = 0; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

__global__ void __launch_bounds__(256) bench(unsigned* sync, float* data, int rounds, int mode, int ngroups, int dirty_floats) {
    unsigned* counter = sync;           // flat
    unsigned* top = sync + 32;
    unsigned* abort = sync + 64;
    unsigned* grp = sync + 128;
    unsigned* rel = sync + 128 + 32 * 64;
    float acc = 0.f;
    for (int r = 1; r <= rounds; ++r) {
        // some dirty data per workgroup and round (what a release has to write back)
        for (int i = threadIdx.x; i < dirty_floats; i += 256) data[(size_t)blockIdx.x * dirty_floats + i] = acc + (float)r;
        bool ok = true;
        if (mode == 0) ok = barrier_flat(counter, (unsigned)r * gridDim.x, abort);
        else if (mode == 1) ok = barrier_two(grp, top, rel, ngroups, (unsigned)r, abort);
        if (!ok) return;
        acc += data[(size_t)((blockIdx.x + 1) % gridDim.x) * dirty_floats + (threadIdx.x % (dirty_floats > 0 ? dirty_floats : 1))] * 1e-9f;
    }
    if (acc == 123.f) data[0] = acc;
}

int main() {
    int dev = 0;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    int per_cu = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bench, 256, 0));
    printf("device %s: %d CUs, cooperative launch %d, resident workgroups of 256 threads per CU %d\n", prop.name,
           prop.multiProcessorCount, prop.cooperativeLaunch, per_cu);
    unsigned* sync;
    float* data;
    const int maxg = 2048, dirty_max = 4096;
    CHECK(hipMalloc(&sync, 16384 * sizeof(unsigned)));
    CHECK(hipMalloc(&data, (size_t)maxg * dirty_max * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int grid : {256, 512, 1024, 2048}) {
        if (grid > per_cu * prop.multiProcessorCount) continue;
        for (int dirty : {0, 132 * 2, 4096}) {
            for (int mode = 0; mode < 2; ++mode) {
                for (int ngroups : {8, 32}) {
                    if (mode == 0 && ngroups != 8) continue;
                    int rounds = 2000;
                    CHECK(hipMemset(sync, 0, 16384 * sizeof(unsigned)));
                    void* args[] = {&sync, &data, &rounds, &mode, &ngroups, &dirty};
                    CHECK(hipEventRecord(e0));
                    CHECK(hipLaunchCooperativeKernel((const void*)bench, dim3(grid), dim3(256), args, 0, 0));
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    float ms = 0;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    unsigned ab = 0;
                    CHECK(hipMemcpy(&ab, sync + 64, 4, hipMemcpyDeviceToHost));
                    printf("grid %4d dirty %5d floats/wg  %-9s groups %2d : %7.2f us per barrier round%s\n", grid, dirty,
                           mode == 0 ? "flat" : "two-level", mode == 0 ? 1 : ngroups, ms * 1e3 / rounds, ab ? "  (ABORTED)" : "");
                }
            }
        }
    }
    return 0;
}
