// What agent-scope (sc1) accesses cost against plain ones, by width, with every CU busy: the numbers behind the data
// layout of the streaming frame kernel.  hipcc --offload-arch=gfx950 -O3 sc1_bench.hip -o sc1_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

template <int MODE> __global__ void __launch_bounds__(256) k_store(double *p, long long n_words, int reps)
{
    // every wavefront writes 512-byte (8 B per lane) or 1024-byte (16 B per lane) pieces scattered over the buffer
    const long long lane = threadIdx.x & 63, wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long waves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (int r = 0; r < reps; ++r) {
        const long long piece = (wave + (long long)r * waves);
        if (MODE == 0) p[(piece * 64 + lane) % n_words] = (double)r;                                         // plain 8 B
        if (MODE == 1) __hip_atomic_store((unsigned long long *)p + (piece * 64 + lane) % n_words, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 2) ((v2d *)p)[(piece * 64 + lane) % (n_words / 2)] = (v2d){ (double)r, 1.0 };           // plain 16 B
        if (MODE == 3) {                                                                                      // sc1 16 B
            v2d *q = (v2d *)p + (piece * 64 + lane) % (n_words / 2);
            const v2d val = { (double)r, 1.0 };
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(q), "v"(val) : "memory");
        }
        if (MODE == 4) __hip_atomic_store((unsigned int *)p + (piece * 64 + lane) % (n_words * 2), (unsigned int)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // sc1 4 B
    }
}
template <int MODE> __global__ void __launch_bounds__(256) k_load(const double *p, long long n_words, int reps, double *sink)
{
    const long long lane = threadIdx.x & 63, wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long waves = ((long long)gridDim.x * blockDim.x) >> 6;
    double acc = 0;
    for (int r = 0; r < reps; ++r) {
        const long long piece = (wave * 7919 + (long long)r * waves);
        if (MODE == 0) acc += p[(piece * 64 + lane) % n_words];
        if (MODE == 1) acc += __longlong_as_double((long long)__hip_atomic_load((unsigned long long *)p + (piece * 64 + lane) % n_words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (MODE == 2) { const v2d v = ((const v2d *)p)[(piece * 64 + lane) % (n_words / 2)]; acc += v.x + v.y; }
        if (MODE == 3) {
            const v2d *q = (const v2d *)p + (piece * 64 + lane) % (n_words / 2);
            v2d v;
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(q) : "memory");
            acc += v.x + v.y;
        }
    }
    if (acc == 12345.678) sink[0] = acc;
}
// dependent round trips: one lane per wavefront chases a chain of sc1 loads while the others keep the memory busy
template <int MODE> __global__ void __launch_bounds__(256) k_latency(const int *chain, int steps, int *out)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int i = (int)(t & 0xfffff);
    for (int s = 0; s < steps; ++s)
        i = MODE ? __hip_atomic_load(const_cast<int *>(chain) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : chain[i];
    if (i == -7) out[0] = i;
}
__global__ void k_atomics(int *ctr, int stride_ints, int n_addr, int reps)
{
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int r = 0; r < reps; ++r) atomicAdd(ctr + (size_t)((wave + r) % n_addr) * stride_ints, 1);
}
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <typename F> static float timed(F f)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}
int main()
{
    const long long n_words = 1ll << 27;       // 1 GiB
    double *buf, *sink;
    CHECK(hipMalloc(&buf, n_words * 8));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 0, n_words * 8));
    const int reps = 64;
    const char *names[] = { "plain 8 B/lane", "sc1 8 B/lane", "plain 16 B/lane", "sc1 16 B/lane", "sc1 4 B/lane" };
    for (int blocks : { 1024, 3072 }) {
        printf("-- %d workgroups of 256\n", blocks);
        const double waves = blocks * 4.0;
        float ms;
        ms = timed([&] { hipLaunchKernelGGL(k_store<0>, dim3(blocks), dim3(256), 0, 0, buf, n_words, reps); });
        printf("store %-16s %8.1f GB/s\n", names[0], waves * reps * 512 / ms / 1e6);
        ms = timed([&] { hipLaunchKernelGGL(k_store<1>, dim3(blocks), dim3(256), 0, 0, buf, n_words, reps); });
        printf("store %-16s %8.1f GB/s\n", names[1], waves * reps * 512 / ms / 1e6);
        ms = timed([&] { hipLaunchKernelGGL(k_store<2>, dim3(blocks), dim3(256), 0, 0, buf, n_words, reps); });
        printf("store %-16s %8.1f GB/s\n", names[2], waves * reps * 1024 / ms / 1e6);
        ms = timed([&] { hipLaunchKernelGGL(k_store<3>, dim3(blocks), dim3(256), 0, 0, buf, n_words, reps); });
        printf("store %-16s %8.1f GB/s\n", names[3], waves * reps * 1024 / ms / 1e6);
        ms = timed([&] { hipLaunchKernelGGL(k_store<4>, dim3(blocks), dim3(256), 0, 0, buf, n_words, reps); });
        printf("store %-16s %8.1f GB/s\n", names[4], waves * reps * 256 / ms / 1e6);
        ms = timed([&] { hipLaunchKernelGGL(k_load<0>, dim3(blocks), dim3(256), 0, 0, buf, n_words, reps, sink); });
        printf("load  %-16s %8.1f GB/s\n", names[0], waves * reps * 512 / ms / 1e6);
        ms = timed([&] { hipLaunchKernelGGL(k_load<1>, dim3(blocks), dim3(256), 0, 0, buf, n_words, reps, sink); });
        printf("load  %-16s %8.1f GB/s\n", names[1], waves * reps * 512 / ms / 1e6);
        ms = timed([&] { hipLaunchKernelGGL(k_load<2>, dim3(blocks), dim3(256), 0, 0, buf, n_words, reps, sink); });
        printf("load  %-16s %8.1f GB/s\n", names[2], waves * reps * 1024 / ms / 1e6);
        ms = timed([&] { hipLaunchKernelGGL(k_load<3>, dim3(blocks), dim3(256), 0, 0, buf, n_words, reps, sink); });
        printf("load  %-16s %8.1f GB/s (waits for every load)\n", names[3], waves * reps * 1024 / ms / 1e6);
    }
    // latency of dependent loads, every lane of 768 workgroups chasing
    {
        std::vector<int> h(1 << 20);
        for (int i = 0; i < (1 << 20); ++i) h[i] = (int)(((long long)i * 7919 + 12345) & 0xfffff);
        int *chain, *out;
        CHECK(hipMalloc(&chain, h.size() * 4));
        CHECK(hipMalloc(&out, 64));
        CHECK(hipMemcpy(chain, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        for (int blocks : { 8, 768 }) {
            float ms = timed([&] { hipLaunchKernelGGL(k_latency<0>, dim3(blocks), dim3(256), 0, 0, chain, 256, out); });
            printf("dependent plain loads, %4d workgroups: %.2f us per step\n", blocks, ms * 1000 / 256);
            ms = timed([&] { hipLaunchKernelGGL(k_latency<1>, dim3(blocks), dim3(256), 0, 0, chain, 256, out); });
            printf("dependent sc1 loads,   %4d workgroups: %.2f us per step\n", blocks, ms * 1000 / 256);
        }
    }
    // non-returning atomics from 3072 wavefronts onto n addresses `stride` bytes apart
    {
        int *ctr;
        CHECK(hipMalloc(&ctr, 64 << 20));
        CHECK(hipMemset(ctr, 0, 64 << 20));
        for (int stride_b : { 64, 4352, 65536 + 256 })
            for (int n_addr : { 1, 8, 64 }) {
                float ms = timed([&] { hipLaunchKernelGGL(k_atomics, dim3(768), dim3(256), 0, 0, ctr, stride_b / 4, n_addr, 200); });
                printf("atomics onto %2d addresses %6d B apart: %.1f per us\n", n_addr, stride_b, 768 * 4 * 200 / (ms * 1000));
            }
    }
    return 0;
}
