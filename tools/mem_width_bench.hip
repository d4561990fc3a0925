// Dev tool: what a wave's access width costs on gfx950.  Streams N bytes (write only / read only) with 2-, 4-, 8- and 16-byte
// accesses per lane, 256-thread workgroups, every wave on its own contiguous run of lines; prints GB/s.
//   hipcc --offload-arch=gfx950 -O3 -o mem_width_bench tools/mem_width_bench.hip && ./mem_width_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <typename T> __global__ __launch_bounds__(256) void wr(T* p, size_t n_per_wg, T v)
{
    T* q = p + (size_t)blockIdx.x * n_per_wg;
    for (size_t i = threadIdx.x; i < n_per_wg; i += 256) q[i] = v;
}
template <typename T> __global__ __launch_bounds__(256) void rd(const T* p, size_t n_per_wg, uint32_t* sink)
{
    const T* q = p + (size_t)blockIdx.x * n_per_wg;
    uint32_t acc = 0;
    for (size_t i = threadIdx.x; i < n_per_wg; i += 256) { T v = q[i]; acc += *reinterpret_cast<const uint8_t*>(&v); }
    if (acc == 0x12345678u) *sink = acc;
}
// 4 row streams per workgroup (like the DWT's four sub-bands): T stores to four distant rows per iteration
template <typename T> __global__ __launch_bounds__(256) void wr4(T* p, size_t n_per_wg, size_t quarter, T v)
{
    T* q = p + (size_t)blockIdx.x * n_per_wg;
    for (size_t i = threadIdx.x; i < n_per_wg; i += 256) { q[i] = v; q[i + quarter] = v; q[i + 2 * quarter] = v; q[i + 3 * quarter] = v; }
}
// the DWT's mix: for every unit read, two written (8-bit pixels in, int16 coefficients out), to two other arrays
template <typename T> __global__ __launch_bounds__(256) void rw12(const T* in, T* o1, T* o2, size_t n_per_wg)
{
    const size_t b = (size_t)blockIdx.x * n_per_wg;
    for (size_t i = threadIdx.x; i < n_per_wg; i += 256) { const T v = in[b + i]; o1[b + i] = v; o2[b + i] = v; }
}
// ... one read, one write (levels >= 1)
template <typename T> __global__ __launch_bounds__(256) void rw11(const T* in, T* o1, size_t n_per_wg)
{
    const size_t b = (size_t)blockIdx.x * n_per_wg;
    for (size_t i = threadIdx.x; i < n_per_wg; i += 256) o1[b + i] = in[b + i];
}
template <typename F> float time_ms(F f)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main()
{
    const size_t bytes = 512u << 20;
    void* buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
    uint32_t* sink; hipMalloc(&sink, 4);
    for (int wgs : {4096, 16384}) {
        const size_t per = bytes / wgs;
#define RUN(T, name) { T v{}; \
        float w = time_ms([&] { hipLaunchKernelGGL(wr<T>, dim3(wgs), dim3(256), 0, 0, (T*)buf, per / sizeof(T), v); }); \
        float r = time_ms([&] { hipLaunchKernelGGL(rd<T>, dim3(wgs), dim3(256), 0, 0, (const T*)buf, per / sizeof(T), sink); }); \
        float w4 = time_ms([&] { hipLaunchKernelGGL(wr4<T>, dim3(wgs), dim3(256), 0, 0, (T*)buf, per / sizeof(T) / 4, bytes / sizeof(T) / 4 / 1, v); }); \
        printf("wgs %5d  %-6s write %7.0f GB/s   read %7.0f GB/s   write4rows %7.0f GB/s\n", wgs, name, bytes / w / 1e6, bytes / r / 1e6, bytes / w4 / 1e6); }
        RUN(uint16_t, "b16") RUN(uint32_t, "b32") RUN(uint2, "b64") RUN(uint4, "b128")
    }
    {
        const size_t third = 200u << 20;          // 200 MB read + 400 MB written, like level 0 of the 8K frame
        char *i0, *o1, *o2; hipMalloc(&i0, third); hipMalloc(&o1, third); hipMalloc(&o2, third); hipMemset(i0, 1, third);
        for (int wgs : {2048, 8192}) {
            float m = time_ms([&] { hipLaunchKernelGGL(rw12<uint32_t>, dim3(wgs), dim3(256), 0, 0, (const uint32_t*)i0, (uint32_t*)o1, (uint32_t*)o2, third / 4 / wgs); });
            float m4 = time_ms([&] { hipLaunchKernelGGL(rw12<uint4>, dim3(wgs), dim3(256), 0, 0, (const uint4*)i0, (uint4*)o1, (uint4*)o2, third / 16 / wgs); });
            float c = time_ms([&] { hipLaunchKernelGGL(rw11<uint32_t>, dim3(wgs), dim3(256), 0, 0, (const uint32_t*)i0, (uint32_t*)o1, third / 4 / wgs); });
            float c4 = time_ms([&] { hipLaunchKernelGGL(rw11<uint4>, dim3(wgs), dim3(256), 0, 0, (const uint4*)i0, (uint4*)o1, third / 16 / wgs); });
            printf("wgs %5d  1 read : 2 writes  b32 %7.0f GB/s  b128 %7.0f GB/s     1 : 1  b32 %7.0f GB/s  b128 %7.0f GB/s   (read + written bytes)\n",
                   wgs, 3.0 * third / m / 1e6, 3.0 * third / m4 / 1e6, 2.0 * third / c / 1e6, 2.0 * third / c4 / 1e6);
        }
    }
    return 0;
}
