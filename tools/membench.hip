// membench.hip — write/read/copy ceilings on this MI355X for the access shapes the GT kernels use.
// Tuning aid only (not part of the library): establishes the known-good reference a kernel's
// achieved GB/s is compared with (cdna_hip_programming.md §5.4 rule 10).
//   hipcc -O3 --offload-arch=gfx950 -o tools/membench tools/membench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void fill16(v4u *dst, uint64_t n_chunks, uint32_t seed)
{
    const uint64_t step = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n_chunks; i += step) {
        v4u v = {seed + (uint32_t)i, seed ^ (uint32_t)i, (uint32_t)i * 3u, 0x302F3009u};
        if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
    }
}

// constant data, same loop as fill16 (is the memset rate a data effect or a loop-shape effect?)
__global__ __launch_bounds__(256) void fill16_const(v4u *dst, uint64_t n_chunks, uint32_t seed)
{
    const uint64_t step = (uint64_t)gridDim.x * 256u;
    const v4u v = {seed, seed, seed, seed};
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n_chunks; i += step) dst[i] = v;
}

// text-like data: every dword one of 4 values (what the GT stream looks like)
__global__ __launch_bounds__(256) void fill16_text(v4u *dst, uint64_t n_chunks, uint32_t seed)
{
    const uint64_t step = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n_chunks; i += step) {
        uint32_t h = (uint32_t)i * 2654435761u + seed;
        v4u v = {0x302F3009u + ((h >> 3) & 1) * 0x01000000u, 0x302F3009u + ((h >> 7) & 1) * 0x01000100u,
                 0x302F3009u + ((h >> 11) & 1) * 0x01000000u, 0x302F3009u + ((h >> 17) & 1) * 0x01000100u};
        dst[i] = v;
    }
}

// tile-blocked like gt_flat: block handles 16 KiB contiguous per step
template <bool NT, int U>
__global__ __launch_bounds__(256) void fill16_tiled(v4u *dst, uint64_t n_chunks, uint32_t seed)
{
    const uint64_t tiles = (n_chunks + 256u * U - 1) / (256u * U);
    for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            uint64_t i = t * 256u * U + u * 256u + threadIdx.x;
            if (i < n_chunks) {
                v4u v = {seed + (uint32_t)i, seed ^ (uint32_t)i, (uint32_t)i * 3u, 0x302F3009u};
                if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void read16(const v4u *src, uint64_t n_chunks, uint32_t *sink)
{
    const uint64_t step = (uint64_t)gridDim.x * 256u;
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n_chunks; i += step) {
        v4u v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) *sink = acc;
}

__global__ __launch_bounds__(256) void copy16(const v4u *src, v4u *dst, uint64_t n_chunks)
{
    const uint64_t step = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n_chunks; i += step) dst[i] = src[i];
}

// 1 read : 16 write, like the all-samples GT path
__global__ __launch_bounds__(256) void expand16(const uint8_t *src, v4u *dst, uint64_t n_chunks)
{
    const uint64_t step = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n_chunks; i += step) {
        uint32_t b = src[i];
        v4u v = {b * 0x01010101u, b + 1u, b + 2u, b + 3u};
        dst[i] = v;
    }
}

// software-pipelined expand: D loads in flight ahead of the stores
template <int D>
__global__ __launch_bounds__(256) void expand16_pipe(const uint8_t *src, v4u *dst, uint64_t n_chunks)
{
    const uint64_t step = (uint64_t)gridDim.x * 256u;
    uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    uint32_t buf[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        uint64_t j = i + d * step;
        buf[d] = j < n_chunks ? src[j] : 0u;
    }
    for (; i < n_chunks; i += D * step) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            uint64_t j = i + d * step;
            uint32_t b = buf[d];
            uint64_t jn = j + D * step;
            buf[d] = jn < n_chunks ? src[jn] : 0u;
            if (j < n_chunks) {
                v4u v = {b * 0x01010101u, b + 1u, b + 2u, b + 3u};
                dst[j] = v;
            }
        }
    }
}

// expand with the byte index wrapped into a small window (reads served by L2 / Infinity Cache)
__global__ __launch_bounds__(256) void expand16_cached(const uint8_t *src, v4u *dst, uint64_t n_chunks, uint64_t src_mask)
{
    const uint64_t step = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n_chunks; i += step) {
        uint32_t b = src[i & src_mask];
        v4u v = {b * 0x01010101u, b + 1u, b + 2u, b + 3u};
        dst[i] = v;
    }
}

// expand with wide reads: a wave loads 1 KiB (16 B/lane) = the input of 16 KiB of output, bounces it
// through LDS, then issues 16 coalesced 1-KiB stores.  Block = 4 waves = 64 KiB of output per step.
__global__ __launch_bounds__(256) void expand16_wide(const uint8_t *src, v4u *dst, uint64_t n_chunks)
{
    __shared__ uint8_t lds[4][1024];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t tiles = (n_chunks + 4095) / 4096;  // 4096 chunks (64 KiB out, 4 KiB in) per block step
    for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const uint64_t base = t * 4096 + wave * 1024;  // this wave's first chunk == first input byte
        if (base + lane * 16 + 16 <= n_chunks) {
            v4u in = *reinterpret_cast<const v4u *>(src + base + lane * 16);
            *reinterpret_cast<v4u *>(&lds[wave][lane * 16]) = in;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const uint64_t i = base + u * 64 + lane;
            if (i < n_chunks) {
                uint32_t b = lds[wave][u * 64 + lane];
                v4u v = {b * 0x01010101u, b + 1u, b + 2u, b + 3u};
                dst[i] = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename F>
static double time_ms(F launch, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main(int argc, char **argv)
{
    const uint64_t bytes = (argc > 1 ? strtoull(argv[1], 0, 10) : 4096ull) << 20;  // MiB
    const uint64_t n = bytes / 16;
    v4u *d, *s;
    uint32_t *sink;
    CK(hipMalloc(&d, bytes));
    CK(hipMalloc(&s, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(s, 1, bytes));
    const int reps = 9;
    printf("buffer %.1f MiB\n", bytes / 1048576.0);
    for (int grid : {256, 512, 1024, 2048, 4096}) {
        double t;
        t = time_ms([&] { hipLaunchKernelGGL(fill16<false>, dim3(grid), dim3(256), 0, 0, d, n, 7u); }, reps);
        printf("fill16 plain      grid %5d: %.3f ms  %.2f TB/s\n", grid, t, bytes / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(fill16_const, dim3(grid), dim3(256), 0, 0, d, n, 0x5a5a5a5au); }, reps);
        printf("fill16 const      grid %5d: %.3f ms  %.2f TB/s\n", grid, t, bytes / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(fill16_text, dim3(grid), dim3(256), 0, 0, d, n, 7u); }, reps);
        printf("fill16 text-like  grid %5d: %.3f ms  %.2f TB/s\n", grid, t, bytes / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(fill16<true>, dim3(grid), dim3(256), 0, 0, d, n, 7u); }, reps);
        printf("fill16 nt         grid %5d: %.3f ms  %.2f TB/s\n", grid, t, bytes / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL((fill16_tiled<false, 4>), dim3(grid), dim3(256), 0, 0, d, n, 7u); }, reps);
        printf("fill16 tiled U=4  grid %5d: %.3f ms  %.2f TB/s\n", grid, t, bytes / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL((fill16_tiled<true, 4>), dim3(grid), dim3(256), 0, 0, d, n, 7u); }, reps);
        printf("fill16 tiled nt   grid %5d: %.3f ms  %.2f TB/s\n", grid, t, bytes / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(read16, dim3(grid), dim3(256), 0, 0, s, n, sink); }, reps);
        printf("read16            grid %5d: %.3f ms  %.2f TB/s\n", grid, t, bytes / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, s, d, n); }, reps);
        printf("copy16            grid %5d: %.3f ms  %.2f TB/s (read+write)\n", grid, t, 2.0 * bytes / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(expand16, dim3(grid), dim3(256), 0, 0, (const uint8_t *)s, d, n); }, reps);
        printf("expand 1:16       grid %5d: %.3f ms  %.2f TB/s (read+write)\n", grid, t, (bytes + bytes / 16.0) / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(expand16_cached, dim3(grid), dim3(256), 0, 0, (const uint8_t *)s, d, n, (uint64_t)((1u << 20) - 1)); }, reps);
        printf("expand src 1 MiB  grid %5d: %.3f ms  %.2f TB/s (write only counted)\n", grid, t, bytes / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(expand16_cached, dim3(grid), dim3(256), 0, 0, (const uint8_t *)s, d, n, (uint64_t)((64u << 20) - 1)); }, reps);
        printf("expand src 64 MiB grid %5d: %.3f ms  %.2f TB/s (write only counted)\n", grid, t, bytes / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(expand16_wide, dim3(grid), dim3(256), 0, 0, (const uint8_t *)s, d, n); }, reps);
        printf("expand wide+LDS   grid %5d: %.3f ms  %.2f TB/s (read+write)\n", grid, t, (bytes + bytes / 16.0) / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(expand16_pipe<2>, dim3(grid), dim3(256), 0, 0, (const uint8_t *)s, d, n); }, reps);
        printf("expand pipe D=2   grid %5d: %.3f ms  %.2f TB/s (read+write)\n", grid, t, (bytes + bytes / 16.0) / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(expand16_pipe<4>, dim3(grid), dim3(256), 0, 0, (const uint8_t *)s, d, n); }, reps);
        printf("expand pipe D=4   grid %5d: %.3f ms  %.2f TB/s (read+write)\n", grid, t, (bytes + bytes / 16.0) / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(expand16_pipe<8>, dim3(grid), dim3(256), 0, 0, (const uint8_t *)s, d, n); }, reps);
        printf("expand pipe D=8   grid %5d: %.3f ms  %.2f TB/s (read+write)\n", grid, t, (bytes + bytes / 16.0) / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(expand16_pipe<16>, dim3(grid), dim3(256), 0, 0, (const uint8_t *)s, d, n); }, reps);
        printf("expand pipe D=16  grid %5d: %.3f ms  %.2f TB/s (read+write)\n", grid, t, (bytes + bytes / 16.0) / t / 1e9);
    }
    double t = time_ms([&] { CK(hipMemsetAsync(d, 0x5a, bytes, 0)); }, reps);
    printf("hipMemsetAsync              : %.3f ms  %.2f TB/s\n", t, bytes / t / 1e9);
    t = time_ms([&] { CK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0)); }, reps);
    printf("hipMemcpyAsync D2D          : %.3f ms  %.2f TB/s (read+write)\n", t, 2.0 * bytes / t / 1e9);
    return 0;
}
