// readbench.hip — which READ PATTERN over a block of long records reaches the HBM read ceiling?
// Tuning aid only (not part of the library).  The compact pass of the two-pass path (gt_scan.hip, BASELINE configs[4]) is a
// pure record reader that sat at 5.1 TB/s (0.64 of 8) in round 2 and moved 12 % between boxes.  This program replays its
// loads — and alternatives — without the pick/compact work, on the same geometry (V rows of R bytes, 4-KiB segments):
//   seg_xcd / seg_plain   the kernel's map: block = (segment, row group), wave = one row, 4 x 1-KiB loads per row piece
//   linear                waves take 4-KiB pieces in memory order (piece q = row * n_seg + seg), any number of waves per CU
//   stream                plain grid-stride 16-B/lane loads over the whole block (the known ceiling, tools/membench.hip read16)
//   hipcc -O3 --offload-arch=gfx950 -o tools/readbench tools/readbench.hip ; tools/readbench [V] [R]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream_read(const v4u *src, uint64_t n_chunks, uint32_t *sink)
{
    const uint64_t step = (uint64_t)gridDim.x * 256u;
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n_chunks; i += step) {
        v4u v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) *sink = acc;
}

__device__ __forceinline__ void load_piece(const uint8_t *rec, uint32_t R, uint32_t seg, uint32_t lane, v4u (&dst)[4])
{
    // 4 x 1 KiB of the row's segment; loads past the row's end are pulled back inside (like the kernel's tail window)
#pragma unroll
    for (uint32_t t = 0; t < 4u; t++) {
        uint32_t off = seg * 4096u + t * 1024u + lane * 16u;
        if (off + 16u > R) off = R - 16u;
        __builtin_memcpy(&dst[t], rec + off, 16);
    }
}

__device__ __forceinline__ uint32_t fold(const v4u (&b)[4])
{
    uint32_t a = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) a ^= b[t].x ^ b[t].y ^ b[t].z ^ b[t].w;
    return a;
}

// the kernel's map (gt_scan.hip): DEPTH row pieces in flight per wave
template <int DEPTH>
__global__ __launch_bounds__(256) void seg_read(const uint8_t *recs, uint64_t V, uint32_t R, uint32_t n_seg, uint32_t row_groups, uint32_t xcd_groups, uint32_t *sink)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool xcd_map = blockIdx.x < xcd_groups * n_seg;
    const uint32_t b_plain = blockIdx.x - xcd_groups * n_seg;
    const uint32_t seg = xcd_map ? (blockIdx.x >> 3) % n_seg : b_plain % n_seg;
    const uint32_t row_group = xcd_map ? ((blockIdx.x >> 3) / n_seg) * 8u + (blockIdx.x & 7u) : xcd_groups + b_plain / n_seg;
    const uint64_t row_step = (uint64_t)row_groups * 4u;
    const uint64_t j0 = (uint64_t)row_group * 4u + wave;
    const uint64_t rows = j0 < V ? (V - j0 + row_step - 1) / row_step : 0;
    if (!rows) return;
    v4u buf[DEPTH][4];
    uint32_t acc = 0;
#pragma unroll
    for (int d = 0; d < DEPTH - 1; d++) load_piece(recs + (j0 + std::min<uint64_t>(d, rows - 1) * row_step) * R, R, seg, lane, buf[d]);
    for (uint64_t n = 0; n < rows; n += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            if (n + d < rows) {
                load_piece(recs + (j0 + std::min<uint64_t>(n + d + DEPTH - 1, rows - 1) * row_step) * R, R, seg, lane, buf[(d + DEPTH - 1) % DEPTH]);
                acc ^= fold(buf[d]);
            }
        }
    }
    if (acc == 0x12345u) *sink = acc;
}

// pieces in memory order
template <int DEPTH>
__global__ __launch_bounds__(256) void linear_read(const uint8_t *recs, uint64_t V, uint32_t R, uint32_t n_seg, uint32_t *sink)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave_gid = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint64_t step = (uint64_t)gridDim.x * 4u;
    const uint64_t pieces = V * n_seg;
    if (wave_gid >= pieces) return;
    const uint64_t mine = (pieces - wave_gid + step - 1) / step;
    v4u buf[DEPTH][4];
    uint32_t acc = 0;
    auto piece = [&](uint64_t n, v4u (&dst)[4]) {
        const uint64_t q = wave_gid + std::min<uint64_t>(n, mine - 1) * step;
        const uint64_t row = q / n_seg;
        load_piece(recs + row * R, R, (uint32_t)(q - row * n_seg), lane, dst);
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; d++) piece(d, buf[d]);
    for (uint64_t n = 0; n < mine; n += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            if (n + d < mine) {
                piece(n + d + DEPTH - 1, buf[(d + DEPTH - 1) % DEPTH]);
                acc ^= fold(buf[d]);
            }
        }
    }
    if (acc == 0x12345u) *sink = acc;
}

// pieces in memory order, but a BLOCK owns a contiguous window of the piece sequence (one CU, one XCD work in one narrow window)
template <int DEPTH>
__global__ __launch_bounds__(256) void window_read(const uint8_t *recs, uint64_t V, uint32_t R, uint32_t n_seg, uint32_t *sink)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t pieces = V * n_seg;
    const uint64_t per_block = (pieces + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * per_block, hi = std::min(pieces, lo + per_block);
    if (lo + wave >= hi) return;
    const uint64_t mine = (hi - lo - wave + 3) / 4;
    v4u buf[DEPTH][4];
    uint32_t acc = 0;
    auto piece = [&](uint64_t n, v4u (&dst)[4]) {
        const uint64_t q = lo + wave + std::min<uint64_t>(n, mine - 1) * 4u;
        const uint64_t row = q / n_seg;
        load_piece(recs + row * R, R, (uint32_t)(q - row * n_seg), lane, dst);
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; d++) piece(d, buf[d]);
    for (uint64_t n = 0; n < mine; n += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            if (n + d < mine) {
                piece(n + d + DEPTH - 1, buf[(d + DEPTH - 1) % DEPTH]);
                acc ^= fold(buf[d]);
            }
        }
    }
    if (acc == 0x12345u) *sink = acc;
}

// a WAVE owns whole rows (gt_rowpick.hip): it streams its row's n_seg pieces in order, row = wave id + i * waves
template <int DEPTH>
__global__ __launch_bounds__(256) void rowwave_read(const uint8_t *recs, uint64_t V, uint32_t R, uint32_t n_seg, uint32_t *sink)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave_gid = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint64_t step = (uint64_t)gridDim.x * 4u;
    if (wave_gid >= V) return;
    const uint64_t rows = (V - wave_gid + step - 1) / step;
    const uint64_t mine = rows * n_seg;
    v4u buf[DEPTH][4];
    uint32_t acc = 0;
    auto piece = [&](uint64_t n, v4u (&dst)[4]) {
        const uint64_t q = std::min<uint64_t>(n, mine - 1);
        const uint64_t i = q / n_seg;
        load_piece(recs + (wave_gid + i * step) * R, R, (uint32_t)(q - i * n_seg), lane, dst);
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; d++) piece(d, buf[d]);
    for (uint64_t n = 0; n < mine; n += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            if (n + d < mine) {
                piece(n + d + DEPTH - 1, buf[(d + DEPTH - 1) % DEPTH]);
                acc ^= fold(buf[d]);
            }
        }
    }
    if (acc == 0x12345u) *sink = acc;
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
    const uint64_t V = argc > 1 ? strtoull(argv[1], 0, 10) : 41667ull;
    const uint32_t R = argc > 2 ? (uint32_t)strtoul(argv[2], 0, 10) : 125000u;
    const uint32_t n_seg = (R + 4095u) / 4096u;
    const uint64_t bytes = V * R;
    uint8_t *d;
    uint32_t *sink;
    CK(hipMalloc(&d, bytes + 64));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(d, 1, bytes));
    const int reps = 7;
    printf("V %llu x R %u = %.2f GB, %u segments per row\n", (unsigned long long)V, R, bytes / 1e9, n_seg);
    auto report = [&](const char *name, int per_cu, double t) { printf("%-28s %2d blocks/CU: %8.3f ms  %.2f TB/s\n", name, per_cu, t, bytes / t / 1e9); fflush(stdout); };
    if (argc > 3) {
        // position sweep: the same pattern on successive sub-blocks of ONE big allocation (does WHERE the records lie matter?)
        const uint64_t parts = strtoull(argv[3], 0, 10);
        const uint64_t Vp = V / parts;
        const uint32_t groups = 512u / n_seg, grid = groups * n_seg, xg = groups & ~7u;
        for (int round = 0; round < 2; round++)
            for (uint64_t q = 0; q < parts; q++) {
                const uint8_t *base = d + q * Vp * R;
                const double t = time_ms([&] { hipLaunchKernelGGL(seg_read<2>, dim3(grid), dim3(256), 0, 0, base, Vp, R, n_seg, groups, xg, sink); }, reps);
                const double t2 = time_ms([&] { hipLaunchKernelGGL(stream_read, dim3(2048), dim3(256), 0, 0, (const v4u *)(base + (16 - ((uintptr_t)base & 15)) % 16), Vp * R / 16 - 1, sink); }, reps);
                printf("part %2llu (rows %llu..): seg_xcd %.3f ms %.2f TB/s   stream %.3f ms %.2f TB/s\n", (unsigned long long)q, (unsigned long long)(q * Vp), t, Vp * R / t / 1e9, t2, Vp * R / t2 / 1e9);
                fflush(stdout);
            }
        return 0;
    }
    for (int per_cu : {4, 8}) {
        const int grid = per_cu * 256;
        report("stream", per_cu, time_ms([&] { hipLaunchKernelGGL(stream_read, dim3(grid), dim3(256), 0, 0, (const v4u *)d, bytes / 16, sink); }, reps));
    }
    for (int per_cu : {2, 3, 4}) {
        const uint32_t groups = (uint32_t)(per_cu * 256) / n_seg;
        const uint32_t grid = groups * n_seg;
        const uint32_t xg = groups & ~7u;
        report("seg_xcd depth 2", per_cu, time_ms([&] { hipLaunchKernelGGL(seg_read<2>, dim3(grid), dim3(256), 0, 0, d, V, R, n_seg, groups, xg, sink); }, reps));
        report("seg_plain depth 2", per_cu, time_ms([&] { hipLaunchKernelGGL(seg_read<2>, dim3(grid), dim3(256), 0, 0, d, V, R, n_seg, groups, 0u, sink); }, reps));
        report("seg_xcd depth 3", per_cu, time_ms([&] { hipLaunchKernelGGL(seg_read<3>, dim3(grid), dim3(256), 0, 0, d, V, R, n_seg, groups, xg, sink); }, reps));
        report("seg_xcd depth 4", per_cu, time_ms([&] { hipLaunchKernelGGL(seg_read<4>, dim3(grid), dim3(256), 0, 0, d, V, R, n_seg, groups, xg, sink); }, reps));
        report("linear depth 2", per_cu, time_ms([&] { hipLaunchKernelGGL(linear_read<2>, dim3(per_cu * 256), dim3(256), 0, 0, d, V, R, n_seg, sink); }, reps));
        report("linear depth 3", per_cu, time_ms([&] { hipLaunchKernelGGL(linear_read<3>, dim3(per_cu * 256), dim3(256), 0, 0, d, V, R, n_seg, sink); }, reps));
        report("linear depth 4", per_cu, time_ms([&] { hipLaunchKernelGGL(linear_read<4>, dim3(per_cu * 256), dim3(256), 0, 0, d, V, R, n_seg, sink); }, reps));
        report("rowwave depth 2", per_cu, time_ms([&] { hipLaunchKernelGGL(rowwave_read<2>, dim3(per_cu * 256), dim3(256), 0, 0, d, V, R, n_seg, sink); }, reps));
        report("rowwave depth 3", per_cu, time_ms([&] { hipLaunchKernelGGL(rowwave_read<3>, dim3(per_cu * 256), dim3(256), 0, 0, d, V, R, n_seg, sink); }, reps));
        report("window depth 2", per_cu, time_ms([&] { hipLaunchKernelGGL(window_read<2>, dim3(per_cu * 256), dim3(256), 0, 0, d, V, R, n_seg, sink); }, reps));
        report("window depth 4", per_cu, time_ms([&] { hipLaunchKernelGGL(window_read<4>, dim3(per_cu * 256), dim3(256), 0, 0, d, V, R, n_seg, sink); }, reps));
    }
    return 0;
}
