// wpattern.hip — pure-write microbenchmark of the WRITE PATTERNS of the GT kernels (tuning aid, not part of the library):
// does the memory side take the segment kernel's pattern (thousands of thin streams, a KiB per wave and step, picks between
// the stores) as fast as the stream kernel's (16-KiB items in sequence, sixteen stores back to back)?
//   hipcc -O3 --offload-arch=gfx950 -o tools/wpattern tools/wpattern.hip
//   tools/wpattern [total_GB] — prints TB/s for a grid of (pattern, waves per CU, chunks per step, VALU work per chunk)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// text-like data from a few dependent VALU operations (WORK of them per chunk: the picks' stand-in)
__device__ __forceinline__ v4u make_text(uint32_t seed, uint32_t work)
{
    uint32_t h = seed * 2654435761u + 12345u;
    for (uint32_t k = 0; k < work; k++) h = h * 1664525u + 1013904223u;   // (v_mul_lo + v_add: ~6 issue slots per round)
    return v4u{0x302F3009u + ((h >> 3) & 1u) * 0x01000000u, 0x302F3009u + ((h >> 7) & 1u) * 0x01000100u,
               0x302F3009u + ((h >> 11) & 1u) * 0x01000000u, 0x302F3009u + ((h >> 17) & 1u) * 0x01000100u};
}

// SEGMENT-kernel pattern: rows of `row_bytes`, cut into `pieces` pieces; block = 4 waves, owns piece p of the rows of its row group;
// wave w of row group g writes piece p of rows 4 g + w, 4 g + w + 4 G, ...: U KiB per step, the piece front to back.
// DRAIN: the wave waits for all its stores at the end of every piece (what a load behind the stores costs the segment kernel:
// gfx9 has one counter for loads and stores, and the compiler waits vmcnt(0) for a load that follows stores in a loop)
// READ: every piece starts with the wave's 4-KiB read of its record segment (4 x 16 B per lane, waited for before the first store), like
// the real kernel's: do a few per cent of reads woven into the write streams cost more than their bytes?
template <int U, bool NT, bool DRAIN = false, bool JITTER = false, bool READ = false>
__global__ __launch_bounds__(256) void seg_pattern(uint8_t *out, uint64_t rows, uint32_t row_bytes, uint32_t pieces, uint32_t groups, uint32_t work, const uint8_t *recs = nullptr)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t p = blockIdx.x % pieces, g = blockIdx.x / pieces;
    const uint32_t piece_bytes = row_bytes / pieces;                       // (host: a multiple of 1024 U)
    const uint32_t steps = piece_bytes / (1024u * U);
    for (uint64_t row = (uint64_t)g * 4u + wave; row < rows; row += (uint64_t)groups * 4u) {
        uint8_t *dst = out + row * row_bytes + (uint64_t)p * piece_bytes + lane * 16u;
        uint32_t mix = 0;
        if (READ) {
            const uint8_t *src = recs + (row * pieces + p) * 4096ull + lane * 16u;
            v4u r0 = *reinterpret_cast<const v4u *>(src), r1 = *reinterpret_cast<const v4u *>(src + 1024), r2 = *reinterpret_cast<const v4u *>(src + 2048),
                r3 = *reinterpret_cast<const v4u *>(src + 3072);
            mix = (r0.x ^ r1.y ^ r2.z ^ r3.w) & 1u;
        }
        for (uint32_t s = 0; s < steps; s++) {
            v4u v[U];
            // JITTER: the work per step varies by +-50 % from wave to wave and step to step (do waves that all take equally long per step
            // fall into lock step and send their stores in bursts?)
            const uint32_t w_s = JITTER ? work / 2u + (((uint32_t)row * 2654435761u + s * 40503u + blockIdx.x * 97u + wave * 31u) >> 7) % (work + 1u) : work;
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = make_text((uint32_t)row + s * 64u + lane + u + mix, w_s);
#pragma unroll
            for (int u = 0; u < U; u++) {
                v4u *q = reinterpret_cast<v4u *>(dst + (s * U + u) * 1024u);
                if (NT) __builtin_nontemporal_store(v[u], q); else *q = v[u];
            }
        }
        if (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// The segment pattern with its reads gathered into BURSTS: the wave reads the 4-KiB record segments of RB of its rows at once
// (RB x 4 loads of 16 B per lane), then writes those RB pieces without another read ("few, large read bursts disturb the HBM write
// stream far less than many small ones", gt_wide.hip).
template <int U, int RB, bool NTLOAD = false>
__global__ __launch_bounds__(256) void seg_pattern_burst(uint8_t *out, uint64_t rows, uint32_t row_bytes, uint32_t pieces, uint32_t groups, uint32_t work, const uint8_t *recs)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t p = blockIdx.x % pieces, g = blockIdx.x / pieces;
    const uint32_t piece_bytes = row_bytes / pieces;
    const uint32_t steps = piece_bytes / (1024u * U);
    const uint64_t row_step = (uint64_t)groups * 4u;
    for (uint64_t row0 = (uint64_t)g * 4u + wave; row0 < rows; row0 += row_step * RB) {
        uint32_t mix[RB];
#pragma unroll
        for (int b = 0; b < RB; b++) {
            const uint64_t row = min(row0 + (uint64_t)b * row_step, rows - 1ull);
            const uint8_t *src = recs + (row * pieces + p) * 4096ull + lane * 16u;
            v4u r0, r1, r2, r3;
            if (NTLOAD) {
                r0 = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(src));
                r1 = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(src + 1024));
                r2 = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(src + 2048));
                r3 = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(src + 3072));
            } else {
                r0 = *reinterpret_cast<const v4u *>(src); r1 = *reinterpret_cast<const v4u *>(src + 1024); r2 = *reinterpret_cast<const v4u *>(src + 2048);
                r3 = *reinterpret_cast<const v4u *>(src + 3072);
            }
            mix[b] = (r0.x ^ r1.y ^ r2.z ^ r3.w) & 1u;
        }
#pragma unroll
        for (int b = 0; b < RB; b++) {
            const uint64_t row = row0 + (uint64_t)b * row_step;
            if (row >= rows) break;
            uint8_t *dst = out + row * row_bytes + (uint64_t)p * piece_bytes + lane * 16u;
            for (uint32_t s = 0; s < steps; s++) {
                v4u v[U];
#pragma unroll
                for (int u = 0; u < U; u++) v[u] = make_text((uint32_t)row + s * 64u + lane + u + mix[b], work);
#pragma unroll
                for (int u = 0; u < U; u++) __builtin_nontemporal_store(v[u], reinterpret_cast<v4u *>(dst + (s * U + u) * 1024u));
            }
        }
    }
}

// STREAM-kernel pattern: the output as consecutive items of `item_kib` KiB, dealt to the waves in order by one atomic counter per
// range (8 ranges); a wave builds and stores its item's KiBs back to back.
// READ: one 16-byte-per-lane load per item from the item's place in a record stream 16 times smaller (the stream kernel's loader)
template <bool NT, bool READ = false>
__global__ __launch_bounds__(256) void stream_pattern(uint8_t *out, uint64_t n_items, uint32_t item_kib, uint32_t work, unsigned long long *heads, const uint8_t *recs = nullptr)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nr = 8u;
    const uint64_t per_range = (n_items + nr - 1u) / nr;
    uint32_t range = blockIdx.x & (nr - 1u);
    for (uint32_t tries = 0; tries < nr;) {
        unsigned long long t = 0;
        if (lane == 0u) t = atomicAdd(&heads[range * 16u], 1ull);
        t = __shfl(t, 0, 64);
        const uint64_t item = (uint64_t)range * per_range + t;
        if (t >= per_range || item >= n_items) { range = (range + 1u) & (nr - 1u); tries++; continue; }
        uint8_t *dst = out + item * item_kib * 1024ull + lane * 16u;
        uint32_t mix = 0;
        if (READ) {
            const v4u r0 = *reinterpret_cast<const v4u *>(recs + item * (item_kib * 64ull) + (lane * 16u) % (item_kib * 64u));
            mix = (r0.x ^ r0.w) & 1u;
        }
        for (uint32_t s = 0; s < item_kib; s++) {
            v4u v = make_text((uint32_t)item + s * 64u + lane + mix, work);
            v4u *q = reinterpret_cast<v4u *>(dst + s * 1024u);
            if (NT) __builtin_nontemporal_store(v, q); else *q = v;
        }
    }
}

template <typename F>
static double time_ms(F launch, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main(int argc, char **argv)
{
    const double gb = argc > 1 ? atof(argv[1]) : 6.0;
    const uint32_t row_bytes = 7u * 32768u;                                // 7 pieces of 32 KiB (N = 100 000 with 50 % kept)
    const uint64_t rows = (uint64_t)(gb * 1e9 / row_bytes);
    const uint64_t total = rows * row_bytes;
    uint8_t *out; CK(hipMalloc(&out, total + 4096));
    unsigned long long *heads; CK(hipMalloc(&heads, 8 * 16 * sizeof(unsigned long long)));
    uint8_t *recs; CK(hipMalloc(&recs, rows * 28ull * 4096ull + 4096));
    CK(hipMemset(recs, 0x5A, rows * 28ull * 4096ull + 4096));
    printf("rows %llu x %u bytes = %.2f GB\n", (unsigned long long)rows, row_bytes, total / 1e9);
    const int cus = 256;
    for (uint32_t work : {0u, 8u, 16u, 24u}) {
        for (int per_cu : {2, 3, 4}) {
            for (uint32_t pieces : {7u, 28u}) {
                const uint32_t groups = std::max(1u, (uint32_t)(per_cu * cus) / pieces);
                auto run = [&](auto kern, const char *name) {
                    double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(groups * pieces), dim3(256), 0, 0, out, rows, row_bytes, pieces, groups, work, (const uint8_t *)nullptr); }, 5);
                    printf("seg   work %2u  blocks/CU %d  pieces %2u  %-8s %7.3f ms  %.2f TB/s\n", work, per_cu, pieces, name, ms, total / ms / 1e9);
                };
                run(seg_pattern<1, false>, "U1");
                run(seg_pattern<2, true>, "U2 nt");
                run(seg_pattern<4, true>, "U4 nt");
                run(seg_pattern<2, true, true>, "U2 nt drain");
                run(seg_pattern<2, true, false, true>, "U2 nt jitter");
                auto run_burst = [&](auto kern, const char *name) {
                    double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(groups * pieces), dim3(256), 0, 0, out, rows, row_bytes, pieces, groups, work, (const uint8_t *)recs); }, 5);
                    printf("seg   work %2u  blocks/CU %d  pieces %2u  %-8s %7.3f ms  %.2f TB/s (writes; + %.0f %% read)\n", work, per_cu, pieces, name, ms, total / ms / 1e9, 100.0 * 4096.0 * pieces / row_bytes);
                };
                run_burst(seg_pattern_burst<2, 1>, "read burst 1");
                run_burst(seg_pattern_burst<2, 4>, "read burst 4");
                run_burst(seg_pattern_burst<2, 8>, "read burst 8");
                run_burst(seg_pattern_burst<2, 16>, "read burst 16");
                run_burst((seg_pattern_burst<2, 1, true>), "read nt-load");
                {
                    double ms = time_ms([&] { hipLaunchKernelGGL((seg_pattern<2, true, false, false, true>), dim3(groups * pieces), dim3(256), 0, 0, out, rows, row_bytes, pieces, groups, work, recs); }, 5);
                    printf("seg   work %2u  blocks/CU %d  pieces %2u  %-8s %7.3f ms  %.2f TB/s (writes; + %.0f %% read)\n", work, per_cu, pieces, "U2 nt read", ms, total / ms / 1e9, 100.0 * 4096.0 * pieces / row_bytes);
                }
            }
            for (uint32_t item_kib : {16u, 4u}) {
                const uint64_t n_items = total / (item_kib * 1024ull);
                double ms = time_ms([&] {
                    CK(hipMemsetAsync(heads, 0, 8 * 16 * sizeof(unsigned long long), 0));
                    hipLaunchKernelGGL(stream_pattern<true>, dim3(per_cu * cus), dim3(256), 0, 0, out, n_items, item_kib, work, heads, (const uint8_t *)nullptr);
                }, 5);
                printf("strm  work %2u  blocks/CU %d  item %2u KiB nt       %7.3f ms  %.2f TB/s\n", work, per_cu, item_kib, ms, total / ms / 1e9);
                ms = time_ms([&] {
                    CK(hipMemsetAsync(heads, 0, 8 * 16 * sizeof(unsigned long long), 0));
                    hipLaunchKernelGGL((stream_pattern<true, true>), dim3(per_cu * cus), dim3(256), 0, 0, out, n_items, item_kib, work, heads, (const uint8_t *)recs);
                }, 5);
                printf("strm  work %2u  blocks/CU %d  item %2u KiB nt read  %7.3f ms  %.2f TB/s (writes; + 6 %% read)\n", work, per_cu, item_kib, ms, total / ms / 1e9);
            }
        }
    }
    return 0;
}
