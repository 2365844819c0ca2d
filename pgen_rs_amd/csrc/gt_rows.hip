// gt_rows.hip — general row-tiled GT decode/emit kernel for gfx950 (MI355X).
//
// Replaces the nested loop of Pfile::output_vcf, /root/reference/src/pfile.rs:156-192:
// one launch covers a block of kept variants; a work item is (output row j, tile of aligned
// 16-byte chunks of that row's output bytes).  Output-driven: every lane owns one 16-B-ALIGNED
// chunk of the output address space, so interior chunks leave as whole global_store_dwordx4 and
// a wave writes 1 KiB of contiguous, 128-B-line-covering text per store instruction — the
// output is 16x the input (2 bits -> 4 bytes), so the kernel is HBM-WRITE-bound and store shape
// is what matters.  A row's GT segment starts at an arbitrary byte address (row pitch 4K+1 is
// odd; full lines start after a variable-length prefix), so the text dwords are funnel-shifted
// (v_alignbyte_b32) by the row's phase; the first/last chunk of a row and prefix seams take a
// per-byte edge path with masked byte stores.
//
// This kernel accepts every argument combination of the C ABI (any alignment/stride, variant
// index gather, kept-sample list gather, optional line prefixes) and is the correctness
// baseline the specialised kernels are checked against.
#include "gt_common.hip.h"
#include "kernels.h"

namespace pgenhip {

namespace {

constexpr int kThreads = 256;
constexpr int kChunksPerLane = 4;                       // 4 x 16 B per lane per work item
constexpr uint32_t kTileChunks = kThreads * kChunksPerLane;  // 16 KiB of output per item

template <bool SUBSET>
__device__ __forceinline__ uint32_t load_code(const uint8_t *__restrict__ rec,
                                              const uint32_t *__restrict__ kept_idx, uint32_t k)
{
    // src/pfile.rs:171-175
    uint32_t s = SUBSET ? kept_idx[k] : k;
    return ((uint32_t)rec[s >> 2] >> ((s & 3u) * 2u)) & 3u;
}

template <bool SUBSET, bool LINES>
__global__ __launch_bounds__(kThreads) void gt_rows_kernel(EmitArgs a, uint32_t tiles_per_row,
                                                           uint64_t total_items)
{
    const uint32_t tid = threadIdx.x;
    const uint64_t gt_bytes = 4ull * a.kept_count;  // text bytes of one row, without '\n'
    const uint32_t last_rec_byte = a.record_size ? a.record_size - 1u : 0u;

    for (uint64_t item = blockIdx.x; item < total_items; item += gridDim.x) {
        const uint32_t j = (uint32_t)(item / tiles_per_row);
        const uint32_t tile = (uint32_t)(item - (uint64_t)j * tiles_per_row);
        const uint8_t *__restrict__ rec = gathered(a) ? gathered_record(a, j) : a.records + (uint64_t)j * a.record_stride;

        uint64_t line_addr, prefix_pos = 0, prefix_len = 0;
        if (LINES) {
            prefix_pos = a.prefix_off[j];
            prefix_len = a.prefix_off[j + 1] - prefix_pos;
            line_addr = (uint64_t)(uintptr_t)a.out + a.line_off[j];
        } else {
            line_addr = (uint64_t)(uintptr_t)a.out + (uint64_t)j * a.out_stride;
        }
        const uint64_t gt_addr = line_addr + prefix_len;
        const uint64_t line_len = prefix_len + gt_bytes + 1ull;
        const uint64_t chunk0 = line_addr >> 4;
        const uint32_t n_chunks = (uint32_t)(((line_addr + line_len - 1ull) >> 4) - chunk0) + 1u;

#pragma unroll
        for (int u = 0; u < kChunksPerLane; u++) {
            const uint32_t i = tile * kTileChunks + (uint32_t)u * kThreads + tid;
            if (i >= n_chunks) break;
            const uint64_t caddr = (chunk0 + i) << 4;
            const int64_t q = (int64_t)(caddr - gt_addr);  // chunk start relative to the GT segment
            u32x4 *dst = reinterpret_cast<u32x4 *>(a.out + (int64_t)(caddr - (uint64_t)(uintptr_t)a.out));  // stays a global pointer

            if (q >= 0 && (uint64_t)q + 16ull <= gt_bytes) {
                // interior chunk: bytes q..q+15 of the GT segment = samples k0..k0+4, phase sh
                if (!SUBSET) {
                    // 10-bit window of samples k0..k0+4 (record bytes q/16 and q/16+1)
                    *dst = gt_text16_from_window(load_window<false>(rec, (int32_t)(q >> 4), last_rec_byte), q);
                } else {
                    const uint32_t k0 = (uint32_t)((uint64_t)q >> 2);
                    const uint32_t sh = (uint32_t)q & 3u;
                    const uint32_t klast = a.kept_count - 1u;
                    const uint32_t t0 = gt_text(load_code<true>(rec, a.kept_idx, k0));
                    const uint32_t t1 = gt_text(load_code<true>(rec, a.kept_idx, k0 + 1u));
                    const uint32_t t2 = gt_text(load_code<true>(rec, a.kept_idx, k0 + 2u));
                    const uint32_t t3 = gt_text(load_code<true>(rec, a.kept_idx, k0 + 3u));
                    const uint32_t t4 = gt_text(load_code<true>(rec, a.kept_idx, min(k0 + 4u, klast)));
                    u32x4 v;
                    v.x = funnel_bytes(t0, t1, sh);
                    v.y = funnel_bytes(t1, t2, sh);
                    v.z = funnel_bytes(t2, t3, sh);
                    v.w = funnel_bytes(t3, t4, sh);
                    *dst = v;
                }
            } else {
                // edge chunk (row head/tail, prefix bytes, '\n'): one byte at a time
                uint32_t d[4] = {0u, 0u, 0u, 0u};
                uint32_t valid = 0u;
#pragma unroll
                for (int b = 0; b < 16; b++) {
                    const int64_t p = (int64_t)(caddr + (uint64_t)b - line_addr);  // offset in line
                    if (p < 0 || (uint64_t)p >= line_len) continue;
                    uint32_t ch;
                    if (LINES && (uint64_t)p < prefix_len) {
                        ch = a.prefix_blob[prefix_pos + (uint64_t)p];              // :157-161
                    } else {
                        const uint64_t g = (uint64_t)p - prefix_len;
                        if (g == gt_bytes) {
                            ch = '\n';                                              // :190
                        } else {
                            const uint32_t code = load_code<SUBSET>(rec, a.kept_idx, (uint32_t)(g >> 2));
                            ch = gt_text_byte(code, (uint32_t)g & 3u);              // :177-187
                        }
                    }
                    d[b >> 2] |= ch << (8 * (b & 3));
                    valid |= 1u << b;
                }
                if (valid == 0xFFFFu) {
                    u32x4 v = {d[0], d[1], d[2], d[3]};
                    *dst = v;
                } else {
                    uint8_t *bp = reinterpret_cast<uint8_t *>(dst);
#pragma unroll
                    for (int b = 0; b < 16; b++) {
                        if (valid & (1u << b)) bp[b] = (uint8_t)(d[b >> 2] >> (8 * (b & 3)));
                    }
                }
            }
        }
    }
}

}  // namespace

hipError_t launch_gt_rows(const EmitArgs &a, int num_cus, hipStream_t stream)
{
    if (a.n_variants == 0) return hipSuccess;
    const bool lines = a.line_off != nullptr;
    const bool subset = a.kept_idx != nullptr;
    const uint64_t max_line = lines ? a.max_line_bytes : 4ull * a.kept_count + 1ull;
    // a line of L bytes at an arbitrary address touches at most ceil((L+15)/16) aligned chunks
    const uint64_t max_chunks = (max_line + 15ull + 15ull) / 16ull;
    const uint32_t tiles_per_row = (uint32_t)((max_chunks + kTileChunks - 1ull) / kTileChunks);
    const uint64_t total_items = (uint64_t)a.n_variants * tiles_per_row;
    const uint64_t max_grid = (uint64_t)num_cus * 8ull;  // 8 blocks of 256 threads fill a CU
    const uint32_t grid = (uint32_t)(total_items < max_grid ? total_items : max_grid);

    void (*kern)(EmitArgs, uint32_t, uint64_t);
    if (subset)
        kern = lines ? gt_rows_kernel<true, true> : gt_rows_kernel<true, false>;
    else
        kern = lines ? gt_rows_kernel<false, true> : gt_rows_kernel<false, false>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), 0, stream, a, tiles_per_row, total_items);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// synthetic records: LE words splitmix64(seed + (v << 20) + word_idx) truncated to R
namespace {
__global__ __launch_bounds__(256) void synth_records_kernel(uint8_t *dst, uint64_t record_stride,
                                                            uint32_t sample_count, uint32_t record_size,
                                                            uint64_t first_variant, uint32_t n_variants,
                                                            uint64_t seed, uint32_t dirty_pad)
{
    const uint32_t words_per_rec = (record_size + 7u) / 8u;
    const uint64_t total = (uint64_t)n_variants * words_per_rec;
    const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += step) {
        const uint32_t j = (uint32_t)(idx / words_per_rec);
        const uint32_t w = (uint32_t)(idx - (uint64_t)j * words_per_rec);
        uint64_t val = splitmix64(seed + ((first_variant + j) << 20) + (uint64_t)w);
        const uint32_t byte0 = w * 8u;
        const uint32_t nbytes = min(8u, record_size - byte0);
        if (!dirty_pad && (sample_count & 3u) != 0u && byte0 + nbytes == record_size) {
            const uint32_t bi = record_size - 1u - byte0;  // last record byte inside this word
            const uint64_t keep = (1ull << ((sample_count & 3u) * 2u)) - 1ull;
            val &= ~(0xFFull << (8u * bi)) | (keep << (8u * bi));
        }
        uint8_t *p = dst + (uint64_t)j * record_stride + byte0;
        if (nbytes == 8u && (((uintptr_t)p) & 7u) == 0u) {
            *reinterpret_cast<uint64_t *>(p) = val;
        } else {
            for (uint32_t b = 0; b < nbytes; b++) p[b] = (uint8_t)(val >> (8u * b));
        }
    }
}

// "hwe" distribution: per-variant allele frequency p16 / 65536 in [0.01, 0.5), two independent allele
// draws per sample (codes 0/1/2 in Hardy-Weinberg proportions), 0.1 % missing; integer-only — the oracle holds the twin.
// One thread per 64-bit word of a record (32 samples).
__global__ __launch_bounds__(256) void synth_records_hwe_kernel(uint8_t *dst, uint64_t record_stride, uint32_t sample_count,
                                                                uint32_t record_size, uint64_t first_variant, uint32_t n_variants, uint64_t seed)
{
    const uint32_t words_per_rec = (record_size + 7u) / 8u;
    const uint64_t total = (uint64_t)n_variants * words_per_rec;
    const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += step) {
        const uint32_t j = (uint32_t)(idx / words_per_rec);
        const uint32_t w = (uint32_t)(idx - (uint64_t)j * words_per_rec);
        const uint64_t v = first_variant + j;
        const uint32_t p16 = 655u + (uint32_t)(splitmix64((seed ^ 0x4D4146ull) + v) % 32113ull);
        const uint64_t kv = splitmix64((seed ^ 0x485745ull) + v);
        uint64_t val = 0ull;
        const uint32_t s0 = w * 32u;
        for (uint32_t i = 0; i < 32u && s0 + i < sample_count; i++) {
            const uint64_t h = splitmix64(kv + (uint64_t)(s0 + i));
            uint32_t code = (uint32_t)((uint32_t)(h & 0xFFFFull) < p16) + (uint32_t)((uint32_t)((h >> 16) & 0xFFFFull) < p16);
            if ((uint32_t)(h >> 32) < 4294967u) code = 3u;
            val |= (uint64_t)code << (2u * i);
        }
        const uint32_t byte0 = w * 8u;
        const uint32_t nbytes = min(8u, record_size - byte0);
        uint8_t *p = dst + (uint64_t)j * record_stride + byte0;
        if (nbytes == 8u && (((uintptr_t)p) & 7u) == 0u) {
            *reinterpret_cast<uint64_t *>(p) = val;
        } else {
            for (uint32_t b = 0; b < nbytes; b++) p[b] = (uint8_t)(val >> (8u * b));
        }
    }
}
}  // namespace

hipError_t launch_synth_records_hwe(uint8_t *dst, uint64_t record_stride, uint32_t sample_count, uint64_t first_variant,
                                    uint32_t n_variants, uint64_t seed, int num_cus, hipStream_t stream)
{
    const uint32_t record_size = (sample_count * 2u) / 8u + (((sample_count * 2u) % 8u) ? 1u : 0u);
    if (n_variants == 0 || record_size == 0) return hipSuccess;
    const uint64_t total = (uint64_t)n_variants * ((record_size + 7u) / 8u);
    const uint64_t blocks_needed = (total + 255ull) / 256ull;
    const uint64_t max_grid = (uint64_t)num_cus * 16ull;
    hipLaunchKernelGGL(synth_records_hwe_kernel, dim3((uint32_t)(blocks_needed < max_grid ? blocks_needed : max_grid)), dim3(256), 0, stream,
                       dst, record_stride, sample_count, record_size, first_variant, n_variants, seed);
    return hipGetLastError();
}

hipError_t launch_synth_records(uint8_t *dst, uint64_t record_stride, uint32_t sample_count,
                                uint64_t first_variant, uint32_t n_variants, uint64_t seed,
                                bool dirty_pad, int num_cus, hipStream_t stream)
{
    const uint32_t record_size = (sample_count * 2u) / 8u + (((sample_count * 2u) % 8u) ? 1u : 0u);
    if (n_variants == 0 || record_size == 0) return hipSuccess;
    const uint64_t total = (uint64_t)n_variants * ((record_size + 7u) / 8u);
    const uint64_t blocks_needed = (total + 255ull) / 256ull;
    const uint64_t max_grid = (uint64_t)num_cus * 8ull;
    const uint32_t grid = (uint32_t)(blocks_needed < max_grid ? blocks_needed : max_grid);
    hipLaunchKernelGGL(synth_records_kernel, dim3(grid), dim3(256), 0, stream, dst, record_stride,
                       sample_count, record_size, first_variant, n_variants, seed, dirty_pad ? 1u : 0u);
    return hipGetLastError();
}

}  // namespace pgenhip
