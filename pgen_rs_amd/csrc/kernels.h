// kernels.h — host-callable launchers of the gfx950 kernels (internal to libpgen_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pgenhip {

// One launch = one block of kept variants (src/pfile.rs:156 outer loop, many iterations at once).
struct EmitArgs {
    const uint8_t *records;       // device; row r at records + r*record_stride
    uint64_t record_stride;
    const uint32_t *variant_idx;  // device or nullptr (identity)
    uint32_t n_variants;
    uint32_t sample_count;        // N
    uint32_t record_size;         // R = ceil(N/4)
    const uint32_t *kept_idx;     // device or nullptr (all samples)
    uint32_t kept_count;          // K (== N when kept_idx is nullptr)
    uint8_t *out;                 // device
    uint64_t out_stride;          // GT-segment mode: row j at out + j*out_stride
    // full-line mode (all three non-null): line j at out + line_off[j], prefix bytes first
    const uint8_t *prefix_blob;
    const uint64_t *prefix_off;
    const uint64_t *line_off;
    uint64_t max_line_bytes;      // upper bound of any line's byte length (prefix + 4K + 1)
    uint64_t *work_counters;      // device scratch owned by the ctx: 8 x 128-B-spaced work-queue heads + a block-exit counter at +1024 B; zero between launches (the last block out re-zeroes them)
};

// General row-tiled kernel: any alignment, any strides, list gather for kept subsets.
hipError_t launch_gt_rows(const EmitArgs &a, int num_cus, hipStream_t stream);

// Dense all-samples stream kernel (K = N, out_stride == 4N+1): every lane owns one aligned
// 16-byte chunk of the whole launch's output stream.
bool gt_flat_applicable(const EmitArgs &a);
hipError_t launch_gt_flat(const EmitArgs &a, int num_cus, hipStream_t stream);

// Same contract as the flat kernel, but a wave stages its span's record bytes with one wide
// (16 B/lane) load into LDS and then issues 16 coalesced 1-KiB stores (rows >= 4 KiB of text).
bool gt_wide_applicable(const EmitArgs &a);
hipError_t launch_copy_prefixes(const EmitArgs &a, int num_cus, hipStream_t stream);  // full-line mode: prefix bytes of every line
bool gt_wide_lines_applicable(const EmitArgs &a);  // full-line mode (line_off/prefix_off set) through the stream kernel
hipError_t launch_gt_wide(const EmitArgs &a, int num_cus, hipStream_t stream);

// Stream-span kernel: like the wide stream kernel, but a work item is a 1-KiB-aligned 16-KiB span of the
// output stream (crossing row ends), so every store step is a full 1 KiB (N >= 2048; needs work_counters).
bool gt_span_applicable(const EmitArgs &a);
hipError_t launch_gt_span(const EmitArgs &a, int num_cus, hipStream_t stream);

// Kept-subset scan kernel: per-context keep bitmap (N bits, zero-padded to whole segments of
// kScanSegmentSamples) + number of kept samples before each segment.
constexpr uint32_t kScanSegmentSamples = 16384u;
struct ScanArgs {
    const uint64_t *keep_words;  // device; n_segments * (kScanSegmentSamples / 64) words
    const uint32_t *seg_rank;    // device; n_segments + 1 entries
    uint32_t max_seg_count;      // most kept samples in any one segment
    uint32_t max_super_count;    // most kept samples in any aligned triple of segments (picks the three-segment gather kernel)
};
hipError_t launch_gt_scan(const EmitArgs &a, const ScanArgs &sc, int num_cus, hipStream_t stream);

// kept subsets on short records (N <= 4096): output-driven pick through the kept list, no compaction (gt_pick.hip)
bool gt_pick_applicable(const EmitArgs &a);
hipError_t launch_gt_pick(const EmitArgs &a, int num_cus, hipStream_t stream);

// Deterministic synthetic records (SURVEY.md §8d counter-based generator).
hipError_t launch_synth_records(uint8_t *dst, uint64_t record_stride, uint32_t sample_count,
                                uint64_t first_variant, uint32_t n_variants, uint64_t seed,
                                bool dirty_pad, int num_cus, hipStream_t stream);

}  // namespace pgenhip
