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
};

// General row-tiled kernel: any alignment, any strides, list gather for kept subsets.
hipError_t launch_gt_rows(const EmitArgs &a, int num_cus, hipStream_t stream);

// Deterministic synthetic records (SURVEY.md §8d counter-based generator).
hipError_t launch_synth_records(uint8_t *dst, uint64_t record_stride, uint32_t sample_count,
                                uint64_t first_variant, uint32_t n_variants, uint64_t seed,
                                bool dirty_pad, int num_cus, hipStream_t stream);

}  // namespace pgenhip
