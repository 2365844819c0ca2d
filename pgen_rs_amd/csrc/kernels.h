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
    const uint64_t *record_off;   // device or nullptr; when set, row j's record starts at records + record_off[j] (byte offsets:
                                  // uncompressed records of a variable-width .pgen) and variant_idx / record_stride are not used
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
    uint64_t *work_counters;      // device scratch owned by the ctx, one block per launch in flight: 8 x 128-B-spaced work-queue heads + a block-exit counter at +1024 B; zero between launches (the last block out re-zeroes them)
};

// Work-queue kernels: up to kMaxQueueRanges contiguous item ranges per launch, one 128-B-spaced head word each, and the block-exit
// counter behind them (a launch's counter block: (kMaxQueueRanges + 1) * 128 bytes; capi.hip keeps a ring of them)
constexpr uint32_t kMaxQueueRanges = 64;

// Launch-shape knobs, resolved ONCE per context: pgenhip_create sets the measured defaults below and
// pgenhip_tune overrides one (tests force small grids to exercise ring reuse; A/B probes).  Nothing on the
// launch path reads the process environment.
struct Tuning {
    int wide_blocks_per_cu = 0;    // stream kernel: 0 = what the occupancy API says is resident
    int wide_ranges = 0;           // stream kernel: work-queue ranges = write fronts of a launch (power of two <= 64); 0 = by shape (8 for rows of several spans, else 2)
    int flat_blocks_per_cu = 64;   // flat kernel: grid cap
    int scan_blocks_per_cu = 0;    // segment kernel: 0 = the measured rule (2 from ~0.6 % kept, else what the occupancy API says)
    int pick_line_seams = 1;       // pick family, full lines of dense records: interiors + seams in one kernel (0 = round 2's row-by-row flush)
    int pick_batch_bytes = 32768;  // short-record pick kernel: text per batch (one store drain per batch)
    int scan_xcd_map = 1;          // segment kernels: all blocks of a row group on one XCD (seam lines merge in one L2); 0 = plain map
    int scan_chunk_rows = 0;       // two-pass path: rows per chunk (0 = as many as the 64-MiB compact scratch holds; tests force small chunks)
    int scan_two_pass = 1;         // sparse keeps on long records: compact pass + all-samples pass (0 = single-pass segment kernel)
    int rowpick_blocks_per_cu = 0; // row-owner kernel: cap on resident blocks per CU (0 = what the occupancy API says)
    int scan_rowpick = 1;          // (1: row-owner compact pass + all-samples pass; 2: row-owner single pass; 0: segment compact pass)
         // sparse keeps on long records, many rows: one wave per row, one pass (0 = the segment kernels / two passes)
    int flush_unroll = 2;          // segment / row-owner kernels: 16-byte chunks per lane and step of the text flush (1 or 2)
    int scan_four_picks = 1;       // segment / row-owner kernels' text flush: four picks per chunk + the fifth text from the next lane (0 = round 2's five picks)
    int align_stores = 1;          // subset kernels: lanes <-> chunks shifted so that every store instruction covers whole 128-byte lines (0 = from the first whole chunk)
    int runs_rows = 0;             // RUNS mode of the stream kernel: rows per work item (0 = as many as one wide load / one span holds)
};

// rows are gathered (variant list or byte offsets): the HAS_VIDX instantiations
__host__ __device__ inline bool gathered(const EmitArgs &a) { return a.variant_idx != nullptr || a.record_off != nullptr; }

// General row-tiled kernel: any alignment, any strides, list gather for kept subsets.
hipError_t launch_gt_rows(const EmitArgs &a, int num_cus, hipStream_t stream);

// Dense all-samples stream kernel (K = N, out_stride == 4N+1): every lane owns one aligned
// 16-byte chunk of the whole launch's output stream.
bool gt_flat_applicable(const EmitArgs &a);
hipError_t launch_gt_flat(const EmitArgs &a, const Tuning &t, int num_cus, hipStream_t stream);

// Same contract as the flat kernel, but records are staged with wide (16 B/lane) loads through LDS by a loader
// wave and storer waves issue 16 coalesced 1-KiB stores per work item (rows >= 4 KiB of text; needs work_counters).
bool gt_wide_applicable(const EmitArgs &a);
bool gt_wide_lines_applicable(const EmitArgs &a);  // full-line mode (line_off/prefix_off set) through the stream kernel
hipError_t launch_gt_wide(const EmitArgs &a, const Tuning &t, int num_cus, hipStream_t stream);
// RUNS mode of the same kernel for SHORT rows (8 <= N <= ~2000, dense records and dense text, no gather): a work item
// is a run of consecutive rows — one wide load of their contiguous record bytes, their text as one contiguous run.
bool gt_runs_applicable(const EmitArgs &a);
bool gt_runs_preferred(const EmitArgs &a);  // what AUTO uses
hipError_t launch_gt_runs(const EmitArgs &a, const Tuning &t, int num_cus, hipStream_t stream);
// Runs of FULL LINES (line_off / prefix_off set) on short rows, all samples kept, dense records: prefixes, GT text and '\n' of a
// run of lines assembled in the storers' LDS stages and stored as whole 128-B lines (gt_wide.hip, gt_lineruns_kernel).
bool gt_lineruns_applicable(const EmitArgs &a);
uint32_t gt_lineruns_rows(const EmitArgs &a);  // lines per work item for this shape (records of one wide load, prefix bytes of the slab, one span of text)
hipError_t launch_gt_lineruns(const EmitArgs &a, const Tuning &t, int num_cus, hipStream_t stream);

// Kept-subset segment kernels: number of kept samples before each segment of kScanSegmentSamples samples.
constexpr uint32_t kScanSegmentSamples = 16384u;
struct ScanArgs {
    const uint32_t *seg_rank;    // device; n_segments + 1 entries
    uint32_t max_seg_count;      // most kept samples in any one segment
    uint32_t align_stores;       // Tuning::align_stores
};
// The compact pass takes segments with at most this many kept samples (a quarter of a segment; the two-pass band is <= 4.5 %
// kept overall, so only a very clustered list has more in one segment: capi.hip then stays with the single-pass kernel)
constexpr uint32_t kCompactMaxSegCodes = 4096u;
// compact = true: write each row's COMPACT record (the K kept codes packed like a mode-0x02 record of K samples) to a.out + j * a.out_stride
// instead of text: the first pass of the two-pass path for sparse keeps (capi.hip)
hipError_t launch_gt_scan(const EmitArgs &a, const ScanArgs &sc, const Tuning &t, int num_cus, hipStream_t stream, bool compact = false);

// Sparse kept subsets on long records, one wave per ROW (gt_rowpick.hip): the whole kept list as the LDS table, the row's compact
// record assembled in LDS segment by segment, its text written in one go.  Any strides / gathers / full lines.
constexpr uint32_t kRowPickMaxKept = 16384u;
bool gt_rowpick_applicable(const EmitArgs &a, int num_cus);   // what AUTO requires (incl. enough rows for every resident wave)
// compact = true: the row's COMPACT record (ceil(K / 4) bytes at a.out + row * a.out_stride) instead of its text: first pass of the two-pass path
uint32_t gt_rowpick_resident_waves(const EmitArgs &a, const Tuning &t, int num_cus, bool compact);   // rows of one round (0: not applicable)
hipError_t launch_gt_rowpick(const EmitArgs &a, const ScanArgs &sc, const Tuning &t, int num_cus, hipStream_t stream, bool compact = false);

// kept subsets on short records (N <= 4096): output-driven pick through the kept list, no compaction (gt_pick.hip)
bool gt_pick_applicable(const EmitArgs &a);
hipError_t launch_gt_pick(const EmitArgs &a, const Tuning &t, int num_cus, hipStream_t stream);

// Deterministic synthetic records (SURVEY.md §8d counter-based generator).
hipError_t launch_synth_records(uint8_t *dst, uint64_t record_stride, uint32_t sample_count,
                                uint64_t first_variant, uint32_t n_variants, uint64_t seed,
                                bool dirty_pad, int num_cus, hipStream_t stream);

// "hwe" value distribution (per-variant allele frequency, Hardy-Weinberg genotype proportions, 0.1 % missing)
hipError_t launch_synth_records_hwe(uint8_t *dst, uint64_t record_stride, uint32_t sample_count, uint64_t first_variant,
                                    uint32_t n_variants, uint64_t seed, int num_cus, hipStream_t stream);

}  // namespace pgenhip
