// capi.hip — the extern "C" boundary of libpgen_hip.so (declared in include/pgen_hip.h).
// Host-side argument checking, context/stream ownership and kernel selection; no CPU compute
// path exists here: every decode goes through a gfx950 kernel or fails with a status code.
#include "../../include/pgen_hip.h"

#include <algorithm>
#include <hip/hip_runtime.h>

#include <new>
#include <string>
#include <vector>

#include "host_pure.h"
#include "kernels.h"

using namespace pgenhip;

struct pgenhip_ctx {
    int device = 0;
    int num_cus = 256;
    uint32_t sample_count = 0;
    uint32_t record_size = 0;
    uint32_t kept_count = 0;
    bool subset = false;
    bool identity = false;             // a kept list that names every sample: AUTO takes the all-samples kernels
    uint32_t *d_kept = nullptr;
    uint32_t *d_seg_rank = nullptr;    // segment kernels: kept samples before each segment
    uint32_t max_seg_count = 0;        // segment kernels: most kept samples in one segment
    uint8_t *d_compact = nullptr;      // two-pass path for sparse keeps on long records: compact records of one chunk of rows,
    size_t compact_bytes = 0;          // one slice of compact_bytes per launch in flight (the slice follows the launch's counter block)
    uint32_t launch_slot = 0;          // ring slot of the launch being queued (claim_counters)
    uint64_t *d_work = nullptr;        // stream kernel: ring of PGENHIP_LAUNCHES_IN_FLIGHT counter blocks (8 work-queue heads 128 B apart + an exit counter)
    uint32_t launch_seq = 0;           // next counter block of the ring
    bool work_dirty = false;           // a launch failed: counters may be non-zero, re-zero the ring before the next launch
    Tuning tune;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr;
    hipEvent_t ev_stop = nullptr;
};

namespace {

int fail(int status, const char *what) { return pgenhip::set_detail(status, what); }

int fail_hip(hipError_t e, const char *where)
{
    pgenhip::set_detail(0, (std::string(where) + ": " + hipGetErrorString(e)).c_str());
    (void)hipGetLastError();  // clear the sticky error
    return (e == hipErrorOutOfMemory) ? PGENHIP_ERR_OOM : PGENHIP_ERR_HIP;
}

#define HIP_TRY(expr)                                      \
    do {                                                   \
        hipError_t e__ = (expr);                           \
        if (e__ != hipSuccess) return fail_hip(e__, #expr); \
    } while (0)

int bind(const pgenhip_ctx *ctx)
{
    if (!ctx) return fail(PGENHIP_ERR_BAD_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    return PGENHIP_OK;
}

// Sparse keeps on long records (BASELINE config 5: 1 % of 500 000 samples) go through TWO passes: the segment kernel writes each
// row's COMPACT record (the K kept codes packed like a mode-0x02 record of K samples: 41-byte pieces instead of 656-byte pieces of
// text per row and segment) and the all-samples kernels turn those into text with whole-line stores.  Worth it while the text is
// small against the records.  Measured band (N = 500 000, profiles/r02_kernel_sweeps.md): 0.5 % kept -2 %, 0.65 % +4 %, 1 % +15 %,
// 2 % +11 %, 4 % +7 %, 5 % level, 10 % -4 %  ->  0.6 % .. 4.5 % kept.
// (Short records, N <= 4 096, were tried through the same two passes with a COMPACT instantiation of the short-record pick
// kernel: 0.42-0.50 of roofline against the single pass's 0.52-0.63 at every density — the compaction costs more than the text
// it saves there; profiles/r02_kernel_sweeps.md.)
bool two_pass_shape(uint32_t sample_count, uint32_t kept_count)
{
    return sample_count > 4096u && kept_count >= 8u && (uint64_t)kept_count * 170ull >= (uint64_t)sample_count &&
           (uint64_t)kept_count * 22ull <= (uint64_t)sample_count;
}

constexpr size_t kWorkBlockBytes = (kMaxQueueRanges + 1u) * 128u;  // the heads 128 B apart + the exit counter
constexpr size_t kWorkBlockWords = kWorkBlockBytes / sizeof(uint64_t);
// Two-pass path: every launch in flight has its own slice of compact-record scratch, like its own counter block (launches of one
// ctx on different streams may overlap: include/pgen_hip.h, "Streams").  A slice holds one chunk of rows between the two passes;
// a chunk stays in the 256-MiB Infinity Cache, so a smaller one costs only its two extra kernel launches.
constexpr size_t kCompactSliceBytes = 32u << 20;

}  // namespace

extern "C" {

int pgenhip_device_count(int *count)
{
    if (!count) return fail(PGENHIP_ERR_BAD_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail_hip(e, "hipGetDeviceCount");
    }
    *count = n;
    return PGENHIP_OK;
}

int pgenhip_create(pgenhip_ctx **out, int device_ordinal, uint32_t sample_count,
                   const uint32_t *kept_idx, uint32_t kept_count, uint32_t flags)
{
    if (!out) return fail(PGENHIP_ERR_BAD_ARG, "ctx out-pointer is NULL");
    *out = nullptr;
    if (flags & ~PGENHIP_CREATE_KEEP_LIST) return fail(PGENHIP_ERR_BAD_ARG, "unknown create flag");
    const bool subset = (flags & PGENHIP_CREATE_KEEP_LIST) != 0u || kept_idx != nullptr;
    if (subset && kept_count && !kept_idx) return fail(PGENHIP_ERR_BAD_ARG, "kept_count > 0 with a NULL kept_idx");
    if (sample_count > 0x7FFFFFFFu) return fail(PGENHIP_ERR_TOO_LARGE, "sample_count > 2^31-1");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(PGENHIP_ERR_NO_DEVICE, "hipGetDeviceCount found no device");
    }
    if (device_ordinal < 0 || device_ordinal >= n) return fail(PGENHIP_ERR_NO_DEVICE, "device ordinal out of range");
    if (subset) {
        for (uint32_t k = 0; k < kept_count; k++) {
            if (kept_idx[k] >= sample_count) return fail(PGENHIP_ERR_INDEX_RANGE, "kept_idx entry >= sample_count");
            if (k && kept_idx[k] <= kept_idx[k - 1]) return fail(PGENHIP_ERR_BAD_ARG, "kept_idx not strictly ascending");
        }
    }
    pgenhip_ctx *ctx = new (std::nothrow) pgenhip_ctx();
    if (!ctx) return fail(PGENHIP_ERR_OOM, "ctx");
    ctx->device = device_ordinal;
    ctx->sample_count = sample_count;
    ctx->record_size = pgenhip_variant_record_size(sample_count);
    ctx->subset = subset;
    ctx->identity = subset && kept_count == sample_count;  // strictly ascending and complete = 0..N-1
    ctx->kept_count = subset ? kept_count : sample_count;

    int rc = PGENHIP_OK;
    do {
        if ((e = hipSetDevice(device_ordinal)) != hipSuccess) { rc = fail_hip(e, "hipSetDevice"); break; }
        hipDeviceProp_t prop;
        if ((e = hipGetDeviceProperties(&prop, device_ordinal)) != hipSuccess) { rc = fail_hip(e, "hipGetDeviceProperties"); break; }
        ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) { rc = fail_hip(e, "hipStreamCreate"); break; }
        ctx->stream = ctx->own_stream;
        if ((e = hipEventCreate(&ctx->ev_start)) != hipSuccess) { rc = fail_hip(e, "hipEventCreate"); break; }
        if ((e = hipEventCreate(&ctx->ev_stop)) != hipSuccess) { rc = fail_hip(e, "hipEventCreate"); break; }
        if ((e = hipMalloc(reinterpret_cast<void **>(&ctx->d_work), PGENHIP_LAUNCHES_IN_FLIGHT * kWorkBlockBytes)) != hipSuccess) { rc = fail_hip(e, "hipMalloc(work counters)"); break; }
        if ((e = hipMemset(ctx->d_work, 0, PGENHIP_LAUNCHES_IN_FLIGHT * kWorkBlockBytes)) != hipSuccess) { rc = fail_hip(e, "hipMemset(work counters)"); break; }  // the kernels leave them zero
        if (ctx->subset) {
            size_t bytes = (size_t)(kept_count ? kept_count : 1u) * sizeof(uint32_t);
            if ((e = hipMalloc(reinterpret_cast<void **>(&ctx->d_kept), bytes)) != hipSuccess) { rc = fail_hip(e, "hipMalloc(kept_idx)"); break; }
            if (kept_count) {
                if ((e = hipMemcpy(ctx->d_kept, kept_idx, (size_t)kept_count * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess) { rc = fail_hip(e, "hipMemcpy(kept_idx)"); break; }
            }
            // kept samples before each 16 384-sample segment (what the segment kernels slice the list by)
            const uint32_t n_seg = (sample_count + kScanSegmentSamples - 1u) / kScanSegmentSamples;
            const uint32_t n_seg_eff = n_seg ? n_seg : 1u;
            std::vector<uint32_t> seg_rank((size_t)n_seg_eff + 1u, 0u);
            for (uint32_t k = 0; k < kept_count; k++) seg_rank[(size_t)(kept_idx[k] / kScanSegmentSamples) + 1u]++;
            for (uint32_t g = 0; g < n_seg_eff; g++) {
                ctx->max_seg_count = std::max(ctx->max_seg_count, seg_rank[g + 1u]);
                seg_rank[g + 1u] += seg_rank[g];
            }
            if ((e = hipMalloc(reinterpret_cast<void **>(&ctx->d_seg_rank), seg_rank.size() * sizeof(uint32_t))) != hipSuccess) { rc = fail_hip(e, "hipMalloc(segment ranks)"); break; }
            if ((e = hipMemcpy(ctx->d_seg_rank, seg_rank.data(), seg_rank.size() * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess) { rc = fail_hip(e, "hipMemcpy(segment ranks)"); break; }
            // two-pass path (sparse keeps on long records): scratch for the compact records of one chunk of rows per launch in flight
            // (config 5's per-GPU shard, 125 000 rows x 1 250 bytes, is five chunks; a chunk stays in the 256-MiB Infinity Cache
            // between the two passes); allocated here so that no launch ever allocates
            if (two_pass_shape(sample_count, kept_count)) {
                ctx->compact_bytes = kCompactSliceBytes;
                if ((e = hipMalloc(reinterpret_cast<void **>(&ctx->d_compact), PGENHIP_LAUNCHES_IN_FLIGHT * ctx->compact_bytes)) != hipSuccess) { rc = fail_hip(e, "hipMalloc(compact records)"); break; }
            }
        }
    } while (0);
    if (rc != PGENHIP_OK) {
        const std::string keep = pgenhip_last_error_detail();
        pgenhip_destroy(ctx);
        pgenhip::set_detail(0, keep.c_str());
        return rc;
    }
    *out = ctx;
    return PGENHIP_OK;
}

int pgenhip_destroy(pgenhip_ctx *ctx)
{
    if (!ctx) return PGENHIP_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    if (ctx->d_kept) (void)hipFree(ctx->d_kept);
    if (ctx->d_work) (void)hipFree(ctx->d_work);
    if (ctx->d_seg_rank) (void)hipFree(ctx->d_seg_rank);
    if (ctx->d_compact) (void)hipFree(ctx->d_compact);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return PGENHIP_OK;
}

int pgenhip_set_stream(pgenhip_ctx *ctx, void *hip_stream)
{
    if (!ctx) return fail(PGENHIP_ERR_BAD_ARG, "ctx is NULL");
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);  // nullptr = the default (null) stream
    return PGENHIP_OK;
}

int pgenhip_reset_stream(pgenhip_ctx *ctx)
{
    if (!ctx) return fail(PGENHIP_ERR_BAD_ARG, "ctx is NULL");
    ctx->stream = ctx->own_stream;
    return PGENHIP_OK;
}

uint32_t pgenhip_sample_count(const pgenhip_ctx *ctx) { return ctx ? ctx->sample_count : 0u; }
uint32_t pgenhip_kept_count(const pgenhip_ctx *ctx) { return ctx ? ctx->kept_count : 0u; }
uint64_t pgenhip_gt_row_bytes(const pgenhip_ctx *ctx) { return ctx ? 4ull * ctx->kept_count + 1ull : 0ull; }

static int fill_args(pgenhip_ctx *ctx, EmitArgs &a, const void *d_records, uint64_t record_stride,
                     const uint32_t *d_variant_idx, uint32_t n_variants, void *d_out)
{
    if (n_variants && (!d_out)) return fail(PGENHIP_ERR_BAD_ARG, "d_out is NULL");
    if (n_variants && ctx->record_size && !d_records) return fail(PGENHIP_ERR_BAD_ARG, "d_records is NULL");
    if (n_variants > 1 && !d_variant_idx && record_stride < ctx->record_size)
        return fail(PGENHIP_ERR_BAD_ARG, "record_stride < record size");
    a.records = static_cast<const uint8_t *>(d_records);
    a.record_stride = record_stride;
    a.variant_idx = d_variant_idx;
    a.record_off = nullptr;
    a.n_variants = n_variants;
    a.sample_count = ctx->sample_count;
    a.record_size = ctx->record_size;
    a.kept_idx = ctx->subset ? ctx->d_kept : nullptr;
    a.kept_count = ctx->kept_count;
    a.out = static_cast<uint8_t *>(d_out);
    a.out_stride = 0;
    a.prefix_blob = nullptr;
    a.prefix_off = nullptr;
    a.line_off = nullptr;
    a.max_line_bytes = 0;
    a.work_counters = nullptr;
    return PGENHIP_OK;
}

// Every launch gets its own counter block from the ring, so launches of one ctx queued on different streams
// never share work-queue heads (include/pgen_hip.h, "Streams").  After a failed launch the ring is re-zeroed
// in stream order first.
static int claim_counters(pgenhip_ctx *ctx, EmitArgs &a)
{
    if (ctx->work_dirty) {
        HIP_TRY(hipMemsetAsync(ctx->d_work, 0, PGENHIP_LAUNCHES_IN_FLIGHT * kWorkBlockBytes, ctx->stream));
        ctx->work_dirty = false;
    }
    ctx->launch_slot = ctx->launch_seq++ % PGENHIP_LAUNCHES_IN_FLIGHT;
    a.work_counters = ctx->d_work + (size_t)ctx->launch_slot * kWorkBlockWords;
    return PGENHIP_OK;
}

#define LAUNCH_TRY(expr)                           \
    do {                                           \
        hipError_t e__ = (expr);                   \
        if (e__ != hipSuccess) {                   \
            ctx->work_dirty = true;                \
            return fail_hip(e__, #expr);           \
        }                                          \
    } while (0)

// measured crossover (profiles/r01_kernel_sweeps.md: N = 500 000, 0.2 % kept list gather 1.13 ms vs 1.49 ms,
// 0.4 % kept 1.51 vs 1.47; round 2, with the faster segment kernel: 0.25 % 1.33 vs 1.51, 0.33 % 1.48 vs 1.52, 0.4 % 1.60 vs 1.57): below ~1/280 kept on long records the list gather touches only the kept
// samples' lines and wins; everywhere else the segment kernels do (they read each record once, wide)
static bool very_sparse(const pgenhip_ctx *ctx)
{
    return ctx->sample_count >= 65536u && (uint64_t)ctx->kept_count * 280ull <= ctx->sample_count;
}

// Kept subsets on long records through the row-owner kernel WRITING TEXT (gt_rowpick.hip, one pass): the whole kept list fits its LDS
// table (K <= 16 384) and the launch has enough rows for every resident wave.  Where it is ahead of the segment kernel / the two passes
// (profiles/r03_kernel_sweeps.md §7, fraction of roofline against the previous dispatch):
//   * records of one full segment and a thin second one (16 384 < N < 24 576): the segment kernel's blocks are unbalanced there —
//     N = 16 385 with 10 / 50 % kept 0.57 / 0.55 against 0.39 / 0.38, N = 20 000 with 10 / 30 / 80 % 0.60 / 0.57 / 0.51 against 0.50 / 0.43 / 0.43;
//   * 2 % .. 20 % kept on longer records: N = 30 000 5 % 0.62 / 0.53, N = 60 000 10 % 0.60 / 0.58, N = 100 000 2 / 5 % 0.65 / 0.62 against 0.62 / 0.56,
//     N = 500 000 3.2 % 0.62 / 0.58 (level from ~20 %: N = 60 000 25 % 0.565 / 0.571; behind at 50 %).
// Below 2 % the two passes keep the sparse band (BASELINE configs[4], 1 % of 500 000: one pass reads 86 % and writes 14 % of its bytes
// everywhere at once and runs at the copy ceiling, 3-6 % behind; N = 200 000 1 %: 0.635 against 0.658).
// PGENHIP_KNOB_SCAN_ROWPICK = 2 takes the one pass wherever it is applicable in the two-pass band too (A/B), -1 nowhere.
static bool rowpick_shape(const pgenhip_ctx *ctx, const EmitArgs &a)
{
    if (ctx->tune.scan_rowpick == 0 || ctx->sample_count <= kScanSegmentSamples || very_sparse(ctx) || !gt_rowpick_applicable(a, ctx->num_cus)) return false;
    const uint64_t N = ctx->sample_count, K = ctx->kept_count;
    if (ctx->tune.scan_rowpick == 2 && two_pass_shape(ctx->sample_count, ctx->kept_count)) return true;
    if (N < 24576ull) return true;
    return K * 50ull >= N && K * 5ull <= N;
}

// AUTO for all samples kept, GT segments at a.out + j * a.out_stride
static int dispatch_all_samples(pgenhip_ctx *ctx, const EmitArgs &a)
{
    const Tuning &t = ctx->tune;
    if (gt_runs_preferred(a))
        // short rows, dense records (8 <= N <= 1915): runs of rows as one work item, text staged through LDS in 4-KiB groups
        // so that every 128-B line leaves whole: 0.68-0.71 of roofline from N = 100 to 1500 where the flat kernel had
        // 0.42-0.56, the pick kernel 0.31-0.62 and the row-item stream kernel 0.51-0.65 (profiles/r02_kernel_sweeps.md)
        LAUNCH_TRY(launch_gt_runs(a, t, ctx->num_cus, ctx->stream));
    else if (gt_pick_applicable(a) && a.sample_count < 2000u)
        // short rows that are gathered or padded (no contiguous runs): batches of rows through gt_pick.hip with the identity for a
        // table, several rows per load instruction.  Gathered rows, fraction of roofline, pick / flat / row-item stream kernel:
        // N = 64 0.46 / 0.28 / -, 300 0.49 / 0.31 / -, 1 399 0.58 / 0.40 / 0.52, 1 500 0.58 / 0.40 / 0.54, 2 504 0.59 / 0.41 / 0.68
        LAUNCH_TRY(launch_gt_pick(a, t, ctx->num_cus, ctx->stream));
    else if (gt_wide_applicable(a))
        LAUNCH_TRY(launch_gt_wide(a, t, ctx->num_cus, ctx->stream));
    else if (gt_flat_applicable(a))
        LAUNCH_TRY(launch_gt_flat(a, t, ctx->num_cus, ctx->stream));
    else
        LAUNCH_TRY(launch_gt_rows(a, ctx->num_cus, ctx->stream));
    return PGENHIP_OK;
}

// AUTO for all samples kept, full lines (a.line_off / a.prefix_off set)
static int dispatch_all_samples_lines(pgenhip_ctx *ctx, const EmitArgs &a)
{
    const Tuning &t = ctx->tune;
    if (gt_wide_lines_applicable(a) && a.sample_count >= 1400u) {
        // long rows: the work-queue stream kernel writes the GT segments in place behind their prefixes (+ a small prefix copy).
        // From N = 1 400: at N = 1 024 / 1 200 it runs at 0.45 / 0.49 of roofline against 0.48-0.51 for the two kernels below, at
        // N = 1 500 / 1 900 at 0.56 / 0.60 against 0.49-0.55 (profiles/r02_kernel_sweeps.md)
        LAUNCH_TRY(launch_gt_wide(a, t, ctx->num_cus, ctx->stream));
    } else if (gt_lineruns_applicable(a) && gt_lineruns_rows(a) >= 7u && a.sample_count < 1000u) {
        // short rows, dense records: runs of whole lines (prefix + GT + '\n') assembled in LDS and stored as whole 128-B lines.
        // Ahead of the pick family's full-line kernel while a run holds seven lines or more (prefixes up to ~90 bytes) and N < 1 000:
        // 30-byte prefixes, N = 100 / 300 / 500: 0.41 / 0.51 / 0.53 of roofline against 0.36 / 0.44 / 0.49; with 166-byte prefixes a run
        // is three or four lines and the pick family is ahead at every N (0.37-0.52 against 0.14-0.50); from N = 1 000 it is level or
        // ahead with short prefixes too (profiles/r03_logs/lines_sweep_after.log)
        LAUNCH_TRY(launch_gt_lineruns(a, t, ctx->num_cus, ctx->stream));
    } else if (gt_pick_applicable(a)) {
        // the pick family (identity for a table): rows' interiors + batched seams (gt_pick_lines_kernel); gathered / padded records: row by row
        LAUNCH_TRY(launch_gt_pick(a, t, ctx->num_cus, ctx->stream));   // (writes the prefixes too)
    } else {
        LAUNCH_TRY(launch_gt_rows(a, ctx->num_cus, ctx->stream));
    }
    return PGENHIP_OK;
}

// Two passes for sparse keeps on long records, chunk by chunk of as many rows as the compact scratch holds: (1) the row-owner kernel
// (chunks of many rows; else the segment kernel) compacts each row's kept codes into a K-sample record, (2) the all-samples
// dispatch above turns those records into text.
static bool two_pass(const pgenhip_ctx *ctx, const EmitArgs &a)
{
    // (full lines: only where the second pass is the stream kernel's LINES mode, K >= 1 024 — below that it would flush row by row)
    return ctx->tune.scan_two_pass != 0 && ctx->d_compact != nullptr && a.kept_idx != nullptr && a.record_size >= 16u &&
           ctx->max_seg_count <= kCompactMaxSegCodes &&
           (a.line_off != nullptr ? a.kept_count >= 1024u : (a.n_variants <= 1u || a.out_stride == 4ull * a.kept_count + 1ull));
}

static int dispatch_two_pass(pgenhip_ctx *ctx, const EmitArgs &a, const ScanArgs &sc)
{
    const uint32_t rc_bytes = (a.kept_count + 3u) / 4u;
    uint8_t *const scratch = ctx->d_compact + (size_t)ctx->launch_slot * ctx->compact_bytes;  // this launch's own slice
    uint64_t chunk_rows = std::max<uint64_t>(1ull, ctx->compact_bytes / rc_bytes);
    if (ctx->tune.scan_chunk_rows > 0) chunk_rows = std::min<uint64_t>(chunk_rows, (uint64_t)ctx->tune.scan_chunk_rows);
    // the row-owner compact pass deals rows to its resident waves round-robin: whole rounds per chunk
    bool row_owner = false;
    if (ctx->tune.scan_rowpick != 0) {
        EmitArgs probe = a;
        probe.n_variants = (uint32_t)std::min<uint64_t>(chunk_rows, a.n_variants);
        row_owner = gt_rowpick_applicable(probe, ctx->num_cus);
        const uint64_t round = gt_rowpick_resident_waves(probe, ctx->tune, ctx->num_cus, true);
        if (row_owner && round && chunk_rows > round && ctx->tune.scan_chunk_rows <= 0) chunk_rows -= chunk_rows % round;
    }
    for (uint64_t row0 = 0; row0 < a.n_variants; row0 += chunk_rows) {
        const uint32_t n = (uint32_t)std::min<uint64_t>(chunk_rows, (uint64_t)a.n_variants - row0);
        EmitArgs c = a;  // pass 1: this chunk's rows -> compact records
        if (a.record_off) c.record_off = a.record_off + row0;
        else if (a.variant_idx) c.variant_idx = a.variant_idx + row0;
        else c.records = a.records + row0 * a.record_stride;
        c.n_variants = n;
        c.out = scratch;
        c.out_stride = rc_bytes;
        c.prefix_blob = nullptr;
        c.prefix_off = nullptr;
        c.line_off = nullptr;
        if (row_owner && (uint64_t)n * 2ull >= chunk_rows)   // (a short last chunk: the segment kernel cuts it finer)
            LAUNCH_TRY(launch_gt_rowpick(c, sc, ctx->tune, ctx->num_cus, ctx->stream, true));   // a wave per row: whole compact records, wide stores
        else
            LAUNCH_TRY(launch_gt_scan(c, sc, ctx->tune, ctx->num_cus, ctx->stream, true));
        EmitArgs d = a;  // pass 2: a block of n mode-0x02 records of K samples, all of them kept
        d.records = scratch;
        d.record_stride = rc_bytes;
        d.variant_idx = nullptr;
        d.record_off = nullptr;
        d.n_variants = n;
        d.sample_count = a.kept_count;
        d.record_size = rc_bytes;
        d.kept_idx = nullptr;
        if (a.line_off) {
            d.prefix_off = a.prefix_off + row0;  // offsets are absolute: the same blob and output base
            d.line_off = a.line_off + row0;
            const int rc = dispatch_all_samples_lines(ctx, d);
            if (rc) return rc;
        } else {
            d.out = a.out + row0 * a.out_stride;
            const int rc = dispatch_all_samples(ctx, d);
            if (rc) return rc;
        }
    }
    return PGENHIP_OK;
}

static int decode_emit_core(pgenhip_ctx *ctx, const void *d_records, uint64_t record_stride, const uint32_t *d_variant_idx,
                            const uint64_t *d_record_off, uint32_t n_variants, void *d_out, uint64_t out_stride, uint32_t flags)
{
    int rc = bind(ctx);
    if (rc) return rc;
    EmitArgs a;
    rc = fill_args(ctx, a, d_records, d_record_off ? (uint64_t)ctx->record_size : record_stride, d_variant_idx, n_variants, d_out);
    if (rc) return rc;
    a.record_off = d_record_off;
    if (n_variants > 1 && out_stride < 4ull * ctx->kept_count + 1ull)
        return fail(PGENHIP_ERR_BAD_ARG, "out_stride < 4K+1");
    if (flags & ~PGENHIP_KERNEL_MASK) return fail(PGENHIP_ERR_BAD_ARG, "unknown decode_emit flag");
    a.out_stride = out_stride;
    if (n_variants == 0) return PGENHIP_OK;
    rc = claim_counters(ctx, a);
    if (rc) return rc;
    const Tuning &t = ctx->tune;
    const ScanArgs sc{ctx->d_seg_rank, ctx->max_seg_count, (uint32_t)ctx->tune.align_stores};

    switch (flags & PGENHIP_KERNEL_MASK) {
        case PGENHIP_KERNEL_AUTO:
            if (ctx->identity) a.kept_idx = nullptr;  // `--include-sam` that keeps everybody: same bytes, the all-samples kernels
            if (a.kept_idx == nullptr) return dispatch_all_samples(ctx, a);
            if (rowpick_shape(ctx, a)) {
                LAUNCH_TRY(launch_gt_rowpick(a, sc, t, ctx->num_cus, ctx->stream));
                return PGENHIP_OK;
            }
            if (two_pass(ctx, a) && !very_sparse(ctx))
                return dispatch_two_pass(ctx, a, sc);
            if (gt_pick_applicable(a))
                // short records (the 1000 Genomes shape with a sample filter): output-driven pick, any density
                LAUNCH_TRY(launch_gt_pick(a, t, ctx->num_cus, ctx->stream));
            else if (very_sparse(ctx) || ctx->record_size < 16u)
                LAUNCH_TRY(launch_gt_rows(a, ctx->num_cus, ctx->stream));
            else
                LAUNCH_TRY(launch_gt_scan(a, sc, t, ctx->num_cus, ctx->stream));
            return PGENHIP_OK;
        case PGENHIP_KERNEL_ROWS:
            LAUNCH_TRY(launch_gt_rows(a, ctx->num_cus, ctx->stream));
            return PGENHIP_OK;
        case PGENHIP_KERNEL_SCAN:
            if (!ctx->subset) return fail(PGENHIP_ERR_BAD_ARG, "segment kernels need a kept-sample list");
            if (ctx->record_size < 16u) return fail(PGENHIP_ERR_BAD_ARG, "segment kernels need N >= 61 (records of >= 16 bytes)");
            LAUNCH_TRY(launch_gt_scan(a, sc, t, ctx->num_cus, ctx->stream));
            return PGENHIP_OK;
        case PGENHIP_KERNEL_ROWPICK:
            if (!ctx->subset || ctx->record_size < 16u || ctx->kept_count < 1u || ctx->kept_count > kRowPickMaxKept)
                return fail(PGENHIP_ERR_BAD_ARG, "row-owner kernel needs a kept-sample list of 1 .. 16384 samples and N >= 61");
            LAUNCH_TRY(launch_gt_rowpick(a, sc, t, ctx->num_cus, ctx->stream));
            return PGENHIP_OK;
        case PGENHIP_KERNEL_PICK:
            if (!gt_pick_applicable(a)) return fail(PGENHIP_ERR_BAD_ARG, "pick kernel needs K >= 1, 61 <= N <= 4096 and out_stride == 4K+1");
            LAUNCH_TRY(launch_gt_pick(a, t, ctx->num_cus, ctx->stream));
            return PGENHIP_OK;
        case PGENHIP_KERNEL_WIDE:
            if (!gt_wide_applicable(a)) return fail(PGENHIP_ERR_BAD_ARG, "wide kernel needs all samples kept, N >= 1024 and out_stride == 4N+1");
            LAUNCH_TRY(launch_gt_wide(a, t, ctx->num_cus, ctx->stream));
            return PGENHIP_OK;
        case PGENHIP_KERNEL_RUNS:
            if (!gt_runs_applicable(a)) return fail(PGENHIP_ERR_BAD_ARG, "runs kernel needs all samples kept, 8 <= N <= ~2000, dense records and text, no variant gather");
            LAUNCH_TRY(launch_gt_runs(a, t, ctx->num_cus, ctx->stream));
            return PGENHIP_OK;
        case PGENHIP_KERNEL_FLAT:
            if (!gt_flat_applicable(a)) return fail(PGENHIP_ERR_BAD_ARG, "flat kernel needs all samples kept and out_stride == 4N+1");
            LAUNCH_TRY(launch_gt_flat(a, t, ctx->num_cus, ctx->stream));
            return PGENHIP_OK;
        default:
            return fail(PGENHIP_ERR_BAD_ARG, "unknown kernel id");
    }
}

int pgenhip_decode_emit(pgenhip_ctx *ctx, const void *d_records, uint64_t record_stride,
                        const uint32_t *d_variant_idx, uint32_t n_variants,
                        void *d_out, uint64_t out_stride, uint32_t flags)
{
    return decode_emit_core(ctx, d_records, record_stride, d_variant_idx, nullptr, n_variants, d_out, out_stride, flags);
}

int pgenhip_decode_emit_at(pgenhip_ctx *ctx, const void *d_base, const uint64_t *d_record_off, uint32_t n_variants,
                           void *d_out, uint64_t out_stride, uint32_t flags)
{
    if (n_variants && !d_record_off) return fail(PGENHIP_ERR_BAD_ARG, "d_record_off is NULL");
    return decode_emit_core(ctx, d_base, 0, nullptr, d_record_off, n_variants, d_out, out_stride, flags);
}

int pgenhip_emit_lines(pgenhip_ctx *ctx, const void *d_records, uint64_t record_stride,
                       const uint32_t *d_variant_idx, uint32_t n_variants,
                       const void *d_prefix_blob, const uint64_t *d_prefix_off,
                       const uint64_t *d_line_off, uint64_t max_prefix_bytes,
                       void *d_out, uint32_t flags)
{
    int rc = bind(ctx);
    if (rc) return rc;
    EmitArgs a;
    rc = fill_args(ctx, a, d_records, record_stride, d_variant_idx, n_variants, d_out);
    if (rc) return rc;
    if (n_variants == 0) return PGENHIP_OK;
    if (!d_prefix_off || !d_line_off) return fail(PGENHIP_ERR_BAD_ARG, "offset arrays are NULL");
    if (max_prefix_bytes && !d_prefix_blob) return fail(PGENHIP_ERR_BAD_ARG, "d_prefix_blob is NULL");
    a.prefix_blob = static_cast<const uint8_t *>(d_prefix_blob);
    a.prefix_off = d_prefix_off;
    a.line_off = d_line_off;
    a.max_line_bytes = max_prefix_bytes + 4ull * ctx->kept_count + 1ull;
    rc = claim_counters(ctx, a);
    if (rc) return rc;
    const Tuning &t = ctx->tune;
    const ScanArgs sc{ctx->d_seg_rank, ctx->max_seg_count, (uint32_t)ctx->tune.align_stores};
    switch (flags) {
        case PGENHIP_KERNEL_AUTO:
            if (ctx->identity) a.kept_idx = nullptr;
            if (a.kept_idx == nullptr) return dispatch_all_samples_lines(ctx, a);
            if (rowpick_shape(ctx, a)) {
                LAUNCH_TRY(launch_gt_rowpick(a, sc, t, ctx->num_cus, ctx->stream));   // (writes the prefixes too)
                return PGENHIP_OK;
            }
            if (two_pass(ctx, a) && !very_sparse(ctx)) return dispatch_two_pass(ctx, a, sc);
            if (gt_lineruns_applicable(a) && gt_lineruns_rows(a) >= 7u && a.sample_count < 300u) {
                // kept subset on VERY short dense records: runs of whole lines through the line-run kernel, picks through its LDS kept table
                // (N = 100, 30-byte prefixes, 50 / 10 % kept: 0.28 / 0.13 of roofline against 0.23 / 0.10 for the pick family's full-line
                // kernel; from N = 300 that kernel is level or ahead — N = 500: 0.48 / 0.27 against 0.39 / 0.15 — and with long prefixes
                // always: profiles/r03_logs/lines_sweep_after.log)
                LAUNCH_TRY(launch_gt_lineruns(a, t, ctx->num_cus, ctx->stream));
            } else if (gt_pick_applicable(a)) {
                // kept subset on short records: the pick family (interiors + batched seams; gathered / padded records: row by row)
                LAUNCH_TRY(launch_gt_pick(a, t, ctx->num_cus, ctx->stream));   // (writes the prefixes too)
            } else if (ctx->record_size >= 16u && !very_sparse(ctx)) {
                // kept subset: the segment kernel writes each GT segment behind its prefix, the prefix kernel the rest
                LAUNCH_TRY(launch_gt_scan(a, sc, t, ctx->num_cus, ctx->stream));   // (writes the prefixes too)
            } else {
                LAUNCH_TRY(launch_gt_rows(a, ctx->num_cus, ctx->stream));
            }
            return PGENHIP_OK;
        case PGENHIP_KERNEL_ROWS:
            LAUNCH_TRY(launch_gt_rows(a, ctx->num_cus, ctx->stream));
            return PGENHIP_OK;
        case PGENHIP_KERNEL_WIDE:
            if (!gt_wide_lines_applicable(a)) return fail(PGENHIP_ERR_BAD_ARG, "PGENHIP_KERNEL_WIDE needs all samples kept and sample_count >= 1024");
            LAUNCH_TRY(launch_gt_wide(a, t, ctx->num_cus, ctx->stream));
            return PGENHIP_OK;
        case PGENHIP_KERNEL_SCAN:
            if (!ctx->subset || ctx->record_size < 16u) return fail(PGENHIP_ERR_BAD_ARG, "PGENHIP_KERNEL_SCAN needs a kept-sample list and N >= 61");
            LAUNCH_TRY(launch_gt_scan(a, sc, t, ctx->num_cus, ctx->stream));
            return PGENHIP_OK;
        case PGENHIP_KERNEL_RUNS:
            if (!gt_lineruns_applicable(a)) return fail(PGENHIP_ERR_BAD_ARG, "PGENHIP_KERNEL_RUNS (lines) needs dense records, >= 8 kept samples (of <= 4096 with a keep list) and two lines per item");
            LAUNCH_TRY(launch_gt_lineruns(a, t, ctx->num_cus, ctx->stream));
            return PGENHIP_OK;
        case PGENHIP_KERNEL_ROWPICK:
            if (!ctx->subset || ctx->record_size < 16u || ctx->kept_count < 1u || ctx->kept_count > kRowPickMaxKept)
                return fail(PGENHIP_ERR_BAD_ARG, "PGENHIP_KERNEL_ROWPICK needs a kept-sample list of 1 .. 16384 samples and N >= 61");
            LAUNCH_TRY(launch_gt_rowpick(a, sc, t, ctx->num_cus, ctx->stream));
            return PGENHIP_OK;
        case PGENHIP_KERNEL_PICK:
            if (!gt_pick_applicable(a)) return fail(PGENHIP_ERR_BAD_ARG, "PGENHIP_KERNEL_PICK needs K >= 1 and 61 <= N <= 4096");
            LAUNCH_TRY(launch_gt_pick(a, t, ctx->num_cus, ctx->stream));
            return PGENHIP_OK;
        default:
            return fail(PGENHIP_ERR_BAD_ARG, "pgenhip_emit_lines supports kernel flags AUTO, ROWS, WIDE, SCAN, PICK, RUNS and ROWPICK");
    }
}

int pgenhip_tune(pgenhip_ctx *ctx, uint32_t knob, int32_t value)
{
    if (!ctx) return fail(PGENHIP_ERR_BAD_ARG, "ctx is NULL");
    const Tuning d;  // the defaults
    Tuning &t = ctx->tune;
    switch (knob) {
        case PGENHIP_KNOB_WIDE_BLOCKS_PER_CU: t.wide_blocks_per_cu = value > 0 ? value : d.wide_blocks_per_cu; break;
        case PGENHIP_KNOB_WIDE_RANGES:
            if (value < 0 || value > (int32_t)kMaxQueueRanges || (value & (value - 1)) != 0) return fail(PGENHIP_ERR_BAD_ARG, "ranges must be a power of two up to 64");
            t.wide_ranges = value ? value : d.wide_ranges;
            break;
        case PGENHIP_KNOB_FLAT_BLOCKS_PER_CU: t.flat_blocks_per_cu = value > 0 ? value : d.flat_blocks_per_cu; break;
        case PGENHIP_KNOB_SCAN_BLOCKS_PER_CU: t.scan_blocks_per_cu = value > 0 ? value : d.scan_blocks_per_cu; break;
        case PGENHIP_KNOB_PICK_BATCH_BYTES: t.pick_batch_bytes = value > 0 ? value : d.pick_batch_bytes; break;
        case PGENHIP_KNOB_SCAN_XCD_MAP: t.scan_xcd_map = value < 0 ? 0 : 1; break;
        case PGENHIP_KNOB_SCAN_CHUNK_ROWS: t.scan_chunk_rows = value > 0 ? value : d.scan_chunk_rows; break;
        case PGENHIP_KNOB_SCAN_TWO_PASS: t.scan_two_pass = value < 0 ? 0 : 1; break;
        case PGENHIP_KNOB_ROWPICK_BLOCKS_PER_CU: t.rowpick_blocks_per_cu = value > 0 ? value : d.rowpick_blocks_per_cu; break;
        case PGENHIP_KNOB_SCAN_ROWPICK: t.scan_rowpick = value < 0 ? 0 : (value == 2 ? 2 : 1); break;
        case PGENHIP_KNOB_PICK_LINE_SEAMS: t.pick_line_seams = value < 0 ? 0 : 1; break;
        case PGENHIP_KNOB_FLUSH_UNROLL: t.flush_unroll = value == 1 || value == 2 || value == 4 ? value : d.flush_unroll; break;
        case PGENHIP_KNOB_SCAN_FOUR_PICKS: t.scan_four_picks = value < 0 ? 0 : 1; break;
        case PGENHIP_KNOB_ALIGN_STORES: t.align_stores = value < 0 ? 0 : 1; break;
        case PGENHIP_KNOB_RUNS_ROWS: t.runs_rows = value > 0 ? value : d.runs_rows; break;
        default: return fail(PGENHIP_ERR_BAD_ARG, "unknown knob");
    }
    return PGENHIP_OK;
}

int pgenhip_wait(pgenhip_ctx *ctx)
{
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PGENHIP_OK;
}

int pgenhip_timer_start(pgenhip_ctx *ctx)
{
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    return PGENHIP_OK;
}

int pgenhip_timer_stop(pgenhip_ctx *ctx, float *elapsed_ms)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!elapsed_ms) return fail(PGENHIP_ERR_BAD_ARG, "elapsed_ms is NULL");
    HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->ev_stop));
    HIP_TRY(hipEventElapsedTime(elapsed_ms, ctx->ev_start, ctx->ev_stop));
    return PGENHIP_OK;
}

int pgenhip_timer_mark(pgenhip_ctx *ctx)
{
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
    return PGENHIP_OK;
}

int pgenhip_timer_read(pgenhip_ctx *ctx, float *elapsed_ms)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!elapsed_ms) return fail(PGENHIP_ERR_BAD_ARG, "elapsed_ms is NULL");
    HIP_TRY(hipEventElapsedTime(elapsed_ms, ctx->ev_start, ctx->ev_stop));
    return PGENHIP_OK;
}

int pgenhip_device_malloc(pgenhip_ctx *ctx, void **d_ptr, size_t bytes)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_ptr) return fail(PGENHIP_ERR_BAD_ARG, "d_ptr is NULL");
    *d_ptr = nullptr;
    HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 1));
    return PGENHIP_OK;
}

int pgenhip_device_free(pgenhip_ctx *ctx, void *d_ptr)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (d_ptr) HIP_TRY(hipFree(d_ptr));
    return PGENHIP_OK;
}

int pgenhip_host_malloc_pinned(pgenhip_ctx *ctx, void **h_ptr, size_t bytes)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!h_ptr) return fail(PGENHIP_ERR_BAD_ARG, "h_ptr is NULL");
    *h_ptr = nullptr;
    HIP_TRY(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return PGENHIP_OK;
}

int pgenhip_host_free_pinned(pgenhip_ctx *ctx, void *h_ptr)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (h_ptr) HIP_TRY(hipHostFree(h_ptr));
    return PGENHIP_OK;
}

int pgenhip_memcpy_h2d(pgenhip_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (bytes && (!d_dst || !h_src)) return fail(PGENHIP_ERR_BAD_ARG, "NULL pointer");
    if (bytes) HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return PGENHIP_OK;
}

int pgenhip_memcpy_d2h(pgenhip_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (bytes && (!h_dst || !d_src)) return fail(PGENHIP_ERR_BAD_ARG, "NULL pointer");
    if (bytes) HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return PGENHIP_OK;
}

int pgenhip_synth_records(pgenhip_ctx *ctx, void *d_dst, uint64_t record_stride,
                          uint64_t first_variant, uint32_t n_variants, uint64_t seed, uint32_t flags)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (n_variants && ctx->record_size && !d_dst) return fail(PGENHIP_ERR_BAD_ARG, "d_dst is NULL");
    if (n_variants > 1 && record_stride < ctx->record_size) return fail(PGENHIP_ERR_BAD_ARG, "record_stride < record size");
    if (flags & ~(PGENHIP_SYNTH_DIRTY_PAD | PGENHIP_SYNTH_HWE)) return fail(PGENHIP_ERR_BAD_ARG, "unknown synth flag");
    if (flags & PGENHIP_SYNTH_HWE) {
        HIP_TRY(launch_synth_records_hwe(static_cast<uint8_t *>(d_dst), record_stride, ctx->sample_count, first_variant, n_variants, seed,
                                         ctx->num_cus, ctx->stream));
        return PGENHIP_OK;
    }
    HIP_TRY(launch_synth_records(static_cast<uint8_t *>(d_dst), record_stride, ctx->sample_count, first_variant,
                                 n_variants, seed, (flags & PGENHIP_SYNTH_DIRTY_PAD) != 0, ctx->num_cus, ctx->stream));
    return PGENHIP_OK;
}

}  // extern "C"
