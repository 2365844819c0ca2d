// gt_scan.hip — kept-subset kernels for records longer than one tile (gfx950 / MI355X).
//
// Replaces /root/reference/src/pfile.rs:165-190 when a sample filter is active
// (`--include-sam`, kept list from filter_metadata :312-335).  The kept list is the same for
// every variant, so the context holds it on the device together with the number of kept samples
// before each SEGMENT of 16 384 samples (4 KiB of every record).  A block owns one segment (or
// three, gather kernel) and stages ITS SLICE of the kept list once, as u16 offsets in LDS: that
// slice is the rank -> sample table.  Waves then walk rows: wide 16-B-per-lane loads of the
// segment's record bytes (next row's loads in flight while this row's text goes out), parked in
// the wave's LDS stage; the output-driven flush (flush_codes, gt_common.hip.h) gives every lane a
// 16-byte-ALIGNED chunk of the row's output bytes: five table reads + five staged-byte reads,
// text, funnel shift by the row's phase, one 16-B store — whole-line coalesced stores however
// irregular the mask is.  Segment and row edges (partial chunks, '\n') go out as one byte-store
// instruction per row piece.
//
// HBM traffic per row: the record once (R bytes, wide loads) + 4K+1 bytes of text; the kept list
// once per block, not once per row.
// (Round 1's first subset kernel — keep bitmap + popcount prefix in LDS, per-lane ctz compaction
// into a code ring — measured 5-10 % behind the table pick at every density
// (profiles/r01_kernel_sweeps.md) and was removed in round 2.)
#include "gt_common.hip.h"
#include "kernels.h"

namespace pgenhip {

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr uint32_t kSegSamples = kScanSegmentSamples;   // 16 384 samples per segment
constexpr uint32_t kTileSamples = 4096;                 // 64 lanes x 64 samples = 1 KiB of record
constexpr uint32_t kTilesPerSeg = kSegSamples / kTileSamples;

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <uint32_t RING>
__device__ __forceinline__ uint32_t ring_code(const uint8_t *ring, uint32_t rel)
{
    return ring[rel & (RING - 1u)];
}

// A lane's 16 record bytes as two 64-bit halves; a window that was pulled back by `tail_shift`
// bytes at the record tail is shifted into place.
__device__ __forceinline__ void window_halves(const v4u &w, uint32_t tail_shift, uint64_t &lo, uint64_t &hi)
{
    lo = (uint64_t)w.x | ((uint64_t)w.y << 32);
    hi = (uint64_t)w.z | ((uint64_t)w.w << 32);
    if (tail_shift != 0u) {
        const uint32_t sh8 = tail_shift * 8u;
        if (sh8 >= 128u) { lo = 0ull; hi = 0ull; }
        else if (sh8 >= 64u) { lo = hi >> (sh8 - 64u); hi = 0ull; }
        else { lo = (lo >> sh8) | (hi << (64u - sh8)); hi >>= sh8; }
    }
}

// Output-driven flush of row bytes [emitted, hi_emit) of one row.  Whole 16-byte-ALIGNED chunks:
// one lane per chunk reads five consecutive kept codes from the ring (rank r of the segment sits
// at ring position ring_base + r), expands them to text (src/pfile.rs:177-190), funnel-shifts by
// the row's phase and stores 16 B.  The up to 15 bytes before the first and after the last whole
// chunk (segment / row edges, shared with the neighbouring segment's block) go out as ONE
// byte-store instruction: lanes 0-15 take the head bytes, lanes 16-31 the tail bytes.
// All 64-bit arithmetic is wave-uniform (scalar unit); a lane only adds a 32-bit offset.
template <uint32_t RING>
__device__ __forceinline__ void flush_range(const uint8_t *ring, uint32_t ring_base, uint8_t *row_out, uint64_t emitted, uint64_t hi_emit,
                                            uint32_t seg_k0, uint32_t K, uint32_t lane)
{
    flush_codes([ring](uint32_t x) { return ring_code<RING>(ring, x); }, ring_base, row_out, emitted, hi_emit, seg_k0, K, lane);
}

// ---- constants of the sparse-keep gather kernel below -------------------------------------------
constexpr uint32_t kGatherRing = 4096;                               // codes per wave
constexpr uint32_t kGatherMaxRows = 48;                               // rows per batch (a store drain per batch)
constexpr uint32_t kGatherMaxSegCodes = (kGatherRing - 8u) / 4u;     // >= 3 whole rows + the row being scanned fit the ring
constexpr uint32_t kStageBytes = kSegSamples / 4u;                   // one row's segment of record bytes

// ---------------------------------------------------------------------------------------------
// gt_scan_gather3_kernel — sparse keeps around 1 % on long records (BASELINE config 5: 1 % of
// 500 000 samples kept, 125 KB read and 20 KB written per variant), where it is ~4 % ahead of the
// segment pick kernel below; everywhere else that kernel is as fast or faster and is the default.
//   * a block owns THREE consecutive segments (49 152 samples, 12 KiB of every record); its slice of
//     the context's kept list (ascending u32 indices, src/pfile.rs:319-333) is the rank -> sample
//     table in LDS (u16 offsets);
//   * a wave keeps the three sub-segments of a row in three register buffers — buffer q always
//     holds sub-segment q and is re-loaded with the NEXT row's sub-segment q as soon as it has been
//     parked in the wave's LDS stage — and, inside a batch, issues nothing but loads, so the
//     compiler's vmcnt waits are exact there (gfx9 counts loads and stores in ONE in-order vmcnt);
//   * lane r fetches kept sample r's code straight from the stage (work proportional to the KEPT
//     samples) into the wave's LDS code ring at a RUNNING position (row n's rank r at
//     n * seg_cnt + r);
//   * a whole batch of rows (up to 48) is flushed at once: per row one run of text (three segments'
//     worth: a row's output arrives as 11 pieces instead of 31, with a third of the partly written
//     128-B lines) = whole aligned 16-B stores + ONE byte-store instruction for its two edges, so
//     the one store drain per batch overlaps the row of loads already in flight.
// History (profiles/r01_kernel_sweeps.md): per-row ctz kernel 4.4 ms (two-round grid) -> 3.4 (one
// round) -> one-segment gather + batched flush 3.05 -> this 2.9 ms on the config-5 geometry.
// Launch precondition: at most kGatherMaxSegCodes kept samples in any aligned triple of segments.
constexpr uint32_t kSubSegs = 3;
constexpr uint32_t kSuperSamples = kSubSegs * kSegSamples;

template <bool HAS_VIDX>
__global__ __launch_bounds__(kThreads) void gt_scan_gather3_kernel(EmitArgs a, ScanArgs sc, uint32_t n_seg, uint32_t n_super, uint32_t row_groups, uint32_t xcd_groups)
{
    __shared__ uint16_t s_idx[kGatherMaxSegCodes + 2];                // rank -> sample index inside the block's 49 152 samples
    __shared__ uint32_t s_live;                                       // bit t: tile t (of 12) holds a kept sample
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[kWaves][kStageBytes];
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[kWaves][kGatherRing];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware block -> (segment triple, row group) map: the pieces of ONE row are written by the blocks of one row group, and
    // neighbouring pieces share a 128-B line at every seam.  Blocks are dealt round-robin over the 8 XCDs (b and b + 8 share
    // one), so with the plain map b = row_group * n_super + ss the two halves of a seam line come from two different L2s and
    // reach memory as two partial writes; with row groups a multiple of 8 all blocks of a row group sit on one XCD and that
    // L2 merges the line.  The first xcd_groups row groups (a multiple of 8) use that map, the rest the plain one.
    const bool xcd_map = blockIdx.x < xcd_groups * n_super;
    const uint32_t b_plain = blockIdx.x - xcd_groups * n_super;
    const uint32_t ss = xcd_map ? (blockIdx.x >> 3) % n_super : b_plain % n_super;
    const uint32_t row_group = xcd_map ? ((blockIdx.x >> 3) / n_super) * 8u + (blockIdx.x & 7u) : xcd_groups + b_plain / n_super;

    // kept samples before each of the block's sub-segment boundaries (global ranks)
    uint32_t kq[kSubSegs + 1];
#pragma unroll
    for (uint32_t q = 0; q <= kSubSegs; q++) kq[q] = __builtin_amdgcn_readfirstlane(sc.seg_rank[min(ss * kSubSegs + q, n_seg)]);
    const uint32_t K = a.kept_count;
    const uint32_t seg_k0 = kq[0];
    const uint32_t seg_cnt = kq[kSubSegs] - kq[0];
    const bool last_seg = ss + 1u == n_super;
    const uint32_t R = a.record_size;
    const uint64_t row_step = (uint64_t)row_groups * kWaves;
    const uint64_t j0 = (uint64_t)row_group * kWaves + wave;
    const uint64_t rows = j0 < a.n_variants ? (a.n_variants - j0 + row_step - 1ull) / row_step : 0ull;

    if (seg_cnt == 0u) {
        // nothing of these segments is kept; the last block of a row still owes the '\n' (:190)
        if (last_seg)
            for (uint64_t n = lane; n < rows; n += 64ull) row_text(a, j0 + n * row_step)[4ull * K] = (uint8_t)'\n';
        return;
    }
    if (tid == 0u) s_live = 0u;
    __syncthreads();
    {
        uint32_t live = 0u;
        for (uint32_t r = tid; r < seg_cnt; r += (uint32_t)kThreads) {
            const uint32_t idx = a.kept_idx[seg_k0 + r] - ss * kSuperSamples;  // < 49 152
            s_idx[r] = (uint16_t)idx;
            live |= 1u << (idx / kTileSamples);
        }
        if (live != 0u) atomicOr(&s_live, live);
    }
    __syncthreads();
    if (rows == 0ull) return;
    const uint32_t live_tiles = __builtin_amdgcn_readfirstlane(s_live);

    uint8_t *const ring = s_ring[wave];
    uint8_t *const stage = s_stage[wave];
    // Loads: a tile that lies wholly inside the record is read at scalar base + lane offset + immediate (no
    // per-tile address registers); the ONE record tile that holds the record's last byte, and tiles behind
    // it, read a window pulled back into the record (R >= 16), shifted into place when it is parked.
    const uint32_t tail_t = (R - 1u) >> 10;                                  // record tile holding the last record byte
    const uint32_t tail_b = tail_t * 1024u + lane * 16u;
    const uint32_t tail_off = min(tail_b, R - 16u);                          // pulled-back window of this lane
    const uint32_t tail_shift = tail_b + 16u <= R ? 0u : min(tail_b - (R - 16u), 16u);
    auto load_sub = [&](uint64_t n, uint32_t q, v4u(&dst)[kTilesPerSeg]) {
        const uint64_t row = j0 + min(n, rows - 1ull) * row_step;
        const uint8_t *__restrict__ rec = HAS_VIDX ? gathered_record(a, row) : a.records + row * a.record_stride;
        const uint32_t tile0 = (ss * kSubSegs + q) * kTilesPerSeg;          // first record tile of the sub-segment
        const uint8_t *__restrict__ sub = rec + (uint64_t)tile0 * 1024u;    // scalar; tiles add an immediate
#pragma unroll
        for (uint32_t t = 0; t < kTilesPerSeg; t++) {
            const uint8_t *src16 = tile0 + t < tail_t ? sub + lane * 16u + t * 1024u : rec + tail_off;
            __builtin_memcpy(&dst[t], src16, 16);  // (streaming `nt` loads were measured: 3.3-3.5 ms instead of 2.9-3.0)
        }
    };
    uint32_t base = 0u;  // ring position of rank 0 (of this block's ranks) of the row being scanned
    auto scan_sub = [&](const v4u(&w)[kTilesPerSeg], uint32_t q) {
        const uint32_t r0 = kq[q] - kq[0], r1 = kq[q + 1u] - kq[0];
        if (r0 == r1) return;  // no kept sample in this sub-segment
#pragma unroll
        for (uint32_t tile = 0; tile < kTilesPerSeg; tile++) {
            if (!(live_tiles & (1u << (q * kTilesPerSeg + tile)))) continue;
            v4u x = w[tile];
            if ((ss * kSubSegs + q) * kTilesPerSeg + tile == tail_t) {  // the record's last tile: windows were pulled back
                uint64_t lo, hi;
                window_halves(x, tail_shift, lo, hi);
                x = v4u{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
            }
            *reinterpret_cast<v4u *>(stage + tile * 1024u + lane * 16u) = x;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // lane r takes kept sample r (src/pfile.rs:171-175): byte idx / 4, bits 2 * (idx % 4)
#pragma unroll 2
        for (uint32_t r = r0 + lane; r < r1; r += 64u) {
            const uint32_t idx = (uint32_t)s_idx[r] - q * kSegSamples;
            const uint32_t byte = stage[idx >> 2];
            ring[(base + r) & (kGatherRing - 1u)] = (uint8_t)((byte >> ((idx & 3u) * 2u)) & 3u);
        }
        // the stage is rewritten by the next sub-segment: its reads above must have returned first
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto landed = [&](const v4u(&w)[kTilesPerSeg]) {
#pragma unroll
        for (uint32_t t = 0; t < kTilesPerSeg; t++) asm volatile("" ::"v"(w[t].x), "v"(w[t].y), "v"(w[t].z), "v"(w[t].w));
    };

    const uint32_t fit = (kGatherRing - 8u) / seg_cnt - 1u;              // >= 3 by the launch precondition
    const uint64_t batch = (uint64_t)min(fit, kGatherMaxRows);
    v4u b0[kTilesPerSeg], b1[kTilesPerSeg], b2[kTilesPerSeg];
    load_sub(0ull, 0u, b0);
    load_sub(0ull, 1u, b1);
    load_sub(0ull, 2u, b2);
    uint64_t n = 0ull;       // rows scanned
    uint64_t flushed = 0ull; // rows written
    while (n < rows) {
        const uint64_t batch_end = min(rows, n + batch);
        // ---- scan: loads and LDS only
        do {
            landed(b0);
            scan_sub(b0, 0u);
            load_sub(n + 1ull, 0u, b0);
            landed(b1);
            scan_sub(b1, 1u);
            load_sub(n + 1ull, 1u, b1);
            landed(b2);
            scan_sub(b2, 2u);
            load_sub(n + 1ull, 2u, b2);
            base += seg_cnt;
        } while (++n < batch_end);
        // ---- flush the batch: one run of text per row
#pragma nounroll
        for (; flushed < n; flushed++) {
            uint8_t *const row_out = row_text(a, j0 + flushed * row_step);
            const uint64_t lo_emit = 4ull * seg_k0;
            const uint64_t hi_emit = 4ull * ((uint64_t)seg_k0 + seg_cnt) + (last_seg ? 1ull : 0ull);  // '\n' closes the row (:190)
            flush_range<kGatherRing>(ring, (uint32_t)flushed * seg_cnt, row_out, lo_emit, hi_emit, seg_k0, K, lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// gt_scan_pick_kernel — the default kernel for kept subsets on records longer than one tile, any
// density.  The per-lane ctz compaction of round 1's first subset kernel
// cost 150-300 VALU instructions per store step there; this kernel has no
// compaction at all (the idea of gt_pick.hip, per segment): the block's slice of the context's
// kept list, as u16 offsets into the segment, IS the rank -> sample table in LDS; a wave parks a
// row's 4 KiB of record bytes in its LDS stage and the output-driven flush (flush_codes) picks
// every genotype straight from there: one table read + one staged-byte read per genotype.
// A row piece is >= 4 KiB of text here, so the one store drain per row piece is amortised.  Against the DENSE
// instantiation of that kernel (> 75 % kept: whole record bytes per step) it was 4-6 % faster as well
// (0.545 -> 0.566 of roofline at all-but-7 kept).
// (A 2 048-entry table instantiation for sparse keeps — 4 KiB instead of 32 KiB of LDS, twice the blocks per CU — measured 1-8 %
// SLOWER at 0.33-5 % kept: this kernel wants few, fat waves; profiles/r02_kernel_sweeps.md.)
constexpr uint32_t kPickMaxSegCodes = kSegSamples;             // up to a fully kept segment: 32 KiB of LDS for the table

template <bool HAS_VIDX>
__global__ __launch_bounds__(kThreads) void gt_scan_pick_kernel(EmitArgs a, ScanArgs sc, uint32_t n_seg, uint32_t row_groups, uint32_t xcd_groups)
{
    __shared__ uint16_t s_idx[kPickMaxSegCodes + 8];
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[kWaves][kStageBytes];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool xcd_map = blockIdx.x < xcd_groups * n_seg;  // XCD-aware block map: see gt_scan_gather3_kernel
    const uint32_t b_plain = blockIdx.x - xcd_groups * n_seg;
    const uint32_t seg = xcd_map ? (blockIdx.x >> 3) % n_seg : b_plain % n_seg;
    const uint32_t row_group = xcd_map ? ((blockIdx.x >> 3) / n_seg) * 8u + (blockIdx.x & 7u) : xcd_groups + b_plain / n_seg;
    const uint32_t K = a.kept_count;
    const uint32_t seg_k0 = __builtin_amdgcn_readfirstlane(sc.seg_rank[seg]);
    const uint32_t seg_cnt = __builtin_amdgcn_readfirstlane(sc.seg_rank[seg + 1u]) - seg_k0;
    const bool last_seg = seg + 1u == n_seg;
    const uint32_t R = a.record_size;
    const uint64_t row_step = (uint64_t)row_groups * kWaves;
    const uint64_t j0 = (uint64_t)row_group * kWaves + wave;
    const uint64_t rows = j0 < a.n_variants ? (a.n_variants - j0 + row_step - 1ull) / row_step : 0ull;

    if (seg_cnt == 0u) {
        // nothing of this segment is kept; the last segment still owes every row its '\n' (:190)
        if (last_seg)
            for (uint64_t n = lane; n < rows; n += 64ull) row_text(a, j0 + n * row_step)[4ull * K] = (uint8_t)'\n';
        return;
    }
    for (uint32_t r = tid; r < seg_cnt + 8u; r += (uint32_t)kThreads)
        s_idx[r] = r < seg_cnt ? (uint16_t)(a.kept_idx[seg_k0 + r] - seg * kSegSamples) : (uint16_t)0;  // 8 entries of slack for the flush's fifth code
    __syncthreads();
    if (rows == 0ull) return;

    uint8_t *const stage = s_stage[wave];
    // loads as in gt_scan_gather3_kernel: tiles before the record's last tile at scalar base + lane offset + immediate,
    // the last tile (and tiles behind it) as one window pulled back into the record
    const uint32_t tile0 = seg * kTilesPerSeg;
    const uint32_t tail_t = (R - 1u) >> 10;
    const uint32_t tail_b = tail_t * 1024u + lane * 16u;
    const uint32_t tail_off = min(tail_b, R - 16u);
    const uint32_t tail_shift = tail_b + 16u <= R ? 0u : min(tail_b - (R - 16u), 16u);
    auto load_row = [&](uint64_t n, v4u(&dst)[kTilesPerSeg]) {
        const uint64_t row = j0 + min(n, rows - 1ull) * row_step;
        const uint8_t *__restrict__ rec = HAS_VIDX ? gathered_record(a, row) : a.records + row * a.record_stride;
        const uint8_t *__restrict__ sub = rec + (uint64_t)tile0 * 1024u;
#pragma unroll
        for (uint32_t t = 0; t < kTilesPerSeg; t++) {
            const uint8_t *src16 = tile0 + t < tail_t ? sub + lane * 16u + t * 1024u : rec + tail_off;
            __builtin_memcpy(&dst[t], src16, 16);
        }
    };
    auto emit_row = [&](uint64_t n, const v4u(&w)[kTilesPerSeg]) {
        // park the row's segment bytes (segment byte b at stage[b])
#pragma unroll
        for (uint32_t tile = 0; tile < kTilesPerSeg; tile++) {
            v4u x = w[tile];
            if (tile0 + tile == tail_t) {
                uint64_t lo, hi;
                window_halves(x, tail_shift, lo, hi);
                x = v4u{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
            }
            if (tile0 + tile <= tail_t) *reinterpret_cast<v4u *>(stage + tile * 1024u + lane * 16u) = x;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint8_t *const row_out = row_text(a, j0 + n * row_step);
        const uint64_t lo_emit = 4ull * seg_k0;
        const uint64_t hi_emit = 4ull * ((uint64_t)seg_k0 + seg_cnt) + (last_seg ? 1ull : 0ull);  // '\n' closes the row (:190)
        const uint16_t *idx = s_idx;
        flush_codes(
            [stage, idx](uint32_t r) {
                const uint32_t s16 = idx[r];  // r <= seg_cnt + 4: inside the slack
                return ((uint32_t)stage[s16 >> 2] >> ((s16 & 3u) * 2u)) & 3u;  // src/pfile.rs:171-175
            },
            0u, row_out, lo_emit, hi_emit, seg_k0, K, lane);
        // the stage is rewritten by the next row: this row's reads must have returned first
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto landed = [&](const v4u(&w)[kTilesPerSeg]) {
#pragma unroll
        for (uint32_t t = 0; t < kTilesPerSeg; t++) asm volatile("" ::"v"(w[t].x), "v"(w[t].y), "v"(w[t].z), "v"(w[t].w));
    };
    // two register buffers, the loop unrolled by two: the next row's loads are in flight while this row's text goes out
    // (three buffers, re-loaded three rows ahead, measured no better: 3.20-3.35 vs 3.03-3.20 ms on the config-5 geometry)
    v4u b0[kTilesPerSeg], b1[kTilesPerSeg];
    load_row(0ull, b0);
    for (uint64_t n = 0;;) {
        landed(b0);
        load_row(n + 1ull, b1);
        emit_row(n, b0);
        if (++n == rows) break;
        landed(b1);
        load_row(n + 1ull, b0);
        emit_row(n, b1);
        if (++n == rows) break;
    }
}

}  // namespace

// Every block walks the same number of rows, so the launch must be exactly ONE resident round: a grid
// that exceeds residency by a few blocks runs those in a second round that takes as long as the first.
template <typename Kern>
static uint32_t resident_blocks(Kern kern, int threads, int num_cus, const Tuning &t)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (t.scan_blocks_per_cu > 0) per_cu = t.scan_blocks_per_cu;
    return (uint32_t)per_cu * (uint32_t)num_cus;
}

hipError_t launch_gt_scan(const EmitArgs &a, const ScanArgs &sc, const Tuning &t, int num_cus, hipStream_t stream)
{
    if (a.n_variants == 0) return hipSuccess;
    if (a.kept_idx == nullptr) return hipErrorInvalidValue;
    const uint32_t n_seg = (a.sample_count + kSegSamples - 1u) / kSegSamples;
    const uint32_t n_seg_eff = n_seg ? n_seg : 1u;
    const uint64_t groups_needed = ((uint64_t)a.n_variants + kWaves - 1ull) / kWaves;
    // Two kernels (interleaved A/B, N = 500 000, 60 000 variants, XCD-aware block map on: profiles/r02_kernel_sweeps.md):
    //  * the segment pick kernel — default for every density;
    //  * the three-segment gather kernel in the one band where it still measures ahead, around 1.5 % kept on records of three
    //    segments or more (0.33 % / 0.5 % / 1 % / 1.5 % / 2 % kept: pick 1.34 / 1.42 / 1.77 / 1.94 / 2.00 ms, gather 1.47 / 1.54 / 1.79 / 1.82 / 2.01);
    //    Tuning::scan_super = 0 / 1 overrides the band (1 still needs the ring precondition).
    const bool band = n_seg_eff >= kSubSegs && (uint64_t)a.kept_count * 80ull >= (uint64_t)a.sample_count &&  // >= 1.25 % kept
                      (uint64_t)a.kept_count * 53ull <= (uint64_t)a.sample_count;                            // <= 1.9 %
    const bool super_kernel = sc.max_super_count <= kGatherMaxSegCodes && (t.scan_super >= 0 ? t.scan_super != 0 : band);
    if (super_kernel) {
        void (*k3)(EmitArgs, ScanArgs, uint32_t, uint32_t, uint32_t, uint32_t) = gathered(a) ? gt_scan_gather3_kernel<true> : gt_scan_gather3_kernel<false>;
        const uint32_t n_super = (n_seg_eff + kSubSegs - 1u) / kSubSegs;
        uint64_t groups = (uint64_t)resident_blocks(k3, kThreads, num_cus, t) / n_super;
        if (groups < 1ull) groups = 1ull;
        if (groups > groups_needed) groups = groups_needed;
        const uint32_t xcd_groups = t.scan_xcd_map != 0 ? (uint32_t)(groups & ~7ull) : 0u;
        hipLaunchKernelGGL(k3, dim3((uint32_t)(groups * n_super)), dim3(kThreads), 0, stream, a, sc, n_seg_eff, n_super, (uint32_t)groups, xcd_groups);
        return hipGetLastError();
    }
    void (*kern)(EmitArgs, ScanArgs, uint32_t, uint32_t, uint32_t) = gathered(a) ? gt_scan_pick_kernel<true> : gt_scan_pick_kernel<false>;
    uint64_t groups = (uint64_t)resident_blocks(kern, kThreads, num_cus, t) / n_seg_eff;  // floor: never a partial second round
    if (groups < 1ull) groups = 1ull;  // more segments than resident blocks (N > ~16 M samples): rounds are unavoidable
    if (groups > groups_needed) groups = groups_needed;
    const uint32_t xcd_groups = t.scan_xcd_map != 0 ? (uint32_t)(groups & ~7ull) : 0u;
    const uint32_t grid = (uint32_t)(groups * n_seg_eff);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), 0, stream, a, sc, n_seg_eff, (uint32_t)groups, xcd_groups);
    return hipGetLastError();
}

}  // namespace pgenhip
