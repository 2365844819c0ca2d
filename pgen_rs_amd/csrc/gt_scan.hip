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
#include <type_traits>

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

constexpr uint32_t kStageBytes = kSegSamples / 4u;                   // one row's segment of record bytes

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

// FOUR (default): a table entry is 2 x position << 12 | byte offset (code by one v_bfe_u32); a chunk's four entries come with ONE
// LDS read, it makes FOUR picks and takes its fifth genotype from the next lane (flush_text4, gt_common.hip.h): 40 VALU + 6.5 LDS
// instructions per chunk where round 2's form (FOUR = false, kept for the A/B: entry = sample offset, five picks by shifts) has 70 +
// 10.  Also tried (profiles/r03_kernel_sweeps.md §8): a 4-KiB byte -> text table in LDS (36 VALU + 15 LDS reads, LDS-bound),
// bit-field extract alone (59 + 10), and a block-cooperative form with a loader wave (level).
template <bool HAS_VIDX, uint32_t U, bool FOUR>
__global__ __launch_bounds__(kThreads) void gt_scan_pick_kernel(EmitArgs a, ScanArgs sc, uint32_t n_seg, uint32_t row_groups, uint32_t xcd_groups, uint32_t bands)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_idx[kPickMaxSegCodes + 16];
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[kWaves][kStageBytes];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware block -> (segment, row group) map: the pieces of ONE row are written by the blocks of one row group, and neighbouring
    // pieces share a 128-B line at every seam.  Blocks are dealt round-robin over the 8 XCDs (b and b + 8 share one), so with the
    // plain map b = row_group * n_seg + seg the two halves of a seam line come from two different L2s and reach memory as two partial
    // writes; with all blocks of a row group on one XCD that L2 merges the line (+1-7 %, profiles/r02_kernel_sweeps.md).  The first
    // xcd_groups row groups (a multiple of 8) use that map, the rest the plain one.
    const bool xcd_map = blockIdx.x < xcd_groups * n_seg;
    const uint32_t b_plain = blockIdx.x - xcd_groups * n_seg;
    const uint32_t seg = xcd_map ? (blockIdx.x >> 3) % n_seg : b_plain % n_seg;
    const uint32_t row_group = xcd_map ? ((blockIdx.x >> 3) / n_seg) * 8u + (blockIdx.x & 7u) : xcd_groups + b_plain / n_seg;
    const uint32_t K = a.kept_count;
    const uint32_t seg_k0 = __builtin_amdgcn_readfirstlane(sc.seg_rank[seg]);
    const uint32_t seg_cnt = __builtin_amdgcn_readfirstlane(sc.seg_rank[seg + 1u]) - seg_k0;
    const bool last_seg = seg + 1u == n_seg;
    const uint32_t R = a.record_size;
    // BANDS: the launch's rows are cut into `bands` contiguous bands and row group g works in band g % bands (with the XCD map: one
    // band per XCD), so the launch reads (and writes) at `bands` distant fronts instead of one window of row_groups * 4 rows that
    // all blocks share — where the driver put the buffers then matters less (see the write fronts of the stream kernel, gt_wide.hip)
    const uint32_t band = row_group % bands, group_in_band = row_group / bands;
    const uint64_t band_rows = ((uint64_t)a.n_variants + bands - 1u) / bands;
    const uint64_t band_lo = (uint64_t)band * band_rows, band_hi = min((uint64_t)a.n_variants, band_lo + band_rows);
    const uint64_t row_step = (uint64_t)(row_groups / bands) * kWaves;
    const uint64_t j0 = band_lo + (uint64_t)group_in_band * kWaves + wave;
    const uint64_t rows = j0 < band_hi ? (band_hi - j0 + row_step - 1ull) / row_step : 0ull;

    // full lines: this wave's share of the prefix bytes first (gt_common.hip.h; every block comes by here, whatever its segment holds)
    if (a.line_off != nullptr) {
        const uint32_t pfx_shift = prefix_copy_shift(a);
        if (pfx_shift != 0u) copy_prefix_rows(a, pfx_shift, (uint64_t)blockIdx.x * kWaves + wave, (uint64_t)gridDim.x * kWaves, lane);
    }
    if (seg_cnt == 0u) {
        // nothing of this segment is kept; the last segment still owes every row its '\n' (:190)
        if (last_seg)
            for (uint64_t n = lane; n < rows; n += 64ull) row_text(a, j0 + n * row_step)[4ull * K] = (uint8_t)'\n';
        return;
    }
    for (uint32_t r = tid; r < seg_cnt + 16u; r += (uint32_t)kThreads) {
        const uint32_t s16 = r < seg_cnt ? a.kept_idx[seg_k0 + r] - seg * kSegSamples : 0u;  // 16 entries of slack for the flush's fifth code / second group
        s_idx[r] = (uint16_t)(FOUR ? ((s16 & 3u) << 13) | (s16 >> 2) : s16);
    }
    __syncthreads();
    if (rows == 0ull) return;

    uint8_t *const stage = s_stage[wave];
    // loads: tiles before the record's last tile at scalar base + lane offset + immediate (no per-tile address registers),
    // the last tile (and tiles behind it) as one window pulled back into the record
    const uint32_t tile0 = seg * kTilesPerSeg;
    const uint32_t tail_t = (R - 1u) >> 10;
    const uint32_t tail_b = tail_t * 1024u + lane * 16u;
    const uint32_t tail_off = min(tail_b, R - 16u);
    const uint32_t tail_shift = tail_b + 16u <= R ? 0u : min(tail_b - (R - 16u), 16u);
    // (full lines: where the row's GT segment starts — line_off + prefix length — travels with the row's loads, a row ahead,
    // instead of three dependent loads in front of every flush)
    const bool lines = a.line_off != nullptr;
    auto load_row = [&](uint64_t n, v4u(&dst)[kTilesPerSeg], uint64_t &toff) {
        const uint64_t row = j0 + min(n, rows - 1ull) * row_step;
        if (lines) toff = a.line_off[row] + (a.prefix_off[row + 1ull] - a.prefix_off[row]);
        const uint8_t *__restrict__ rec = HAS_VIDX ? gathered_record(a, row) : a.records + row * a.record_stride;
        const uint8_t *__restrict__ sub = rec + (uint64_t)tile0 * 1024u;
#pragma unroll
        for (uint32_t t = 0; t < kTilesPerSeg; t++) {
            const uint8_t *src16 = tile0 + t < tail_t ? sub + lane * 16u + t * 1024u : rec + tail_off;
            __builtin_memcpy(&dst[t], src16, 16);
        }
    };
    auto emit_row = [&](uint64_t n, const v4u(&w)[kTilesPerSeg], uint64_t toff) {
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
        uint8_t *const row_out = lines ? a.out + toff : a.out + (j0 + n * row_step) * a.out_stride;
        const uint64_t lo_emit = 4ull * seg_k0;
        const uint64_t hi_emit = 4ull * ((uint64_t)seg_k0 + seg_cnt) + (last_seg ? 1ull : 0ull);  // '\n' closes the row (:190)
        const uint16_t *idx = s_idx;
        if (FOUR) {
            auto code_of = [stage](uint32_t e, uint32_t byte_lo, uint32_t shift_lo) {   // entry in bits [byte_lo, byte_lo + 12) and [shift_lo, shift_lo + 3)
                return __builtin_amdgcn_ubfe((uint32_t)stage[__builtin_amdgcn_ubfe(e, byte_lo, 12u)], __builtin_amdgcn_ubfe(e, shift_lo, 3u), 2u);   // src/pfile.rs:171-175
            };
            flush_text4<U>(
                [idx, code_of](auto c0, uint32_t g, uint32_t &k0, uint32_t &k1, uint32_t &k2, uint32_t &k3) {
                    constexpr uint32_t C0 = decltype(c0)::value;
                    const uint32_t *grp = reinterpret_cast<const uint32_t *>(idx) + 2u * g;     // 8-byte aligned; the second group may be slack
                    uint32_t p01, p23;                                                            // entries (r, r+1), (r+2, r+3)
                    if (C0 == 0u) { p01 = grp[0]; p23 = grp[1]; }
                    else if (C0 == 2u) { p01 = grp[1]; p23 = grp[2]; }
                    else {
                        const uint32_t w0 = grp[C0 == 1u ? 0 : 1], w1 = grp[C0 == 1u ? 1 : 2], w2 = grp[C0 == 1u ? 2 : 3];
                        p01 = __builtin_amdgcn_alignbyte(w1, w0, 2u);
                        p23 = __builtin_amdgcn_alignbyte(w2, w1, 2u);
                    }
                    k0 = code_of(p01, 0u, 12u);
                    k1 = code_of(p01, 16u, 28u);
                    k2 = code_of(p23, 0u, 12u);
                    k3 = code_of(p23, 16u, 28u);
                },
                [idx, code_of](uint32_t r) -> uint32_t { return code_of(idx[r], 0u, 12u); },
                0u, row_out, lo_emit, hi_emit, seg_k0, K, lane, sc.align_stores != 0u);
        } else {
            flush_codes<U>(
                [stage, idx](uint32_t r) -> uint32_t {
                    const uint32_t s16 = idx[r];  // r <= seg_cnt + 4: inside the slack
                    return ((uint32_t)stage[s16 >> 2] >> ((s16 & 3u) * 2u)) & 3u;  // src/pfile.rs:171-175
                },
                0u, row_out, lo_emit, hi_emit, seg_k0, K, lane);
        }
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
    uint64_t t0 = 0ull, t1 = 0ull;
    load_row(0ull, b0, t0);
    for (uint64_t n = 0;;) {
        landed(b0);
        load_row(n + 1ull, b1, t1);
        emit_row(n, b0, t0);
        if (++n == rows) break;
        landed(b1);
        load_row(n + 1ull, b0, t0);
        emit_row(n, b1, t1);
        if (++n == rows) break;
    }
}

// ---------------------------------------------------------------------------------------------
// gt_compact_kernel — first pass of the two-pass path for sparse keeps on long records (capi.hip; BASELINE configs[4]).
// Same block map, same table, same wide loads as gt_scan_pick_kernel, but instead of text the block writes its part of each
// row's COMPACT record to a.out + j * a.out_stride: the K kept codes packed four to a byte exactly like a mode-0x02 record of
// K samples (src/pfile.rs:171-175 applied here; :177-190 by the all-samples kernels in the second pass).  Byte b of a compact
// record belongs to the segment that owns rank 4b: a block writes bytes ceil(k0/4) .. ceil(k1/4)-1 of its rank slice [k0, k1)
// and fetches the up to three ranks behind k1 that share its last byte straight from the record (their samples live in later
// segments) — no byte is written twice, nothing needs zeroing.
//
// This kernel is a pure record READER (125 000 bytes in, 1 250 out per row at 1 % kept), so the one thing that matters is that
// its loads never wait for anything but loads.  gfx9 counts loads and stores in ONE in-order vmcnt, and the compiler cannot
// count stores that sit in a loop or behind a branch: round 2's form — next row's loads, this row's picks and byte stores, wait
// for the loads — compiled to `s_waitcnt vmcnt(0)` in front of every row, i.e. every row also waited for the acknowledgement of
// the bytes it had just stored (1.02 ms per 5.2-GB chunk = 5.1 TB/s, while the same loads alone run at 6.3-6.4 TB/s:
// tools/readbench.hip, profiles/r03_kernel_sweeps.md).  Here a row's compact bytes are parked in the wave's LDS out-stage and
// leave ONE ROW LATER, right behind the wait and in front of the next loads: in program order nothing is ever younger than the
// loads a wave waits for, so vmcnt(0) costs nothing.
template <bool HAS_VIDX>
__global__ __launch_bounds__(kThreads) void gt_compact_kernel(EmitArgs a, ScanArgs sc, uint32_t n_seg, uint32_t row_groups, uint32_t xcd_groups)
{
    __shared__ uint16_t s_idx[kCompactMaxSegCodes + 8];
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[kWaves][kStageBytes];
    __shared__ uint8_t s_out[kWaves][kCompactMaxSegCodes / 4u + 64u];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the XCD-aware block -> (segment, row group) map of gt_scan_pick_kernel: a row's 31 pieces leave through ONE L2
    const bool xcd_map = blockIdx.x < xcd_groups * n_seg;
    const uint32_t b_plain = blockIdx.x - xcd_groups * n_seg;
    const uint32_t seg = xcd_map ? (blockIdx.x >> 3) % n_seg : b_plain % n_seg;
    const uint32_t row_group = xcd_map ? ((blockIdx.x >> 3) / n_seg) * 8u + (blockIdx.x & 7u) : xcd_groups + b_plain / n_seg;
    const uint32_t K = a.kept_count;
    const uint32_t seg_k0 = __builtin_amdgcn_readfirstlane(sc.seg_rank[seg]);
    const uint32_t seg_cnt = __builtin_amdgcn_readfirstlane(sc.seg_rank[seg + 1u]) - seg_k0;
    const uint32_t R = a.record_size;
    const uint64_t row_step = (uint64_t)row_groups * kWaves;
    const uint64_t j0 = (uint64_t)row_group * kWaves + wave;
    const uint64_t rows = j0 < (uint64_t)a.n_variants ? ((uint64_t)a.n_variants - j0 + row_step - 1ull) / row_step : 0ull;
    if (seg_cnt == 0u) return;   // nothing of this segment is kept
    // bytes of the compact record this block owns, and the ranks behind its slice that share its last byte
    const uint32_t seg_k1 = seg_k0 + seg_cnt;
    const uint32_t cb0 = (seg_k0 + 3u) >> 2, cb1 = (seg_k1 + 3u) >> 2;
    if (cb0 == cb1) return;      // every rank of the slice sits in a byte an earlier segment owns
    const uint32_t n_foreign = min((cb1 << 2) - seg_k1, K - seg_k1);   // 0 .. 3
    uint32_t f_smp = 0u;         // lane i < n_foreign: sample of rank seg_k1 + i
    if (lane < n_foreign) f_smp = a.kept_idx[seg_k1 + lane];
    for (uint32_t r = tid; r < seg_cnt + 8u; r += (uint32_t)kThreads)
        s_idx[r] = r < seg_cnt ? (uint16_t)(a.kept_idx[seg_k0 + r] - seg * kSegSamples) : (uint16_t)0;  // slack entries read as sample 0
    __syncthreads();
    if (rows == 0ull) return;

    uint8_t *const stage = s_stage[wave];
    uint8_t *const ostage = s_out[wave];
    const uint32_t tile0 = seg * kTilesPerSeg;
    const uint32_t tail_t = (R - 1u) >> 10;
    const uint32_t tail_b = tail_t * 1024u + lane * 16u;
    const uint32_t tail_off = min(tail_b, R - 16u);
    const uint32_t tail_shift = tail_b + 16u <= R ? 0u : min(tail_b - (R - 16u), 16u);
    auto load_row = [&](uint64_t n, v4u(&dst)[kTilesPerSeg], uint32_t &fbyte) {
        const uint64_t row = j0 + min(n, rows - 1ull) * row_step;
        const uint8_t *__restrict__ rec = HAS_VIDX ? gathered_record(a, row) : a.records + row * a.record_stride;
        const uint8_t *__restrict__ sub = rec + (uint64_t)tile0 * 1024u;
#pragma unroll
        for (uint32_t t = 0; t < kTilesPerSeg; t++) {
            const uint8_t *src16 = tile0 + t < tail_t ? sub + lane * 16u + t * 1024u : rec + tail_off;
            __builtin_memcpy(&dst[t], src16, 16);
        }
        if (lane < n_foreign) fbyte = rec[f_smp >> 2];  // the record byte of a rank behind the slice (a later segment's sample)
    };
    // row n's compact bytes, picked from the parked record bytes into the out-stage (lane <-> byte b = ranks 4b .. 4b+3)
    auto pick_row = [&](const v4u(&w)[kTilesPerSeg], uint32_t fbyte) {
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
        // the foreign ranks' codes, wave-uniform: rank seg_k1 + i sits at bits 2 * ((seg_k1 + i) & 3) of the last owned byte
        uint32_t f_bits = 0u;
        const uint32_t f_code = (fbyte >> ((f_smp & 3u) * 2u)) & 3u;
        for (uint32_t i = 0; i < n_foreign; i++)
            f_bits |= (uint32_t)__builtin_amdgcn_readlane((int)f_code, (int)i) << (2u * ((seg_k1 + i) & 3u));
#pragma clang loop unroll(disable)
        for (uint32_t b = cb0 + lane; b < cb1; b += 64u) {
            uint32_t byte = 0u;
#pragma unroll
            for (uint32_t j = 0; j < 4u; j++) {
                const uint32_t r = 4u * b + j - seg_k0;           // rank inside the slice (>= 0: b >= ceil(seg_k0 / 4))
                const uint32_t s16 = s_idx[min(r, seg_cnt)];      // entries behind the slice are 0 (slack)
                const uint32_t code = ((uint32_t)stage[s16 >> 2] >> ((s16 & 3u) * 2u)) & 3u;   // src/pfile.rs:171-175
                byte |= (r < seg_cnt ? code : 0u) << (2u * j);
            }
            if (b + 1u == cb1) byte |= f_bits;
            ostage[b - cb0] = (uint8_t)byte;   // (the same lane takes it out again: no cross-lane traffic through the out-stage)
        }
        // the stage is rewritten by the next row: this row's reads must have returned first
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // the parked bytes of row n leave for HBM (one row after they were picked)
    auto store_row = [&](uint64_t n) {
        uint8_t *const crow = a.out + (j0 + n * row_step) * a.out_stride;
#pragma clang loop unroll(disable)
        for (uint32_t b = cb0 + lane; b < cb1; b += 64u) crow[b] = ostage[b - cb0];
    };
    auto landed = [&](const v4u(&w)[kTilesPerSeg]) {
#pragma unroll
        for (uint32_t t = 0; t < kTilesPerSeg; t++) asm volatile("" ::"v"(w[t].x), "v"(w[t].y), "v"(w[t].z), "v"(w[t].w));
    };
    v4u b0[kTilesPerSeg], b1[kTilesPerSeg];
    uint32_t f0 = 0u, f1 = 0u;
    load_row(0ull, b0, f0);
    for (uint64_t n = 0;;) {
        landed(b0);                       // vmcnt(0): only row n's loads (and long-gone stores) are outstanding
        if (n != 0ull) store_row(n - 1ull);
        load_row(n + 1ull, b1, f1);
        pick_row(b0, f0);
        if (++n == rows) break;
        landed(b1);
        store_row(n - 1ull);
        load_row(n + 1ull, b0, f0);
        pick_row(b1, f1);
        if (++n == rows) break;
    }
    store_row(rows - 1ull);
}

}  // namespace

// Every block walks the same number of rows, so the launch must be exactly ONE resident round: a grid
// that exceeds residency by a few blocks runs those in a second round that takes as long as the first.
template <typename Kern>
static uint32_t resident_blocks(Kern kern, int threads, int num_cus, const Tuning &t, int preferred)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (preferred > 0 && preferred < per_cu) per_cu = preferred;
    if (t.scan_blocks_per_cu > 0) per_cu = t.scan_blocks_per_cu;
    return (uint32_t)per_cu * (uint32_t)num_cus;
}

hipError_t launch_gt_scan(const EmitArgs &a, const ScanArgs &sc, const Tuning &t, int num_cus, hipStream_t stream, bool compact)
{
    if (a.n_variants == 0) return hipSuccess;
    if (a.kept_idx == nullptr) return hipErrorInvalidValue;
    const uint32_t n_seg = (a.sample_count + kSegSamples - 1u) / kSegSamples;
    const uint32_t n_seg_eff = n_seg ? n_seg : 1u;
    const uint64_t groups_needed = ((uint64_t)a.n_variants + kWaves - 1ull) / kWaves;
    // One kernel for every density (round 1's three-segment gather kernel kept a 0.8-2 % band; with the XCD-aware block map and
    // two blocks per CU the pick kernel is ahead there too: 1.5 % kept 1.85 vs 1.99 ms, profiles/r02_kernel_sweeps.md).
    // Blocks per CU: the occupancy API says 3 (32 KiB table + 16 KiB of stages); from ~0.6 % kept upwards 2 measure the same or
    // better (+5-9 % at 1-3 % kept, level from 30 %), below that 3 do (the kernel is then a pure record reader).
    if (compact) {
        // first pass of the two-pass path: a.out / a.out_stride address the compact records (ceil(K / 4) bytes per row)
        if (sc.max_seg_count > kCompactMaxSegCodes) return hipErrorInvalidValue;  // (capi.hip takes the single pass then)
        void (*ckern)(EmitArgs, ScanArgs, uint32_t, uint32_t, uint32_t) = gathered(a) ? gt_compact_kernel<true> : gt_compact_kernel<false>;
        uint64_t groups = (uint64_t)resident_blocks(ckern, kThreads, num_cus, t, 2) / n_seg_eff;  // floor: never a partial second round; two blocks per CU (3: -1 %, 5: -6 %)
        if (groups < 1ull) groups = 1ull;
        if (groups > groups_needed) groups = groups_needed;
        // one narrow window of rows (a record reader: banded -3 %, profiles/r02_kernel_sweeps.md)
        const uint32_t xcd_groups = t.scan_xcd_map != 0 ? (uint32_t)(groups & ~7ull) : 0u;
        hipLaunchKernelGGL(ckern, dim3((uint32_t)(groups * n_seg_eff)), dim3(kThreads), 0, stream, a, sc, n_seg_eff, (uint32_t)groups, xcd_groups);
        return hipGetLastError();
    }
    typedef void (*Kern)(EmitArgs, ScanArgs, uint32_t, uint32_t, uint32_t, uint32_t);
    const bool g = gathered(a);
    auto by_mode = [&](auto u) -> Kern {
        constexpr uint32_t U = decltype(u)::value;
        if (t.scan_four_picks != 0) return g ? gt_scan_pick_kernel<true, U, true> : gt_scan_pick_kernel<false, U, true>;
        return g ? gt_scan_pick_kernel<true, U, false> : gt_scan_pick_kernel<false, U, false>;
    };
    Kern kern = t.flush_unroll == 1 ? by_mode(std::integral_constant<uint32_t, 1>{}) : t.flush_unroll == 4 ? by_mode(std::integral_constant<uint32_t, 4>{}) : by_mode(std::integral_constant<uint32_t, 2>{});
    const int preferred = (uint64_t)a.kept_count * 170ull >= (uint64_t)a.sample_count ? 2 : 0;
    uint64_t groups = (uint64_t)resident_blocks(kern, kThreads, num_cus, t, preferred) / n_seg_eff;  // floor: never a partial second round
    if (groups < 1ull) groups = 1ull;  // more segments than resident blocks (N > ~16 M samples): rounds are unavoidable
    if (groups > groups_needed) groups = groups_needed;
    const uint32_t xcd_groups = t.scan_xcd_map != 0 ? (uint32_t)(groups & ~7ull) : 0u;
    const uint32_t grid = (uint32_t)(groups * n_seg_eff);
    // Eight bands where the text dominates the traffic (>= 10 % kept: N = 500 000, 10 % kept 0.599 -> 0.626 of roofline, 50 % 0.589 ->
    // 0.593); one front where the launch is mostly a record reader (1 % kept, compact pass: 0.634 vs 0.614 banded; 0.3 % single pass:
    // 0.741 vs 0.693) — reads like the one narrow window, writes like several fronts (profiles/r02_kernel_sweeps.md)
    const bool banded = (uint64_t)a.kept_count * 10ull >= (uint64_t)a.sample_count && groups % 8ull == 0ull && groups_needed >= 64ull * groups;
    const uint32_t bands = banded ? 8u : 1u;   // (every band holds the same number of row groups)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), 0, stream, a, sc, n_seg_eff, (uint32_t)groups, xcd_groups, bands);
    return hipGetLastError();
}

}  // namespace pgenhip
