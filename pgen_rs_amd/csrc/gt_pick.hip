// gt_pick.hip — kept subsets on SHORT records (N <= 4096 samples, R <= 1 KiB: the 1000 Genomes
// shape with a sample filter), any density (gfx950 / MI355X).
//
// Replaces /root/reference/src/pfile.rs:165-190 when `--include-sam` is active and a record is
// at most one 1-KiB tile.  The scan kernels of gt_scan.hip are built around 16 384-sample segments
// and a per-row code ring; on a 626-byte record they spend a wave on one row piece at a time and
// most of their loads on nothing.  Here everything is OUTPUT-driven and there is no compaction
// step at all:
//   * the kept-sample list the context already holds (ascending u32 indices, what filter_metadata
//     src/pfile.rs:319-333 yields) is copied once per block into LDS as u16: it IS the
//     rank -> sample table;
//   * a wave takes BATCHES of B consecutive rows: one 16-B-per-lane load instruction per row
//     (issued for the next batch before this batch's text goes out), the rows parked in the
//     wave's LDS stage at a 16-B-padded pitch;
//   * the batch's output is one contiguous run of B x (4K+1) bytes (dense pitch is a launch
//     precondition).  Lanes own 16-byte-ALIGNED chunks of it; a chunk at run offset o lies in row
//     i = o / (4K+1) at row byte p: its five genotypes are ranks p/4 .. p/4+4 of that row — five
//     u16 reads of the table, five byte reads of the staged record (src/pfile.rs:171-175), the
//     usual text dwords and funnel shift (:177-190).  A chunk that holds a row's '\n' merges the
//     tail of row i with the head of row i+1 in registers, so it is still one 16-B store;
//   * the run's first and last partial chunk (shared with the neighbouring waves' runs) go out as
//     ONE byte-store instruction (lanes 0-15 head bytes, 16-31 tail bytes);
//   * full-line mode (pgenhip_emit_lines): the rows sit behind their prefixes, so each parked row
//     is flushed on its own through flush_codes with the same table + staged-byte pick.
// HBM traffic per row: the record once + 4K+1 bytes of text; the kept list once per block.
#include "gt_common.hip.h"
#include "kernels.h"

namespace pgenhip {

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr uint32_t kMaxSamples = 4096;          // R <= 1024: one 16-B piece per lane covers a record
constexpr uint32_t kStageBytes = 8192;          // per wave: B rows at the padded pitch (PACKED: the batch's contiguous record bytes)
constexpr int kMaxBatchRows = 12;               // load instructions per batch (one per row; PACKED: one per KiB of the batch's records)
constexpr uint32_t kMaxPackedRows = 64;         // PACKED batches: rows (their heads are prepared by lane <-> row)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

struct PickParams {
    uint32_t pieces;      // 16-B pieces per record = ceil(R / 16) (<= 64)
    uint32_t pitch;       // bytes per staged row = pieces * 16
    uint32_t batch_rows;  // B
    uint32_t row_bytes;   // S = 4K + 1
    uint32_t magic;       // floor(2^32 / S) + 1: o / S = umulhi(o, magic) for o < 2^20 (checked on the host), fixed up by one compare
    uint32_t n_batches;
    uint32_t pfx_shift;   // full lines: lanes per line (log2) of the in-kernel prefix copy, 0 = none
    uint32_t align_stores; // Tuning::align_stores
    uint32_t packed;      // 1: dense records, no gather — a batch's B records are ONE contiguous run of B*R bytes, fetched KiB by KiB and
                          // parked as they lie (pitch = R): short records no longer cost a load instruction and a register quad per
                          // ROW, and a batch is up to 64 rows instead of 12 (N = 300 with 30 samples kept: 7.7 KB of text per batch, not 1.4)
};

// run offset -> (row in batch, row byte)
__device__ __forceinline__ void split_offset(uint32_t o, const PickParams &p, uint32_t &i, uint32_t &pos)
{
    i = __umulhi(o, p.magic);
    if (i * p.row_bytes > o) i--;
    pos = o - i * p.row_bytes;
}

// 2-bit code of kept sample `rank` of staged row `row`.  `idx` points at table entry 0; the table has kPadBefore
// entries of slack in front and kPadAfter behind (ranks -4 .. K+4 occur next to a row's ends: their bytes are
// never stored), so no clamp is needed.  IDENT (all samples kept): rank r IS sample r, no table.
constexpr uint32_t kPadBefore = 4, kPadAfter = 12;
template <bool IDENT>
__device__ __forceinline__ uint32_t pick_code(const uint8_t *row, const uint16_t *idx, int32_t rank, uint32_t K)
{
    (void)K;
    if (!IDENT) {
        // table entry = 2 x (position in the byte) << 12 | byte offset in the record (pick_entry below): one v_bfe_u32 per code
        const uint32_t e = idx[rank];
        return __builtin_amdgcn_ubfe((uint32_t)row[e & 0xFFFu], e >> 12, 2u);   // src/pfile.rs:171-175
    }
    const uint32_t s = (uint32_t)max(rank, 0);
    return ((uint32_t)row[s >> 2] >> ((s & 3u) * 2u)) & 3u;
}
__device__ __forceinline__ uint16_t pick_entry(uint32_t sample) { return (uint16_t)(((sample & 3u) << 13) | (sample >> 2)); }

// 16 text bytes of staged row `row` starting at row byte q (q may be negative: the bytes before the row are don't-care)
template <bool IDENT>
__device__ __forceinline__ u32x4 pick_text16(const uint8_t *row, const uint16_t *idx, int32_t q, uint32_t K)
{
    if (IDENT) {
        // consecutive samples: one 16-bit window of the record holds the five codes (as in the stream kernel)
        uint32_t window;
        if (q >= 0) {
            uint16_t h;
            __builtin_memcpy(&h, row + (q >> 4), 2);
            window = h;
        } else {
            window = (uint32_t)row[0] << 8;  // record byte -1 (none) and byte 0
        }
        return gt_text16_from_window(window, (int64_t)q);
    }
    // five picks, no text dwords: the variable bytes of genotypes (0, 1), (2, 3), (4) in three registers (gt_vars2), their 2-byte
    // realignments for the odd dwords, and ONE selector for all four output dwords — bytes [sh, sh + 4) of the selector sequence
    // '\t', a1, '/', a2, '\t', a1', '/', a2' of a genotype pair (4 = '\t', 5 = '/', 0-3 = the pair register's bytes): 36 VALU
    // instructions per chunk where text dwords + funnel shifts + the old shift-and-mask picks took 56
    const int32_t g = q >> 2;  // floor
    const uint32_t sh = (uint32_t)q & 3u;
    const uint32_t V0 = gt_vars2(pick_code<false>(row, idx, g, K), pick_code<false>(row, idx, g + 1, K));
    const uint32_t V1 = gt_vars2(pick_code<false>(row, idx, g + 2, K), pick_code<false>(row, idx, g + 3, K));
    const uint32_t V2 = gt_vars2(pick_code<false>(row, idx, g + 4, K), 0u);
    const uint32_t X1 = __builtin_amdgcn_alignbyte(V1, V0, 2u), X3 = __builtin_amdgcn_alignbyte(V2, V1, 2u);
    const uint32_t sel = __builtin_amdgcn_alignbyte(0x03050204u, 0x01050004u, sh);
    constexpr uint32_t kConst = 0x00002F09u;                            // byte 0 = '\t', byte 1 = '/'
    u32x4 v;
    v.x = __builtin_amdgcn_perm(kConst, V0, sel);
    v.y = __builtin_amdgcn_perm(kConst, X1, sel);
    v.z = __builtin_amdgcn_perm(kConst, V1, sel);
    v.w = __builtin_amdgcn_perm(kConst, X3, sel);
    return v;
}

template <bool HAS_VIDX, bool LINES, bool IDENT>
__global__ __launch_bounds__(kThreads) void gt_pick_kernel(EmitArgs a, PickParams p)
{
    __shared__ uint16_t s_tab[kPadBefore + kMaxSamples + kPadAfter];
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[kWaves][kStageBytes];
    // per wave and batch row: 16 zero bytes, then the row's first four genotypes as text (its first 16 bytes) — what the chunk
    // that holds the PREVIOUS row's '\n' needs behind it
    __shared__ __attribute__((aligned(16))) uint32_t s_heads[kWaves][(kMaxPackedRows + 1) * 8];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t K = a.kept_count;
    if (!IDENT) {
        for (uint32_t r = tid; r < kPadBefore + K + kPadAfter; r += (uint32_t)kThreads)
            s_tab[r] = r >= kPadBefore && r < kPadBefore + K ? pick_entry(a.kept_idx[r - kPadBefore]) : (uint16_t)0;
    }
    for (uint32_t r = tid; r < (uint32_t)kWaves * (kMaxPackedRows + 1) * 8u; r += (uint32_t)kThreads) (&s_heads[0][0])[r] = 0u;
    __syncthreads();
    const uint16_t *const s_idx = s_tab + kPadBefore;

    uint8_t *const stage = s_stage[wave];
    const uint32_t R = a.record_size;
    const uint32_t S = p.row_bytes;
    const uint32_t B = p.batch_rows;
    // Rows that are NOT one dense byte run (gathered / padded records) are fetched row by row, G = 64 / lanes_per_row rows per load
    // instruction: lane -> (row `grp` of the instruction, 16-byte piece `piece` of its record); pitch = lanes_per_row * 16.
    // The record's last piece is read pulled back into the record (R >= 16) and shifted into place when parked.
    const uint32_t lpr = p.pitch >> 4;                                  // lanes per row: pieces rounded up to a power of two (not PACKED)
    const uint32_t grp = lane / lpr, piece = lane & (lpr - 1u), G = 64u / lpr;
    const uint32_t piece_off = min(piece * 16u, R - 16u);
    const uint32_t tail_shift = piece * 16u + 16u <= R ? 0u : min(piece * 16u - (R - 16u), 16u);

    const uint32_t batch_step = gridDim.x * kWaves;
    uint32_t bi = blockIdx.x * kWaves + wave;
    // full lines: this wave's share of the prefix bytes first (gt_common.hip.h)
    if (LINES && p.pfx_shift != 0u) copy_prefix_rows(a, p.pfx_shift, (uint64_t)bi, (uint64_t)batch_step, lane);
    if (bi >= p.n_batches) return;

    v4u buf[kMaxBatchRows];
    const bool packed = p.packed != 0u;
    // full-line mode: where row (batch row0 + lane)'s GT segment starts in the output (line_off + prefix length), fetched with the
    // batch's records — one coalesced load per batch instead of three dependent scalar loads in front of every row's flush
    uint64_t text_off = 0ull;
    auto load_batch = [&](uint32_t b) {
        const uint64_t row0 = (uint64_t)b * B;
        if (LINES) {
            const uint64_t jr = min(row0 + (uint64_t)lane, (uint64_t)a.n_variants - 1ull);
            text_off = a.line_off[jr] + (a.prefix_off[jr + 1ull] - a.prefix_off[jr]);
        }
        if (packed) {
            // the batch's records as one byte run, from the 16-B boundary below its first byte
            const uint8_t *__restrict__ run0 = a.records + row0 * (uint64_t)R;
            const uint32_t mis = (uint32_t)(uintptr_t)run0 & 15u;
            const uint32_t n_bytes = mis + (uint32_t)min((uint64_t)B, (uint64_t)a.n_variants - row0) * R;   // <= kStageBytes (host)
#pragma unroll
            for (int i = 0; i < kMaxBatchRows; i++) {
                buf[i] = v4u{0u, 0u, 0u, 0u};
                const uint32_t off = (uint32_t)i * 1024u + lane * 16u;
                if ((uint32_t)i * 1024u < n_bytes && off < n_bytes) buf[i] = *reinterpret_cast<const v4u *>(run0 - mis + off);
            }
            return;
        }
        // where the batch's rows lie: lane l looks up row row0 + l once (B <= 64), every lane then takes its row's with a shuffle
        uint64_t place = 0ull;
        {
            const uint64_t row = min(row0 + (uint64_t)lane, (uint64_t)a.n_variants - 1ull);  // rows past the end re-load the last row
            place = HAS_VIDX ? (uint64_t)(gathered_record(a, row) - a.records) : row * a.record_stride;
        }
#pragma unroll
        for (int i = 0; i < kMaxBatchRows; i++) {
            buf[i] = v4u{0u, 0u, 0u, 0u};
            if ((uint32_t)i * G < B) {
                const uint32_t bi_row = (uint32_t)i * G + grp;           // row of the batch this lane works for in instruction i
                const uint64_t off = (uint64_t)__shfl((unsigned long long)place, (int)min(bi_row, 63u), 64);
                if (bi_row < B && piece < p.pieces) __builtin_memcpy(&buf[i], a.records + off + piece_off, 16);
            }
        }
    };
    load_batch(bi);

    for (;;) {
        // ---- park this batch's rows (waits for their loads — and, gfx9 having one in-order vmcnt, for the
        // previous batch's stores: one drain per batch of ~32 KiB of text or 12 rows)
        const uint64_t row0 = (uint64_t)bi * B;
        const uint32_t rows_here = (uint32_t)min((uint64_t)B, (uint64_t)a.n_variants - row0);
        // staged row i starts at rows0 + i * pitch (PACKED: pitch = R behind the run's misalignment)
        const uint8_t *const rows0 = stage + (packed ? (uint32_t)(uintptr_t)(a.records + row0 * (uint64_t)R) & 15u : 0u);
        if (packed) {
            const uint32_t n_bytes = (uint32_t)(rows0 - stage) + rows_here * R;
#pragma unroll
            for (int i = 0; i < kMaxBatchRows; i++)
                if ((uint32_t)i * 1024u < n_bytes) *reinterpret_cast<v4u *>(stage + (uint32_t)i * 1024u + lane * 16u) = buf[i];
        }
#pragma unroll
        for (int i = 0; i < kMaxBatchRows; i++) {
            if (!packed && (uint32_t)i * G + grp < B && piece < p.pieces) {
                v4u x = buf[i];
                if (tail_shift != 0u) {
                    uint64_t lo = (uint64_t)x.x | ((uint64_t)x.y << 32), hi = (uint64_t)x.z | ((uint64_t)x.w << 32);
                    const uint32_t sh8 = tail_shift * 8u;  // < 128: a lane with a whole piece outside the record does not exist (lane < pieces)
                    if (sh8 >= 64u) { lo = hi >> (sh8 - 64u); hi = 0ull; }
                    else { lo = (lo >> sh8) | (hi << (64u - sh8)); hi >>= sh8; }
                    x = v4u{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
                }
                *reinterpret_cast<v4u *>(stage + ((uint32_t)i * G + grp) * p.pitch + piece * 16u) = x;
            }
        }
        const uint64_t text_off_cur = text_off;   // (this batch's; load_batch below overwrites it with the next batch's)
        const uint32_t bi_next = bi + batch_step;
        const bool more = bi_next < p.n_batches;
        if (more) load_batch(bi_next);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        if (LINES) {
            // ---- full-line mode: the rows' GT segments sit behind their prefixes, so every row is flushed on its own
            // (whole aligned chunks + one byte-store instruction for its two edges; flush_codes, gt_common.hip.h)
            for (uint32_t i = 0; i < rows_here; i++) {
                const uint8_t *row = rows0 + i * p.pitch;
                const uint16_t *idx = s_idx;
                const uint64_t off = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(text_off_cur >> 32), (int)i) << 32) |
                                     (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)text_off_cur, (int)i);
                flush_codes([row, idx, K](uint32_t r) { return pick_code<IDENT>(row, idx, (int32_t)r, K); }, 0u, a.out + off, 0ull,
                            (uint64_t)S, 0u, K, lane);
            }
            if (!more) break;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            bi = bi_next;
            continue;
        }
        // ---- emit: one contiguous run of rows_here x S bytes
        const uint32_t len = rows_here * S;
        uint8_t *const run = a.out + row0 * (uint64_t)S;
        const uint32_t mis = (uint32_t)(uintptr_t)run & 15u;
        const uint32_t head = min((16u - mis) & 15u, len);     // bytes before the first whole chunk
        const uint32_t n_chunks = (len - head) >> 4;
        const uint32_t tail_off = head + (n_chunks << 4);
        const uint32_t tail = len - tail_off;                   // bytes after the last whole chunk (< 16)
        // the heads of the batch's rows 1 .. rows_here-1 (lane i: row i), once per batch: the chunk loop below then builds a
        // row-crossing chunk from ONE 16-byte text of this row, five dwords of the next row's head and a byte merge instead of
        // a second five-genotype gather (that path runs in every store step on rows of ~1 KB; in-process A/B on the chr22 shape:
        // 5 % kept 0.514 -> 0.545 of roofline, 10 % 0.512 -> 0.563, 20 % 0.565 -> 0.577, >= 30 % +1 %)
        uint32_t *const heads = s_heads[wave];
        if (S >= 17u && lane >= 1u && lane < rows_here) {
            const uint8_t *hrow = rows0 + lane * p.pitch;
            v4u ht;
            ht.x = gt_text(pick_code<IDENT>(hrow, s_idx, 0, K));
            ht.y = gt_text(pick_code<IDENT>(hrow, s_idx, 1, K));
            ht.z = gt_text(pick_code<IDENT>(hrow, s_idx, 2, K));
            ht.w = gt_text(pick_code<IDENT>(hrow, s_idx, 3, K));
            *reinterpret_cast<v4u *>(heads + lane * 8u + 4u) = ht;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (S < 17u) {
            // K = 1, 2, 3 (`--include-sam 'IID == "NA20900"'`, /root/reference/README.md:16-21): rows of 5 / 9 / 13 bytes, a 16-byte
            // chunk spans up to four of them, so it is built byte by byte.  The launch is a record READER here (626 bytes in for 5
            // out at N = 2 504): what counts is that the batch's records arrive as one wide run and its text leaves as whole chunks.
            for (uint32_t c = lane; c < n_chunks; c += 64u) {
                const uint32_t o = head + (c << 4);
                uint32_t i, pos;
                split_offset(o, p, i, pos);
                uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (uint32_t t = 0; t < 16u; t++) {
                    const uint32_t code = pick_code<IDENT>(rows0 + i * p.pitch, s_idx, (int32_t)(pos >> 2), K);  // (rank K under '\n': table slack)
                    const uint32_t byte = pos == S - 1u ? 0x0Au : gt_text_byte(code, pos & 3u);
                    w[t >> 2] |= byte << (8u * (t & 3u));
                    if (++pos == S) { pos = 0u; i++; }   // a whole chunk lies inside the run: i stays < rows_here
                }
                v4u t4 = {w[0], w[1], w[2], w[3]};
                *reinterpret_cast<v4u *>(run + o) = t4;
            }
        } else
        // (lanes <-> chunks shifted by `lead` so that every store instruction covers eight WHOLE 128-byte lines: one that starts
        // mid-line touches nine, two of them partially — 5 % on write-dominated launches, profiles/r03_kernel_sweeps.md §8)
        for (uint32_t c0 = 0, lead = p.align_stores ? ((uint32_t)(uintptr_t)(run + head) >> 4) & 7u : 0u; c0 < n_chunks + lead; c0 += 64u) {
            const uint32_t c = c0 + lane - lead;                // (wraps for the lanes in front of chunk 0)
            if (c >= n_chunks) continue;
            const uint32_t o = head + (c << 4);
            uint32_t i, pos;
            split_offset(o, p, i, pos);
            const uint8_t *row = rows0 + i * p.pitch;
            u32x4 v = pick_text16<IDENT>(row, s_idx, (int32_t)pos, K);
            const uint32_t nl = S - 1u - pos;                   // chunk byte of this row's '\n' if < 16
            if (nl < 16u) {
                // the chunk holds '\n' at byte nl and the head of the next row behind it (a whole chunk never ends the run)
                u32x4 y = {0u, 0u, 0u, 0u};
                if (nl < 15u) {
                    // bytes nl+1 .. 15 of the chunk = bytes 0 .. of row i+1: its head shifted up by sh = nl + 1 bytes (zeros come in below)
                    const uint32_t sh = nl + 1u, bsh = sh & 3u;
                    const uint32_t *e = heads + (i + 1u) * 8u + 3u - (sh >> 2);
                    const uint32_t e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3], e4 = e[4];
                    y.x = bsh ? funnel_bytes(e0, e1, 4u - bsh) : e1;
                    y.y = bsh ? funnel_bytes(e1, e2, 4u - bsh) : e2;
                    y.z = bsh ? funnel_bytes(e2, e3, 4u - bsh) : e3;
                    y.w = bsh ? funnel_bytes(e3, e4, 4u - bsh) : e4;
                }
                uint32_t xs[4] = {v.x, v.y, v.z, v.w};
                uint32_t ys[4] = {y.x, y.y, y.z, y.w};
                uint32_t os[4];
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const int32_t nb = (int32_t)nl - 4 * m;     // bytes of dword m taken from this row
                    const uint32_t mask = nb >= 4 ? 0xFFFFFFFFu : (nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u));
                    uint32_t d = (xs[m] & mask) | (ys[m] & ~mask);
                    if (nb >= 0 && nb < 4) d = (d & ~(0xFFu << (8 * nb))) | (0x0Au << (8 * nb));
                    os[m] = d;
                }
                v = u32x4{os[0], os[1], os[2], os[3]};
            }
            v4u t = {v.x, v.y, v.z, v.w};
            subset_store16(run + o, gt_v4u{t.x, t.y, t.z, t.w});
        }
        {
            // edge bytes of the run: lanes 0-15 the head, lanes 16-31 the tail
            const uint32_t o = lane < 16u ? lane : tail_off + (lane - 16u);
            const bool on = lane < 16u ? lane < head : (lane < 32u && lane - 16u < tail);
            if (on) {
                uint32_t i, pos;
                split_offset(o, p, i, pos);
                const uint32_t code = pick_code<IDENT>(rows0 + i * p.pitch, s_idx, (int32_t)(pos >> 2), K);
                run[o] = (uint8_t)(pos == S - 1u ? 0x0Au : gt_text_byte(code, pos & 3u));
            }
        }
        if (!more) break;
        // the stage is rewritten by the next batch: this batch's reads must have returned first
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bi = bi_next;
    }
}


// ---------------------------------------------------------------------------------------------
// gt_pick_lines_kernel (round 3) — FULL LINES (pgenhip_emit_lines, src/pfile.rs:156-192) of short DENSE records, kept subset or
// all samples: the CLI with `--include-sam` on the 1000 Genomes shape.
//
// Round 2 flushed every row of a batch on its own behind its prefix (whole chunks + one byte-store instruction for the row's two
// ragged ends) and copied the prefixes separately.  SQ counters at N = 2 504 with 10 % kept (profiles/r03_kernel_sweeps.md): 94 VALU
// per KiB of output against 145 in GT-segment mode — and still 0.44 of roofline against 0.60, 52 % of the wave cycles waiting: every
// row left three separately written partial 128-B lines behind (its head, its tail, its prefix).
//
// Lines are packed back to back, so between the last WHOLE 16-byte chunk of row r's GT text and the first whole chunk of row r+1's
// lies a SEAM that is itself a whole number of chunks: [last GT bytes of r | '\n' | prefix of r+1 | first GT bytes of r+1].  A batch
// now writes (1) every row's interior as whole chunks (one pass of 64 lanes per KiB, the pick of gt_pick_kernel) and (2) all its
// seams in a few wave passes, 4 .. 16 seams per pass by the launch's longest prefix, each lane building one chunk of one seam from
// the two rows' texts and the prefix bytes (the batch's piece of the prefix blob is staged in LDS with the records).  Every byte
// is written once, as part of a whole chunk; only the launch's first and last chunk can be ragged.
// A batch owns the seams BEHIND its rows, so it stages one record more than it has rows (the row behind it), and the launch's first
// batch also owns the piece in front of row 0.  Offsets are fetched two batches ahead (lane <-> row), so that the prefix bytes of
// the next batch can be requested together with its records.
constexpr uint32_t kPfxBytes = 2048;                         // prefix bytes of a batch (+ the row behind it) that the stage holds
constexpr uint32_t kPfxStage = 16 + 16 + kPfxBytes + 16;     // 16 of slack in front, 15 of misalignment, 16 of slack behind

struct PickLinesParams {
    uint32_t batch_rows;   // B <= 62 (lanes 0 .. B + 1 hold the offsets of rows 0 .. B + 1 of the batch)
    uint32_t n_batches;
    uint32_t cps_shift;    // log2 of the lanes (= chunk slots) per seam: 2 .. 6
};

template <bool IDENT>
__global__ __launch_bounds__(kThreads) void gt_pick_lines_kernel(EmitArgs a, PickLinesParams p)
{
    __shared__ uint16_t s_tab[kPadBefore + kMaxSamples + kPadAfter];
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[kWaves][kStageBytes];
    __shared__ __attribute__((aligned(16))) uint8_t s_pfx[kWaves][kPfxStage];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t K = a.kept_count;
    if (!IDENT) {
        for (uint32_t r = tid; r < kPadBefore + K + kPadAfter; r += (uint32_t)kThreads)
            s_tab[r] = r >= kPadBefore && r < kPadBefore + K ? pick_entry(a.kept_idx[r - kPadBefore]) : (uint16_t)0;
    }
    __syncthreads();
    const uint16_t *const s_idx = s_tab + kPadBefore;
    uint8_t *const stage = s_stage[wave];
    uint8_t *const pstage = s_pfx[wave];
    const uint32_t R = a.record_size, B = p.batch_rows;
    const uint64_t V = a.n_variants;
    const int32_t N4 = (int32_t)(4u * K);
    const uint32_t step = gridDim.x * kWaves;
    uint32_t bi = blockIdx.x * kWaves + wave;
    if (bi >= p.n_batches) return;

    // lane l: line and prefix offsets of row b * B + l (both arrays have V + 1 entries)
    auto load_offs = [&](uint32_t b, uint64_t &lo, uint64_t &po) {
        const uint64_t jr = min((uint64_t)b * B + (uint64_t)lane, V);
        lo = a.line_off[jr];
        po = a.prefix_off[jr];
    };
    auto lane64 = [&](uint64_t v, uint32_t l) {   // wave-uniform lane index
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
    };
    // the records of rows b * B .. b * B + B (the row behind the batch comes along) as one byte run, from the 16-B boundary below it
    v4u rbuf[kStageBytes / 1024];
    auto load_records = [&](uint32_t b) {
        const uint64_t row0 = (uint64_t)b * B;
        const uint8_t *__restrict__ run0 = a.records + row0 * (uint64_t)R;
        const uint32_t mis = (uint32_t)(uintptr_t)run0 & 15u;
        const uint32_t n_bytes = mis + (uint32_t)min((uint64_t)B + 1ull, V - row0) * R;   // <= kStageBytes (host)
#pragma unroll
        for (uint32_t i = 0; i < kStageBytes / 1024u; i++) {
            rbuf[i] = v4u{0u, 0u, 0u, 0u};
            const uint32_t off = i * 1024u + lane * 16u;
            if (i * 1024u < n_bytes && off < n_bytes) rbuf[i] = *reinterpret_cast<const v4u *>(run0 - mis + off);
        }
    };
    // the prefixes of the same rows: blob bytes [po(row 0), po(row n)), from the 16-B boundary below them
    v4u pbuf[2];
    auto load_blob = [&](uint32_t b, uint64_t po) {
        const uint32_t n = (uint32_t)min((uint64_t)B + 1ull, V - (uint64_t)b * B);
        const uint64_t p0 = lane64(po, 0u), p1 = lane64(po, n);
        const uint8_t *__restrict__ src = a.prefix_blob + p0;
        const uint32_t mis = (uint32_t)(uintptr_t)src & 15u;
        const uint32_t n_bytes = mis + (uint32_t)(p1 - p0);                                // <= 15 + kPfxBytes (host)
#pragma unroll
        for (uint32_t i = 0; i < 2u; i++) {
            pbuf[i] = v4u{0u, 0u, 0u, 0u};
            const uint32_t off = i * 1024u + lane * 16u;
            if (p1 > p0 && off < n_bytes) pbuf[i] = *reinterpret_cast<const v4u *>(src - mis + off);
        }
    };

    uint64_t lo0, po0, lo1 = 0ull, po1 = 0ull, lo2 = 0ull, po2 = 0ull;
    load_offs(bi, lo0, po0);
    if (bi + step < p.n_batches) load_offs(bi + step, lo1, po1);
    load_records(bi);
    load_blob(bi, po0);
    for (;;) {
        const uint64_t row0 = (uint64_t)bi * B;
        const uint32_t rows_here = (uint32_t)min((uint64_t)B, V - row0);
        // ---- park the records and the prefix bytes (waits for their loads and, one in-order vmcnt, for the previous batch's stores)
        const uint32_t rmis = (uint32_t)(uintptr_t)(a.records + row0 * (uint64_t)R) & 15u;
        {
            const uint32_t n_bytes = rmis + (uint32_t)min((uint64_t)B + 1ull, V - row0) * R;
#pragma unroll
            for (uint32_t i = 0; i < kStageBytes / 1024u; i++)
                if (i * 1024u < n_bytes) *reinterpret_cast<v4u *>(stage + i * 1024u + lane * 16u) = rbuf[i];
        }
        const uint64_t pbase = lane64(po0, 0u);
        const uint32_t pmis = (uint32_t)(uintptr_t)(a.prefix_blob + pbase) & 15u;
#pragma unroll
        for (uint32_t i = 0; i < 2u; i++) *reinterpret_cast<v4u *>(pstage + 16u + i * 1024u + lane * 16u) = pbuf[i];
        const uint8_t *const rows0 = stage + rmis;              // staged row i at rows0 + i * R
        const uint8_t *const pfx0 = pstage + 16u + pmis;        // prefix byte t of the batch (relative to row 0's prefix start) at pfx0[t]
        // ---- this batch's geometry, relative to its first line (32 bits: a batch is at most 63 lines)
        const uint64_t lbase = lane64(lo0, 0u);
        const uint32_t lrel = (uint32_t)(lo0 - lbase);          // lane l: start of line l
        const uint32_t prel = (uint32_t)(po0 - pbase);          //         start of its prefix in the batch's prefix bytes
        const uint32_t plen = (uint32_t)__shfl_down((int)prel, 1, 64) - prel;   // (every lane takes part in the shuffle)
        const uint32_t grel = lrel + plen;                      //         first byte of its GT text
        uint8_t *const out0 = a.out + lbase;
        const uint32_t A0 = (uint32_t)(uintptr_t)out0 & 15u;
        auto adown = [&](int32_t x) { return x - (int32_t)(((uint32_t)x + A0) & 15u); };
        auto aup = [&](int32_t x) { return x + (int32_t)((0u - ((uint32_t)x + A0)) & 15u); };
        // ---- the next batch's loads (its offsets arrived a batch ago), and the offsets of the batch after it
        const uint32_t bi_next = bi + step;
        const bool more = bi_next < p.n_batches;
        if (more) {
            load_records(bi_next);
            load_blob(bi_next, po1);
            if (bi_next + step < p.n_batches) load_offs(bi_next + step, lo2, po2);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- (1) the rows' interiors: whole chunks inside [g, g + 4K)
        for (uint32_t i = 0; i < rows_here; i++) {
            const int32_t g = (int32_t)(uint32_t)__builtin_amdgcn_readlane((int)grel, (int)i);
            const int32_t q0 = aup(g) - g;                                       // 0 .. 15
            const int32_t n_int = (N4 - q0) >> 4;
            const uint8_t *row = rows0 + i * R;
            for (int32_t c = (int32_t)lane; c < n_int; c += 64) {
                const int32_t q = q0 + 16 * c;
                const u32x4 v = pick_text16<IDENT>(row, s_idx, q, K);
                *reinterpret_cast<v4u *>(out0 + g + q) = v4u{v.x, v.y, v.z, v.w};
            }
        }
        // ---- (2) the seams behind rows 0 .. rows_here - 1 (and, first batch of the launch, the piece in front of row 0: seam -1)
        const int32_t s_first = row0 == 0ull ? -1 : 0;
        const uint32_t cps = 1u << p.cps_shift, spp = 64u >> p.cps_shift;
        for (int32_t s_pass = s_first; s_pass < (int32_t)rows_here; s_pass += (int32_t)spp) {
            const int32_t s = s_pass + (int32_t)(lane >> p.cps_shift);         // this lane's seam: between rows s and s + 1 of the batch
            const uint32_t sl = (uint32_t)max(s, 0), sr = (uint32_t)(s + 1);
            // (shuffles outside every branch: all lanes take part)
            const int32_t g_l = (int32_t)__shfl((int)grel, (int)sl, 64);
            const int32_t g_r = (int32_t)__shfl((int)grel, (int)sr, 64);
            const int32_t l_r = (int32_t)__shfl((int)lrel, (int)sr, 64);
            const int32_t p_r = (int32_t)__shfl((int)prel, (int)sr, 64);
            const bool l_valid = s >= 0;
            const bool r_valid = row0 + (uint64_t)sr < V;
            const int32_t e_l = g_l + N4;                                        // row s's '\n'
            const int32_t S0 = l_valid ? adown(e_l) : adown(l_r);
            const int32_t S1 = r_valid ? aup(g_r) : aup(e_l + 1);
            const int32_t V0 = l_valid ? S0 : l_r, V1 = r_valid ? S1 : e_l + 1;  // bytes of the seam that exist (ragged only at the launch's two ends)
            const int32_t c0 = S0 + 16 * (int32_t)(lane & (cps - 1u));
            if (s < (int32_t)rows_here && c0 < S1) {
                const int32_t n1 = l_valid ? e_l - c0 : -1;                      // chunk bytes [0, n1): row s's text; byte n1: its '\n'
                const int32_t n2 = r_valid ? g_r - c0 : 16;                      // chunk bytes (n1, n2): row s+1's prefix; [n2, 16): its text
                u32x4 T = {0u, 0u, 0u, 0u}, H = {0u, 0u, 0u, 0u};
                if (n1 > 0) T = pick_text16<IDENT>(rows0 + sl * R, s_idx, c0 - g_l, K);
                if (n2 < 16) H = pick_text16<IDENT>(rows0 + sr * R, s_idx, c0 - g_r, K);
                uint32_t P[4] = {0u, 0u, 0u, 0u};
                if (r_valid && n2 > 0 && n1 < 15) {
                    const uint8_t *pb = pfx0 + p_r + (c0 - l_r);                 // (>= 16 bytes of slack on both sides of the staged prefixes)
#pragma unroll
                    for (int t = 0; t < 16; t++) P[t >> 2] |= (uint32_t)pb[t] << (8 * (t & 3));
                }
                const uint32_t Ts[4] = {T.x, T.y, T.z, T.w}, Hs[4] = {H.x, H.y, H.z, H.w};
                uint32_t o[4];
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const int32_t a1 = n1 - 4 * m, a2 = n2 - 4 * m;
                    const uint32_t m1 = a1 >= 4 ? 0xFFFFFFFFu : (a1 <= 0 ? 0u : ((1u << (8 * a1)) - 1u));
                    const uint32_t m2 = a2 >= 4 ? 0xFFFFFFFFu : (a2 <= 0 ? 0u : ((1u << (8 * a2)) - 1u));
                    uint32_t d = (Ts[m] & m1) | (P[m] & ~m1 & m2) | (Hs[m] & ~m2);
                    if (a1 >= 0 && a1 < 4) d = (d & ~(0xFFu << (8 * a1))) | (0x0Au << (8 * a1));   // (a1 < 0 for every m when there is no row s)
                    o[m] = d;
                }
                uint8_t *const dst = out0 + c0;
                if (c0 >= V0 && c0 + 16 <= V1) {
                    *reinterpret_cast<v4u *>(dst) = v4u{o[0], o[1], o[2], o[3]};
                } else {
#pragma unroll
                    for (int t = 0; t < 16; t++)
                        if (c0 + t >= V0 && c0 + t < V1) dst[t] = (uint8_t)(o[t >> 2] >> (8 * (t & 3)));
                }
            }
        }
        if (!more) break;
        // the stages are rewritten by the next batch: this batch's reads must have returned first
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bi = bi_next;
        lo0 = lo1; po0 = po1;
        lo1 = lo2; po1 = po2;
    }
}

}  // namespace

bool gt_pick_applicable(const EmitArgs &a)
{
    // kept subset, record of one tile (16 <= R <= 1024), at least one kept sample (rows of >= 17 bytes: a 16-B chunk touches at
    // most two rows; K = 1, 2, 3: up to four, built byte by byte), dense output pitch or full-line mode (rows then go out one
    // by one behind their prefixes)
    // (kept_idx == NULL: all samples kept — the same kernel with the identity in place of the table)
    return a.sample_count <= kMaxSamples && a.record_size >= 16u && a.kept_count >= 1u &&
           (a.line_off != nullptr || a.n_variants <= 1u || a.out_stride == 4ull * a.kept_count + 1ull);
}

hipError_t launch_gt_pick(const EmitArgs &a, const Tuning &t, int num_cus, hipStream_t stream)
{
    if (a.n_variants == 0) return hipSuccess;
    PickParams p;
    p.pieces = (a.record_size + 15u) / 16u;
    p.pitch = p.pieces * 16u;
    p.row_bytes = 4u * a.kept_count + 1u;
    p.pfx_shift = prefix_copy_shift(a);
    p.align_stores = t.align_stores != 0 ? 1u : 0u;
    // rows per batch: ~32 KiB of text per batch (8 / 16 / 32 / 64 KiB at 50 % kept on the chr22 shape: 1.44 / 1.41 / 1.37 / 1.36 ms),
    // what the stage holds, what the register buffer holds
    const uint32_t batch_bytes = t.pick_batch_bytes > 0 ? (uint32_t)t.pick_batch_bytes : 32768u;
    uint32_t b = (batch_bytes + p.row_bytes - 1u) / p.row_bytes;
    b = b < 1u ? 1u : b;
    // dense records and no gather: the batch's records are one contiguous run (rows at pitch R, up to 64 of them)
    p.packed = !gathered(a) && (a.n_variants <= 1u || a.record_stride == a.record_size) ? 1u : 0u;
    if (p.packed) {
        p.pitch = a.record_size;
        if (b > kMaxPackedRows) b = kMaxPackedRows;
        if (b > (kStageBytes - 16u) / p.pitch) b = (kStageBytes - 16u) / p.pitch;  // 15 bytes of misalignment + B * R <= 8 192: >= 7 rows, <= 8 loads
    } else {
        // row-by-row loads, G rows per instruction: lanes per row = the record's pieces rounded up to a power of two
        uint32_t lpr = 1u;
        while (lpr < p.pieces) lpr <<= 1;
        p.pitch = lpr * 16u;
        const uint32_t rows_per_load = 64u / lpr;
        if (b > (uint32_t)kMaxBatchRows * rows_per_load) b = (uint32_t)kMaxBatchRows * rows_per_load;
        if (b > kMaxPackedRows) b = kMaxPackedRows;
        if (b > kStageBytes / p.pitch) b = kStageBytes / p.pitch;  // >= 8 (pitch <= 1024)
    }
    p.batch_rows = b;
    p.magic = (uint32_t)(0x100000000ull / p.row_bytes) + 1u;    // exact up to one compare for run offsets < 2^20 (<= 12 rows of <= 16 385 bytes, or <= 64 rows within 32 KiB + one row)
    p.n_batches = (uint32_t)(((uint64_t)a.n_variants + b - 1u) / b);
    // full lines of dense records: interiors + seams, every byte written once as part of a whole chunk (gt_pick_lines_kernel)
    if (a.line_off != nullptr && p.packed && t.pick_line_seams != 0 && a.kept_count >= 4u && a.prefix_blob != nullptr) {
        const uint64_t max_prefix = a.max_line_bytes - (uint64_t)p.row_bytes;
        PickLinesParams lp;
        uint32_t bl = b;
        if (bl > 62u) bl = 62u;
        if (15u + (bl + 1u) * a.record_size > kStageBytes) bl = (kStageBytes - 15u) / a.record_size - 1u;           // the row behind the batch comes along
        if (max_prefix != 0ull && (uint64_t)(bl + 1u) * max_prefix > kPfxBytes - 16u) bl = (uint32_t)((kPfxBytes - 16u) / max_prefix) - 1u;   // and its prefix (two 1-KiB loads from the 16-B boundary below)
        const uint32_t seam_chunks = (uint32_t)((max_prefix + 31ull) / 16ull);                                         // a seam is at most P + 31 bytes
        lp.cps_shift = seam_chunks <= 4u ? 2u : (seam_chunks <= 8u ? 3u : (seam_chunks <= 16u ? 4u : (seam_chunks <= 32u ? 5u : 6u)));
        if (max_prefix <= 993ull && (int32_t)bl >= 2 && bl <= 62u) {
            lp.batch_rows = bl;
            lp.n_batches = (uint32_t)(((uint64_t)a.n_variants + bl - 1u) / bl);
            void (*lk)(EmitArgs, PickLinesParams) = a.kept_idx == nullptr ? gt_pick_lines_kernel<true> : gt_pick_lines_kernel<false>;
            int lper_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&lper_cu, lk, kThreads, 0) != hipSuccess || lper_cu < 1) lper_cu = 1;
            const uint64_t lneed = ((uint64_t)lp.n_batches + kWaves - 1ull) / kWaves;
            const uint64_t lcap = (uint64_t)lper_cu * (uint64_t)num_cus;
            hipLaunchKernelGGL(lk, dim3((uint32_t)(lneed < lcap ? lneed : lcap)), dim3(kThreads), 0, stream, a, lp);
            return hipGetLastError();
        }
    }
    void (*kern)(EmitArgs, PickParams);
    if (a.kept_idx == nullptr) {
        if (a.line_off)
            kern = gathered(a) ? gt_pick_kernel<true, true, true> : gt_pick_kernel<false, true, true>;
        else
            kern = gathered(a) ? gt_pick_kernel<true, false, true> : gt_pick_kernel<false, false, true>;
    } else if (a.line_off)
        kern = gathered(a) ? gt_pick_kernel<true, true, false> : gt_pick_kernel<false, true, false>;
    else
        kern = gathered(a) ? gt_pick_kernel<true, false, false> : gt_pick_kernel<false, false, false>;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kThreads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    const uint64_t need = ((uint64_t)p.n_batches + kWaves - 1ull) / kWaves;
    const uint64_t cap = (uint64_t)per_cu * (uint64_t)num_cus;
    hipLaunchKernelGGL(kern, dim3((uint32_t)(need < cap ? need : cap)), dim3(kThreads), 0, stream, a, p);
    return hipGetLastError();
}

}  // namespace pgenhip
