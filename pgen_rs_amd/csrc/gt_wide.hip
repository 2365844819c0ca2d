// gt_wide.hip — the work-queue STREAM kernel: all samples kept, dense text, wide LDS-staged record loads (gfx950 / MI355X).
//
// Same contract and output-stream view as gt_flat.hip (K = N, rows packed at 4N+1 bytes; every lane stores one
// 16-byte-ALIGNED chunk of the launch's output stream), but the 2-bit words are no longer fetched with two byte loads per
// chunk.  Measured on MI355X (tools/membench): a kernel that pairs each 1-KiB wave store with a 64-byte wave load tops out
// at ~4.3 TB/s whatever the pipelining depth, while one 1-KiB wide load feeding sixteen 1-KiB stores reaches ~5.3 TB/s —
// few, large read bursts disturb the HBM write stream far less than many small ones.
//
// One kernel (gt_stream_dyn_kernel), three kinds of work item:
//   * ROW ITEMS  (row j, span k): up to 1 024 consecutive chunks (16 KiB of text) that start inside row j — N >= 1 916;
//   * LINES      the same items with every row's GT segment behind its own prefix (pgenhip_emit_lines, N >= 1 024);
//   * RUNS       a run of B consecutive SHORT rows as one item (8 <= N <= 1 915, dense records): one wide load for the
//                run's contiguous records, the text staged through LDS in 4-KiB groups so that the row-crossing chunks are
//                built in one pass and every 128-B line leaves whole.
// and its sibling for FULL LINES of short rows (gt_lineruns_kernel, N < 1 024): runs of whole lines — prefixes, GT text, '\n' —
// assembled in the same kind of LDS stage, one loader wave per three storer waves.
// Roles inside a 512-thread block: wave 0 only loads (16 B per lane per item -> LDS slab ring), waves 1-7 only store
// (ds_read_u16 window -> text -> global_store_dwordx4 nt, 1 KiB of contiguous text per wave instruction); items come from a
// self-cleaning work queue in global memory.  Details at each piece below.
#include <algorithm>

#include "gt_common.hip.h"
#include "kernels.h"

namespace pgenhip {

namespace {

constexpr uint32_t kSpanChunks = 1024;  // chunks per work item: 16 stores x 64 lanes
constexpr uint32_t kSlabBytes = 1088;   // 66 lanes x 16 B staged at most, rounded to 64
constexpr uint32_t kSlabExtra = 16;     // stream kernels: +0 u8 = first record byte of row j+1 (for the chunk holding row j's '\n')
// Item descriptors travel in their own LDS ring (ring slots + 1 per storer, 64 B each), so the
// loader can write them while it issues the loads, before it has to wait for the slab slot to be free:
//   +8  u64 item index t (work-queue kernel; ~0 = no item, ~0-1 = launch is out of work)
//   +16 u64 g0   +24 i64 c_first   +32 u64 row   +40 u32 cnt   +44 u32 lead   +48 i32 delta
// and a storer wave does not redo the loader's 64-bit scalar item arithmetic.
constexpr uint32_t kDescBytes = 64;

struct WideParams {
    uint64_t row_bytes;      // S = 4N + 1
    uint64_t total_bytes;    // T = V * S
    uint64_t n_items;        // V * spans_per_row
    uint32_t spans_per_row;
    uint32_t head;           // out address & 127: the chunk grid is anchored at a 128-B line boundary
    uint32_t n_ranges;       // work-queue kernel: contiguous item ranges with a head word each (1, 2, 4 or 8)
    uint32_t run_rows;       // RUNS mode: rows per work item (B)
    uint32_t run_rec;        // RUNS mode: bytes per row in the slab (R)
    uint32_t magic;          // RUNS mode: floor(2^32 / S) + 1: o / S = umulhi(o, magic), fixed up by one compare (o < 2^14)
    uint32_t pfx_shift;      // LINES through the stream kernel: log2 of the lanes per line of the in-kernel prefix copy (0 = no prefixes to copy)
};

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <bool HAS_VIDX>
__device__ __forceinline__ const uint8_t *row_record(const EmitArgs &a, uint64_t r)
{
    return HAS_VIDX ? gathered_record(a, r) : a.records + r * a.record_stride;
}

template <bool NT>
__device__ __forceinline__ void store_chunk(uint8_t *dst, const u32x4 &v)
{
    v4u t = {v.x, v.y, v.z, v.w};
    if (NT)
        __builtin_nontemporal_store(t, reinterpret_cast<v4u *>(dst));
    else
        *reinterpret_cast<v4u *>(dst) = t;
}

// geometry of one work item (all wave-uniform)
struct Item {
    uint64_t g0;          // first chunk (index into the aligned chunk space of the stream)
    uint32_t cnt;         // chunks in this item (0 = nothing to do)
    uint32_t lead;        // g0 - span base: store step u covers chunks base + 64u .. +63 (1-KiB aligned)
    int64_t c_first;      // row-relative byte offset of chunk g0's first byte (>= -15)
    const uint8_t *rec;   // record of row j
    const uint8_t *base;  // 16-B-aligned address the staged bytes start at
    uint32_t n_load;      // 16-byte pieces to stage (<= 66)
    int32_t delta;        // slab offset of record byte 0 (= rec - base)
    uint64_t row;         // j
    bool row_tail;        // the item holds the row's last chunk (the one with its '\n')
};

// Item (row j, span k).  Only the row's place in the stream needs 64-bit arithmetic; everything
// else is computed RELATIVE to the row's first chunk in 32 bits (a row has at most 2^30 + 1 chunks),
// because gfx9's scalar unit has no 64-bit ordered compare — every 64-bit min/max becomes a VALU
// compare plus a round trip through VCC, and the loader wave computes seven items per step.
// LINES (full VCF body lines, src/pfile.rs:156-192): row j's GT segment starts behind its own prefix, at
// line_off[j] + prefix length — any byte offset — and the bytes around it belong to the prefixes, so a
// row owns the chunk that holds its first GT byte (like row 0 of the dense stream) and writes only
// its own bytes of its first and last chunk.
// (`lines_row_start` = line_off[j] + prefix length, fetched by the caller: one coalesced load per step for all of
// a step's rows instead of two dependent round trips per item)
template <bool HAS_VIDX, bool LINES = false>
__device__ __forceinline__ Item make_item_at(const EmitArgs &a, const WideParams &p, uint64_t j, uint32_t k, uint64_t lines_row_start = 0ull,
                                             const uint8_t *rec_known = nullptr)
{
    Item it;
    const uint64_t S = p.row_bytes;
    const uint64_t row_start = LINES ? lines_row_start : j * S;
    // row j owns the chunks whose first byte lies in [j*S, (j+1)*S); row 0 also the chunk that
    // holds stream byte 0.  Spans are cut on 64-chunk (1 KiB) boundaries of the chunk grid so every
    // store instruction of a wave covers eight WHOLE 128-B lines (except at the two row ends).
    const uint64_t g_first = (LINES || j == 0) ? (row_start + p.head) >> 4 : (row_start + p.head + 15ull) >> 4;
    const uint64_t g_end = (row_start + S + p.head + 15ull) >> 4;
    const int32_t row_chunks = (int32_t)(uint32_t)(g_end - g_first);
    const int32_t base_rel = (int32_t)(k * kSpanChunks) - (int32_t)((uint32_t)g_first & 63u);  // span base - g_first
    const int32_t end_rel = min(row_chunks, base_rel + (int32_t)kSpanChunks);
    const int32_t g0_rel = max(base_rel, 0);
    // row-relative byte offset of chunk g_first: 0..15 (row 0 and LINES: -15..0); exact modulo 2^32
    const int32_t row_c_first = (int32_t)((uint32_t)g_first * 16u - p.head - (uint32_t)row_start);
    it.row = j;
    it.g0 = g_first + (uint64_t)(uint32_t)g0_rel;
    it.cnt = end_rel > g0_rel ? (uint32_t)(end_rel - g0_rel) : 0u;
    it.lead = (uint32_t)(g0_rel - base_rel);
    it.row_tail = it.cnt != 0u && end_rel == row_chunks;  // <=> c_first + 16 cnt >= S
    it.c_first = (int64_t)row_c_first + ((int64_t)g0_rel << 4);
    it.rec = HAS_VIDX ? rec_known : a.records + j * a.record_stride;   // (gathered rows: the loader fetched the step's record places with one load and passes this row's)
    const int32_t last = (int32_t)(a.record_size - 1u);
    const int32_t bf = g0_rel + (row_c_first >> 4);  // = c_first >> 4: first record byte needed (-1 for the head chunk)
    const uint32_t b_first = bf > 0 ? (uint32_t)min(bf, last - 1) : 0u;  // R >= 2
    const uint32_t b_last = (uint32_t)min(bf + (int32_t)it.cnt, last);  // window hi byte of the last chunk
    const uint64_t addr_first = (uint64_t)(uintptr_t)(it.rec + b_first);
    const uint32_t mis = (uint32_t)(addr_first & 15ull);
    it.base = it.rec + b_first - mis;
    it.n_load = it.cnt ? (mis + (b_last - b_first)) / 16u + 1u : 0u;
    it.delta = (int32_t)mis - (int32_t)b_first;
    return it;
}


// ---- item descriptor hand-over through the slab's extra area (layout at kSlabExtra) ----------
// NB: __builtin_amdgcn_readfirstlane returns a signed int — widen through uint32_t, or a low half
// with bit 31 set sign-extends into the high half (found by the full-size config-3 test: g0 >= 2^31)
__device__ __forceinline__ uint32_t sgpr32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t sgpr64(uint64_t v)
{
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
    return ((uint64_t)hi << 32) | (uint64_t)lo;
}

__device__ __forceinline__ void desc_put_item(uint8_t *x, const Item &it, uint64_t t)
{
    *reinterpret_cast<uint64_t *>(x + 8) = t;
    *reinterpret_cast<uint64_t *>(x + 16) = it.g0;
    *reinterpret_cast<int64_t *>(x + 24) = it.c_first;
    *reinterpret_cast<uint64_t *>(x + 32) = it.row;
    *reinterpret_cast<uint32_t *>(x + 40) = it.cnt;
    *reinterpret_cast<uint32_t *>(x + 44) = it.lead;
    *reinterpret_cast<int32_t *>(x + 48) = it.delta;
}

__device__ __forceinline__ Item desc_get_item(const uint8_t *x)
{
    Item it;
    it.g0 = sgpr64(*reinterpret_cast<const uint64_t *>(x + 16));
    it.c_first = (int64_t)sgpr64((uint64_t)*reinterpret_cast<const int64_t *>(x + 24));
    it.row = sgpr64(*reinterpret_cast<const uint64_t *>(x + 32));
    it.cnt = sgpr32(*reinterpret_cast<const uint32_t *>(x + 40));
    it.lead = sgpr32(*reinterpret_cast<const uint32_t *>(x + 44));
    it.delta = (int32_t)sgpr32((uint32_t)*reinterpret_cast<const int32_t *>(x + 48));
    it.rec = nullptr;
    it.base = nullptr;
    it.row_tail = false;
    it.n_load = 0u;
    return it;
}

// One item's 16 store steps: lanes read their 10-bit window from the staged record bytes in
// `slab`, expand to text and store aligned 16-byte chunks.  NEXT_IN_SLAB: the first byte of row
// j+1 (needed by the chunk that holds row j's '\n') was parked at slab[kSlabBytes] by the loader
// wave, so this wave never issues a global load (gfx9 has one in-order vmcnt for loads and stores:
// a wave that waits for a load also drains all its older stores).
template <bool HAS_VIDX, bool NT, bool NEXT_IN_SLAB, bool LINES = false, int BURST = 1>
__device__ __forceinline__ void emit_item(const EmitArgs &a, const WideParams &p, const Item &it,
                                          const uint8_t *slab, uint32_t lane)
{
    const uint64_t S = p.row_bytes;
    const uint64_t gt_bytes = S - 1ull;
    const uint32_t last = a.record_size - 1u;
    uint8_t *const chunk0 = a.out - p.head;
    // ---- 16 store steps over the staged bytes
    const int32_t bf = (int32_t)(it.c_first >> 4);                 // record byte of chunk 0's window (>= -1)
    const int32_t delta = it.delta;                                 // slab offset of record byte 0
    const uint32_t phase = (uint32_t)it.c_first & 15u;             // same for every chunk of the row
    // chunks 0 .. n_interior-1 lie wholly inside the row's GT text
    const int64_t room = (int64_t)gt_bytes - 16 - it.c_first;      // c_first + 16*i + 16 <= gt_bytes
    const uint32_t n_interior = room < 0 ? 0u : (uint32_t)min((int64_t)it.cnt, room / 16 + 1);
    const bool head_chunk = it.c_first < 0;                        // only (row 0, span 0) with an unaligned out
    uint8_t *const span_ptr = chunk0 + (it.g0 - it.lead) * 16ull + lane * 16u;  // lane's chunk in step 0
    const uint32_t first_plain = it.lead + (head_chunk ? 1u : 0u);               // steps at/after this position ...
    const uint32_t end_plain = it.lead + n_interior;                              // ... and before this one are all-interior
    // the item holds the row's last chunk (the descriptor hand-over does not carry Item::row_tail)
    const bool row_tail = it.cnt != 0u && it.c_first + 16ll * (int64_t)it.cnt >= (int64_t)S;
    const uint32_t pshift = ((phase >> 2) & 3u) * 2u;                             // bit offset of sample k0 in its byte (row-uniform)
    const uint32_t psh = phase & 3u;                                              // byte phase of the text (row-uniform)
    const int32_t slab_b0 = bf + delta - (int32_t)it.lead + (int32_t)lane;        // slab offset of the lane's window in step 0
    for (uint32_t u = 0; u < kSpanChunks / 64u; u++) {
        if (u * 64u >= it.lead + it.cnt) break;
        if (BURST > 1 && u * 64u >= first_plain && (u + (uint32_t)BURST) * 64u <= end_plain) {
            // BURST plain steps: all the text first, then the stores back to back, so that the memory system gets
            // BURST KiB of one contiguous run at once instead of 1 KiB every few hundred cycles
            u32x4 vb[BURST];
#pragma unroll
            for (int q = 0; q < BURST; q++) {
                uint16_t h;
                __builtin_memcpy(&h, slab + slab_b0 + (int32_t)((u + (uint32_t)q) * 64u), 2);
                const uint32_t w = (uint32_t)h >> pshift;
                const uint32_t t0 = gt_text(w & 3u), t1 = gt_text((w >> 2) & 3u), t2 = gt_text((w >> 4) & 3u);
                const uint32_t t3 = gt_text((w >> 6) & 3u), t4 = gt_text((w >> 8) & 3u);
                vb[q].x = funnel_bytes(t0, t1, psh);
                vb[q].y = funnel_bytes(t1, t2, psh);
                vb[q].z = funnel_bytes(t2, t3, psh);
                vb[q].w = funnel_bytes(t3, t4, psh);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < BURST; q++) store_chunk<NT>(span_ptr + (u + (uint32_t)q) * 1024u, vb[q]);
            u += (uint32_t)BURST - 1u;
            continue;
        }
        if (u * 64u >= first_plain && u * 64u + 64u <= end_plain) {
            // every lane's chunk lies inside the row's text and inside the record: no clamps, no masks
            uint16_t h;
            __builtin_memcpy(&h, slab + slab_b0 + (int32_t)(u * 64u), 2);
            const uint32_t w = (uint32_t)h >> pshift;
            const uint32_t t0 = gt_text(w & 3u), t1 = gt_text((w >> 2) & 3u), t2 = gt_text((w >> 4) & 3u);
            const uint32_t t3 = gt_text((w >> 6) & 3u), t4 = gt_text((w >> 8) & 3u);
            u32x4 v;
            v.x = funnel_bytes(t0, t1, psh);
            v.y = funnel_bytes(t1, t2, psh);
            v.z = funnel_bytes(t2, t3, psh);
            v.w = funnel_bytes(t3, t4, psh);
            store_chunk<NT>(span_ptr + u * 1024u, v);
            continue;
        }
        const uint32_t idx = u * 64u + lane;      // position inside the 1-KiB-aligned span
        const uint32_t i = idx - it.lead;         // chunk of this item (wraps to huge when idx < lead)
        if (i >= it.cnt) continue;
        uint8_t *dst = chunk0 + (it.g0 + i) * 16ull;
        const int32_t b0 = bf + (int32_t)i;
        uint32_t window;
        {
            // bytes b0, b0+1 of the record (clamped to the staged range for don't-care positions)
            const int32_t bb = max(0, min(b0, (int32_t)last - 1));
            uint16_t h;
            __builtin_memcpy(&h, slab + bb + delta, 2);
            const int32_t d = b0 - bb;
            window = d < 0 ? ((uint32_t)h << 8) & 0xFFFFu : (uint32_t)h >> (8u * (uint32_t)min(d, 2));
        }
        // interior lanes and the row-tail lane build their 16 bytes on different paths but leave through ONE
        // store instruction (a second, one-lane store per row was 8 % of the kernel's store instructions)
        u32x4 v = gt_text16_from_window(window, (int64_t)phase);
        bool whole = i < n_interior && !(head_chunk && i == 0u);
        if (LINES) {
            // a row's first and last chunk are shared with the prefixes around it: only a last chunk whose
            // '\n' is its 16th byte is whole; the other two are written byte-wise below (edge bytes)
            if (!whole && i + 1u == it.cnt && row_tail && (int64_t)gt_bytes - (it.c_first + 16ll * (int64_t)i) == 15) {
                v.w = (v.w & 0x00FFFFFFu) | 0x0A000000u;
                whole = true;
            }
        } else if (!whole) {
            // ---- row tail: the chunk holds '\n' at byte nl (and the head of row j+1 behind it)
            const int64_t c = it.c_first + 16ll * (int64_t)i;
            const int64_t o = (int64_t)((it.g0 + i) * 16ull) - (int64_t)p.head;
            const uint32_t nl = (uint32_t)((int64_t)gt_bytes - c);  // 0..15 when c >= 0
            if (c >= 0 && (it.row + 1ull < a.n_variants || nl == 15u)) {
                u32x4 y = {0u, 0u, 0u, 0u};
                if (nl < 15u) {
                    const int64_t qy = -(int64_t)nl - 1;  // row j+1 starts nl+1 bytes into the chunk
                    // window of row j+1's first samples: record byte -1 (none) and byte 0
                    const uint32_t nb0 = NEXT_IN_SLAB ? (uint32_t)slab[kSlabBytes] : (uint32_t)row_record<HAS_VIDX>(a, it.row + 1ull)[0];
                    y = gt_text16_from_window(nb0 << 8, qy);
                }
                uint32_t xs[4] = {v.x, v.y, v.z, v.w};  // v = row j's text from this chunk's first byte (c = phase mod 16)
                uint32_t ys[4] = {y.x, y.y, y.z, y.w};
                uint32_t os[4];
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const int32_t nb = (int32_t)nl - 4 * m;  // bytes of dword m taken from row j
                    const uint32_t mask = nb >= 4 ? 0xFFFFFFFFu : (nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u));
                    uint32_t d = (xs[m] & mask) | (ys[m] & ~mask);
                    if (nb >= 0 && nb < 4) d = (d & ~(0xFFu << (8 * nb))) | (0x0Au << (8 * nb));
                    os[m] = d;
                }
                v = u32x4{os[0], os[1], os[2], os[3]};
                whole = true;
            } else {
                // first/last chunk of the whole stream: byte-wise with a validity mask
                uint64_t rr = it.row;
                int64_t cc = c;
#pragma unroll
                for (int b = 0; b < 16; b++) {
                    const int64_t ob = o + b;
                    if (ob >= 0 && (uint64_t)ob < p.total_bytes) {
                        if (cc >= (int64_t)S) {
                            cc -= (int64_t)S;
                            rr++;
                        }
                        uint32_t ch;
                        if ((uint64_t)cc == gt_bytes) {
                            ch = '\n';
                        } else {
                            const uint32_t s = (uint32_t)((uint64_t)cc >> 2);
                            const uint32_t code = ((uint32_t)row_record<HAS_VIDX>(a, rr)[s >> 2] >> ((s & 3u) * 2u)) & 3u;
                            ch = gt_text_byte(code, (uint32_t)cc & 3u);
                        }
                        dst[b] = (uint8_t)ch;
                    }
                    cc++;
                }
            }
        }
        if (whole) store_chunk<NT>(dst, v);
    }
    if (LINES) {
        // ---- edge bytes: the row's own bytes of its first chunk (lanes 0-15) and of its last chunk
        // (lanes 16-31) in ONE byte-store instruction, read straight from the staged record bytes
        const bool head_edge = head_chunk && it.cnt != 0u;
        const uint32_t i_tail = it.cnt - 1u;
        const int64_t c_tail = it.c_first + 16ll * (int64_t)i_tail;
        const uint32_t nl = (uint32_t)((int64_t)gt_bytes - c_tail);                 // byte of '\n' in the last chunk (row_tail)
        const bool tail_edge = row_tail && nl < 15u;
        if (head_edge || tail_edge) {
            const bool hl = lane < 16u;
            const uint32_t b = lane & 15u;
            const int64_t c = hl ? it.c_first : c_tail;                               // row byte of the chunk's byte 0
            const bool on = hl ? (head_edge && (int32_t)b >= -(int32_t)it.c_first) : (lane < 32u && tail_edge && b <= nl);
            if (on) {
                const uint32_t x = ((uint32_t)c & 15u) + b;                           // row byte = 16 * (c >> 4) + x
                const int32_t rb = (int32_t)(c >> 4) + (int32_t)(x >> 4);              // record byte of that sample
                const uint32_t code = ((uint32_t)slab[rb + delta] >> (((x >> 2) & 3u) * 2u)) & 3u;
                const uint32_t ch = (!hl && b == nl) ? 0x0Au : gt_text_byte(code, x & 3u);
                chunk0[(it.g0 + (hl ? 0u : i_tail)) * 16ull + b] = (uint8_t)ch;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// RUNS mode — SHORT rows (N <= ~2000: the reference's own 300-sample datasets, data/basic2 and
// data/random1).  A row of 1 201 bytes is a poor work item: 75 chunks, a quarter of a record tile,
// a '\n' merge in every store step.  With dense records (record_stride == R, no variant gather) B
// consecutive rows are ONE contiguous run of B*R record bytes and ONE contiguous run of B*S text
// bytes, so the item becomes (rows t*B .. t*B+B-1): the loader fetches the whole run — plus the
// first byte of the row behind it — with a single 16-B-per-lane load (B*R <= 1 040), and a storer
// walks the run's 16-byte-ALIGNED chunks exactly like a long row's: chunk at run offset o lies in
// row o / S (one multiply-high by a host-computed reciprocal) at row byte o % S; its window is two
// bytes of the slab.  The chunks that hold a '\n' (one per row) are not built in the store steps, where one
// such lane would drag the whole wave through the merge path every step: the text leaves through the storer's
// 4-KiB LDS stage (emit_run below: plain text of every chunk, one fix-up pass over the group's '\n' chunks,
// four stores of whole 128-B lines).
__device__ __forceinline__ Item make_run_item(const EmitArgs &a, const WideParams &p, uint64_t t)
{
    Item it;
    const uint32_t S = (uint32_t)p.row_bytes;
    const uint32_t R = p.run_rec;
    const uint64_t row0 = t * (uint64_t)p.run_rows;
    const uint32_t nrows = (uint32_t)min((uint64_t)p.run_rows, (uint64_t)a.n_variants - row0);
    const uint64_t run_start = row0 * (uint64_t)S;
    const uint32_t run_len = nrows * S;
    // the run owns the chunks whose first byte lies inside it; run 0 also the chunk that holds stream byte 0
    const uint64_t g_first = t == 0ull ? (run_start + p.head) >> 4 : (run_start + p.head + 15ull) >> 4;
    const uint64_t g_end = (run_start + run_len + p.head + 15ull) >> 4;
    it.row = row0;
    it.g0 = g_first;
    it.cnt = (uint32_t)(g_end - g_first);
    it.lead = (uint32_t)g_first & 63u;  // one span per item: store step u covers chunks (g_first - lead) + 64u .. +63
    it.c_first = (int64_t)(int32_t)((uint32_t)g_first * 16u - p.head - (uint32_t)run_start);  // run-relative offset of chunk g_first: -15 .. 15
    it.row_tail = false;
    it.rec = a.records + row0 * (uint64_t)R;  // dense records: the run's record bytes are contiguous
    // record bytes of the run + the first byte of the row behind it (head of the chunk holding the run's last '\n')
    const uint32_t n_bytes = nrows * R + (row0 + nrows < (uint64_t)a.n_variants ? 1u : 0u);
    const uint32_t mis = (uint32_t)((uint64_t)(uintptr_t)it.rec & 15ull);
    it.base = it.rec - mis;
    it.n_load = (mis + n_bytes + 15u) / 16u;  // <= 66 (host: B * R <= 1 040)
    it.delta = (int32_t)mis;
    return it;
}

// 16 text bytes of the run starting at row byte `pos` of the row whose record starts at slab offset `rec_off`
__device__ __forceinline__ u32x4 run_text16(const uint8_t *slab, uint32_t rec_off, uint32_t pos)
{
    uint16_t h;
    __builtin_memcpy(&h, slab + rec_off + (pos >> 4), 2);
    return gt_text16_from_window((uint32_t)h, (int64_t)pos);
}

// A run's text leaves in GROUPS of four store steps (256 chunks, 4 KiB) through the wave's LDS stage:
//   A. every lane builds its four chunks as plain text of the row its first byte lies in (valid up to that row's
//      '\n') and parks them in the stage (aligned ds_write_b128);
//   B. the chunks of the group that hold a '\n' — one per row, their positions follow from S alone — are fixed
//      up in ONE pass, lane i on the i-th of them: tail of row r from the stage, '\n', head of row r+1 from the
//      slab, back into the stage.  (Done inside the store steps this merge runs in every step for two or three
//      lanes — the whole wave pays for it 16 times per item instead of 4 times at N = 100; done as a separate
//      16-byte store per row it leaves a hole in the step's store, and a 128-B line written in two pieces costs
//      as much as half a KiB of whole lines: 0.31 of roofline at N = 100, measured.)
//   C. the stage goes out as four 1-KiB stores of whole 128-B lines.
constexpr uint32_t kStageBytes = 4096;

template <bool NT>
__device__ __forceinline__ void emit_run(const EmitArgs &a, const WideParams &p, const Item &it, const uint8_t *slab, uint8_t *stage, uint32_t lane)
{
    const uint32_t S = (uint32_t)p.row_bytes;
    const uint32_t R = p.run_rec;
    uint8_t *const chunk0 = a.out - p.head;
    const int32_t c_first = (int32_t)it.c_first;
    const uint32_t delta = (uint32_t)it.delta;
    const uint32_t nrows = (uint32_t)min((uint64_t)p.run_rows, (uint64_t)a.n_variants - it.row);
    const uint32_t run_len = nrows * S;
    const bool has_next_run = it.row + (uint64_t)nrows < (uint64_t)a.n_variants;
    uint8_t *const span_ptr = chunk0 + (it.g0 - it.lead) * 16ull + lane * 16u;  // lane's chunk in step 0
    const uint32_t end = it.lead + it.cnt;
    // the stream's very last chunk ends at the last '\n': if that is not its 16th byte the chunk is written byte-wise (phase B)
    const uint32_t e_last = run_len - 1u;
    const bool ragged_end = !has_next_run && (((uint32_t)((int32_t)e_last - c_first)) & 15u) != 15u;
    const uint32_t cnt_whole = it.cnt - (ragged_end ? 1u : 0u);
    v4u *const st = reinterpret_cast<v4u *>(stage);
    for (uint32_t g = 0; g * 256u < end; g++) {
        // ---- A: plain text of every chunk of the group
#pragma unroll
        for (uint32_t s4 = 0; s4 < 4u; s4++) {
            const uint32_t idx = g * 256u + s4 * 64u + lane;
            if (g * 256u + s4 * 64u >= end) break;
            const int32_t o_s = c_first + 16 * (int32_t)(idx - it.lead);
            const uint32_t o = (uint32_t)min(max(o_s, 0), (int32_t)run_len - 1);  // lanes outside the item build a harmless chunk
            uint32_t r = __umulhi(o, p.magic);
            uint32_t rs = __umul24(r, S);
            if (rs > o) {
                r--;
                rs -= S;
            }
            const u32x4 x = run_text16(slab, delta + __umul24(r, R), o - rs);
            st[s4 * 64u + lane] = v4u{x.x, x.y, x.z, x.w};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- B: the group's '\n' chunks.  Chunks [i_lo, i_hi) of the item are in the group; row rr's '\n' sits at run
        // offset e = rr * S + S - 1, and e >= lo  <=>  rr >= floor(lo / S)
        const uint32_t i_lo = g * 256u > it.lead ? g * 256u - it.lead : 0u;
        const uint32_t i_hi = min(it.cnt, g * 256u + 256u - it.lead);
        const uint32_t lo = (uint32_t)max(c_first + 16 * (int32_t)i_lo, 0);
        const uint32_t hi = (uint32_t)(c_first + 16 * (int32_t)i_hi);   // > 0: the group holds at least one chunk of the item
        // (wave-uniform: s_mul_hi_u32 by the host's reciprocal instead of a division sequence per group)
        uint32_t rr_lo = __umulhi(lo, p.magic);
        if (rr_lo * S > lo) rr_lo--;
        uint32_t rr_hi = __umulhi(hi, p.magic);
        if (rr_hi * S > hi) rr_hi--;
        rr_hi = min(rr_hi, nrows);
        for (uint32_t rr = rr_lo + lane; rr < rr_hi; rr += 64u) {
            const uint32_t e = rr * S + S - 1u;                            // run offset of the row's '\n'
            const uint32_t i = (uint32_t)((int32_t)e - c_first) >> 4;      // its chunk (e >= 32 > c_first)
            const uint32_t nl = (uint32_t)((int32_t)e - c_first) & 15u;    // its byte inside that chunk
            const uint32_t slot = i + it.lead - g * 256u;                  // 0 .. 255
            const v4u x = st[slot];
            const bool has_next = rr + 1u < nrows || has_next_run;
            u32x4 y = {0u, 0u, 0u, 0u};
            if (nl < 15u && has_next) y = gt_text16_from_window((uint32_t)slab[delta + (rr + 1u) * R] << 8, -(int64_t)nl - 1);  // record byte -1 (none) and byte 0 of row rr+1
            const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
            const uint32_t ys[4] = {y.x, y.y, y.z, y.w};
            uint32_t os[4];
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int32_t nb = (int32_t)nl - 4 * m;  // bytes of dword m taken from row rr
                const uint32_t mask = nb >= 4 ? 0xFFFFFFFFu : (nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u));
                uint32_t d = (xs[m] & mask) | (ys[m] & ~mask);
                if (nb >= 0 && nb < 4) d = (d & ~(0xFFu << (8 * nb))) | (0x0Au << (8 * nb));
                os[m] = d;
            }
            if (has_next || nl == 15u) {
                st[slot] = v4u{os[0], os[1], os[2], os[3]};
            } else {
                // last row of the whole stream: the chunk ends at the '\n', the bytes behind it are not ours
                uint8_t *const dst = chunk0 + (it.g0 + i) * 16ull;
#pragma unroll
                for (int b = 0; b < 16; b++)
                    if ((uint32_t)b <= nl) dst[b] = (uint8_t)(os[b >> 2] >> (8 * (b & 3)));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- C: four 1-KiB stores
#pragma unroll
        for (uint32_t s4 = 0; s4 < 4u; s4++) {
            const uint32_t u = g * 4u + s4;
            if (u * 64u >= end) break;
            const uint32_t i = u * 64u + lane - it.lead;  // wraps to huge before `lead`
            const v4u v = st[s4 * 64u + lane];
            if (i < cnt_whole && !(c_first < 0 && i == 0u)) store_chunk<NT>(span_ptr + u * 1024u, u32x4{v.x, v.y, v.z, v.w});
        }
        // the stage is rewritten by the next group: this group's reads must have returned first
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // ---- run 0 with an unaligned output pointer: the bytes of chunk 0 that belong to the stream (all in row 0's text: S >= 33)
    if (c_first < 0 && it.cnt != 0u && lane < 16u && (int32_t)lane + c_first >= 0) {
        const uint32_t ob = (uint32_t)((int32_t)lane + c_first);
        const uint32_t code = ((uint32_t)slab[delta + (ob >> 4)] >> (((ob >> 2) & 3u) * 2u)) & 3u;
        chunk0[it.g0 * 16ull + lane] = (uint8_t)gt_text_byte(code, ob & 3u);
    }
}

// ---------------------------------------------------------------------------------------------
// gt_stream_dyn_kernel — ROLES: wave 0 of a block only loads (global -> registers -> LDS slabs),
// waves 1..NS only store (LDS -> text -> global).  gfx9 counts loads and stores in one in-order
// vmcnt, so a wave that consumes a load must also wait for all of its older stores; a storer wave
// here never waits on vmcnt, and its 1-KiB stores stream back to back like a fill kernel's, while
// the loader keeps NS wide loads in flight one ring slot ahead.
// Hand-off through LDS: per storer a ring of kRingSlots slabs, `full` / `done` sequence words
// (LDS operations of one wave execute in order, and both waves live on one CU, so a flag written
// after the slab's ds_writes is seen after them; compiler ordering is pinned with asm barriers).
// (Round 1 also carried a symmetric-wave kernel and a static-partition role kernel as A/B partners:
// 0.54 and 0.61 of roofline against this kernel's 0.66-0.73 on the chr22 block, profiles/r01_*; removed.)
constexpr int kRingSlots = 3;
// descriptor ring: ring slots + 1 (slot s % 4 was last read in step s-4, whose `done` the loader saw in step s-1)

// WORK QUEUE instead of a static item partition: only 4 of these 512-thread blocks fit on a CU, so
// a static grid-stride split of a big grid runs in rounds and the last, partly filled round is a
// tail of several hundred microseconds on a 2.4 ms launch (measured with per-wave s_memrealtime
// stamps).  Here the items are cut into
// contiguous ranges (two by default — one per XCD measured 1 % slower; picked by blockIdx & (n - 1)); the
// loader wave claims NS consecutive items per step with one returning atomicAdd on its range's
// head word (heads live 128 B apart; ~10 claims/us per word, far below the ~88/us a word takes)
// and steals from the next range when its own is drained.  Every block therefore runs until the
// whole launch is out of work and all of them finish within one step of each other.
template <int NS, bool HAS_VIDX, bool NT, bool LINES = false, int BURST = 2, bool RUNS = false>
__global__ __launch_bounds__(64 * (NS + 1)) void gt_stream_dyn_kernel(EmitArgs a, WideParams p)
{
    // RUNS mode trades one slab slot per storer for the storers' 4-KiB text stages (LDS per block: 25 KB -> 45 KB)
    constexpr int RS = RUNS ? 2 : kRingSlots;
    constexpr int DS = RS + 1;
    __shared__ __attribute__((aligned(16))) uint8_t slabs[NS][RS][kSlabBytes + kSlabExtra];
    __shared__ __attribute__((aligned(16))) uint8_t s_desc[NS][DS][kDescBytes];
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[RUNS ? NS : 1][RUNS ? kStageBytes : 16u];
    __shared__ uint32_t s_full[NS][RS];
    __shared__ uint32_t s_done[NS][RS];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < NS * RS) {
        (&s_full[0][0])[threadIdx.x] = 0u;
        (&s_done[0][0])[threadIdx.x] = 0u;
    }
    __syncthreads();
    constexpr uint64_t kNoItem = ~0ull;

    if (wave == 0u) {
        // ------------------------------ loader wave ------------------------------
        const uint32_t nr = p.n_ranges;  // power of two <= kMaxQueueRanges
        const uint64_t per_range = (p.n_items + (uint64_t)nr - 1ull) / (uint64_t)nr;
        uint32_t range = blockIdx.x & (nr - 1u);
        uint32_t drained = 0u;  // consecutive ranges found empty
        // The claim for step s+1 is ISSUED right after step s's record loads and only read at the top of step
        // s+1, so the atomic's round trip (1-2 us under load) hides behind the load latency instead of adding
        // to every step of this wave — the storers wait for nothing else.
        auto issue_claim = [&](uint32_t rng) -> uint64_t {
            uint64_t got = 0;
            if (lane == 0u) got = atomicAdd(reinterpret_cast<unsigned long long *>(a.work_counters + rng * 16u), (unsigned long long)NS);
            return got;
        };
        uint64_t pending = issue_claim(range);
        for (uint32_t step = 0;; step++) {
            // ---- resolve the claim issued one step ago (lane 0 asked, the wave shares the answer)
            uint64_t t0 = kNoItem;
            uint64_t got = sgpr64(pending);
            while (drained < nr) {
                const uint64_t lo = (uint64_t)range * per_range;
                const uint64_t hi = min(lo + per_range, p.n_items);
                if (lo + got < hi) {
                    t0 = lo + got;
                    break;
                }
                range = (range + 1u) & (nr - 1u);  // own range drained: steal from the next one
                drained++;
                if (drained < nr) got = sgpr64(issue_claim(range));
            }
            const uint64_t t_end = t0 == kNoItem ? 0ull : min(t0 + NS, min(((uint64_t)range + 1ull) * per_range, p.n_items));
            const uint32_t slot = (uint32_t)(step % RS);
            // in0[w]: pieces 0-63 of item w; `ext`: pieces 64 and 65 of ALL items, item w in lanes 2w and 2w+1, fetched
            // by ONE load after the item loop (seven masked loads into one register quad would each wait for the
            // one before: the compiler orders writes to a register it cannot prove lane-disjoint)
            v4u in0[NS];
            v4u ext = v4u{0u, 0u, 0u, 0u};
            const uint8_t *ext_addr = a.records;
            bool ext_on = false;
            uint32_t nb[NS];
            // one division per step: the step's items are consecutive, so (row, span) just counts on
            uint64_t j_it = 0ull;
            uint32_t k_it = 0u;
            if (!RUNS && t0 != kNoItem) {
                j_it = p.spans_per_row == 1u ? t0 : t0 / p.spans_per_row;
                k_it = p.spans_per_row == 1u ? 0u : (uint32_t)(t0 - j_it * p.spans_per_row);
            }
            const uint32_t n_here = t0 == kNoItem ? 0u : (uint32_t)(t_end - t0);  // items of this step (<= NS)
            // LINES: where each row's GT segment starts.  A step's items cover at most NS consecutive rows:
            // lane r fetches line_off / prefix_off of row j_it + r (one coalesced load each) and keeps that row's
            // start; items pick theirs with v_readlane.  The value is passed through an asm move so that the
            // compiler does not tie the later readlanes to the loads (it would wait for the previous item's
            // record load before every item).
            const uint64_t j_step = j_it;
            uint32_t rs_lo = 0u, rs_hi = 0u;
            if (LINES && t0 != kNoItem) {
                const uint64_t jr = min(j_it + (uint64_t)lane, (uint64_t)a.n_variants);  // both arrays have n_variants + 1 entries
                const uint64_t lo_v = a.line_off[jr];
                const uint64_t po_v = a.prefix_off[jr];
                const uint64_t po_next = __shfl_down((unsigned long long)po_v, 1, 64);
                const uint64_t rs = lo_v + (po_next - po_v);
                asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(rs_lo), "=v"(rs_hi) : "v"((uint32_t)rs), "v"((uint32_t)(rs >> 32)));
            }
            // GATHERED rows (variant list or byte offsets): where the records of rows j_step .. j_step + 63 lie, relative to a.records —
            // one coalesced load per step (lane r: row j_step + r) instead of a dependent index load in front of every item's record
            // load and another for the row behind it (seven serial round trips per step: 0.42 of roofline on the chr22 block against
            // 0.72 for dense rows).  Passed through an asm move like the LINES offsets above.
            uint32_t go_lo = 0u, go_hi = 0u;
            if (HAS_VIDX && !RUNS && t0 != kNoItem) {
                const uint64_t jr = min(j_it + (uint64_t)lane, (uint64_t)a.n_variants - 1ull);
                const uint64_t off = a.record_off != nullptr ? a.record_off[jr] : (uint64_t)a.variant_idx[jr] * a.record_stride;
                asm volatile("v_mov_b32 %1, %3\n\tv_mov_b32 %0, %2" : "=v"(go_lo), "=v"(go_hi) : "v"((uint32_t)off), "v"((uint32_t)(off >> 32)));
            }
            auto gathered_at = [&](uint32_t r) -> const uint8_t * {
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)go_hi, (int)r);
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)go_lo, (int)r);
                return a.records + (((uint64_t)hi << 32) | (uint64_t)lo);
            };
#pragma unroll
            for (int w = 0; w < NS; w++) {
                nb[w] = 0u;
                in0[w] = v4u{0u, 0u, 0u, 0u};
                // descriptor first (its ring has one slot more than the slab ring, so no wait is needed here)
                if ((uint32_t)w >= n_here) {
                    // ~0-1 = "launch is out of work", ~0 = no item for this storer in this (last) step of a range
                    if (lane == 0u) desc_put_item(s_desc[w][step % DS], Item{}, t0 == kNoItem ? kNoItem - 1ull : kNoItem);
                } else {
                    uint64_t row_start = 0ull;
                    if (LINES) {
                        const uint32_t r = (uint32_t)(j_it - j_step);  // < NS
                        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)rs_hi, (int)r);
                        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)rs_lo, (int)r);
                        row_start = ((uint64_t)hi << 32) | (uint64_t)lo;
                    }
                    const uint32_t r_it = (uint32_t)(j_it - j_step);   // row of the step: < NS
                    const uint8_t *const rec_next = HAS_VIDX ? gathered_at(r_it + 1u) : nullptr;   // (wave-uniform, outside the lane-0 branch below)
                    const Item it = RUNS ? make_run_item(a, p, t0 + (uint64_t)w)
                                         : make_item_at<HAS_VIDX, LINES>(a, p, j_it, k_it, row_start, HAS_VIDX ? gathered_at(r_it) : nullptr);
                    if (!RUNS && ++k_it == p.spans_per_row) {
                        k_it = 0u;
                        j_it++;
                    }
                    if (lane == 0u) desc_put_item(s_desc[w][step % DS], it, t0 + (uint64_t)w);
                    if (lane < it.n_load) in0[w] = *reinterpret_cast<const v4u *>(it.base + lane * 16u);
                    if ((lane >> 1) == (uint32_t)w) {
                        ext_addr = it.base + (64u + (lane & 1u)) * 16u;
                        ext_on = 64u + (lane & 1u) < it.n_load;
                    }
                    // rows < n_variants <= 2^32 - 1, so row + 1 fits 32 bits
                    if (!LINES && !RUNS && it.row_tail && (uint32_t)it.row + 1u < a.n_variants && lane == 0u)
                        nb[w] = (uint32_t)(HAS_VIDX ? rec_next : row_record<HAS_VIDX>(a, it.row + 1ull))[0];
                }
            }
            if (ext_on) ext = *reinterpret_cast<const v4u *>(ext_addr);
            if (t0 != kNoItem) pending = issue_claim(range);  // for the next step; read at the top of the loop
#pragma unroll
            for (int w = 0; w < NS; w++) {
                if (step >= (uint32_t)RS) {
                    const uint32_t want = step - (uint32_t)RS + 1u;
                    while (lds_flag_read(lds_offset(&s_done[w][slot])) != want) __builtin_amdgcn_s_sleep(1);
                }
                uint8_t *slab = slabs[w][slot];
                *reinterpret_cast<v4u *>(slab + lane * 16u) = in0[w];
                if ((lane >> 1) == (uint32_t)w) *reinterpret_cast<v4u *>(slab + (64u + (lane & 1u)) * 16u) = ext;
                if (lane == 0u) slab[kSlabBytes] = (uint8_t)nb[w];
                if (lane == 0u) lds_flag_write(lds_offset(&s_full[w][slot]), step + 1u);
            }
            if (t0 == kNoItem) break;
        }
        // ---- self-cleaning queue: the last loader wave to leave re-zeroes the heads for the next launch (every block's
        // claims precede its exit count; no memset node in front of the kernel, and a captured graph can be replayed)
        {
            unsigned long long *const exits = reinterpret_cast<unsigned long long *>(a.work_counters + kMaxQueueRanges * 16u);
            uint32_t last = 0u;
            if (lane == 0u) last = atomicAdd(exits, 1ull) == (unsigned long long)gridDim.x - 1ull ? 1u : 0u;
            if (__builtin_amdgcn_readfirstlane(last)) {
                if (lane < kMaxQueueRanges) atomicExch(reinterpret_cast<unsigned long long *>(a.work_counters + lane * 16u), 0ull);   // lane h: head h
                if (lane == 0u) atomicExch(exits, 0ull);
            }
        }
    } else {
        // ------------------------------ storer waves -----------------------------
        const uint32_t w = wave - 1u;
        // LINES: this wave's share of the lines' prefix bytes first (bytes no item writes; the loader's first loads are in flight meanwhile)
        if (LINES && p.pfx_shift != 0u) copy_prefix_rows(a, p.pfx_shift, (uint64_t)blockIdx.x * NS + w, (uint64_t)gridDim.x * NS, lane);
        for (uint32_t step = 0;; step++) {  // 32-bit: a block that ran 2^32 steps would have written > 2^47 bytes
            const uint32_t slot = step % RS;
            while (lds_flag_read(lds_offset(&s_full[w][slot])) != step + 1u) __builtin_amdgcn_s_sleep(1);
            const uint8_t *slab = slabs[w][slot];
            const uint8_t *desc = s_desc[w][step % DS];
            uint64_t t = *reinterpret_cast<const uint64_t *>(desc + 8u);
            t = sgpr64(t);
            if (t == kNoItem - 1ull) break;          // the loader found every range drained
            if (t != kNoItem) {                       // kNoItem: this storer has no item in this (last) step of a range
                const Item it = desc_get_item(desc);
                if (RUNS)
                    emit_run<NT>(a, p, it, slab, s_stage[RUNS ? w : 0u], lane);
                else
                    emit_item<HAS_VIDX, NT, true, LINES, BURST>(a, p, it, slab, lane);
            }
            if (lane == 0u) lds_flag_write(lds_offset(&s_done[w][slot]), step + 1u);
        }
    }
}


// ---------------------------------------------------------------------------------------------
// gt_lineruns_kernel — FULL LINES (pgenhip_emit_lines, src/pfile.rs:156-192) on SHORT rows, all samples kept, dense records:
// the RUNS idea for whole VCF body lines.  Lines are packed back to back, so a run of B lines is ONE contiguous piece of the
// output: prefix_0 GT_0 '\n' prefix_1 GT_1 '\n' ...  A work item is such a run; the loader hands a storer the run's records
// (one wide load of the contiguous record bytes), its prefix bytes (one wide load of the contiguous piece of the prefix blob) and
// the lines' start and prefix offsets relative to the run.  The storer emits the run in 4-KiB groups through its LDS stage:
//   A. lane <-> 16-byte-aligned chunk: chunks that lie wholly inside one line's GT text are built as in the RUNS mode (the line is
//      found by comparing with the few line starts of the group);
//   B. everything else — the last bytes of a line's GT text, its '\n', the next line's prefix, the first bytes of that line's GT
//      text up to the next chunk boundary — is written into the stage byte by byte, one wave pass per line of the group (all
//      bounds wave-uniform);
//   C. four 1-KiB stores of whole 128-B lines.
// Against flushing row by row behind a separate prefix copy (gt_pick.hip LINES: 0.24 of roofline at N = 300) every 128-B line
// leaves whole, once.
constexpr uint32_t kLrMaxRows = 30;                       // lines per item (+ the line behind the run)
constexpr uint32_t kLrRecBytes = kSlabBytes;              // slab: [records | prefix bytes | lrel[32] | prel[32]]
constexpr uint32_t kLrPfxBytes = 768;
constexpr uint32_t kLrSlab = kLrRecBytes + kLrPfxBytes + 2u * 32u * 4u;

// one text byte of GT-segment offset x of the record at slab offset rec_off (src/pfile.rs:171-187)
__device__ __forceinline__ uint32_t run_text_byte(const uint8_t *slab, uint32_t rec_off, uint32_t x)
{
    const uint32_t code = ((uint32_t)slab[rec_off + (x >> 4)] >> (((x >> 2) & 3u) * 2u)) & 3u;
    return gt_text_byte(code, x & 3u);
}

// kept-subset form (PICK): the 2-bit code of kept rank `rank` of the record at `rec` — `tab` is the block's LDS copy of the ascending
// kept list as u16 (what filter_metadata src/pfile.rs:319-333 yields), with kTabFront / kTabBack entries of slack around it (ranks
// -4 .. K + 4 occur next to a line's ends; their bytes are never stored)
constexpr uint32_t kTabFront = 4, kTabBack = 12, kTabMaxSamples = 4096;
__device__ __forceinline__ uint32_t tab_code(const uint8_t *rec, const uint16_t *tab, int32_t rank)
{
    const uint32_t s = (uint32_t)tab[rank];
    return ((uint32_t)rec[s >> 2] >> ((s & 3u) * 2u)) & 3u;
}

// 16 text bytes of the line's GT segment from byte x (x >= -15) for a kept list
__device__ __forceinline__ u32x4 tab_text16(const uint8_t *rec, const uint16_t *tab, int32_t x)
{
    const int32_t g = x >> 2;  // floor
    const uint32_t sh = (uint32_t)x & 3u;
    const uint32_t t0 = gt_text(tab_code(rec, tab, g)), t1 = gt_text(tab_code(rec, tab, g + 1)), t2 = gt_text(tab_code(rec, tab, g + 2));
    const uint32_t t3 = gt_text(tab_code(rec, tab, g + 3)), t4 = gt_text(tab_code(rec, tab, g + 4));
    return u32x4{funnel_bytes(t0, t1, sh), funnel_bytes(t1, t2, sh), funnel_bytes(t2, t3, sh), funnel_bytes(t3, t4, sh)};
}

template <bool PICK>
__device__ __forceinline__ void emit_lines_run(const EmitArgs &a, const WideParams &p, const Item &it, const uint8_t *slab, uint8_t *stage,
                                               uint32_t pdelta, uint32_t lane, const uint16_t *tab)
{
    const int32_t N4 = (int32_t)(4u * a.kept_count);                // GT text bytes of a line, without its '\n'
    const uint32_t R = a.record_size;
    const uint32_t delta = (uint32_t)it.delta;                      // slab offset of the run's first record
    const uint8_t *const pfx = slab + kLrRecBytes + pdelta;         // line r's prefix bytes at pfx + prel[r]
    const uint32_t *const lrel = reinterpret_cast<const uint32_t *>(slab + kLrRecBytes + kLrPfxBytes);  // line starts, run-relative
    const uint32_t *const prel = lrel + 32;                         // prefix starts, relative to the run's first prefix
    const uint32_t nrows = (uint32_t)min((uint64_t)p.run_rows, (uint64_t)a.n_variants - it.row);
    const bool has_next = it.row + (uint64_t)nrows < (uint64_t)a.n_variants;
    const int32_t run_len = (int32_t)lrel[nrows];
    uint8_t *const chunk0 = a.out - p.head;
    const int32_t c_first = (int32_t)it.c_first;
    uint8_t *const span_ptr = chunk0 + (it.g0 - it.lead) * 16ull + lane * 16u;   // lane's chunk in step 0
    const uint32_t end = it.lead + it.cnt;
    // the chunk grid is anchored at c_first (mod 16): start of the chunk that holds run offset x / first chunk start >= x
    auto adown = [&](int32_t x) { return x - ((x - c_first) & 15); };
    auto aup = [&](int32_t x) { return x + ((c_first - x) & 15); };
    // only whole chunks inside [0, run_len (+ the line behind the run)) leave as 16-byte stores; the stream's first and last chunk, if
    // ragged, are written byte-wise from the stage
    const int32_t own_end = c_first + 16 * (int32_t)it.cnt;         // end of the item's last chunk
    const bool ragged_head = c_first < 0;
    const bool ragged_tail = !has_next && own_end > run_len;
    v4u *const st = reinterpret_cast<v4u *>(stage);
    // lane l holds line l's start and prefix start (read once per item; wave-uniform values come out with v_readlane, no LDS wait)
    const int32_t lr_v = lane <= nrows ? (int32_t)lrel[lane] : 0x7FFFFFFF;
    const uint32_t pr_v = prel[min(lane, 31u)];
    auto line_start = [&](uint32_t l) { return (int32_t)__builtin_amdgcn_readlane(lr_v, (int)l); };                     // l <= nrows
    auto line_plen = [&](uint32_t l) { return (int32_t)((uint32_t)__builtin_amdgcn_readlane((int)pr_v, (int)(l + 1u)) - (uint32_t)__builtin_amdgcn_readlane((int)pr_v, (int)l)); };
    const float inv_mean = (float)nrows / (float)max(run_len, 1);   // lines per byte of the run (phase A's first guess of a chunk's line)
    // phase B: lanes per seam (log2) by the launch's longest prefix (+ the '\n' in front of it), four bytes per lane
    const uint32_t seam_bytes = (uint32_t)(a.max_line_bytes - (uint64_t)(4u * a.kept_count + 1u)) + 1u;
    const uint32_t seam_shift = seam_bytes <= 16u ? 2u : (seam_bytes <= 32u ? 3u : (seam_bytes <= 64u ? 4u : (seam_bytes <= 128u ? 5u : 6u)));
    // (only a prefix shorter than 15 bytes can leave a line's first GT bytes in a chunk that starts back in the previous line)
    const uint32_t plen_v = (uint32_t)__shfl_down((int)pr_v, 1, 64) - pr_v;   // lane l: prefix length of line l (every lane takes part in the shuffle)
    const bool short_prefix = __ballot(lane < nrows + (has_next ? 1u : 0u) && plen_v < 15u) != 0ull;
    for (uint32_t g = 0; g * 256u < end; g++) {
        const int32_t gb = c_first + 16 * ((int32_t)(g * 256u) - (int32_t)it.lead);        // run offset of stage byte 0
        const int32_t glo = max(gb, c_first), ghi = min(gb + 4096, own_end);               // the item's bytes in this group
        // lines that start inside (glo, ghi): r_lo = last line starting at or before glo
        uint32_t r_lo = 0u, r_hi = 0u;
        r_lo = (uint32_t)__popcll(__ballot(lane >= 1u && lr_v <= max(glo, 0)));            // (lanes > nrows hold INT_MAX)
        r_hi = (uint32_t)__popcll(__ballot(lane >= 1u && lr_v < ghi));
        // ---- A: every chunk that overlaps the GT text of the line its FIRST byte lies in, as 16 bytes of that line's text at the
        // chunk's phase (bytes of the chunk outside the GT text are filler: phase B overwrites them).
        // Which line: the lines of a run differ only by their prefixes, so offset / (mean line length) is off by a line at most; two LDS
        // reads of the run's line starts and a (rarely taken) walk settle it.  (Round 2 compared every chunk with up to 12 line starts
        // held in scalars: 33 VALU per 64 chunks and 24 v_readlane per group — a quarter of this issue-bound kernel's instructions.)
#pragma unroll
        for (uint32_t s4 = 0; s4 < 4u; s4++) {
            if (g * 256u + s4 * 64u >= end) break;
            const int32_t o = gb + 16 * (int32_t)(s4 * 64u + lane);
            uint32_t rr = min((uint32_t)((float)max(o, 0) * inv_mean), nrows);
            int32_t ls = (int32_t)lrel[rr];
            while (o < ls && rr > 0u) ls = (int32_t)lrel[--rr];
            while (rr < nrows && o >= (int32_t)lrel[rr + 1u]) ls = (int32_t)lrel[++rr];
            const int32_t gt_lo = ls + (int32_t)(prel[rr + 1u] - prel[rr]);   // (rr = nrows: not used below; prel has 32 entries, rr + 1 <= 31)
            const int32_t x = o - gt_lo;
            if (x > -16 && x < N4 && rr < nrows) {
                const uint8_t *const rec = slab + delta + rr * R;
                u32x4 v;
                if (PICK) {
                    v = tab_text16(rec, tab, x);
                } else {
                    uint32_t window;
                    if (x >= 0) {
                        uint16_t h;
                        __builtin_memcpy(&h, rec + (x >> 4), 2);
                        window = h;
                    } else {
                        window = (uint32_t)rec[0] << 8;  // record byte -1 (none) and byte 0
                    }
                    v = gt_text16_from_window(window, (int64_t)x);
                }
                st[s4 * 64u + lane] = v4u{v.x, v.y, v.z, v.w};
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- B: what lies between two lines' GT texts, on top of A's chunks: line l-1's '\n' and line l's prefix.  A lane takes FOUR
        // bytes of a seam and a seam takes 4 .. 64 lanes by the launch's longest prefix, so a wave pass covers up to 16 seams (round 2: one
        // pass per seam, 45 instructions for ~31 bytes, 3.3 seams per 4-KiB group at N = 300 and 9.5 at N = 100)
        const uint32_t l_hi = min(r_hi + 1u, nrows);
        for (uint32_t l0 = r_lo; l0 <= l_hi; l0 += 64u >> seam_shift) {
            const uint32_t l = l0 + (lane >> seam_shift);                       // this lane's line (64 lanes per seam: wave-uniform)
            if (l <= l_hi) {
                const uint32_t lc = min(l, nrows);
                const bool line_l = l < nrows || has_next;                        // line l's prefix is in the slab
                const int32_t ls = (int32_t)lrel[lc];                             // start of line l (= run_len for l == nrows)
                const uint32_t pr = prel[lc];
                const int32_t plen = line_l ? (int32_t)(prel[lc + 1u] - pr) : 0;
                const int32_t lim_hi = min(ghi, l == nrows && !has_next ? run_len : own_end);
                const int32_t lo_ok = l >= 1u ? 0 : 1;                           // (run line 0: the '\n' in front of it belongs to the run before)
                for (int32_t b4 = 4 * (int32_t)(lane & ((1u << seam_shift) - 1u)); b4 <= plen; b4 += 4 << seam_shift) {
#pragma unroll
                    for (int32_t j = 0; j < 4; j++) {                             // b = 0: the '\n' of line l-1, b = 1 .. plen: the prefix
                        const int32_t b = b4 + j, ob = ls - 1 + b;
                        if (b <= plen && b >= lo_ok && ob >= glo && ob < lim_hi) stage[ob - gb] = b == 0 ? (uint8_t)'\n' : pfx[pr + (uint32_t)(b - 1)];
                    }
                }
            }
        }
        // and — only when a prefix is so short that the chunk holding line l's first GT byte started back in line l-1 — line l's first GT
        // bytes up to the chunk boundary (lanes 48 .. 62), one pass per such line
        if (short_prefix) {
            for (uint32_t l = r_lo; l <= l_hi; l++) {
                const bool line_l = l < nrows || has_next;
                const int32_t ls = line_start(min(l, nrows));
                const int32_t plen = line_l ? line_plen(l) : 0;
                const int32_t gt_lo = ls + plen;
                const int32_t lim_hi = min(ghi, l == nrows && !has_next ? run_len : own_end);
                if (line_l && adown(gt_lo) < ls && lane >= 48u) {
                    const int32_t ob = gt_lo + (int32_t)(lane - 48u);
                    if (ob < aup(gt_lo) && ob >= glo && ob < lim_hi)
                        stage[ob - gb] = (uint8_t)(PICK ? gt_text_byte(tab_code(slab + delta + l * R, tab, (int32_t)((lane - 48u) >> 2)), (lane - 48u) & 3u)
                                                        : run_text_byte(slab, delta + l * R, lane - 48u));
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- C: four 1-KiB stores
#pragma unroll
        for (uint32_t s4 = 0; s4 < 4u; s4++) {
            const uint32_t u = g * 4u + s4;
            if (u * 64u >= end) break;
            const uint32_t i = u * 64u + lane - it.lead;  // wraps to huge before `lead`
            const v4u v = st[s4 * 64u + lane];
            const bool whole = i < it.cnt && !(ragged_head && i == 0u) && !(ragged_tail && i + 1u == it.cnt);
            if (whole) store_chunk<true>(span_ptr + u * 1024u, u32x4{v.x, v.y, v.z, v.w});
            else if (i < it.cnt) {
                // the stream's first / last chunk: only the bytes inside [0, run_len)
                const int32_t o = c_first + 16 * (int32_t)i;
                uint8_t *const dst = chunk0 + (it.g0 + i) * 16ull;
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int b = 0; b < 16; b++)
                    if (o + b >= 0 && o + b < run_len) dst[b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
            }
        }
        // the stage is rewritten by the next group: this group's reads must have returned first
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int NS, bool PICK = false>
__global__ __launch_bounds__(64 * (NS + 1)) void gt_lineruns_kernel(EmitArgs a, WideParams p)
{
    constexpr int RS = 2, DS = RS + 1;
    __shared__ uint16_t s_tab[PICK ? kTabFront + kTabMaxSamples + kTabBack : 1];
    if (PICK) {
        for (uint32_t r = threadIdx.x; r < kTabFront + a.kept_count + kTabBack; r += 64u * (NS + 1))
            s_tab[r] = r >= kTabFront && r < kTabFront + a.kept_count ? (uint16_t)a.kept_idx[r - kTabFront] : (uint16_t)0;
    }
    __shared__ __attribute__((aligned(16))) uint8_t slabs[NS][RS][kLrSlab];
    __shared__ __attribute__((aligned(16))) uint8_t s_desc[NS][DS][kDescBytes];
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[NS][kStageBytes];
    __shared__ uint32_t s_full[NS][RS];
    __shared__ uint32_t s_done[NS][RS];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < NS * RS) {
        (&s_full[0][0])[threadIdx.x] = 0u;
        (&s_done[0][0])[threadIdx.x] = 0u;
    }
    __syncthreads();
    constexpr uint64_t kNoItem = ~0ull;

    if (wave == 0u) {
        // ------------------------------ loader wave ------------------------------
        const uint32_t nr = p.n_ranges;
        const uint64_t per_range = (p.n_items + (uint64_t)nr - 1ull) / (uint64_t)nr;
        uint32_t range = blockIdx.x & (nr - 1u);
        uint32_t drained = 0u;
        auto issue_claim = [&](uint32_t rng) -> uint64_t {
            uint64_t got = 0;
            if (lane == 0u) got = atomicAdd(reinterpret_cast<unsigned long long *>(a.work_counters + rng * 16u), (unsigned long long)NS);
            return got;
        };
        const uint32_t R = a.record_size, B = p.run_rows;
        uint64_t pending = issue_claim(range);
        for (uint32_t step = 0;; step++) {
            uint64_t t0 = kNoItem;
            uint64_t got = sgpr64(pending);
            while (drained < nr) {
                const uint64_t lo = (uint64_t)range * per_range;
                const uint64_t hi = min(lo + per_range, p.n_items);
                if (lo + got < hi) {
                    t0 = lo + got;
                    break;
                }
                range = (range + 1u) & (nr - 1u);
                drained++;
                if (drained < nr) got = sgpr64(issue_claim(range));
            }
            const uint64_t t_end = t0 == kNoItem ? 0ull : min(t0 + NS, min(((uint64_t)range + 1ull) * per_range, p.n_items));
            const uint32_t n_here = t0 == kNoItem ? 0u : (uint32_t)(t_end - t0);
            const uint32_t slot = step % RS;
            // ---- round trip 1: every item's line / prefix offsets (lane l: row row0 + l, rows row0 .. row0 + nrows + 1) and records
            uint64_t lo_v[NS], po_v[NS];
            v4u in0[NS];
            v4u ext = v4u{0u, 0u, 0u, 0u};
            const uint8_t *ext_addr = a.records;
            bool ext_on = false;
#pragma unroll
            for (int w = 0; w < NS; w++) {
                lo_v[w] = 0ull;
                po_v[w] = 0ull;
                in0[w] = v4u{0u, 0u, 0u, 0u};
                if ((uint32_t)w < n_here) {
                    const uint64_t row0 = (t0 + (uint64_t)w) * (uint64_t)B;
                    const uint64_t jr = min(row0 + (uint64_t)lane, (uint64_t)a.n_variants);  // both arrays have n_variants + 1 entries
                    lo_v[w] = a.line_off[jr];
                    po_v[w] = a.prefix_off[jr];
                    const uint32_t nrows = (uint32_t)min((uint64_t)B, (uint64_t)a.n_variants - row0);
                    const uint8_t *const rec = a.records + row0 * (uint64_t)R;
                    // (+ the first byte of the record behind the run: the first GT bytes of that line may share the run's last chunk; with a
                    // kept list its first kept samples can sit anywhere in the record, so the whole record comes along)
                    const uint32_t n_bytes = nrows * R + (row0 + nrows < (uint64_t)a.n_variants ? (PICK ? R : 1u) : 0u);
                    const uint32_t mis = (uint32_t)((uint64_t)(uintptr_t)rec & 15ull);
                    const uint32_t n_load = (mis + n_bytes + 15u) / 16u;   // <= 66
                    if (lane < n_load) in0[w] = *reinterpret_cast<const v4u *>(rec - mis + lane * 16u);
                    if ((lane >> 1) == (uint32_t)w) {
                        ext_addr = rec - mis + (64u + (lane & 1u)) * 16u;
                        ext_on = 64u + (lane & 1u) < n_load;
                    }
                }
            }
            if (ext_on) ext = *reinterpret_cast<const v4u *>(ext_addr);
            if (t0 != kNoItem) pending = issue_claim(range);
            // ---- round trip 2: the prefix bytes of every item (their place in the blob is known now) + item geometry
            v4u pin[NS];
            uint32_t pmis_w[NS];
#pragma unroll
            for (int w = 0; w < NS; w++) {
                pin[w] = v4u{0u, 0u, 0u, 0u};
                pmis_w[w] = 0u;
                if ((uint32_t)w >= n_here) {
                    if (lane == 0u) desc_put_item(s_desc[w][step % DS], Item{}, t0 == kNoItem ? kNoItem - 1ull : kNoItem);
                    continue;
                }
                const uint64_t row0 = (t0 + (uint64_t)w) * (uint64_t)B;
                const uint32_t nrows = (uint32_t)min((uint64_t)B, (uint64_t)a.n_variants - row0);
                const bool has_next = row0 + nrows < (uint64_t)a.n_variants;
                const uint64_t run_start = sgpr64((uint64_t)__shfl((unsigned long long)lo_v[w], 0, 64));
                const uint64_t run_end = sgpr64((uint64_t)__shfl((unsigned long long)lo_v[w], (int)nrows, 64));
                const uint64_t p_start = sgpr64((uint64_t)__shfl((unsigned long long)po_v[w], 0, 64));
                const uint64_t p_end = sgpr64((uint64_t)__shfl((unsigned long long)po_v[w], (int)(nrows + (has_next ? 1u : 0u)), 64));
                Item it;
                const uint64_t g_first = (run_start + p.head) >> 4;   // every run owns the chunk that holds its first byte ... see below
                const uint64_t g_own = (t0 + (uint64_t)w) == 0ull ? g_first : (run_start + p.head + 15ull) >> 4;
                const uint64_t g_end = (run_end + p.head + 15ull) >> 4;
                it.row = row0;
                it.g0 = g_own;
                it.cnt = (uint32_t)(g_end - g_own);
                it.lead = (uint32_t)g_own & 63u;
                it.c_first = (int64_t)(int32_t)((uint32_t)g_own * 16u - p.head - (uint32_t)run_start);
                it.delta = (int32_t)((uint32_t)((uint64_t)(uintptr_t)(a.records + row0 * (uint64_t)R) & 15ull));
                it.rec = nullptr;
                it.base = nullptr;
                it.n_load = 0u;
                it.row_tail = false;
                const uint8_t *const pb = a.prefix_blob + p_start;
                const uint32_t pmis = (uint32_t)((uint64_t)(uintptr_t)pb & 15ull);
                const uint32_t p_pieces = (pmis + (uint32_t)(p_end - p_start) + 15u) / 16u;   // <= kLrPfxBytes / 16
                if (lane < p_pieces) pin[w] = *reinterpret_cast<const v4u *>(pb - pmis + lane * 16u);
                pmis_w[w] = pmis;
                if (lane == 0u) {
                    uint8_t *d = s_desc[w][step % DS];
                    desc_put_item(d, it, t0 + (uint64_t)w);
                    *reinterpret_cast<uint32_t *>(d + 52) = pmis;
                }
            }
            // ---- park
#pragma unroll
            for (int w = 0; w < NS; w++) {
                if (step >= (uint32_t)RS) {
                    const uint32_t want = step - (uint32_t)RS + 1u;
                    while (lds_flag_read(lds_offset(&s_done[w][slot])) != want) __builtin_amdgcn_s_sleep(1);
                }
                if ((uint32_t)w < n_here) {
                    uint8_t *slab = slabs[w][slot];
                    *reinterpret_cast<v4u *>(slab + lane * 16u) = in0[w];
                    if ((lane >> 1) == (uint32_t)w) *reinterpret_cast<v4u *>(slab + (64u + (lane & 1u)) * 16u) = ext;
                    if (lane < kLrPfxBytes / 16u) *reinterpret_cast<v4u *>(slab + kLrRecBytes + lane * 16u) = pin[w];
                    if (lane < 32u) {
                        const uint64_t l0 = (uint64_t)__shfl((unsigned long long)lo_v[w], 0, 64), p0 = (uint64_t)__shfl((unsigned long long)po_v[w], 0, 64);
                        uint32_t *offs = reinterpret_cast<uint32_t *>(slab + kLrRecBytes + kLrPfxBytes);
                        offs[lane] = (uint32_t)(lo_v[w] - l0);
                        offs[32u + lane] = (uint32_t)(po_v[w] - p0);
                    }
                }
                if (lane == 0u) lds_flag_write(lds_offset(&s_full[w][slot]), step + 1u);
            }
            if (t0 == kNoItem) break;
        }
        {
            unsigned long long *const exits = reinterpret_cast<unsigned long long *>(a.work_counters + kMaxQueueRanges * 16u);
            uint32_t last = 0u;
            if (lane == 0u) last = atomicAdd(exits, 1ull) == (unsigned long long)gridDim.x - 1ull ? 1u : 0u;
            if (__builtin_amdgcn_readfirstlane(last)) {
                if (lane < kMaxQueueRanges) atomicExch(reinterpret_cast<unsigned long long *>(a.work_counters + lane * 16u), 0ull);   // lane h: head h
                if (lane == 0u) atomicExch(exits, 0ull);
            }
        }
    } else {
        // ------------------------------ storer waves -----------------------------
        const uint32_t w = wave - 1u;
        for (uint32_t step = 0;; step++) {
            const uint32_t slot = step % RS;
            while (lds_flag_read(lds_offset(&s_full[w][slot])) != step + 1u) __builtin_amdgcn_s_sleep(1);
            const uint8_t *slab = slabs[w][slot];
            const uint8_t *desc = s_desc[w][step % DS];
            uint64_t t = *reinterpret_cast<const uint64_t *>(desc + 8u);
            t = sgpr64(t);
            if (t == kNoItem - 1ull) break;
            if (t != kNoItem) {
                const Item it = desc_get_item(desc);
                const uint32_t pdelta = sgpr32(*reinterpret_cast<const uint32_t *>(desc + 52));
                emit_lines_run<PICK>(a, p, it, slab, s_stage[w], pdelta, lane, s_tab + (PICK ? kTabFront : 0u));
            }
            if (lane == 0u) lds_flag_write(lds_offset(&s_done[w][slot]), step + 1u);
        }
    }
}

}  // namespace

bool gt_wide_applicable(const EmitArgs &a)
{
    // rows of >= 4 KiB keep a wave's span reasonably full; R >= 16 for the clamped window reads
    return a.kept_idx == nullptr && a.line_off == nullptr && a.sample_count >= 1024u &&
           (a.n_variants <= 1 || a.out_stride == 4ull * a.kept_count + 1ull);
}

bool gt_wide_lines_applicable(const EmitArgs &a)
{
    // full lines through the work-queue stream kernel: all samples kept, rows of >= 4 KiB, queue heads present
    return a.kept_idx == nullptr && a.line_off != nullptr && a.prefix_off != nullptr && a.sample_count >= 1024u &&
           a.work_counters != nullptr;
}

hipError_t launch_gt_wide(const EmitArgs &a, const Tuning &t, int num_cus, hipStream_t stream)
{
    if (a.n_variants == 0) return hipSuccess;
    if (a.work_counters == nullptr) return hipErrorInvalidValue;
    WideParams p;
    p.row_bytes = 4ull * a.kept_count + 1ull;
    p.total_bytes = (uint64_t)a.n_variants * p.row_bytes;
    p.head = (uint32_t)(((uint64_t)(uintptr_t)a.out) & 127ull);
    // interleaved in one process (profiles/r01_kernel_sweeps.md, chr22 block): 8 ranges 2.058 ms, 4: 2.045, 2: 2.038, 1: 2.039
    p.run_rows = 0u;
    p.run_rec = 0u;
    p.magic = 0u;
    p.pfx_shift = 0u;
    // a row owns floor(S/16) or ceil(S/16) chunks (one more for row 0 with an unaligned pointer), and
    // its first span starts up to 63 chunks before them (1-KiB-aligned span grid)
    const uint64_t max_row_chunks = (p.row_bytes + 15ull) / 16ull + 1ull + 63ull;
    p.spans_per_row = (uint32_t)((max_row_chunks + kSpanChunks - 1ull) / kSpanChunks);
    p.n_items = (uint64_t)a.n_variants * p.spans_per_row;
    // Write fronts of the launch (set below, once the grid is known).  Rows of several spans on a launch of >= 64 steps per block: 8 ranges — on the 212-GB launch of BASELINE configs[2] two fronts ran at 0.73
    // or 0.78 of roofline depending on where the driver had put the buffer (stable per allocation, 5-7 % apart), eight at 0.784 on
    // every box; 0.765 -> 0.784 on the 266-GB shard of configs[3]; +0.4-2.4 % on 12-GB launches of N = 5 000 .. 100 000.  Rows of one
    // span (the chr22 shape): 2 — four or eight were level on one box and 1.6-2.4 % behind on another; short launches (the 0.8-GB second
    // pass of the two-pass path: 15 steps per block): 2 — eight cost 9 % there (profiles/r02_kernel_sweeps.md).

    // 1 loader + 7 storer waves, nontemporal stores, plain store steps in bursts of 2 (text of both first, then both
    // stores): the measured best of the round-1 A/Bs (3 storers, plain stores, bursts of 1/4/8: profiles/r01_kernel_sweeps.md)
    const uint64_t need = (p.n_items + 6ull) / 7ull;
    void (*dk)(EmitArgs, WideParams);
    if (a.line_off)
        dk = gathered(a) ? gt_stream_dyn_kernel<7, true, true, true> : gt_stream_dyn_kernel<7, false, true, true>;
    else
        dk = gathered(a) ? gt_stream_dyn_kernel<7, true, true> : gt_stream_dyn_kernel<7, false, true>;
    // launch exactly what is resident (62 VGPRs -> 8 waves/SIMD -> four 512-thread blocks per CU; 25 KB of LDS
    // each): blocks beyond that would only start when the queue is already empty.  Interleaved A/B on the chr22
    // block: 2 blocks/CU 2.12 ms, 4 blocks/CU 2.00 ms (profiles/r01_kernel_sweeps.md)
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dk, 512, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    // ... on long launches.  SHORT launches (up to ~1 GB of text: a CLI block, a chunk of the two passes) run faster with two blocks per
    // CU (round 3, profiles/r03_kernel_sweeps.md §6).  On a re-used output buffer (what the CLI does) the advantage is large and holds up to
    // ~4.5 GB (N = 2 504: 134 MB 0.51 against 0.45 of roofline, 2.2 GB 0.77 / 0.71, 6 GB 0.69 / 0.73); on FRESH output regions (one call cut into
    // pieces of that size, tools/split_probe.py) it is small and ends earlier: 134 MB 0.457 / 0.442, 0.5 GB level (K = 4 940: 0.584 / 0.569),
    // 2 GB 0.613 / 0.642 — so the rule takes the size both agree on.
    if ((uint64_t)a.n_variants * p.row_bytes <= 1000000000ull && per_cu > 2) per_cu = 2;
    if (t.wide_blocks_per_cu > 0) per_cu = t.wide_blocks_per_cu;
    const uint64_t cap = (uint64_t)num_cus * (uint64_t)per_cu;
    const uint32_t g = (uint32_t)(need < cap ? need : cap);
    p.n_ranges = t.wide_ranges > 0 ? (uint32_t)t.wide_ranges : (p.spans_per_row > 1u && need >= 64ull * g ? 8u : 2u);
    // LINES: the storer waves copy the prefixes themselves before their first item (lanes per line by the longest prefix: prefix_copy_shift)
    p.pfx_shift = prefix_copy_shift(a);
    hipLaunchKernelGGL(dk, dim3(g), dim3(512), 0, stream, a, p);
    return hipGetLastError();
}

// ---- RUNS mode: short rows, B rows per work item ---------------------------------------------------------
static uint32_t run_rows_for(const EmitArgs &a)
{
    if (a.record_size == 0u) return 0u;
    const uint32_t S = 4u * a.kept_count + 1u;
    // one 16-B-per-lane load (+ the two shared extra pieces) covers the run's record bytes at any alignment + 1 byte:
    // 15 + B*R + 1 <= 66 * 16; the run's chunks (+ up to 63 of lead) fit one 1 024-chunk span: B*S/16 + 2 + 63 <= 1 024
    const uint32_t by_load = 1040u / a.record_size;
    const uint32_t by_span = 15328u / S;
    return by_load < by_span ? by_load : by_span;
}

bool gt_runs_applicable(const EmitArgs &a)
{
    // all samples kept, dense records AND dense text (both are then contiguous over a run of rows), rows of >= 33 bytes
    // (a 16-byte chunk meets at most one '\n') and at least one whole row per item (N <= 3 831)
    return a.kept_idx == nullptr && a.line_off == nullptr && !gathered(a) && a.sample_count >= 8u &&
           (a.n_variants <= 1 || (a.out_stride == 4ull * a.kept_count + 1ull && a.record_stride == a.record_size)) &&
           a.work_counters != nullptr && run_rows_for(a) >= 1u;
}

// AUTO takes the RUNS mode where an item holds at least two rows (N <= 1 915); with one row per item it is the row-item
// kernel's work item with LDS-staged text, measured against it in profiles/r02_kernel_sweeps.md
bool gt_runs_preferred(const EmitArgs &a) { return gt_runs_applicable(a) && run_rows_for(a) >= 2u; }

hipError_t launch_gt_runs(const EmitArgs &a, const Tuning &t, int num_cus, hipStream_t stream)
{
    if (a.n_variants == 0) return hipSuccess;
    if (!gt_runs_applicable(a)) return hipErrorInvalidValue;
    WideParams p;
    p.row_bytes = 4ull * a.kept_count + 1ull;
    p.total_bytes = (uint64_t)a.n_variants * p.row_bytes;
    p.head = (uint32_t)(((uint64_t)(uintptr_t)a.out) & 127ull);
    p.n_ranges = t.wide_ranges > 0 ? (uint32_t)t.wide_ranges : 2u;   // (1 .. 64 measured level in this mode)
    p.spans_per_row = 1u;
    p.run_rows = t.runs_rows > 0 && (uint32_t)t.runs_rows < run_rows_for(a) ? (uint32_t)t.runs_rows : run_rows_for(a);
    p.run_rec = a.record_size;
    p.magic = (uint32_t)(0x100000000ull / p.row_bytes) + 1u;
    p.pfx_shift = 0u;
    p.n_items = ((uint64_t)a.n_variants + p.run_rows - 1ull) / p.run_rows;
    const uint64_t need = (p.n_items + 6ull) / 7ull;
    void (*dk)(EmitArgs, WideParams) = gt_stream_dyn_kernel<7, false, true, false, 2, true>;
    // two blocks per CU, not the three the occupancy API allows (45 KB of LDS each): level on 11-GB launches (N = 100 / 300 / 1 000 /
    // 1 900: 0.686 / 0.680 / 0.698 / 0.707 against 0.683 / 0.681 / 0.703 / 0.690) and 8-9 % ahead on the 255-MB launch of the
    // reference's own dataset shape (0.505 against 0.469), whose blocks run only three steps each
    // (round 3: on one box three were ahead again from ~5.5 GB of re-used output — 6.4 M rows of N = 300: 0.73 against 0.69 — which round 2's
    // boxes did not show; left at two)
    int per_cu = 2;
    if (t.wide_blocks_per_cu > 0) per_cu = t.wide_blocks_per_cu;
    const uint64_t cap = (uint64_t)num_cus * (uint64_t)per_cu;
    hipLaunchKernelGGL(dk, dim3((uint32_t)(need < cap ? need : cap)), dim3(512), 0, stream, a, p);
    return hipGetLastError();
}

// ---- runs of full LINES: short rows, all samples kept, dense records ---------------------------------------------------------
static uint32_t lineruns_rows_for(const EmitArgs &a)
{
    if (a.record_size == 0u || a.max_line_bytes == 0ull || a.max_line_bytes > 15328ull) return 0u;
    const uint32_t row_text = 4u * a.kept_count + 1u;
    const uint32_t max_prefix = (uint32_t)a.max_line_bytes - row_text;
    uint32_t b = 1040u / a.record_size;                                        // one wide load of the run's records (+ 1 byte)
    if (a.kept_idx && b) b--;                                                   // kept list: + the whole record behind the run
    b = std::min<uint32_t>(b, 15328u / (uint32_t)a.max_line_bytes);            // the run's chunks (+ lead) fit one span
    b = std::min<uint32_t>(b, kLrMaxRows);
    if (max_prefix) b = std::min<uint32_t>(b, (kLrPfxBytes - 16u) / max_prefix - (((kLrPfxBytes - 16u) / max_prefix) ? 1u : 0u));  // prefixes of B + 1 lines
    return b;
}

uint32_t gt_lineruns_rows(const EmitArgs &a) { return lineruns_rows_for(a); }

bool gt_lineruns_applicable(const EmitArgs &a)
{
    // all samples, or a kept list of >= 8 samples out of <= 4 096 (its u16 copy lives in LDS); lines of >= 33 bytes of GT text (a
    // 16-byte chunk meets at most one '\n'); dense records; at least two lines per item
    const bool samples_ok = a.kept_idx == nullptr ? a.sample_count >= 8u : (a.sample_count <= kTabMaxSamples && a.kept_count >= 8u);
    return samples_ok && a.line_off != nullptr && a.prefix_off != nullptr && !gathered(a) &&
           (a.n_variants <= 1 || a.record_stride == a.record_size) && a.work_counters != nullptr && lineruns_rows_for(a) >= 2u;
}

hipError_t launch_gt_lineruns(const EmitArgs &a, const Tuning &t, int num_cus, hipStream_t stream)
{
    if (a.n_variants == 0) return hipSuccess;
    if (!gt_lineruns_applicable(a)) return hipErrorInvalidValue;
    WideParams p;
    p.row_bytes = 4ull * a.kept_count + 1ull;
    p.total_bytes = 0ull;
    p.head = (uint32_t)(((uint64_t)(uintptr_t)a.out) & 127ull);
    p.n_ranges = t.wide_ranges > 0 ? (uint32_t)t.wide_ranges : 2u;   // (1 .. 64 measured level in this mode)
    p.spans_per_row = 1u;
    const uint32_t b_max = lineruns_rows_for(a);
    p.run_rows = t.runs_rows > 0 && (uint32_t)t.runs_rows < b_max ? (uint32_t)t.runs_rows : b_max;
    p.run_rec = a.record_size;
    p.magic = 0u;
    p.pfx_shift = 0u;
    p.n_items = ((uint64_t)a.n_variants + p.run_rows - 1ull) / p.run_rows;
    // THREE storer waves per loader wave, not the stream kernel's seven: this loader has two dependent round trips and about as many
    // instructions per item as a storer has per 4-KiB group, so with seven storers it is the wave everybody waits for; and its
    // per-item registers (offsets, records, prefix pieces) set the kernel's VGPR count (110 with seven items in flight, 63 with
    // three: six 256-thread blocks per CU).  Storers per loader, fraction of roofline at N = 100 / 300 / 1 000 with 30-byte prefixes:
    // 1: 0.18 / 0.32 / 0.26, 2: 0.28 / 0.41 / 0.45, 3: 0.30 / 0.44 / 0.49, 4: 0.24 / 0.36 / 0.40, 5: 0.20 / 0.29 / 0.32, 7: 0.26 / 0.38 / 0.43
    // (profiles/r02_kernel_sweeps.md; the GT-only RUNS mode, one round trip and a lighter loader, stays best with seven)
    constexpr int kStorers = 3;
    const uint64_t need = (p.n_items + (uint64_t)kStorers - 1ull) / (uint64_t)kStorers;
    void (*dk)(EmitArgs, WideParams) = a.kept_idx ? gt_lineruns_kernel<kStorers, true> : gt_lineruns_kernel<kStorers, false>;
    constexpr int threads = 64 * (kStorers + 1);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dk, threads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (t.wide_blocks_per_cu > 0) per_cu = t.wide_blocks_per_cu;
    const uint64_t cap = (uint64_t)num_cus * (uint64_t)per_cu;
    hipLaunchKernelGGL(dk, dim3((uint32_t)(need < cap ? need : cap)), dim3(threads), 0, stream, a, p);
    return hipGetLastError();
}

}  // namespace pgenhip
