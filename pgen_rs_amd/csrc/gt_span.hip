// gt_span.hip — dense all-samples kernel over STREAM SPANS (gfx950 / MI355X): the bench kernel
// for rows of >= 8 KiB of text (N >= 2048).
//
// Same contract as gt_wide.hip's stream kernels (K = N, rows packed at 4N+1 bytes; roles: wave 0
// of a 512-thread block only loads, waves 1-7 only store; LDS slab ring + descriptor ring; work
// queue), but the unit of work is no longer "the chunks of one row": it is one 1024-chunk
// (16 KiB), 1-KiB-aligned SPAN of the launch's output stream, whatever rows it crosses.  A
// 10 017-byte row (N = 2 504) is 626-627 chunks: with per-row items the first and last of its
// 10-11 store steps are partly masked (+10 % store instructions, measured 1.20e7 vs 1.08e7 ideal
// on the chr22 block); with spans EVERY store step is a full 1 KiB of eight whole 128-B lines,
// rows simply begin and end inside it.  A span touches at most three rows here (S >= 8193): the
// loader stages up to three record pieces in one slab (1-2 wide load instructions for all of
// them together), the storer picks each lane's piece with two compares.
//
// Reference semantics: /root/reference/src/pfile.rs:165-190.
#include <stdlib.h>

#include "gt_common.hip.h"
#include "kernels.h"

namespace pgenhip {

namespace {

constexpr int kNS = 7;                        // storer waves per block
constexpr int kThreads = 64 * (kNS + 1);
constexpr uint32_t kSpanChunks = 1024;        // 16 stores x 64 lanes
constexpr uint32_t kMaxPieces = 3;
constexpr uint32_t kSlabBytes = 1216;         // <= 1024 + 3 x (15 + 16 + 1) staged bytes, 16-B pieces (76 x 16)
constexpr uint32_t kSlabExtra = 16;           // +0: first record byte of the row after the span's last row
constexpr uint32_t kExtSlots = kSlabBytes / 16u - 64u;  // 12 staged 16-B slots beyond the first 64 of a span
constexpr uint32_t kDescBytes = 128;
constexpr int kRingSlots = 3;
constexpr int kDescSlots = kRingSlots + 1;
constexpr uint32_t kNone = 0xFFFFu;           // "no such chunk / piece in this span"

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

struct SpanParams {
    uint64_t row_bytes;    // S = 4N + 1
    uint64_t total_bytes;  // T = V * S
    uint64_t n_chunks;     // chunks of the 128-B-anchored grid touched by the stream
    uint64_t n_items;      // spans
    uint32_t head;         // out address & 127
};

// one row's share of a span (all wave-uniform)
struct Piece {
    uint32_t start;   // first chunk (relative to the span) owned by this row
    uint32_t tail;    // relative chunk that holds this row's '\n', kNone if it lies outside the span
    int32_t bf;       // record byte of the first chunk's window (-1 only for the stream's head chunk)
    uint32_t phase;   // c_first & 15: text phase of the row inside 16-B chunks
    int32_t sdelta;   // slab offset of this row's record byte 0
};

// NB: the three pieces are kept as separate named members, not as an array: a piece is picked per lane / per
// step at run time, and hipcc turns `q == 0 ? pc[0] : ...` on an array back into an indexed load from a
// SCRATCH copy (scratch_load_dword = VMEM: 5.7e7 extra loads per launch and a vmcnt wait in every store
// step, measured).  Selects over individual registers stay s_cselect / v_cndmask.
struct Span {
    uint64_t base_chunk;  // absolute index of relative chunk 0
    uint32_t lo, hi;      // valid relative chunks [lo, hi)
    uint32_t n_pieces;
    uint64_t row0;        // row of piece 0
    Piece p0, p1, p2;
};

__device__ __forceinline__ uint32_t pick(uint32_t q, uint32_t a0, uint32_t a1, uint32_t a2) { return q == 0u ? a0 : (q == 1u ? a1 : a2); }
__device__ __forceinline__ int32_t pick(uint32_t q, int32_t a0, int32_t a1, int32_t a2) { return q == 0u ? a0 : (q == 1u ? a1 : a2); }

// loader-side view: where each piece's bytes come from
struct SpanLoads {
    const uint8_t *src[kMaxPieces];  // 16-B-aligned global address of the piece's first staged byte
    uint32_t cum[kMaxPieces + 1];    // 16-B slots before each piece (cum[n_pieces] = total)
};

template <bool HAS_VIDX>
__device__ __forceinline__ const uint8_t *row_record(const EmitArgs &a, uint64_t r)
{
    const uint64_t src = HAS_VIDX ? (uint64_t)a.variant_idx[r] : r;
    return a.records + src * a.record_stride;
}

// row_hint: the row that holds the span's first byte when the caller knows it (consecutive spans: the row the
// previous span ended in, or the one after it), kNoHint = find it with a 64-bit division.
constexpr uint64_t kNoHint = ~0ull;

template <bool HAS_VIDX, bool WITH_LOADS>
__device__ __forceinline__ Span make_span(const EmitArgs &a, const SpanParams &p, uint64_t t, SpanLoads *ld, uint64_t row_hint)
{
    Span sp;
    const uint64_t S = p.row_bytes;
    const uint32_t last = a.record_size - 1u;
    sp.base_chunk = t * kSpanChunks;
    const uint64_t first_chunk = (uint64_t)(p.head >> 4);
    sp.lo = sp.base_chunk < first_chunk ? (uint32_t)(first_chunk - sp.base_chunk) : 0u;
    const uint64_t end_abs = min(sp.base_chunk + kSpanChunks, p.n_chunks);
    sp.hi = (uint32_t)(end_abs - sp.base_chunk);
    const int64_t o_lo = (int64_t)((sp.base_chunk + sp.lo) * 16ull) - (int64_t)p.head;  // stream offset of chunk lo
    uint64_t r = row_hint != kNoHint ? row_hint : (o_lo <= 0 ? 0ull : (uint64_t)o_lo / S);
    sp.row0 = r;
    uint32_t start = sp.lo;
    uint32_t slots = 0u;
    sp.n_pieces = 0u;
    auto one_piece = [&](uint32_t q, Piece &pc) {
        pc.start = kNone;
        pc.tail = kNone;
        pc.bf = 0;
        pc.phase = 0u;
        pc.sdelta = 0;
        if (WITH_LOADS) {
            ld->src[q] = a.records;
            ld->cum[q] = slots;
        }
        if (start < sp.hi && q == sp.n_pieces) {
            const uint64_t row_start = r * S;
            // row r owns the chunks whose first byte lies in [r*S, (r+1)*S)
            const uint64_t g_end = (row_start + S + p.head + 15ull) >> 4;  // first chunk of row r+1
            const uint32_t end_rel = g_end - sp.base_chunk < (uint64_t)sp.hi ? (uint32_t)(g_end - sp.base_chunk) : sp.hi;
            const int64_t c_first = (int64_t)((sp.base_chunk + start) * 16ull) - (int64_t)p.head - (int64_t)row_start;
            const int64_t bf = c_first >> 4;
            const uint32_t cnt = end_rel - start;
            pc.start = start;
            pc.bf = (int32_t)bf;
            pc.phase = (uint32_t)c_first & 15u;
            // the row's '\n' sits in its last owned chunk; it is in this span iff the row ends here
            pc.tail = (g_end - sp.base_chunk <= (uint64_t)sp.hi) ? end_rel - 1u : kNone;
            const uint32_t b_first = bf > 0 ? (uint32_t)min(bf, (int64_t)last - 1) : 0u;
            const uint32_t b_last = (uint32_t)min(bf + (int64_t)cnt, (int64_t)last);
            const uint8_t *rec = row_record<HAS_VIDX>(a, r);
            const uint32_t mis = (uint32_t)(((uint64_t)(uintptr_t)(rec + b_first)) & 15ull);
            pc.sdelta = (int32_t)(slots * 16u + mis) - (int32_t)b_first;
            if (WITH_LOADS) ld->src[q] = rec + b_first - mis;
            slots += (mis + (b_last - b_first)) / 16u + 1u;
            sp.n_pieces = q + 1u;
            start = end_rel;
            r++;
        }
    };
    one_piece(0u, sp.p0);
    one_piece(1u, sp.p1);
    one_piece(2u, sp.p2);
    if (WITH_LOADS) ld->cum[kMaxPieces] = slots;
    return sp;
}

// ---- LDS flag words and descriptor hand-over (explicit DS instructions; see gt_wide.hip) ------
// readfirstlane returns a signed int: widen through uint32_t
__device__ __forceinline__ uint32_t sgpr32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t sgpr64(uint64_t v)
{
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
    return ((uint64_t)hi << 32) | (uint64_t)lo;
}

// descriptor layout: +0 u64 tag (item index; ~0-1 = out of work)  +8 u64 base_chunk  +16 u64 row0
// +24 u32 lo  +28 u32 hi  +32 u32 n_pieces  then per piece at +40 + 20*q: start, tail, bf, phase, sdelta
__device__ __forceinline__ void desc_put(uint8_t *x, const Span &sp, uint64_t tag)
{
    *reinterpret_cast<uint64_t *>(x) = tag;
    *reinterpret_cast<uint64_t *>(x + 8) = sp.base_chunk;
    *reinterpret_cast<uint64_t *>(x + 16) = sp.row0;
    *reinterpret_cast<uint32_t *>(x + 24) = sp.lo;
    *reinterpret_cast<uint32_t *>(x + 28) = sp.hi;
    *reinterpret_cast<uint32_t *>(x + 32) = sp.n_pieces;
    auto put = [&](uint32_t q, const Piece &pc) {
        uint32_t *w = reinterpret_cast<uint32_t *>(x + 40 + 20 * q);
        w[0] = pc.start;
        w[1] = pc.tail;
        w[2] = (uint32_t)pc.bf;
        w[3] = pc.phase;
        w[4] = (uint32_t)pc.sdelta;
    };
    put(0u, sp.p0);
    put(1u, sp.p1);
    put(2u, sp.p2);
}

__device__ __forceinline__ Span desc_get(const uint8_t *x)
{
    Span sp;
    sp.base_chunk = sgpr64(*reinterpret_cast<const uint64_t *>(x + 8));
    sp.row0 = sgpr64(*reinterpret_cast<const uint64_t *>(x + 16));
    sp.lo = sgpr32(*reinterpret_cast<const uint32_t *>(x + 24));
    sp.hi = sgpr32(*reinterpret_cast<const uint32_t *>(x + 28));
    sp.n_pieces = sgpr32(*reinterpret_cast<const uint32_t *>(x + 32));
    auto get = [&](uint32_t q, Piece &pc) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(x + 40 + 20 * q);
        pc.start = sgpr32(w[0]);
        pc.tail = sgpr32(w[1]);
        pc.bf = (int32_t)sgpr32(w[2]);
        pc.phase = sgpr32(w[3]);
        pc.sdelta = (int32_t)sgpr32(w[4]);
    };
    get(0u, sp.p0);
    get(1u, sp.p1);
    get(2u, sp.p2);
    return sp;
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

__device__ __forceinline__ u32x4 text_from_window(uint32_t w, uint32_t psh)
{
    const uint32_t t0 = gt_text(w & 3u), t1 = gt_text((w >> 2) & 3u), t2 = gt_text((w >> 4) & 3u);
    const uint32_t t3 = gt_text((w >> 6) & 3u), t4 = gt_text((w >> 8) & 3u);
    u32x4 v;
    v.x = funnel_bytes(t0, t1, psh);
    v.y = funnel_bytes(t1, t2, psh);
    v.z = funnel_bytes(t2, t3, psh);
    v.w = funnel_bytes(t3, t4, psh);
    return v;
}

// The span's 16 store steps.  `slab` holds the staged record pieces, slab[kSlabBytes] the first
// record byte of the row that follows the span's last row.
template <bool HAS_VIDX, bool NT>
__device__ __forceinline__ void emit_span(const EmitArgs &a, const SpanParams &p, const Span &sp, const uint8_t *slab, uint32_t lane)
{
    const uint64_t S = p.row_bytes;
    const uint64_t gt_bytes = S - 1ull;
    const uint32_t last = a.record_size - 1u;
    uint8_t *const chunk0 = a.out - p.head;
    uint8_t *const lane_ptr = chunk0 + sp.base_chunk * 16ull + lane * 16u;  // lane's chunk in step 0
    const bool head_partial = sp.base_chunk == 0ull && (p.head & 15u) != 0u;  // chunk `lo` starts before the stream

    // ---- plain steps: all 64 chunks inside ONE row's text (no '\n', no row change, no stream edge).  Their
    // range is computed once per piece, so the per-step scalar work is a loop counter and an address bump
    // (the first version re-derived the piece in every step and the kernel became SALU-bound: 1.0e9 scalar
    // instructions per launch, ~80 % of the scalar unit).
    auto plain_range = [&](uint32_t q, const Piece &pc, uint32_t &ub, uint32_t &ue) {
        const uint32_t from = max(pc.start + ((head_partial && q == 0u) ? 1u : 0u), sp.lo);
        const uint32_t lim = pc.tail == kNone ? sp.hi : pc.tail;  // first chunk that is no longer plain interior
        ub = (from + 63u) >> 6;
        ue = lim >> 6;
        if (pc.start == kNone || ue < ub) ue = ub = 0u;
    };
    auto plain_steps = [&](uint32_t ub, uint32_t ue, const Piece &pc) {
        const int32_t off0 = pc.bf + pc.sdelta - (int32_t)pc.start + (int32_t)lane;
        const uint32_t sh = ((pc.phase >> 2) & 3u) * 2u, psh = pc.phase & 3u;
        for (uint32_t u = ub; u < ue; u++) {
            uint16_t h;
            __builtin_memcpy(&h, slab + off0 + (int32_t)(u * 64u), 2);
            store_chunk<NT>(lane_ptr + u * 1024u, text_from_window((uint32_t)h >> sh, psh));
        }
    };
    uint32_t ub0, ue0, ub1, ue1, ub2, ue2;
    plain_range(0u, sp.p0, ub0, ue0);
    plain_range(1u, sp.p1, ub1, ue1);
    plain_range(2u, sp.p2, ub2, ue2);
    plain_steps(ub0, ue0, sp.p0);
    plain_steps(ub1, ue1, sp.p1);
    plain_steps(ub2, ue2, sp.p2);

    // ---- the few remaining steps (row changes, stream edges)
    for (uint32_t u = 0; u < kSpanChunks / 64u; u++) {
        const uint32_t s0 = u * 64u;
        if (s0 >= sp.hi) break;
        if (s0 + 64u <= sp.lo) continue;
        if ((u >= ub0 && u < ue0) || (u >= ub1 && u < ue1) || (u >= ub2 && u < ue2)) continue;
        // ---- general step: lanes pick their row piece; row tails merge two rows; stream edges go byte-wise
        const uint32_t idx = s0 + lane;
        if (idx < sp.lo || idx >= sp.hi) continue;
        const uint32_t q = (idx >= sp.p1.start ? 1u : 0u) + (idx >= sp.p2.start ? 1u : 0u);
        const uint32_t start = pick(q, sp.p0.start, sp.p1.start, sp.p2.start);
        const uint32_t tail = pick(q, sp.p0.tail, sp.p1.tail, sp.p2.tail);
        const int32_t bf = pick(q, sp.p0.bf, sp.p1.bf, sp.p2.bf);
        const uint32_t phase = pick(q, sp.p0.phase, sp.p1.phase, sp.p2.phase);
        const int32_t sdelta = pick(q, sp.p0.sdelta, sp.p1.sdelta, sp.p2.sdelta);
        const uint32_t i = idx - start;
        const int32_t b0 = bf + (int32_t)i;
        uint32_t window;
        {
            const int32_t bb = max(0, min(b0, (int32_t)last - 1));
            uint16_t h;
            __builtin_memcpy(&h, slab + bb + sdelta, 2);
            const int32_t d = b0 - bb;
            window = d < 0 ? ((uint32_t)h << 8) & 0xFFFFu : (uint32_t)h >> (8u * (uint32_t)min(d, 2));
        }
        uint8_t *dst = lane_ptr + u * 1024u;
        u32x4 v = text_from_window(window >> (((phase >> 2) & 3u) * 2u), phase & 3u);
        bool whole = idx != tail && !(head_partial && idx == sp.lo);
        bool bytewise = false;
        if (!whole) {
            const uint64_t row = sp.row0 + q;
            const int64_t o = (int64_t)((sp.base_chunk + idx) * 16ull) - (int64_t)p.head;  // stream offset of the chunk
            const int64_t c = o - (int64_t)(row * S);                                       // offset inside the row (may be < 0 at the head)
            const uint32_t nl = (uint32_t)((int64_t)gt_bytes - c);                          // 0..15 at a row tail
            if (idx == tail && c >= 0 && (row + 1ull < a.n_variants || nl == 15u)) {
                // row tail: bytes < nl from this row, '\n' at nl, the rest is the head of row+1
                u32x4 y = {0u, 0u, 0u, 0u};
                if (nl < 15u) {
                    // first record byte of row+1: staged with the next piece when that piece is in this span
                    const bool next_here = q + 1u < sp.n_pieces;
                    const int32_t nsd = q == 0u ? sp.p1.sdelta : sp.p2.sdelta;
                    const uint32_t nb0 = next_here ? (uint32_t)slab[nsd] : (uint32_t)slab[kSlabBytes];
                    const uint32_t ph = (16u - (nl + 1u)) & 15u;  // phase of row+1's text in this chunk: it starts nl+1 bytes in
                    y = text_from_window((nb0 << 8) >> (((ph >> 2) & 3u) * 2u), ph & 3u);
                }
                uint32_t xs[4] = {v.x, v.y, v.z, v.w};
                uint32_t ys[4] = {y.x, y.y, y.z, y.w};
                uint32_t os[4];
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const int32_t nb = (int32_t)nl - 4 * m;
                    const uint32_t mask = nb >= 4 ? 0xFFFFFFFFu : (nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u));
                    uint32_t d = (xs[m] & mask) | (ys[m] & ~mask);
                    if (nb >= 0 && nb < 4) d = (d & ~(0xFFu << (8 * nb))) | (0x0Au << (8 * nb));
                    os[m] = d;
                }
                v = u32x4{os[0], os[1], os[2], os[3]};
                whole = true;
            } else {
                bytewise = true;
            }
        }
        // first / last chunk of the whole stream (twice per launch): byte-wise with direct loads.  Guarded by a
        // wave-UNIFORM branch: this block contains s_waitcnt vmcnt(0), which a storer must not execute in the
        // common case (scalar instructions ignore EXEC, and a plain divergent `if` was not branched over here —
        // measured: 5.7e7 VMEM read instructions per launch instead of 1.4e6 and every general step drained)
        if (__ballot(bytewise) != 0ull) {
            if (bytewise) {
                const uint64_t row = sp.row0 + q;
                const int64_t o = (int64_t)((sp.base_chunk + idx) * 16ull) - (int64_t)p.head;
                uint64_t rr = row;
                int64_t cc = o - (int64_t)(row * S);
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
}

template <bool HAS_VIDX, bool NT>
__global__ __launch_bounds__(kThreads, 8) void gt_span_kernel(EmitArgs a, SpanParams p)
{
    __shared__ __attribute__((aligned(16))) uint8_t slabs[kNS][kRingSlots][kSlabBytes + kSlabExtra];
    __shared__ __attribute__((aligned(16))) uint8_t s_desc[kNS][kDescSlots][kDescBytes];
    __shared__ uint32_t s_full[kNS][kRingSlots];
    __shared__ uint32_t s_done[kNS][kRingSlots];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < kNS * kRingSlots) {
        (&s_full[0][0])[threadIdx.x] = 0u;
        (&s_done[0][0])[threadIdx.x] = 0u;
    }
    __syncthreads();
    constexpr uint64_t kNoItem = ~0ull;

    if (wave == 0u) {
        // ------------------------------ loader wave ------------------------------
        const uint64_t per_range = (p.n_items + 7ull) / 8ull;
        uint32_t range = blockIdx.x & 7u;
        uint32_t drained = 0u;
        for (uint64_t step = 0;; step++) {
            // claim kNS consecutive spans from this range's head word (steal from the next range when drained)
            uint64_t t0 = kNoItem;
            while (drained < 8u) {
                const uint64_t lo = (uint64_t)range * per_range;
                const uint64_t hi = min(lo + per_range, p.n_items);
                uint64_t got = 0;
                if (lane == 0u) got = atomicAdd(reinterpret_cast<unsigned long long *>(a.work_counters + range * 16u), (unsigned long long)kNS);
                got = sgpr64(got);
                if (lo + got < hi) {
                    t0 = lo + got;
                    break;
                }
                range = (range + 1u) & 7u;
                drained++;
            }
            const uint64_t t_end = t0 == kNoItem ? 0ull : min(t0 + kNS, min(((uint64_t)range + 1ull) * per_range, p.n_items));
            const uint32_t slot = (uint32_t)(step % kRingSlots);
            // in0[w]: slots 0-63 of span w.  Slots 64-75 of ALL spans share two register quads: lane group g = lane / 12
            // holds the extra slots of span g in ext_a (spans 0-3) and of span 4 + g in ext_b (spans 4-6), fetched by ONE
            // load each after the span loop (per-span in1 registers cost 28 VGPRs and a block of occupancy)
            v4u in0[kNS];
            v4u ext_a = v4u{0u, 0u, 0u, 0u}, ext_b = v4u{0u, 0u, 0u, 0u};
            const uint8_t *ext_a_addr = a.records, *ext_b_addr = a.records;
            bool ext_a_on = false, ext_b_on = false;
            const uint32_t ext_group = lane / kExtSlots, ext_e = lane - ext_group * kExtSlots;
            uint32_t nb[kNS];
            uint64_t row_hint = kNoHint;  // one 64-bit division per claim; the following spans continue where the last ended
#pragma unroll
            for (int w = 0; w < kNS; w++) {
                const uint64_t t = t0 == kNoItem ? kNoItem : t0 + (uint64_t)w;
                nb[w] = 0u;
                in0[w] = v4u{0u, 0u, 0u, 0u};
                const bool have = t0 != kNoItem && t < t_end;
                const uint64_t tag = t0 == kNoItem ? kNoItem - 1ull : (have ? t : kNoItem);
                if (!have) {
                    if (lane == 0u) *reinterpret_cast<uint64_t *>(s_desc[w][step % kDescSlots]) = tag;
                } else {
                    SpanLoads ld;
                    const Span sp = make_span<HAS_VIDX, true>(a, p, t, &ld, row_hint);
                    if (lane == 0u) desc_put(s_desc[w][step % kDescSlots], sp, tag);
                    // one 16-B slot per lane; a lane finds its piece with two compares
                    {
                        const uint32_t s = lane;
                        const uint32_t q = (s >= ld.cum[1] ? 1u : 0u) + (s >= ld.cum[2] ? 1u : 0u);
                        const uint8_t *src = q == 0u ? ld.src[0] : (q == 1u ? ld.src[1] : ld.src[2]);
                        const uint32_t c0 = q == 0u ? ld.cum[0] : (q == 1u ? ld.cum[1] : ld.cum[2]);
                        if (s < ld.cum[kMaxPieces]) in0[w] = *reinterpret_cast<const v4u *>(src + (s - c0) * 16u);
                    }
                    if (ext_group == (uint32_t)(w < 4 ? w : w - 4)) {
                        const uint32_t s = ext_e + 64u;
                        const uint32_t q = (s >= ld.cum[1] ? 1u : 0u) + (s >= ld.cum[2] ? 1u : 0u);
                        const uint8_t *src = q == 0u ? ld.src[0] : (q == 1u ? ld.src[1] : ld.src[2]);
                        const uint32_t c0 = q == 0u ? ld.cum[0] : (q == 1u ? ld.cum[1] : ld.cum[2]);
                        if (w < 4) {
                            ext_a_addr = src + (s - c0) * 16u;
                            ext_a_on = s < ld.cum[kMaxPieces];
                        } else {
                            ext_b_addr = src + (s - c0) * 16u;
                            ext_b_on = s < ld.cum[kMaxPieces];
                        }
                    }
                    // the span's last chunk may hold its last row's '\n': then the head of the NEXT row (not staged) is needed
                    const uint32_t lp_tail = pick(sp.n_pieces - 1u, sp.p0.tail, sp.p1.tail, sp.p2.tail);
                    const uint64_t last_row = sp.row0 + sp.n_pieces - 1ull;
                    if (lp_tail != kNone && last_row + 1ull < a.n_variants && lane == 0u)
                        nb[w] = (uint32_t)row_record<HAS_VIDX>(a, last_row + 1ull)[0];
                    // the next span starts in the row this one stopped in, or in the next one if that row ended here
                    row_hint = last_row + (lp_tail != kNone ? 1ull : 0ull);
                }
            }
            if (ext_a_on) ext_a = *reinterpret_cast<const v4u *>(ext_a_addr);
            if (ext_b_on) ext_b = *reinterpret_cast<const v4u *>(ext_b_addr);
#pragma unroll
            for (int w = 0; w < kNS; w++) {
                if (step >= (uint64_t)kRingSlots) {
                    const uint32_t want = (uint32_t)(step - kRingSlots) + 1u;
                    while (lds_flag_read(lds_offset(&s_done[w][slot])) != want) __builtin_amdgcn_s_sleep(1);
                }
                uint8_t *slab = slabs[w][slot];
                *reinterpret_cast<v4u *>(slab + lane * 16u) = in0[w];
                if (ext_group == (uint32_t)(w < 4 ? w : w - 4) && lane < 5u * kExtSlots) *reinterpret_cast<v4u *>(slab + (ext_e + 64u) * 16u) = w < 4 ? ext_a : ext_b;
                if (lane == 0u) slab[kSlabBytes] = (uint8_t)nb[w];
                if (lane == 0u) lds_flag_write(lds_offset(&s_full[w][slot]), (uint32_t)step + 1u);
            }
            if (t0 == kNoItem) break;
        }
        // ---- self-cleaning queue: the last loader wave to leave re-zeroes the heads for the next launch (every block's
        // claims precede its exit count; no memset node in front of the kernel, and a captured graph can be replayed)
        if (lane == 0u) {
            unsigned long long *const exits = reinterpret_cast<unsigned long long *>(a.work_counters + 8u * 16u);
            if (atomicAdd(exits, 1ull) == (unsigned long long)gridDim.x - 1ull) {
#pragma unroll
                for (uint32_t h = 0; h < 8u; h++) atomicExch(reinterpret_cast<unsigned long long *>(a.work_counters + h * 16u), 0ull);
                atomicExch(exits, 0ull);
            }
        }
    } else {
        // ------------------------------ storer waves -----------------------------
        const uint32_t w = wave - 1u;
        for (uint64_t step = 0;; step++) {
            const uint32_t slot = (uint32_t)(step % kRingSlots);
            while (lds_flag_read(lds_offset(&s_full[w][slot])) != (uint32_t)step + 1u) __builtin_amdgcn_s_sleep(1);
            const uint8_t *slab = slabs[w][slot];
            const uint8_t *desc = s_desc[w][step % kDescSlots];
            const uint64_t tag = sgpr64(*reinterpret_cast<const uint64_t *>(desc));
            if (tag == kNoItem - 1ull) break;
            if (tag != kNoItem) {
                const Span sp = desc_get(desc);
                emit_span<HAS_VIDX, NT>(a, p, sp, slab, lane);
            }
            if (lane == 0u) lds_flag_write(lds_offset(&s_done[w][slot]), (uint32_t)step + 1u);
        }
    }
}

}  // namespace

bool gt_span_applicable(const EmitArgs &a)
{
    // S >= 8193: a 16-KiB span touches at most three rows; the work queue needs the ctx's counters
    return a.kept_idx == nullptr && a.line_off == nullptr && a.sample_count >= 2048u && a.work_counters != nullptr &&
           (a.n_variants <= 1 || a.out_stride == 4ull * a.kept_count + 1ull);
}

hipError_t launch_gt_span(const EmitArgs &a, int num_cus, hipStream_t stream)
{
    if (a.n_variants == 0) return hipSuccess;
    SpanParams p;
    p.row_bytes = 4ull * a.kept_count + 1ull;
    p.total_bytes = (uint64_t)a.n_variants * p.row_bytes;
    p.head = (uint32_t)(((uint64_t)(uintptr_t)a.out) & 127ull);
    p.n_chunks = (p.head + p.total_bytes + 15ull) >> 4;
    p.n_items = (p.n_chunks + kSpanChunks - 1ull) / kSpanChunks;
    const char *en = getenv("PGENHIP_WIDE_NT");
    const bool nt = en ? atoi(en) != 0 : true;
    const char *eb = getenv("PGENHIP_WIDE_BLOCKS_PER_CU");
    void (*k)(EmitArgs, SpanParams);
    if (a.variant_idx) k = nt ? gt_span_kernel<true, true> : gt_span_kernel<true, false>;
    else k = nt ? gt_span_kernel<false, true> : gt_span_kernel<false, false>;
    int per_cu = 0;  // exactly the resident blocks: later ones would find the queue empty
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, kThreads, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    if (eb && atoi(eb) > 0) per_cu = atoi(eb);
    const uint64_t cap = (uint64_t)num_cus * (uint64_t)per_cu;
    const uint64_t need = (p.n_items + kNS - 1ull) / kNS;
    const uint32_t grid = (uint32_t)(need < cap ? need : cap);
    hipLaunchKernelGGL(k, dim3(grid), dim3(kThreads), 0, stream, a, p);
    return hipGetLastError();
}

}  // namespace pgenhip
