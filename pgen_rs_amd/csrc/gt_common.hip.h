// gt_common.hip.h — device helpers shared by the GT decode/emit kernels (gfx950 only).
//
// Reference semantics (teoremma/pgen-rs, src/pfile.rs):
//   :172-175  code(s) = (record[s/4] >> ((s%4)*2)) & 0b11         (LSB-first 2-bit hard calls)
//   :177-183  00 -> "0/0", 01 -> "0/1", 10 -> "1/1", 11 -> "./."
//   :186-190  each kept sample emits '\t' + the 3 GT bytes; the row ends with '\n'
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "kernels.h"

namespace pgenhip {

// One genotype = one little-endian dword of text:
//   "\t0/0" = 09 30 2F 30, "\t0/1" = 09 30 2F 31, "\t1/1" = 09 31 2F 31, "\t./." = 09 2E 2F 2E.
// Bytes 0 and 2 are constant; byte 1 is "001."[code], byte 3 is "011."[code].
// v_perm_b32 picks both from two 4-byte tables with one selector (selector values 0-3 = bytes of
// the second operand, 4-7 = bytes of the first, 0x0C = constant 0x00), so a genotype costs
// v_mul_u32_u24 + v_lshl_or_b32 + v_perm_b32 + v_or_b32.
__device__ __forceinline__ uint32_t gt_text(uint32_t code)
{
    constexpr uint32_t kTabA = 0x2E313030u;  // '0','0','1','.'  (allele 1 char by code)
    constexpr uint32_t kTabB = 0x2E313130u;  // '0','1','1','.'  (allele 2 char by code)
    // selector = 0x000C040C | code << 8 | code << 24.  Written as a 24-bit multiply + shift-or: a plain
    // `code * 0x01000100` compiles to v_mul_lo_u32 (quarter rate) + v_add, and these kernels are issue-bound.
    // The multiply is pinned with inline asm: hipcc folds `(umul24(code, 0x010001) << 8)` straight back into
    // v_mul_lo_u32 by 0x01000100.
    uint32_t spread;  // code | code << 16
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(spread) : "v"(code), "v"(0x010001u));
    const uint32_t sel = (spread << 8) | 0x000C040Cu;  // v_lshl_or_b32
    return __builtin_amdgcn_perm(kTabA, kTabB, sel) | 0x002F0009u;
}

// Funnel: bytes [shift, shift+4) of the 8-byte little-endian pair {lo, hi}.
__device__ __forceinline__ uint32_t funnel_bytes(uint32_t lo, uint32_t hi, uint32_t shift)
{
    return __builtin_amdgcn_alignbyte(hi, lo, shift);
}

// single text byte of the GT segment at segment offset p (0 <= p < 4K): used only on the
// per-byte edge path (row heads/tails, prefix seams)
__device__ __forceinline__ uint32_t gt_text_byte(uint32_t code, uint32_t byte_in_gt)
{
    return (gt_text(code) >> (8u * byte_in_gt)) & 0xFFu;
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct alignas(16) u32x4 {
    uint32_t x, y, z, w;
};

// Record of output row r when rows are gathered: byte offsets (uncompressed records of a variable-width .pgen,
// src/pgen.rs's tables) or the kept-variant list (src/pfile.rs:165 with var_idx from the list).
__device__ __forceinline__ const uint8_t *gathered_record(const EmitArgs &a, uint64_t r)
{
    if (a.record_off != nullptr) return a.records + a.record_off[r];
    return a.records + (uint64_t)a.variant_idx[r] * a.record_stride;
}

// ---- all-samples text from a 16-bit record window ---------------------------------------------
// 16 bytes of one row's GT text starting at segment offset q (q >= -15) cover samples
// k0 = floor(q/4) .. k0+4 = 10 bits at bit 2*k0 of the record = record bytes b0 = floor(q/16)
// and b0+1.  load_window fetches those two bytes (out-of-record bytes, needed only for
// don't-care positions, are clamped/zeroed), gt_text16_from_window turns them into the 16 text
// bytes, funnel-shifted by the phase q & 3 so the caller can store to a 16-B-aligned address.
// Bytes whose segment offset q+i lies in [0, 4N) are exact; the others are filler.
//
// LOAD16: one (possibly odd-address) global_load_ushort instead of two byte loads; the address
// is clamped to [0, R-2] (requires R >= 2) so no byte outside the record is touched.
template <bool LOAD16>
__device__ __forceinline__ uint32_t load_window(const uint8_t *__restrict__ rec, int32_t b0, uint32_t last_rec_byte)
{
    if (LOAD16) {
        const int32_t bb = max(0, min(b0, (int32_t)last_rec_byte - 1));
        uint16_t h;
        __builtin_memcpy(&h, rec + bb, 2);
        const int32_t d = b0 - bb;  // -1, 0 or +1 (more only for don't-care windows)
        return d < 0 ? ((uint32_t)h << 8) & 0xFFFFu : (uint32_t)h >> (8u * (uint32_t)min(d, 2));
    } else {
        const uint32_t lo = b0 >= 0 ? (uint32_t)rec[min((uint32_t)b0, last_rec_byte)] : 0u;
        const uint32_t hi = (uint32_t)rec[min((uint32_t)(b0 + 1), last_rec_byte)];
        return lo | (hi << 8);
    }
}

__device__ __forceinline__ u32x4 gt_text16_from_window(uint32_t window, int64_t q)
{
    const uint32_t k0 = (uint32_t)(q >> 2);  // floor(q/4) mod 2^32; only its low 2 bits matter here
    const uint32_t sh = (uint32_t)q & 3u;
    const uint32_t w = window >> ((k0 & 3u) * 2u);
    const uint32_t t0 = gt_text(w & 3u);
    const uint32_t t1 = gt_text((w >> 2) & 3u);
    const uint32_t t2 = gt_text((w >> 4) & 3u);
    const uint32_t t3 = gt_text((w >> 6) & 3u);
    const uint32_t t4 = gt_text((w >> 8) & 3u);
    u32x4 v;
    v.x = funnel_bytes(t0, t1, sh);
    v.y = funnel_bytes(t1, t2, sh);
    v.z = funnel_bytes(t2, t3, sh);
    v.w = funnel_bytes(t3, t4, sh);
    return v;
}

// ---- LDS flag words for wave-to-wave hand-over inside a block -----------------------------------
// Flag words are touched with explicit DS instructions: a volatile C++ access through a generic
// pointer would become flat_load + s_waitcnt vmcnt(0), i.e. exactly the store drain the role-split
// kernels exist to avoid.  The low 32 bits of a generic pointer into LDS are the LDS byte offset.
// (LDS operations of one wave execute in order and both waves live on one CU, so a flag written
// after a wave's data ds_writes is seen after them; compiler ordering is pinned by the asm barriers.)
__device__ __forceinline__ uint32_t lds_offset(const void *p) { return (uint32_t)(uintptr_t)p; }

__device__ __forceinline__ uint32_t lds_flag_read(uint32_t off)
{
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(off) : "memory");
    return v;
}

__device__ __forceinline__ void lds_flag_write(uint32_t off, uint32_t value)
{
    // everything this wave sent to the LDS before (data writes / data reads) has completed first
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" ::"v"(off), "v"(value) : "memory");
}

// ---- shared by the subset kernels (gt_scan.hip, gt_pick.hip) -------------------------------------
typedef uint32_t gt_v4u __attribute__((ext_vector_type(4)));

// The subset kernels' chunk stores: non-temporal like the stream kernel's (the text is not read again on the device; two libraries
// alternated on one box, profiles/r03_logs/subset_nt_stores_ab.log: level to +3 %)
__device__ __forceinline__ void subset_store16(uint8_t *dst, gt_v4u v) { __builtin_nontemporal_store(v, reinterpret_cast<gt_v4u *>(dst)); }

// Byte 0 of row j's GT segment: out + j * out_stride, or — full-line mode (src/pfile.rs:156-192), any kept
// subset — behind the line's prefix at out + line_off[j] + prefix length (the prefixes are copied by
// copy_prefix_rows below, run by the GT kernels' own waves).  j is wave-uniform or per-lane.
__device__ __forceinline__ uint8_t *row_text(const EmitArgs &a, uint64_t j)
{
    if (a.line_off != nullptr) return a.out + a.line_off[j] + (a.prefix_off[j + 1ull] - a.prefix_off[j]);
    return a.out + j * a.out_stride;
}

// Output-driven flush of row bytes [emitted, hi_emit) of one row.  Whole 16-byte-ALIGNED chunks: one lane per chunk
// takes five consecutive kept codes from `code_of`, expands them to text (src/pfile.rs:177-190), funnel-shifts by the
// row's phase and stores 16 B.  The up to 15 bytes before the first and after the last whole chunk (shared with
// whoever writes the neighbouring bytes) go out as ONE byte-store instruction: lanes 0-15 the head bytes, lanes
// 16-31 the tail bytes.  All 64-bit arithmetic is wave-uniform (scalar unit); a lane only adds a 32-bit offset.
// `code_of(x)` yields the 2-bit code at position x = base + segment rank (ring position, or rank for the pick kernel).
// U chunks per lane and loop step: their 5 U table reads, then their 5 U staged-byte reads leave together, so a step costs two LDS
// round trips whatever U is (the compiler does not unroll this loop by itself; with the segment kernel's two waves per SIMD the
// round trips of U = 1 are exposed).
template <uint32_t U = 1, typename CodeFn>
__device__ __forceinline__ void flush_codes(CodeFn code_of, uint32_t base, uint8_t *row_out, uint64_t emitted, uint64_t hi_emit,
                                            uint32_t seg_k0, uint32_t K, uint32_t lane)
{
    uint8_t *const out0 = row_out + emitted;                           // first byte of the flush
    const uint32_t len = (uint32_t)(hi_emit - emitted);                // <= 4 * 16 384 + 1
    const uint32_t mis = (uint32_t)(uintptr_t)out0 & 15u;
    const uint32_t head = min((16u - mis) & 15u, len);                 // bytes before the first whole chunk
    const uint32_t n_chunks = (len - head) >> 4;
    const uint32_t tail_off = head + (n_chunks << 4);
    const uint32_t tail = len - tail_off;                              // bytes after the last whole chunk (< 16)
    const uint32_t e4 = base + (uint32_t)(emitted >> 2) - seg_k0;      // position of the code under byte `emitted`
    const uint32_t em = (uint32_t)emitted & 3u;
    const uint64_t nl64 = 4ull * K - emitted;                          // flush offset of the row's '\n' (row byte 4K)
    const uint32_t nl = nl64 < (uint64_t)len ? (uint32_t)nl64 : 0xFFFFFFFFu;
    for (uint32_t i0 = 0; i0 < n_chunks; i0 += 64u * U) {
        uint32_t offv[U], shv[U], c[U][5];
        bool ok[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t i = i0 + u * 64u + lane;
            ok[u] = i < n_chunks;
            offv[u] = head + ((ok[u] ? i : 0u) << 4);                  // (a lane without a chunk re-reads chunk 0's codes and stores nothing)
            const uint32_t x = em + offv[u];                           // byte offset from the dword boundary under `emitted`
            const uint32_t rel = e4 + (x >> 2);
            shv[u] = x & 3u;
#pragma unroll
            for (uint32_t k = 0; k < 5u; k++) c[u][k] = code_of(rel + k);  // the fifth may be past the flush: then it feeds no byte (sh = 0) or only '\n''s place
        }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t t0 = gt_text(c[u][0]), t1 = gt_text(c[u][1]), t2 = gt_text(c[u][2]), t3 = gt_text(c[u][3]), t4 = gt_text(c[u][4]);
            const uint32_t sh = shv[u];
            gt_v4u v = {funnel_bytes(t0, t1, sh), funnel_bytes(t1, t2, sh), funnel_bytes(t2, t3, sh), funnel_bytes(t3, t4, sh)};
            // the row's '\n' can only be a whole chunk's last byte (hi_emit <= 4K + 1)
            if (offv[u] + 15u == nl) v.w = (v.w & 0x00FFFFFFu) | 0x0A000000u;
            if (ok[u]) subset_store16(out0 + offv[u], v);
        }
    }
    const uint32_t off = lane < 16u ? lane : tail_off + (lane - 16u);
    const bool on = lane < 16u ? lane < head : (lane < 32u && lane - 16u < tail);
    if (on) {
        const uint32_t x = em + off;
        const uint32_t code = code_of(e4 + (x >> 2));
        out0[off] = (uint8_t)(off == nl ? 0x0Au : gt_text_byte(code, x & 3u));
    }
}

// The four VARIABLE bytes of two genotypes' text — (allele 1, allele 2) of code ca, then of code cb — with three instructions
// (v_lshl_or_b32, v_mad_u32_u24, v_perm_b32): selector byte 0 = ca (table A), byte 1 = 4 + ca (table B), bytes 2-3 the same for cb.
__device__ __forceinline__ uint32_t gt_vars2(uint32_t ca, uint32_t cb)
{
    constexpr uint32_t kTabA = 0x2E313030u;  // '0','0','1','.'  (allele 1 char by code)
    constexpr uint32_t kTabB = 0x2E313130u;  // '0','1','1','.'  (allele 2 char by code)
    const uint32_t x = ca | (cb << 16);
    uint32_t sel;   // (ca | cb << 16) * 0x0101 + 0x04000400; pinned: hipcc turns a plain multiply into the quarter-rate v_mul_lo_u32
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(sel) : "v"(x), "v"(0x0101u), "v"(0x04000400u));
    return __builtin_amdgcn_perm(kTabB, kTabA, sel);
}

// The same flush for kernels whose picks come four at a time.  A 16-byte chunk holds the text of ranks rel .. rel+3 and, when the
// flush's phase is not zero, leading bytes of rank rel+4 — which is the FIRST rank of the next chunk, i.e. of the next lane: it comes
// over by a wavefront shift (v_mov_b32_dpp wave_shl:1) instead of a fifth pick, lane 63's from the next group of 64 chunks (one
// extra single pick per step, for the chunk behind the step).  The four table entries of a chunk are consecutive:
// `codes4(c0, g, a, b, c, d)` gets the flush-uniform C0 = rel & 3 as a std::integral_constant (four copies of the loop, chosen per
// flush by a scalar branch) and the aligned group g = rel >> 2, and returns the 2-bit codes of ranks 4g + C0 .. 4g + C0 + 3;
// `code1(rel)` returns one.
// Lane <-> chunk: lane l of group u of a step takes chunk i0 + 64 u + l - lead, where `lead` = the chunks between the 128-byte line
// boundary at or below chunk 0 and chunk 0: every store instruction then covers eight WHOLE lines (a store that starts mid-line
// touches nine, two of them partially).  `part` of `n_parts` cooperating waves takes the steps part, part + n_parts, ...
// Steps whose every lane has a chunk and none the flush's last one — all but the first and the last — take the SHORT path: no
// text dwords at all.  The chunk's variable bytes sit in three registers (gt_vars2: V0 = ranks 0-1, V1 = ranks 2-3, V2 = the next
// lane's V0), and each of its four output dwords is ONE v_perm_b32 of a constant ('\t', '/') and V0 / V1 or their 2-byte
// realignments, with a selector that depends only on the flush's phase: 13 instructions from codes to chunk where text dwords +
// funnel shifts take 21, and no validity selects.
template <uint32_t U, uint32_t C0, typename Codes4Fn, typename Code1Fn>
__device__ __forceinline__ void flush_text4_loop(Codes4Fn codes4, Code1Fn code1, uint8_t *out0, uint32_t head, uint32_t n_chunks, uint32_t rel0,
                                                 uint32_t sh, uint32_t rmax, uint32_t nl, uint32_t lane, uint32_t part, uint32_t n_parts, bool align)
{
    const uint32_t lead = align ? __builtin_amdgcn_readfirstlane(((uint32_t)(uintptr_t)(out0 + head) >> 4) & 7u) : 0u;
    const uint32_t t_last = gt_text(code1(rmax));                       // the rank behind the last whole chunk
    // phase -> selectors of the even / odd output dwords (source bytes 0-3 = the variable-byte register, 4 = '\t', 5 = '/') and the
    // realignment of the odd ones
    const uint32_t sel_even = sh == 0u ? 0x01050004u : sh == 1u ? 0x04010500u : sh == 2u ? 0x02040105u : 0x05020401u;
    const uint32_t sel_odd = sh == 0u ? 0x03050204u : sh == 1u ? 0x04030502u : sel_even;
    const uint32_t realign = sh < 2u ? 0u : 2u;
    constexpr uint32_t kConst = 0x00002F09u;                            // byte 0 = '\t', byte 1 = '/'
    for (uint32_t i0 = part * 64u * U; i0 < n_chunks + lead; i0 += n_parts * 64u * U) {
        if ((i0 != 0u || lead == 0u) && i0 + 64u * U - lead < n_chunks) {
            uint32_t V0[U], V1[U];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                uint32_t c0, c1, c2, c3;
                codes4(std::integral_constant<uint32_t, C0>{}, (rel0 >> 2) + (i0 + u * 64u + lane - lead), c0, c1, c2, c3);
                V0[u] = gt_vars2(c0, c1);
                V1[u] = gt_vars2(c2, c3);
            }
            const uint32_t v_next = gt_vars2(code1(rel0 + 4u * (i0 + 64u * U - lead)), 0u);   // first rank of the chunk behind this step (wave-uniform)
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t from_next_lane = __builtin_amdgcn_update_dpp(0u, V0[u], 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
                const uint32_t from_next_group = u + 1u < U ? (uint32_t)__builtin_amdgcn_readfirstlane(V0[u + 1u < U ? u + 1u : u]) : v_next;
                const uint32_t V2 = lane == 63u ? from_next_group : from_next_lane;
                const uint32_t X1 = __builtin_amdgcn_alignbyte(V1[u], V0[u], realign), X3 = __builtin_amdgcn_alignbyte(V2, V1[u], realign);
                const gt_v4u v = {__builtin_amdgcn_perm(kConst, V0[u], sel_even), __builtin_amdgcn_perm(kConst, X1, sel_odd),
                                  __builtin_amdgcn_perm(kConst, V1[u], sel_even), __builtin_amdgcn_perm(kConst, X3, sel_odd)};
                subset_store16(out0 + head + ((i0 + u * 64u + lane - lead) << 4), v);
            }
            continue;
        }
        uint32_t t[U][5], offv[U];
        bool ok[U], last[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t i = i0 + u * 64u + lane - lead;              // (wraps for the lanes in front of chunk 0: not ok)
            ok[u] = i < n_chunks;
            last[u] = i + 1u == n_chunks;
            const uint32_t ii = ok[u] ? i : 0u;                         // (a lane without a chunk re-reads chunk 0's entries and stores nothing)
            offv[u] = head + (ii << 4);
            uint32_t c0, c1, c2, c3;
            codes4(std::integral_constant<uint32_t, C0>{}, (rel0 >> 2) + ii, c0, c1, c2, c3);
            t[u][0] = gt_text(c0);
            t[u][1] = gt_text(c1);
            t[u][2] = gt_text(c2);
            t[u][3] = gt_text(c3);
        }
        const uint32_t t_next = gt_text(code1(min(rel0 + 4u * (i0 + 64u * U - lead), rmax)));   // first rank of the chunk behind this step (wave-uniform)
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t from_next_lane = __builtin_amdgcn_update_dpp(0u, t[u][0], 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
            const uint32_t from_next_group = u + 1u < U ? (uint32_t)__builtin_amdgcn_readfirstlane(t[u + 1u < U ? u + 1u : u][0]) : t_next;
            t[u][4] = last[u] ? t_last : lane == 63u ? from_next_group : from_next_lane;
            gt_v4u v = {funnel_bytes(t[u][0], t[u][1], sh), funnel_bytes(t[u][1], t[u][2], sh), funnel_bytes(t[u][2], t[u][3], sh), funnel_bytes(t[u][3], t[u][4], sh)};
            if (offv[u] + 15u == nl) v.w = (v.w & 0x00FFFFFFu) | 0x0A000000u;   // the row's '\n' can only be a whole chunk's last byte
            if (ok[u]) subset_store16(out0 + offv[u], v);
        }
    }
}

template <uint32_t U, typename Codes4Fn, typename Code1Fn>
__device__ __forceinline__ void flush_text4(Codes4Fn codes4, Code1Fn code1, uint32_t base, uint8_t *row_out, uint64_t emitted, uint64_t hi_emit,
                                            uint32_t seg_k0, uint32_t K, uint32_t lane, bool align = true, uint32_t part = 0u, uint32_t n_parts = 1u)
{
    uint8_t *const out0 = row_out + emitted;
    const uint32_t len = (uint32_t)(hi_emit - emitted);
    const uint32_t mis = (uint32_t)(uintptr_t)out0 & 15u;
    const uint32_t head = min((16u - mis) & 15u, len);
    const uint32_t n_chunks = (len - head) >> 4;
    const uint32_t tail_off = head + (n_chunks << 4);
    const uint32_t tail = len - tail_off;
    const uint32_t e4 = base + (uint32_t)(emitted >> 2) - seg_k0;
    const uint32_t em = (uint32_t)emitted & 3u;
    const uint64_t nl64 = 4ull * K - emitted;
    const uint32_t nl = nl64 < (uint64_t)len ? (uint32_t)nl64 : 0xFFFFFFFFu;
    if (n_chunks != 0u) {
        const uint32_t x0 = em + head;                                  // byte offset of chunk 0 from the dword boundary under `emitted`
        const uint32_t rel0 = __builtin_amdgcn_readfirstlane(e4 + (x0 >> 2));
        const uint32_t sh = __builtin_amdgcn_readfirstlane(x0 & 3u);
        const uint32_t rmax = __builtin_amdgcn_readfirstlane(e4 + ((em + tail_off) >> 2));
        switch (rel0 & 3u) {
            case 0u: flush_text4_loop<U, 0>(codes4, code1, out0, head, n_chunks, rel0, sh, rmax, nl, lane, part, n_parts, align); break;
            case 1u: flush_text4_loop<U, 1>(codes4, code1, out0, head, n_chunks, rel0, sh, rmax, nl, lane, part, n_parts, align); break;
            case 2u: flush_text4_loop<U, 2>(codes4, code1, out0, head, n_chunks, rel0, sh, rmax, nl, lane, part, n_parts, align); break;
            default: flush_text4_loop<U, 3>(codes4, code1, out0, head, n_chunks, rel0, sh, rmax, nl, lane, part, n_parts, align); break;
        }
    }
    if (part != 0u) return;                                             // the edges: the first of the cooperating waves
    const uint32_t off = lane < 16u ? lane : tail_off + (lane - 16u);
    const bool on = lane < 16u ? lane < head : (lane < 32u && lane - 16u < tail);
    if (on) {
        const uint32_t x = em + off;
        out0[off] = (uint8_t)(off == nl ? 0x0Au : gt_text_byte(code1(e4 + (x >> 2)), x & 3u));
    }
}

// ---- full-line mode: the lines' prefix bytes (pvar fields + "GT", src/pfile.rs:157-161) -------------------------------------
// from the blob to out + line_off[j]: 1-2 % of the output bytes with plink2-made .pvar files, more with 1000 Genomes INFO columns
// (130-250 bytes per line), byte granular at both ends because the neighbouring GT bytes belong to other waves.  The share of wave
// `wave` of `n_waves`: 64 consecutive lines per round — their offsets with ONE coalesced load each (no dependent load chain per
// line) — then eight or four lines per pass (8 / 16 lanes each, by the longest prefix: `shift`): whole destination-aligned dwords
// (the four source bytes gathered), the up to three bytes before and behind them as bytes.
// Round 2 ran this as a kernel of its own behind every GT kernel; at the CLI's launch granularity (13 000 lines per launch) that
// cost 10-17 us beside a GT kernel of 43-52, for 0.4 % of the bytes (profiles/r02_cli_kernels.md).  Round 3: the GT kernels' own
// waves do their share before their first work item (stream kernel: the STORER waves, while the loader's first record loads are in
// flight and they would only wait).  Disjoint bytes, any order.
__device__ __forceinline__ void copy_prefix_rows(const EmitArgs &a, uint32_t shift, uint64_t wave, uint64_t n_waves, uint32_t lane)
{
    const uint32_t group = lane >> shift, sub = lane & ((1u << shift) - 1u), per_pass = 64u >> shift, lanes = 1u << shift;
    const uint64_t V = a.n_variants;
    // lines per round: 64 on big launches; on small ones (the CLI's 13 000 lines over 7 000 waves) one pass' worth per wave, so that
    // the copy is one offset load + one pass deep on many waves instead of eight dependent passes deep on a few
    const uint64_t share = (V + n_waves - 1ull) / n_waves;
    const uint32_t rpr = (uint32_t)min(64ull, max((uint64_t)per_pass, (share + per_pass - 1ull) / per_pass * per_pass));
    for (uint64_t base = wave * rpr; base < V; base += n_waves * rpr) {
        const uint64_t j = min(base + (uint64_t)min(lane, rpr - 1u), V - 1ull);
        const uint64_t p0_l = a.prefix_off[j];
        const uint32_t len_l = (uint32_t)(a.prefix_off[j + 1ull] - p0_l);
        const uint64_t lo_l = a.line_off[j];
        const uint32_t count = (uint32_t)min((uint64_t)rpr, V - base);
        for (uint32_t i = 0; i < count; i += per_pass) {
            const uint32_t li = min(i + group, count - 1u);      // (a short last pass: the spare groups repeat the last line, same bytes)
            const uint64_t p0 = (uint64_t)__shfl((unsigned long long)p0_l, (int)li, 64);
            const uint32_t len = (uint32_t)__shfl((int)len_l, (int)li, 64);
            const uint64_t lo = (uint64_t)__shfl((unsigned long long)lo_l, (int)li, 64);
            const uint8_t *__restrict__ src = a.prefix_blob + p0;
            uint8_t *__restrict__ dst = a.out + lo;
            const uint32_t head = min((uint32_t)(-(int32_t)(uint32_t)(uintptr_t)dst) & 3u, len);   // bytes before the first aligned dword
            const uint32_t nd = (len - head) >> 2, tail = (len - head) & 3u;
            for (uint32_t d = sub; d < nd; d += lanes) {
                uint32_t w;
                __builtin_memcpy(&w, src + head + 4u * d, 4);
                *reinterpret_cast<uint32_t *>(dst + head + 4u * d) = w;
            }
            if (sub < head) dst[sub] = src[sub];
            if (sub >= 4u && sub - 4u < tail) dst[head + 4u * nd + (sub - 4u)] = src[head + 4u * nd + (sub - 4u)];
        }
    }
}


// lanes per line of copy_prefix_rows as log2, by the launch's longest prefix (ms for 1 M lines of 2 504 samples with 30 / 100 / 166-byte
// prefixes: 8 lanes 0.055 / 0.123 / 0.217, 16 lanes 0.076 / 0.117 / 0.162, 32 lanes - / 0.136 / 0.171); 0 = the launch has no prefix bytes
__host__ __device__ inline uint32_t prefix_copy_shift(const EmitArgs &a)
{
    if (a.line_off == nullptr || a.prefix_blob == nullptr) return 0u;
    return a.max_line_bytes - (4ull * a.kept_count + 1ull) <= 48ull ? 3u : 4u;
}

}  // namespace pgenhip
