// gt_common.hip.h — device helpers shared by the GT decode/emit kernels (gfx950 only).
//
// Reference semantics (teoremma/pgen-rs, src/pfile.rs):
//   :172-175  code(s) = (record[s/4] >> ((s%4)*2)) & 0b11         (LSB-first 2-bit hard calls)
//   :177-183  00 -> "0/0", 01 -> "0/1", 10 -> "1/1", 11 -> "./."
//   :186-190  each kept sample emits '\t' + the 3 GT bytes; the row ends with '\n'
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
template <typename CodeFn>
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
    for (uint32_t i = lane; i < n_chunks; i += 64u) {
        const uint32_t off = head + (i << 4);
        const uint32_t x = em + off;                                   // byte offset from the dword boundary under `emitted`
        const uint32_t rel = e4 + (x >> 2);
        const uint32_t sh = x & 3u;
        const uint32_t t0 = gt_text(code_of(rel));
        const uint32_t t1 = gt_text(code_of(rel + 1u));
        const uint32_t t2 = gt_text(code_of(rel + 2u));
        const uint32_t t3 = gt_text(code_of(rel + 3u));
        const uint32_t t4 = gt_text(code_of(rel + 4u));  // may be past the flush: then it feeds no byte (sh = 0) or only '\n''s place
        gt_v4u v = {funnel_bytes(t0, t1, sh), funnel_bytes(t1, t2, sh), funnel_bytes(t2, t3, sh), funnel_bytes(t3, t4, sh)};
        // the row's '\n' can only be a whole chunk's last byte (hi_emit <= 4K + 1)
        if (off + 15u == nl) v.w = (v.w & 0x00FFFFFFu) | 0x0A000000u;
        *reinterpret_cast<gt_v4u *>(out0 + off) = v;
    }
    const uint32_t off = lane < 16u ? lane : tail_off + (lane - 16u);
    const bool on = lane < 16u ? lane < head : (lane < 32u && lane - 16u < tail);
    if (on) {
        const uint32_t x = em + off;
        const uint32_t code = code_of(e4 + (x >> 2));
        out0[off] = (uint8_t)(off == nl ? 0x0Au : gt_text_byte(code, x & 3u));
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
