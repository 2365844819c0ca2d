// gt_flat.hip — dense all-samples stream kernel for gfx950 (MI355X): the K = N fast path.
//
// Replaces /root/reference/src/pfile.rs:165-190 when every sample is kept and the GT segments
// are packed back to back (out_stride == 4N+1, i.e. exactly the bytes the reference's
// BufWriter emits for the genotype part of consecutive lines).  The whole output of a launch is
// then ONE byte stream of V*(4N+1) bytes, and the kernel is organised around that stream, not
// around rows: lane l of a tile owns the 16-byte-ALIGNED chunk (tile_base + l) of the output
// address space, derives (row, column) of its first byte, builds the 16 bytes in registers and
// issues one global_store_dwordx4.  Every wave store instruction therefore writes 1 KiB of
// contiguous memory covering eight whole 128-B lines, whatever N is — rows of 10 017 bytes
// (N = 2 504) fill the machine exactly like rows of 2 MB (N = 500 000), and no lane idles at row
// ends.  Output is 16x the input, so the kernel is HBM-write-bound; the 2-bit words are read
// with two byte loads per chunk (the 10-bit window of five samples), L1/L2-resident.
//
// A chunk that contains a row's '\n' (one in (4N+1)/16 chunks) merges the tail of row r, the
// newline and the head of row r+1 in registers (two funnel-shifted text vectors + v_bfi), so it
// still leaves as one aligned 16-byte store.  Only the first/last chunk of the stream (when the
// output pointer or length is not 16-B aligned) and degenerate N < 8 use masked byte stores.
//
// (row, col) bookkeeping is incremental: one 64-bit division per block at kernel start, then
// add-with-carry per grid-stride step — no per-lane division for rows >= 8 KiB.
#include "gt_common.hip.h"
#include "kernels.h"

namespace pgenhip {

namespace {

constexpr int kThreads = 256;

struct FlatParams {
    uint64_t row_bytes;    // S = 4N + 1
    uint64_t total_bytes;  // T = V * S
    uint64_t n_chunks;     // aligned 16-B chunks touched by the stream
    uint64_t n_tiles;
    uint64_t step_rows;    // (gridDim * kTileBytes) / S
    uint64_t step_cols;    // (gridDim * kTileBytes) % S
    uint32_t head;         // out address & 15: stream byte 0 sits at chunk 0 byte `head`
};

template <bool HAS_VIDX>
__device__ __forceinline__ const uint8_t *row_record(const EmitArgs &a, uint64_t r)
{
    return HAS_VIDX ? gathered_record(a, r) : a.records + r * a.record_stride;
}

// store one 16-byte chunk; NT = nontemporal hint (the text is written once and never re-read here)
template <bool NT>
__device__ __forceinline__ void store_chunk(u32x4 *dst, const u32x4 &v)
{
    if (NT) {
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        v4u t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<v4u *>(dst));
    } else {
        *dst = v;
    }
}

// U chunks per lane per tile; WAVE_CONTIG: a wave's U stores cover one contiguous U KiB span
// (lane chunk = wave*64*U + u*64 + lane) instead of interleaving the block's four waves.
// HAS_VIDX: rows are gathered through variant_idx.  WRAP: how many row ends one tile can span
// (0: S >= tile bytes -> at most 1; 1: S >= tile bytes / 2 -> at most 2; 2: small rows, divide).
template <int U, bool NT, bool LOAD16, bool WAVE_CONTIG, bool HAS_VIDX, int WRAP>
__global__ __launch_bounds__(kThreads) void gt_flat_kernel(EmitArgs a, FlatParams p)
{
    constexpr uint32_t kTileChunks = kThreads * U;
    constexpr uint32_t kTileBytes = kTileChunks * 16u;
    const uint32_t tid = threadIdx.x;
    const uint64_t S = p.row_bytes;
    const uint64_t gt_bytes = S - 1ull;
    const uint32_t last_rec_byte = a.record_size ? a.record_size - 1u : 0u;
    uint8_t *const chunk0 = a.out - p.head;  // 16-B aligned; pointer arithmetic keeps the global address space

    uint64_t tile = blockIdx.x;
    if (tile >= p.n_tiles) return;

    // (row, col) of the first byte of this block's first tile; col < 0 only for tile 0 when the
    // output pointer is not 16-B aligned (the chunk then starts `head` bytes before the stream)
    uint64_t row;
    int64_t col;
    {
        const int64_t o0 = (int64_t)(tile * kTileBytes) - (int64_t)p.head;
        if (o0 < 0) {
            row = 0;
            col = o0;
        } else {
            row = (uint64_t)o0 / S;
            col = (int64_t)((uint64_t)o0 - row * S);
        }
    }

    for (; tile < p.n_tiles; tile += gridDim.x) {
        // ---- phase A (branch-free): locate the U chunks of this lane and issue their window loads
        int64_t cs[U];
        uint64_t rs[U];
        uint32_t ws[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t lane_chunk = WAVE_CONTIG ? ((tid >> 6) * 64u * U + (uint32_t)u * 64u + (tid & 63u))
                                                    : ((uint32_t)u * kThreads + tid);
            int64_t c = col + (int64_t)(lane_chunk * 16u);
            uint64_t r = row;
            if (WRAP == 2) {
                // small rows: c < S + tile bytes < 2^31; branch-free 32-bit divide
                const uint32_t cu = c > 0 ? (uint32_t)c : 0u;
                const uint32_t qd = cu / (uint32_t)S;
                c -= (int64_t)((uint64_t)qd * S);
                r += qd;
            } else {
                const bool w1 = c >= (int64_t)S;
                c -= w1 ? (int64_t)S : 0;
                r += w1 ? 1u : 0u;
                if (WRAP == 1) {
                    const bool w2 = c >= (int64_t)S;
                    c -= w2 ? (int64_t)S : 0;
                    r += w2 ? 1u : 0u;
                }
            }
            cs[u] = c;
            rs[u] = r;
            // rows past the end (lanes beyond the stream) are clamped so the load stays in bounds
            const uint64_t r_safe = r < a.n_variants ? r : (uint64_t)a.n_variants - 1ull;
            ws[u] = load_window<LOAD16>(row_record<HAS_VIDX>(a, r_safe), (int32_t)(c >> 4), last_rec_byte);
        }
        // ---- phase B: build and store
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t lane_chunk = WAVE_CONTIG ? ((tid >> 6) * 64u * U + (uint32_t)u * 64u + (tid & 63u))
                                                    : ((uint32_t)u * kThreads + tid);
            const uint64_t g = tile * kTileChunks + lane_chunk;
            if (g >= p.n_chunks) continue;
            const int64_t c = cs[u];
            const uint64_t r = rs[u];
            u32x4 *dst = reinterpret_cast<u32x4 *>(chunk0 + g * 16ull);
            const int64_t o = (int64_t)(g * 16ull) - (int64_t)p.head;  // stream offset of the chunk

            if (c >= 0 && (uint64_t)c + 16ull <= gt_bytes) {
                // interior of one row's GT text
                store_chunk<NT>(dst, gt_text16_from_window(ws[u], c));
            } else if (c >= 0 && o >= 0 && (uint64_t)o + 16ull <= p.total_bytes) {
                // the chunk holds row r's '\n' at byte nl, row r's text before, row r+1's after
                const uint32_t nl = (uint32_t)(gt_bytes - (uint64_t)c);  // 0..15
                u32x4 x = gt_text16_from_window(ws[u], c);
                u32x4 y = {0u, 0u, 0u, 0u};
                if (nl < 15u) {
                    const int64_t qy = c - (int64_t)S;  // -15..-1: row r+1 starts inside this chunk
                    y = gt_text16_from_window(load_window<false>(row_record<HAS_VIDX>(a, r + 1ull), (int32_t)(qy >> 4), last_rec_byte), qy);
                }
                // byte i < nl from x, byte nl = '\n', byte i > nl from y
                uint32_t xs[4] = {x.x, x.y, x.z, x.w};
                uint32_t ys[4] = {y.x, y.y, y.z, y.w};
                uint32_t os[4];
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const int32_t nb = (int32_t)nl - 4 * m;  // bytes of dword m that come from x
                    uint32_t mask = nb >= 4 ? 0xFFFFFFFFu : (nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u));
                    uint32_t v = (xs[m] & mask) | (ys[m] & ~mask);
                    if (nb >= 0 && nb < 4) v = (v & ~(0xFFu << (8 * nb))) | (0x0Au << (8 * nb));
                    os[m] = v;
                }
                u32x4 v = {os[0], os[1], os[2], os[3]};
                store_chunk<NT>(dst, v);
            } else {
                // first/last chunk of the stream (unaligned pointer or length): one byte at a time, masked
                uint32_t d[4] = {0u, 0u, 0u, 0u};
                uint32_t valid = 0u;
                uint64_t rr = r;
                int64_t cc = c;
#pragma unroll
                for (int b = 0; b < 16; b++) {
                    const int64_t ob = o + b;
                    if (ob >= 0 && (uint64_t)ob < p.total_bytes) {
                        while (cc >= (int64_t)S) {
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
                        d[b >> 2] |= ch << (8 * (b & 3));
                        valid |= 1u << b;
                    }
                    cc++;
                }
                if (valid == 0xFFFFu) {
                    u32x4 v = {d[0], d[1], d[2], d[3]};
                    *dst = v;
                } else {
                    uint8_t *bp = reinterpret_cast<uint8_t *>(dst);
#pragma unroll
                    for (int b = 0; b < 16; b++) {
                        if (valid & (1u << b)) bp[b] = (uint8_t)(d[b >> 2] >> (8 * (b & 3)));
                    }
                }
            }
        }
        // advance (row, col) by one grid stride of tiles
        col += (int64_t)p.step_cols;
        row += p.step_rows;
        if (col >= (int64_t)S) {
            col -= (int64_t)S;
            row++;
        }
        if (col < 0 && row > 0) {
            col += (int64_t)S;
            row--;
        }
    }
}

}  // namespace

bool gt_flat_applicable(const EmitArgs &a)
{
    // N >= 8: a row is >= 33 bytes, so a 16-byte chunk meets at most one '\n', and R >= 2
    return a.kept_idx == nullptr && a.line_off == nullptr && a.sample_count >= 8u &&
           (a.n_variants <= 1 || a.out_stride == 4ull * a.kept_count + 1ull);
}

hipError_t launch_gt_flat(const EmitArgs &a, const Tuning &t, int num_cus, hipStream_t stream)
{
    if (a.n_variants == 0) return hipSuccess;
    FlatParams p;
    p.row_bytes = 4ull * a.kept_count + 1ull;
    p.total_bytes = (uint64_t)a.n_variants * p.row_bytes;
    p.head = (uint32_t)(((uint64_t)(uintptr_t)a.out) & 15ull);
    p.n_chunks = (p.head + p.total_bytes + 15ull) / 16ull;
    // U = 4 chunks per lane, nontemporal stores, byte window loads: the best of the 13 builds of the round-1 sweep
    // (profiles/r01_kernel_sweeps.md; plain stores and U = 8 + 16-bit loads + wave-contiguous spans were the runners-up)
    constexpr uint32_t kU = 4;
    static void (*const kern[2][3])(EmitArgs, FlatParams) = {
        {gt_flat_kernel<kU, true, false, false, false, 0>, gt_flat_kernel<kU, true, false, false, false, 1>, gt_flat_kernel<kU, true, false, false, false, 2>},
        {gt_flat_kernel<kU, true, false, false, true, 0>, gt_flat_kernel<kU, true, false, false, true, 1>, gt_flat_kernel<kU, true, false, false, true, 2>}};
    const uint32_t tile_chunks = kThreads * kU;
    const uint32_t tile_bytes = tile_chunks * 16u;
    p.n_tiles = (p.n_chunks + tile_chunks - 1ull) / tile_chunks;
    const uint64_t max_grid = (uint64_t)num_cus * (uint64_t)(t.flat_blocks_per_cu > 0 ? t.flat_blocks_per_cu : 64);
    const uint32_t grid = (uint32_t)(p.n_tiles < max_grid ? p.n_tiles : max_grid);
    const uint64_t step = (uint64_t)grid * tile_bytes;
    p.step_rows = step / p.row_bytes;
    p.step_cols = step % p.row_bytes;
    const int wrap = p.row_bytes >= tile_bytes ? 0 : (p.row_bytes >= tile_bytes / 2u ? 1 : 2);
    hipLaunchKernelGGL(kern[gathered(a) ? 1 : 0][wrap], dim3(grid), dim3(kThreads), 0, stream, a, p);
    return hipGetLastError();
}

}  // namespace pgenhip
