// gt_scan.hip — kept-subset kernel: input-driven scan + LDS compaction (gfx950 / MI355X).
//
// Replaces /root/reference/src/pfile.rs:165-190 when a sample filter is active
// (`--include-sam`, kept list from filter_metadata :312-335).  The keep mask is the same for
// every variant, so it is turned once per context into an N-bit bitmap plus per-segment ranks;
// the kernel then never touches the kept-index list:
//
//   * a block owns one SEGMENT of 16 384 samples (4 KiB of every record) and stages that
//     segment's 256 keep words into LDS once, together with their exclusive popcount prefix
//     (computed in-kernel: per-lane __popcll + a wave-level shuffle scan);
//   * each WAVE of the block then walks rows: per 1-KiB tile it issues one coalesced 16-B-per-
//     lane load of the packed 2-bit words (64 samples per lane), takes its 64-bit keep word and
//     rank from LDS, and drops the kept codes into its private LDS ring at their final rank
//     (ctz loop over the set bits — lanes with an empty word skip, a tile whose ballot of
//     non-empty words is zero is skipped whole);
//   * output-driven flush: lanes own 16-byte-ALIGNED chunks of the row's output bytes, read five
//     consecutive codes from the ring, expand to text, funnel-shift by the row's phase and store
//     16 B — whole-line coalesced stores regardless of how irregular the mask is.  Segment and
//     row edges (partial chunks, '\n') use masked byte stores.
//
// HBM traffic per row: the record once (R bytes, wide loads) + 4K+1 bytes of text; the bitmap
// (N/8 bytes) is read once per block, not once per row.
#include "gt_common.hip.h"
#include "kernels.h"

namespace pgenhip {

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr uint32_t kSegSamples = kScanSegmentSamples;   // 16 384 samples per segment
constexpr uint32_t kSegWords = kSegSamples / 64;        // 256 keep words
constexpr uint32_t kTileSamples = 4096;                 // 64 lanes x 64 samples = 1 KiB of record
constexpr uint32_t kTilesPerSeg = kSegSamples / kTileSamples;
constexpr uint32_t kRing = 8192;                        // code ring per wave (bytes, power of two)
constexpr uint32_t kFlushCodes = 2048;                  // pending codes that trigger a mid-segment flush (ring holds 2048 + 4096 + carry)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t ring_code(const uint8_t *ring, uint32_t rel)
{
    return ring[rel & (kRing - 1u)];
}

// DENSE: most samples are kept (host decides from K/N): compact one record byte (4 samples) per step
// instead of one kept sample per step.  A separate instantiation keeps the sparse build's registers low.
template <bool HAS_VIDX, bool DENSE>
__global__ __launch_bounds__(kThreads) void gt_scan_kernel(EmitArgs a, ScanArgs sc, uint32_t n_seg, uint32_t row_groups)
{
    __shared__ uint64_t s_mask[kSegWords];
    __shared__ uint32_t s_pre[kSegWords + 1];
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[kWaves][kRing];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t seg = blockIdx.x % n_seg;
    const uint32_t row_group = blockIdx.x / n_seg;

    // ---- stage this segment's keep words and their exclusive popcount prefix (once per block)
    s_mask[tid] = sc.keep_words[(uint64_t)seg * kSegWords + tid];
    __syncthreads();
    if (wave == 0u) {
        // lane handles words 4*lane .. 4*lane+3; wave scan over the lane totals
        uint32_t c0 = __popcll(s_mask[4u * lane]), c1 = __popcll(s_mask[4u * lane + 1u]);
        uint32_t c2 = __popcll(s_mask[4u * lane + 2u]), c3 = __popcll(s_mask[4u * lane + 3u]);
        uint32_t tot = c0 + c1 + c2 + c3;
        uint32_t incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t up = __shfl_up(incl, d, 64);
            if ((int)lane >= d) incl += up;
        }
        uint32_t excl = incl - tot;
        s_pre[4u * lane] = excl;
        s_pre[4u * lane + 1u] = excl + c0;
        s_pre[4u * lane + 2u] = excl + c0 + c1;
        s_pre[4u * lane + 3u] = excl + c0 + c1 + c2;
        if (lane == 63u) s_pre[kSegWords] = incl;
    }
    __syncthreads();

    uint8_t *const ring = s_ring[wave];
    const uint32_t K = a.kept_count;
    const uint32_t seg_k0 = sc.seg_rank[seg];                 // kept samples before this segment
    const uint32_t seg_cnt = s_pre[kSegWords];                // kept samples inside it
    const bool last_seg = seg + 1u == n_seg;
    const uint32_t seg_byte0 = seg * (kSegSamples / 4u);      // first record byte of the segment
    const uint32_t R = a.record_size;
    if (seg_cnt == 0u && !last_seg) return;                   // nothing of this segment is kept

    // the keep word and rank of this lane's 64 samples in each tile do not depend on the row:
    // take them out of LDS once
    uint64_t m[kTilesPerSeg];
    uint32_t pre[kTilesPerSeg], tile_end[kTilesPerSeg];
    uint32_t live_tiles = 0u;  // tiles with at least one kept sample (wave-uniform bit set)
#pragma unroll
    for (uint32_t t = 0; t < kTilesPerSeg; t++) {
        m[t] = s_mask[t * 64u + lane];
        pre[t] = s_pre[t * 64u + lane];
        tile_end[t] = s_pre[t * 64u + 64u];
        if (__ballot(m[t] != 0ull) != 0ull) live_tiles |= 1u << t;
    }

    const uint64_t row_step = (uint64_t)row_groups * kWaves;
    uint64_t j = (uint64_t)row_group * kWaves + wave;
    if (j >= a.n_variants) return;

    // ---- coalesced wide loads of the packed 2-bit words: 16 B (64 samples) per lane and tile,
    // all tiles of the segment in flight at once, and the NEXT row's requested before this row is
    // processed (register double buffering)
    v4u cur[kTilesPerSeg], nxt[kTilesPerSeg];
    // Branch-free: every lane always loads 16 bytes from inside the record (R >= 16 is a launch
    // precondition).  A window that would run past the record end (last lane of the last tile) is
    // pulled back to R-16 and the bytes are shifted into place at use time (`tail_shift`), so no
    // conditional byte loop — and with it no early s_waitcnt — sits between the loads.
    uint32_t tail_shift[kTilesPerSeg];
#pragma unroll
    for (uint32_t t = 0; t < kTilesPerSeg; t++) {
        const uint32_t b = seg_byte0 + t * 1024u + lane * 16u;
        tail_shift[t] = b + 16u <= R ? 0u : min(b - (R - 16u), 16u);
    }
    auto load_row = [&](uint64_t row, v4u(&dst)[kTilesPerSeg]) {
        const uint64_t src = HAS_VIDX ? (uint64_t)a.variant_idx[row] : row;
        const uint8_t *__restrict__ rec = a.records + src * a.record_stride;
#pragma unroll
        for (uint32_t t = 0; t < kTilesPerSeg; t++) {
            const uint32_t b = min(seg_byte0 + t * 1024u + lane * 16u, R - 16u);
            __builtin_memcpy(&dst[t], rec + b, 16);
        }
    };
    load_row(j, cur);

    for (;;) {
        const uint64_t j_next = j + row_step;
        const bool more = j_next < a.n_variants;
        // gfx9 has ONE in-order vmcnt: make this row's words land before the next row's are
        // requested (they were issued a whole row ago, so this wait is short), otherwise the first
        // use of `cur` below would also wait for the loads issued here
#pragma unroll
        for (uint32_t t = 0; t < kTilesPerSeg; t++) asm volatile("" ::"v"(cur[t].x), "v"(cur[t].y), "v"(cur[t].z), "v"(cur[t].w));
        if (more) load_row(j_next, nxt);

        uint8_t *const row_out = a.out + j * a.out_stride;    // byte 0 of this row's GT segment
        const uint64_t row_addr = (uint64_t)(uintptr_t)row_out;
        uint64_t emitted = 4ull * seg_k0;                      // next row byte this wave must write
        uint32_t produced = 0u;                                // codes in the ring (rank relative to seg_k0)

#pragma unroll
        for (uint32_t tile = 0; tile < kTilesPerSeg; tile++) {
            if (live_tiles & (1u << tile)) {
                // ---- compaction: kept codes go to the ring at their rank (src/pfile.rs:171-175)
                uint64_t lo = (uint64_t)cur[tile].x | ((uint64_t)cur[tile].y << 32);
                uint64_t hi = (uint64_t)cur[tile].z | ((uint64_t)cur[tile].w << 32);
                if (tail_shift[tile] != 0u) {
                    // record tail: the window was pulled back by tail_shift bytes; drop them
                    const uint32_t sh8 = tail_shift[tile] * 8u;
                    if (sh8 >= 128u) { lo = 0ull; hi = 0ull; }
                    else if (sh8 >= 64u) { lo = hi >> (sh8 - 64u); hi = 0ull; }
                    else { lo = (lo >> sh8) | (hi << (64u - sh8)); hi >>= sh8; }
                }
                uint32_t pos = pre[tile];
                uint64_t mm = m[tile];
                if (DENSE) {
                    // dense masks: one record byte (4 samples) at a time; a fully kept byte becomes
                    // one 4-byte ring write (codes spread to one per byte), partial bytes go bit-wise
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        const uint32_t nib = (uint32_t)(mm >> (4 * q)) & 0xFu;
                        if (nib == 0u) continue;
                        const uint32_t x = (uint32_t)((q < 8 ? lo : hi) >> (8 * (q & 7))) & 0xFFu;
                        if (nib == 0xFu) {
                            const uint32_t d = (x & 3u) | ((x & 0xCu) << 6) | ((x & 0x30u) << 12) | ((x & 0xC0u) << 18);
                            const uint32_t idx = pos & (kRing - 1u);
                            if (idx <= kRing - 4u) {
                                __builtin_memcpy(ring + idx, &d, 4);
                            } else {
                                ring[idx] = (uint8_t)d;
                                ring[(idx + 1u) & (kRing - 1u)] = (uint8_t)(d >> 8);
                                ring[(idx + 2u) & (kRing - 1u)] = (uint8_t)(d >> 16);
                                ring[(idx + 3u) & (kRing - 1u)] = (uint8_t)(d >> 24);
                            }
                            pos += 4u;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                if (nib & (1u << e)) {
                                    ring[pos & (kRing - 1u)] = (uint8_t)((x >> (2 * e)) & 3u);
                                    pos++;
                                }
                            }
                        }
                    }
                } else {
                    while (mm != 0ull) {
                        const uint32_t bit = (uint32_t)__builtin_ctzll(mm);
                        mm &= mm - 1ull;
                        const uint64_t half = bit < 32u ? lo : hi;
                        ring[pos & (kRing - 1u)] = (uint8_t)((half >> ((bit & 31u) * 2u)) & 3ull);
                        pos++;
                    }
                }
            }
            produced = tile_end[tile];
            const bool final = tile + 1u == kTilesPerSeg || produced == seg_cnt;
            // flush when the segment is done, or when enough text is pending to fill whole stores
            const uint64_t avail = 4ull * ((uint64_t)seg_k0 + produced);
            if (!final && avail - emitted < 4ull * kFlushCodes) continue;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // ---- flush: row bytes [emitted, hi_emit) are now determined
            uint64_t hi_emit;
            if (final)
                hi_emit = avail + (last_seg ? 1ull : 0ull);             // '\n' closes the row (:190)
            else
                hi_emit = ((row_addr + avail) & ~15ull) - row_addr;     // keep the partial chunk for later
            if ((int64_t)hi_emit > (int64_t)emitted) {
                const uint64_t lo_addr = row_addr + emitted, hi_addr = row_addr + hi_emit;
                const uint64_t c_first = lo_addr >> 4, c_last = (hi_addr - 1ull) >> 4;
                const uint32_t n_chunks = (uint32_t)(c_last - c_first) + 1u;
                for (uint32_t i = lane; i < n_chunks; i += 64u) {
                    const uint64_t caddr = (c_first + i) << 4;
                    const int64_t q = (int64_t)(caddr - row_addr);      // row byte of the chunk start
                    uint8_t *dst = row_out + q;
                    // five consecutive kept codes from the ring (ranks outside this flush read stale
                    // ring bytes: they only feed bytes that are masked out below)
                    const int64_t k0s = q >> 2;                           // floor; may be < seg_k0 in the first chunk
                    const uint32_t rel = (uint32_t)((int64_t)k0s - (int64_t)seg_k0);
                    const uint32_t sh = (uint32_t)q & 3u;
                    const uint32_t t0 = gt_text(ring_code(ring, rel));
                    const uint32_t t1 = gt_text(ring_code(ring, rel + 1u));
                    const uint32_t t2 = gt_text(ring_code(ring, rel + 2u));
                    const uint32_t t3 = gt_text(ring_code(ring, rel + 3u));
                    const uint32_t t4 = gt_text(ring_code(ring, rel + 4u));
                    uint32_t d[4] = {funnel_bytes(t0, t1, sh), funnel_bytes(t1, t2, sh), funnel_bytes(t2, t3, sh), funnel_bytes(t3, t4, sh)};
                    // the row's '\n' (row byte 4K) if it falls into this chunk
                    const int64_t nl = (int64_t)(4ull * K) - q;
                    if (nl >= 0 && nl < 16) {
#pragma unroll
                        for (int mth = 0; mth < 4; mth++) {
                            if ((nl >> 2) == mth) d[mth] = (d[mth] & ~(0xFFu << (8 * (nl & 3)))) | (0x0Au << (8 * (nl & 3)));
                        }
                    }
                    // valid bytes of this chunk: [vb, ve) within 0..16
                    const uint32_t vb = caddr >= lo_addr ? 0u : (uint32_t)(lo_addr - caddr);
                    const uint32_t ve = caddr + 16ull <= hi_addr ? 16u : (uint32_t)(hi_addr - caddr);
                    if (vb == 0u && ve == 16u) {
                        v4u v = {d[0], d[1], d[2], d[3]};
                        *reinterpret_cast<v4u *>(dst) = v;
                    } else {
                        // segment / row edge: whole dwords where possible, single bytes otherwise
#pragma unroll
                        for (int mth = 0; mth < 4; mth++) {
                            const uint32_t b0 = 4u * mth;
                            if (vb <= b0 && ve >= b0 + 4u) {
                                *reinterpret_cast<uint32_t *>(dst + b0) = d[mth];
                            } else {
#pragma unroll
                                for (int bb = 0; bb < 4; bb++) {
                                    if (b0 + bb >= vb && b0 + bb < ve) dst[b0 + bb] = (uint8_t)(d[mth] >> (8 * bb));
                                }
                            }
                        }
                    }
                }
                emitted = hi_emit;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (final) break;
        }
        if (!more) break;
#pragma unroll
        for (uint32_t t = 0; t < kTilesPerSeg; t++) cur[t] = nxt[t];
        j = j_next;
    }
}

}  // namespace

hipError_t launch_gt_scan(const EmitArgs &a, const ScanArgs &sc, int num_cus, hipStream_t stream)
{
    if (a.n_variants == 0) return hipSuccess;
    const uint32_t n_seg = (a.sample_count + kSegSamples - 1u) / kSegSamples;
    const uint32_t n_seg_eff = n_seg ? n_seg : 1u;
    const uint64_t groups_needed = ((uint64_t)a.n_variants + kWaves - 1ull) / kWaves;
    uint64_t groups = ((uint64_t)num_cus * 4ull + n_seg_eff - 1ull) / n_seg_eff;
    if (groups < 1ull) groups = 1ull;
    if (groups > groups_needed) groups = groups_needed;
    const uint32_t grid = (uint32_t)(groups * n_seg_eff);
    const bool dense = (uint64_t)a.kept_count * 4ull > (uint64_t)a.sample_count * 3ull;  // > 75 % kept
    void (*kern)(EmitArgs, ScanArgs, uint32_t, uint32_t);
    if (a.variant_idx)
        kern = dense ? gt_scan_kernel<true, true> : gt_scan_kernel<true, false>;
    else
        kern = dense ? gt_scan_kernel<false, true> : gt_scan_kernel<false, false>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), 0, stream, a, sc, n_seg_eff, (uint32_t)groups);
    return hipGetLastError();
}

}  // namespace pgenhip
