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

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t ring_code(const uint8_t *ring, uint32_t rel)
{
    return ring[rel & (kRing - 1u)];
}

template <bool HAS_VIDX>
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

    for (uint64_t j = (uint64_t)row_group * kWaves + wave; j < a.n_variants; j += (uint64_t)row_groups * kWaves) {
        const uint64_t src = HAS_VIDX ? (uint64_t)a.variant_idx[j] : j;
        const uint8_t *__restrict__ rec = a.records + src * a.record_stride;
        uint8_t *const row_out = a.out + j * a.out_stride;    // byte 0 of this row's GT segment
        const uint64_t row_addr = (uint64_t)(uintptr_t)row_out;
        uint64_t emitted = 4ull * seg_k0;                      // next row byte this wave must write
        uint32_t produced = 0u;                                // codes in the ring (rank relative to seg_k0)

        for (uint32_t tile = 0; tile < kTilesPerSeg; tile++) {
            const uint32_t w = tile * 64u + lane;
            const uint64_t m = s_mask[w];
            const bool any = __ballot(m != 0ull) != 0ull;
            if (any) {
                // ---- coalesced wide load of the packed 2-bit words: 16 B (64 samples) per lane
                const uint32_t b = seg_byte0 + tile * 1024u + lane * 16u;
                uint64_t lo = 0ull, hi = 0ull;
                if (m != 0ull) {
                    if (b + 16u <= R) {
                        v4u v;
                        __builtin_memcpy(&v, rec + b, 16);
                        lo = (uint64_t)v.x | ((uint64_t)v.y << 32);
                        hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
                    } else {
                        for (uint32_t t = 0; t < 16u && b + t < R; t++) {
                            const uint64_t byte = rec[b + t];
                            if (t < 8u) lo |= byte << (8u * t); else hi |= byte << (8u * (t - 8u));
                        }
                    }
                }
                // ---- compaction: kept codes go to the ring at their rank (src/pfile.rs:171-175)
                uint32_t pos = s_pre[w];
                uint64_t mm = m;
                while (mm != 0ull) {
                    const uint32_t bit = (uint32_t)__builtin_ctzll(mm);
                    mm &= mm - 1ull;
                    const uint64_t half = bit < 32u ? lo : hi;
                    ring[pos & (kRing - 1u)] = (uint8_t)((half >> ((bit & 31u) * 2u)) & 3ull);
                    pos++;
                }
            }
            produced = s_pre[tile * 64u + 64u];  // == prefix of the next tile's first word
            const bool final = tile + 1u == kTilesPerSeg || produced == seg_cnt;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // ---- flush: row bytes [emitted, hi_emit) are now determined
            const uint64_t avail = 4ull * ((uint64_t)seg_k0 + produced);
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
                    if (caddr >= lo_addr && caddr + 16ull <= hi_addr && (uint64_t)q + 16ull <= 4ull * K) {
                        // interior: five consecutive kept codes from the ring
                        const uint32_t rel = (uint32_t)((uint64_t)q >> 2) - seg_k0;
                        const uint32_t sh = (uint32_t)q & 3u;
                        const uint32_t t0 = gt_text(ring_code(ring, rel));
                        const uint32_t t1 = gt_text(ring_code(ring, rel + 1u));
                        const uint32_t t2 = gt_text(ring_code(ring, rel + 2u));
                        const uint32_t t3 = gt_text(ring_code(ring, rel + 3u));
                        const uint32_t t4 = gt_text(ring_code(ring, rel + 4u));
                        v4u v = {funnel_bytes(t0, t1, sh), funnel_bytes(t1, t2, sh), funnel_bytes(t2, t3, sh), funnel_bytes(t3, t4, sh)};
                        *reinterpret_cast<v4u *>(dst) = v;
                    } else {
#pragma unroll
                        for (int bb = 0; bb < 16; bb++) {
                            const uint64_t addr = caddr + (uint64_t)bb;
                            if (addr >= lo_addr && addr < hi_addr) {
                                const uint64_t p = addr - row_addr;
                                uint32_t ch;
                                if (p == 4ull * K)
                                    ch = '\n';
                                else
                                    ch = gt_text_byte(ring_code(ring, (uint32_t)(p >> 2) - seg_k0), (uint32_t)p & 3u);
                                dst[bb] = (uint8_t)ch;
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
    void (*kern)(EmitArgs, ScanArgs, uint32_t, uint32_t) = a.variant_idx ? gt_scan_kernel<true> : gt_scan_kernel<false>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), 0, stream, a, sc, n_seg_eff, (uint32_t)groups);
    return hipGetLastError();
}

}  // namespace pgenhip
