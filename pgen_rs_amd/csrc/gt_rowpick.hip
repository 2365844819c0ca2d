// gt_rowpick.hip — sparse kept subsets on LONG records, one wave per ROW (gfx950 / MI355X).
//
// Replaces /root/reference/src/pfile.rs:165-190 for `--include-sam` runs that keep a few per cent of very many samples
// (BASELINE configs[4]: 1 % of 500 000).  Such a launch is a record READER: 125 000 bytes in, 20 001 bytes of text out per
// variant.  The segment kernels of gt_scan.hip cut a row into 31 pieces handled by 31 blocks, and whatever leaves per piece —
// 656 bytes of text in the single pass, 41 bytes of compact record in round 2's two-pass path — leaves as its own small,
// partial-line store plus (compact form) a stray byte load for the ranks that straddle two pieces.  Measured by ablation on
// the two-pass path (profiles/r03_kernel_sweeps.md): its record loads alone run at 6.27 TB/s, the 41-byte stores cost 6 %,
// the stray byte loads 3 %, the second pass another 18 %.
//
// Here a WAVE owns whole rows:
//   * the block stages the whole kept list once as u16 offsets into 16 384-sample segments (K <= 16 384) plus the number of
//     kept samples before each segment: the rank -> sample table of every segment;
//   * the wave walks its row segment by segment — 4 x 16 B per lane, the next piece's loads in flight while this piece is
//     picked — parks the piece in its 4-KiB LDS stage and picks the segment's kept codes (src/pfile.rs:171-175) into the
//     row's COMPACT record in LDS (four codes to a byte: lane <-> byte, a byte that straddles two segments is completed
//     by the second one);
//   * after the last segment the compact record is a mode-0x02 record of K samples sitting in LDS: the wave writes the row's
//     whole text (:177-190) from it with 16-byte-aligned chunk stores (flush_codes) — 20 KB of whole 128-B lines, the only
//     partial lines are the two row ends — and, in full-line mode, the row's prefix (:157-161) as well.
// One pass, no scratch, no second kernel; HBM traffic per row: the record once + the text once; the kept list once per block.
#include "gt_common.hip.h"
#include "kernels.h"

namespace pgenhip {

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr uint32_t kSegSamples = kScanSegmentSamples;   // 16 384 samples = 4 KiB of record per piece
constexpr uint32_t kTilesPerSeg = 4;
constexpr uint32_t kStageBytes = kSegSamples / 4u;

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t sgpr64_(uint64_t v)   // a wave-uniform value into scalar registers (readfirstlane returns int: widen through uint32_t)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
}

// bytes of dynamic LDS: table (K + 8 u16, 16-B rounded) | kept-before-segment (n_seg + 1 u32, 16-B rounded) | per wave: stage + compact record
__host__ __device__ inline uint32_t table_bytes(uint32_t K) { return (2u * (K + 8u) + 15u) & ~15u; }
__host__ __device__ inline uint32_t rank_bytes(uint32_t n_seg) { return (4u * (n_seg + 1u) + 15u) & ~15u; }
__host__ __device__ inline uint32_t codes_bytes(uint32_t K) { return ((K + 3u) / 4u + 16u + 15u) & ~15u; }   // + slack behind the record for the flush's fifth code

// COMPACT instantiation: instead of text the wave writes the row's compact record itself — ceil(K / 4) bytes at a.out + row *
// a.out_stride, whole 16-byte chunks — the first pass of the two-pass path (capi.hip): an almost pure record reader, the text is
// then written by the all-samples kernels from those records in a second, write-only pass.
template <bool HAS_VIDX, bool COMPACT, uint32_t U, bool FOUR>
__global__ __launch_bounds__(kThreads) void gt_rowpick_kernel(EmitArgs a, ScanArgs sc, uint32_t n_seg)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem[];
    const uint32_t K = a.kept_count;
    uint16_t *const s_idx = reinterpret_cast<uint16_t *>(s_mem);
    uint32_t *const s_rank = reinterpret_cast<uint32_t *>(s_mem + table_bytes(K));
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint8_t *const stage = s_mem + table_bytes(K) + rank_bytes(n_seg) + wave * (kStageBytes + codes_bytes(K));
    uint8_t *const codes = stage + kStageBytes;

    for (uint32_t r = tid; r < K + 8u; r += (uint32_t)kThreads)
        s_idx[r] = r < K ? (uint16_t)(a.kept_idx[r] & (kSegSamples - 1u)) : (uint16_t)0;   // offset inside the sample's segment
    for (uint32_t g = tid; g <= n_seg; g += (uint32_t)kThreads) s_rank[g] = sc.seg_rank[g];
    for (uint32_t i = lane; i < codes_bytes(K); i += 64u) codes[i] = 0;                     // (the slack bytes are read, never stored)
    __syncthreads();

    const uint32_t R = a.record_size;
    const uint64_t wave_step = (uint64_t)gridDim.x * kWaves;
    const uint64_t j0 = (uint64_t)blockIdx.x * kWaves + wave;
    if (j0 >= (uint64_t)a.n_variants) return;
    const uint64_t rows = ((uint64_t)a.n_variants - j0 + wave_step - 1ull) / wave_step;
    const uint64_t pieces = rows * n_seg;

    // loads: tiles before the record's last tile at base + lane offset; the last tile as one window pulled back into the record
    const uint32_t tail_t = (R - 1u) >> 10;
    const uint32_t tail_b = tail_t * 1024u + lane * 16u;
    const uint32_t tail_off = min(tail_b, R - 16u);
    const uint32_t tail_shift = tail_b + 16u <= R ? 0u : min(tail_b - (R - 16u), 16u);
    const bool lines = !COMPACT && a.line_off != nullptr;

    // piece p of this wave = (row j0 + (p / n_seg) * wave_step, segment p % n_seg), walked with two counters
    auto load_piece = [&](uint64_t row, uint32_t seg, v4u(&dst)[kTilesPerSeg]) {
        const uint8_t *__restrict__ rec = HAS_VIDX ? gathered_record(a, row) : a.records + row * a.record_stride;
        const uint32_t tile0 = seg * kTilesPerSeg;
        const uint8_t *__restrict__ sub = rec + (uint64_t)tile0 * 1024u;
#pragma unroll
        for (uint32_t t = 0; t < kTilesPerSeg; t++) {
            const uint8_t *src16 = tile0 + t < tail_t ? sub + lane * 16u + t * 1024u : rec + tail_off;
            __builtin_memcpy(&dst[t], src16, 16);
        }
    };
    auto pick_piece = [&](uint32_t seg, const v4u(&w)[kTilesPerSeg]) {
        const uint32_t tile0 = seg * kTilesPerSeg;
#pragma unroll
        for (uint32_t tile = 0; tile < kTilesPerSeg; tile++) {
            v4u x = w[tile];
            if (tile0 + tile == tail_t && tail_shift != 0u) {
                uint64_t lo = (uint64_t)x.x | ((uint64_t)x.y << 32), hi = (uint64_t)x.z | ((uint64_t)x.w << 32);
                const uint32_t sh8 = tail_shift * 8u;
                if (sh8 >= 128u) { lo = 0ull; hi = 0ull; }
                else if (sh8 >= 64u) { lo = hi >> (sh8 - 64u); hi = 0ull; }
                else { lo = (lo >> sh8) | (hi << (64u - sh8)); hi >>= sh8; }
                x = v4u{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
            }
            if (tile0 + tile <= tail_t) *reinterpret_cast<v4u *>(stage + tile * 1024u + lane * 16u) = x;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ranks [k0, k1) of the kept list live in this segment; compact byte b holds ranks 4b .. 4b+3
        const uint32_t k0 = __builtin_amdgcn_readfirstlane(s_rank[seg]);
        const uint32_t k1 = __builtin_amdgcn_readfirstlane(s_rank[seg + 1u]);
        const uint32_t cb0 = k0 >> 2, cb1 = (k1 + 3u) >> 2;
#pragma clang loop unroll(disable)
        for (uint32_t b = cb0 + lane; b < cb1 && k0 < k1; b += 64u) {
            uint32_t byte = 0u;
#pragma unroll
            for (uint32_t j = 0; j < 4u; j++) {
                const uint32_t r = 4u * b + j;
                const bool mine = r >= k0 && r < k1;
                const uint32_t s16 = s_idx[mine ? r : k0];
                const uint32_t code = ((uint32_t)stage[s16 >> 2] >> ((s16 & 3u) * 2u)) & 3u;   // src/pfile.rs:171-175
                byte |= (mine ? code : 0u) << (2u * j);
            }
            // the byte that straddles this segment and an earlier one was started there (its high codes zero)
            if (b == cb0 && (k0 & 3u) != 0u) byte |= codes[b];
            codes[b] = (uint8_t)byte;
        }
        // the stage is rewritten by the next piece: this piece's reads must have returned first
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // full lines: offsets and prefix bytes travel with the row's record loads — L_*: the row whose first piece was requested last (its
    // line_off / prefix_off entries, in flight); C_*: the row being picked (its prefix bytes were requested with its second piece, when
    // L_* had landed); P_*: the finished row waiting to leave
    uint64_t L_lo = 0ull, L_po = 0ull, L_po1 = 0ull, C_lo = 0ull, C_po = 0ull, C_plen = 0ull, P_lo = 0ull, P_po = 0ull, P_plen = 0ull;
    uint32_t C_pfx = 0u, P_pfx = 0u;
    auto load_offsets = [&](uint64_t row) {      // with piece 0 of `row`
        if (lines) {
            L_lo = a.line_off[row];
            L_po = a.prefix_off[row];
            L_po1 = a.prefix_off[row + 1ull];
        }
    };
    auto load_prefix = [&]() {                    // with piece 1 of the same row: its offsets have landed (they are older than piece 0's data)
        if (lines) {
            C_lo = sgpr64_(L_lo);
            C_po = sgpr64_(L_po);
            C_plen = sgpr64_(L_po1) - C_po;
            C_pfx = lane < C_plen ? (uint32_t)a.prefix_blob[C_po + lane] : 0u;
        }
    };
    // the row's text (and prefix) from its compact record in LDS
    auto emit_row = [&](uint64_t row) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (COMPACT) {
            // the compact record as it lies in LDS: head bytes up to the first 16-byte boundary of the destination, whole chunks, tail bytes
            uint8_t *const crow = a.out + row * a.out_stride;
            const uint32_t len = (K + 3u) >> 2;
            const uint32_t head = min((16u - ((uint32_t)(uintptr_t)crow & 15u)) & 15u, len);
            const uint32_t n_chunks = (len - head) >> 4;
            for (uint32_t c = lane; c < n_chunks; c += 64u) {
                v4u v;
                __builtin_memcpy(&v, codes + head + (c << 4), 16);   // (LDS side unaligned: four dword reads)
                *reinterpret_cast<v4u *>(crow + head + (c << 4)) = v;
            }
            const uint32_t tail_off = head + (n_chunks << 4);
            const uint32_t off = lane < 16u ? lane : tail_off + (lane - 16u);
            if (lane < 16u ? lane < head : (lane < 32u && off < len)) crow[off] = codes[off];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            return;
        }
        uint8_t *row_out;
        if (lines) {
            // the row's offsets came with its first piece and the first 64 bytes of its prefix with its second (P_*): nothing is
            // fetched here unless the prefix is longer than a wave or the record has a single piece
            uint64_t lo = P_lo, p0 = P_po, plen = P_plen;
            if (n_seg < 2u) {
                p0 = a.prefix_off[row];
                plen = a.prefix_off[row + 1ull] - p0;
                lo = a.line_off[row];
            }
            uint8_t *const line = a.out + lo;
            if (n_seg >= 2u && lane < plen) line[lane] = (uint8_t)P_pfx;                                              // :157-161
            for (uint64_t i = (n_seg >= 2u ? 64ull : 0ull) + lane; i < plen; i += 64ull) line[i] = a.prefix_blob[p0 + i];
            row_out = line + plen;
        } else {
            row_out = a.out + row * a.out_stride;
        }
        const uint8_t *cd = codes;
        if (FOUR) {
            // four picks per chunk from ONE or two compact bytes, the fifth text from the next lane (flush_text4, gt_common.hip.h)
            flush_text4<U>(
                [cd](auto c0, uint32_t g, uint32_t &k0, uint32_t &k1, uint32_t &k2, uint32_t &k3) {
                    constexpr uint32_t C0 = decltype(c0)::value;
                    const uint32_t win = C0 == 0u ? (uint32_t)cd[g] : (uint32_t)cd[g] | ((uint32_t)cd[g + 1u] << 8);   // (the byte behind the record is slack)
                    k0 = __builtin_amdgcn_ubfe(win, 2u * C0, 2u);
                    k1 = __builtin_amdgcn_ubfe(win, 2u * C0 + 2u, 2u);
                    k2 = __builtin_amdgcn_ubfe(win, 2u * C0 + 4u, 2u);
                    k3 = __builtin_amdgcn_ubfe(win, 2u * C0 + 6u, 2u);
                },
                [cd](uint32_t r) -> uint32_t { return ((uint32_t)cd[r >> 2] >> ((r & 3u) * 2u)) & 3u; },
                0u, row_out, 0ull, 4ull * K + 1ull, 0u, K, lane, sc.align_stores != 0u);
        } else {
            flush_codes<U>([cd](uint32_t r) { return ((uint32_t)cd[r >> 2] >> ((r & 3u) * 2u)) & 3u; }, 0u, row_out, 0ull, 4ull * K + 1ull, 0u, K, lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto landed = [&](const v4u(&w)[kTilesPerSeg]) {
#pragma unroll
        for (uint32_t t = 0; t < kTilesPerSeg; t++) asm volatile("" ::"v"(w[t].x), "v"(w[t].y), "v"(w[t].z), "v"(w[t].w));
    };

    // two register buffers, the loop unrolled by two: the next piece's loads are in flight while this piece is picked
    v4u b0[kTilesPerSeg], b1[kTilesPerSeg];
    uint64_t row = j0, nrow = j0;     // row of the current piece / of the piece being loaded
    uint32_t seg = 0u, nseg = 0u;
    auto advance = [&](uint64_t &r, uint32_t &s) {
        if (++s == n_seg) { s = 0u; r += wave_step; }
    };
    // A finished row leaves ONE PIECE LATER: right behind the wait for the next piece's loads and in front of the loads after
    // those.  gfx9 counts loads and stores in one in-order vmcnt, so a wave that stored and then waits for loads it issued
    // BEFORE the stores also waits for the stores' acknowledgement (microseconds, once per row); issued in this order nothing
    // is ever younger than the loads a wave waits for.
    bool pending = false;
    uint64_t pending_row = 0ull;
    auto issue_next = [&](uint64_t p, v4u(&dst)[kTilesPerSeg]) {
        advance(nrow, nseg);
        const bool real = p + 1ull < pieces;                                        // (behind the last piece: a harmless re-load)
        if (real && nseg == 0u) load_offsets(nrow);                                   // (in front of the piece's own loads: landed(piece) then covers them)
        if (real && nseg == 1u) load_prefix();
        load_piece(real ? nrow : row, real ? nseg : seg, dst);
    };
    auto finish_piece = [&]() {
        if (seg + 1u == n_seg) {
            pending = true;
            pending_row = row;
            P_lo = C_lo; P_po = C_po; P_plen = C_plen; P_pfx = C_pfx;
        }
        advance(row, seg);
    };
    load_offsets(nrow);
    load_piece(nrow, nseg, b0);
    for (uint64_t p = 0;;) {
        landed(b0);
        if (pending) { emit_row(pending_row); pending = false; }
        issue_next(p, b1);
        pick_piece(seg, b0);
        finish_piece();
        if (++p == pieces) break;
        landed(b1);
        if (pending) { emit_row(pending_row); pending = false; }
        issue_next(p, b0);
        pick_piece(seg, b1);
        finish_piece();
        if (++p == pieces) break;
    }
    if (pending) emit_row(pending_row);
}

}  // namespace

// K <= 16 384 kept samples (the table + a compact record per wave fit the LDS with several blocks per CU), records of at least one
// 16-byte piece, at most 4 096 segments (N <= 64 M), and enough rows that every resident wave owns some
bool gt_rowpick_applicable(const EmitArgs &a, int num_cus)
{
    const uint32_t n_seg = (a.sample_count + kSegSamples - 1u) / kSegSamples;
    return a.kept_idx != nullptr && a.kept_count >= 1u && a.kept_count <= kRowPickMaxKept && a.record_size >= 16u && n_seg >= 1u && n_seg <= 4096u &&
           (uint64_t)a.n_variants >= 8ull * (uint64_t)num_cus * kWaves;
}

namespace {

struct RowPickLaunch {
    void (*kern)(EmitArgs, ScanArgs, uint32_t);
    uint32_t n_seg, lds, max_blocks;   // max_blocks: one resident round
};

bool plan(const EmitArgs &a, const Tuning &t, int num_cus, bool compact, RowPickLaunch &L)
{
    if (a.kept_idx == nullptr || a.kept_count > kRowPickMaxKept) return false;
    L.n_seg = (a.sample_count + kSegSamples - 1u) / kSegSamples;
    if (L.n_seg < 1u || L.n_seg > 4096u || a.record_size < 16u) return false;
    L.lds = table_bytes(a.kept_count) + rank_bytes(L.n_seg) + (uint32_t)kWaves * (kStageBytes + codes_bytes(a.kept_count));
    const bool g = gathered(a);
    if (compact) L.kern = g ? gt_rowpick_kernel<true, true, 1, false> : gt_rowpick_kernel<false, true, 1, false>;
    else if (t.scan_four_picks != 0)
        L.kern = t.flush_unroll == 1 ? (g ? gt_rowpick_kernel<true, false, 1, true> : gt_rowpick_kernel<false, false, 1, true>)
                                     : (g ? gt_rowpick_kernel<true, false, 2, true> : gt_rowpick_kernel<false, false, 2, true>);
    else
        L.kern = t.flush_unroll == 1 ? (g ? gt_rowpick_kernel<true, false, 1, false> : gt_rowpick_kernel<false, false, 1, false>)
                                     : (g ? gt_rowpick_kernel<true, false, 2, false> : gt_rowpick_kernel<false, false, 2, false>);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, L.kern, kThreads, L.lds) != hipSuccess || per_cu < 1) per_cu = 1;
    // two blocks per CU measure best on long records (the loads of 8 waves per CU already saturate the read path: tools/readbench.hip; 3:
    // level, 5: -3 %); on records of barely more than one segment, where a row is mostly text to write, all that fit (N = 16 385 with
    // 10 % kept: 0.51 / 0.57 / 0.56 of roofline at 2 / 3 / 4 per CU; N = 20 000 with 30 %: 0.52 / 0.56 / 0.57)
    const int want = t.rowpick_blocks_per_cu > 0 ? t.rowpick_blocks_per_cu : (!compact && a.sample_count < 24576u ? 4 : 2);
    if (want < per_cu) per_cu = want;
    L.max_blocks = (uint32_t)per_cu * (uint32_t)num_cus;
    return true;
}

}  // namespace

// Rows are dealt to the resident waves round-robin, so a launch takes as long as its busiest wave: a launch of a few rows per
// wave (a chunk of the two-pass path: 13.1 rows per wave) should hold a MULTIPLE of this many rows (14 instead of 13.1: -6 %).
uint32_t gt_rowpick_resident_waves(const EmitArgs &a, const Tuning &t, int num_cus, bool compact)
{
    RowPickLaunch L;
    return plan(a, t, num_cus, compact, L) ? L.max_blocks * (uint32_t)kWaves : 0u;
}

hipError_t launch_gt_rowpick(const EmitArgs &a, const ScanArgs &sc, const Tuning &t, int num_cus, hipStream_t stream, bool compact)
{
    if (a.n_variants == 0) return hipSuccess;
    RowPickLaunch L;
    if (!plan(a, t, num_cus, compact, L)) return hipErrorInvalidValue;
    const uint64_t need = ((uint64_t)a.n_variants + kWaves - 1ull) / kWaves;
    hipLaunchKernelGGL(L.kern, dim3((uint32_t)(need < L.max_blocks ? need : L.max_blocks)), dim3(kThreads), L.lds, stream, a, sc, L.n_seg);
    return hipGetLastError();
}

}  // namespace pgenhip
