/*
 * pgen_hip.h — C ABI of libpgen_hip.so, the MI355X (gfx950) engine for the
 * one hot path of teoremma/pgen-rs: fixed-width .pgen (storage mode 0x02)
 * variant records -> 2-bit hard-call unpack -> kept-sample select -> VCF GT
 * text.  This is the drop-in boundary: plain pointers and sizes, no C++ or
 * torch types.  The reference has no FFI of its own (it is one private Rust
 * function), so each entry point cites the reference lines it replaces;
 * INTEGRATION.md shows the Rust `extern "C"` block a maintainer would add.
 *
 * Reference seam: /root/reference/src/pfile.rs:156-192 (Pfile::output_vcf hot
 * loop), :196-200 (variant_record_size), :38-76 (header), :165 (record offset).
 *
 * Error convention: every call returns int, 0 = PGENHIP_OK, negative =
 * pgenhip_status.  Nothing throws or aborts across the boundary (the
 * reference panics at src/pfile.rs:47,53,69,169,170,173; a host maps a
 * negative status to a non-zero exit + stderr).  There is NO CPU fallback:
 * without a usable HIP device pgenhip_create fails with
 * PGENHIP_ERR_NO_DEVICE / PGENHIP_ERR_HIP.
 *
 * Threading: a ctx is bound to one device and is not thread-safe; use one
 * ctx per device per host thread.  Distinct ctxs are independent.
 *
 * Streams: launches are asynchronous on the ctx stream (its own, or the one
 * given to pgenhip_set_stream).  A ctx may be moved between streams at any
 * time and launches queued on different streams may overlap: every launch
 * takes its own block of work-queue counters from a ring of
 * PGENHIP_LAUNCHES_IN_FLIGHT, so at most that many launches of ONE ctx may be
 * in flight at once (a host that keeps more queued must pgenhip_wait in
 * between).  The scratch of the two-pass path (sparse keeps on long
 * records) is sliced the same way: each launch in flight has its own slice.
 * A launch captured into a HIP graph keeps the ring slot it was captured
 * with, so a REPLAY of that graph must not overlap other launches of the same
 * ctx (they come round to its slot every PGENHIP_LAUNCHES_IN_FLIGHT
 * launches): replay on the stream the ctx launches on, or give the graph a
 * ctx of its own.  The kernels leave their counter block zeroed; after a launch
 * that FAILED (any negative status from a launch call) the ctx re-zeroes the
 * whole ring in stream order before its next launch.  A kernel that faults
 * on the device takes the process down like any HIP fault; nothing is
 * recoverable in that case.
 *
 * Nothing in this library reads the process environment: launch-shape knobs
 * are per ctx (pgenhip_tune) and default to the measured best.
 */
#ifndef PGEN_HIP_H
#define PGEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGENHIP_ABI_VERSION 2u
#define PGENHIP_LAUNCHES_IN_FLIGHT 16u

typedef enum pgenhip_status {
    PGENHIP_OK = 0,
    PGENHIP_ERR_BAD_ARG = -1,     /* NULL/zero/misordered argument */
    PGENHIP_ERR_HIP = -2,         /* a HIP runtime call failed; see pgenhip_last_error_detail */
    PGENHIP_ERR_OOM = -3,         /* host or device allocation failed */
    PGENHIP_ERR_INDEX_RANGE = -4, /* kept sample index >= sample_count (ref: slice panic, src/pfile.rs:173) */
    PGENHIP_ERR_BAD_MAGIC = -5,   /* src/pfile.rs:47 */
    PGENHIP_ERR_BAD_MODE = -6,    /* src/pfile.rs:53 (only storage mode 0x02) */
    PGENHIP_ERR_BAD_FLAGS = -7,   /* src/pfile.rs:69 (byte 11 must be 0x40) */
    PGENHIP_ERR_NO_DEVICE = -8,   /* no HIP device / ordinal out of range */
    PGENHIP_ERR_TOO_LARGE = -9,   /* a size does not fit the kernel's index types */
    PGENHIP_ERR_IO = -10,         /* file read/write failed (ref: unwrap at src/pfile.rs:169-170) */
    PGENHIP_ERR_BAD_INDEX = -11,  /* variable-width file: block offsets not ascending (src/pgen.rs:160-165), tables truncated, records overlap */
    PGENHIP_ERR_COMPRESSED_RECORD = -12 /* variable-width file: a selected variant's record is not a plain 2-bit record (type != 0 or length != R) */
} pgenhip_status;

typedef struct pgenhip_ctx pgenhip_ctx;

/* ---- library-level ---------------------------------------------------- */
uint32_t pgenhip_abi_version(void);
const char *pgenhip_strerror(int status);
/* thread-local detail string of the last failing call (HIP error text etc.) */
const char *pgenhip_last_error_detail(void);
int pgenhip_device_count(int *count);

/* ---- format geometry (host-side, pure) -------------------------------- */
/* src/pfile.rs:196-200  Pfile::variant_record_size: ceil(2*N/8), u32 like the reference */
uint32_t pgenhip_variant_record_size(uint32_t sample_count);
/* src/pfile.rs:44-69  the 12-byte header: magic 6C 1B, mode 02, u32 LE variants, u32 LE samples, 0x40 */
int pgenhip_parse_header(const uint8_t header[12], uint32_t *variant_count, uint32_t *sample_count);
/* src/pfile.rs:165  byte offset of record var_idx in the file; computed in u64
 * (the reference multiplies in u32 and wraps at var_idx*R >= 2^32 — SURVEY.md F5). */
uint64_t pgenhip_record_offset(uint64_t var_idx, uint32_t record_size);
/* src/pfile.rs:156  the outer loop walks the kept-variant list in file order; a shard is a
 * contiguous slice of that iteration space.  [*begin, *end) of the n_variants
 * kept variants owned by `rank` of `world`; sizes differ by at most one, low ranks take the
 * extra.  The ONE partitioner: the C++ host's device threads, bench.py's ranks and the tests
 * all call this.  PGENHIP_ERR_BAD_ARG for world == 0 or rank >= world. */
int pgenhip_shard_range(uint64_t n_variants, uint32_t world, uint32_t rank, uint64_t *begin, uint64_t *end);

/* ---- variable-width storage modes: header and offset-table walk ----------------------
 * The reference only VALIDATES these tables (src/pgen.rs, the dead `Pgen` type: header bits :52-67, block offsets
 * :140-169, per-block type / length arrays :172-258) and its tool refuses every mode but 0x02 (src/pfile.rs:53).
 * This slice turns the same tables into per-variant byte offsets so that the records such a file stores
 * UNCOMPRESSED (record type 0: the mode-0x02 2-bit layout, length R) can be decoded in place by the kernels
 * (pgenhip_decode_emit_at); any other record type is reported, never guessed at.  Allele-count arrays
 * (allele_count_bytes != 0) are outside the slice: PGENHIP_ERR_BAD_FLAGS. */
typedef struct pgenhip_vw_header {
    uint32_t variant_count;           /* src/pgen.rs:42 */
    uint32_t sample_count;            /* :47 */
    uint8_t storage_mode;             /* :34 (0x10 = standard variable-width; the reference prints it, asserts nothing) */
    uint8_t record_type_bits;         /* :61-65  4 or 8 */
    uint8_t record_length_bytes;      /* :67     1..4 */
    uint8_t allele_count_bytes;       /* :56 */
    uint8_t provisional_ref_storage;  /* :57 */
    uint8_t reserved[3];
    uint64_t block_count;             /* :100-102  ceil(variant_count / 65 536) */
    uint64_t main_header_body_offset; /* :112-114  12 + 8 * block_count */
    uint64_t variant_records_offset;  /* :135-137  end of the type/length tables (type arrays rounded up per block, as the file stores them) */
} pgenhip_vw_header;
/* src/pgen.rs:21-98: BAD_MAGIC (:30), BAD_FLAGS (provisional_ref_storage != 1 :58, record storage mode >= 8 :64, allele counts present) */
int pgenhip_vw_parse_header(const uint8_t header[12], pgenhip_vw_header *out);
/* src/pgen.rs:140-258 turned into per-variant tables.  `index` = the file's bytes [12, variant_records_offset)
 * (index_len of them); outputs are HOST arrays of variant_count entries: the record's type (4- or 8-bit value),
 * its length and its byte offset in the file (block offset + lengths of the block's earlier records).
 * PGENHIP_ERR_BAD_INDEX: table truncated, block offsets not strictly ascending (:160-165), a block's records
 * run into the next block, the first record starts inside the tables, or offsets overflow 64 bits.
 * PGENHIP_ERR_BAD_ARG: *h is not what pgenhip_vw_parse_header produces — record_type_bits not 4 / 8, record_length_bytes
 * not 1..4, or block_count / variant_records_offset that do not follow from variant_count and those widths (the walk
 * recomputes them; nothing derived is trusted). */
int pgenhip_vw_walk_index(const pgenhip_vw_header *h, const uint8_t *index, uint64_t index_len,
                          uint8_t *record_type, uint32_t *record_len, uint64_t *record_off);
/* The selected variants (variant_idx[0..n), or the first n when NULL) must all be plain 2-bit records:
 * type 0 and length == R; writes their offsets to sel_off (n entries).  Else PGENHIP_ERR_COMPRESSED_RECORD
 * (pgenhip_last_error_detail names the first offending variant and its type). */
int pgenhip_vw_select_uncompressed(const uint8_t *record_type, const uint32_t *record_len, const uint64_t *record_off,
                                   uint32_t variant_count, const uint32_t *variant_idx, uint32_t n,
                                   uint32_t record_size, uint64_t *sel_off);

/* ---- context ---------------------------------------------------------- */
/* Binds a device and the kept-sample list (src/pfile.rs:128 sam_idx_rcs; the
 * list filter_metadata :319-333 builds is strictly ascending, and that is
 * required here).  Without PGENHIP_CREATE_KEEP_LIST, kept_idx == NULL means
 * "all samples" (K = N fast path) and a non-NULL kept_idx is a HOST array of
 * kept_count indices, copied.  With PGENHIP_CREATE_KEEP_LIST in `flags` the
 * pair (kept_idx, kept_count) IS the list whatever the pointer: kept_count
 * may be 0 (every row is then just "\n") and kept_idx may then be NULL — a
 * host whose filter kept nobody must say so with the flag, not with the
 * pointer (an empty std::vector's data() is NULL). */
#define PGENHIP_CREATE_KEEP_LIST 1u
int pgenhip_create(pgenhip_ctx **ctx, int device_ordinal, uint32_t sample_count,
                   const uint32_t *kept_idx, uint32_t kept_count, uint32_t flags);
int pgenhip_destroy(pgenhip_ctx *ctx);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream) instead of the ctx's own.
 * NULL means HIP's default (null) stream — that IS torch's current stream unless a stream
 * context is active.  pgenhip_reset_stream goes back to the ctx's own non-blocking stream. */
int pgenhip_set_stream(pgenhip_ctx *ctx, void *hip_stream);
int pgenhip_reset_stream(pgenhip_ctx *ctx);
uint32_t pgenhip_sample_count(const pgenhip_ctx *ctx);
uint32_t pgenhip_kept_count(const pgenhip_ctx *ctx);
/* 4*K + 1: bytes one variant's GT segment occupies (src/pfile.rs:186-190) */
uint64_t pgenhip_gt_row_bytes(const pgenhip_ctx *ctx);

/* ---- the hot path (device-resident, asynchronous on the ctx stream) ---- */
/* kernel selection for pgenhip_decode_emit flags (0 = automatic) */
#define PGENHIP_KERNEL_AUTO 0u
#define PGENHIP_KERNEL_ROWS 1u   /* general row-tiled kernel (any stride/alignment, list gather) */
#define PGENHIP_KERNEL_FLAT 2u   /* dense all-samples stream kernel (out_stride == 4N+1) */
#define PGENHIP_KERNEL_SCAN 3u   /* kept subset on long records: per-segment rank->sample table pick (N >= 61) */
#define PGENHIP_KERNEL_WIDE 4u   /* dense all-samples, wide LDS-staged record loads, one row piece per item (N >= 1024) */
/* 5u was round 1's stream-span kernel (measured level with WIDE, removed) */
#define PGENHIP_KERNEL_PICK 6u   /* kept subset on short records (61 <= N <= 4096, K >= 1, dense pitch): output-driven pick through the kept list */
#define PGENHIP_KERNEL_RUNS 7u   /* dense all-samples on SHORT rows (8 <= N <= ~2000, dense records, no gather): runs of rows as one work item */
#define PGENHIP_KERNEL_ROWPICK 8u /* sparse kept subset (1 <= K <= 16384) on long records: one wave per row, the row's compact record assembled in LDS, text written in one go; any strides */
#define PGENHIP_KERNEL_MASK 0xFu

/* src/pfile.rs:165-190 for a block of n_variants kept variants.
 *   row j reads the record at d_records + r*record_stride, r = d_variant_idx ? d_variant_idx[j] : j
 *   (record layout: sample s in byte s/4, bits 2*(s%4), LSB first — :172-175)
 *   and writes K x {'\t',a,'/',b} + '\n' = 4K+1 bytes at d_out + j*out_stride (:177-190).
 * All pointers are DEVICE pointers; any alignment and any strides with
 * record_stride >= R, out_stride >= 4K+1 (or n_variants <= 1).  Bytes of d_out
 * outside the n_variants segments are not touched.  No allocation, no
 * synchronisation: the launch is queued on the ctx stream (graph-capturable). */
int pgenhip_decode_emit(pgenhip_ctx *ctx, const void *d_records, uint64_t record_stride,
                        const uint32_t *d_variant_idx, uint32_t n_variants,
                        void *d_out, uint64_t out_stride, uint32_t flags);

/* Same, with the record of row j at d_base + d_record_off[j] (DEVICE array of n_variants byte offsets): the
 * uncompressed records of a variable-width file staged to HBM as it lies on disk, or any other gapped layout.
 * Every kernel family takes this gather (it replaces variant_idx * record_stride); RUNS needs dense records. */
int pgenhip_decode_emit_at(pgenhip_ctx *ctx, const void *d_base, const uint64_t *d_record_off, uint32_t n_variants,
                           void *d_out, uint64_t out_stride, uint32_t flags);

/* Full VCF body lines (src/pfile.rs:156-192): line j = prefix bytes
 * d_prefix_blob[d_prefix_off[j] .. d_prefix_off[j+1]) (pvar columns + '\t' each,
 * then "GT", :157-161) + GT segment + '\n', written at d_out + d_line_off[j].
 * d_prefix_off/d_line_off are device arrays of n_variants+1 u64 with
 * d_line_off[j+1]-d_line_off[j] == prefix_len(j) + 4K + 1 (lines packed back to back);
 * max_prefix_bytes is a host-known upper bound of any prefix length.  It must be a TRUE bound: the kernels size their LDS staging
 * and their seam passes by it; with a smaller value lines come out wrong (nothing is read or written out of bounds).
 * flags: PGENHIP_KERNEL_AUTO picks by shape — all samples kept: the work-queue stream kernel from 1 400 samples (GT segments
 * in place behind their prefixes), below it runs of whole lines assembled in LDS (short prefixes, N < 1 000) or the pick
 * family's interiors + seams kernel; a kept subset: the pick family on records of up to 4 096 samples, the segment kernels or
 * the two passes on longer ones, the general kernel for tiny shapes.  Every kernel writes the prefixes itself (no separate
 * copy).  PGENHIP_KERNEL_ROWS, _WIDE, _SCAN, _PICK, _RUNS or _ROWPICK force one (tests, A/B). */
int pgenhip_emit_lines(pgenhip_ctx *ctx, const void *d_records, uint64_t record_stride,
                       const uint32_t *d_variant_idx, uint32_t n_variants,
                       const void *d_prefix_blob, const uint64_t *d_prefix_off,
                       const uint64_t *d_line_off, uint64_t max_prefix_bytes,
                       void *d_out, uint32_t flags);

/* Launch-shape knobs of one ctx (tests force small grids to exercise ring re-use; A/B probes).
 * value 0 restores the built-in default of a knob unless noted. */
typedef enum pgenhip_knob {
    PGENHIP_KNOB_WIDE_BLOCKS_PER_CU = 1, /* stream kernel: resident blocks per CU (default: occupancy API) */
    PGENHIP_KNOB_WIDE_RANGES = 2,        /* stream kernel: work-queue ranges (write fronts of a launch), a power of two up to 64; default 0 = by shape (8 for rows of more than 16 KiB of text, else 2) */
    PGENHIP_KNOB_FLAT_BLOCKS_PER_CU = 3, /* flat kernel: grid cap per CU (default 64) */
    PGENHIP_KNOB_SCAN_BLOCKS_PER_CU = 4, /* segment kernel: resident blocks per CU (default: 2 from ~0.6 % kept, else the occupancy API's 3) */
    /* 5 was the band override of round 1's three-segment gather kernel (removed) */
    PGENHIP_KNOB_PICK_BATCH_BYTES = 6,   /* short-record pick kernel: text bytes per batch (default 32768) */
    PGENHIP_KNOB_SCAN_XCD_MAP = 8,       /* segment kernels: 1 (default) all blocks of a row group on one XCD, -1 plain block map */
    PGENHIP_KNOB_SCAN_TWO_PASS = 9,      /* sparse keeps on long records: 1 (default) compact pass + all-samples pass, -1 single-pass segment kernel */
    PGENHIP_KNOB_SCAN_CHUNK_ROWS = 10,   /* two-pass path: rows per chunk (default: as many as the 64-MiB compact scratch holds) */
    PGENHIP_KNOB_ROWPICK_BLOCKS_PER_CU = 11, /* row-owner kernel: cap on resident blocks per CU (default: occupancy API) */
    PGENHIP_KNOB_SCAN_ROWPICK = 12,      /* kept subsets on long records, launches of many rows: 1 (default) the row-owner kernel where it measures ahead (text in one pass for 2-20 % kept and for records of barely more than one segment; compact pass of the two passes below 2 %), 2 also in ONE pass across the two-pass band, -1 never (segment kernels) */
    PGENHIP_KNOB_PICK_LINE_SEAMS = 13,   /* short dense records, full lines: 1 (default) rows' interiors + batched seams (every byte written once in a whole chunk), -1 round 2's row-by-row flush */
    PGENHIP_KNOB_FLUSH_UNROLL = 14,      /* segment / row-owner kernels: 16-byte chunks per lane and step of the text flush (1, 2 or 4 — 4 in the segment kernel only; default 2) */
    PGENHIP_KNOB_SCAN_FOUR_PICKS = 15,   /* segment / row-owner kernels' text flush: 1 (default) four picks per 16-byte chunk, the fifth text from the next lane, table entries fetched four at a time; -1 round 2's five picks per chunk */
    PGENHIP_KNOB_ALIGN_STORES = 16,      /* subset kernels (segment, row-owner, pick): 1 (default) lanes <-> chunks shifted so that every store instruction covers whole 128-byte lines, -1 from the run's first whole chunk */
    PGENHIP_KNOB_RUNS_ROWS = 7           /* RUNS kernel: rows per work item (default: as many as one wide load / one span holds) */
} pgenhip_knob;
int pgenhip_tune(pgenhip_ctx *ctx, uint32_t knob, int32_t value);

/* Block until everything queued on the ctx stream has finished. */
int pgenhip_wait(pgenhip_ctx *ctx);

/* hipEvent pair on the ctx stream: start ... launches ... stop -> elapsed ms (stop synchronises). */
int pgenhip_timer_start(pgenhip_ctx *ctx);
int pgenhip_timer_stop(pgenhip_ctx *ctx, float *elapsed_ms);
/* Pipelined form: mark records the stop event without blocking; read returns the elapsed ms once
 * the stream has passed the mark (e.g. after pgenhip_wait). */
int pgenhip_timer_mark(pgenhip_ctx *ctx);
int pgenhip_timer_read(pgenhip_ctx *ctx, float *elapsed_ms);

/* ---- device / pinned memory for hosts that do not link HIP themselves --- */
int pgenhip_device_malloc(pgenhip_ctx *ctx, void **d_ptr, size_t bytes);
int pgenhip_device_free(pgenhip_ctx *ctx, void *d_ptr);
int pgenhip_host_malloc_pinned(pgenhip_ctx *ctx, void **h_ptr, size_t bytes);
int pgenhip_host_free_pinned(pgenhip_ctx *ctx, void *h_ptr);
/* asynchronous on the ctx stream (truly async only from/to pinned host memory) */
int pgenhip_memcpy_h2d(pgenhip_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int pgenhip_memcpy_d2h(pgenhip_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);

/* ---- synthetic records generated on the device --------- */
#define PGENHIP_SYNTH_DIRTY_PAD 1u
/* "hwe" value distribution instead of uniform codes: variant v has allele frequency
 * p = (655 + splitmix64((seed ^ 0x4D4146) + v) % 32113) / 65536 in [0.01, 0.5); sample s draws two alleles from
 * h = splitmix64(splitmix64((seed ^ 0x485745) + v) + s) (low two 16-bit fields < p * 65536) -> codes 0/1/2 in
 * Hardy-Weinberg proportions, and is missing (code 3) when h >> 32 < 4294967 (0.1 %).  Pad bits zero. */
#define PGENHIP_SYNTH_HWE 2u
/* record bytes of variant v = LE words splitmix64(seed + (v << 20) + word_idx), truncated to R;
 * pad bits of the last byte zeroed unless PGENHIP_SYNTH_DIRTY_PAD (the tests hold a CPU twin). */
int pgenhip_synth_records(pgenhip_ctx *ctx, void *d_dst, uint64_t record_stride,
                          uint64_t first_variant, uint32_t n_variants,
                          uint64_t seed, uint32_t flags);

#ifdef __cplusplus
}
#endif
#endif
