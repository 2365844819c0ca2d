/*
 * pgen_oracle.h — CPU oracle for the pgen-rs GT decode/emit hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (pgen_rs_amd/, the
 * libpgen_hip.so C-ABI, the host CLI) may include, link, dlopen or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / the timed CPU baseline.
 *
 * It is a plain-C restatement of the reference's algorithm; every function
 * cites the reference lines (teoremma/pgen-rs, /root/reference) it follows.
 *
 * PARITY PIN STATUS: *parity unpinned by the reference*.  The reference has
 * no tests and no golden vectors for this path, its .pgen blobs are missing
 * from the mount, and there is no Rust toolchain to run it.
 * The oracle is pinned instead by (1) line-by-line review against
 * src/pfile.rs:165-190,196-200, (2) an independent numpy decoder
 * (tests/golden/make_golden.py) whose outputs are committed under
 * tests/golden/, (3) the hand-written truth table 0xE4 -> "\t0/0\t0/1\t1/1\t./."
 * and (4) metadata known-answers computed from data/basic1/basic1.{pvar,psam}.
 */
#ifndef PGEN_ORACLE_H
#define PGEN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/pfile.rs:196-200  Pfile::variant_record_size (u32 arithmetic). */
uint32_t pgo_variant_record_size(uint32_t num_samples);

/* src/pfile.rs:44-69  header checks of Pfile::from_prefix.
 * 0 = ok; -1 bad magic (:47); -2 storage mode != 0x02 (:53); -3 byte 11 != 0x40 (:69).
 * (The reference panics at those lines; the oracle reports which assert fired.) */
int pgo_parse_header(const uint8_t hdr[12], uint32_t *num_variants, uint32_t *num_samples);

/* src/pfile.rs:165  record offset.  _exact widens before multiplying (what the
 * build matches); _ref_u32_wrap reproduces the reference's u32 multiply, which
 * wraps silently in --release once var_idx*R >= 2^32. */
uint64_t pgo_record_offset_exact(uint64_t var_idx, uint32_t record_size);
uint64_t pgo_record_offset_ref_u32_wrap(uint64_t var_idx, uint32_t record_size);

/* src/pfile.rs:171-190  the hot loop on an in-memory block of records.
 * For output row j: record = records + (variant_idx ? variant_idx[j] : j) * record_stride;
 * for each kept sample index s (ascending list kept_idx[0..kept_count), or all
 * s in 0..num_samples when kept_idx == NULL) emit '\t' + {"0/0","0/1","1/1","./."}
 * [(rec[s/4] >> (s%4*2)) & 3]; then '\n'.  Row j goes to out + j*out_stride
 * (exactly 4*kept_count+1 bytes written per row).  Returns 0, or -1 on a kept
 * index >= num_samples (the reference would panic on the slice index, :173). */
int pgo_decode_emit(const uint8_t *records, uint64_t record_stride,
                    const uint32_t *variant_idx, uint32_t n_variants,
                    uint32_t num_samples,
                    const uint32_t *kept_idx, uint32_t kept_count,
                    uint8_t *out, uint64_t out_stride);

/* src/pfile.rs:156-192  full-line body on an in-memory block: for row j the
 * bytes prefix_blob[prefix_off[j] .. prefix_off[j+1]) (the pvar columns each
 * followed by '\t', then "GT", :157-161) followed by the GT segment and '\n'
 * (:171-190), rows packed back to back at out + line_off[j].  line_off has
 * n_variants+1 entries and must satisfy
 * line_off[j+1]-line_off[j] == prefix_len(j) + 4*kept_count + 1. */
int pgo_emit_lines(const uint8_t *records, uint64_t record_stride,
                   const uint32_t *variant_idx, uint32_t n_variants,
                   uint32_t num_samples,
                   const uint32_t *kept_idx, uint32_t kept_count,
                   const uint8_t *prefix_blob, const uint64_t *prefix_off,
                   const uint64_t *line_off, uint8_t *out);

/* src/pfile.rs:149-192  file-to-file literal restatement, used as the timed
 * CPU baseline: File::open (unbuffered, :149-152); per kept variant a fresh
 * zeroed Vec (:168), seek (:169), read_exact (:170); per genotype two
 * BufWriter::write calls (:186-187) into an 8 KiB BufWriter (std default).
 * prefixes may be NULL (GT segments only) or n_var NUL-terminated strings
 * written before each row's genotypes.  wrap_u32 != 0 selects the reference's
 * u32-wrapping offset (:165).  Returns 0 or -errno-style negative. */
int pgo_output_vcf_body_file(const char *pgen_path, uint32_t num_samples,
                             const uint32_t *var_idx, uint32_t n_var,
                             const uint32_t *kept_idx, uint32_t kept_count,
                             const char *const *prefixes,
                             const char *out_path, int append, int wrap_u32);

/* Synthetic inputs.  Not from the reference: the counter-based
 * generator the HIP library's pgenhip_synth_records must reproduce bit-exactly. */
uint64_t pgo_splitmix64(uint64_t x);
void pgo_synth_records(uint8_t *dst, uint64_t record_stride, uint32_t num_samples,
                       uint64_t first_variant, uint32_t n_variants,
                       uint64_t seed, int dirty_pad);
/* "hwe" value distribution: per-variant allele frequency in [0.01, 0.5), codes 0/1/2 in Hardy-Weinberg
 * proportions, 0.1 % missing; integer-only, see pgen_oracle.c.  Twin of PGENHIP_SYNTH_HWE. */
void pgo_synth_records_hwe(uint8_t *dst, uint64_t record_stride, uint32_t num_samples,
                           uint64_t first_variant, uint32_t n_variants, uint64_t seed);
/* keep sample i iff splitmix64(seed ^ i) % modulus == 0; writes ascending
 * indices to kept_idx (capacity cap) and returns the kept count. */
uint32_t pgo_synth_keep(uint32_t num_samples, uint64_t seed, uint32_t modulus,
                        uint32_t *kept_idx, uint32_t cap);

/* ---- variable-width storage modes: header / offset-table walk (pgen_vw_oracle.c) ------------------------
 * Restates the reference's `Pgen` validator, /root/reference/src/pgen.rs (dead code there: SURVEY.md F1). */
typedef struct pgo_vw_header {
    uint8_t storage_mode;            /* src/pgen.rs:34 */
    uint32_t variant_count;          /* :42 */
    uint32_t sample_count;           /* :47 */
    uint8_t record_type_bits;        /* :61-65  4 or 8 */
    uint8_t record_length_bytes;     /* :67     1..4 */
    uint8_t allele_count_bytes;      /* :56 */
    uint8_t provisional_ref_storage; /* :57 */
} pgo_vw_header;
/* :21-98  0 ok; -1 magic (:30); -2 provisional_ref_storage != 1 (:58); -3 invalid record storage mode (:64) */
int pgo_vw_parse_header(const uint8_t hdr[12], pgo_vw_header *h);
uint64_t pgo_vw_variant_block_count(const pgo_vw_header *h);     /* :100-102 */
uint64_t pgo_vw_main_header_body_offset(const pgo_vw_header *h); /* :104-114 */
uint64_t pgo_vw_main_header_body_size(const pgo_vw_header *h);   /* :116-133 */
uint64_t pgo_vw_variant_records_offset(const pgo_vw_header *h);  /* :135-137 */
/* :140-169  position behind the block offsets; -1 short file, -2 not strictly ascending */
int64_t pgo_vw_check_variant_block_offsets(const pgo_vw_header *h, const uint8_t *file, uint64_t file_len, uint64_t offset);
/* :172-258  position behind the last block (with the reference's last-block quirk); sets of distinct type values
 * and distinct length bytes as 256-entry flag arrays; -1 short file */
int64_t pgo_vw_check_main_header_body(const pgo_vw_header *h, const uint8_t *file, uint64_t file_len, uint64_t offset,
                                      uint8_t types_seen[256], uint8_t length_bytes_seen[256]);
/* NOT in the reference (parity unpinned): per-variant record type / length / file offset from the same tables.
 * 0 ok; -1 short file; -2 block offsets not ascending; -3 a block's records overrun the next block's offset */
int pgo_vw_index(const pgo_vw_header *h, const uint8_t *file, uint64_t file_len,
                 uint8_t *types, uint32_t *lens, uint64_t *offs);

/* src/pfile.rs:171-190 on records addressed by BYTE OFFSET (a variable-width file's uncompressed records,
 * or any gapped layout): row j's record starts at base + record_off[j].  Otherwise as pgo_decode_emit. */
int pgo_decode_emit_at(const uint8_t *base, const uint64_t *record_off, uint32_t n_variants,
                       uint32_t num_samples, const uint32_t *kept_idx, uint32_t kept_count,
                       uint8_t *out, uint64_t out_stride);

#ifdef __cplusplus
}
#endif
#endif
