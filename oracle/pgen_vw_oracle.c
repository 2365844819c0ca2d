/*
 * pgen_vw_oracle.c — CPU oracle for the header / offset-table walk of the VARIABLE-WIDTH .pgen
 * storage modes.  TEST INFRASTRUCTURE ONLY (see pgen_oracle.h).
 *
 * Part 1 restates /root/reference/src/pgen.rs as literally as C allows — the `Pgen` header
 * validator (dead code in the reference: declared at src/main.rs:3, never called).  The reference
 * reads from a file through BufReader; here the file's bytes are a buffer and positions are
 * offsets into it.  Its two quirks are KEPT and documented where they occur.
 *
 * Part 2 (pgo_vw_index) is NOT in the reference: pgen-rs never computes per-variant record
 * offsets for these modes (the tool asserts mode 0x02, src/pfile.rs:53).  It is the plain
 * reading of the same tables the validator walks — record i of a 65 536-variant block starts
 * where the block starts plus the lengths of the block's earlier records — written here so the
 * HIP library's index walk has an independent twin.  PARITY UNPINNED: no reference output, test
 * or fixture covers it (every .pgen is missing from the mount).
 */
#include "pgen_oracle.h"

#include <string.h>

#define VARIANT_BLOCK_SIZE ((uint64_t)1 << 16) /* src/pgen.rs:19 */

/* src/pgen.rs:21-98  Pgen::from_file_path up to the struct literal */
int pgo_vw_parse_header(const uint8_t hdr[12], pgo_vw_header *h)
{
    memset(h, 0, sizeof *h);
    if (hdr[0] != 0x6c || hdr[1] != 0x1b) return -1; /* :30 assert_eq!(magic_number, [0x6c, 0x1b]) */
    h->storage_mode = hdr[2];                        /* :34  (the assert on 0x10 is commented out, :36) */
    /* :42, :47  u32::from_le_bytes */
    h->variant_count = (uint32_t)hdr[3] | ((uint32_t)hdr[4] << 8) | ((uint32_t)hdr[5] << 16) | ((uint32_t)hdr[6] << 24);
    h->sample_count = (uint32_t)hdr[7] | ((uint32_t)hdr[8] << 8) | ((uint32_t)hdr[9] << 16) | ((uint32_t)hdr[10] << 24);
    const uint8_t header_format_byte = hdr[11];                             /* :52 */
    const uint8_t record_storage_mode = header_format_byte & 0x0F;          /* :55 */
    h->allele_count_bytes = (uint8_t)((header_format_byte & (3u << 4)) >> 4);      /* :56 */
    h->provisional_ref_storage = (uint8_t)((header_format_byte & (3u << 6)) >> 6); /* :57 */
    if (h->provisional_ref_storage != 1) return -2;                         /* :58 assert_eq!(provisional_ref_storage, 0b01) */
    switch (record_storage_mode / 4) {                                      /* :61-65 */
        case 0: h->record_type_bits = 4; break;
        case 1: h->record_type_bits = 8; break;
        default: return -3;                                                 /* :64 panic!("invalid record storage mode") */
    }
    h->record_length_bytes = (uint8_t)(record_storage_mode % 4 + 1);        /* :67 */
    return 0;
}

/* src/pgen.rs:100-102 */
uint64_t pgo_vw_variant_block_count(const pgo_vw_header *h)
{
    return ((uint64_t)h->variant_count + VARIANT_BLOCK_SIZE - 1) / VARIANT_BLOCK_SIZE;
}

/* src/pgen.rs:104-114 */
uint64_t pgo_vw_main_header_body_offset(const pgo_vw_header *h)
{
    return 12u + pgo_vw_variant_block_count(h) * 8u;
}

/* src/pgen.rs:116-133.  QUIRK kept: the type array is sized for ALL variants at once ("+= 4" bits
 * when the total is odd), although the file rounds it up per block — equal unless an inner block
 * has an odd variant count, which blocks of 65 536 never have. */
uint64_t pgo_vw_main_header_body_size(const pgo_vw_header *h)
{
    uint64_t all_record_types_size = (uint64_t)h->variant_count * h->record_type_bits;
    if (all_record_types_size % 8 != 0) all_record_types_size += 4;
    all_record_types_size /= 8;
    const uint64_t all_record_lengths_size = (uint64_t)h->variant_count * h->record_length_bytes;
    return all_record_types_size + all_record_lengths_size;
}

/* src/pgen.rs:135-137 */
uint64_t pgo_vw_variant_records_offset(const pgo_vw_header *h)
{
    return pgo_vw_main_header_body_offset(h) + pgo_vw_main_header_body_size(h);
}

/* src/pgen.rs:140-169  check_variant_block_offsets: reads variant_block_count u64 LE values at `offset`,
 * panics unless strictly ascending (:160-166), returns the position behind them.
 * -1 = short file (read_exact unwrap), -2 = not ascending. */
int64_t pgo_vw_check_variant_block_offsets(const pgo_vw_header *h, const uint8_t *file, uint64_t file_len, uint64_t offset)
{
    const uint64_t n = pgo_vw_variant_block_count(h);
    uint64_t prev = 0;
    for (uint64_t b = 0; b < n; b++) {
        if (offset + 8 > file_len) return -1;
        uint64_t v = 0;
        for (int k = 0; k < 8; k++) v |= (uint64_t)file[offset + (uint64_t)k] << (8 * k); /* :151 u64::from_le_bytes */
        if (b > 0 && !(prev < v)) return -2;                                               /* :160-165 windows(2).all(w[0] < w[1]) */
        prev = v;
        offset += 8;
    }
    return (int64_t)offset; /* :168 */
}

/* src/pgen.rs:172-258  check_main_header_body: per block a packed array of record types, then a packed array of
 * record lengths; collects the distinct type values and the distinct length BYTES (:238-240 inserts every byte of
 * the length array, not the little-endian values) and returns the position behind the last block.
 * QUIRK kept (:200-204): the last block's variant count is variant_count % 65 536 — ZERO when the count is a
 * multiple of 65 536, so the walk then stops one block short of variant_records_offset and the caller's
 * assert_eq (:92) fails.  -1 = short file. */
int64_t pgo_vw_check_main_header_body(const pgo_vw_header *h, const uint8_t *file, uint64_t file_len, uint64_t offset,
                                      uint8_t types_seen[256], uint8_t length_bytes_seen[256])
{
    const uint64_t n_blocks = pgo_vw_variant_block_count(h);
    memset(types_seen, 0, 256);
    memset(length_bytes_seen, 0, 256);
    uint64_t types_pos = offset, lengths_pos = offset; /* :188-193 two readers at the same position */
    for (uint64_t block = 0; block < n_blocks; block++) {
        const uint64_t block_variant_count =
            block == n_blocks - 1 ? (uint64_t)h->variant_count % VARIANT_BLOCK_SIZE : VARIANT_BLOCK_SIZE; /* :200-204 */
        uint64_t types_block_size = block_variant_count * h->record_type_bits; /* :207-212 */
        if (types_block_size % 8 != 0) types_block_size += 4;
        types_block_size /= 8;
        const uint64_t lengths_block_size = block_variant_count * h->record_length_bytes; /* :214 */
        lengths_pos += types_block_size;                                                  /* :217-219 */
        if (types_pos + types_block_size > file_len) return -1;
        for (uint64_t k = 0; k < types_block_size; k++) {                                 /* :222-233 */
            const uint8_t byte = file[types_pos + k];
            if (h->record_type_bits == 4) {
                types_seen[byte >> 4] = 1;
                types_seen[byte & 0x0F] = 1;
            } else {
                types_seen[byte] = 1;
            }
        }
        if (lengths_pos + lengths_block_size > file_len) return -1;
        for (uint64_t k = 0; k < lengths_block_size; k++) length_bytes_seen[file[lengths_pos + k]] = 1; /* :236-240 */
        lengths_pos += lengths_block_size;
        types_pos = lengths_pos; /* :244-247 */
    }
    return (int64_t)lengths_pos; /* :257 */
}

/* ---- Part 2: not in the reference (see the header comment) ---------------------------------------------------
 * Per-variant record type, length and file offset from the tables above.  Layout as the validator walks it
 * (block b = variants b*65 536 ..): types packed record_type_bits each — for 4 bits the EVEN variant of a pair in
 * the low nibble (the PLINK 2 convention; the reference inserts both nibbles into a set and never orders them, :226-233) —
 * then lengths, record_length_bytes each, little endian.  The last block holds what is left of variant_count
 * (65 536 when the count is a multiple of the block size: the intended reading, not the quirk of :200-204).
 * Returns 0; -1 short file; -2 block offsets not ascending; -3 a block's records overrun the next block's offset. */
int pgo_vw_index(const pgo_vw_header *h, const uint8_t *file, uint64_t file_len,
                 uint8_t *types, uint32_t *lens, uint64_t *offs)
{
    const uint64_t n_blocks = pgo_vw_variant_block_count(h);
    if (12u + n_blocks * 8u > file_len) return -1;
    uint64_t pos = 12u + n_blocks * 8u;
    uint64_t prev_block_off = 0;
    for (uint64_t b = 0; b < n_blocks; b++) {
        uint64_t block_off = 0;
        for (int k = 0; k < 8; k++) block_off |= (uint64_t)file[12u + b * 8u + (uint64_t)k] << (8 * k);
        if (b > 0 && !(prev_block_off < block_off)) return -2;
        const uint64_t first = b * VARIANT_BLOCK_SIZE;
        const uint64_t left = (uint64_t)h->variant_count - first;
        const uint64_t cnt = left < VARIANT_BLOCK_SIZE ? left : VARIANT_BLOCK_SIZE;
        const uint64_t types_bytes = (cnt * h->record_type_bits + 7u) / 8u;
        const uint64_t lens_bytes = cnt * h->record_length_bytes;
        if (pos + types_bytes + lens_bytes > file_len) return -1;
        uint64_t rec_off = block_off;
        if (b > 0 && offs[first - 1] + lens[first - 1] > block_off) return -3;
        for (uint64_t i = 0; i < cnt; i++) {
            uint8_t t;
            if (h->record_type_bits == 4) {
                const uint8_t byte = file[pos + i / 2u];
                t = (i & 1u) ? (uint8_t)(byte >> 4) : (uint8_t)(byte & 0x0F);
            } else {
                t = file[pos + i];
            }
            uint32_t len = 0;
            for (uint32_t k = 0; k < h->record_length_bytes; k++)
                len |= (uint32_t)file[pos + types_bytes + i * h->record_length_bytes + k] << (8u * k);
            types[first + i] = t;
            lens[first + i] = len;
            offs[first + i] = rec_off;
            rec_off += len;
        }
        pos += types_bytes + lens_bytes;
        prev_block_off = block_off;
    }
    return 0;
}
