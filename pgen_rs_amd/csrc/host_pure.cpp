// host_pure.cpp — the entry points of include/pgen_hip.h that never touch a device: format geometry, the variant
// partitioner, the header / offset-table walk of the variable-width storage modes, status strings.  Plain C++ (no HIP):
// compiled into libpgen_hip.so by hipcc and, for the CPU sanitizer leg, on its own with g++ -fsanitize=address,undefined
// (tests/test_sanitizers.py fuzzes the table walk with it — these functions parse bytes that come straight from a file).
#include "../../include/pgen_hip.h"

#include <algorithm>
#include <string>

#include "host_pure.h"

namespace pgenhip {

namespace {
thread_local std::string g_detail;
}

int set_detail(int status, const char *what)
{
    g_detail = what ? what : "";
    return status;
}

const char *g_detail_c_str() { return g_detail.c_str(); }

}  // namespace pgenhip

namespace {
int fail(int status, const char *what) { return pgenhip::set_detail(status, what); }
}  // namespace

extern "C" {

uint32_t pgenhip_abi_version(void) { return PGENHIP_ABI_VERSION; }

const char *pgenhip_strerror(int status)
{
    switch (status) {
        case PGENHIP_OK: return "ok";
        case PGENHIP_ERR_BAD_ARG: return "bad argument";
        case PGENHIP_ERR_HIP: return "HIP runtime error";
        case PGENHIP_ERR_OOM: return "out of memory";
        case PGENHIP_ERR_INDEX_RANGE: return "kept sample index out of range";
        case PGENHIP_ERR_BAD_MAGIC: return "not a .pgen file (magic bytes != 6C 1B)";
        case PGENHIP_ERR_BAD_MODE: return "unsupported .pgen storage mode (only 0x02 fixed-width)";
        case PGENHIP_ERR_BAD_FLAGS: return "unexpected .pgen header flag byte (expected 0x40)";
        case PGENHIP_ERR_NO_DEVICE: return "no usable HIP device";
        case PGENHIP_ERR_TOO_LARGE: return "size exceeds kernel index range";
        case PGENHIP_ERR_IO: return "I/O error";
        case PGENHIP_ERR_BAD_INDEX: return "variable-width .pgen: bad block-offset / record-length tables";
        case PGENHIP_ERR_COMPRESSED_RECORD: return "variable-width .pgen: selected record is not a plain 2-bit record";
        default: return "unknown status";
    }
}

const char *pgenhip_last_error_detail(void) { return pgenhip::g_detail_c_str(); }

// src/pfile.rs:196-200
uint32_t pgenhip_variant_record_size(uint32_t sample_count)
{
    uint32_t bit_size = sample_count * 2u;
    return bit_size / 8u + ((bit_size % 8u) ? 1u : 0u);
}

// src/pfile.rs:44-69
int pgenhip_parse_header(const uint8_t header[12], uint32_t *variant_count, uint32_t *sample_count)
{
    if (!header || !variant_count || !sample_count) return fail(PGENHIP_ERR_BAD_ARG, "NULL argument");
    if (header[0] != 0x6C || header[1] != 0x1B) return fail(PGENHIP_ERR_BAD_MAGIC, "magic");
    if (header[2] != 0x02) return fail(PGENHIP_ERR_BAD_MODE, "storage mode");
    *variant_count = (uint32_t)header[3] | (uint32_t)header[4] << 8 | (uint32_t)header[5] << 16 | (uint32_t)header[6] << 24;
    *sample_count = (uint32_t)header[7] | (uint32_t)header[8] << 8 | (uint32_t)header[9] << 16 | (uint32_t)header[10] << 24;
    if (header[11] != 0x40) return fail(PGENHIP_ERR_BAD_FLAGS, "flag byte");
    return PGENHIP_OK;
}

// src/pfile.rs:165, widened before the multiply
uint64_t pgenhip_record_offset(uint64_t var_idx, uint32_t record_size)
{
    return 12ull + var_idx * (uint64_t)record_size;
}

// src/pfile.rs:156: contiguous slices of the kept-variant iteration space
int pgenhip_shard_range(uint64_t n_variants, uint32_t world, uint32_t rank, uint64_t *begin, uint64_t *end)
{
    if (!begin || !end) return fail(PGENHIP_ERR_BAD_ARG, "NULL argument");
    if (world == 0u || rank >= world) return fail(PGENHIP_ERR_BAD_ARG, "rank outside [0, world)");
    const uint64_t base = n_variants / world, extra = n_variants % world;
    *begin = (uint64_t)rank * base + std::min<uint64_t>(rank, extra);
    *end = *begin + base + (rank < extra ? 1u : 0u);
    return PGENHIP_OK;
}

// ---- variable-width storage modes (src/pgen.rs) ------------------------------------------------------------
// src/pgen.rs:21-98
int pgenhip_vw_parse_header(const uint8_t header[12], pgenhip_vw_header *out)
{
    if (!header || !out) return fail(PGENHIP_ERR_BAD_ARG, "NULL argument");
    *out = pgenhip_vw_header{};
    if (header[0] != 0x6C || header[1] != 0x1B) return fail(PGENHIP_ERR_BAD_MAGIC, "magic");  // :30
    out->storage_mode = header[2];                                                             // :34
    out->variant_count = (uint32_t)header[3] | (uint32_t)header[4] << 8 | (uint32_t)header[5] << 16 | (uint32_t)header[6] << 24;
    out->sample_count = (uint32_t)header[7] | (uint32_t)header[8] << 8 | (uint32_t)header[9] << 16 | (uint32_t)header[10] << 24;
    const uint8_t fmt = header[11];                       // :52
    const uint8_t record_storage_mode = fmt & 0x0Fu;      // :55
    out->allele_count_bytes = (uint8_t)((fmt >> 4) & 3u); // :56
    out->provisional_ref_storage = (uint8_t)(fmt >> 6);   // :57
    if (out->provisional_ref_storage != 1u) return fail(PGENHIP_ERR_BAD_FLAGS, "provisional_ref_storage != 1");  // :58
    if (record_storage_mode >= 8u) return fail(PGENHIP_ERR_BAD_FLAGS, "invalid record storage mode");             // :61-65
    out->record_type_bits = record_storage_mode < 4u ? 4u : 8u;
    out->record_length_bytes = (uint8_t)(record_storage_mode % 4u + 1u);  // :67
    if (out->allele_count_bytes != 0u) return fail(PGENHIP_ERR_BAD_FLAGS, "allele-count arrays are outside this slice");
    constexpr uint64_t kBlock = 1ull << 16;               // :19
    out->block_count = ((uint64_t)out->variant_count + kBlock - 1ull) / kBlock;  // :100-102
    out->main_header_body_offset = 12ull + 8ull * out->block_count;              // :112-114
    uint64_t body = 0;
    for (uint64_t b = 0; b < out->block_count; b++) {
        const uint64_t cnt = std::min<uint64_t>(kBlock, (uint64_t)out->variant_count - b * kBlock);
        body += (cnt * out->record_type_bits + 7ull) / 8ull + cnt * out->record_length_bytes;  // per block, as the file stores it (:205-214)
    }
    out->variant_records_offset = out->main_header_body_offset + body;           // :135-137
    return PGENHIP_OK;
}

// src/pgen.rs:140-258, producing per-variant tables instead of statistics
int pgenhip_vw_walk_index(const pgenhip_vw_header *h, const uint8_t *index, uint64_t index_len,
                          uint8_t *record_type, uint32_t *record_len, uint64_t *record_off)
{
    if (!h || (!index && index_len) || ((!record_type || !record_len || !record_off) && h && h->variant_count))
        return fail(PGENHIP_ERR_BAD_ARG, "NULL argument");
    // *h may be caller-made (it is a plain struct of the public ABI): nothing derived is trusted.  The table geometry is
    // recomputed from variant_count, record_type_bits and record_length_bytes and must agree with what the struct says.
    if ((h->record_type_bits != 4u && h->record_type_bits != 8u) || h->record_length_bytes < 1u || h->record_length_bytes > 4u)
        return fail(PGENHIP_ERR_BAD_ARG, "record_type_bits must be 4 or 8 and record_length_bytes 1..4");
    constexpr uint64_t kBlock = 1ull << 16;
    {
        const uint64_t blocks = ((uint64_t)h->variant_count + kBlock - 1ull) / kBlock;
        uint64_t tables = 8ull * blocks;
        for (uint64_t b = 0; b < blocks; b++) {
            const uint64_t cnt = std::min<uint64_t>(kBlock, (uint64_t)h->variant_count - b * kBlock);
            tables += (cnt * h->record_type_bits + 7ull) / 8ull + cnt * h->record_length_bytes;
        }
        if (h->block_count != blocks || h->variant_records_offset != 12ull + tables)
            return fail(PGENHIP_ERR_BAD_ARG, "block_count / variant_records_offset do not follow from variant_count and the record widths");
        if (index_len < tables) return fail(PGENHIP_ERR_BAD_INDEX, "index shorter than the header says");
    }
    auto le = [&](uint64_t pos, uint32_t n) {  // little-endian value of n <= 8 bytes at file offset 12 + pos
        uint64_t v = 0;
        for (uint32_t k = 0; k < n; k++) v |= (uint64_t)index[pos + k] << (8u * k);
        return v;
    };
    uint64_t pos = 8ull * h->block_count;  // first block's tables, relative to file offset 12
    uint64_t prev_end = h->variant_records_offset, prev_block_off = 0;
    for (uint64_t b = 0; b < h->block_count; b++) {
        const uint64_t block_off = le(8ull * b, 8);                                        // :147-152
        if (b > 0 && !(prev_block_off < block_off)) return fail(PGENHIP_ERR_BAD_INDEX, "variant block offsets are not in ascending order");  // :160-165
        if (block_off < prev_end) return fail(PGENHIP_ERR_BAD_INDEX, b ? "a block's records run into the next block" : "first record inside the tables");
        const uint64_t first = b * kBlock;
        const uint64_t cnt = std::min<uint64_t>(kBlock, (uint64_t)h->variant_count - first);
        const uint64_t types_bytes = (cnt * h->record_type_bits + 7ull) / 8ull;            // :207-212
        uint64_t off = block_off;
        for (uint64_t i = 0; i < cnt; i++) {
            uint8_t t;
            if (h->record_type_bits == 4u) {
                const uint8_t byte = index[pos + i / 2ull];                                 // :226-233: two types per byte, even variant low
                t = (i & 1ull) ? (uint8_t)(byte >> 4) : (uint8_t)(byte & 0x0Fu);
            } else {
                t = index[pos + i];
            }
            const uint32_t len = (uint32_t)le(pos + types_bytes + i * h->record_length_bytes, h->record_length_bytes);  // :214, :236-240
            record_type[first + i] = t;
            record_len[first + i] = len;
            record_off[first + i] = off;
            if (off + len < off) return fail(PGENHIP_ERR_BAD_INDEX, "record offsets overflow 64 bits");  // a block offset near 2^64
            off += len;
        }
        pos += types_bytes + cnt * h->record_length_bytes;
        prev_end = off;
        prev_block_off = block_off;
    }
    return PGENHIP_OK;
}

int pgenhip_vw_select_uncompressed(const uint8_t *record_type, const uint32_t *record_len, const uint64_t *record_off,
                                   uint32_t variant_count, const uint32_t *variant_idx, uint32_t n,
                                   uint32_t record_size, uint64_t *sel_off)
{
    if (n && (!record_type || !record_len || !record_off || !sel_off)) return fail(PGENHIP_ERR_BAD_ARG, "NULL argument");
    for (uint32_t j = 0; j < n; j++) {
        const uint32_t v = variant_idx ? variant_idx[j] : j;
        if (v >= variant_count) return fail(PGENHIP_ERR_INDEX_RANGE, "variant index >= variant_count");
        if (record_type[v] != 0u || record_len[v] != record_size) {
            const std::string msg = "variant " + std::to_string(v) + ": record type " + std::to_string(record_type[v]) + ", length " + std::to_string(record_len[v]) +
                       " (a plain 2-bit record has type 0 and length " + std::to_string(record_size) + ")";
            return fail(PGENHIP_ERR_COMPRESSED_RECORD, msg.c_str());
        }
        sel_off[j] = record_off[v];
    }
    return PGENHIP_OK;
}

}  // extern "C"
