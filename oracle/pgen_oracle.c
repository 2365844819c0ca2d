/*
 * pgen_oracle.c — CPU oracle (plain C).  TEST INFRASTRUCTURE ONLY; see
 * pgen_oracle.h for who may use it and for the "parity unpinned" statement.
 *
 * Each function restates the cited lines of /root/reference/src/pfile.rs as
 * literally as C allows: scalar, one genotype at a time, no tricks.
 */
#define _FILE_OFFSET_BITS 64
#include "pgen_oracle.h"

#include <errno.h>
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <unistd.h>

/* src/pfile.rs:196-200
 *   let bit_size = self.num_samples * 2;
 *   (bit_size / 8) + if bit_size % 8 == 0 { 0 } else { 1 }
 * u32 arithmetic as in the reference (num_samples*2 would itself wrap above
 * 2^31 samples; the .pgen header cannot express a useful file of that size). */
uint32_t pgo_variant_record_size(uint32_t num_samples)
{
    uint32_t bit_size = num_samples * 2u;
    return (bit_size / 8u) + ((bit_size % 8u == 0u) ? 0u : 1u);
}

/* src/pfile.rs:44-69 */
int pgo_parse_header(const uint8_t hdr[12], uint32_t *num_variants, uint32_t *num_samples)
{
    if (hdr[0] != 0x6C || hdr[1] != 0x1B) return -1; /* :47 assert_eq!(buf, [0x6C, 0x1B]) */
    if (hdr[2] != 0x02) return -2;                   /* :53 assert!(storage_mode == 0x02) */
    /* :57, :62  u32::from_le_bytes */
    *num_variants = (uint32_t)hdr[3] | ((uint32_t)hdr[4] << 8) | ((uint32_t)hdr[5] << 16) | ((uint32_t)hdr[6] << 24);
    *num_samples = (uint32_t)hdr[7] | ((uint32_t)hdr[8] << 8) | ((uint32_t)hdr[9] << 16) | ((uint32_t)hdr[10] << 24);
    if (hdr[11] != 0x40) return -3;                  /* :69 assert_eq!(buf, [0x40]) */
    return 0;
}

/* src/pfile.rs:165 with the multiply widened first (the mathematically intended offset). */
uint64_t pgo_record_offset_exact(uint64_t var_idx, uint32_t record_size)
{
    return 12u + var_idx * (uint64_t)record_size;
}

/* src/pfile.rs:165 literally:  12 + (*var_idx as u32 * self.variant_record_size()) as u64
 * `as u32` truncates var_idx, the product is a wrapping u32 multiply in --release. */
uint64_t pgo_record_offset_ref_u32_wrap(uint64_t var_idx, uint32_t record_size)
{
    uint32_t prod = (uint32_t)var_idx * record_size;
    return 12u + (uint64_t)prod;
}

/* src/pfile.rs:177-183  the 4-arm match */
static const char *const GT_STR[4] = {"0/0", "0/1", "1/1", "./."};

/* src/pfile.rs:171-190 */
int pgo_decode_emit(const uint8_t *records, uint64_t record_stride,
                    const uint32_t *variant_idx, uint32_t n_variants,
                    uint32_t num_samples,
                    const uint32_t *kept_idx, uint32_t kept_count,
                    uint8_t *out, uint64_t out_stride)
{
    uint32_t k_total = kept_idx ? kept_count : num_samples;
    for (uint32_t j = 0; j < n_variants; j++) {
        uint64_t row = variant_idx ? (uint64_t)variant_idx[j] : (uint64_t)j;
        const uint8_t *record_buf = records + row * record_stride;
        uint8_t *w = out + (uint64_t)j * out_stride;
        for (uint32_t k = 0; k < k_total; k++) {
            uint32_t sam_idx = kept_idx ? kept_idx[k] : k;            /* :171 */
            if (sam_idx >= num_samples) return -1;
            uint32_t sample_offset = sam_idx / 4;                     /* :172 */
            uint8_t host_byte = record_buf[sample_offset];            /* :173 */
            uint32_t in_byte_offset = sam_idx % 4;                    /* :174 */
            uint8_t encoded_genotype = (uint8_t)((host_byte >> (in_byte_offset * 2)) & 0x3); /* :175 */
            const char *genotype = GT_STR[encoded_genotype];          /* :177-183 */
            *w++ = '\t';                                              /* :186 */
            *w++ = (uint8_t)genotype[0];                              /* :187 */
            *w++ = (uint8_t)genotype[1];
            *w++ = (uint8_t)genotype[2];
        }
        *w++ = '\n';                                                  /* :190 */
    }
    return 0;
}

/* src/pfile.rs:156-192 on memory blocks: prefix (:157-161) + GT segment (:171-190). */
int pgo_emit_lines(const uint8_t *records, uint64_t record_stride,
                   const uint32_t *variant_idx, uint32_t n_variants,
                   uint32_t num_samples,
                   const uint32_t *kept_idx, uint32_t kept_count,
                   const uint8_t *prefix_blob, const uint64_t *prefix_off,
                   const uint64_t *line_off, uint8_t *out)
{
    uint32_t k_total = kept_idx ? kept_count : num_samples;
    for (uint32_t j = 0; j < n_variants; j++) {
        uint64_t plen = prefix_off[j + 1] - prefix_off[j];
        if (line_off[j + 1] - line_off[j] != plen + 4ull * k_total + 1ull) return -2;
        uint8_t *w = out + line_off[j];
        memcpy(w, prefix_blob + prefix_off[j], plen);
        uint64_t row = variant_idx ? (uint64_t)variant_idx[j] : (uint64_t)j;
        int rc = pgo_decode_emit(records + row * record_stride, record_stride, NULL, 1,
                                 num_samples, kept_idx, kept_count, w + plen, 0);
        if (rc) return rc;
    }
    return 0;
}

/* ---- std::io::BufWriter<File> restated (Rust std, default capacity 8 KiB) ----
 * BufWriter::write(buf): if buf.len() > spare capacity -> flush_buf();
 * if buf.len() >= capacity -> write straight through; else copy into the buffer. */
#define PGO_BUFWRITER_CAP 8192
typedef struct {
    int fd;
    size_t len;
    uint8_t buf[PGO_BUFWRITER_CAP];
} pgo_bufwriter;

static int pgo_write_all(int fd, const uint8_t *p, size_t n)
{
    while (n) {
        ssize_t w = write(fd, p, n);
        if (w < 0) {
            if (errno == EINTR) continue;
            return -errno;
        }
        p += (size_t)w;
        n -= (size_t)w;
    }
    return 0;
}

static int pgo_bw_flush(pgo_bufwriter *bw)
{
    int rc = pgo_write_all(bw->fd, bw->buf, bw->len);
    bw->len = 0;
    return rc;
}

static inline int pgo_bw_write(pgo_bufwriter *bw, const void *data, size_t n)
{
    if (n > PGO_BUFWRITER_CAP - bw->len) {
        int rc = pgo_bw_flush(bw);
        if (rc) return rc;
    }
    if (n >= PGO_BUFWRITER_CAP) return pgo_write_all(bw->fd, (const uint8_t *)data, n);
    memcpy(bw->buf + bw->len, data, n);
    bw->len += n;
    return 0;
}

/* src/pfile.rs:149-192 */
int pgo_output_vcf_body_file(const char *pgen_path, uint32_t num_samples,
                             const uint32_t *var_idx, uint32_t n_var,
                             const uint32_t *kept_idx, uint32_t kept_count,
                             const char *const *prefixes,
                             const char *out_path, int append, int wrap_u32)
{
    int rc = 0;
    uint32_t record_size = pgo_variant_record_size(num_samples);
    uint32_t k_total = kept_idx ? kept_count : num_samples;
    int pgen = open(pgen_path, O_RDONLY);                       /* :149 File::open, unbuffered (:150-152) */
    if (pgen < 0) return -errno;
    int ofd = open(out_path, O_WRONLY | O_CREAT | (append ? O_APPEND : O_TRUNC), 0644); /* :136 */
    if (ofd < 0) {
        rc = -errno;
        close(pgen);
        return rc;
    }
    pgo_bufwriter *bw = (pgo_bufwriter *)malloc(sizeof(pgo_bufwriter)); /* :137 */
    if (!bw) {
        close(pgen);
        close(ofd);
        return -ENOMEM;
    }
    bw->fd = ofd;
    bw->len = 0;
    for (uint32_t j = 0; j < n_var && rc == 0; j++) {            /* :156 */
        uint64_t vi = var_idx ? (uint64_t)var_idx[j] : (uint64_t)j;
        if (prefixes) {                                          /* :157-161, already joined by the caller */
            rc = pgo_bw_write(bw, prefixes[j], strlen(prefixes[j]));
            if (rc) break;
        }
        uint64_t record_offset = wrap_u32 ? pgo_record_offset_ref_u32_wrap(vi, record_size)
                                          : pgo_record_offset_exact(vi, record_size); /* :165 */
        uint8_t *record_buf = (uint8_t *)calloc(record_size ? record_size : 1, 1);    /* :168 vec![0u8; R] */
        if (!record_buf) {
            rc = -ENOMEM;
            break;
        }
        if (lseek(pgen, (off_t)record_offset, SEEK_SET) < 0) rc = -errno;             /* :169 */
        size_t got = 0;
        while (rc == 0 && got < record_size) {                                       /* :170 read_exact */
            ssize_t r = read(pgen, record_buf + got, record_size - got);
            if (r < 0) {
                if (errno == EINTR) continue;
                rc = -errno;
            } else if (r == 0) {
                rc = -EIO; /* UnexpectedEof -> unwrap() panic in the reference */
            } else {
                got += (size_t)r;
            }
        }
        for (uint32_t k = 0; rc == 0 && k < k_total; k++) {                           /* :171 */
            uint32_t sam_idx = kept_idx ? kept_idx[k] : k;
            uint32_t sample_offset = sam_idx / 4;                                     /* :172 */
            uint8_t host_byte = record_buf[sample_offset];                            /* :173 */
            uint32_t in_byte_offset = sam_idx % 4;                                    /* :174 */
            uint8_t encoded_genotype = (uint8_t)((host_byte >> (in_byte_offset * 2)) & 0x3); /* :175 */
            const char *genotype = GT_STR[encoded_genotype];                          /* :177-183 */
            rc = pgo_bw_write(bw, "\t", 1);                                           /* :186 */
            if (rc == 0) rc = pgo_bw_write(bw, genotype, 3);                          /* :187 */
        }
        if (rc == 0) rc = pgo_bw_write(bw, "\n", 1);                                  /* :190 */
        free(record_buf);                                                             /* Vec drop at end of iteration */
    }
    if (rc == 0) rc = pgo_bw_flush(bw); /* BufWriter drop flushes */
    free(bw);
    close(ofd);
    close(pgen);
    return rc;
}

/* ---- synthetic inputs (not from the reference) ---- */
uint64_t pgo_splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* record bytes of variant v = little-endian words splitmix64(seed + (v << 20) + word_idx),
 * truncated to R = ceil(N/4); pad bits of the last byte zeroed unless dirty_pad. */
void pgo_synth_records(uint8_t *dst, uint64_t record_stride, uint32_t num_samples,
                       uint64_t first_variant, uint32_t n_variants,
                       uint64_t seed, int dirty_pad)
{
    uint32_t R = pgo_variant_record_size(num_samples);
    for (uint32_t j = 0; j < n_variants; j++) {
        uint64_t v = first_variant + j;
        uint8_t *rec = dst + (uint64_t)j * record_stride;
        for (uint32_t b = 0; b < R; b++) {
            uint64_t w = pgo_splitmix64(seed + (v << 20) + (uint64_t)(b >> 3));
            rec[b] = (uint8_t)(w >> (8 * (b & 7)));
        }
        if (!dirty_pad && (num_samples % 4u) != 0u && R > 0) {
            uint32_t used_bits = (num_samples % 4u) * 2u;
            rec[R - 1] &= (uint8_t)((1u << used_bits) - 1u);
        }
    }
}

/* SURVEY.md §8d "hwe" value distribution (not from the reference: pgen-rs has no generator; real genotype data is
 * mostly 0/0, and write bandwidth on MI355X is data dependent, so the bench needs it).  Integer-only so that the
 * device twin is bit-exact:
 *   variant v:  p16 = 655 + splitmix64((seed ^ "MAF") + v) % 32113          allele frequency p = p16 / 65536 in [0.01, 0.5)
 *               kv  = splitmix64((seed ^ "HWE") + v)
 *   sample s:   h = splitmix64(kv + s);  two independent allele draws a = (h & 0xFFFF) < p16, b = ((h >> 16) & 0xFFFF) < p16
 *               code = a + b  (0 = 0/0, 1 = 0/1, 2 = 1/1: Hardy-Weinberg proportions q^2, 2pq, p^2)
 *               missing (code 3) when the top 32 bits of h are < 4294967 (0.1 %).
 * Pad bits of the last byte are zero. */
#define PGO_SEED_MAF 0x4D4146ull
#define PGO_SEED_HWE 0x485745ull
void pgo_synth_records_hwe(uint8_t *dst, uint64_t record_stride, uint32_t num_samples,
                           uint64_t first_variant, uint32_t n_variants, uint64_t seed)
{
    uint32_t R = pgo_variant_record_size(num_samples);
    for (uint32_t j = 0; j < n_variants; j++) {
        uint64_t v = first_variant + j;
        uint8_t *rec = dst + (uint64_t)j * record_stride;
        uint32_t p16 = 655u + (uint32_t)(pgo_splitmix64((seed ^ PGO_SEED_MAF) + v) % 32113ull);
        uint64_t kv = pgo_splitmix64((seed ^ PGO_SEED_HWE) + v);
        for (uint32_t b = 0; b < R; b++) rec[b] = 0;
        for (uint32_t s = 0; s < num_samples; s++) {
            uint64_t h = pgo_splitmix64(kv + (uint64_t)s);
            uint32_t code = (uint32_t)((h & 0xFFFFu) < p16) + (uint32_t)(((h >> 16) & 0xFFFFu) < p16);
            if ((uint32_t)(h >> 32) < 4294967u) code = 3u;
            rec[s >> 2] |= (uint8_t)(code << ((s & 3u) * 2u));
        }
    }
}

uint32_t pgo_synth_keep(uint32_t num_samples, uint64_t seed, uint32_t modulus,
                        uint32_t *kept_idx, uint32_t cap)
{
    uint32_t n = 0;
    for (uint32_t i = 0; i < num_samples; i++) {
        if (pgo_splitmix64(seed ^ (uint64_t)i) % modulus == 0) {
            if (n < cap) kept_idx[n] = i;
            n++;
        }
    }
    return n;
}

/* src/pfile.rs:171-190 with the record of output row j at base + record_off[j] (byte offsets instead of the
 * fixed-width offset formula of :165): the records of a variable-width file that are stored uncompressed. */
int pgo_decode_emit_at(const uint8_t *base, const uint64_t *record_off, uint32_t n_variants,
                       uint32_t num_samples, const uint32_t *kept_idx, uint32_t kept_count,
                       uint8_t *out, uint64_t out_stride)
{
    for (uint32_t j = 0; j < n_variants; j++) {
        int rc = pgo_decode_emit(base + record_off[j], 0, NULL, 1, num_samples, kept_idx, kept_count, out + (uint64_t)j * out_stride, out_stride);
        if (rc) return rc;
    }
    return 0;
}
