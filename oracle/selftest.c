/* selftest.c — exercises the oracle under AddressSanitizer + UBSan (make -C oracle check).
 * CPU-only; part of the test infrastructure. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pgen_oracle.h"

#define CHECK(cond)                                                         \
    do {                                                                    \
        if (!(cond)) {                                                      \
            fprintf(stderr, "selftest failed at line %d: %s\n", __LINE__, #cond); \
            return 1;                                                       \
        }                                                                   \
    } while (0)

int main(void)
{
    /* truth table (src/pfile.rs:172-183) */
    uint8_t rec = 0xE4, out[17];
    CHECK(pgo_decode_emit(&rec, 1, NULL, 1, 4, NULL, 0, out, 17) == 0);
    CHECK(memcmp(out, "\t0/0\t0/1\t1/1\t./.\n", 17) == 0);
    /* exact-size buffers for every N in 0..70 with a ragged keep list: any overrun trips ASan */
    for (uint32_t n = 0; n <= 70; n++) {
        uint32_t r = pgo_variant_record_size(n);
        uint32_t v = 3;
        uint8_t *recs = malloc((size_t)v * r + 1);
        pgo_synth_records(recs, r, n, 5, v, 42, 1);
        uint32_t *kept = malloc(sizeof(uint32_t) * (n + 1));
        uint32_t k = pgo_synth_keep(n, 7, 3, kept, n);
        uint8_t *o = malloc((size_t)v * (4u * k + 1u) + 1);
        CHECK(pgo_decode_emit(recs, r, NULL, v, n, kept, k, o, 4u * k + 1u) == 0);
        for (uint32_t j = 0; j < v; j++) CHECK(o[(size_t)j * (4u * k + 1u) + 4u * k] == '\n');
        uint8_t *o2 = malloc((size_t)v * (4u * n + 1u) + 1);
        CHECK(pgo_decode_emit(recs, r, NULL, v, n, NULL, 0, o2, 4u * n + 1u) == 0);
        free(o2);
        free(o);
        free(kept);
        free(recs);
    }
    /* out-of-range kept index is reported, not read */
    uint32_t bad = 8;
    uint8_t two[2] = {0, 0}, o9[9];
    CHECK(pgo_decode_emit(two, 2, NULL, 1, 8, &bad, 1, o9, 5) == -1);
    /* header + offsets */
    uint8_t hdr[12] = {0x6C, 0x1B, 0x02, 1, 0, 0, 0, 2, 0, 0, 0, 0x40};
    uint32_t nv, ns;
    CHECK(pgo_parse_header(hdr, &nv, &ns) == 0 && nv == 1 && ns == 2);
    CHECK(pgo_record_offset_ref_u32_wrap(34360, 125000) != pgo_record_offset_exact(34360, 125000));
    puts("oracle selftest ok");
    return 0;
}
