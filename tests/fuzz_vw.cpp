// fuzz_vw.cpp — mutation fuzzer for the variable-width header / offset-table walk (pgenhip_vw_*), built by
// tests/test_sanitizers.py with g++ -fsanitize=address,undefined together with pgen_rs_amd/csrc/host_pure.cpp (the same
// source libpgen_hip.so is built from).  These functions parse bytes that come straight from a file: whatever the bytes, they
// must return a status, never read outside `index` and never write outside the caller's arrays.
//   usage: fuzz_vw <seed file .pgen> <iterations> <rng seed>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>

#include "../include/pgen_hip.h"

static uint64_t rng_state;
static uint64_t rnd()
{
    rng_state += 0x9E3779B97F4A7C15ull;
    uint64_t z = rng_state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    const std::vector<uint8_t> seed((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (seed.size() < 12) return 2;
    const long iters = std::atol(argv[2]);
    rng_state = std::strtoull(argv[3], nullptr, 10);
    long ok = 0, bad_header = 0, bad_index = 0, compressed = 0, skipped = 0, struct_refused = 0;
    for (long it = 0; it < iters; it++) {
        std::vector<uint8_t> d = seed;
        const int n_mut = 1 + (int)(rnd() % 4);
        for (int m = 0; m < n_mut; m++) {
            const uint64_t r = rnd();
            const size_t span = (r & 1) ? std::min<size_t>(d.size(), 64) : d.size();  // half of the mutations hit the header and the first tables
            const size_t pos = (size_t)((r >> 8) % span);
            switch ((r >> 1) & 3) {
                case 0: d[pos] ^= (uint8_t)(1u << ((r >> 40) & 7)); break;
                case 1: d[pos] = (uint8_t)(r >> 48); break;
                case 2: d[pos] = 0xFF; break;
                default: d[pos] = 0x00; break;
            }
        }
        if (rnd() % 8 == 0) d.resize(12 + (size_t)(rnd() % (d.size() - 11)));  // truncation
        pgenhip_vw_header h;
        int rc = pgenhip_vw_parse_header(d.data(), &h);
        if (rc != PGENHIP_OK) {
            bad_header++;
            continue;
        }
        if (h.variant_count > 2000000u) {  // the caller's arrays are variant_count long: a real host bounds this by the file size first
            skipped++;
            continue;
        }
        // exactly what a host has: the bytes between the header and where the header says the records start (if the file is that long)
        const uint64_t want = h.variant_records_offset - 12;
        const uint64_t have = d.size() - 12;
        const uint64_t index_len = std::min(want, have);
        std::vector<uint8_t> index(d.begin() + 12, d.begin() + 12 + (ptrdiff_t)index_len);  // own allocation: ASan sees any over-read
        std::vector<uint8_t> types(h.variant_count);
        std::vector<uint32_t> lens(h.variant_count);
        std::vector<uint64_t> offs(h.variant_count);
        // the struct is part of the public ABI and a caller may hand in one it made itself: perturb the DERIVED fields of a copy
        // (the arrays stay variant_count long).  The walk must refuse what does not follow from variant_count and the widths
        // — a record_length_bytes > 8 would shift by >= 64, a wrong block_count would walk past the index — or walk clean.
        if (it % 3 == 0) {
            pgenhip_vw_header g = h;
            const uint64_t r = rnd();
            switch (r & 7) {
                case 0: g.record_length_bytes = (uint8_t)(r >> 8); break;
                case 1: g.record_type_bits = (uint8_t)(r >> 8); break;
                case 2: g.block_count = (r >> 8) % 70000u; break;
                case 3: g.block_count = r; break;
                case 4: g.variant_records_offset = (r >> 8) % (2 * h.variant_records_offset + 64); break;
                case 5: g.variant_records_offset = r; break;
                case 6: g.main_header_body_offset = r; break;  // not used by the walk: must not matter
                default: g.record_length_bytes = 0; break;
            }
            const int grc = pgenhip_vw_walk_index(&g, index.data(), index.size(), types.data(), lens.data(), offs.data());
            const bool same = g.record_length_bytes == h.record_length_bytes && g.record_type_bits == h.record_type_bits && g.block_count == h.block_count &&
                              g.variant_records_offset == h.variant_records_offset;
            if (!same && h.variant_count && grc == PGENHIP_OK) {
                std::fprintf(stderr, "iteration %ld: an inconsistent header struct was walked\n", it);
                return 1;
            }
            if (grc == PGENHIP_ERR_BAD_ARG) struct_refused++;
        }
        rc = pgenhip_vw_walk_index(&h, index.data(), index.size(), types.data(), lens.data(), offs.data());
        if (rc != PGENHIP_OK) {
            bad_index++;
            continue;
        }
        // a walk that succeeded yields ascending, non-overlapping records behind the tables
        uint64_t prev_end = h.variant_records_offset;
        for (uint32_t v = 0; v < h.variant_count; v++) {
            if (offs[v] < prev_end) {
                std::fprintf(stderr, "iteration %ld: record %u at %llu overlaps the one before (ends %llu)\n", it, v, (unsigned long long)offs[v], (unsigned long long)prev_end);
                return 1;
            }
            prev_end = offs[v] + lens[v];
        }
        std::vector<uint64_t> sel(h.variant_count);
        const uint32_t r_bytes = pgenhip_variant_record_size(h.sample_count);
        rc = pgenhip_vw_select_uncompressed(types.data(), lens.data(), offs.data(), h.variant_count, nullptr, h.variant_count, r_bytes, sel.data());
        if (rc == PGENHIP_ERR_COMPRESSED_RECORD) compressed++;
        else if (rc == PGENHIP_OK) ok++;
        else return 1;
    }
    std::printf("fuzz_vw: %ld iterations: %ld walked clean, %ld compressed records, %ld bad header, %ld bad index, %ld skipped, %ld perturbed structs refused\n", iters, ok, compressed, bad_header,
                bad_index, skipped, struct_refused);
    return 0;
}
