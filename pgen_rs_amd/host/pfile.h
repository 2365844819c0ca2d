// pfile.h — C++ host restatement of pgen-rs's `Pfile` (src/pfile.rs:19-336) above the C ABI of
// include/pgen_hip.h.  Same names and argument meaning as the reference so a pgen-rs user finds
// what they expect; the per-variant decode/emit body (src/pfile.rs:165-190) is NOT here — it runs
// on the GPU through pgenhip_emit_lines.  The reference is compiled Rust and no Rust toolchain
// exists in this image, hence C++.
#pragma once
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "csvlite.h"

namespace pgenhost {

// What the reference expresses as panic!/unwrap()/assert! (exit status 101).
struct PfileError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct OutputOptions {
    int n_gpus = 1;                         // variant ranges shard over devices 0..n_gpus-1 (no collective)
    int n_shards = 0;                       // 0 = one shard per GPU; > 0: that many variant ranges, dealt round-robin over the GPUs
    uint64_t block_text_bytes = 128ull << 20;  // VCF bytes produced (and record bytes read) per block and device: the unit of staging, D2H copy and file write.  End to end on the chr22 shape
                                               // (tmpfs): 512 / 256 / 128 / 64 MiB -> 2.87 / 2.56 / 2.47 / 2.39 s keeping everybody, 1.03 / 0.76 / 0.65 / 0.65 s
                                               // keeping 20 samples: pinning the staging buffers costs more than bigger launches give (profiles/r02_e2e.md)
    uint64_t launch_bytes = 2048ull << 20;     // text (or record) bytes one kernel launch covers: several consecutive blocks, staged and copied out block by block
    int write_threads = 1;                  // parallel pwrite()s per block and device
    int read_threads = 4;                   // parallel pread()s of one run of consecutive records (runs of >= 64 MiB)
    bool bgzf = false;                      // write BGZF (`.vcf.gz`) instead of plain text
    int bgzf_level = 6;                     // zlib level of the BGZF members (bgzip's default)
    int compress_threads = 0;               // deflate threads per device shard (0 = the host's cores, at most 32)
    int filter_threads = 0;                 // pieces the metadata walk is cut into (0 = by file size and host cores, 1 = serial like the reference)
    bool verbose = false;
};

struct OutputStats {
    uint64_t variants = 0, samples_kept = 0, header_bytes = 0, body_bytes = 0;   // header / body: bytes of VCF text
    uint64_t file_bytes = 0;                                                     // what the output file holds (BGZF: compressed)
    double seconds_filter = 0, seconds_body = 0, seconds_kernel = 0;
    double seconds_setup = 0;   // inside seconds_body: HIP runtime start, contexts, device and pinned allocations of the slowest shard, before its first block is staged
};

class Pfile {
  public:
    std::string pfile_prefix;  // src/pfile.rs:20-22
    uint32_t num_variants = 0;
    uint32_t num_samples = 0;
    // Variable-width storage modes (the reference refuses them at src/pfile.rs:53 and only validates
    // their tables in the dead `Pgen` type, src/pgen.rs): per-variant record type / length / file offset from the
    // header walk (pgenhip_vw_walk_index).  Empty for a fixed-width (0x02) file.
    uint8_t storage_mode = 0x02;
    std::shared_ptr<const std::vector<uint8_t>> vw_record_type;
    std::shared_ptr<const std::vector<uint32_t>> vw_record_len;
    std::shared_ptr<const std::vector<uint64_t>> vw_record_off;
    bool variable_width() const { return static_cast<bool>(vw_record_off); }
    // file offset of variant var_idx's record: src/pfile.rs:165 (u64), or the walked table
    uint64_t record_offset(uint64_t var_idx) const;

    std::string pgen_path() const { return pfile_prefix + ".pgen"; }  // :26-36
    std::string psam_path() const { return pfile_prefix + ".psam"; }
    std::string pvar_path() const { return pfile_prefix + ".pvar"; }

    // :38-76 — opens PREFIX.pgen and checks magic / storage mode 0x02 / flag byte 0x40; storage mode 0x10 goes through the
    // variable-width header walk (src/pgen.rs:21-258) and is accepted when the file holds its tables and they are sound;
    // every other mode is refused like the reference does (:53)
    static Pfile from_prefix(const std::string &pfile_prefix);

    // :196-200
    uint32_t variant_record_size() const;

    // :202-220 — (all leading '#' lines but the last, the last one = column names line)
    std::pair<std::string, std::string> read_pvar_header() const;

    // :248-268 — byte offset just after the '#' of the column-header line
    static uint64_t find_metadata_file_header_start(const std::string &file_contents);

    using IdxRecords = std::vector<std::pair<size_t, StringRecord>>;
    // :312-335 — rows (file order) whose predicate is true; all rows without a query
    // (filter_threads: 0 = as many pieces as the file size and the host's cores suggest, 1 = the reference's serial walk)
    static IdxRecords filter_metadata(TsvReader &reader, const std::optional<std::string> &query, int filter_threads = 0);

    // :78-102 — prints f_string evaluated on each kept row to `out` (stdout in the CLI)
    static void query_metadata(TsvReader &reader, const std::optional<std::string> &query, const std::string &f_string,
                               std::string &out);

    // :104-194 — header on the host, body lines assembled on the GPU(s)
    OutputStats output_vcf(const std::optional<std::string> &sam_query, const std::optional<std::string> &var_query,
                           const std::string &filename, const OutputOptions &opt = OutputOptions()) const;

    // the header part of output_vcf (:110-146) on its own: used by output_vcf and by the CPU tests
    std::string vcf_header(const IdxRecords &sam_idx_rcs, const StringRecord &sam_header) const;
};

// Synthetic PREFIX.{pgen,pvar,psam}: pvar rows "22\t{16050000+7i}\tsnp{i}\tA\tG\t100\tPASS\t.",
// psam "#IID\tSEX\tKEEP" with "S{i:06d}\tNA\t{keep}", records from pgenhip_synth_records on device 0.
void synth_pfile(const std::string &prefix, uint32_t variants, uint32_t samples, uint32_t keep_modulus, uint64_t seed);

// whole-file read helper (metadata files are read once; the reference streams them through BufReader)
std::string read_file(const std::string &path);

}  // namespace pgenhost
