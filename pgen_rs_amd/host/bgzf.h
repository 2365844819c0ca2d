// bgzf.h — BGZF (blocked gzip) writer for `.vcf.gz` output (the reference git-ignores `*.vcf.gz` next to
// its data, /root/reference/.gitignore:3, and README.md:170-189 compares against bcftools on `.vcf.gz` — it writes none itself:
// FORMAT PARITY UNPINNED by the reference).  BGZF as bgzip / htslib write it (SAM spec §4.1): a series of gzip members of at
// most 64 KiB, each carrying its own compressed size in a 'BC' extra subfield, closed by the 28-byte empty EOF member.  Every
// member is independent, so a block of VCF text is cut into 65 280-byte pieces that are deflated by a pool of threads and
// appended in order.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace pgenhost {

class BgzfWriter {
  public:
    // appends to `fd` at its current end (the caller owns fd); level 1..9 (zlib), threads >= 1
    BgzfWriter(int fd, std::string path, int level, unsigned threads);
    // deflate `n` bytes as ceil(n / 65 280) members, in parallel, and append them in order
    void write(const void *data, size_t n);
    // the EOF marker member; nothing may be written afterwards
    void finish();
    uint64_t bytes_in() const { return bytes_in_; }
    uint64_t bytes_out() const { return bytes_out_; }

    static constexpr size_t kPiece = 65280;   // input bytes per member (bgzip's 0xff00)

  private:
    int fd_;
    std::string path_;
    int level_;
    unsigned threads_;
    uint64_t bytes_in_ = 0, bytes_out_ = 0;
    std::vector<uint8_t> out_;   // one 64-KiB slot per piece of the current write()
    std::vector<uint32_t> len_;
};

}  // namespace pgenhost
