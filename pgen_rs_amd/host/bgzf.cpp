// bgzf.cpp — see bgzf.h.
#include "bgzf.h"

#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <cerrno>
#include <cstring>
#include <mutex>
#include <thread>

#include "pfile.h"

namespace pgenhost {

namespace {

constexpr size_t kSlot = 65536;        // a member never exceeds 64 KiB (BSIZE is 16 bits)
constexpr size_t kHeader = 18, kTrailer = 8;

void put16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
void put32(uint8_t *p, uint32_t v) { put16(p, v); put16(p + 2, v >> 16); }

void member_header(uint8_t *p)
{
    static const uint8_t h[kHeader] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0, 0};  // BSIZE filled in later
    std::memcpy(p, h, kHeader);
}

// one member for `n` <= kPiece input bytes into `dst` (kSlot bytes); returns its total size
uint32_t deflate_piece(z_stream &zs, const uint8_t *src, size_t n, uint8_t *dst)
{
    member_header(dst);
    deflateReset(&zs);
    zs.next_in = const_cast<Bytef *>(src);
    zs.avail_in = (uInt)n;
    zs.next_out = dst + kHeader;
    zs.avail_out = (uInt)(kSlot - kHeader - kTrailer);
    size_t clen;
    if (deflate(&zs, Z_FINISH) == Z_STREAM_END) {
        clen = (kSlot - kHeader - kTrailer) - zs.avail_out;
    } else {
        // incompressible piece: one stored deflate block (5 bytes of framing; 65 280 + 5 + 26 < 65 536)
        uint8_t *q = dst + kHeader;
        q[0] = 1;  // BFINAL = 1, BTYPE = 00
        put16(q + 1, (uint32_t)n);
        put16(q + 3, ~(uint32_t)n & 0xFFFFu);
        std::memcpy(q + 5, src, n);
        clen = n + 5;
    }
    const uint32_t total = (uint32_t)(kHeader + clen + kTrailer);
    put16(dst + 16, total - 1u);
    put32(dst + kHeader + clen, (uint32_t)crc32(crc32(0L, Z_NULL, 0), src, (uInt)n));
    put32(dst + kHeader + clen + 4, (uint32_t)n);
    return total;
}

void write_all(int fd, const uint8_t *p, size_t n, const std::string &path)
{
    while (n) {
        ssize_t w = ::write(fd, p, n);
        if (w < 0) {
            if (errno == EINTR) continue;
            throw PfileError("write " + path + ": " + std::strerror(errno));
        }
        if (w == 0) throw PfileError("write " + path + ": failed to write whole buffer");
        p += w;
        n -= (size_t)w;
    }
}

}  // namespace

BgzfWriter::BgzfWriter(int fd, std::string path, int level, unsigned threads)
    : fd_(fd), path_(std::move(path)), level_(level < 1 ? 1 : (level > 9 ? 9 : level)), threads_(threads ? threads : 1u)
{
}

void BgzfWriter::write(const void *data, size_t n)
{
    if (!n) return;
    const uint8_t *src = static_cast<const uint8_t *>(data);
    const size_t pieces = (n + kPiece - 1) / kPiece;
    if (out_.size() < pieces * kSlot) out_.resize(pieces * kSlot);
    if (len_.size() < pieces) len_.resize(pieces);
    std::atomic<size_t> next{0};
    std::mutex mu;
    std::string err;
    auto work = [&] {
        z_stream zs;
        std::memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, level_, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) {  // raw deflate, 32-KiB window
            std::lock_guard<std::mutex> lk(mu);
            err = "deflateInit2 failed";
            return;
        }
        for (size_t i; (i = next.fetch_add(1)) < pieces;) {
            const size_t lo = i * kPiece, len = std::min(kPiece, n - lo);
            len_[i] = deflate_piece(zs, src + lo, len, out_.data() + i * kSlot);
        }
        deflateEnd(&zs);
    };
    const unsigned nt = (unsigned)std::min<size_t>(threads_, pieces);
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt; t++) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    if (!err.empty()) throw PfileError(path_ + ": " + err);
    // append in order; neighbouring members are gathered so that one write() carries ~1 MiB
    std::vector<uint8_t> gather;
    gather.reserve(1u << 20);
    for (size_t i = 0; i < pieces; i++) {
        gather.insert(gather.end(), out_.data() + i * kSlot, out_.data() + i * kSlot + len_[i]);
        bytes_out_ += len_[i];
        if (gather.size() >= (1u << 20) - kSlot || i + 1 == pieces) {
            write_all(fd_, gather.data(), gather.size(), path_);
            gather.clear();
        }
    }
    bytes_in_ += n;
}

void BgzfWriter::finish()
{
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    write_all(fd_, eof, sizeof eof, path_);
    bytes_out_ += sizeof eof;
}

}  // namespace pgenhost
