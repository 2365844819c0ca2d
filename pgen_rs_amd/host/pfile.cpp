// pfile.cpp — see pfile.h.  Line references are to /root/reference/src/pfile.rs.
#include "pfile.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <fstream>
#include <mutex>
#include <sstream>
#include <thread>

#include "../../include/pgen_hip.h"
#include "bgzf.h"
#include "expr.h"

namespace pgenhost {

namespace {

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

[[noreturn]] void fail_status(int rc, const std::string &what)
{
    std::string msg = what + ": " + pgenhip_strerror(rc);
    const char *detail = pgenhip_last_error_detail();
    if (detail && *detail) msg += std::string(" (") + detail + ")";
    throw PfileError(msg);
}

void check(int rc, const char *what)
{
    if (rc != PGENHIP_OK) fail_status(rc, what);
}

// std::io::BufRead::read_line: up to and including '\n'; empty at EOF
std::string read_line(const std::string &data, size_t &pos)
{
    if (pos >= data.size()) return std::string();
    size_t nl = data.find('\n', pos);
    size_t end = nl == std::string::npos ? data.size() : nl + 1;
    std::string line = data.substr(pos, end - pos);
    pos = end;
    return line;
}

// str::trim() of Rust for the ASCII whitespace that can occur here
std::string trim(const std::string &s)
{
    size_t b = 0, e = s.size();
    auto ws = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; };
    while (b < e && ws(s[b])) b++;
    while (e > b && ws(s[e - 1])) e--;
    return s.substr(b, e - b);
}

std::string hex2(uint8_t b)
{
    static const char *d = "0123456789abcdef";
    return std::string(1, d[b >> 4]) + d[b & 15];
}

void pread_exact(int fd, void *dst, size_t bytes, uint64_t offset, const std::string &path)
{
    uint8_t *p = static_cast<uint8_t *>(dst);
    while (bytes) {
        ssize_t r = pread(fd, p, bytes, (off_t)offset);
        if (r < 0) {
            if (errno == EINTR) continue;
            throw PfileError("read " + path + ": " + std::strerror(errno));
        }
        if (r == 0) throw PfileError("read " + path + ": failed to fill whole buffer (record past end of file)");  // :170 read_exact
        p += r;
        bytes -= (size_t)r;
        offset += (uint64_t)r;
    }
}

void pwrite_exact(int fd, const void *src, size_t bytes, uint64_t offset, const std::string &path)
{
    const uint8_t *p = static_cast<const uint8_t *>(src);
    while (bytes) {
        ssize_t w = pwrite(fd, p, bytes, (off_t)offset);
        if (w < 0) {
            if (errno == EINTR) continue;
            throw PfileError("write " + path + ": " + std::strerror(errno));
        }
        if (w == 0) throw PfileError("write " + path + ": failed to write whole buffer");  // BufWriter::write_all -> WriteZero
        p += w;
        bytes -= (size_t)w;
        offset += (uint64_t)w;
    }
}

}  // namespace

std::string read_file(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw PfileError("open " + path + ": " + std::strerror(errno));  // File::open(..).unwrap()
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

// :38-76
Pfile Pfile::from_prefix(const std::string &pfile_prefix)
{
    Pfile pf;
    pf.pfile_prefix = pfile_prefix;
    const std::string path = pf.pgen_path();
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) throw PfileError("open " + path + ": " + std::strerror(errno));  // :41
    struct FdCloser {
        int fd;
        ~FdCloser() { close(fd); }
    } closer{fd};
    uint8_t hdr[12];
    ssize_t got = pread(fd, hdr, sizeof hdr, 0);
    if (got != (ssize_t)sizeof hdr) throw PfileError("read " + path + ": failed to fill whole buffer");  // :45 read_exact
    if (hdr[0] == 0x6C && hdr[1] == 0x1B && hdr[2] == 0x10) {
        // the standard variable-width mode: the header walk of src/pgen.rs:21-258.  (Every other mode byte — 0x01 .bed, 0x03 / 0x04
        // fixed-width dosage, 0x20 / 0x21 — keeps the reference's refusal below, :53: their bytes 3.. are not this header.)
        pgenhip_vw_header vh;
        int vrc = pgenhip_vw_parse_header(hdr, &vh);
        if (vrc == PGENHIP_ERR_BAD_FLAGS)
            throw PfileError(path + ": storage mode 0x" + hex2(hdr[2]) + ": unsupported header format byte (" + pgenhip_last_error_detail() + ")");  // src/pgen.rs:58, :64
        check(vrc, "pgenhip_vw_parse_header");
        // the header's counts are 32 untrusted bits: nothing is allocated from them before the file has shown that it holds the tables
        struct stat sb;
        if (fstat(fd, &sb) != 0) throw PfileError("stat " + path + ": " + std::strerror(errno));
        const uint64_t file_size = (uint64_t)sb.st_size;
        if (vh.variant_records_offset > file_size)
            throw PfileError(path + ": failed to fill whole buffer (the header promises " + std::to_string(vh.variant_records_offset) +
                             " bytes of offset and type / length tables, the file has " + std::to_string(file_size) + ")");  // src/pgen.rs:147, :219 read_exact
        std::vector<uint8_t> index((size_t)(vh.variant_records_offset - 12ull));
        pread_exact(fd, index.data(), index.size(), 12, path);
        auto types = std::make_shared<std::vector<uint8_t>>(vh.variant_count);
        auto lens = std::make_shared<std::vector<uint32_t>>(vh.variant_count);
        auto offs = std::make_shared<std::vector<uint64_t>>(vh.variant_count);
        vrc = pgenhip_vw_walk_index(&vh, index.data(), index.size(), types->data(), lens->data(), offs->data());
        if (vrc == PGENHIP_ERR_BAD_INDEX) throw PfileError(path + ": " + pgenhip_last_error_detail());  // src/pgen.rs:160-165 panic
        check(vrc, "pgenhip_vw_walk_index");
        // records ascend and do not overlap (the walk checked that), so the last one bounds them all
        if (vh.variant_count && offs->back() + (uint64_t)lens->back() > file_size)
            throw PfileError(path + ": variant records run past the end of the file (" + std::to_string(offs->back() + (uint64_t)lens->back()) + " > " +
                             std::to_string(file_size) + ")");
        pf.storage_mode = hdr[2];
        pf.num_variants = vh.variant_count;
        pf.num_samples = vh.sample_count;
        pf.vw_record_type = types;
        pf.vw_record_len = lens;
        pf.vw_record_off = offs;
        return pf;
    }
    int rc = pgenhip_parse_header(hdr, &pf.num_variants, &pf.num_samples);
    if (rc == PGENHIP_ERR_BAD_MAGIC) throw PfileError(path + ": assertion failed: magic number is not [0x6C, 0x1B]");          // :47
    if (rc == PGENHIP_ERR_BAD_MODE) throw PfileError(path + ": assertion failed: storage_mode == 0x02 (only the fixed-width mode is supported)");  // :53
    if (rc == PGENHIP_ERR_BAD_FLAGS) throw PfileError(path + ": assertion failed: header byte 11 is not 0x40");                // :69
    check(rc, "pgenhip_parse_header");
    return pf;
}

// :196-200
uint32_t Pfile::variant_record_size() const { return pgenhip_variant_record_size(num_samples); }

uint64_t Pfile::record_offset(uint64_t var_idx) const
{
    if (variable_width()) return (*vw_record_off)[(size_t)var_idx];
    return pgenhip_record_offset(var_idx, variant_record_size());  // :165, widened before the multiply
}

// :202-220
std::pair<std::string, std::string> Pfile::read_pvar_header() const
{
    const std::string data = read_file(pvar_path());
    std::vector<std::string> header_lines;
    size_t pos = 0;
    for (;;) {
        std::string buf = read_line(data, pos);
        if (!buf.empty() && buf[0] == '#')
            header_lines.push_back(buf);
        else
            break;
    }
    if (header_lines.empty()) throw PfileError(pvar_path() + ": no '#' header line (called `Option::unwrap()` on a `None` value)");  // :217
    std::string header = header_lines.back();
    header_lines.pop_back();
    std::string joined;
    for (const auto &l : header_lines) joined += l;
    return {joined, header};
}

// :248-268
uint64_t Pfile::find_metadata_file_header_start(const std::string &data)
{
    size_t pos = 0;
    std::string prev_buf, buf;
    for (;;) {
        prev_buf = buf;
        buf = read_line(data, pos);
        if (buf.empty() || buf[0] != '#') {
            const uint64_t current_pos = pos;
            const uint64_t offset = (uint64_t)(buf.size() + prev_buf.size()) - 1ull;  // wraps like Rust would panic only in debug
            return current_pos - offset;
        }
    }
}

// :312-335
namespace {

// src/pfile.rs:312-335 for the records of one reader, indices starting at `first_idx`.
Pfile::IdxRecords filter_records(TsvReader &reader, const std::optional<std::string> &query, size_t first_idx)
{
    Pfile::IdxRecords kept;
    std::optional<Expr> expr;
    if (query) {
        expr.emplace(*query);
        expr->bind(reader.headers());
    }
    StringRecord rcd;
    size_t idx = first_idx;
    while (reader.next(rcd)) {
        const bool keep = expr ? expr->eval_boolean(rcd) : true;  // :321-329
        if (keep) kept.emplace_back(idx, rcd);                    // :330-332
        idx++;
    }
    return kept;
}

}  // namespace

// N2: the reference walks the metadata file on one thread (2.7 s of its chr22 runs,
// README.md:164-168).  Records are independent, so a big file without quotes — a quoted field may
// hold a line break, which would make a split point ambiguous — is cut at line ends into one piece
// per thread; every piece is parsed and filtered like the whole (same header row, same expression),
// and the kept records are concatenated in file order with their indices shifted by the number of
// records before the piece.  Any error re-runs the serial walk so that the message (record and line
// numbers) is the one the serial reader gives.
Pfile::IdxRecords Pfile::filter_metadata(TsvReader &reader, const std::optional<std::string> &query, int filter_threads)
{
    const std::string &data = reader.data();
    const size_t begin = reader.position(), end = reader.end_position();
    const size_t kMinBytesPerThread = 1u << 20;
    size_t n_threads = std::min<size_t>({(size_t)std::max(1u, std::thread::hardware_concurrency()), 16u, (end - begin) / kMinBytesPerThread});
    if (filter_threads > 0) n_threads = (size_t)filter_threads;  // `--filter-threads` (1 = the serial walk of the reference)
    if (n_threads < 2 || begin >= end || memchr(data.data() + begin, '"', end - begin) != nullptr)
        return filter_records(reader, query, 0);

    // piece boundaries: just behind a '\n' (files that end records with a lone '\r' stay in one piece)
    std::vector<size_t> cuts{begin};
    for (size_t t = 1; t < n_threads; t++) {
        const size_t want = begin + (end - begin) / n_threads * t;
        if (want <= cuts.back()) continue;
        const void *nl = memchr(data.data() + want, '\n', end - want);
        if (!nl) break;
        const size_t cut = (size_t)(static_cast<const char *>(nl) - data.data()) + 1u;
        if (cut > cuts.back() && cut < end) cuts.push_back(cut);
    }
    cuts.push_back(end);
    const size_t n_pieces = cuts.size() - 1u;
    if (n_pieces < 2) return filter_records(reader, query, 0);

    std::vector<IdxRecords> kept(n_pieces);
    std::vector<size_t> n_records(n_pieces, 0);
    std::vector<char> failed(n_pieces, 0);
    std::vector<std::thread> workers;
    for (size_t t = 0; t < n_pieces; t++) {
        workers.emplace_back([&, t] {
            try {
                TsvReader piece(reader, cuts[t], cuts[t + 1u]);
                kept[t] = filter_records(piece, query, 0);  // indices local to the piece
                n_records[t] = piece.records_read();
            } catch (...) {
                failed[t] = 1;
            }
        });
    }
    for (auto &w : workers) w.join();
    if (std::find(failed.begin(), failed.end(), (char)1) != failed.end()) return filter_records(reader, query, 0);  // throws the serial error

    size_t total = 0;
    for (const auto &k : kept) total += k.size();
    IdxRecords all;
    all.reserve(total);
    size_t base = 0;
    for (size_t t = 0; t < n_pieces; t++) {
        for (auto &kv : kept[t]) all.emplace_back(kv.first + base, std::move(kv.second));
        base += n_records[t];
    }
    return all;
}

void Pfile::query_metadata(TsvReader &reader, const std::optional<std::string> &query, const std::string &f_string, std::string &out)
{
    std::optional<Expr> filter;
    if (query) {
        filter.emplace(*query);
        filter->bind(reader.headers());
    }
    Expr fmt(f_string);
    fmt.bind(reader.headers());
    StringRecord rcd;
    while (reader.next(rcd)) {
        const bool keep = filter ? filter->eval_boolean(rcd) : true;  // :93-95
        if (keep) {
            out += fmt.eval_string(rcd);  // :97
            out += '\n';                  // println!
        }
    }
}

// :110-146 minus the file handling
std::string Pfile::vcf_header(const IdxRecords &sam_idx_rcs, const StringRecord &sam_header) const
{
    auto [pvar_header, pvar_column_names] = read_pvar_header();  // :110
    size_t iid = sam_header.size();
    for (size_t c = 0; c < sam_header.size(); c++) {  // :114-124 find_map: first match
        if (sam_header[c] == "IID") {
            iid = c;
            break;
        }
    }
    if (iid == sam_header.size()) throw PfileError("IID not among the headers of " + psam_path());  // :125-126
    std::string sam_ids;  // :130-134
    for (size_t k = 0; k < sam_idx_rcs.size(); k++) {
        if (k) sam_ids += '\t';
        sam_ids += sam_idx_rcs[k].second.at(iid);
    }
    std::string h = "##fileformat=VCFv4.2\n##source=pgen-rs\n";  // :139-140
    h += pvar_header;                                              // :141
    h += trim(pvar_column_names);                                  // :144-145
    h += "\tFORMAT\t" + sam_ids + "\n";                            // :146
    return h;
}

namespace {

struct DeviceBuffers {
    pgenhip_ctx *ctx = nullptr;
    void *h_rec = nullptr, *h_blob = nullptr, *h_off = nullptr, *h_text = nullptr, *h_text2 = nullptr;
    void *d_rec = nullptr, *d_blob = nullptr, *d_off = nullptr, *d_text = nullptr;
    ~DeviceBuffers()
    {
        if (!ctx) return;
        pgenhip_host_free_pinned(ctx, h_rec);
        pgenhip_host_free_pinned(ctx, h_blob);
        pgenhip_host_free_pinned(ctx, h_off);
        pgenhip_host_free_pinned(ctx, h_text);
        pgenhip_host_free_pinned(ctx, h_text2);
        pgenhip_device_free(ctx, d_rec);
        pgenhip_device_free(ctx, d_blob);
        pgenhip_device_free(ctx, d_off);
        pgenhip_device_free(ctx, d_text);
        pgenhip_destroy(ctx);
    }
};

}  // namespace

// :104-194
OutputStats Pfile::output_vcf(const std::optional<std::string> &sam_query, const std::optional<std::string> &var_query,
                              const std::string &filename, const OutputOptions &opt) const
{
    OutputStats st;
    const double t0 = now_s();
    // The HIP runtime's first call (device discovery, loading the code object) costs 30-80 ms: let it run beside the metadata walk
    // instead of in front of the first block.  A throw-away ctx on every device this run will use; errors are left to the real
    // creates below, which report them.
    std::thread hip_warm_up([n = std::max(1, opt.n_gpus)] {
        int n_dev = 0;
        if (pgenhip_device_count(&n_dev) != PGENHIP_OK) return;
        for (int d = 0; d < std::min(n, n_dev); d++) {
            pgenhip_ctx *c = nullptr;
            if (pgenhip_create(&c, d, 4, nullptr, 0, 0) == PGENHIP_OK) pgenhip_destroy(c);
        }
    });
    struct WarmUpJoin {
        std::thread &t;
        ~WarmUpJoin()
        {
            if (t.joinable()) t.join();
        }
    } warm_up_join{hip_warm_up};
    const std::string psam = read_file(psam_path());  // :111
    TsvReader psam_reader(psam, find_metadata_file_header_start(psam));
    const StringRecord sam_header = psam_reader.headers();  // :112
    const std::string pvar = read_file(pvar_path());
    TsvReader pvar_reader(pvar, find_metadata_file_header_start(pvar));
    const IdxRecords var_idx_rcds = filter_metadata(pvar_reader, var_query, opt.filter_threads);  // :127
    const IdxRecords sam_idx_rcs = filter_metadata(psam_reader, sam_query, opt.filter_threads);   // :128
    const std::string header = vcf_header(sam_idx_rcs, sam_header);
    st.seconds_filter = now_s() - t0;

    int fd = open(filename.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);  // :136 File::create
    if (fd < 0) throw PfileError("create " + filename + ": " + std::strerror(errno));
    struct FdGuard {
        int fd;
        ~FdGuard()
        {
            if (fd >= 0) close(fd);
        }
    } guard{fd};
    // a short or failed close (ENOSPC/EIO surfacing late) must not leave a truncated VCF behind an exit code 0
    auto close_checked = [&] {
        const int cfd = guard.fd;
        guard.fd = -1;
        if (close(cfd) != 0) throw PfileError("close " + filename + ": " + std::strerror(errno));
    };
    const unsigned n_compress = (unsigned)(opt.compress_threads > 0 ? opt.compress_threads : std::min(32u, std::max(1u, std::thread::hardware_concurrency())));
    if (opt.bgzf) {
        BgzfWriter hw(fd, filename, opt.bgzf_level, 1);
        hw.write(header.data(), header.size());   // its own BGZF member(s); the body's members follow
        st.file_bytes = hw.bytes_out();
    } else {
        pwrite_exact(fd, header.data(), header.size(), 0, filename);  // :139-146
    }
    // BGZF: every member is independent, so the shards' streams simply follow each other; shard 0 appends to the file itself, the
    // others to temporary files that are appended in shard order at the end (one device: no temporary file)
    auto shard_tmp = [&](int g) { return filename + ".shard" + std::to_string(g) + ".tmp"; };
    auto finish_bgzf = [&](int n_shards) {
        std::vector<uint8_t> buf(8u << 20);
        for (int g = 1; g < n_shards; g++) {
            const std::string tmp = shard_tmp(g);
            int tfd = open(tmp.c_str(), O_RDONLY);
            if (tfd < 0) continue;   // a shard without variants wrote nothing
            for (;;) {
                ssize_t r = read(tfd, buf.data(), buf.size());
                if (r < 0 && errno == EINTR) continue;
                if (r < 0) { close(tfd); throw PfileError("read " + tmp + ": " + std::strerror(errno)); }
                if (r == 0) break;
                const uint8_t *p = buf.data();
                for (size_t left = (size_t)r; left;) {
                    ssize_t w = write(fd, p, left);
                    if (w < 0 && errno == EINTR) continue;
                    if (w <= 0) { close(tfd); throw PfileError("write " + filename + ": " + std::strerror(errno)); }
                    p += w;
                    left -= (size_t)w;
                }
                st.file_bytes += (uint64_t)r;
            }
            close(tfd);
            unlink(tmp.c_str());
        }
        BgzfWriter ew(fd, filename, opt.bgzf_level, 1);
        ew.finish();
        st.file_bytes += ew.bytes_out();
    };

    // ---- geometry of the body (:156-192): line j = prefix_j + K x "\tA/B" + "\n"
    const uint32_t N = num_samples;
    const uint32_t R = variant_record_size();
    const size_t V = var_idx_rcds.size();
    std::vector<uint32_t> kept;
    kept.reserve(sam_idx_rcs.size());
    for (const auto &ir : sam_idx_rcs) {
        if (ir.first >= N) throw PfileError("index out of bounds: sample row " + std::to_string(ir.first) + " but the .pgen holds " + std::to_string(N) + " samples");  // :173
        kept.push_back((uint32_t)ir.first);
    }
    const bool all_samples = kept.size() == (size_t)N;  // every row kept: the K = N fast path
    const uint64_t K = kept.size();
    std::vector<uint64_t> file_off(V + 1, 0);  // body-relative offset of each line
    uint64_t max_prefix = 0;
    for (size_t j = 0; j < V; j++) {
        uint64_t plen = 2;  // "GT" (:161)
        for (const auto &col : var_idx_rcds[j].second) plen += col.size() + 1;  // col + '\t' (:157-160)
        max_prefix = std::max(max_prefix, plen);
        file_off[j + 1] = file_off[j] + plen + 4ull * K + 1ull;
        if (var_idx_rcds[j].first >= num_variants)
            throw PfileError("variant row " + std::to_string(var_idx_rcds[j].first) + " is past the " + std::to_string(num_variants) + " records of " + pgen_path());
        if (variable_width()) {
            // only records stored as plain 2-bit hard calls (type 0, length R) can take the path of :165-190; anything else is reported
            const size_t vi = var_idx_rcds[j].first;
            if ((*vw_record_type)[vi] != 0 || (*vw_record_len)[vi] != R)
                throw PfileError(pgen_path() + ": variant row " + std::to_string(vi) + " is stored compressed (record type " +
                                 std::to_string((*vw_record_type)[vi]) + ", " + std::to_string((*vw_record_len)[vi]) + " bytes); only uncompressed 2-bit records are supported");
        }
    }
    st.variants = V;
    st.samples_kept = K;
    st.header_bytes = header.size();
    st.body_bytes = file_off[V];
    if (!opt.bgzf) st.file_bytes = st.header_bytes + st.body_bytes;
    if (V == 0) {
        if (opt.bgzf) finish_bgzf(0);
        close_checked();
        return st;
    }

    if (hip_warm_up.joinable()) hip_warm_up.join();
    int n_dev = 0;
    check(pgenhip_device_count(&n_dev), "pgenhip_device_count");
    if (n_dev <= 0) throw PfileError("no HIP device: the GT decode/emit path has no CPU fallback");
    const int n_use = std::max(1, std::min(opt.n_gpus, n_dev));      // devices actually used
    const int G = opt.n_shards > 0 ? opt.n_shards : n_use;            // variant ranges (shards)
    const double t_body = now_s();
    std::mutex err_mu;
    std::string err;
    std::vector<double> kernel_s((size_t)G, 0.0), setup_s((size_t)G, 0.0);
    std::vector<uint64_t> shard_out((size_t)G, 0);   // BGZF: compressed bytes per shard
    const std::string pgen = pgen_path();

    // Per device: two buffer sets, each with its own ctx/stream.  The producer (this thread) reads
    // records + builds prefixes for block k and queues H2D -> kernel -> D2H on set k%2; a consumer
    // thread waits for that set, writes its text with a few parallel pwrite()s at the precomputed
    // file offset, and hands the set back.  File staging, PCIe copies, the kernel and the file
    // writes of neighbouring blocks overlap.
    // parallel pwrite()s of one block: measured on tmpfs they only fight over the page-allocation lock
    // (8 writers: sys 9.6 s vs 2.3 s, wall unchanged), so the default is one writer per device
    const unsigned n_writers = (unsigned)std::max(1, opt.write_threads);

    auto worker = [&](int g) {
        try {
            const double t_worker = now_s();
            // contiguous range of the kept-variant list per device, sizes differ by <= 1: the one partitioner
            uint64_t begin64 = 0, end64 = 0;
            check(pgenhip_shard_range(V, (uint32_t)G, (uint32_t)g, &begin64, &end64), "pgenhip_shard_range");
            const size_t begin = (size_t)begin64, end = (size_t)end64;
            if (begin == end) return;
            int pfd = open(pgen.c_str(), O_RDONLY);  // :149 (unbuffered on purpose, :150-152)
            if (pfd < 0) throw PfileError("open " + pgen + ": " + std::strerror(errno));
            FdGuard pg{pfd};
            // BGZF: this shard's ordered, parallel deflate writer (shard 0: the output file behind the header; others: a temporary file)
            int zfd = -1;
            if (opt.bgzf) {
                zfd = g == 0 ? fd : open(shard_tmp(g).c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
                if (zfd < 0) throw PfileError("create " + shard_tmp(g) + ": " + std::strerror(errno));
            }
            FdGuard zg{g == 0 ? -1 : zfd};
            std::unique_ptr<BgzfWriter> zw;
            if (opt.bgzf) zw.reset(new BgzfWriter(zfd, g == 0 ? filename : shard_tmp(g), opt.bgzf_level, std::max(1u, n_compress / (unsigned)std::max(1, std::min(G, n_use)))));
            // variants per block: bounded by the text budget — and by the same budget of RECORD bytes, so that a run that keeps few
            // samples (little text per record) still moves in several blocks and its file reads overlap the copies and the kernel
            const uint64_t max_line = max_prefix + 4ull * K + 1ull;
            const uint64_t bv = std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(opt.block_text_bytes / max_line, opt.block_text_bytes / std::max<uint32_t>(R, 1u)), end - begin));
            const size_t n_blocks = (end - begin + (size_t)bv - 1) / (size_t)bv;
            // A block is the unit of file staging, of the D2H copy and of the file write.  A LAUNCH covers several consecutive blocks
            // (up to the launch budget of text or record bytes): its records are staged block by block while the blocks of the launch
            // before it are copied out and written, so the host pipeline keeps the cadence of small blocks and the kernel sees launches
            // of up to 2 GiB (a 128-MiB launch of the chr22 shape is two work items deep: 39 us where its share of a large launch costs
            // 25 us, profiles/r03_cli_kernels.md).  Launch sizes ramp 1, 2, 4, ... blocks so the first byte leaves as early as before.
            const uint64_t blocks_per_launch_max = std::max<uint64_t>(1, std::min<uint64_t>(opt.launch_bytes / std::max<uint64_t>(1, std::max<uint64_t>(bv * max_line, bv * R)), n_blocks));
            struct LaunchPlan {
                size_t first_block, n_blocks;
            };
            std::vector<LaunchPlan> plan;
            size_t widest = 1;
            for (size_t k = 0, step = 1; k < n_blocks; step = (size_t)std::min<uint64_t>(2 * (uint64_t)step, blocks_per_launch_max)) {
                const size_t n = (size_t)std::min<uint64_t>(std::min<uint64_t>(step, blocks_per_launch_max), n_blocks - k);
                plan.push_back(LaunchPlan{k, n});
                widest = std::max(widest, n);
                k += n;
            }
            const uint64_t lv = std::min<uint64_t>(bv * widest, end - begin);   // variants of the widest launch
            const size_t rec_bytes = (size_t)(bv * R), blob_bytes = (size_t)(bv * max_prefix), text_bytes = (size_t)(bv * max_line);
            const size_t off_bytes = (size_t)(2 * (bv + 1) * sizeof(uint64_t));
            const int n_sets = plan.size() > 1 ? 2 : 1;     // device-side sets (one launch each), each with its own ctx / stream
            const int n_text = n_blocks > 1 ? 2 : 1;        // pinned text buffers (one block each)
            DeviceBuffers sets[2];
            void *h_text[2] = {nullptr, nullptr};
            for (int s = 0; s < n_sets; s++) {
                DeviceBuffers &B = sets[s];
                // a filter that kept NOBODY is an empty list, not "all samples": say so with the flag (kept.data() is NULL then)
                check(pgenhip_create(&B.ctx, g % n_use, N, all_samples ? nullptr : kept.data(), (uint32_t)K,
                                     all_samples ? 0u : PGENHIP_CREATE_KEEP_LIST), "pgenhip_create");
                check(pgenhip_device_malloc(B.ctx, &B.d_rec, (size_t)(lv * R)), "device records");
                check(pgenhip_device_malloc(B.ctx, &B.d_blob, (size_t)(lv * max_prefix)), "device prefixes");
                check(pgenhip_device_malloc(B.ctx, &B.d_off, (size_t)(2 * (lv + 1) * sizeof(uint64_t))), "device offsets");
                check(pgenhip_device_malloc(B.ctx, &B.d_text, (size_t)(lv * max_line)), "device text");
            }
            // pinned staging: ONE set of input buffers (each block's H2D is waited for before the next block is staged) ...
            check(pgenhip_host_malloc_pinned(sets[0].ctx, &sets[0].h_rec, rec_bytes), "pinned records");
            check(pgenhip_host_malloc_pinned(sets[0].ctx, &sets[0].h_blob, blob_bytes), "pinned prefixes");
            check(pgenhip_host_malloc_pinned(sets[0].ctx, &sets[0].h_off, off_bytes), "pinned offsets");
            // ... and two text buffers (freed with the sets: DeviceBuffers owns h_text)
            for (int t = 0; t < n_text; t++) {
                check(pgenhip_host_malloc_pinned(sets[0].ctx, &h_text[t], text_bytes), "pinned text");
                (t == 0 ? sets[0].h_text : sets[0].h_text2) = h_text[t];
            }
            struct InFlight {
                int text;      // pinned text buffer
                int set;       // device set (stream) the copy was queued on
                bool first;    // first block of its launch: the kernel's time is read here
                size_t b0;
                uint64_t bytes;
            };
            std::mutex mu;
            std::condition_variable cv;
            std::deque<InFlight> inflight;
            bool text_busy[2] = {false, false};
            uint64_t pushed[2] = {0, 0}, consumed[2] = {0, 0};   // blocks per device set
            bool producer_done = false;
            std::string consumer_err;

            std::thread consumer([&] {
                try {
                    for (;;) {
                        InFlight job;
                        {
                            std::unique_lock<std::mutex> lk(mu);
                            cv.wait(lk, [&] { return !inflight.empty() || producer_done; });
                            if (inflight.empty()) return;
                            job = inflight.front();
                            inflight.pop_front();
                        }
                        DeviceBuffers &B = sets[job.set];
                        check(pgenhip_wait(B.ctx), "pgenhip_wait");
                        float ms = 0;
                        if (job.first && pgenhip_timer_read(B.ctx, &ms) == PGENHIP_OK) kernel_s[(size_t)g] += ms * 1e-3;
                        // every line has a known length, so ranges land at precomputed offsets in any order
                        const uint64_t file_pos = header.size() + file_off[job.b0];
                        const uint8_t *text = static_cast<const uint8_t *>(h_text[job.text]);
                        if (zw) {
                            zw->write(text, (size_t)job.bytes);   // blocks arrive in order: the members are appended in order
                        } else if (n_writers <= 1 || job.bytes < (8ull << 20)) {
                            pwrite_exact(fd, text, (size_t)job.bytes, file_pos, filename);
                        } else {
                            std::vector<std::thread> ws;
                            std::string werr;
                            std::mutex wmu;
                            const uint64_t slice = (job.bytes + n_writers - 1) / n_writers;
                            for (unsigned t = 0; t < n_writers; t++) {
                                const uint64_t lo = std::min<uint64_t>((uint64_t)t * slice, job.bytes), hi = std::min<uint64_t>(lo + slice, job.bytes);
                                if (lo == hi) continue;
                                ws.emplace_back([&, lo, hi] {
                                    try {
                                        pwrite_exact(fd, text + lo, (size_t)(hi - lo), file_pos + lo, filename);
                                    } catch (const std::exception &e) {
                                        std::lock_guard<std::mutex> lk(wmu);
                                        werr = e.what();
                                    }
                                });
                            }
                            for (auto &t : ws) t.join();
                            if (!werr.empty()) throw PfileError(werr);
                        }
                        {
                            std::lock_guard<std::mutex> lk(mu);
                            text_busy[job.text] = false;
                            consumed[job.set]++;
                        }
                        cv.notify_all();
                    }
                } catch (const std::exception &e) {
                    std::lock_guard<std::mutex> lk(mu);
                    consumer_err = e.what();
                    text_busy[0] = text_busy[1] = false;
                    cv.notify_all();
                }
            });
            struct Joiner {
                std::thread &t;
                std::mutex &mu;
                std::condition_variable &cv;
                bool &done;
                ~Joiner()
                {
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        done = true;
                    }
                    cv.notify_all();
                    if (t.joinable()) t.join();
                }
            } joiner{consumer, mu, cv, producer_done};

            uint8_t *h_rec = static_cast<uint8_t *>(sets[0].h_rec);
            char *h_blob = static_cast<char *>(sets[0].h_blob);
            uint64_t *h_poff = static_cast<uint64_t *>(sets[0].h_off);
            uint64_t *h_loff = h_poff + (bv + 1);
            std::vector<uint64_t> blob_len(plan.size(), 0);   // prefix bytes staged so far, per launch
            auto launch_b0 = [&](size_t u) { return begin + plan[u].first_block * (size_t)bv; };
            auto launch_nv = [&](size_t u) { return std::min<size_t>(plan[u].n_blocks * (size_t)bv, end - launch_b0(u)); };

            // block j of launch u: records from the file, prefixes joined, offsets — into the pinned staging set, then to their place
            // in the launch's device buffers (on the launch's own stream, idle by now: see the wait in the schedule below)
            auto stage_block = [&](size_t u, size_t j) {
                DeviceBuffers &B = sets[u % (size_t)n_sets];
                const size_t r0 = j * (size_t)bv, b0 = launch_b0(u) + r0;
                const size_t nv = std::min<size_t>((size_t)bv, end - b0);
                // :165-170 once per run of consecutive variant indices instead of once per variant
                for (size_t j2 = 0; j2 < nv;) {
                    size_t run = 1;
                    // (variable-width files: consecutive plain records are adjacent on disk too when nothing compressed lies between them)
                    while (j2 + run < nv && var_idx_rcds[b0 + j2 + run].first == var_idx_rcds[b0 + j2].first + run &&
                           record_offset(var_idx_rcds[b0 + j2 + run].first) == record_offset(var_idx_rcds[b0 + j2].first) + run * (uint64_t)R)
                        run++;
                    // (a long run is read by a few threads at once: one thread copies ~2-3 GB/s out of the page cache)
                    const size_t run_bytes = run * (size_t)R;
                    const uint64_t run_off = record_offset(var_idx_rcds[b0 + j2].first);
                    const unsigned n_readers = run_bytes >= (64u << 20) ? (unsigned)std::max(1, opt.read_threads) : 1u;
                    if (n_readers <= 1) {
                        pread_exact(pfd, h_rec + j2 * R, run_bytes, run_off, pgen);
                    } else {
                        std::vector<std::thread> rs;
                        std::string rerr;
                        std::mutex rmu;
                        const size_t slice = (run_bytes + n_readers - 1) / n_readers;
                        for (unsigned t = 0; t < n_readers; t++) {
                            const size_t lo = std::min(run_bytes, (size_t)t * slice), hi = std::min(run_bytes, lo + slice);
                            if (lo == hi) continue;
                            rs.emplace_back([&, lo, hi] {
                                try {
                                    pread_exact(pfd, h_rec + j2 * R + lo, hi - lo, run_off + lo, pgen);
                                } catch (const std::exception &e) {
                                    std::lock_guard<std::mutex> lk(rmu);
                                    rerr = e.what();
                                }
                            });
                        }
                        for (auto &t : rs) t.join();
                        if (!rerr.empty()) throw PfileError(rerr);
                    }
                    j2 += run;
                }
                // :157-161 joined once per variant: col '\t' col '\t' ... "GT"; offsets count from the start of the LAUNCH's blob / text
                const uint64_t blob0 = blob_len[u], text0 = file_off[launch_b0(u)];
                uint64_t bp = 0;
                for (size_t i = 0; i < nv; i++) {
                    h_poff[i] = blob0 + bp;
                    h_loff[i] = file_off[b0 + i] - text0;
                    for (const auto &col : var_idx_rcds[b0 + i].second) {
                        std::memcpy(h_blob + bp, col.data(), col.size());
                        bp += col.size();
                        h_blob[bp++] = '\t';
                    }
                    h_blob[bp++] = 'G';
                    h_blob[bp++] = 'T';
                }
                h_poff[nv] = blob0 + bp;                       // (the next block's first entry, or the launch's end)
                h_loff[nv] = file_off[b0 + nv] - text0;
                blob_len[u] = blob0 + bp;
                uint64_t *d_poff = static_cast<uint64_t *>(B.d_off), *d_loff = d_poff + (lv + 1);
                check(pgenhip_memcpy_h2d(B.ctx, static_cast<uint8_t *>(B.d_rec) + r0 * (size_t)R, h_rec, nv * (size_t)R), "H2D records");
                check(pgenhip_memcpy_h2d(B.ctx, static_cast<char *>(B.d_blob) + blob0, h_blob, (size_t)bp), "H2D prefixes");
                check(pgenhip_memcpy_h2d(B.ctx, d_poff + r0, h_poff, (nv + 1) * sizeof(uint64_t)), "H2D prefix offsets");
                check(pgenhip_memcpy_h2d(B.ctx, d_loff + r0, h_loff, (nv + 1) * sizeof(uint64_t)), "H2D line offsets");
                check(pgenhip_wait(B.ctx), "pgenhip_wait");   // the staging set is free again
            };
            auto launch = [&](size_t u) {
                DeviceBuffers &B = sets[u % (size_t)n_sets];
                uint64_t *d_poff = static_cast<uint64_t *>(B.d_off), *d_loff = d_poff + (lv + 1);
                check(pgenhip_timer_start(B.ctx), "timer");
                check(pgenhip_emit_lines(B.ctx, B.d_rec, R, nullptr, (uint32_t)launch_nv(u), B.d_blob, d_poff, d_loff, max_prefix, B.d_text, 0), "pgenhip_emit_lines");
                check(pgenhip_timer_mark(B.ctx), "timer");
            };
            // block j of launch u leaves: D2H into a free pinned text buffer, then the consumer's
            auto drain_block = [&](size_t u, size_t j) {
                const int si = (int)(u % (size_t)n_sets), ti = (int)((plan[u].first_block + j) % (size_t)n_text);
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !text_busy[ti] || !consumer_err.empty(); });
                    if (!consumer_err.empty()) throw PfileError(consumer_err);
                    text_busy[ti] = true;
                }
                const size_t b0 = launch_b0(u) + j * (size_t)bv, nv = std::min<size_t>((size_t)bv, end - b0);
                const uint64_t block_bytes = file_off[b0 + nv] - file_off[b0];
                check(pgenhip_memcpy_d2h(sets[si].ctx, h_text[ti], static_cast<uint8_t *>(sets[si].d_text) + (file_off[b0] - file_off[launch_b0(u)]), (size_t)block_bytes), "D2H text");
                {
                    std::lock_guard<std::mutex> lk(mu);
                    inflight.push_back(InFlight{ti, si, j == 0, b0, block_bytes});
                    pushed[si]++;
                }
                cv.notify_all();
            };
            // a device set is staged into again only once every block of its previous launch has been written: its stream is idle
            // then, so the producer's waits in stage_block never meet the consumer's on the same stream
            auto wait_set_idle = [&](size_t u) {
                const int si = (int)(u % (size_t)n_sets);
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return consumed[si] == pushed[si] || !consumer_err.empty(); });
                if (!consumer_err.empty()) throw PfileError(consumer_err);
            };

            setup_s[(size_t)g] = now_s() - t_worker;
            for (size_t j = 0; j < plan[0].n_blocks; j++) stage_block(0, j);
            launch(0);
            for (size_t u = 0; u < plan.size(); u++) {
                const size_t n_u = plan[u].n_blocks, n_next = u + 1 < plan.size() ? plan[u + 1].n_blocks : 0;
                size_t staged = 0;
                for (size_t j = 0; j < n_u; j++) {
                    drain_block(u, j);
                    // the next launch's share of staging, so that it is complete when this launch's last block has been queued
                    while (staged < n_next && staged * n_u < (j + 1) * n_next) {
                        if (staged == 0) wait_set_idle(u + 1);
                        stage_block(u + 1, staged++);
                    }
                }
                if (n_next) launch(u + 1);
            }
            {
                std::unique_lock<std::mutex> lk(mu);
                producer_done = true;
                cv.notify_all();
            }
            if (consumer.joinable()) consumer.join();
            if (!consumer_err.empty()) throw PfileError(consumer_err);
            if (zw) shard_out[(size_t)g] = zw->bytes_out();
        } catch (const std::exception &e) {
            std::lock_guard<std::mutex> lk(err_mu);
            if (err.empty()) err = e.what();
        }
    };
    std::vector<std::thread> threads;
    for (int g = 1; g < G; g++) threads.emplace_back(worker, g);
    worker(0);
    for (auto &t : threads) t.join();
    if (!err.empty()) {
        if (opt.bgzf)
            for (int g = 1; g < G; g++) unlink(shard_tmp(g).c_str());
        throw PfileError(err);
    }
    if (opt.bgzf) {
        st.file_bytes += shard_out[0];
        finish_bgzf(G);
    }
    close_checked();
    st.seconds_body = now_s() - t_body;
    st.seconds_kernel = *std::max_element(kernel_s.begin(), kernel_s.end());
    st.seconds_setup = *std::max_element(setup_s.begin(), setup_s.end());
    return st;
}


namespace {
uint64_t splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
}  // namespace

void synth_pfile(const std::string &prefix, uint32_t variants, uint32_t samples, uint32_t keep_modulus, uint64_t seed)
{
    const uint32_t R = pgenhip_variant_record_size(samples);
    {
        FILE *f = std::fopen((prefix + ".pvar").c_str(), "wb");
        if (!f) throw PfileError("create " + prefix + ".pvar: " + std::strerror(errno));
        std::fputs("##fileformat=VCFv4.2\n##source=pgen-hip synth\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n", f);
        for (uint32_t i = 0; i < variants; i++) std::fprintf(f, "22\t%llu\tsnp%u\tA\tG\t100\tPASS\t.\n", 16050000ull + 7ull * i, i);
        std::fclose(f);
        f = std::fopen((prefix + ".psam").c_str(), "wb");
        if (!f) throw PfileError("create " + prefix + ".psam: " + std::strerror(errno));
        std::fputs("#IID\tSEX\tKEEP\n", f);
        for (uint32_t i = 0; i < samples; i++)
            std::fprintf(f, "S%06u\tNA\t%d\n", i, keep_modulus && splitmix64(0x4D41534Bull ^ (uint64_t)i) % keep_modulus == 0 ? 1 : 0);
        std::fclose(f);
    }
    int fd = open((prefix + ".pgen").c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) throw PfileError("create " + prefix + ".pgen: " + std::strerror(errno));
    struct FdGuard {
        int fd;
        ~FdGuard() { close(fd); }
    } guard{fd};
    uint8_t hdr[12] = {0x6C, 0x1B, 0x02, 0, 0, 0, 0, 0, 0, 0, 0, 0x40};
    for (int b = 0; b < 4; b++) {
        hdr[3 + b] = (uint8_t)(variants >> (8 * b));
        hdr[7 + b] = (uint8_t)(samples >> (8 * b));
    }
    pwrite_exact(fd, hdr, sizeof hdr, 0, prefix + ".pgen");
    if (variants == 0 || R == 0) return;
    DeviceBuffers B;
    check(pgenhip_create(&B.ctx, 0, samples, nullptr, 0, 0), "pgenhip_create");
    const uint64_t bv = std::max<uint64_t>(1, std::min<uint64_t>((256ull << 20) / R, variants));
    check(pgenhip_host_malloc_pinned(B.ctx, &B.h_rec, (size_t)(bv * R)), "pinned records");
    check(pgenhip_device_malloc(B.ctx, &B.d_rec, (size_t)(bv * R)), "device records");
    for (uint64_t v0 = 0; v0 < variants; v0 += bv) {
        const uint32_t nv = (uint32_t)std::min<uint64_t>(bv, variants - v0);
        check(pgenhip_synth_records(B.ctx, B.d_rec, R, v0, nv, seed, 0), "pgenhip_synth_records");
        check(pgenhip_memcpy_d2h(B.ctx, B.h_rec, B.d_rec, (size_t)nv * R), "D2H records");
        check(pgenhip_wait(B.ctx), "pgenhip_wait");
        pwrite_exact(fd, B.h_rec, (size_t)nv * R, 12ull + v0 * R, prefix + ".pgen");
    }
}

}  // namespace pgenhost
