// cli.cpp — `pgen-hip`: the pgen-rs command-line surface (src/cli.rs:1-62, dispatch
// src/main.rs:92-127) on top of the MI355X engine.  Same subcommands, flags and defaults:
//
//   pgen-hip query  <PFILE_PREFIX> -f|--fstring <EXPR> [-i|--include <EXPR>] [-s|--samples]
//   pgen-hip filter <PFILE_PREFIX> [--include-var <EXPR>] [--include-sam <EXPR>] [-o|--out <FILE>]
//
// Additions (opt-in, not in the reference): --gpus <N>, --block-mib <M>, --launch-mib <M>, --filter-threads <T>, --stats, --dry-run
// (filter: write the VCF header only and report the body geometry; needs no GPU); BGZF output (`-o x.vcf.gz` or --bgzf,
// --bgzf-level <1-9>, --compress-threads <T>) and `pgen-hip bgzf <IN> <OUT>`, the same writer on a file.
// Exit codes: 0 ok; 2 usage error (clap's code); 101 where the reference would panic.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <optional>
#include <string>
#include <vector>

#include <fcntl.h>
#include <unistd.h>

#include "bgzf.h"
#include "expr.h"
#include "pfile.h"

using namespace pgenhost;

namespace {

struct Fd {
    int fd;
    ~Fd()
    {
        if (fd >= 0) close(fd);
    }
};

void close_or_throw(Fd &f, const std::string &path)
{
    const int fd = f.fd;
    f.fd = -1;
    if (close(fd) != 0) throw PfileError("close " + path + ": " + std::strerror(errno));
}

// --dry-run: the VCF header alone, as text or as a complete BGZF file (header members + EOF marker)
void write_header_only(const std::string &out_file, const std::string &header, bool bgzf, int level)
{
    Fd f{open(out_file.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644)};
    if (f.fd < 0) throw PfileError("create " + out_file + ": " + std::strerror(errno));
    if (bgzf) {
        BgzfWriter w(f.fd, out_file, level, 1);
        w.write(header.data(), header.size());
        w.finish();
    } else {
        const char *p = header.data();
        for (size_t left = header.size(); left;) {
            ssize_t n = write(f.fd, p, left);
            if (n < 0 && errno == EINTR) continue;
            if (n <= 0) throw PfileError("write " + out_file + ": " + std::strerror(errno));
            p += n;
            left -= (size_t)n;
        }
    }
    close_or_throw(f, out_file);
}

void bgzf_file(const std::string &in, const std::string &out, int level, unsigned threads, size_t chunk)
{
    Fd i{open(in.c_str(), O_RDONLY)};
    if (i.fd < 0) throw PfileError("open " + in + ": " + std::strerror(errno));
    Fd o{open(out.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644)};
    if (o.fd < 0) throw PfileError("create " + out + ": " + std::strerror(errno));
    BgzfWriter w(o.fd, out, level, threads);
    std::vector<uint8_t> buf(chunk);
    for (;;) {
        size_t got = 0;
        while (got < buf.size()) {
            ssize_t r = read(i.fd, buf.data() + got, buf.size() - got);
            if (r < 0 && errno == EINTR) continue;
            if (r < 0) throw PfileError("read " + in + ": " + std::strerror(errno));
            if (r == 0) break;
            got += (size_t)r;
        }
        if (!got) break;
        w.write(buf.data(), got);
    }
    w.finish();
    close_or_throw(o, out);
}

const char *kUsage =
    "Usage: pgen-hip <COMMAND>\n\n"
    "Commands:\n"
    "  query   Queries the pgen, outputting to stdout\n"
    "  filter  Filters the pgen, outputting to a VCF\n"
    "  help    Print this message\n\n"
    "query  <PFILE_PREFIX> -f, --fstring <QUERY_FSTRING> [-i, --include <QUERY>] [-s, --samples]\n"
    "filter <PFILE_PREFIX> [--include-var <VAR_QUERY>] [--include-sam <SAM_QUERY>] [-o, --out <OUT_FILE>]\n"
    "       [--gpus <N>] [--shards <S>] [--block-mib <M>] [--launch-mib <M>] [--write-threads <T>] [--read-threads <T>] [--filter-threads <T>] [--stats] [--dry-run]\n"
    "       [--bgzf] [--bgzf-level <1-9>] [--compress-threads <T>]   (BGZF `.vcf.gz`; implied by an OUT_FILE ending in .gz)\n"
    "bgzf   <IN_FILE> <OUT_FILE> [--level <1-9>] [--threads <T>] [--chunk-mib <M>]\n";

[[noreturn]] void usage_error(const std::string &msg)
{
    std::fprintf(stderr, "error: %s\n\n%s", msg.c_str(), kUsage);
    std::exit(2);
}

struct Args {
    std::vector<std::string> positional;
    std::vector<std::pair<std::string, std::string>> options;  // name (without dashes) -> value ("" for flags)
    bool has(const std::string &n) const
    {
        for (const auto &o : options)
            if (o.first == n) return true;
        return false;
    }
    std::optional<std::string> get(const std::string &n) const
    {
        std::optional<std::string> v;
        for (const auto &o : options)
            if (o.first == n) v = o.second;
        return v;
    }
};

// clap-style: long `--name value` / `--name=value`, short `-f value` / `-fvalue`, flags without values
Args parse(int argc, char **argv, int first, const std::vector<std::pair<std::string, char>> &valued,
           const std::vector<std::pair<std::string, char>> &flags)
{
    Args a;
    auto long_of_short = [&](char c, bool &is_flag) -> std::string {
        for (const auto &v : valued)
            if (v.second == c) { is_flag = false; return v.first; }
        for (const auto &f : flags)
            if (f.second == c) { is_flag = true; return f.first; }
        return std::string();
    };
    auto is_valued = [&](const std::string &n) {
        for (const auto &v : valued)
            if (v.first == n) return true;
        return false;
    };
    auto is_flag = [&](const std::string &n) {
        for (const auto &f : flags)
            if (f.first == n) return true;
        return false;
    };
    bool only_positional = false;
    for (int i = first; i < argc; i++) {
        std::string s = argv[i];
        if (only_positional || s.size() < 2 || s[0] != '-') {
            a.positional.push_back(s);
            continue;
        }
        if (s == "--") {
            only_positional = true;
            continue;
        }
        if (s[1] == '-') {
            std::string name = s.substr(2), value;
            const size_t eq = name.find('=');
            const bool inline_value = eq != std::string::npos;
            if (inline_value) {
                value = name.substr(eq + 1);
                name = name.substr(0, eq);
            }
            if (is_valued(name)) {
                if (!inline_value) {
                    if (i + 1 >= argc) usage_error("a value is required for '--" + name + "' but none was supplied");
                    value = argv[++i];
                }
                a.options.emplace_back(name, value);
            } else if (is_flag(name)) {
                a.options.emplace_back(name, "");
            } else {
                usage_error("unexpected argument '--" + name + "' found");
            }
        } else {
            bool flag = false;
            const std::string name = long_of_short(s[1], flag);
            if (name.empty()) usage_error(std::string("unexpected argument '-") + s[1] + "' found");
            if (flag) {
                a.options.emplace_back(name, "");
            } else {
                std::string value = s.substr(2);
                if (!value.empty() && value[0] == '=') value = value.substr(1);
                if (s.size() == 2) {
                    if (i + 1 >= argc) usage_error("a value is required for '-" + std::string(1, s[1]) + "' but none was supplied");
                    value = argv[++i];
                }
                a.options.emplace_back(name, value);
            }
        }
    }
    return a;
}

}  // namespace

int main(int argc, char **argv)
{
    const auto t_main = std::chrono::steady_clock::now();
    if (argc < 2) usage_error("a subcommand is required");
    const std::string cmd = argv[1];
    try {
        if (cmd == "help" || cmd == "--help" || cmd == "-h") {
            std::fputs(kUsage, stdout);
            return 0;
        }
        if (cmd == "--version" || cmd == "-V") {
            std::puts("pgen-hip 0.1.0 (MI355X engine for pgen-rs's filter/query surface)");
            return 0;
        }
        if (cmd == "query") {  // src/main.rs:95-113
            Args a = parse(argc, argv, 2, {{"fstring", 'f'}, {"include", 'i'}}, {{"samples", 's'}});
            if (a.positional.size() != 1) usage_error("the following required arguments were not provided: <PFILE_PREFIX>");
            if (!a.has("fstring")) usage_error("the following required arguments were not provided: --fstring <QUERY_FSTRING>");
            const Pfile pfile = Pfile::from_prefix(a.positional[0]);
            const std::string path = a.has("samples") ? pfile.psam_path() : pfile.pvar_path();
            const std::string data = read_file(path);
            TsvReader reader(data, Pfile::find_metadata_file_header_start(data));
            std::string out;
            Pfile::query_metadata(reader, a.get("include"), *a.get("fstring"), out);
            std::fwrite(out.data(), 1, out.size(), stdout);
            return 0;
        }
        if (cmd == "filter") {  // src/main.rs:114-124
            Args a = parse(argc, argv, 2, {{"include-var", 0}, {"include-sam", 0}, {"out", 'o'}, {"gpus", 0}, {"shards", 0}, {"block-mib", 0}, {"launch-mib", 0}, {"write-threads", 0}, {"read-threads", 0}, {"filter-threads", 0}, {"bgzf-level", 0}, {"compress-threads", 0}},
                           {{"stats", 0}, {"dry-run", 0}, {"bgzf", 0}});
            if (a.positional.size() != 1) usage_error("the following required arguments were not provided: <PFILE_PREFIX>");
            const Pfile pfile = Pfile::from_prefix(a.positional[0]);
            auto ends_with = [](const std::string &x, const char *suffix) { const size_t n = std::strlen(suffix); return x.size() >= n && x.compare(x.size() - n, n, suffix) == 0; };
            const bool bgzf = a.has("bgzf") || (a.get("out") && ends_with(*a.get("out"), ".gz"));
            const std::string out_file = a.get("out").value_or(pfile.pfile_prefix + (bgzf ? ".pgen-rs.vcf.gz" : ".pgen-rs.vcf"));  // :121-122
            const int bgzf_level = a.get("bgzf-level") ? std::atoi(a.get("bgzf-level")->c_str()) : 6;
            if (bgzf_level < 1 || bgzf_level > 9) usage_error("--bgzf-level takes 1 .. 9");
            const int filter_threads = a.get("filter-threads") ? std::max(1, std::atoi(a.get("filter-threads")->c_str())) : 0;
            if (a.has("dry-run")) {
                // header + geometry only: the plumbing of BASELINE config 1 without touching a GPU
                const std::string psam = read_file(pfile.psam_path());
                TsvReader psam_reader(psam, Pfile::find_metadata_file_header_start(psam));
                const StringRecord sam_header = psam_reader.headers();
                const std::string pvar = read_file(pfile.pvar_path());
                TsvReader pvar_reader(pvar, Pfile::find_metadata_file_header_start(pvar));
                const auto vars = Pfile::filter_metadata(pvar_reader, a.get("include-var"), filter_threads);
                const auto sams = Pfile::filter_metadata(psam_reader, a.get("include-sam"), filter_threads);
                const std::string header = pfile.vcf_header(sams, sam_header);
                write_header_only(out_file, header, bgzf, bgzf_level);
                unsigned long long prefix = 0;
                for (const auto &v : vars) {
                    prefix += 2;
                    for (const auto &c : v.second) prefix += c.size() + 1;
                }
                const unsigned long long body = prefix + (unsigned long long)vars.size() * (4ull * sams.size() + 1ull);
                std::printf("{\"variants_kept\": %zu, \"samples_kept\": %zu, \"header_bytes\": %zu, \"prefix_bytes\": %llu, \"body_bytes\": %llu, \"file_bytes\": %llu}\n",
                            vars.size(), sams.size(), header.size(), prefix, body, (unsigned long long)header.size() + body);
                return 0;
            }
            OutputOptions opt;
            opt.filter_threads = filter_threads;
            opt.bgzf = bgzf;
            opt.bgzf_level = bgzf_level;
            if (auto c = a.get("compress-threads")) opt.compress_threads = std::max(1, std::atoi(c->c_str()));
            if (auto g = a.get("gpus")) opt.n_gpus = std::max(1, std::atoi(g->c_str()));
            if (auto sh = a.get("shards")) opt.n_shards = std::max(1, std::atoi(sh->c_str()));
            if (auto w = a.get("write-threads")) opt.write_threads = std::max(1, std::atoi(w->c_str()));
            if (auto w = a.get("read-threads")) opt.read_threads = std::max(1, std::atoi(w->c_str()));
            if (auto m = a.get("block-mib")) opt.block_text_bytes = (uint64_t)std::max(1, std::atoi(m->c_str())) << 20;
            if (auto m = a.get("launch-mib")) opt.launch_bytes = (uint64_t)std::max(1, std::atoi(m->c_str())) << 20;
            const OutputStats st = pfile.output_vcf(a.get("include-sam"), a.get("include-var"), out_file, opt);  // :123
            if (a.has("stats")) {
                std::fprintf(stderr,
                             "{\"variants_kept\": %llu, \"samples_kept\": %llu, \"header_bytes\": %llu, \"body_bytes\": %llu, "
                             "\"file_bytes\": %llu, \"seconds_filter\": %.6f, \"seconds_body\": %.6f, \"seconds_setup\": %.6f, \"seconds_kernel\": %.6f, \"seconds_main\": %.6f}\n",
                             (unsigned long long)st.variants, (unsigned long long)st.samples_kept, (unsigned long long)st.header_bytes,
                             (unsigned long long)st.body_bytes, (unsigned long long)st.file_bytes, st.seconds_filter, st.seconds_body, st.seconds_setup, st.seconds_kernel,
                             std::chrono::duration<double>(std::chrono::steady_clock::now() - t_main).count());
            }
            return 0;
        }
        if (cmd == "bgzf") {
            // not in the reference: the BGZF writer of `filter ... -o x.vcf.gz` applied to a file (what `bgzip -c IN > OUT` does)
            Args a = parse(argc, argv, 2, {{"level", 0}, {"threads", 0}, {"chunk-mib", 0}}, {});
            if (a.positional.size() != 2) usage_error("bgzf <IN_FILE> <OUT_FILE> [--level <1-9>] [--threads <T>] [--chunk-mib <M>]");
            const int level = a.get("level") ? std::atoi(a.get("level")->c_str()) : 6;
            if (level < 1 || level > 9) usage_error("--level takes 1 .. 9");
            const unsigned threads = (unsigned)std::max(1, std::atoi(a.get("threads").value_or("8").c_str()));
            const size_t chunk = (size_t)std::max(1, std::atoi(a.get("chunk-mib").value_or("64").c_str())) << 20;
            bgzf_file(a.positional[0], a.positional[1], level, threads, chunk);
            return 0;
        }
        if (cmd == "synth") {
            // not in the reference: writes a synthetic PREFIX.{pgen,pvar,psam} triple of the SURVEY §8d shapes
            // (records from the device generator pgenhip_synth_records; KEEP column = the 1-in-M keep mask)
            Args a = parse(argc, argv, 2, {{"variants", 0}, {"samples", 0}, {"keep-modulus", 0}, {"seed", 0}}, {});
            if (a.positional.size() != 1 || !a.has("variants") || !a.has("samples"))
                usage_error("synth <PFILE_PREFIX> --variants <V> --samples <N> [--keep-modulus <M>] [--seed <S>]");
            synth_pfile(a.positional[0], (uint32_t)std::strtoul(a.get("variants")->c_str(), nullptr, 10),
                        (uint32_t)std::strtoul(a.get("samples")->c_str(), nullptr, 10),
                        (uint32_t)std::strtoul(a.get("keep-modulus").value_or("100").c_str(), nullptr, 10),
                        std::strtoull(a.get("seed").value_or("1346848078").c_str(), nullptr, 10));  // 0x5047454E "PGEN": seed_data of SURVEY.md 8(d) (round 2 had a typo here)
            return 0;
        }
        usage_error("unrecognized subcommand '" + cmd + "'");
    } catch (const PfileError &e) {
        std::fprintf(stderr, "pgen-hip: %s\n", e.what());
        return 101;
    } catch (const CsvError &e) {
        std::fprintf(stderr, "pgen-hip: %s\n", e.what());
        return 101;
    } catch (const ExprError &e) {
        std::fprintf(stderr, "pgen-hip: %s\n", e.what());
        return 101;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "pgen-hip: %s\n", e.what());
        return 101;
    }
    return 0;
}
