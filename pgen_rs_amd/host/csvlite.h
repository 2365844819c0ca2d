// csvlite.h — the slice of the `csv` crate (1.3.0, csv-core 0.1.11; Cargo.lock:100-118) that
// pgen-rs relies on, restated in C++ for the host side.  Call sites in the reference:
// src/pfile.rs:275-282 (ReaderBuilder: delimiter b'\t', has_headers(true), everything else
// default), :84-86 / :317-320 (headers() then records()).
//
// Defaults restated (from the crate's documentation; the crate source is not in /root/reference
// and the reference has no test that pins it — "parity unpinned" at this boundary):
//   * quote '"' enabled, doubled quote inside a quoted field = one quote, no escape char;
//     a quote only opens a quoted field at the start of a field;
//   * record terminator: "\n", "\r\n" or a lone "\r"; empty lines are skipped;
//   * no trimming; last record needs no trailing newline;
//   * flexible(false): a record whose field count differs from the first record is an error.
#pragma once
#include <cstddef>
#include <stdexcept>
#include <string>
#include <vector>

namespace pgenhost {

struct CsvError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

using StringRecord = std::vector<std::string>;

class TsvReader {
  public:
    // Parses `data` starting at byte `start` (the reference seeks the file to just after the '#'
    // of the column-header line, src/pfile.rs:271-273).  The first record becomes the headers.
    TsvReader(const std::string &data, size_t start, char delimiter = '\t');
    // A reader over the data records in [start, end) of a file whose header row was already read
    // by `parent` (parallel filtering: [start, end) must begin and end on record boundaries).
    TsvReader(const TsvReader &parent, size_t start, size_t end);

    const StringRecord &headers() const { return headers_; }
    // Next data record; false at end of input.  Throws CsvError on a ragged record.
    bool next(StringRecord &out);
    size_t records_read() const { return n_records_; }
    // Byte range not yet consumed, and the text itself (for splitting the rest between threads).
    size_t position() const { return pos_; }
    size_t end_position() const { return end_; }
    const std::string &data() const { return data_; }

  private:
    bool read_record(StringRecord &out);
    const std::string &data_;
    size_t pos_;
    size_t end_;
    char delim_;
    StringRecord headers_;
    size_t n_records_ = 0;
    size_t line_ = 1;
};

}  // namespace pgenhost
