// csvlite.cpp — see csvlite.h for the restated `csv` crate semantics.
#include "csvlite.h"

namespace pgenhost {

TsvReader::TsvReader(const TsvReader &parent, size_t start, size_t end)
    : data_(parent.data_), pos_(start), end_(end), delim_(parent.delim_), headers_(parent.headers_)
{
}

TsvReader::TsvReader(const std::string &data, size_t start, char delimiter) : data_(data), pos_(start), end_(data.size()), delim_(delimiter)
{
    // has_headers(true): the first record is the header row (src/pfile.rs:280, :84, :317)
    if (!read_record(headers_)) headers_.clear();
}

bool TsvReader::next(StringRecord &out)
{
    if (!read_record(out)) return false;
    if (out.size() != headers_.size()) {
        throw CsvError("CSV error: record " + std::to_string(n_records_ + 1) + " (line " + std::to_string(line_) +
                       "): found record with " + std::to_string(out.size()) + " fields, but the previous record has " +
                       std::to_string(headers_.size()) + " fields");
    }
    n_records_++;
    return true;
}

bool TsvReader::read_record(StringRecord &out)
{
    out.clear();
    const size_t n = end_;
    // skip empty lines
    while (pos_ < n && (data_[pos_] == '\n' || data_[pos_] == '\r')) {
        if (data_[pos_] == '\n') line_++;
        pos_++;
    }
    if (pos_ >= n) return false;
    std::string field;
    bool at_field_start = true;
    bool in_quotes = false;
    for (;;) {
        if (pos_ >= n) {  // end of input terminates the last record
            out.push_back(field);
            return true;
        }
        const char c = data_[pos_];
        if (in_quotes) {
            if (c == '"') {
                if (pos_ + 1 < n && data_[pos_ + 1] == '"') {  // doubled quote
                    field.push_back('"');
                    pos_ += 2;
                } else {
                    in_quotes = false;
                    pos_++;
                }
            } else {
                if (c == '\n') line_++;
                field.push_back(c);
                pos_++;
            }
            continue;
        }
        if (c == delim_) {
            out.push_back(field);
            field.clear();
            at_field_start = true;
            pos_++;
            continue;
        }
        if (c == '\n' || c == '\r') {
            out.push_back(field);
            if (c == '\r' && pos_ + 1 < n && data_[pos_ + 1] == '\n') pos_++;
            pos_++;
            line_++;
            return true;
        }
        if (c == '"' && at_field_start) {
            in_quotes = true;
            at_field_start = false;
            pos_++;
            continue;
        }
        field.push_back(c);
        at_field_start = false;
        pos_++;
    }
}

}  // namespace pgenhost
