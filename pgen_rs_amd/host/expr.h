// expr.h — the slice of `evalexpr` (11.3.0; Cargo.lock:121-124) pgen-rs uses for its
// --include / --include-var / --include-sam / -f expressions, restated in C++.
// Call sites in the reference: src/pfile.rs:87-97 (query_metadata), :322-328 (filter_metadata):
// a HashMapContext is filled with one String variable per column (Value::String, :90, :325) and
// the expression is evaluated with eval_boolean_with_context / eval_string_with_context.
//
// Restated from the crate's documentation (crate source absent, no reference test pins it —
// "parity unpinned" at this boundary): values String / Int / Float / Boolean; operators with
// precedence  ^ (120) > unary - ! (110) > * / % (100) > + - (95) > < > <= >= == != (80) >
// && (75) > || (70); `+` concatenates two strings; == / != compare any two values (different
// types are unequal); < > <= >= on numbers and on strings; && || ! need booleans; both
// operands are always evaluated (no short circuit); an unknown identifier is an error.
// Not restated: functions, tuples, assignment, `;` chains — rejected with a clear error.
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace pgenhost {

struct ExprError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct Value {
    enum Kind { STRING, INT, FLOAT, BOOL } kind = BOOL;
    std::string s;
    int64_t i = 0;
    double f = 0.0;
    bool b = false;
    static Value str(std::string v) { Value x; x.kind = STRING; x.s = std::move(v); return x; }
    static Value integer(int64_t v) { Value x; x.kind = INT; x.i = v; return x; }
    static Value floating(double v) { Value x; x.kind = FLOAT; x.f = v; return x; }
    static Value boolean(bool v) { Value x; x.kind = BOOL; x.b = v; return x; }
    std::string describe() const;
};

class Expr {
  public:
    // Parses once; throws ExprError on syntax the restatement does not cover.
    explicit Expr(const std::string &source);
    ~Expr();
    Expr(Expr &&) noexcept;
    Expr(const Expr &) = delete;

    // Resolve identifiers against the column names (a later duplicate wins, like repeated
    // HashMapContext::set_value calls at src/pfile.rs:88-92).
    void bind(const std::vector<std::string> &headers);

    Value eval(const std::vector<std::string> &row) const;
    // eval_boolean_with_context / eval_string_with_context: type mismatch is an error (the
    // reference unwrap()s it into a panic, src/pfile.rs:94, :97, :328).
    bool eval_boolean(const std::vector<std::string> &row) const;
    std::string eval_string(const std::vector<std::string> &row) const;

    struct Node;

  private:
    std::unique_ptr<Node> root_;
    std::string source_;
};

}  // namespace pgenhost
