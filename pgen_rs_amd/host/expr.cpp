// expr.cpp — see expr.h for the restated `evalexpr` semantics.
#include "expr.h"

#include <cerrno>
#include <cmath>
#include <cstdlib>

namespace pgenhost {

std::string Value::describe() const
{
    switch (kind) {
        case STRING: return "String(\"" + s + "\")";
        case INT: return "Int(" + std::to_string(i) + ")";
        case FLOAT: return "Float(" + std::to_string(f) + ")";
        default: return std::string("Boolean(") + (b ? "true" : "false") + ")";
    }
}

namespace {

enum class Tok { END, IDENT, STRING, INT, FLOAT, TRUE_, FALSE_, LPAREN, RPAREN, PLUS, MINUS, STAR, SLASH, PERCENT, HAT, EQ, NEQ, LT, GT, LEQ, GEQ, AND, OR, NOT };

struct Token {
    Tok t = Tok::END;
    std::string text;
    int64_t i = 0;
    double f = 0.0;
};

std::vector<Token> tokenize(const std::string &src)
{
    std::vector<Token> out;
    size_t p = 0;
    const size_t n = src.size();
    auto bad = [&](const std::string &why) -> ExprError { return ExprError("expression \"" + src + "\": " + why); };
    while (p < n) {
        const char c = src[p];
        if (c == ' ' || c == '\t' || c == '\n' || c == '\r') {
            p++;
            continue;
        }
        Token tk;
        if (c == '"') {
            p++;
            std::string s;
            bool closed = false;
            while (p < n) {
                if (src[p] == '\\' && p + 1 < n && (src[p + 1] == '"' || src[p + 1] == '\\')) {
                    s.push_back(src[p + 1]);
                    p += 2;
                } else if (src[p] == '"') {
                    closed = true;
                    p++;
                    break;
                } else {
                    s.push_back(src[p++]);
                }
            }
            if (!closed) throw bad("unterminated string literal");
            tk.t = Tok::STRING;
            tk.text = s;
        } else if ((c >= '0' && c <= '9') || (c == '.' && p + 1 < n && src[p + 1] >= '0' && src[p + 1] <= '9')) {
            size_t q = p;
            bool is_float = false;
            while (q < n && ((src[q] >= '0' && src[q] <= '9') || src[q] == '.' || src[q] == 'e' || src[q] == 'E' ||
                             ((src[q] == '+' || src[q] == '-') && q > p && (src[q - 1] == 'e' || src[q - 1] == 'E')))) {
                if (src[q] == '.' || src[q] == 'e' || src[q] == 'E') is_float = true;
                q++;
            }
            const std::string num = src.substr(p, q - p);
            if (is_float) {
                tk.t = Tok::FLOAT;
                tk.f = std::strtod(num.c_str(), nullptr);
            } else {
                tk.t = Tok::INT;
                errno = 0;
                tk.i = std::strtoll(num.c_str(), nullptr, 10);
                // evalexpr 11.3.0 (Cargo.lock:121-124; source not in the mount, PARITY UNPINNED) tokenises a literal as i64 first and,
                // when that parse fails, as f64: a digit string that does not fit 64 bits is a Float, never a saturated Int
                if (errno == ERANGE) {
                    tk.t = Tok::FLOAT;
                    tk.f = std::strtod(num.c_str(), nullptr);
                }
            }
            p = q;
        } else if ((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '_') {
            size_t q = p;
            while (q < n && ((src[q] >= 'A' && src[q] <= 'Z') || (src[q] >= 'a' && src[q] <= 'z') || (src[q] >= '0' && src[q] <= '9') ||
                             src[q] == '_' || src[q] == '.' || src[q] == ':'))
                q++;
            tk.text = src.substr(p, q - p);
            tk.t = tk.text == "true" ? Tok::TRUE_ : (tk.text == "false" ? Tok::FALSE_ : Tok::IDENT);
            p = q;
        } else {
            auto two = [&](char a, char b) { return c == a && p + 1 < n && src[p + 1] == b; };
            if (two('=', '=')) { tk.t = Tok::EQ; p += 2; }
            else if (two('!', '=')) { tk.t = Tok::NEQ; p += 2; }
            else if (two('<', '=')) { tk.t = Tok::LEQ; p += 2; }
            else if (two('>', '=')) { tk.t = Tok::GEQ; p += 2; }
            else if (two('&', '&')) { tk.t = Tok::AND; p += 2; }
            else if (two('|', '|')) { tk.t = Tok::OR; p += 2; }
            else {
                switch (c) {
                    case '(': tk.t = Tok::LPAREN; break;
                    case ')': tk.t = Tok::RPAREN; break;
                    case '+': tk.t = Tok::PLUS; break;
                    case '-': tk.t = Tok::MINUS; break;
                    case '*': tk.t = Tok::STAR; break;
                    case '/': tk.t = Tok::SLASH; break;
                    case '%': tk.t = Tok::PERCENT; break;
                    case '^': tk.t = Tok::HAT; break;
                    case '<': tk.t = Tok::LT; break;
                    case '>': tk.t = Tok::GT; break;
                    case '!': tk.t = Tok::NOT; break;
                    default: throw bad(std::string("unsupported character '") + c + "' (only the evalexpr subset pgen-rs queries use is restated)");
                }
                p++;
            }
        }
        out.push_back(tk);
    }
    out.push_back(Token{});
    return out;
}

int binary_precedence(Tok t)
{
    switch (t) {
        case Tok::HAT: return 120;
        case Tok::STAR: case Tok::SLASH: case Tok::PERCENT: return 100;
        case Tok::PLUS: case Tok::MINUS: return 95;
        case Tok::LT: case Tok::GT: case Tok::LEQ: case Tok::GEQ: case Tok::EQ: case Tok::NEQ: return 80;
        case Tok::AND: return 75;
        case Tok::OR: return 70;
        default: return -1;
    }
}

}  // namespace

struct Expr::Node {
    enum Kind { CONST, VAR, UNARY, BINARY } kind = CONST;
    Value constant;
    std::string name;
    int column = -1;
    Tok op = Tok::END;
    std::unique_ptr<Node> lhs, rhs;
};

namespace {

struct Parser {
    const std::vector<Token> &toks;
    const std::string &src;
    size_t p = 0;
    ExprError bad(const std::string &why) const { return ExprError("expression \"" + src + "\": " + why); }

    std::unique_ptr<Expr::Node> primary()
    {
        const Token &t = toks[p];
        auto node = std::make_unique<Expr::Node>();
        switch (t.t) {
            case Tok::STRING: node->constant = Value::str(t.text); p++; return node;
            case Tok::INT: node->constant = Value::integer(t.i); p++; return node;
            case Tok::FLOAT: node->constant = Value::floating(t.f); p++; return node;
            case Tok::TRUE_: node->constant = Value::boolean(true); p++; return node;
            case Tok::FALSE_: node->constant = Value::boolean(false); p++; return node;
            case Tok::IDENT:
                if (toks[p + 1].t == Tok::LPAREN) throw bad("function calls are not part of the restated evalexpr subset");
                node->kind = Expr::Node::VAR;
                node->name = t.text;
                p++;
                return node;
            case Tok::LPAREN: {
                p++;
                auto inner = expression(0);
                if (toks[p].t != Tok::RPAREN) throw bad("missing ')'");
                p++;
                return inner;
            }
            case Tok::MINUS:
            case Tok::NOT: {
                node->kind = Expr::Node::UNARY;
                node->op = t.t;
                p++;
                node->lhs = expression(110);  // unary binds tighter than * / %
                return node;
            }
            default: throw bad("unexpected token");
        }
    }

    std::unique_ptr<Expr::Node> expression(int min_prec)
    {
        auto lhs = primary();
        for (;;) {
            const Tok op = toks[p].t;
            const int prec = binary_precedence(op);
            if (prec < 0 || prec < min_prec) return lhs;
            p++;
            // left-associative except ^
            auto rhs = expression(op == Tok::HAT ? prec : prec + 1);
            auto node = std::make_unique<Expr::Node>();
            node->kind = Expr::Node::BINARY;
            node->op = op;
            node->lhs = std::move(lhs);
            node->rhs = std::move(rhs);
            lhs = std::move(node);
        }
    }
};

void bind_node(Expr::Node *n, const std::vector<std::string> &headers)
{
    if (!n) return;
    if (n->kind == Expr::Node::VAR) {
        n->column = -1;
        for (size_t c = 0; c < headers.size(); c++)
            if (headers[c] == n->name) n->column = (int)c;  // last one wins
    }
    bind_node(n->lhs.get(), headers);
    bind_node(n->rhs.get(), headers);
}

bool is_num(const Value &v) { return v.kind == Value::INT || v.kind == Value::FLOAT; }
double as_f(const Value &v) { return v.kind == Value::INT ? (double)v.i : v.f; }

Value eval_node(const Expr::Node *n, const std::vector<std::string> &row, const std::string &src)
{
    auto bad = [&](const std::string &why) -> ExprError { return ExprError("expression \"" + src + "\": " + why); };
    switch (n->kind) {
        case Expr::Node::CONST: return n->constant;
        case Expr::Node::VAR:
            if (n->column < 0 || (size_t)n->column >= row.size()) throw bad("VariableIdentifierNotFound(\"" + n->name + "\")");
            return Value::str(row[(size_t)n->column]);  // every column is a Value::String (src/pfile.rs:90, :325)
        case Expr::Node::UNARY: {
            const Value v = eval_node(n->lhs.get(), row, src);
            if (n->op == Tok::NOT) {
                if (v.kind != Value::BOOL) throw bad("expected a boolean for '!', got " + v.describe());
                return Value::boolean(!v.b);
            }
            if (v.kind == Value::INT) {
                if (v.i == INT64_MIN) throw bad("integer overflow in unary '-'");  // evalexpr: checked arithmetic
                return Value::integer(-v.i);
            }
            if (v.kind == Value::FLOAT) return Value::floating(-v.f);
            throw bad("expected a number for unary '-', got " + v.describe());
        }
        default: break;
    }
    const Value a = eval_node(n->lhs.get(), row, src);
    const Value b = eval_node(n->rhs.get(), row, src);  // no short circuit
    switch (n->op) {
        case Tok::EQ:
        case Tok::NEQ: {
            bool eq;
            if (a.kind != b.kind) eq = false;
            else if (a.kind == Value::STRING) eq = a.s == b.s;
            else if (a.kind == Value::INT) eq = a.i == b.i;
            else if (a.kind == Value::FLOAT) eq = a.f == b.f;
            else eq = a.b == b.b;
            return Value::boolean(n->op == Tok::EQ ? eq : !eq);
        }
        case Tok::LT: case Tok::GT: case Tok::LEQ: case Tok::GEQ: {
            int cmp;
            if (a.kind == Value::STRING && b.kind == Value::STRING) cmp = a.s < b.s ? -1 : (a.s == b.s ? 0 : 1);
            else if (is_num(a) && is_num(b)) {
                if (a.kind == Value::INT && b.kind == Value::INT) cmp = a.i < b.i ? -1 : (a.i == b.i ? 0 : 1);
                else cmp = as_f(a) < as_f(b) ? -1 : (as_f(a) == as_f(b) ? 0 : 1);
            } else throw bad("cannot order " + a.describe() + " and " + b.describe());
            const bool r = n->op == Tok::LT ? cmp < 0 : n->op == Tok::GT ? cmp > 0 : n->op == Tok::LEQ ? cmp <= 0 : cmp >= 0;
            return Value::boolean(r);
        }
        case Tok::AND:
        case Tok::OR:
            if (a.kind != Value::BOOL || b.kind != Value::BOOL) throw bad("expected booleans for '&&'/'||', got " + a.describe() + " and " + b.describe());
            return Value::boolean(n->op == Tok::AND ? (a.b && b.b) : (a.b || b.b));
        case Tok::PLUS:
            if (a.kind == Value::STRING && b.kind == Value::STRING) return Value::str(a.s + b.s);
            [[fallthrough]];
        case Tok::MINUS: case Tok::STAR: case Tok::SLASH: case Tok::PERCENT: case Tok::HAT: {
            if (!is_num(a) || !is_num(b)) throw bad("expected numbers (or two strings for '+'), got " + a.describe() + " and " + b.describe());
            if (n->op == Tok::HAT) return Value::floating(std::pow(as_f(a), as_f(b)));
            if (a.kind == Value::INT && b.kind == Value::INT) {
                switch (n->op) {
                    // evalexpr's integer operators are checked (an overflow is an error, i.e. a panic / exit 101 in pgen-rs, not
                    // a wrapped or saturated value); C++ signed overflow would be undefined behaviour, INT64_MIN / -1 a trap
                    case Tok::PLUS: { int64_t r; if (__builtin_add_overflow(a.i, b.i, &r)) throw bad("integer overflow in '+'"); return Value::integer(r); }
                    case Tok::MINUS: { int64_t r; if (__builtin_sub_overflow(a.i, b.i, &r)) throw bad("integer overflow in '-'"); return Value::integer(r); }
                    case Tok::STAR: { int64_t r; if (__builtin_mul_overflow(a.i, b.i, &r)) throw bad("integer overflow in '*'"); return Value::integer(r); }
                    case Tok::SLASH:
                        if (b.i == 0) throw bad("division by zero");
                        if (a.i == INT64_MIN && b.i == -1) throw bad("integer overflow in '/'");
                        return Value::integer(a.i / b.i);
                    default:
                        if (b.i == 0) throw bad("modulo by zero");
                        if (a.i == INT64_MIN && b.i == -1) throw bad("integer overflow in '%'");
                        return Value::integer(a.i % b.i);
                }
            }
            const double x = as_f(a), y = as_f(b);
            switch (n->op) {
                case Tok::PLUS: return Value::floating(x + y);
                case Tok::MINUS: return Value::floating(x - y);
                case Tok::STAR: return Value::floating(x * y);
                case Tok::SLASH: return Value::floating(x / y);
                default: return Value::floating(std::fmod(x, y));
            }
        }
        default: throw bad("internal: unknown operator");
    }
}

}  // namespace

Expr::Expr(const std::string &source) : source_(source)
{
    const std::vector<Token> toks = tokenize(source);
    Parser ps{toks, source_};
    root_ = ps.expression(0);
    if (toks[ps.p].t != Tok::END) throw ps.bad("trailing tokens");
}

Expr::~Expr() = default;
Expr::Expr(Expr &&) noexcept = default;

void Expr::bind(const std::vector<std::string> &headers) { bind_node(root_.get(), headers); }

Value Expr::eval(const std::vector<std::string> &row) const { return eval_node(root_.get(), row, source_); }

bool Expr::eval_boolean(const std::vector<std::string> &row) const
{
    const Value v = eval(row);
    if (v.kind != Value::BOOL) throw ExprError("expression \"" + source_ + "\": ExpectedBoolean { actual: " + v.describe() + " }");
    return v.b;
}

std::string Expr::eval_string(const std::vector<std::string> &row) const
{
    const Value v = eval(row);
    if (v.kind != Value::STRING) throw ExprError("expression \"" + source_ + "\": ExpectedString { actual: " + v.describe() + " }");
    return v.s;
}

}  // namespace pgenhost
