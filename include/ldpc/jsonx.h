// jsonx.h -- reader / writer for ldpc-lib's "jsonx" configuration format (header only, C++17).
//
// Own implementation of the format upstream's settings.cpp:178-345 parses and :430-520 writes, so that the scenario and
// result files of `ldpc-lib simulation` (files/*.jsonx upstream) round-trip with this repository's driver
// (ldpc-lib_amd/csrc/cli/ldpc_sim.cpp).  Grammar, as upstream reads it:
//
//   value   := record | array | matrix | sparse | string | fileref | number
//   record  := '{' { identifier '=' value } '}'            a repeated key keeps its FIRST value (std::map::insert, :314)
//   array   := 'array' '{' { value } '}'  |  'array' '@' string      (file: a sequence of values, :221-233)
//   matrix  := 'matrix' '(' rows cols ')' '{' rows*cols values '}'
//   sparse  := 'sparse' 'matrix' '(' rows cols ')' '{' { row col value } '}'      (absent cells: empty record)
//   string  := '"' chars '"'          fileref := '@' string  (the file's single value, relative to the including file)
//   number  := [-.0-9eE]+             kept as text (upstream stores numbers as strings and converts on cast_to)
//   comment := '/' ... end of line    (a single slash is enough, :111-113); anything <= ' ' is white space
//
// select("a/b/c") falls back to the record's "defaults" member at every level (:372-398), like upstream.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace jsonx {

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

class Value {
public:
    enum Type { RECORD, STRING, ARRAY, MATRIX };
    Type type = RECORD;
    std::string str;                                       // STRING (numbers too)
    std::vector<Value> items;                              // ARRAY, or MATRIX cells row-major
    int rows = 0, cols = 0;                                // MATRIX
    std::vector<std::pair<std::string, Value>> fields;     // RECORD, in file order (upstream sorts by key; order carries no meaning)

    Value() = default;
    static Value string(std::string s) { Value v; v.type = STRING; v.str = std::move(s); return v; }
    static Value number(long long x) { return string(std::to_string(x)); }
    static Value number(double x) {                          // shortest text that reads back to the same double
        char buf[40];
        for (int prec = 6; prec <= 17; ++prec) {
            snprintf(buf, sizeof buf, "%.*g", prec, x);
            if (strtod(buf, nullptr) == x) break;
        }
        return string(buf);
    }
    static Value array(std::vector<Value> v = {}) { Value a; a.type = ARRAY; a.items = std::move(v); return a; }
    template <class T> static Value numbers(const std::vector<T> &xs) {
        Value a = array();
        for (const T &x : xs) a.items.push_back(number(x));
        return a;
    }
    static Value matrix(int r, int c, const std::vector<int> &cells) {
        Value m; m.type = MATRIX; m.rows = r; m.cols = c;
        for (int x : cells) m.items.push_back(number((long long)x));
        return m;
    }

    // ---- record access -------------------------------------------------------------------------------------
    const Value *find(const std::string &key) const {
        if (type != RECORD) return nullptr;
        for (const auto &f : fields) if (f.first == key) return &f.second;
        return nullptr;
    }
    Value &set(const std::string &key, Value v) {           // overwrite or append
        for (auto &f : fields) if (f.first == key) { f.second = std::move(v); return f.second; }
        fields.emplace_back(key, std::move(v));
        return fields.back().second;
    }
    // upstream settings::select: path "a/b"; a missing key is looked up in the record's "defaults" (recursively)
    const Value &select(const std::string &path) const {
        if (path.empty()) return *this;
        const size_t slash = path.find('/');
        const std::string head = path.substr(0, slash), rest = slash == std::string::npos ? "" : path.substr(slash + 1);
        if (type == RECORD) {
            if (const Value *v = find(head)) return v->select(rest);
            if (const Value *d = find("defaults")) return d->select(path);
        }
        if (type == ARRAY && !head.empty() && head.find_first_not_of("0123456789") == std::string::npos) {   // (extension: array index)
            const size_t i = (size_t)strtoull(head.c_str(), nullptr, 10);
            if (i < items.size()) return items[i].select(rest);
        }
        throw Error("jsonx: no field '" + path + "'");
    }
    bool has(const std::string &path) const {
        try { (void)select(path); return true; } catch (const Error &) { return false; }
    }

    // ---- conversions (upstream cast_to) ---------------------------------------------------------------------
    long long as_int() const {
        if (type != STRING) throw Error("jsonx: number expected");
        char *end = nullptr;
        const long long v = strtoll(str.c_str(), &end, 10);
        if (end == str.c_str() || *end) {                   // "1e5" and the like: go through double
            const double d = strtod(str.c_str(), &end);
            if (end == str.c_str() || *end) throw Error("jsonx: '" + str + "' is not a number");
            if (!(d > -9.0e18 && d < 9.0e18)) throw Error("jsonx: '" + str + "' does not fit an integer");
            return (long long)d;
        }
        return v;
    }
    double as_double() const {
        if (type != STRING) throw Error("jsonx: number expected");
        char *end = nullptr;
        const double d = strtod(str.c_str(), &end);
        if (end == str.c_str() || *end) throw Error("jsonx: '" + str + "' is not a number");
        return d;
    }
    const std::string &as_string() const {
        if (type != STRING) throw Error("jsonx: string expected");
        return str;
    }
    std::vector<double> as_doubles() const {
        if (type != ARRAY) throw Error("jsonx: array expected");
        std::vector<double> v;
        for (const Value &x : items) v.push_back(x.as_double());
        return v;
    }
    std::vector<int> as_int_matrix(int &r, int &c) const {
        if (type != MATRIX) throw Error("jsonx: matrix expected");
        r = rows; c = cols;
        std::vector<int> v;
        for (const Value &x : items) v.push_back(x.type == STRING ? (int)x.as_int() : 0);
        return v;
    }

    // ---- writer (upstream settings::write, :430-520: two-space indentation, "key = value") ----------------------
    void write(std::ostream &os, int indent = 0) const {
        const std::string pad((size_t)indent, ' ');
        switch (type) {
        case STRING: {
            bool numeric = !str.empty();
            for (char ch : str) numeric = numeric && (ch == '.' || (ch >= '0' && ch <= '9') || ch == '-' || ch == 'e' || ch == 'E' || ch == '+');
            if (numeric && str.find('+') == std::string::npos) os << str;
            else if (numeric) { std::string t; for (char ch : str) if (ch != '+') t += ch; os << t; }   // upstream's number lexer has no '+'
            else os << '"' << str << '"';
            break;
        }
        case ARRAY: {
            bool flat = true;
            for (const Value &x : items) flat = flat && x.type == STRING;
            if (flat) {
                os << "array {";
                for (const Value &x : items) { os << ' '; x.write(os, 0); }
                os << " }";
            } else {
                os << "array {\n";
                for (const Value &x : items) { os << pad << "  "; x.write(os, indent + 2); os << '\n'; }
                os << pad << '}';
            }
            break;
        }
        case MATRIX: {
            os << "matrix (" << rows << ' ' << cols << ") {\n";
            for (int r = 0; r < rows; ++r) {
                os << pad << "  ";
                for (int c = 0; c < cols; ++c) {
                    const Value &x = items[(size_t)r * cols + c];
                    std::ostringstream cell;
                    x.write(cell, 0);
                    std::string t = cell.str();
                    while (t.size() < 4) t = " " + t;
                    os << ' ' << t;
                }
                os << '\n';
            }
            os << pad << '}';
            break;
        }
        case RECORD:
            os << "{\n";
            for (const auto &f : fields) { os << pad << "  " << f.first << " = "; f.second.write(os, indent + 2); os << '\n'; }
            os << pad << '}';
            break;
        }
    }
    std::string dump() const { std::ostringstream os; write(os, 0); os << '\n'; return os.str(); }
};

namespace detail {

struct Reader {
    std::istream &in;
    std::string dir;   // directory of the file being read: '@' references are relative to it

    void skip_ws() {
        for (;;) {
            while (in.peek() != EOF && in.peek() <= ' ') in.get();
            if (in.peek() == '/') { int ch; while ((ch = in.get()) != EOF && ch != '\n') {} }
            else break;
        }
    }
    void expect(const std::string &text, const std::string &ctx) {
        for (char want : text) {
            const int ch = in.get();
            if (ch == EOF) throw Error("jsonx: unexpected end of file where '" + text + "' expected (" + ctx + ")");
            if ((char)ch != want) throw Error("jsonx: expected '" + text + "' (" + ctx + ")");
        }
    }
    std::string quoted(const std::string &ctx) {
        expect("\"", ctx);
        std::string s;
        while (in.peek() != EOF && in.peek() != '"') s += (char)in.get();
        expect("\"", ctx);
        return s;
    }
    static bool ident_char(int v) { return v == '_' || (v >= 'a' && v <= 'z') || (v >= 'A' && v <= 'Z') || (v >= '0' && v <= '9') || v == '-'; }
    static bool number_char(int v) { return v == '.' || (v >= '0' && v <= '9') || v == '-' || v == 'e' || v == 'E'; }
    std::pair<std::string, std::string> resolve(const std::string &rel) const {   // (file, its directory)
        const std::string file = dir.empty() ? rel : dir + "/" + rel;
        const size_t slash = file.rfind('/');
        return {file, slash == std::string::npos ? std::string() : file.substr(0, slash)};
    }

    int depth = 0;
    struct DepthGuard { int &d; explicit DepthGuard(int &x) : d(x) { ++d; } ~DepthGuard() { --d; } };

    Value value(const std::string &ctx) {
        DepthGuard guard(depth);
        if (depth > 200) throw Error("jsonx: nesting deeper than 200 levels (" + ctx + ")");
        skip_ws();
        Value v;
        const int c = in.peek();
        if (c == EOF) return v;                                   // an empty stream reads as "{}" (settings.cpp:192-195)
        if (c == '"') return Value::string(quoted(ctx));
        if (c == 'a') {
            expect("array", ctx);
            skip_ws();
            v.type = Value::ARRAY;
            if (in.peek() == '{') {
                expect("{", ctx);
                skip_ws();
                while (in.peek() != EOF && in.peek() != '}') { v.items.push_back(value(ctx + "/" + std::to_string(v.items.size()))); skip_ws(); }
                expect("}", ctx);
            } else {
                expect("@", ctx);
                skip_ws();
                const auto p = resolve(quoted(ctx));
                std::ifstream f(p.first);
                if (!f) throw Error("jsonx: cannot open '" + p.first + "' (" + ctx + ")");
                Reader sub{f, p.second, depth};
                sub.skip_ws();
                while (f.peek() != EOF) { v.items.push_back(sub.value(ctx + "/" + std::to_string(v.items.size()))); sub.skip_ws(); }
            }
            return v;
        }
        if (c == 'm' || c == 's') {
            const bool sparse = c == 's';
            if (sparse) { expect("sparse", ctx); skip_ws(); }
            expect("matrix", ctx);
            skip_ws();
            expect("(", ctx);
            v.type = Value::MATRIX;
            v.rows = (int)value(ctx + "/#rows").as_int();
            v.cols = (int)value(ctx + "/#cols").as_int();
            skip_ws();
            expect(")", ctx);
            skip_ws();
            expect("{", ctx);
            if (v.rows < 0 || v.cols < 0) throw Error("jsonx: negative matrix size (" + ctx + ")");
            if ((long long)v.rows * v.cols > (1LL << 24)) throw Error("jsonx: matrix larger than 2^24 cells (" + ctx + ")");
            v.items.resize((size_t)v.rows * v.cols);
            if (!sparse) {
                for (Value &cell : v.items) cell = value(ctx);
                skip_ws();
            } else {
                skip_ws();
                while (in.peek() != EOF && in.peek() != '}') {
                    const long long r = value(ctx).as_int(), cc = value(ctx).as_int();
                    if (r < 0 || r >= v.rows || cc < 0 || cc >= v.cols) throw Error("jsonx: sparse matrix index out of bounds (" + ctx + ")");
                    v.items[(size_t)r * v.cols + cc] = value(ctx);
                    skip_ws();
                }
            }
            expect("}", ctx);
            return v;
        }
        if (c == '{') {
            expect("{", ctx);
            skip_ws();
            while (in.peek() != EOF && in.peek() != '}') {
                std::string id;
                while (in.peek() != EOF && ident_char(in.peek())) id += (char)in.get();
                skip_ws();
                expect("=", ctx + "/" + id);
                Value field = value(ctx + "/" + id);
                skip_ws();
                if (!v.find(id)) v.fields.emplace_back(id, std::move(field));   // the first occurrence wins
            }
            expect("}", ctx);
            return v;
        }
        if (c == '@') {
            expect("@", ctx);
            const auto p = resolve(quoted(ctx));
            std::ifstream f(p.first);
            if (!f) throw Error("jsonx: cannot open '" + p.first + "' (" + ctx + ")");
            Reader sub{f, p.second, depth};   // a file that includes itself runs into the depth limit
            return sub.value(ctx);
        }
        std::string num;
        while (in.peek() != EOF && number_char(in.peek())) num += (char)in.get();
        if (num.empty()) throw Error(std::string("jsonx: unexpected character '") + (char)c + "' (" + ctx + ")");
        return Value::string(num);
    }
};

}  // namespace detail

inline Value parse(std::istream &in, const std::string &dir = "") {
    detail::Reader r{in, dir, 0};
    return r.value("");
}
inline Value parse_string(const std::string &text, const std::string &dir = "") {
    std::istringstream in(text);
    return parse(in, dir);
}
inline Value parse_file(const std::string &path) {
    std::ifstream f(path);
    if (!f) throw Error("jsonx: cannot open '" + path + "'");
    const size_t slash = path.rfind('/');
    return parse(f, slash == std::string::npos ? std::string() : path.substr(0, slash));
}

}  // namespace jsonx
