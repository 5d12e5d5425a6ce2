// host_sanitize.cpp -- the header-only host pieces (jsonx reader/writer, interleaver maps) under AddressSanitizer and
// UndefinedBehaviorSanitizer: malformed and hostile inputs must end in jsonx::Error / `false`, never in a crash.
// Built and run by tests/test_host_cpu.py (CPU only; GPU sanitizers are not available on this pool).
#include <cstdio>
#include <random>
#include <string>
#include <vector>

#include "ldpc/encoder.h"
#include "ldpc/interleaver.h"
#include "ldpc/jsonx.h"

static int failures = 0;
#define CHECK(c) do { if (!(c)) { printf("CHECK failed: %s (line %d)\n", #c, __LINE__); ++failures; } } while (0)

static bool parses(const std::string &text) {
    try { (void)jsonx::parse_string(text).dump(); return true; } catch (const jsonx::Error &) { return false; }
}

int main() {
    // ---- well-formed
    CHECK(parses("{ a = 1 b = \"x y\" c = array { 1 2.5 -3e-2 } m = matrix (2 2) { 1 2 3 4 } s = sparse matrix (2 3) { 0 0 1  1 2 -1 } }"));
    CHECK(parses(""));                       // empty stream == {}
    CHECK(parses("// only a comment\n"));
    CHECK(parses("{ a = { b = { c = array { array { } array { { } } } } } }"));
    // ---- malformed: every one must throw, not crash
    const char *bad[] = {"{", "{ a", "{ a = ", "{ a = \"unterminated", "{ a = array { 1 2", "{ a = array @", "{ a = array @\"/nonexistent/file.jsonx\" }",
                         "{ a = @\"/nonexistent/file.jsonx\" }", "{ m = matrix (2 2) { 1 2 3 } }", "{ m = matrix (-1 2) { } }", "{ m = matrix (2", "{ m = matrix 2 2) { } }",
                         "{ m = matrix (100000 100000) { } }", "{ s = sparse matrix (2 2) { 5 0 1 } }", "{ s = sparse matrix (2 2) { 0 } }", "{ a 3 }",
                         "{ a = ? }", "}", "array", "arra { }", "{ a = matrix (1e400 2) { } }", "{ a = matrix (x 2) { } }"};
    for (const char *t : bad) CHECK(!parses(t));
    // ---- conversions and select
    {
        const jsonx::Value v = jsonx::parse_string("{ defaults = { d = 7 } n = 12 x = 1e3 s = \"str\" arr = array { 1 2 } }");
        CHECK(v.select("n").as_int() == 12 && v.select("x").as_int() == 1000 && v.select("d").as_int() == 7);
        CHECK(v.select("arr/1").as_double() == 2.0);
        bool threw = false;
        try { (void)v.select("s").as_int(); } catch (const jsonx::Error &) { threw = true; }
        CHECK(threw);
        threw = false;
        try { (void)v.select("arr/2"); } catch (const jsonx::Error &) { threw = true; }
        CHECK(threw);
        CHECK(jsonx::Value::number(0.1).as_double() == 0.1 && jsonx::Value::number(1e-300).as_double() == 1e-300);
    }
    // ---- random byte soup around a valid skeleton
    std::mt19937 rng(5);
    const std::string seed_text = "{ a = array { 1 2 3 } m = matrix (2 2) { 1 2 3 4 } r = { s = \"q\" } }";
    for (int it = 0; it < 20000; ++it) {
        std::string t = seed_text;
        const int edits = 1 + (int)(rng() % 4);
        for (int e = 0; e < edits; ++e) {
            const size_t pos = rng() % t.size();
            switch (rng() % 3) {
            case 0: t[pos] = (char)(rng() % 96 + 32); break;
            case 1: t.erase(pos, 1 + rng() % 3); break;
            default: t.insert(pos, 1, "{}()=\"@/ am1-e"[rng() % 15]); break;
            }
            if (t.empty()) t = "{";
        }
        (void)parses(t);   // either outcome is fine; sanitizers watch the rest
    }
    // ---- interleaver: every mode on random small codes is a pair of mutually inverse permutations or a clean refusal
    for (int it = 0; it < 3000; ++it) {
        const int b = 1 + (int)(rng() % 6), c = b + 1 + (int)(rng() % 10), M = 1 + (int)(rng() % 12), h = 1 + (int)(rng() % 4), mode = (int)(rng() % 6);
        std::vector<int> hd((size_t)b * c);
        for (int &x : hd) x = (rng() % 3) ? -1 : (int)(rng() % M);
        ldpc::Interleaver il;
        std::string err;
        const int bs = (int)(rng() % 40) - 2, st = (int)(rng() % 9) - 1;
        if (ldpc::build_interleaver(b, c, M, h, mode, bs, st, hd.data(), il, err)) {
            const int N = c * M;
            CHECK((int)il.direct.size() == N && (int)il.inverse.size() == N);
            for (int i = 0; i < N; ++i) {
                CHECK(il.direct[(size_t)i] >= 0 && il.direct[(size_t)i] < N);
                CHECK(il.inverse[(size_t)il.direct[(size_t)i]] == i);
            }
        } else {
            CHECK(!err.empty());
        }
    }
    // ---- encoder: arbitrary base matrices either encode to a word that satisfies every check, or are refused
    int encoded = 0;
    for (int it = 0; it < 4000; ++it) {
        const int b = 1 + (int)(rng() % 6), c = b + 1 + (int)(rng() % 8), M = 1 + (int)(rng() % 9);
        std::vector<int> mx((size_t)b * c, -1);
        const bool structured = (rng() % 2) != 0;
        for (int i = 0; i < b; ++i)
            for (int j = 0; j < c; ++j) {
                if (structured && j < b) mx[(size_t)(i * c + j)] = (j == i || j + 1 == i) ? 0 : -1;      // dual diagonal
                else mx[(size_t)(i * c + j)] = (rng() % 3) ? -1 : (int)(rng() % M);
            }
        if (structured && b > 1) { mx[(size_t)(b - 1)] = M > 1 ? 1 : 0; mx[(size_t)((b / 2) * c + b - 1)] = 0; }
        std::vector<unsigned char> info((size_t)(c - b) * M + 64), cw;
        for (auto &x : info) x = (unsigned char)(rng() & 1);
        const int rc = ldpc::encode(mx.data(), b, c, M, info.data(), cw);
        if (rc == 0) {
            ++encoded;
            for (int i = 0; i < b; ++i)
                for (int h = 0; h < M; ++h) {
                    unsigned char sy = 0;
                    for (int j = 0; j < c; ++j)
                        if (mx[(size_t)(i * c + j)] >= 0) sy ^= cw[(size_t)(j * M + (h + mx[(size_t)(i * c + j)]) % M)];
                    CHECK(sy == 0);
                }
        }
    }
    CHECK(encoded > 100);
    printf("%s\n", failures ? "FAILED" : "ok");
    return failures ? 1 : 0;
}
