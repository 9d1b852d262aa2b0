// zkc_json.h -- a strict JSON reader for the documents that cross the C ABI as text (host only, no GPU types): verification_key.json, proof.json, signals.json
// (zk_census_test.go:110-122: prover.ParseProof / proof.Verify go through encoding/json; snarkjs through JSON.parse) and the circuit inputs (zk_census_test.go:85-89:
// prover.Prove's third argument is the file image of inputs_example.json).  RFC 8259, nothing more: one value, whitespace around it and nothing else; objects, arrays,
// strings, numbers, true / false / null; no trailing commas, no comments, no NaN / Infinity, no raw control characters in strings, escapes \" \\ \/ \b \f \n \r \t \uXXXX
// only, strings valid UTF-8.  Everything Go's encoding/json, JavaScript's JSON.parse and Python's json.loads reject is rejected here (tests/host/parse_asan.cc fuzzes the
// verify boundary against exactly that).  Numbers keep their text: the callers decide what a number may be (a circuit input may be an integer literal of any length).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

namespace zkc { namespace json {
struct Value {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool b = false;
    std::string s;                                               // String: the decoded text (UTF-8); Number: the literal as written
    std::vector<Value> a;                                        // Array
    std::vector<std::pair<std::string, Value>> o;                // Object, in document order (duplicate names are kept: find() returns the LAST, as JSON.parse and Go do)
    const Value* find(const char* name) const { const Value* r = nullptr; for (auto& kv : o) if (kv.first == name) r = &kv.second; return r; }
};
struct Parser {
    const char* p; const char* e; std::string err; int depth = 0;
    static constexpr int MAX_DEPTH = 64;
    bool fail(const char* m) { if (err.empty()) err = m; return false; }
    void ws() { while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
    static int hexv(char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; }
    static void utf8(std::string& out, uint32_t cp) {
        if (cp < 0x80) out.push_back((char)cp);
        else if (cp < 0x800) { out.push_back((char)(0xc0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 63))); }
        else if (cp < 0x10000) { out.push_back((char)(0xe0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 63))); out.push_back((char)(0x80 | (cp & 63))); }
        else { out.push_back((char)(0xf0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 63))); out.push_back((char)(0x80 | ((cp >> 6) & 63))); out.push_back((char)(0x80 | (cp & 63))); }
    }
    bool string(std::string& out) {                              // at the opening quote
        p++;
        for (;;) {
            if (p >= e) return fail("JSON: unterminated string");
            const unsigned char c = (unsigned char)*p;
            if (c == '"') { p++; return true; }
            if (c < 0x20) return fail("JSON: control character in a string");
            if (c == '\\') {
                if (p + 1 >= e) return fail("JSON: unterminated escape");
                const char x = p[1]; p += 2;
                switch (x) {
                    case '"': out.push_back('"'); break; case '\\': out.push_back('\\'); break; case '/': out.push_back('/'); break;
                    case 'b': out.push_back('\b'); break; case 'f': out.push_back('\f'); break; case 'n': out.push_back('\n'); break;
                    case 'r': out.push_back('\r'); break; case 't': out.push_back('\t'); break;
                    case 'u': {
                        if (e - p < 4) return fail("JSON: short \\u escape");
                        uint32_t cp = 0; for (int i = 0; i < 4; i++) { const int h = hexv(p[i]); if (h < 0) return fail("JSON: bad \\u escape"); cp = cp * 16 + (uint32_t)h; }
                        p += 4;
                        if (cp >= 0xd800 && cp < 0xdc00 && e - p >= 6 && p[0] == '\\' && p[1] == 'u') {      // a surrogate pair
                            uint32_t lo = 0; bool ok = true; for (int i = 0; i < 4; i++) { const int h = hexv(p[2 + i]); if (h < 0) { ok = false; break; } lo = lo * 16 + (uint32_t)h; }
                            if (ok && lo >= 0xdc00 && lo < 0xe000) { cp = 0x10000 + ((cp - 0xd800) << 10) + (lo - 0xdc00); p += 6; }
                        }
                        utf8(out, cp); break;                    // (a lone surrogate is passed through as the parsers named above do)
                    }
                    default: return fail("JSON: bad escape");
                }
                continue;
            }
            if (c < 0x80) { out.push_back((char)c); p++; continue; }
            // a UTF-8 sequence, checked: length by the lead byte, continuation bytes, no overlong forms, no code points past U+10FFFF or in the surrogate range
            const int n = c >= 0xf0 ? 4 : c >= 0xe0 ? 3 : c >= 0xc2 ? 2 : 0;
            if (n == 0 || c > 0xf4 || e - p < n) return fail("JSON: invalid UTF-8");
            uint32_t cp = n == 2 ? c & 0x1f : n == 3 ? c & 0x0f : c & 0x07;
            for (int i = 1; i < n; i++) { const unsigned char d = (unsigned char)p[i]; if ((d & 0xc0) != 0x80) return fail("JSON: invalid UTF-8"); cp = (cp << 6) | (d & 0x3f); }
            if ((n == 3 && cp < 0x800) || (n == 4 && (cp < 0x10000 || cp > 0x10ffff)) || (cp >= 0xd800 && cp < 0xe000)) return fail("JSON: invalid UTF-8");
            out.append(p, (size_t)n); p += n;
        }
    }
    bool number(std::string& out) {
        const char* s = p;
        if (p < e && *p == '-') p++;
        if (p >= e) return fail("JSON: bad number");
        if (*p == '0') p++;
        else if (*p >= '1' && *p <= '9') { while (p < e && *p >= '0' && *p <= '9') p++; }
        else return fail("JSON: bad number");
        if (p < e && *p == '.') { p++; if (p >= e || *p < '0' || *p > '9') return fail("JSON: bad number"); while (p < e && *p >= '0' && *p <= '9') p++; }
        if (p < e && (*p == 'e' || *p == 'E')) { p++; if (p < e && (*p == '+' || *p == '-')) p++; if (p >= e || *p < '0' || *p > '9') return fail("JSON: bad number"); while (p < e && *p >= '0' && *p <= '9') p++; }
        out.assign(s, (size_t)(p - s)); return true;
    }
    bool lit(const char* w) { const size_t n = strlen(w); if ((size_t)(e - p) < n || memcmp(p, w, n)) return fail("JSON: unexpected token"); p += n; return true; }
    bool value(Value& v) {
        ws();
        if (p >= e) return fail("JSON: unexpected end");
        if (++depth > MAX_DEPTH) return fail("JSON: nested too deeply");
        bool ok = true;
        switch (*p) {
            case '{': {
                v.type = Value::Object; p++; ws();
                if (p < e && *p == '}') { p++; break; }
                for (;;) {
                    ws(); if (p >= e || *p != '"') { ok = fail("JSON: expected a member name"); break; }
                    std::string name; if (!string(name)) { ok = false; break; }
                    ws(); if (p >= e || *p != ':') { ok = fail("JSON: expected ':'"); break; }
                    p++; v.o.emplace_back(std::move(name), Value());
                    if (!value(v.o.back().second)) { ok = false; break; }
                    ws(); if (p < e && *p == ',') { p++; continue; }
                    if (p < e && *p == '}') { p++; break; }
                    ok = fail("JSON: expected ',' or '}'"); break;
                }
                break;
            }
            case '[': {
                v.type = Value::Array; p++; ws();
                if (p < e && *p == ']') { p++; break; }
                for (;;) {
                    v.a.emplace_back();
                    if (!value(v.a.back())) { ok = false; break; }
                    ws(); if (p < e && *p == ',') { p++; continue; }
                    if (p < e && *p == ']') { p++; break; }
                    ok = fail("JSON: expected ',' or ']'"); break;
                }
                break;
            }
            case '"': v.type = Value::String; ok = string(v.s); break;
            case 't': v.type = Value::Bool; v.b = true; ok = lit("true"); break;
            case 'f': v.type = Value::Bool; v.b = false; ok = lit("false"); break;
            case 'n': v.type = Value::Null; ok = lit("null"); break;
            default: v.type = Value::Number; ok = number(v.s); break;
        }
        depth--;
        return ok;
    }
};
// the whole text must be ONE JSON value (a UTF-8 byte order mark in front is not JSON); false: err says why
inline bool parse(const char* text, size_t len, Value& out, std::string& err) {
    Parser ps{text, text + len, std::string(), 0};
    if (!ps.value(out)) { err = ps.err; return false; }
    ps.ws();
    if (ps.p != ps.e) { err = "JSON: text after the document"; return false; }
    return true;
}
}}  // namespace zkc::json
