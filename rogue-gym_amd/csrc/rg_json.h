// rg_json.h -- the small JSON reader / writer of the config parser (serde_json's role in GameConfig::from_json / to_json,
// core/src/lib.rs:144-149).  Header-only; used by rg_config.cpp and rg_items.cpp.
#pragma once
#include <cctype>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

namespace rgjson {

struct JVal {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    bool neg = false, is_int = true;
    unsigned __int128 mag = 0; // integer magnitude
    double d = 0;
    std::string s;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal *get(const char *k) const {
        for (auto &kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
};

struct Parser {
    const char *p, *end;
    std::string err;
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
    bool fail(const std::string &m) { if (err.empty()) err = m; return false; }
    bool parse(JVal &v) {
        ws();
        if (p >= end) return fail("EOF while parsing a value");
        char c = *p;
        if (c == '{') {
            v.kind = JVal::Obj; p++; ws();
            if (p < end && *p == '}') { p++; return true; }
            for (;;) {
                ws();
                JVal k;
                if (p >= end || *p != '"') return fail("key must be a string");
                if (!parse(k)) return false;
                ws();
                if (p >= end || *p != ':') return fail("expected `:`");
                p++;
                JVal x;
                if (!parse(x)) return false;
                v.obj.emplace_back(k.s, std::move(x));
                ws();
                if (p < end && *p == ',') { p++; ws(); if (p < end && *p == '}') return fail("trailing comma"); continue; }
                if (p < end && *p == '}') { p++; return true; }
                return fail("expected `,` or `}`");
            }
        }
        if (c == '[') {
            v.kind = JVal::Arr; p++; ws();
            if (p < end && *p == ']') { p++; return true; }
            for (;;) {
                JVal x;
                if (!parse(x)) return false;
                v.arr.push_back(std::move(x));
                ws();
                if (p < end && *p == ',') { p++; ws(); if (p < end && *p == ']') return fail("trailing comma"); continue; }
                if (p < end && *p == ']') { p++; return true; }
                return fail("expected `,` or `]`");
            }
        }
        if (c == '"') {
            v.kind = JVal::Str; p++;
            while (p < end && *p != '"') {
                if (*p == '\\' && p + 1 < end) { p++; char e = *p++; v.s += (e == 'n' ? '\n' : e == 't' ? '\t' : e); }
                else v.s += *p++;
            }
            if (p >= end) return fail("EOF while parsing a string");
            p++;
            return true;
        }
        if (!strncmp(p, "true", 4) && end - p >= 4) { v.kind = JVal::Bool; v.b = true; p += 4; return true; }
        if (!strncmp(p, "false", 5) && end - p >= 5) { v.kind = JVal::Bool; v.b = false; p += 5; return true; }
        if (!strncmp(p, "null", 4) && end - p >= 4) { v.kind = JVal::Null; p += 4; return true; }
        if (c == '-' || isdigit((unsigned char)c)) {
            v.kind = JVal::Num;
            const char *s = p;
            if (c == '-') { v.neg = true; p++; }
            if (p >= end || !isdigit((unsigned char)*p)) return fail("invalid number");
            while (p < end && isdigit((unsigned char)*p)) { v.mag = v.mag * 10 + (unsigned)(*p - '0'); p++; }
            if (p < end && (*p == '.' || *p == 'e' || *p == 'E')) {
                v.is_int = false;
                while (p < end && (isdigit((unsigned char)*p) || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) p++;
                v.d = strtod(std::string(s, p).c_str(), nullptr);
            }
            return true;
        }
        return fail(std::string("expected value, found `") + c + "`");
    }
};


inline std::string quote(const std::string &s) {
    std::string o = "\"";
    for (char ch : s) {
        if (ch == '"' || ch == '\\') { o += '\\'; o += ch; }
        else if (ch == '\n') o += "\\n";
        else if (ch == '\t') o += "\\t";
        else o += ch;
    }
    return o + "\"";
}
inline std::string u128_str(unsigned __int128 v) {
    if (v == 0) return "0";
    std::string s;
    while (v) { s.insert(s.begin(), (char)('0' + (int)(v % 10))); v /= 10; }
    return s;
}
// re-serialise a parsed value (used for sections the stepper carries through untouched, e.g. `keymap`)
inline std::string to_string(const JVal &v) {
    switch (v.kind) {
    case JVal::Null: return "null";
    case JVal::Bool: return v.b ? "true" : "false";
    case JVal::Num:
        if (!v.is_int) { char b[64]; snprintf(b, sizeof b, "%.17g", v.d); return b; }
        return std::string(v.neg && v.mag ? "-" : "") + u128_str(v.mag);
    case JVal::Str: return quote(v.s);
    case JVal::Arr: { std::string s = "["; for (size_t i = 0; i < v.arr.size(); i++) s += (i ? ", " : "") + to_string(v.arr[i]); return s + "]"; }
    case JVal::Obj: { std::string s = "{"; for (size_t i = 0; i < v.obj.size(); i++) s += (i ? ", " : "") + quote(v.obj[i].first) + ": " + to_string(v.obj[i].second); return s + "}"; }
    }
    return "null";
}

}  // namespace rgjson
