// Minimal JSON DOM for the reference's scene / technique configuration files (the reference uses
// nlohmann::json, rc/json/json.hpp; only the subset its configs need is implemented: objects,
// arrays, strings, numbers, booleans, null).
#pragma once
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <cerrno>
#include <string>
#include <utility>
#include <vector>

namespace evplp {

struct JsonError : std::runtime_error { using std::runtime_error::runtime_error; };

class Json {
public:
    enum Type { Null, Bool, Number, String, Array, Object };
    Type type = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    static Json parse(const std::string &text) {
        size_t i = 0; Json v = parse_value(text, i); skip_ws(text, i);
        if (i != text.size()) throw JsonError("trailing characters after JSON value at offset " + std::to_string(i));
        return v;
    }
    bool is_null() const { return type == Null; }
    bool is_object() const { return type == Object; }
    bool is_array() const { return type == Array; }
    int kind() const { return (int)type; }
    const Json *find(const std::string &key) const {
        if (type != Object) return nullptr;
        for (auto &kv : obj) if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool has(const std::string &key) const { return find(key) != nullptr; }
    // like nlohmann's operator[] on a const object, but a missing REQUIRED key is an error with its name
    const Json &at(const std::string &key) const {
        const Json *v = find(key);
        if (!v) throw JsonError("missing required key \"" + key + "\"");
        return *v;
    }
    const Json &at(size_t i) const { if (type != Array || i >= arr.size()) throw JsonError("array index out of range"); return arr[i]; }
    // nlohmann::json 2.1.1 (the reference's reader) semantics, pinned by tests/golden/json_pins.json: a scalar has size 1, null 0;
    // a boolean converts to the arithmetic types (0 / 1), nothing else converts to bool or string
    size_t size() const { return type == Array ? arr.size() : type == Object ? obj.size() : type == Null ? 0 : 1; }
    double as_number(const char *what = "value") const {
        if (type == Bool) return b ? 1.0 : 0.0;
        if (type != Number) throw JsonError(std::string(what) + ": expected a number");
        return num;
    }
    float as_float(const char *what = "value") const { return (float)as_number(what); }
    long long as_int(const char *what = "value") const { return (long long)as_number(what); }
    bool as_bool(const char *what = "value") const { if (type != Bool) throw JsonError(std::string(what) + ": expected a boolean"); return b; }
    const std::string &as_string(const char *what = "value") const { if (type != String) throw JsonError(std::string(what) + ": expected a string"); return str; }
    // shallow merge of another object over this one
    void merge(const Json &o) {
        if (o.type != Object) return;
        if (type != Object) { *this = o; return; }
        for (auto &kv : o.obj) {
            bool done = false;
            for (auto &mine : obj) if (mine.first == kv.first) {
                if (mine.second.type == Object && kv.second.type == Object) mine.second.merge(kv.second); else mine.second = kv.second;
                done = true; break;
            }
            if (!done) obj.push_back(kv);
        }
    }
    std::string dump(int indent = 4, int level = 0) const {
        std::string pad((size_t)indent * (level + 1), ' '), padc((size_t)indent * level, ' ');
        switch (type) {
        case Null: return "null";
        case Bool: return b ? "true" : "false";
        case Number: { char buf[64]; if (num == (long long)num && std::abs(num) < 1e15) snprintf(buf, sizeof buf, "%lld", (long long)num); else snprintf(buf, sizeof buf, "%.9g", num); return buf; }
        case String: { std::string s = "\""; for (char c : str) { if (c == '"' || c == '\\') s += '\\'; s += c; } return s + "\""; }
        case Array: { if (arr.empty()) return "[]"; std::string s = "[\n"; for (size_t i = 0; i < arr.size(); i++) s += pad + arr[i].dump(indent, level + 1) + (i + 1 < arr.size() ? ",\n" : "\n"); return s + padc + "]"; }
        case Object: { if (obj.empty()) return "{}"; std::string s = "{\n"; for (size_t i = 0; i < obj.size(); i++) { Json k; k.type = String; k.str = obj[i].first; s += pad + k.dump() + ": " + obj[i].second.dump(indent, level + 1) + (i + 1 < obj.size() ? ",\n" : "\n"); } return s + padc + "}"; }
        }
        return "";
    }
    static Json number(double v) { Json j; j.type = Number; j.num = v; return j; }
    static Json string(const std::string &s) { Json j; j.type = String; j.str = s; return j; }
    static Json boolean(bool v) { Json j; j.type = Bool; j.b = v; return j; }
    static Json object() { Json j; j.type = Object; return j; }
    void set(const std::string &k, const Json &v) { type = Object; for (auto &kv : obj) if (kv.first == k) { kv.second = v; return; } obj.emplace_back(k, v); }

private:
    static void skip_ws(const std::string &s, size_t &i) { while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) i++; }
    static Json parse_value(const std::string &s, size_t &i) {
        skip_ws(s, i);
        if (i >= s.size()) throw JsonError("unexpected end of JSON");
        char c = s[i];
        Json v;
        if (c == '{') {
            v.type = Object; i++; skip_ws(s, i);
            if (i < s.size() && s[i] == '}') { i++; return v; }
            for (;;) {
                skip_ws(s, i);
                if (i >= s.size() || s[i] != '"') throw JsonError("expected string key at offset " + std::to_string(i));
                std::string k = parse_string(s, i);
                skip_ws(s, i);
                if (i >= s.size() || s[i] != ':') throw JsonError("expected ':' at offset " + std::to_string(i));
                i++;
                { Json item = parse_value(s, i); bool dup = false;                      // a repeated key keeps its LAST value
                  for (auto &kv : v.obj) if (kv.first == k) { kv.second = item; dup = true; break; }
                  if (!dup) v.obj.emplace_back(k, std::move(item)); }
                skip_ws(s, i);
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == '}') { i++; break; }
                throw JsonError("expected ',' or '}' at offset " + std::to_string(i));
            }
        } else if (c == '[') {
            v.type = Array; i++; skip_ws(s, i);
            if (i < s.size() && s[i] == ']') { i++; return v; }
            for (;;) {
                v.arr.push_back(parse_value(s, i));
                skip_ws(s, i);
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == ']') { i++; break; }
                throw JsonError("expected ',' or ']' at offset " + std::to_string(i));
            }
        } else if (c == '"') { v.type = String; v.str = parse_string(s, i); }
        else if (!s.compare(i, 4, "true")) { v.type = Bool; v.b = true; i += 4; }
        else if (!s.compare(i, 5, "false")) { v.type = Bool; v.b = false; i += 5; }
        else if (!s.compare(i, 4, "null")) { v.type = Null; i += 4; }
        else {
            // RFC 7159 number: -?(0|[1-9][0-9]*)(\.[0-9]+)?([eE][+-]?[0-9]+)?  (strtod alone also takes "inf", "0x10", ".5", "+1")
            size_t k = i;
            auto digit = [&](size_t q) { return q < s.size() && s[q] >= '0' && s[q] <= '9'; };
            if (k < s.size() && s[k] == '-') k++;
            if (!digit(k)) throw JsonError("unexpected character '" + std::string(1, c) + "' at offset " + std::to_string(i));
            if (s[k] == '0') k++; else while (digit(k)) k++;
            if (k < s.size() && s[k] == '.') { k++; if (!digit(k)) throw JsonError("digit expected after '.' at offset " + std::to_string(k)); while (digit(k)) k++; }
            if (k < s.size() && (s[k] == 'e' || s[k] == 'E')) {
                k++; if (k < s.size() && (s[k] == '+' || s[k] == '-')) k++;
                if (!digit(k)) throw JsonError("digit expected in exponent at offset " + std::to_string(k));
                while (digit(k)) k++;
            }
            // a token without fraction and exponent is an INTEGER (so "-0" is 0, not -0.0); beyond 64 bits it is read as a float
            const std::string tok = s.substr(i, k - i);
            bool integer = tok.find_first_of(".eE") == std::string::npos;
            if (integer) { errno = 0; char *end = nullptr; const long long iv = std::strtoll(tok.c_str(), &end, 10); if (errno == ERANGE) integer = false; else v.num = (double)iv; }
            if (!integer) v.num = std::strtod(tok.c_str(), nullptr);
            v.type = Number; i = k;
        }
        return v;
    }
    static void put_utf8(std::string &out, unsigned cp) {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
        else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
    }
    static unsigned hex4(const std::string &s, size_t at) {
        if (at + 4 > s.size()) throw JsonError("truncated \\u escape");
        unsigned v = 0;
        for (size_t k = 0; k < 4; k++) {
            const char h = s[at + k]; v <<= 4;
            if (h >= '0' && h <= '9') v |= (unsigned)(h - '0'); else if (h >= 'a' && h <= 'f') v |= (unsigned)(h - 'a' + 10); else if (h >= 'A' && h <= 'F') v |= (unsigned)(h - 'A' + 10);
            else throw JsonError("bad \\u escape at offset " + std::to_string(at));
        }
        return v;
    }
    static std::string parse_string(const std::string &s, size_t &i) {
        std::string out; i++;
        while (i < s.size() && s[i] != '"') {
            const unsigned char ch = (unsigned char)s[i];
            if (ch < 0x20) throw JsonError("control character in string at offset " + std::to_string(i));
            if (ch == '\\') {
                if (i + 1 >= s.size()) break;
                const char e = s[++i];
                switch (e) {
                case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break; case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                case '"': case '\\': case '/': out += e; break;
                case 'u': {
                    unsigned cp = hex4(s, i + 1); i += 4;
                    if (cp >= 0xD800 && cp <= 0xDBFF) {          // high surrogate: a low one must follow
                        if (i + 2 < s.size() && s[i + 1] == '\\' && s[i + 2] == 'u') {
                            const unsigned lo = hex4(s, i + 3);
                            if (lo < 0xDC00 || lo > 0xDFFF) throw JsonError("missing low surrogate at offset " + std::to_string(i));
                            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00); i += 6;
                        } else throw JsonError("missing low surrogate at offset " + std::to_string(i));
                    } else if (cp >= 0xDC00 && cp <= 0xDFFF) throw JsonError("lone low surrogate at offset " + std::to_string(i));
                    put_utf8(out, cp);
                    break;
                }
                default: throw JsonError(std::string("bad escape '\\") + e + "' at offset " + std::to_string(i));
                }
                i++;
            } else out += s[i++];
        }
        if (i >= s.size()) throw JsonError("unterminated string");
        i++;
        return out;
    }
};

} // namespace evplp
