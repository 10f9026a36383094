// oracle/_ref/libref_pin.so, second translation unit (no -Dexception here): the reference's vendored nlohmann::json 2.1.1
// (reflectcuts/json/json.hpp; main.cpp:105-121 and rtcomphoton.h:107-223 read the scene files with it) behind one
// extern "C" probe, used by tests/golden/make_golden.py to pin the product's own JSON reader.
// This file contains no reference code; it calls it.
#include "json/json.hpp"

#include <cstring>
#include <string>

extern "C" {

// nlohmann::json 2.1.1 exactly as the technique blocks use it: a const json&, find() for optional keys, operator[] and the
// implicit conversion `int v = json["k"]` / `float` / `bool` / `std::string` (rtcomphoton.h:114-222).  `path` = keys
// separated by '/', decimal indices into arrays.  want: 0 int, 1 float, 2 bool, 3 string, 4 size(), 5 kind (0 null, 1 bool,
// 2 number, 3 string, 4 array, 5 object).  Returns 0 ok, 1 parse error, 2 key missing / index out of range, 3 conversion error.
int ref_json_query(const char *text, const char *path, int want, double *num, char *str, int cap) {
    nlohmann::json root;
    try { root = nlohmann::json::parse(std::string(text)); } catch (...) { return 1; }
    const nlohmann::json *j = &root;
    std::string p(path);
    size_t at = 0;
    while (at < p.size()) {
        size_t e = p.find('/', at); if (e == std::string::npos) e = p.size();
        const std::string key = p.substr(at, e - at);
        at = e + 1;
        if (j->is_array()) {
            const size_t i = (size_t)std::stoul(key);
            if (i >= j->size()) return 2;
            j = &(*j)[i];
        } else if (j->is_object()) {
            auto it = j->find(key);
            if (it == j->end()) return 2;
            j = &*it;
        } else return 2;
    }
    try {
        switch (want) {
        case 0: { int v = *j; *num = v; break; }
        case 1: { float v = *j; *num = v; break; }
        case 2: { bool v = *j; *num = v ? 1.0 : 0.0; break; }
        case 3: { std::string v = *j; if ((int)v.size() + 1 > cap) return 3; std::memcpy(str, v.data(), v.size()); str[v.size()] = 0; *num = (double)v.size(); break; }
        case 4: *num = (double)j->size(); break;
        default: *num = j->is_null() ? 0 : j->is_boolean() ? 1 : j->is_number() ? 2 : j->is_string() ? 3 : j->is_array() ? 4 : 5; break;
        }
    } catch (...) { return 3; }
    return 0;
}
}
