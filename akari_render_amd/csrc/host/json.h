// json.h -- a small recursive-descent JSON reader (objects keep key order sorted, like the reference's
// BTreeMap-backed `Collection`, crates/akari_scenegraph/src/lib.rs:71).
#pragma once
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace akr {

struct JsonValue;
using JsonPtr = std::shared_ptr<JsonValue>;

struct JsonValue {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<JsonPtr> arr;
    std::map<std::string, JsonPtr> obj;  // std::map iterates in byte-wise key order == BTreeMap<String, _>

    bool is_null() const { return type == Null; }
    bool has(const std::string& k) const { return type == Object && obj.count(k) && !obj.at(k)->is_null(); }
    const JsonValue& at(const std::string& k) const {
        if (type != Object) throw std::runtime_error("JSON: expected an object while looking up '" + k + "'");
        auto it = obj.find(k);
        if (it == obj.end()) throw std::runtime_error("JSON: missing key '" + k + "'");
        return *it->second;
    }
    const JsonValue& at(size_t i) const {
        if (type != Array || i >= arr.size()) throw std::runtime_error("JSON: array index out of range");
        return *arr[i];
    }
    double as_number() const {
        if (type != Number) throw std::runtime_error("JSON: expected a number");
        return num;
    }
    float as_f32() const { return (float)as_number(); }
    const std::string& as_string() const {
        if (type != String) throw std::runtime_error("JSON: expected a string");
        return str;
    }
    bool as_bool() const {
        if (type != Bool) throw std::runtime_error("JSON: expected a bool");
        return b;
    }
};

class JsonParser {
   public:
    static JsonPtr parse(const std::string& text) {
        JsonParser p(text);
        p.skip_ws();
        JsonPtr v = p.value();
        p.skip_ws();
        if (p.pos_ != p.s_.size()) p.fail("trailing characters");
        return v;
    }

   private:
    explicit JsonParser(const std::string& s) : s_(s) {}
    [[noreturn]] void fail(const std::string& what) { throw std::runtime_error("JSON parse error at byte " + std::to_string(pos_) + ": " + what); }
    void skip_ws() {
        while (pos_ < s_.size() && (s_[pos_] == ' ' || s_[pos_] == '\n' || s_[pos_] == '\t' || s_[pos_] == '\r')) pos_++;
    }
    char peek() { return pos_ < s_.size() ? s_[pos_] : '\0'; }
    void expect(char c) {
        if (peek() != c) fail(std::string("expected '") + c + "'");
        pos_++;
    }
    // serde_json, which reads these files in the reference, refuses documents nested deeper than 128 levels; so does this reader
    // (a recursive-descent parser must not let a file of brackets overflow the stack)
    struct Depth {
        int& d;
        explicit Depth(int& dd) : d(dd) { d++; }
        ~Depth() { d--; }
    };
    JsonPtr value() {
        Depth guard(depth_);
        if (depth_ > 128) fail("recursion limit exceeded");
        skip_ws();
        auto v = std::make_shared<JsonValue>();
        char c = peek();
        if (c == '{') {
            v->type = JsonValue::Object;
            pos_++;
            skip_ws();
            if (peek() == '}') { pos_++; return v; }
            for (;;) {
                skip_ws();
                std::string k = string_lit();
                skip_ws();
                expect(':');
                v->obj[k] = value();
                skip_ws();
                if (peek() == ',') { pos_++; continue; }
                expect('}');
                break;
            }
        } else if (c == '[') {
            v->type = JsonValue::Array;
            pos_++;
            skip_ws();
            if (peek() == ']') { pos_++; return v; }
            for (;;) {
                v->arr.push_back(value());
                skip_ws();
                if (peek() == ',') { pos_++; continue; }
                expect(']');
                break;
            }
        } else if (c == '"') {
            v->type = JsonValue::String;
            v->str = string_lit();
        } else if (c == 't' && s_.compare(pos_, 4, "true") == 0) {
            v->type = JsonValue::Bool; v->b = true; pos_ += 4;
        } else if (c == 'f' && s_.compare(pos_, 5, "false") == 0) {
            v->type = JsonValue::Bool; v->b = false; pos_ += 5;
        } else if (c == 'n' && s_.compare(pos_, 4, "null") == 0) {
            v->type = JsonValue::Null; pos_ += 4;
        } else if (c == '-' || (c >= '0' && c <= '9')) {
            const char* start = s_.c_str() + pos_;
            char* end = nullptr;
            v->type = JsonValue::Number;
            v->num = std::strtod(start, &end);
            if (end == start) fail("bad number");
            pos_ += (size_t)(end - start);
        } else {
            fail("unexpected character");
        }
        return v;
    }
    std::string string_lit() {
        expect('"');
        std::string out;
        while (pos_ < s_.size()) {
            char c = s_[pos_++];
            if (c == '"') return out;
            if (c == '\\') {
                if (pos_ >= s_.size()) break;
                char e = s_[pos_++];
                switch (e) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {
                        if (pos_ + 4 > s_.size()) fail("bad \\u escape");
                        unsigned cp = (unsigned)std::strtoul(s_.substr(pos_, 4).c_str(), nullptr, 16);
                        pos_ += 4;
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                        else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: out += e; break;  // \" \\ \/
                }
            } else {
                out += c;
            }
        }
        fail("unterminated string");
    }
    const std::string& s_;
    size_t pos_ = 0;
    int depth_ = 0;
};

}  // namespace akr
