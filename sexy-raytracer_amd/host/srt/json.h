// json.h -- a small JSON reader for the glTF scene-build path (replaces cgltf's JSON side).
#ifndef SRT_HOST_JSON_H
#define SRT_HOST_JSON_H

#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

struct srtJson {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  bool b = false;
  double num = 0;
  std::string str;
  std::vector<srtJson> arr;
  std::map<std::string, srtJson> obj;

  bool has(const std::string& k) const { return kind == Object && obj.count(k); }
  const srtJson& operator[](const std::string& k) const {
    static const srtJson null;
    auto it = obj.find(k);
    return (kind == Object && it != obj.end()) ? it->second : null;
  }
  const srtJson& operator[](size_t i) const {
    static const srtJson null;
    return (kind == Array && i < arr.size()) ? arr[i] : null;
  }
  size_t size() const { return kind == Array ? arr.size() : 0; }
  double number(double dflt) const { return kind == Number ? num : dflt; }
  bool isNull() const { return kind == Null; }
};

class srtJsonParser {
 public:
  explicit srtJsonParser(const std::string& text) : s(text) {}
  bool parse(srtJson& out) {
    ws();
    if (!value(out)) return false;
    ws();
    return p == s.size();
  }

 private:
  const std::string& s;
  size_t p = 0;
  void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\n' || s[p] == '\r' || s[p] == '\t')) ++p; }
  bool lit(const char* w) {
    size_t n = strlen(w);
    if (s.compare(p, n, w) != 0) return false;
    p += n;
    return true;
  }
  bool string(std::string& out) {
    if (p >= s.size() || s[p] != '"') return false;
    ++p;
    while (p < s.size() && s[p] != '"') {
      if (s[p] == '\\' && p + 1 < s.size()) {
        char c = s[p + 1];
        p += 2;
        switch (c) {
          case 'n': out += '\n'; break;
          case 't': out += '\t'; break;
          case 'r': out += '\r'; break;
          case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'u': {  // basic multilingual plane -> UTF-8
            unsigned cp = (unsigned)strtoul(s.substr(p, 4).c_str(), nullptr, 16);
            p += 4;
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: out += c;
        }
      } else
        out += s[p++];
    }
    if (p >= s.size()) return false;
    ++p;
    return true;
  }
  bool value(srtJson& v) {
    if (p >= s.size()) return false;
    char c = s[p];
    if (c == '{') {
      v.kind = srtJson::Object;
      ++p; ws();
      if (p < s.size() && s[p] == '}') { ++p; return true; }
      while (true) {
        std::string k;
        ws();
        if (!string(k)) return false;
        ws();
        if (p >= s.size() || s[p] != ':') return false;
        ++p; ws();
        if (!value(v.obj[k])) return false;
        ws();
        if (p < s.size() && s[p] == ',') { ++p; continue; }
        if (p < s.size() && s[p] == '}') { ++p; return true; }
        return false;
      }
    }
    if (c == '[') {
      v.kind = srtJson::Array;
      ++p; ws();
      if (p < s.size() && s[p] == ']') { ++p; return true; }
      while (true) {
        v.arr.emplace_back();
        ws();
        if (!value(v.arr.back())) return false;
        ws();
        if (p < s.size() && s[p] == ',') { ++p; continue; }
        if (p < s.size() && s[p] == ']') { ++p; return true; }
        return false;
      }
    }
    if (c == '"') { v.kind = srtJson::String; return string(v.str); }
    if (lit("true")) { v.kind = srtJson::Bool; v.b = true; return true; }
    if (lit("false")) { v.kind = srtJson::Bool; v.b = false; return true; }
    if (lit("null")) { v.kind = srtJson::Null; return true; }
    char* end = nullptr;
    v.num = strtod(s.c_str() + p, &end);
    if (end == s.c_str() + p) return false;
    p = end - s.c_str();
    v.kind = srtJson::Number;
    return true;
  }
};

#endif
