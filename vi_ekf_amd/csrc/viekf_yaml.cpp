// viekf_yaml.cpp -- minimal YAML-subset reader for the filter parameter file.
//
// Replaces the get_yaml_node / get_yaml_eigen / get_yaml_diag calls of VIEKF::load
// (reference src/vi_ekf/vi_ekf.cpp:114-131; the helpers live in the absent multirotor_sim
// submodule and wrap yaml-cpp).  Supports what params/ekf.yaml uses: top-level `key: scalar`
// and `key: [flow, sequence]` (possibly spanning lines), `#` comments, true/false booleans.
#include "viekf_host.hpp"

#include <cctype>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace viekf {

static std::string strip_comment(const std::string& s) {
  bool in_s = false, in_d = false;
  for (size_t i = 0; i < s.size(); i++) {
    const char c = s[i];
    if (c == '\'' && !in_d) in_s = !in_s;
    else if (c == '"' && !in_s) in_d = !in_d;
    else if (c == '#' && !in_s && !in_d && (i == 0 || std::isspace((unsigned char)s[i - 1]))) return s.substr(0, i);
  }
  return s;
}

static std::string trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && std::isspace((unsigned char)s[a])) a++;
  while (b > a && std::isspace((unsigned char)s[b - 1])) b--;
  return s.substr(a, b - a);
}

bool yaml_parse_file(const std::string& path, YamlMap& out, std::string& err) {
  std::ifstream f(path);
  if (!f) {
    err = "cannot open parameter file '" + path + "'";
    return false;
  }
  std::string line, key, acc;
  int depth = 0;  // open '[' count of the value being accumulated
  int lineno = 0;
  while (std::getline(f, line)) {
    lineno++;
    std::string s = strip_comment(line);
    if (trim(s).empty()) continue;
    if (depth == 0) {
      if (std::isspace((unsigned char)s[0]) || s[0] == '-') continue;  // nested / block items: not used by load()
      const size_t c = s.find(':');
      if (c == std::string::npos) {
        err = path + ":" + std::to_string(lineno) + ": expected 'key: value'";
        return false;
      }
      key = trim(s.substr(0, c));
      acc = trim(s.substr(c + 1));
    } else {
      acc += " " + trim(s);
    }
    depth = 0;
    for (char ch : acc) {
      if (ch == '[') depth++;
      else if (ch == ']') depth--;
    }
    if (depth < 0) {
      err = path + ":" + std::to_string(lineno) + ": unbalanced ']'";
      return false;
    }
    if (depth == 0) out[key] = acc;
  }
  if (depth != 0) {
    err = path + ": unterminated '[' in value of '" + key + "'";
    return false;
  }
  return true;
}

static bool parse_scalar(const std::string& tok, double& v) {
  std::string t = trim(tok);
  if (t.size() >= 2 && ((t.front() == '"' && t.back() == '"') || (t.front() == '\'' && t.back() == '\'')))
    t = t.substr(1, t.size() - 2);
  if (t == "true" || t == "True" || t == "TRUE") { v = 1.0; return true; }
  if (t == "false" || t == "False" || t == "FALSE") { v = 0.0; return true; }
  if (t == ".nan" || t == ".NaN" || t == "nan") { v = std::strtod("nan", nullptr); return true; }
  if (t.empty()) return false;
  char* end = nullptr;
  v = std::strtod(t.c_str(), &end);
  return end && *end == '\0';
}

bool yaml_get_doubles(const YamlMap& m, const std::string& key, double* out, int count, std::string& err) {
  auto it = m.find(key);
  if (it == m.end()) {
    err = "missing key '" + key + "'";
    return false;
  }
  std::string v = trim(it->second);
  std::vector<std::string> toks;
  if (!v.empty() && v.front() == '[') {
    if (v.back() != ']') {
      err = "malformed sequence for '" + key + "'";
      return false;
    }
    std::stringstream ss(v.substr(1, v.size() - 2));
    std::string t;
    while (std::getline(ss, t, ',')) {
      if (!trim(t).empty()) toks.push_back(t);
    }
  } else {
    toks.push_back(v);
  }
  if ((int)toks.size() != count) {
    err = "key '" + key + "' has " + std::to_string(toks.size()) + " values, expected " + std::to_string(count);
    return false;
  }
  for (int i = 0; i < count; i++) {
    if (!parse_scalar(toks[i], out[i])) {
      err = "key '" + key + "': cannot parse '" + trim(toks[i]) + "' as a number";
      return false;
    }
  }
  return true;
}

bool yaml_get_string(const YamlMap& m, const std::string& key, std::string& out, std::string& err) {
  auto it = m.find(key);
  if (it == m.end()) {
    err = "missing key '" + key + "'";
    return false;
  }
  std::string t = trim(it->second);
  if (t.size() >= 2 && ((t.front() == '"' && t.back() == '"') || (t.front() == '\'' && t.back() == '\'')))
    t = t.substr(1, t.size() - 2);
  out = t;
  return true;
}

}  // namespace viekf
