// viekf_host.hpp -- host-side declarations shared by the C-ABI implementation files.
#pragma once
#include <map>
#include <string>
#include <vector>

namespace viekf {

typedef std::map<std::string, std::string> YamlMap;
bool yaml_parse_file(const std::string& path, YamlMap& out, std::string& err);
bool yaml_get_doubles(const YamlMap& m, const std::string& key, double* out, int count, std::string& err);
bool yaml_get_string(const YamlMap& m, const std::string& key, std::string& out, std::string& err);

}  // namespace viekf
