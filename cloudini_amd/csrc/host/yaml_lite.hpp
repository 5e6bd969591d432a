// yaml_lite: the small subset of YAML that EncodingInfoToYAML emits -- top-level "key: value" lines and one
// block sequence of mappings ("fields:" followed by "  - key: value" / "    key: value" lines). Not a general
// YAML parser (neither is the reference's, include/cloudini_lib/yaml_parser.hpp).
#pragma once

#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

namespace yaml_lite {

struct Mapping {
  std::vector<std::pair<std::string, std::string>> entries;

  bool has(std::string_view key) const {
    for (const auto& e : entries)
      if (e.first == key) return true;
    return false;
  }
  const std::string& scalar(std::string_view key) const {
    for (const auto& e : entries)
      if (e.first == key) return e.second;
    throw std::runtime_error("YAML: missing key '" + std::string(key) + "'");
  }
  long long integer(std::string_view key) const {
    const std::string& s = scalar(key);
    try {
      size_t used = 0;
      const long long v = std::stoll(s, &used);
      if (used != s.size()) throw std::invalid_argument(s);
      return v;
    } catch (const std::exception&) {
      throw std::runtime_error("YAML: '" + std::string(key) + "' is not an integer: " + s);
    }
  }
};

struct Document : Mapping {
  std::vector<Mapping> items;  // the entries of the (single) block sequence
};

inline std::string_view strip(std::string_view s) {
  while (!s.empty() && (s.front() == ' ' || s.front() == '\t')) s.remove_prefix(1);
  while (!s.empty() && (s.back() == ' ' || s.back() == '\t' || s.back() == '\r')) s.remove_suffix(1);
  return s;
}

inline Document parse(std::string_view text) {
  Document doc;
  bool in_sequence = false;
  while (!text.empty()) {
    const size_t eol = text.find('\n');
    std::string_view line = text.substr(0, eol);
    text.remove_prefix(eol == std::string_view::npos ? text.size() : eol + 1);
    if (strip(line).empty() || strip(line).front() == '#') continue;
    const bool indented = line.front() == ' ';
    std::string_view body = strip(line);
    bool new_item = false;
    if (body.size() >= 2 && body[0] == '-' && body[1] == ' ') {
      new_item = true;
      body = strip(body.substr(2));
    }
    const size_t colon = body.find(':');
    if (colon == std::string_view::npos) throw std::runtime_error("YAML: expected 'key: value', got: " + std::string(body));
    const std::string key(strip(body.substr(0, colon)));
    const std::string value(strip(body.substr(colon + 1)));
    if (!indented && !new_item) {
      in_sequence = value.empty();  // "fields:" opens the sequence, any other top-level key closes it
      if (!in_sequence) doc.entries.emplace_back(key, value);
      continue;
    }
    if (!in_sequence) throw std::runtime_error("YAML: unexpected indentation at key '" + key + "'");
    if (new_item) doc.items.emplace_back();
    if (doc.items.empty()) throw std::runtime_error("YAML: sequence entry without '-'");
    doc.items.back().entries.emplace_back(key, value);
  }
  return doc;
}

}  // namespace yaml_lite
