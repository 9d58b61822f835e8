#include "obj_loader.h"

#include <algorithm>
#include <cmath>
#include <fstream>
#include <string_view>
#include <vector>

namespace ptw {
namespace {

using Fields = std::vector<std::string_view>;

bool isBlank(char c) {
  // \s of the reference's regex: space, \t, \n, \v, \f, \r
  return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r';
}
bool endsToken(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '#'; }

// Splits one line into directive + parameters, dropping comments.
void tokenize(std::string_view line, Fields &out) {
  out.clear();
  size_t i = 0;
  const size_t n = line.size();
  while (i < n) {
    while (i < n && isBlank(line[i])) ++i;
    if (i >= n || line[i] == '#') break; // a comment swallows the rest of the line
    size_t start = i;
    while (i < n && !endsToken(line[i])) ++i;
    out.emplace_back(line.substr(start, i - start));
  }
}

// Runs `handler(command, params, lineNumber)` for every non-empty line.
template <typename Handler>
void forEachDirective(std::istream &in, Handler &&handler) {
  std::string line;
  Fields fields;
  int lineNumber = 0;
  while (std::getline(in, line)) {
    ++lineNumber;
    tokenize(line, fields);
    if (fields.empty()) continue;
    std::string_view command = fields.front();
    fields.erase(fields.begin());
    if (!handler(command, fields))
      throw ParseError("Unknown directive '" + std::string(command) + "' on line " +
                       std::to_string(lineNumber));
  }
}

double toDouble(std::string_view sv) {
  try {
    return std::stod(std::string(sv)); // same conversion (prefix parse) as upstream
  } catch (const std::exception &) {
    throw ParseError("Bad number '" + std::string(sv) + "'");
  }
}
long toLong(std::string_view sv) {
  try {
    return std::stol(std::string(sv)); // stops at '/', so "7/1/3" reads as 7
  } catch (const std::exception &) {
    throw ParseError("Bad integer '" + std::string(sv) + "'");
  }
}

Vec3d toVec3(const Fields &params, const char *what) {
  if (params.size() != 3)
    throw ParseError(std::string("Wrong number of params for ") + what);
  return {toDouble(params[0]), toDouble(params[1]), toDouble(params[2])};
}

void requireOne(const Fields &params, const char *what) {
  if (params.size() != 1)
    throw ParseError(std::string("Wrong number of params for ") + what);
}

} // namespace

MtlOpener dirRelativeOpener(std::string dir) {
  return [dir = std::move(dir)](const std::string &name) -> std::unique_ptr<std::istream> {
    const std::string full = dir + "/" + name;
    auto stream = std::make_unique<std::ifstream>(full);
    if (!*stream) throw IoError("Unable to open " + full);
    return stream;
  };
}

std::unordered_map<std::string, ptw_material> loadMaterials(std::istream &in) {
  if (!in) throw IoError("Bad input stream");
  std::unordered_map<std::string, ptw_material> table;
  ptw_material *current = nullptr;
  int illum = 2;    // sticky across blocks
  Vec3d ambient{};  // sticky across blocks

  auto closeBlock = [&] {
    if (current && illum == 3) current->reflectivity = length(ambient);
    current = nullptr;
  };
  auto need = [&](const char *what) -> ptw_material & {
    if (!current) throw ParseError(std::string("Unexpected ") + what);
    return *current;
  };

  forEachDirective(in, [&](std::string_view cmd, const Fields &params) {
    if (cmd == "newmtl") {
      closeBlock();
      requireOne(params, "newmtl");
      // emplace: a repeated name keeps (and keeps editing) the first definition
      current = &table.emplace(std::string(params[0]), material::defaults()).first->second;
    } else if (cmd == "Ke") {
      ptw_material &m = need("Ke");
      toVec3(params, "Ke").store(m.emission);
    } else if (cmd == "Kd") {
      ptw_material &m = need("Kd");
      toVec3(params, "Kd").store(m.diffuse);
    } else if (cmd == "Ka") {
      need("Ka");
      ambient = toVec3(params, "Ka");
    } else if (cmd == "Ni") {
      ptw_material &m = need("Ni");
      requireOne(params, "Ni");
      m.index_of_refraction = toDouble(params[0]);
    } else if (cmd == "Ns") {
      ptw_material &m = need("Ns");
      requireOne(params, "Ns");
      const double val = toDouble(params[0]) / 100;
      m.reflection_cone_angle_rad = M_PI * std::clamp(1 - val, 0.0, 1.0);
    } else if (cmd == "illum") {
      need("illum");
      requireOne(params, "illum");
      illum = static_cast<int>(toLong(params[0]));
    } else if (cmd == "Ks" || cmd == "d") {
      // ignored
    } else {
      return false;
    }
    return true;
  });
  closeBlock();
  return table;
}

void loadObj(std::istream &in, const MtlOpener &opener, SceneBuilder &sb) {
  if (!in) throw IoError("Bad input stream");
  std::vector<Vec3d> vertices;
  std::unordered_map<std::string, ptw_material> materials;
  ptw_material currentMaterial = material::defaults();
  std::vector<size_t> corner;

  forEachDirective(in, [&](std::string_view cmd, const Fields &params) {
    if (cmd == "v") {
      vertices.push_back(toVec3(params, "v"));
    } else if (cmd == "f") {
      corner.clear();
      for (auto p : params) {
        const long raw = toLong(p);
        // negative: relative to the vertices seen so far; positive: 1-based
        const size_t idx = raw < 0 ? static_cast<size_t>(raw + static_cast<long>(vertices.size()))
                                   : static_cast<size_t>(raw - 1);
        if (idx >= vertices.size())
          throw ParseError("Vertex index " + std::string(p) + " out of range");
        corner.push_back(idx);
      }
      // fan around the first corner
      for (size_t k = 1; k + 1 < corner.size(); ++k)
        sb.addTriangle(vertices[corner[0]], vertices[corner[k]], vertices[corner[k + 1]],
                       currentMaterial);
    } else if (cmd == "g" || cmd == "o" || cmd == "s") {
      // groups, object names, smoothing: ignored
    } else if (cmd == "usemtl") {
      if (params.empty()) throw ParseError("Wrong number of params for usemtl");
      const std::string name(params[0]);
      auto it = materials.find(name);
      if (it == materials.end()) throw ParseError("Can't find material " + name);
      currentMaterial = it->second;
    } else if (cmd == "mtllib") {
      if (params.empty()) throw ParseError("Wrong number of params for mtllib");
      if (!opener) throw IoError("Unexpected");
      auto file = opener(std::string(params[0]));
      materials = loadMaterials(*file);
    } else {
      return false;
    }
    return true;
  });
}

void loadObjFile(const std::string &dir, const std::string &file, SceneBuilder &sb) {
  MtlOpener opener = dirRelativeOpener(dir);
  auto in = opener(file);
  loadObj(*in, opener, sb);
}

} // namespace ptw
