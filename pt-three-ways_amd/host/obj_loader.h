// obj_loader.h — Wavefront OBJ/MTL reader for the hip way's SceneBuilder.
//
// Behavioural mirror of the reference loader (src/util/ObjLoaderImpl.h:55-103,
// src/util/ObjLoader.cpp:7-108) so that the bundled scenes produce the same primitive list:
//  * tokens: runs of characters other than space, tab, CR, LF and '#'; '#' starts a comment
//    that runs to end of line (the reference's regex  \s*((#.*)|[^ \t\n\r#]+) );
//  * directives `v f g o s usemtl mtllib`; anything else is
//    "Unknown directive '<x>' on line <n>";
//  * faces are fanned (i0, i, i+1); indices are 1-based or negative-relative; `a/b/c` keeps `a`;
//  * MTL: newmtl Ke Kd Ka Ni Ns illum (Ks, d ignored); Ns -> cone angle
//    pi * clamp(1 - Ns/100, 0, 1); a material whose block ends while illum == 3 gets
//    reflectivity = |Ka| (illum and Ka carry across blocks, as upstream).
// Errors are reported as ptw::ParseError / ptw::IoError (std::runtime_error subclasses).
#pragma once

#include "scene_builder.h"

#include <functional>
#include <istream>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>

namespace ptw {

struct ParseError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct IoError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// Resolves the file named by a `mtllib` directive (the reference's ObjLoaderOpener).
using MtlOpener = std::function<std::unique_ptr<std::istream>(const std::string &)>;

// Opener that resolves names relative to a directory ("Unable to open <dir>/<name>").
MtlOpener dirRelativeOpener(std::string dir);

std::unordered_map<std::string, ptw_material> loadMaterials(std::istream &in);

void loadObj(std::istream &in, const MtlOpener &opener, SceneBuilder &sb);

// Convenience: open `<dir>/<file>` and load it with a dir-relative opener.
void loadObjFile(const std::string &dir, const std::string &file, SceneBuilder &sb);

} // namespace ptw
