// scenes.h — camera construction and the built-in scene catalogue of the hip way.
//
// Mirrors what the reference's driver does before calling a renderer:
//  * Camera(eye, lookAt, up, w, h, vfov) + setFocus           (src/math/Camera.h:40-51)
//  * createScene(sb, name, params)                             (src/main/main.cpp:291-309)
// The result is the POD `ptw_camera` (everything Camera keeps private) and a filled
// SceneBuilder.  Scene constants are restated from main.cpp:69-289 as data tables.
#pragma once

#include "scene_builder.h"

#include <stdexcept>
#include <string>

namespace ptw {

// `up` is normalised here, as every caller in the reference passes camUp.normalised().
ptw_camera makeCamera(const Vec3d &eye, const Vec3d &lookAt, const Vec3d &up, int width,
                      int height, double verticalFovDegrees);
void setFocus(ptw_camera &camera, const Vec3d &focalPoint, double apertureRadius);

// Throws UnknownScene for a name outside the catalogue, IoError/ParseError from the loader.
struct UnknownScene : std::runtime_error {
  using std::runtime_error::runtime_error;
};
ptw_camera buildNamedScene(SceneBuilder &sb, const std::string &name,
                           const std::string &scenesDir, int width, int height);

// gamma-2.2 decode of a 0xRRGGBB colour, main.cpp:40-43.
Vec3d hexColour(uint32_t hex);
// 12 triangles of an axis-aligned box in the reference's emission order, main.cpp:45-67.
void addCube(SceneBuilder &sb, const Vec3d &low, const Vec3d &high, const ptw_material &mat);

} // namespace ptw
