// test stand-in (integration/shim/README.md): the fields of MaterialSpec, in the reference's order
#pragma once
#include "math/Vec3.h"
struct MaterialSpec {
  Vec3 emission;
  Vec3 diffuse;
  double indexOfRefraction{1.0};
  double reflectivity{-1};
  double reflectionConeAngleRadians{0.0};
};
