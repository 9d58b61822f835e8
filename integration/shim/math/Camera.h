// test stand-in (integration/shim/README.md): a Camera with the reference's private member names
// and the one-line friend patch of INTEGRATION.md; built from the values the C ABI computes.
#pragma once
#include "math/Vec3.h"
namespace hip {
class Scene;
}
class OrthoNormalBasis {
  Norm3 x_, y_, z_;

public:
  OrthoNormalBasis() = default;
  OrthoNormalBasis(const Norm3 &x, const Norm3 &y, const Norm3 &z) : x_(x), y_(y), z_(z) {}
  [[nodiscard]] const Norm3 &x() const { return x_; }
  [[nodiscard]] const Norm3 &y() const { return y_; }
  [[nodiscard]] const Norm3 &z() const { return z_; }
};
class Camera {
  friend class hip::Scene;
  Vec3 centre_;
  OrthoNormalBasis axis_;
  double aspectRatio_{};
  double cameraPlaneDist_{};
  double reciprocalHeight_{};
  double reciprocalWidth_{};
  double apertureRadius_{};
  double focalDistance_{};

public:
  Camera(const Vec3 &centre, const OrthoNormalBasis &axis, double aspectRatio, double cameraPlaneDist,
         double reciprocalHeight, double reciprocalWidth, double apertureRadius, double focalDistance)
      : centre_(centre), axis_(axis), aspectRatio_(aspectRatio), cameraPlaneDist_(cameraPlaneDist),
        reciprocalHeight_(reciprocalHeight), reciprocalWidth_(reciprocalWidth), apertureRadius_(apertureRadius),
        focalDistance_(focalDistance) {}
};
