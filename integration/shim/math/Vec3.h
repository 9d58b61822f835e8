// test stand-in (integration/shim/README.md): the members of Vec3 / Norm3 that hip::Scene uses
#pragma once
class Vec3 {
  double x_{}, y_{}, z_{};

public:
  constexpr Vec3() = default;
  constexpr Vec3(double x, double y, double z) : x_(x), y_(y), z_(z) {}
  [[nodiscard]] constexpr double x() const { return x_; }
  [[nodiscard]] constexpr double y() const { return y_; }
  [[nodiscard]] constexpr double z() const { return z_; }
};
class Norm3 {
  Vec3 v_;

public:
  constexpr Norm3() = default;
  constexpr Norm3(double x, double y, double z) : v_(x, y, z) {}
  [[nodiscard]] constexpr double x() const { return v_.x(); }
  [[nodiscard]] constexpr double y() const { return v_.y(); }
  [[nodiscard]] constexpr double z() const { return v_.z(); }
};
