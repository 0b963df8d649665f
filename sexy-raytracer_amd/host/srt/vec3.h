// vec3.h -- the vector vocabulary of the reference's vec3.h (vec2f/vec3f/vec4f/color3f with
// operator() access), without Eigen.  Only what scene-building code needs.
#ifndef SRT_HOST_VEC3_H
#define SRT_HOST_VEC3_H

#include <cstdint>

template <int N>
struct srtVec {
  float e[N];
  srtVec() { for (int i = 0; i < N; ++i) e[i] = 0; }
  srtVec(float a, float b) { static_assert(N == 2, ""); e[0] = a; e[1] = b; }
  srtVec(float a, float b, float c) { static_assert(N == 3, ""); e[0] = a; e[1] = b; e[2] = c; }
  srtVec(float a, float b, float c, float d) { static_assert(N == 4, ""); e[0] = a; e[1] = b; e[2] = c; e[3] = d; }
  float operator()(int i) const { return e[i]; }
  float& operator()(int i) { return e[i]; }
  float operator[](int i) const { return e[i]; }
  float& operator[](int i) { return e[i]; }
  static srtVec UnitX() { srtVec v; v.e[0] = 1; return v; }
  static srtVec UnitY() { srtVec v; v.e[1] = 1; return v; }
  static srtVec UnitZ() { srtVec v; v.e[2] = 1; return v; }
  srtVec operator+(const srtVec& o) const { srtVec r; for (int i = 0; i < N; ++i) r.e[i] = e[i] + o.e[i]; return r; }
  srtVec operator-(const srtVec& o) const { srtVec r; for (int i = 0; i < N; ++i) r.e[i] = e[i] - o.e[i]; return r; }
  srtVec operator-() const { srtVec r; for (int i = 0; i < N; ++i) r.e[i] = -e[i]; return r; }
  srtVec operator*(float s) const { srtVec r; for (int i = 0; i < N; ++i) r.e[i] = e[i] * s; return r; }
  srtVec operator/(float s) const { srtVec r; for (int i = 0; i < N; ++i) r.e[i] = e[i] / s; return r; }
  // Eigen's 3-vector reduction order, x*x' + (y*y' + z*z') (the oracle's and the kernels' dot3)
  float dot(const srtVec& o) const {
    static_assert(N == 3, "");
    return e[0] * o.e[0] + (e[1] * o.e[1] + e[2] * o.e[2]);
  }
  bool operator!=(const srtVec& o) const { for (int i = 0; i < N; ++i) if (e[i] != o.e[i]) return true; return false; }
};
template <int N>
inline srtVec<N> operator*(float s, const srtVec<N>& v) { return v * s; }

using vec2f = srtVec<2>;
using vec3f = srtVec<3>;
using vec4f = srtVec<4>;
using color3f = srtVec<3>;
using color4f = srtVec<4>;

#endif
