// Host-side f64 vector / matrix helpers for scene authoring.
// Semantics follow the reference's vector.rs / matrix.rs (operation order kept so that world-space
// vertices are the same doubles the Rust host would produce): normalize = v * (1/len) (vector.rs:39-46),
// Matrix44 row-major, M*v affine (matrix.rs:180-189), A*B standard product (matrix.rs:162-178).
#pragma once
#include <cmath>
#include "hanamaru_hip.h"

namespace hh {

struct V3 {
    double x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(double a, double b, double c) : x(a), y(b), z(c) {}
    explicit V3(const hr_vec3 &v) : x(v.x), y(v.y), z(v.z) {}
    hr_vec3 c() const { return hr_vec3{x, y, z}; }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 normalize(V3 a) {
    double inv = 1.0 / std::sqrt(dot(a, a));
    return {a.x * inv, a.y * inv, a.z * inv};
}

struct M44 {
    double e[4][4];
    static M44 identity() {
        M44 m{};
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m.e[i][j] = (i == j) ? 1.0 : 0.0;
        return m;
    }
    static M44 scale(double sx, double sy, double sz) {
        M44 m = identity(); m.e[0][0] = sx; m.e[1][1] = sy; m.e[2][2] = sz; return m;
    }
    static M44 scale_linear(double s) { return scale(s, s, s); }
    static M44 rotate_x(double t) {  // matrix.rs:35-44
        double s = std::sin(t), c = std::cos(t);
        M44 m = identity(); m.e[1][1] = c; m.e[1][2] = -s; m.e[2][1] = s; m.e[2][2] = c; return m;
    }
    static M44 rotate_y(double t) {  // matrix.rs:47-56
        double s = std::sin(t), c = std::cos(t);
        M44 m = identity(); m.e[0][0] = c; m.e[0][2] = s; m.e[2][0] = -s; m.e[2][2] = c; return m;
    }
    static M44 translate(double tx, double ty, double tz) {
        M44 m = identity(); m.e[0][3] = tx; m.e[1][3] = ty; m.e[2][3] = tz; return m;
    }
};
inline M44 operator*(const M44 &a, const M44 &b) {
    M44 r = M44::identity();
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.e[i][j] = a.e[i][0] * b.e[0][j] + a.e[i][1] * b.e[1][j] + a.e[i][2] * b.e[2][j] + a.e[i][3] * b.e[3][j];
    return r;
}
inline V3 operator*(const M44 &m, V3 v) {
    return {v.x * m.e[0][0] + v.y * m.e[0][1] + v.z * m.e[0][2] + m.e[0][3],
            v.x * m.e[1][0] + v.y * m.e[1][1] + v.z * m.e[1][2] + m.e[1][3],
            v.x * m.e[2][0] + v.y * m.e[2][1] + v.z * m.e[2][2] + m.e[2][3]};
}

constexpr double PI = 3.14159265358979323846;
constexpr double PI2 = 2.0 * PI;
inline double to_radians(double deg) { return deg * (PI / 180.0); }  // f64::to_radians

void set_error(const char *fmt, ...);

}  // namespace hh
