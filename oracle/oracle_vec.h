/*
 * Vector primitives of the CPU oracle.  TEST INFRASTRUCTURE ONLY.
 *
 * Each helper reproduces, operation for operation, what the fixed-size Eigen 3.3.7 expression of
 * the same name evaluates to in binary32 (no FMA).  eigen_pin.cpp checks them bit for bit against
 * the Eigen vendored by the reference.  Noteworthy: Eigen's unrolled reduction of a 3-vector
 * associates as x + (y + z)  (Eigen/src/Core/Redux.h, redux_novec_unroller splits 3 into 1 + 2),
 * and normalized()/normalize() divide by sqrt(squaredNorm) only when squaredNorm > 0
 * (Eigen/src/Core/Dot.h:124-134).
 */
#ifndef ORACLE_VEC_H
#define ORACLE_VEC_H

#include <math.h>

typedef struct {
    float x, y, z;
} v3;

static inline v3 v3_make(float x, float y, float z) {
    v3 r = {x, y, z};
    return r;
}
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(float s, v3 a) { return v3_make(s * a.x, s * a.y, s * a.z); }
static inline v3 v3_divs(v3 a, float s) { return v3_make(a.x / s, a.y / s, a.z / s); }
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
static inline float v3_sqnorm(v3 a) { return a.x * a.x + (a.y * a.y + a.z * a.z); }
static inline float v3_norm(v3 a) { return sqrtf(v3_sqnorm(a)); }
static inline v3 v3_normalized(v3 a) {
    const float z = v3_sqnorm(a);
    if (z > 0.0f) return v3_divs(a, sqrtf(z));
    return a;
}
static inline v3 v3_cross(v3 a, v3 b) {
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* std::max / std::min as Eigen's cwiseMax/cwiseMin use them: max(a,b) = a<b ? b : a. */
static inline float f_max(float a, float b) { return a < b ? b : a; }
static inline float f_min(float a, float b) { return b < a ? b : a; }
static inline v3 v3_clamp01(v3 a) {
    return v3_make(f_min(f_max(a.x, 0.0f), 1.0f), f_min(f_max(a.y, 0.0f), 1.0f), f_min(f_max(a.z, 0.0f), 1.0f));
}

/* Affine3f * Vector3f: translation + linear * v (Eigen/src/Geometry/Transform.h, Dim-sized rhs).
 * m = row-major 3x4. */
static inline v3 affine_point(const float *m, v3 p) {
    return v3_make(m[3] + (m[0] * p.x + m[1] * p.y + m[2] * p.z), m[7] + (m[4] * p.x + m[5] * p.y + m[6] * p.z),
                   m[11] + (m[8] * p.x + m[9] * p.y + m[10] * p.z));
}

/* shape.scale(s); shape.translate(-c); model(=I) * shape  (model.hpp:102-105,169-173). */
static inline void model_matrix(float s, v3 c, float *m) {
    for (int i = 0; i < 12; ++i) m[i] = 0.0f;
    m[0] = s;
    m[5] = s;
    m[10] = s;
    m[3] = s * -c.x;
    m[7] = s * -c.y;
    m[11] = s * -c.z;
}
/* (Affine3f * Vector4f(v,1)).head(3) for a model matrix built by model_matrix(). */
static inline v3 model_point(const float *m, v3 p) {
    return v3_make(m[0] * p.x + m[3], m[5] * p.y + m[7], m[10] * p.z + m[11]);
}

#endif
