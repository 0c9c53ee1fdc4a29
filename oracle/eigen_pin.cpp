// Pins the oracle's vector primitives against the reference's vendored Eigen 3.3.7.
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile with
//   g++ -O2 -ffp-contract=off -I/root/reference/dependencies/eigen/include
// into oracle/_ref/eigen_pin (never shipped, never linked into the product).
//
// The program evaluates, with real Eigen, the fixed-size expressions the reference's hot path is
// written in (src/flyscene.cpp:186,494-500,507,545-546,758,773-785,822; camera.hpp:173) on seeded
// random inputs, and compares every result BIT FOR BIT with the plain-C helpers of oracle_vec.h.
// Exit code 0 = every comparison identical.
#include <Eigen/Dense>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

extern "C" {
#include "oracle_vec.h"
}

static uint32_t bits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

static int failures = 0;
static void check(const char *what, float a, float b) {
    if (bits(a) != bits(b) && !(a != a && b != b)) {
        if (failures < 20) std::printf("MISMATCH %s: eigen %.9g (%08x) oracle %.9g (%08x)\n", what, a, bits(a), b, bits(b));
        ++failures;
    }
}
static void check3(const char *what, const Eigen::Vector3f &e, v3 o) {
    check(what, e.x(), o.x);
    check(what, e.y(), o.y);
    check(what, e.z(), o.z);
}
static v3 tov(const Eigen::Vector3f &e) {
    v3 r = {e.x(), e.y(), e.z()};
    return r;
}

int main() {
    std::mt19937 rng(20251111u);
    std::uniform_real_distribution<float> U(-2.0f, 2.0f);
    const int N = 200000;
    for (int it = 0; it < N; ++it) {
        Eigen::Vector3f a(U(rng), U(rng), U(rng)), b(U(rng), U(rng), U(rng)), c(U(rng), U(rng), U(rng));
        Eigen::Vector3f n = Eigen::Vector3f(U(rng), U(rng), U(rng)).normalized();
        if (it % 97 == 0) a = Eigen::Vector3f::Zero();
        const float s = U(rng), t = U(rng);

        check("dot", a.dot(b), v3_dot(tov(a), tov(b)));
        check("squaredNorm", a.squaredNorm(), v3_sqnorm(tov(a)));
        check("norm", a.norm(), v3_norm(tov(a)));
        check3("normalized", a.normalized(), v3_normalized(tov(a)));
        {
            Eigen::Vector3f m = b;
            m.normalize();
            check3("normalize", m, v3_normalized(tov(b)));
        }
        check3("cross", a.cross(b), v3_cross(tov(a), tov(b)));
        check3("sub", a - b, v3_sub(tov(a), tov(b)));
        check3("add", a + b, v3_add(tov(a), tov(b)));
        check3("scale", s * a, v3_scale(s, tov(a)));
        check3("cwiseProduct", a.cwiseProduct(b), v3_mul(tov(a), tov(b)));

        // flyscene.cpp:758  tHit = (D - o.n) / (d.n)
        {
            const float D = t;
            const auto tHit = (D - a.dot(n)) / (b.dot(n));
            check("tHit", tHit, (D - v3_dot(tov(a), tov(n))) / v3_dot(tov(b), tov(n)));
            // flyscene.cpp:771-773  lazily evaluated hit point inside the edge function
            const auto hitPoint = a + tHit * b;
            const auto e = (c - a).cross(hitPoint - a).dot(n);
            const v3 hp = v3_add(tov(a), v3_scale(tHit, tov(b)));
            check("edge", e, v3_dot(v3_cross(v3_sub(tov(c), tov(a)), v3_sub(hp, tov(a))), tov(n)));
            Eigen::Vector3f hpe = hitPoint;
            check3("hitpoint", hpe, hp);
        }
        // flyscene.cpp:785  reflection = d - 2 * d.n * n
        {
            Eigen::Vector3f r = b - 2 * b.dot(n) * n;
            check3("reflection", r, v3_sub(tov(b), v3_scale(2 * v3_dot(tov(b), tov(n)), tov(n))));
        }
        // flyscene.cpp:546  reflectedLight = (-L + 2.f * L.N * N).normalized()
        {
            Eigen::Vector3f L = a.normalized();
            Eigen::Vector3f R = (-L + 2.f * L.dot(n) * n).normalized();
            v3 l = v3_normalized(tov(a));
            check3("reflectedLight", R, v3_normalized(v3_add(v3_neg(l), v3_scale(2.f * v3_dot(l, tov(n)), tov(n)))));
        }
        // flyscene.cpp:494-500  area-weighted normal
        {
            const auto A0 = (b - a).cross(c - a).norm() * 0.5;
            const auto A1 = (c - a).cross(n - a).norm() * 0.5;
            Eigen::Vector3f fn = A0 * b + A1 * c + A0 * n;
            const float a0 = v3_norm(v3_cross(v3_sub(tov(b), tov(a)), v3_sub(tov(c), tov(a)))) * 0.5f;
            const float a1 = v3_norm(v3_cross(v3_sub(tov(c), tov(a)), v3_sub(tov(n), tov(a)))) * 0.5f;
            check3("areaNormal", fn, v3_add(v3_add(v3_scale(a0, tov(b)), v3_scale(a1, tov(c))), v3_scale(a0, tov(n))));
        }
        // camera.hpp:173  inverseViewMatrix * norm_coords  (Affine3f * Vector3f)
        {
            Eigen::Affine3f T = Eigen::Affine3f::Identity();
            T.linear() << U(rng), U(rng), U(rng), U(rng), U(rng), U(rng), U(rng), U(rng), U(rng);
            T.translation() = c;
            Eigen::Vector3f w = T * a;
            float m[12];
            for (int r = 0; r < 3; ++r) {
                for (int k = 0; k < 3; ++k) m[r * 4 + k] = T.linear()(r, k);
                m[r * 4 + 3] = T.translation()(r);
            }
            check3("affine*vec3", w, affine_point(m, tov(a)));
        }
        // model.hpp:169-173 + flyscene.cpp:186  (model * shape) * Vector4f(v, 1)
        {
            Eigen::Affine3f shape = Eigen::Affine3f::Identity();
            Eigen::Affine3f model = Eigen::Affine3f::Identity();
            const float sc = 0.01f + (s < 0 ? -s : s);
            shape.scale(sc);
            shape.translate(-c);
            Eigen::Affine3f M = model * shape;
            Eigen::Vector4f v4(a.x(), a.y(), a.z(), 1.0f);
            Eigen::Vector3f p = (M * v4).head(3);
            float m[12];
            model_matrix(sc, tov(c), m);
            for (int r = 0; r < 3; ++r) {
                for (int k = 0; k < 3; ++k) check("modelLinear", M.linear()(r, k), m[r * 4 + k]);
                check("modelTranslation", M.translation()(r), m[r * 4 + 3]);
            }
            check3("model*vec4", p, model_point(m, tov(a)));
        }
        // Vector3f built from doubles (camera.hpp:162-165, mtl.hpp:22-28)
        {
            const double d0 = 2.0 * (double)s / 3.0 - 1.0;
            Eigen::Vector3f v(d0, 0.3, -1.0);
            check("fromDouble", v.x(), (float)d0);
            check("fromDouble", v.y(), (float)0.3);
        }
        // cwiseMax(0.0).cwiseMin(1.0)  (flyscene.cpp:555)
        {
            Eigen::Vector3f v = (a + b).cwiseMax(0.0).cwiseMin(1.0);
            check3("clamp01", v, v3_clamp01(v3_add(tov(a), tov(b))));
        }
    }
    std::printf("eigen_pin: %d iterations, %d mismatches\n", N, failures);
    return failures ? 1 : 0;
}
