// Probe of the reference's VENDORED third-party arithmetic (Eigen 3.4.90 under
// /root/reference/include/Eigen) and of the one reference header that compiles
// as plain C++ without any CUDA stand-in (include/Camera.h).
//
// Built only in the authoring container (needs /root/reference); emits golden
// vectors as JSON on stdout.  See oracle/Makefile target `ref_probe` and
// tests/golden/make_eigen_golden.py.  Nothing here is product code.
//
// Each record: op name, input float bit patterns, output float bit patterns.
#include <Eigen/Dense>
#include <Camera.h>   // reference: include/Camera.h:9-36 get_inverse_view_matrix
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint32_t next_u32() {  // splitmix64
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 32);
}
static float rnd(float scale) {  // uniform in [-scale, scale), varied magnitudes
    float u = (float)(next_u32() >> 8) * (1.0f / 16777216.0f);
    float v = (2.0f * u - 1.0f) * scale;
    if ((next_u32() & 7) == 0) v *= 1e-3f;
    return v;
}
static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static bool first = true;
static void emit(const char* op, const std::vector<float>& in, const std::vector<float>& out) {
    std::printf("%s{\"op\":\"%s\",\"in\":[", first ? "" : ",\n", op);
    first = false;
    for (size_t i = 0; i < in.size(); i++) std::printf("%s%u", i ? "," : "", bits(in[i]));
    std::printf("],\"out\":[");
    for (size_t i = 0; i < out.size(); i++) std::printf("%s%u", i ? "," : "", bits(out[i]));
    std::printf("]}");
}
// noinline + volatile-ish inputs so the compiler cannot fold anything
template <class T> __attribute__((noinline)) T opaque(const T& v) { return v; }
using V = Eigen::Vector3f;
using M = Eigen::Matrix3f;
static std::vector<float> f3(const V& v) { return {v.x(), v.y(), v.z()}; }

int main() {
    std::printf("[\n");
    for (int it = 0; it < 96; it++) {
        float sc = (it % 4 == 0) ? 600.0f : (it % 4 == 1 ? 30.0f : 1.0f);
        V a(rnd(sc), rnd(sc), rnd(sc)), b(rnd(sc), rnd(sc), rnd(sc)), c(rnd(sc), rnd(sc), rnd(sc));
        a = opaque(a); b = opaque(b); c = opaque(c);
        float k = opaque(rnd(3.0f)), k2 = opaque(rnd(2.0f));
        std::vector<float> in = {a.x(), a.y(), a.z(), b.x(), b.y(), b.z(), c.x(), c.y(), c.z(), k, k2};
        emit("dot", in, {a.dot(b)});
        emit("squaredNorm", in, {a.squaredNorm()});
        emit("norm", in, {a.norm()});
        emit("normalized", in, f3(a.normalized()));
        emit("cross", in, f3(a.cross(b)));
        emit("cwiseProduct", in, f3(a.cwiseProduct(b)));
        { V r = a / k; emit("div_scalar", in, f3(r)); }
        { V r = k * a; emit("scalar_mul", in, f3(r)); }
        { V r = a * k; emit("mul_scalar", in, f3(r)); }
        { V r = a - b; emit("sub", in, f3(r)); }
        // reference: include/Global.h:49  a.x()*B + a.y()*C + a.z()*N
        { V r = k * a + k2 * b + c.x() * c; emit("lincomb3", in, f3(r)); }
        // reference: include/Render.cuh:299  in - 2.f * in.dot(n) * n
        { V r = a - 2.f * a.dot(b) * b; emit("reflect", in, f3(r)); }
        // reference: include/DeviceTriangle.cuh:50  origin + t * dir
        { V r = a + k * b; emit("madd", in, f3(r)); }
        // reference: include/Render.cuh:283 chain
        { int lsn = 2; V r = a.cwiseProduct(b) * k * k2 * c.x() / c.y() / lsn; emit("nee_chain", in, f3(r)); }
        // reference: include/Render.cuh:293 chain
        { V r = a.cwiseProduct(b) * k * k2 / c.x(); emit("indir_chain", in, f3(r)); }
        // reference: include/Render.cuh:311 chain (with the eager-evaluation fix)
        { V r = k * a.cwiseProduct(b) * k2 * c.x(); emit("probe_chain", in, f3(r)); }
        // reference: include/Render.cuh:348  L / spp with unsigned spp
        { unsigned spp = 7; V r = a / spp; emit("div_unsigned7", in, f3(r)); }
        // reference: include/Triangle.h:26  (v1+v2+v3)/3
        { V r = (a + b + c) / 3; emit("centroid", in, f3(r)); }
        // reference: include/Loader.h:89-103  kd = mean of three texels, each Vector3f(u8, u8, u8) / 255.  (evaluated eagerly)
        {
            unsigned char tx[9];
            for (int q = 0; q < 9; q++) tx[q] = (unsigned char)(next_u32() & 255u);
            unsigned char* t = opaque(&tx[0]);
            V k1 = V(t[0], t[1], t[2]) / 255.;
            V k2 = V(t[3], t[4], t[5]) / 255.;
            V k3 = V(t[6], t[7], t[8]) / 255.;
            V r = (k1 + k2 + k3) / 3;
            std::vector<float> inb = {(float)t[0], (float)t[1], (float)t[2], (float)t[3], (float)t[4], (float)t[5], (float)t[6], (float)t[7], (float)t[8], 0.0f, 0.0f};
            emit("texel_kd", inb, f3(r));
        }
        // reference: include/Triangle.h:27,39  normal and area
        { V r = (b - a).cross(c - a).normalized(); emit("tri_normal", in, f3(r)); }
        { float r = (b - a).cross(c - a).norm() * 0.5f; emit("tri_area", in, {r}); }
        // reference: include/Render.cuh:291
        { float r = (a - b).normalized().dot(c); emit("cos_between", in, {r}); }
        M m;
        m << a.x(), b.x(), c.x(), a.y(), b.y(), c.y(), a.z(), b.z(), c.z();
        m = opaque(m);
        V v(k, k2, opaque(rnd(1.0f)));
        std::vector<float> inm = {a.x(), a.y(), a.z(), b.x(), b.y(), b.z(), c.x(), c.y(), c.z(), v.x(), v.y(), v.z()};
        // reference: include/Render.cuh:346  inv_view_mat * Vector3f(-x,y,1).normalized()
        { V r = m * v.normalized(); emit("mat3_mul_normalized", inm, f3(r)); }
        { V r = m * v; emit("mat3_mul", inm, f3(r)); }
    }
    // Camera: reference include/Camera.h:9-36
    for (int it = 0; it < 64; it++) {
        V eye(rnd(800.f), rnd(300.f), rnd(800.f)), look(rnd(100.f), rnd(100.f), rnd(100.f)), up(0, 1, 0);
        if (it == 0) { eye = V(278.0f, 273.0f, -800.0f); look = V(278.0f, 273.0f, -799.0f); }
        if (it == 1) { eye = V(28.2792f, 5.2f, 1.23612e-06f); look = V(0.0f, 2.8f, 0.0f); }
        if (it >= 32) up = V(rnd(1.f), 1.0f + rnd(0.5f), rnd(1.f));
        M r = get_inverse_view_matrix(opaque(eye), opaque(look), opaque(up));
        std::vector<float> in = {eye.x(), eye.y(), eye.z(), look.x(), look.y(), look.z(), up.x(), up.y(), up.z()};
        std::vector<float> out(r.data(), r.data() + 9);  // column-major storage
        emit("inverse_view", in, out);
    }
    std::printf("\n]\n");
    return 0;
}
