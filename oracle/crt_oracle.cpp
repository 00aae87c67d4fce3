/*
 * oracle/crt_oracle.cpp -- TEST INFRASTRUCTURE, not product code.
 *
 * Single-threaded CPU restatement of the reference path, written from the
 * reference's behaviour (no code copied).  Every function cites the reference
 * file:line it follows.  See crt_oracle.h for the parity status and
 * philox.h / det_math.h for the two third-party substitutions (cuRAND ->
 * Philox4x32-10 with explicit draw addressing; libdevice -> Cephes
 * polynomials).
 *
 * Build: oracle/Makefile (g++ -O2 -ffp-contract=off).
 */
#include "crt_oracle.h"
#include "det_math.h"
#include "philox.h"

#include <algorithm>
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace {

/* ------------------------------------------------------------------ vec3 --
 * Arithmetic conventions of the reference's vendored Eigen 3.4.90, pinned bit
 * for bit by tests/golden/eigen_ops.json:
 *   dot / squaredNorm reduce as p0 + (p1 + p2)   (Eigen/src/Core/Redux.h:101-115)
 *   normalized(): z = squaredNorm; z > 0 ? v / sqrt(z) : v  (Eigen/src/Core/Dot.h:121-131)
 *   cross: (a1b2-a2b1, a2b0-a0b2, a0b1-a1b0)  (Eigen/src/Geometry/OrthoMethods.h:106-108)
 */
struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
inline V3 cwise(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline float dot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
inline float sqnorm(V3 a) { return dot(a, a); }
inline float norm(V3 a) { return sqrtf(sqnorm(a)); }
inline V3 normalized(V3 a)
{
    float z = sqnorm(a);
    if (z > 0.0f) return a / sqrtf(z);
    return a;
}
inline V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
/* column-major 3x3 times vector: row i = m(i,0)v0 + (m(i,1)v1 + m(i,2)v2) */
inline V3 mat3_mul(const float m[9], V3 v)
{
    return v3(m[0] * v.x + (m[3] * v.y + m[6] * v.z), m[1] * v.x + (m[4] * v.y + m[7] * v.z),
              m[2] * v.x + (m[5] * v.y + m[8] * v.z));
}

const float EPSILON = 0.00001f; /* reference: include/Global.h:11 */
const int BVH_STACK_SIZE = 256;   /* Global.h:17 */
const int BOUNCE_STACK_SIZE = 64; /* Global.h:18 */
enum { DIFFUSE = 0, SPECULAR = 1 }; /* include/Material.h:7-10 */

/* reference: include/Material.h:11-40 / include/DeviceMaterial.cuh:5-37 */
struct Material {
    V3 kd{0.1f, 0.1f, 0.1f}, ks{0.1f, 0.1f, 0.1f}, ka{0.1f, 0.1f, 0.1f}, ke{0.0f, 0.0f, 0.0f};
    float ns = 1.0f;
    bool has_emit = false;
    int mode = DIFFUSE;
    Material() {}
    Material(V3 kd_, V3 ks_, V3 ka_, V3 ke_, float ns_, int mode_) : kd(kd_), ks(ks_), ka(ka_), ke(ke_), ns(ns_), mode(mode_)
    {
        /* Material.h:36-39 */
        has_emit = !(ke.x < EPSILON && ke.y < EPSILON && ke.z < EPSILON);
    }
};

/* reference: include/Triangle.h:9-132 and include/DeviceTriangle.cuh:22-37 */
struct Triangle {
    V3 v1, v2, v3_, center, normal, e1, e2;
    Material m;
    float area, area_of_obj;
    float max_x, min_x, max_y, min_y, max_z, min_z;
    Triangle(V3 a, V3 b, V3 c, const Material& mat) : v1(a), v2(b), v3_(c), m(mat)
    {
        center = ((v1 + v2) + v3_) / 3.0f;                    /* Triangle.h:26 */
        normal = normalized(cross(v2 - v1, v3_ - v1));        /* Triangle.h:27 (OBJ normal ignored, :28) */
        max_x = std::max(std::max(v1.x, v2.x), v3_.x);        /* Triangle.h:30-37 */
        min_x = std::min(std::min(v1.x, v2.x), v3_.x);
        max_y = std::max(std::max(v1.y, v2.y), v3_.y);
        min_y = std::min(std::min(v1.y, v2.y), v3_.y);
        max_z = std::max(std::max(v1.z, v2.z), v3_.z);
        min_z = std::min(std::min(v1.z, v2.z), v3_.z);
        area = norm(cross(v2 - v1, v3_ - v1)) * 0.5f;         /* Triangle.h:39 */
        area_of_obj = 0.0f;
        e1 = v2 - v1;                                         /* DeviceTriangle.cuh:27-28 */
        e2 = v3_ - v1;
    }
};
bool cmp_x(const Triangle& a, const Triangle& b) { return a.center.x < b.center.x; } /* Triangle.h:43-56 */
bool cmp_y(const Triangle& a, const Triangle& b) { return a.center.y < b.center.y; }
bool cmp_z(const Triangle& a, const Triangle& b) { return a.center.z < b.center.z; }

/* reference: include/Object.h:7-31 */
struct Object {
    std::vector<Triangle> triangles;
    float area = 0.0f;
    bool is_light = false;
    Object(const std::vector<Triangle>& ts, bool light) : triangles(ts), is_light(light)
    {
        float a = 0.0f;
        for (size_t i = 0; i < triangles.size(); i++) a += triangles[i].area; /* Object.h:16-19 */
        for (size_t i = 0; i < triangles.size(); i++) triangles[i].area_of_obj = a; /* :20-23 */
        area = a;
    }
};

/* reference: include/BVH.h:9-20 */
struct BVHNode {
    int lc = -1, rc = -1;
    unsigned n = 0;
    int it = -1;
    V3 AA{FLT_MAX, FLT_MAX, FLT_MAX}, BB{-FLT_MAX, -FLT_MAX, -FLT_MAX};
};

/* reference: include/OBJLoader.h:12-39 */
struct Shape {
    std::string material_id;
    std::vector<std::vector<uint64_t>> vs;
    float kd[3] = {0, 0, 0}, ke[3] = {0, 0, 0}, ks[3] = {0, 0, 0};
    bool has_map_kd = false;
    float ns = 1.0f; /* uninitialised in the reference when the MTL has no Ns (OBJLoader.h:23); the oracle fixes it at 1 */
    std::string map_kd;
};

} // namespace

struct orc_scene {
    std::vector<Triangle> triangles; /* Scene::triangles, reordered in place by the BVH build (BVH.h:27) */
    std::vector<Object> objects;     /* creation order, normal and light interleaved */
    std::vector<int> light_objs;     /* indices into objects (Scene::light_objs) */
    std::vector<BVHNode> nodes;
    int root = -1;
    unsigned thresh_n = 0;
    std::string error;
};

namespace {

/* reference: include/OBJLoader.h:61-203 */
bool parse_obj(const char* obj_path, const char* mtl_dir, std::vector<V3>& vertices, std::vector<V3>& normals, std::vector<float>& textures,
               std::vector<Shape>& shapes, std::string& err)
{
    std::ifstream obj(obj_path);
    if (!obj.is_open()) { err = std::string("cannot open OBJ ") + obj_path; return false; }
    std::map<std::string, std::vector<uint64_t>> mts;
    std::string mtl_path, line;
    while (std::getline(obj, line)) {
        std::istringstream ls(line);
        std::string prefix;
        ls >> prefix;
        if (prefix == "v") {
            V3 p{0, 0, 0};
            ls >> p.x >> p.y >> p.z;
            vertices.push_back(p);
        } else if (prefix == "vn") {
            V3 p{0, 0, 0};
            ls >> p.x >> p.y >> p.z;
            normals.push_back(p);
        } else if (prefix == "vt") {                                         /* OBJLoader.h:88-93 */
            float u = 0.0f, v = 0.0f;
            ls >> u >> v;
            textures.push_back(u); textures.push_back(v);
        } else if (prefix == "f") {
            std::vector<uint64_t> vi;
            std::string tok;
            while (ls >> tok) {
                std::string first = tok.substr(0, tok.find('/'));
                uint64_t idx = std::stoull(first);                           /* OBJLoader.h:105 */
                vi.push_back(idx > 0 ? idx - 1 : vertices.size() + idx);     /* :106 */
            }
            if (!shapes.empty()) shapes.back().vs.push_back(vi);             /* :120-123 faces before usemtl are dropped */
        } else if (prefix == "mtllib") {
            std::string name;
            ls >> name;
            mtl_path = std::string(mtl_dir) + "/" + name;                    /* :129 */
        } else if (prefix == "usemtl") {
            std::string id;
            ls >> id;
            mts[id].push_back(shapes.size());                                /* :135 */
            Shape s;
            s.material_id = id;
            shapes.push_back(s);
        }
    }
    std::ifstream mtl(mtl_path);
    if (!mtl.is_open()) { err = "cannot open MTL " + mtl_path; return false; }
    std::vector<uint64_t> sids;
    while (std::getline(mtl, line)) {                                        /* :154-201 */
        std::istringstream ls(line);
        std::string prefix;
        ls >> prefix;
        if (prefix == "newmtl") {
            std::string id;
            ls >> id;
            sids = mts[id];
        } else if (prefix == "Kd" || prefix == "Ks" || prefix == "Ke") {
            float v[3] = {0, 0, 0};
            ls >> v[0] >> v[1] >> v[2];
            for (uint64_t s : sids) {
                float* dst = prefix == "Kd" ? shapes[s].kd : (prefix == "Ks" ? shapes[s].ks : shapes[s].ke);
                dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
            }
        } else if (prefix == "map_Kd") {
            std::string name;
            ls >> name;
            for (uint64_t s : sids) { shapes[s].has_map_kd = true; shapes[s].map_kd = std::string(mtl_dir) + "/" + name; }
        } else if (prefix == "Ns") {
            float ns = 1.0f;
            ls >> ns;
            for (uint64_t s : sids) shapes[s].ns = ns;
        }
    }
    return true;
}

/* What stbi_load(path, &x, &y, &comp, 0) returns for a map_Kd file (Loader.h:58).  stb_image is a vendored third-party
 * decoder of the reference and is not restated: the test harness decodes the file with an independent decoder (PIL) and
 * registers the samples here before the scene is loaded. */
struct Texture { int x = 0, y = 0, comp = 0; std::vector<uint8_t> px; };
std::map<std::string, Texture>& texture_registry()
{
    static std::map<std::string, Texture> r;
    return r;
}

/* reference: include/Loader.h:86-89 -- texel under one vertex; `width` / `height` carry the reference's swapped meaning */
bool texel(const Texture& t, int width, int height, int channel, float tu, float tv, V3& out)
{
    float intpart;
    int u = static_cast<int>(std::modf(std::modf(tu, &intpart) + 1, &intpart) * (width - 1));
    int v = static_cast<int>(std::modf(std::modf(tv, &intpart) + 1, &intpart) * (height - 1));
    long long offset = ((long long)v * width + u) * channel;
    if (offset < 0 || (size_t)offset + 2 >= t.px.size()) return false; /* the reference would read out of bounds */
    out = v3((float)t.px[offset] / 255.0f, (float)t.px[offset + 1] / 255.0f, (float)t.px[offset + 2] / 255.0f); /* Vector3f(...) / 255. */
    return true;
}

/* reference: include/Loader.h:40-124 */
bool load_object(const Shape& s, const std::vector<V3>& vertices, const std::vector<V3>& normals, const std::vector<float>& textures,
                 std::vector<Triangle>& tris, std::vector<Triangle>& light_tris, std::string& err)
{
    tris.clear();
    light_tris.clear();
    const Texture* tex = nullptr;
    int width = 0, height = 0, channel = 0;
    if (s.has_map_kd) {
        auto it = texture_registry().find(s.map_kd);
        if (it == texture_registry().end()) { err = "map_Kd texture was not registered with the oracle (orc_register_texture): " + s.map_kd; return false; }
        tex = &it->second;
        /* Loader.h:58: stbi_load(map_kd.c_str(), &height, &width, &channel, 0) -- x lands in `height`, y in `width` */
        height = tex->x; width = tex->y; channel = tex->comp;
    }
    for (size_t i = 0; i < s.vs.size(); i++) {
        if (s.vs[i].size() < 3) { err = "face with fewer than 3 vertices"; return false; }
        uint64_t i1 = s.vs[i][0], i2 = s.vs[i][1], i3 = s.vs[i][2]; /* Loader.h:62-64: first three only */
        if (i1 >= vertices.size() || i2 >= vertices.size() || i3 >= vertices.size()) { err = "vertex index out of range"; return false; }
        /* Loader.h:70-72 reads normals[] with the VERTEX index; the values are unused (Triangle.h:27-28) but an
         * OBJ with fewer vn than v makes the reference read out of bounds -- reject it. */
        if (i1 >= normals.size() || i2 >= normals.size() || i3 >= normals.size()) { err = "OBJ needs one vn per v (Loader.h:70-72)"; return false; }
        V3 kd = v3(s.kd[0], s.kd[1], s.kd[2]);
        V3 ke = v3(s.ke[0], s.ke[1], s.ke[2]);
        if (tex) { /* Loader.h:79-105; textures[] is read with the VERTEX index (:81-83) */
            if (2 * i1 + 1 >= textures.size() || 2 * i2 + 1 >= textures.size() || 2 * i3 + 1 >= textures.size()) { err = "textured OBJ needs one vt per v (Loader.h:81-83)"; return false; }
            V3 k1, k2, k3;
            if (!texel(*tex, width, height, channel, textures[2 * i1], textures[2 * i1 + 1], k1) ||
                !texel(*tex, width, height, channel, textures[2 * i2], textures[2 * i2 + 1], k2) ||
                !texel(*tex, width, height, channel, textures[2 * i3], textures[2 * i3 + 1], k3)) { err = "texture lookup outside the image (Loader.h:58 swaps width and height)"; return false; }
            kd = ((k1 + k2) + k3) / 3.0f; /* Loader.h:103, the `auto` expressions of :89,:95,:101 evaluated eagerly */
        }
        V3 zero = v3(0.0f, 0.0f, 0.0f); /* ks, ka are always zero: Loader.h:45,47,107 */
        Material m(kd, zero, zero, ke, s.ns, s.ns > 1 ? SPECULAR : DIFFUSE); /* Loader.h:107 */
        Triangle t(vertices[i1], vertices[i2], vertices[i3], m);
        if (m.has_emit) light_tris.push_back(t); else tris.push_back(t);     /* Loader.h:119-122 */
    }
    return true;
}

/* reference: include/BVH.h:37-84 */
int build_node(orc_scene& sc, int l, int r)
{
    if (l >= r) return -1;
    BVHNode node;
    std::vector<Triangle>& T = sc.triangles;
    for (int i = l; i < r; i++) {
        node.AA.x = std::min(T[i].min_x, node.AA.x);
        node.AA.y = std::min(T[i].min_y, node.AA.y);
        node.AA.z = std::min(T[i].min_z, node.AA.z);
        node.BB.x = std::max(T[i].max_x, node.BB.x);
        node.BB.y = std::max(T[i].max_y, node.BB.y);
        node.BB.z = std::max(T[i].max_z, node.BB.z);
    }
    node.it = l;
    node.n = (unsigned)(r - l);
    if (node.n <= sc.thresh_n) {
        sc.nodes.push_back(node);
        return (int)sc.nodes.size() - 1;
    }
    V3 d = node.BB - node.AA;
    if (d.x >= d.y && d.x >= d.z) std::sort(T.begin() + l, T.begin() + r, cmp_x);
    else if (d.y >= d.x && d.y >= d.z) std::sort(T.begin() + l, T.begin() + r, cmp_y);
    else if (d.z >= d.x && d.z >= d.y) std::sort(T.begin() + l, T.begin() + r, cmp_z);
    int mid = (l + r) / 2;
    node.lc = build_node(sc, l, mid);
    node.rc = build_node(sc, mid, r);
    sc.nodes.push_back(node); /* post-order: root is last */
    return (int)sc.nodes.size() - 1;
}

/* ------------------------------------------------------------- payloads -- */
/* reference: include/Payload.cuh:8-36 */
struct HitPayload {
    V3 from_pos{0, 0, 0}, from_dir{0, 0, 0};
    bool happend = false;
    V3 pos{0, 0, 0}, normal{0, 0, 0};
    float t = FLT_MAX;
    float area = 0.0f;
    Material m;
    int tri = -1; /* oracle-only bookkeeping */
};
/* reference: include/Payload.cuh:48-72 */
struct LightSample {
    V3 pos, normal, emit;
    float inv_pdf;
};
/* reference: include/Ray.cuh:12-15 */
static std::vector<float>* g_ray_log = nullptr;
static float g_ray_log_limit = std::numeric_limits<float>::quiet_NaN(); /* the visibility limit of the ray being logged (NaN: a closest-hit ray) */
#define ORC_RAY_LOG_FLOATS 10
/* diagnostic (orc_margin_hist): how far in front of its own leaf box the hit a query's answer rests on lies, in units of reach x steep
 * (the pruning slack of the HIP path's CRT_TRAVERSAL_FAST is a multiple of that product): bin b counts answers with
 * 10^(b/2 - 10) <= (box entry - t_ref) / (reach * steep) < 10^((b+1)/2 - 10); t_ref = the hit distance of a closest-hit ray, the limit of a
 * blocked visibility ray.  Bin 0 also takes everything below, bin 23 everything above; [24] = answers with a hit, [25] = all rays. */
static uint64_t* g_margin_hist = nullptr;

struct Ray {
    V3 origin, dir, inv_dir;
    Ray(V3 o, V3 d) : origin(o)
    {
        dir = normalized(d);
        inv_dir = v3(1 / dir.x, 1 / dir.y, 1 / dir.z);
    }
};
inline float maxf(float x, float y) { return x > y ? x : y; } /* Global.h:111-114 */
inline float minf(float x, float y) { return x < y ? x : y; } /* Global.h:116-119 */
inline float clampf(float lo, float hi, float v) { return maxf(lo, minf(hi, v)); } /* Global.h:121-124 */

/* reference: include/DeviceStack.cuh:4-61 (overflow drops, underflow yields T()) */
template <typename T, int cap> struct Stack {
    T data[cap];
    int top = -1;
    void clear() { top = -1; }
    void push(const T& v) { if (top < cap - 1) { top++; data[top] = v; } }
    T pop() { if (top >= 0) { T v = data[top]; top--; return v; } return T(); }
    bool is_full() const { return top >= cap - 1; }
    bool is_empty() const { return top == -1; }
    int size() const { return top + 1; }
};

struct Tracer {
    const orc_scene& sc;
    orc_stats st{};
    Stack<int, BVH_STACK_SIZE> bvh_stack;
    explicit Tracer(const orc_scene& s) : sc(s) {}

    /* reference: include/DeviceTriangle.cuh:39-65 */
    HitPayload tri_intersect(int ti, V3 origin, V3 dir) const
    {
        const Triangle& T = sc.triangles[ti];
        V3 s = origin - T.v1;
        V3 s1 = cross(dir, T.e2);
        V3 s2 = cross(s, T.e1);
        float reciprocal = 1 / dot(s1, T.e1);
        float beta = dot(s1, s) * reciprocal;
        float gamma = dot(s2, dir) * reciprocal;
        float t = dot(s2, T.e2) * reciprocal;
        float alpha = 1 - beta - gamma;
        V3 pos = origin + t * dir;
        if (0 < alpha && alpha < 1 && 0 < beta && beta < 1 && 0 < gamma && gamma < 1) {
            HitPayload h;
            h.from_pos = origin; h.from_dir = dir; h.happend = true; h.pos = pos; h.normal = T.normal;
            h.t = t; h.area = T.area_of_obj; h.m = T.m; h.tri = ti;
            return h;
        }
        return HitPayload();
    }
    /* reference: include/DeviceBVH.cuh:87-126 */
    bool hit_aabb(int node_index, V3 origin, V3 dir, V3 inv_dir) const
    {
        if (node_index < 0) return false;
        const BVHNode& node = sc.nodes[node_index];
        V3 OA = node.AA - origin, OB = node.BB - origin;
        V3 t_min = cwise(OA, inv_dir), t_max = cwise(OB, inv_dir);
        if (dir.x < 0) std::swap(t_min.x, t_max.x);
        if (dir.y < 0) std::swap(t_min.y, t_max.y);
        if (dir.z < 0) std::swap(t_min.z, t_max.z);
        float t_enter = maxf(maxf(t_min.x, t_min.y), t_min.z);
        float t_exit = minf(minf(t_max.x, t_max.y), t_max.z);
        return t_enter <= t_exit + EPSILON && t_exit >= 0;
    }
    /* reference: include/DeviceBVH.cuh:128-170 (traversal) and :31-43 (leaf) */
    /* diagnostic: t_enter of hit_aabb for the leaf the closest hit of the last intersect_() was found in */
    int hit_leaf = -1;
    float leaf_entry(V3 origin, V3 dir, V3 inv_dir) const
    {
        if (hit_leaf < 0) return 0.0f;
        const BVHNode& node = sc.nodes[hit_leaf];
        V3 OA = node.AA - origin, OB = node.BB - origin;
        V3 t_min = cwise(OA, inv_dir), t_max = cwise(OB, inv_dir);
        if (dir.x < 0) std::swap(t_min.x, t_max.x);
        if (dir.y < 0) std::swap(t_min.y, t_max.y);
        if (dir.z < 0) std::swap(t_min.z, t_max.z);
        return maxf(maxf(t_min.x, t_min.y), t_min.z);
    }
    HitPayload intersect(V3 origin, V3 dir, V3 inv_dir)
    {
        HitPayload h = intersect_(origin, dir, inv_dir);
        if (g_margin_hist) {
            g_margin_hist[25]++;
            const bool vis = g_ray_log_limit == g_ray_log_limit;
            if (h.happend && (!vis || g_ray_log_limit - h.t > EPSILON)) {
                g_margin_hist[24]++;
                const double t_ref = vis ? g_ray_log_limit : h.t;
                const double reach = std::max(std::max(std::fabs((double)origin.x), std::fabs((double)origin.y)), std::fabs((double)origin.z)) + std::fabs(t_ref);
                const double steep = std::max(std::max(std::fabs((double)inv_dir.x), std::fabs((double)inv_dir.y)), std::fabs((double)inv_dir.z));
                const double m = ((double)leaf_entry(origin, dir, inv_dir) - t_ref) / (reach * steep);
                if (m > 0) {
                    int b = (int)std::floor((std::log10(m) + 10.0) * 2.0);
                    g_margin_hist[b < 0 ? 0 : (b > 23 ? 23 : b)]++;
                }
            }
        }
        if (g_ray_log) { /* diagnostic ray log (orc_ray_log_*): origin, direction, hit distance, hit triangle */
            const float rec[ORC_RAY_LOG_FLOATS] = {origin.x, origin.y, origin.z, dir.x, dir.y, dir.z, h.t, (float)(h.happend ? h.tri : -1), g_ray_log_limit,
                                                   h.happend ? leaf_entry(origin, dir, inv_dir) : 0.0f};
            g_ray_log->insert(g_ray_log->end(), rec, rec + ORC_RAY_LOG_FLOATS);
        }
        return h;
    }
    HitPayload intersect_(V3 origin, V3 dir, V3 inv_dir)
    {
        st.rays++;
        hit_leaf = -1;
        bvh_stack.clear();
        bvh_stack.push(sc.root);
        HitPayload closest;
        while (!bvh_stack.is_empty()) {
            if ((uint32_t)bvh_stack.size() > st.max_bvh_stack) st.max_bvh_stack = bvh_stack.size();
            int cur = bvh_stack.pop();
            if (cur < 0) continue;
            const BVHNode& node = sc.nodes[cur];
            if (node.lc < 0 && node.rc < 0) {
                st.leaf_pops++;
                HitPayload payload;
                for (int i = node.it; i < (int)(node.it + node.n); i++) {
                    st.tri_tests++;
                    HitPayload tmp = tri_intersect(i, origin, dir);
                    if (tmp.t > EPSILON && tmp.t < payload.t) payload = tmp;
                }
                if (payload.t < closest.t) { closest = payload; hit_leaf = cur; }
            } else {
                st.inner_pops++;
                bool hl = hit_aabb(node.lc, origin, dir, inv_dir);
                bool hr = hit_aabb(node.rc, origin, dir, inv_dir);
                if (hl && hr) { bvh_stack.push(node.lc); bvh_stack.push(node.rc); }
                else if (hl) bvh_stack.push(node.lc);
                else if (hr) bvh_stack.push(node.rc);
            }
        }
        if (closest.happend) st.hits++;
        return closest;
    }
};

/* reference: include/Global.h:35-50 */
V3 to_world(V3 a, V3 N)
{
    V3 B, C;
    if (fabsf(N.x) > fabsf(N.y)) {
        float invLen = 1.0f / sqrtf(N.x * N.x + N.z * N.z);
        C = v3(N.z * invLen, 0.0f, -N.x * invLen);
    } else {
        float invLen = 1.0f / sqrtf(N.y * N.y + N.z * N.z);
        C = v3(0.0f, N.z * invLen, -N.y * invLen);
    }
    B = cross(C, N);
    return (a.x * B + a.y * C) + a.z * N;
}
/* reference: include/Global.h:57-66 */
V3 sample_hemisphere(V3 N, float x_1, float x_2)
{
    float z = fabsf(1.0f - 2.0f * x_1);
    float r = sqrtf(1.0f - z * z);
    float phi = (float)(2 * M_PI * x_2); /* double product narrowed: Global.h:63 */
    V3 local = v3(r * om_cosf(phi), r * om_sinf(phi), z);
    return to_world(local, N);
}
/* reference: include/Global.h:68-94 */
V3 sample_lobe(V3 out, float delta_theta, float delta_phi, float u1, float u2)
{
    float eta_1 = 2 * u1 - 1;
    float eta_2 = 2 * u2 - 1;
    float r = norm(out);
    float theta_0 = om_acosf(out.z / r);
    float phi_0;
    if ((double)fabsf(out.x) < 1e-5) /* compared in double: Global.h:77 */
        phi_0 = out.y > 0.0f ? (float)M_PI_2 : -(float)M_PI_2;
    else
        phi_0 = om_atan2f(out.y, out.x); /* both remaining branches are identical: Global.h:81-89 */
    float theta = theta_0 + eta_1 * delta_theta;
    float phi = phi_0 + eta_2 * delta_phi;
    return v3(om_sinf(theta) * om_cosf(phi), om_sinf(theta) * om_sinf(phi), om_cosf(theta));
}

struct PathCtx {
    Tracer& tr;
    const orc_scene& sc;
    uint64_t seed;
    uint32_t pixel, k;
    int light_sample_n;
    float P_RR;
};

/* reference: include/DeviceLights.cuh:33-37 + include/DeviceTriangle.cuh:67-74 */
LightSample sample_light(const PathCtx& c, int light, uint32_t depth, uint32_t idx)
{
    const Object& obj = c.sc.objects[c.sc.light_objs[light]];
    uint32_t r[4];
    orc_draw(c.seed, c.pixel, c.k, depth, ORC_RNG_NEE, idx, r);
    uint32_t ti = (uint32_t)((uint64_t)r[0] % (uint64_t)obj.triangles.size());
    const Triangle& T = obj.triangles[ti];
    float alpha = orc_uniform(r[1]);
    float beta = orc_uniform(r[2]) * (1 - alpha);
    float gamma = 1 - alpha - beta;
    LightSample s;
    s.pos = (alpha * T.v1 + beta * T.v2) + gamma * T.v3_;
    s.normal = T.normal;
    s.emit = T.m.ke;
    s.inv_pdf = T.area_of_obj;
    return s;
}

/* reference: include/Render.cuh:199-328 (cast_ray_v2) */
V3 cast_ray_v2(PathCtx& c, Ray ray)
{
    Tracer& tr = c.tr;
    Stack<HitPayload, BOUNCE_STACK_SIZE> bounce;
    bool done = false;
    Ray tmp = ray;
    /* forward loop: Render.cuh:205-230 */
    while (!done) {
        HitPayload hit = tr.intersect(tmp.origin, tmp.dir, tmp.inv_dir);
        bounce.push(hit);
        uint32_t depth = (uint32_t)bounce.size() - 1;
        if (hit.happend) tr.st.vertices++;
        if (depth > tr.st.max_depth) tr.st.max_depth = depth;
        if (!hit.happend || hit.m.has_emit || bounce.is_full()) {
            done = true;
        } else {
            uint32_t r[4];
            orc_draw(c.seed, c.pixel, c.k, depth, ORC_RNG_BOUNCE, 0, r);
            float RR = orc_uniform(r[0]);
            if (RR > c.P_RR) {
                done = true;
            } else {
                V3 reflect_dir = normalized(sample_hemisphere(hit.normal, orc_uniform(r[1]), orc_uniform(r[2])));
                tmp = Ray(hit.pos, reflect_dir);
            }
        }
    }
    /* backward loop: Render.cuh:232-327 */
    HitPayload to_hit, pre_hit;
    bool is_final_hit = true;
    V3 L = v3(0.0f, 0.0f, 0.0f);
    while (!bounce.is_empty()) {
        to_hit = bounce.pop();
        uint32_t depth = (uint32_t)bounce.size(); /* index of the popped vertex */
        if (!to_hit.happend) continue;            /* :242-245 */
        V3 L_dir = v3(0.0f, 0.0f, 0.0f), L_indir = v3(0.0f, 0.0f, 0.0f);
        if (to_hit.m.has_emit) {                  /* :249-255 */
            if (bounce.is_empty()) L_dir = to_hit.m.ke;
            else is_final_hit = false;
        } else {
            V3 pos = to_hit.pos;
            V3 f_r = to_hit.m.kd / (float)M_PI;   /* :259 */
            V3 normal = to_hit.normal;
            int n_lights = (int)c.sc.light_objs.size();
            for (int i = 0; i < n_lights; i++) {  /* :262-286 */
                for (int j = 0; j < c.light_sample_n; j++) {
                    LightSample ls = sample_light(c, i, depth, (uint32_t)(i * c.light_sample_n + j));
                    V3 dist = ls.pos - pos;
                    V3 dir = normalized(dist);
                    Ray back(pos, dir);
                    /* blocked(): Render.cuh:19-27 with t_to_light = dist.x()/dir.x() (:272) */
                    tr.st.shadow_rays++;
                    g_ray_log_limit = dist.x / dir.x;
                    HitPayload sh = tr.intersect(back.origin, back.dir, back.inv_dir);
                    g_ray_log_limit = std::numeric_limits<float>::quiet_NaN();
                    bool is_blocked = (dist.x / dir.x) - sh.t > EPSILON;
                    if (!is_blocked) {
                        V3 L_i = ls.emit;
                        float t_to_light = norm(dist);
                        float t2 = t_to_light * t_to_light;
                        float inv_pdf = ls.inv_pdf;
                        float cos_theta = dot(dir, normal);
                        float cos_theta_2 = -dot(dir, ls.normal);
                        cos_theta = cos_theta > 0.0f ? cos_theta : 0.0f;
                        cos_theta_2 = cos_theta_2 > 0.0f ? cos_theta_2 : 0.0f;
                        /* :283, left-to-right scalar chain */
                        L_dir = L_dir + cwise(L_i, f_r) * cos_theta * cos_theta_2 * inv_pdf / t2 / (float)c.light_sample_n;
                    }
                }
            }
            if (!is_final_hit) {                  /* :288-315 */
                float inv_pdf = (float)(2.0f * M_PI); /* Global.h:96-99 */
                float cos_theta = dot(normalized(pre_hit.pos - pos), normal);
                cos_theta = cos_theta > 0.0f ? cos_theta : 0.0f;
                L_indir = cwise(L, f_r) * cos_theta * inv_pdf / c.P_RR; /* :293 */
                if (to_hit.m.mode == SPECULAR) {
                    float ns = to_hit.m.ns;
                    float delta_coeff = (float)((om_expf(25 / ns) - 1) / (M_E - 1)); /* :297 */
                    V3 in = normalized(to_hit.from_dir);
                    V3 out = in - 2.f * dot(in, to_hit.normal) * to_hit.normal;    /* :299 */
                    /* DELTA_THETA / DELTA_PHI are unparenthesised macros (Global.h:20-21):
                     * delta_coeff * 30 is a float product, then * M_PI / 180 in double, narrowed at the call */
                    float d_theta = (float)(delta_coeff * 30 * M_PI / 180);
                    float d_phi = (float)(delta_coeff * 120 * M_PI / 180);
                    uint32_t r[4];
                    orc_draw(c.seed, c.pixel, c.k, depth, ORC_RNG_PROBE, 0, r);
                    V3 ref = normalized(sample_lobe(out, d_theta, d_phi, orc_uniform(r[0]), orc_uniform(r[1])));
                    Ray probe(to_hit.pos, ref);
                    tr.st.probe_rays++;
                    HitPayload hit = tr.intersect(probe.origin, probe.dir, probe.inv_dir);
                    if (hit.m.has_emit) {         /* :304 */
                        float log_shininess = om_log10f(to_hit.m.ns);
                        float shininess_coeff = (float)(log_shininess * 0.5 + 1); /* :307 double */
                        float ip = (float)(2.0f * M_PI) / 8.f;                     /* :308 */
                        float ct = dot(normalized(hit.pos - to_hit.pos), normal);
                        ct = ct > 0.0f ? ct : 0.0f;
                        /* :311 with the dangling-expression fix (eager evaluation) */
                        V3 temp = shininess_coeff * cwise(hit.m.ke, to_hit.m.kd) * ct * ip;
                        L_dir = L_dir + temp;
                    }
                }
            } else {
                is_final_hit = false;             /* :316-319 */
            }
        }
        pre_hit = to_hit;
        L = L_indir + L_dir;                      /* :323 */
    }
    return L;
}

inline uint8_t to_u8(float v)
{
    if (!(v == v)) return 0; /* NaN -> 0 (GPU float->int conversion semantics; UB on CPU) */
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v; /* truncation, Render.cuh:350 */
}
inline uint8_t tonemap1(float c) { return to_u8(255 * om_powf(clampf(0, 1, c), 0.6f)); }

} // namespace

extern "C" {

/* Registers the decoded samples of a map_Kd file (what stbi_load would return: x, y, components, 8-bit samples). */
void orc_register_texture(const char* path, int x, int y, int comp, const uint8_t* px)
{
    Texture t;
    t.x = x; t.y = y; t.comp = comp;
    t.px.assign(px, px + (size_t)x * y * comp);
    texture_registry()[path] = t;
}

orc_scene* orc_scene_new(void) { return new orc_scene(); }
void orc_scene_free(orc_scene* s) { delete s; }

/* reference: src/main.cu:122-145 */
int orc_scene_add_obj(orc_scene* sc, const char* obj_path, const char* mtl_dir)
{
    std::vector<V3> vertices, normals;
    std::vector<float> textures;
    std::vector<Shape> shapes;
    if (!parse_obj(obj_path, mtl_dir, vertices, normals, textures, shapes, sc->error)) { fprintf(stderr, "oracle: %s\n", sc->error.c_str()); return -1; }
    std::vector<Triangle> tris, light_tris;
    for (size_t i = 0; i < shapes.size(); i++) {
        if (!load_object(shapes[i], vertices, normals, textures, tris, light_tris, sc->error)) { fprintf(stderr, "oracle: %s\n", sc->error.c_str()); return -2; }
        if (!tris.empty()) { /* Scene::add_normal_obj, Scene.h:44-48 */
            Object o(tris, false);
            sc->triangles.insert(sc->triangles.end(), o.triangles.begin(), o.triangles.end());
            sc->objects.push_back(o);
        }
        if (!light_tris.empty()) { /* Scene::add_light_obj, Scene.h:38-42 */
            Object o(light_tris, true);
            sc->triangles.insert(sc->triangles.end(), o.triangles.begin(), o.triangles.end());
            sc->light_objs.push_back((int)sc->objects.size());
            sc->objects.push_back(o);
        }
    }
    return 0;
}
int orc_scene_build(orc_scene* sc, uint32_t thresh_n)
{
    sc->thresh_n = thresh_n;
    sc->nodes.clear();
    sc->root = build_node(*sc, 0, (int)sc->triangles.size()); /* BVH.h:30-34 */
    return sc->root >= 0 ? 0 : -1;
}
uint32_t orc_scene_num_tris(const orc_scene* s) { return (uint32_t)s->triangles.size(); }
uint32_t orc_scene_num_nodes(const orc_scene* s) { return (uint32_t)s->nodes.size(); }
int32_t orc_scene_root(const orc_scene* s) { return s->root; }
uint32_t orc_scene_num_lights(const orc_scene* s) { return (uint32_t)s->light_objs.size(); }
uint32_t orc_scene_light_size(const orc_scene* s, uint32_t l) { return (uint32_t)s->objects[s->light_objs[l]].triangles.size(); }
uint32_t orc_scene_num_objects(const orc_scene* s) { return (uint32_t)s->objects.size(); }
float orc_scene_object_area(const orc_scene* s, uint32_t o) { return s->objects[o].area; }
int orc_scene_object_is_light(const orc_scene* s, uint32_t o) { return s->objects[o].is_light ? 1 : 0; }
void orc_scene_get_nodes(const orc_scene* s, orc_node* out)
{
    for (size_t i = 0; i < s->nodes.size(); i++) {
        const BVHNode& n = s->nodes[i];
        out[i].lc = n.lc; out[i].rc = n.rc; out[i].n = n.n; out[i].it = n.it;
        out[i].aa[0] = n.AA.x; out[i].aa[1] = n.AA.y; out[i].aa[2] = n.AA.z;
        out[i].bb[0] = n.BB.x; out[i].bb[1] = n.BB.y; out[i].bb[2] = n.BB.z;
    }
}
static void fill_tri(const Triangle& T, orc_tri* o)
{
    const V3* src[6] = {&T.v1, &T.v2, &T.v3_, &T.e1, &T.e2, &T.normal};
    float* dst[6] = {o->v1, o->v2, o->v3, o->e1, o->e2, o->normal};
    for (int k = 0; k < 6; k++) { dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z; }
    o->kd[0] = T.m.kd.x; o->kd[1] = T.m.kd.y; o->kd[2] = T.m.kd.z;
    o->ke[0] = T.m.ke.x; o->ke[1] = T.m.ke.y; o->ke[2] = T.m.ke.z;
    o->ns = T.m.ns; o->has_emit = T.m.has_emit ? 1 : 0; o->mode = T.m.mode;
    o->area = T.area; o->area_of_obj = T.area_of_obj;
}
void orc_scene_get_tris(const orc_scene* s, orc_tri* out)
{
    for (size_t i = 0; i < s->triangles.size(); i++) fill_tri(s->triangles[i], &out[i]);
}
void orc_scene_get_light_tris(const orc_scene* s, uint32_t l, orc_tri* out)
{
    const Object& o = s->objects[s->light_objs[l]];
    for (size_t i = 0; i < o.triangles.size(); i++) fill_tri(o.triangles[i], &out[i]);
}

/* reference: include/Camera.h:9-36 */
void orc_inverse_view(const float eye[3], const float lookat[3], const float up[3], float out[9])
{
    V3 e = v3(eye[0], eye[1], eye[2]), l = v3(lookat[0], lookat[1], lookat[2]), u0 = v3(up[0], up[1], up[2]);
    V3 f = normalized(l - e);
    V3 r = normalized(cross(u0, f));
    V3 u = normalized(cross(f, r));
    out[0] = r.x; out[1] = r.y; out[2] = r.z; /* column 0 */
    out[3] = u.x; out[4] = u.y; out[5] = u.z; /* column 1 */
    out[6] = f.x; out[7] = f.y; out[8] = f.z; /* column 2 */
}

/* diagnostic: every ray orc_render traces between orc_ray_log_begin() and orc_ray_log_end(); ORC_RAY_LOG_FLOATS (10) floats per ray:
 * origin, direction, hit distance, hit triangle, visibility limit (NaN for a closest-hit ray), box-entry distance of the hit leaf */
extern "C" void orc_ray_log_begin(void) { delete g_ray_log; g_ray_log = new std::vector<float>(); }
extern "C" uint64_t orc_ray_log_end(float* out, uint64_t cap_rays)
{
    if (!g_ray_log) return 0;
    const uint64_t n = g_ray_log->size() / ORC_RAY_LOG_FLOATS;
    if (out) std::memcpy(out, g_ray_log->data(), (size_t)std::min<uint64_t>(n, cap_rays) * ORC_RAY_LOG_FLOATS * sizeof(float));
    if (out) { delete g_ray_log; g_ray_log = nullptr; }
    return n;
}

/* diagnostic: see g_margin_hist.  orc_margin_hist(NULL) starts (and zeroes) the histogram, orc_margin_hist(out26) copies and stops it */
extern "C" void orc_margin_hist(uint64_t* out)
{
    if (!out) { delete[] g_margin_hist; g_margin_hist = new uint64_t[26](); return; }
    if (g_margin_hist) { std::memcpy(out, g_margin_hist, 26 * sizeof(uint64_t)); delete[] g_margin_hist; g_margin_hist = nullptr; }
}

/* reference: include/Render.cuh:330-354 (view_render_kernel) driven as a pixel loop */
int orc_render(const orc_scene* sc, const orc_camera* cam, const orc_params* p, uint8_t* out_rgb, float* out_mean,
               float* out_L, orc_stats* stats)
{
    if (!sc || sc->root < 0 || !cam || !p) return -1;
    if (p->x0 + p->cw > p->width || p->y0 + p->ch > p->height || p->spp == 0) return -2;
    Tracer tr(*sc);
    V3 eye = v3(cam->eye[0], cam->eye[1], cam->eye[2]);
    float scale = om_tanf(cam->fov_y / 2);                 /* :338 */
    float ar = (float)p->width / p->height;                /* :339 */
    for (uint32_t jj = 0; jj < p->ch; jj++) {
        for (uint32_t ii = 0; ii < p->cw; ii++) {
            int i = (int)(p->x0 + ii), j = (int)(p->y0 + jj);
            uint32_t pixel_index = (uint32_t)j * p->width + (uint32_t)i; /* :336 */
            V3 color = v3(0.0f, 0.0f, 0.0f);
            for (uint32_t k = 0; k < p->spp; k++) {
                uint32_t r[4];
                orc_draw(p->seed, pixel_index, k, 0, ORC_RNG_JITTER, 0, r);
                float x = (2 * (i + orc_uniform(r[0])) / p->width - 1) * scale * ar; /* :344 */
                float y = (1 - 2 * (j + orc_uniform(r[1])) / p->height) * scale;     /* :345 */
                V3 dir = mat3_mul(cam->inv_view, normalized(v3(-x, y, 1)));          /* :346 */
                Ray ray(eye, dir);
                PathCtx c{tr, *sc, p->seed, pixel_index, k, p->light_sample_n, p->p_rr};
                tr.st.paths++;
                V3 L = cast_ray_v2(c, ray);
                if (out_L) {
                    float* o = out_L + ((size_t)(jj * p->cw + ii) * p->spp + k) * 3;
                    o[0] = L.x; o[1] = L.y; o[2] = L.z;
                }
                color = color + L / (float)p->spp;                                   /* :348 */
            }
            size_t o = (size_t)(jj * p->cw + ii) * 3;
            if (out_mean) { out_mean[o] = color.x; out_mean[o + 1] = color.y; out_mean[o + 2] = color.z; }
            if (out_rgb) { out_rgb[o] = tonemap1(color.x); out_rgb[o + 1] = tonemap1(color.y); out_rgb[o + 2] = tonemap1(color.z); } /* :350 */
        }
    }
    if (stats) *stats = tr.st;
    return 0;
}

void orc_intersect(const orc_scene* sc, uint32_t n, const float* origins, const float* dirs, int32_t* out_tri,
                   float* out_t, orc_stats* stats)
{
    Tracer tr(*sc);
    for (uint32_t i = 0; i < n; i++) {
        Ray r(v3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]), v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]));
        HitPayload h = tr.intersect(r.origin, r.dir, r.inv_dir);
        out_tri[i] = h.happend ? h.tri : -1;
        out_t[i] = h.t;
    }
    if (stats) *stats = tr.st;
}

void orc_math(const char* fn, uint32_t n, const float* a, const float* b, float* out)
{
    std::string f(fn);
    for (uint32_t i = 0; i < n; i++) {
        float x = a[i], y = b ? b[i] : 0.0f;
        if (f == "sin") out[i] = om_sinf(x);
        else if (f == "cos") out[i] = om_cosf(x);
        else if (f == "tan") out[i] = om_tanf(x);
        else if (f == "acos") out[i] = om_acosf(x);
        else if (f == "atan2") out[i] = om_atan2f(x, y);
        else if (f == "exp") out[i] = om_expf(x);
        else if (f == "log") out[i] = om_logf(x);
        else if (f == "log10") out[i] = om_log10f(x);
        else if (f == "pow") out[i] = om_powf(x, y);
        else out[i] = om_nan();
    }
}
void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { orc_philox4x32_10(ctr, key, out); }
void orc_rng_draw(uint64_t seed, uint32_t pixel, uint32_t k, uint32_t depth, uint32_t purpose, uint32_t idx,
                  uint32_t out_u32[4], float out_uniform[4])
{
    orc_draw(seed, pixel, k, depth, purpose, idx, out_u32);
    for (int i = 0; i < 4; i++) out_uniform[i] = orc_uniform(out_u32[i]);
}
void orc_sample_hemisphere(const float n[3], float x1, float x2, float out[3])
{
    V3 r = sample_hemisphere(v3(n[0], n[1], n[2]), x1, x2);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_sample_lobe(const float o[3], float dt, float dp, float u1, float u2, float res[3])
{
    V3 r = sample_lobe(v3(o[0], o[1], o[2]), dt, dp, u1, u2);
    res[0] = r.x; res[1] = r.y; res[2] = r.z;
}
void orc_tonemap(uint32_t n, const float* c, uint8_t* out)
{
    for (uint32_t i = 0; i < n; i++) out[i] = tonemap1(c[i]);
}

/* in: a(0-2) b(3-5) c(6-8) k(9) k2(10) for vector ops; for mat3 ops: m columns a,b,c (0-8), v (9-11) */
int orc_vec_op(const char* op_, const float* in, float* out)
{
    std::string op(op_);
    V3 a = v3(in[0], in[1], in[2]), b = v3(in[3], in[4], in[5]), c = v3(in[6], in[7], in[8]);
    float k = in[9], k2 = in[10];
    auto put = [&](V3 r) { out[0] = r.x; out[1] = r.y; out[2] = r.z; return 3; };
    if (op == "dot") { out[0] = dot(a, b); return 1; }
    if (op == "squaredNorm") { out[0] = sqnorm(a); return 1; }
    if (op == "norm") { out[0] = norm(a); return 1; }
    if (op == "normalized") return put(normalized(a));
    if (op == "cross") return put(cross(a, b));
    if (op == "cwiseProduct") return put(cwise(a, b));
    if (op == "div_scalar") return put(a / k);
    if (op == "scalar_mul") return put(k * a);
    if (op == "mul_scalar") return put(a * k);
    if (op == "sub") return put(a - b);
    if (op == "lincomb3") return put((k * a + k2 * b) + c.x * c);
    if (op == "reflect") return put(a - 2.f * dot(a, b) * b);
    if (op == "madd") return put(a + k * b);
    if (op == "nee_chain") return put(cwise(a, b) * k * k2 * c.x / c.y / (float)2);
    if (op == "indir_chain") return put(cwise(a, b) * k * k2 / c.x);
    if (op == "probe_chain") return put(k * cwise(a, b) * k2 * c.x);
    if (op == "div_unsigned7") return put(a / (float)7u);
    if (op == "centroid") return put(((a + b) + c) / 3.0f);
    if (op == "texel_kd") { /* Loader.h:89-103: the nine inputs are 8-bit samples */
        V3 k1 = v3(a.x / 255.0f, a.y / 255.0f, a.z / 255.0f), k2 = v3(b.x / 255.0f, b.y / 255.0f, b.z / 255.0f), k3 = v3(c.x / 255.0f, c.y / 255.0f, c.z / 255.0f);
        return put(((k1 + k2) + k3) / 3.0f);
    }
    if (op == "tri_normal") return put(normalized(cross(b - a, c - a)));
    if (op == "tri_area") { out[0] = norm(cross(b - a, c - a)) * 0.5f; return 1; }
    if (op == "cos_between") { out[0] = dot(normalized(a - b), c); return 1; }
    if (op == "mat3_mul" || op == "mat3_mul_normalized") {
        /* probe builds m << a.x,b.x,c.x, a.y,b.y,c.y, a.z,b.z,c.z : columns are a, b, c */
        float m[9] = {a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z};
        V3 v = v3(in[9], in[10], in[11]);
        return put(mat3_mul(m, op == "mat3_mul" ? v : normalized(v)));
    }
    if (op == "inverse_view") {
        orc_inverse_view(in, in + 3, in + 6, out);
        return 9;
    }
    return -1;
}

} // extern "C"
