// cudaraytracing_amd/csrc/crt_host.cpp -- host layer of libcrt.so (see crt_host.hpp).
//
// Float evaluation order follows the reference's vendored Eigen:
// dot = p0 + (p1 + p2), normalized = v / sqrt(squaredNorm) with true division,
// cross as in OrthoMethods.h.  Compiled with -ffp-contract=off.
#include "crt_host.hpp"
#include "crt_bvh_build.h"
#include "crt_image.h"

#include <algorithm>
#include <array>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace crt {

namespace {
inline Vec3 mk(float x, float y, float z) { Vec3 v; v.x = x; v.y = y; v.z = z; return v; }
inline Vec3 sub(Vec3 a, Vec3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 add(Vec3 a, Vec3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 divs(Vec3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
inline float dot3(Vec3 a, Vec3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
inline Vec3 cross3(Vec3 a, Vec3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline Vec3 unit(Vec3 a)
{
    float z = dot3(a, a);
    return z > 0.0f ? divs(a, std::sqrt(z)) : a;
}
const float kEps = 0.00001f; // reference: include/Global.h:11
} // namespace

// ---------------------------------------------------------------- Material --
Material::Material() : kd_(mk(0.1f, 0.1f, 0.1f)), ke_(mk(0.0f, 0.0f, 0.0f)), ns_(1.0f), has_emit_(false), mode_(DIFFUSE) {}

Material::Material(Vec3 kd, Vec3 ke, float ns, Illum mode) : kd_(kd), ke_(ke), ns_(ns), mode_(mode)
{
    // reference: include/Material.h:36-39
    has_emit_ = !(ke.x < kEps && ke.y < kEps && ke.z < kEps);
}

bool Material::same_as(const Material& o) const
{
    return std::memcmp(&kd_, &o.kd_, sizeof(Vec3)) == 0 && std::memcmp(&ke_, &o.ke_, sizeof(Vec3)) == 0 &&
           std::memcmp(&ns_, &o.ns_, sizeof(float)) == 0 && has_emit_ == o.has_emit_ && mode_ == o.mode_;
}

// ---------------------------------------------------------------- Triangle --
Triangle::Triangle(Vec3 v1, Vec3 v2, Vec3 v3, const Material& m) : v1_(v1), v2_(v2), v3_(v3), material_(m)
{
    center_ = divs(add(add(v1, v2), v3), 3.0f);           // Triangle.h:26
    Vec3 n = cross3(sub(v2, v1), sub(v3, v1));
    normal_ = unit(n);                                    // Triangle.h:27 -- winding normal, OBJ vn ignored
    hi_ = mk(std::max(std::max(v1.x, v2.x), v3.x), std::max(std::max(v1.y, v2.y), v3.y), std::max(std::max(v1.z, v2.z), v3.z));
    lo_ = mk(std::min(std::min(v1.x, v2.x), v3.x), std::min(std::min(v1.y, v2.y), v3.y), std::min(std::min(v1.z, v2.z), v3.z));
    area_ = std::sqrt(dot3(n, n)) * 0.5f;                 // Triangle.h:39
}

// ------------------------------------------------------------------ Object --
Object::Object(const std::vector<Triangle>& ts) : triangles_(ts)
{
    float a = 0.0f;
    for (const Triangle& t : triangles_) a += t.get_area(); // Object.h:16-19, sequential float sum
    for (Triangle& t : triangles_) t.set_area_of_obj(a);    // Object.h:20-23
    area_ = a;
}

// --------------------------------------------------------------------- BVH --
BVH::BVH(unsigned thresh_n, std::vector<Triangle>& triangles) : thresh_n_(thresh_n), triangles_(triangles)
{
    if (thresh_n == 0) throw Error(CRT_ERR_INVALID_ARG, "bvh_thresh_n must be >= 1 (0 recurses forever in the reference, BVH.h:57-81)");
    // The reference std::sort's the 152-byte Triangle objects themselves at every level
    // (BVH.h:65-76).  Sorting 16-byte (centroid, index) keys with the same comparator runs
    // the identical comparison sequence, hence yields the identical permutation; the
    // triangles are gathered once at the end.
    std::vector<Key> keys(triangles.size());
    for (size_t i = 0; i < triangles.size(); i++) {
        Vec3 c = triangles[i].get_center();
        keys[i].c[0] = c.x; keys[i].c[1] = c.y; keys[i].c[2] = c.z;
        keys[i].idx = (uint32_t)i;
    }
    root_ = build_node(keys, 0, (int)keys.size(), 1);
    std::vector<Triangle> sorted;
    sorted.reserve(triangles.size());
    for (const Key& k : keys) sorted.push_back(triangles[k.idx]);
    triangles.swap(sorted);
}

BVH::BVH(unsigned thresh_n, std::vector<Triangle>& triangles, int device, crt_bvh_build_info* info) : thresh_n_(thresh_n), triangles_(triangles)
{
    if (thresh_n == 0) throw Error(CRT_ERR_INVALID_ARG, "bvh_thresh_n must be >= 1 (0 recurses forever in the reference, BVH.h:57-81)");
    if (triangles.empty()) throw Error(CRT_ERR_INVALID_ARG, "BVH over an empty triangle list");
    const auto t0 = std::chrono::steady_clock::now();
    const size_t n = triangles.size();
    std::vector<float> tmin(n * 3), tmax(n * 3), cen(n * 3);
    bool plain = true; // no -0.0 / NaN / inf: std::min, std::max and the comparator do not depend on scan order
    for (size_t i = 0; i < n; i++) {
        const Vec3 lo = triangles[i].get_min(), hi = triangles[i].get_max(), c = triangles[i].get_center();
        const float v[9] = {lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, c.x, c.y, c.z};
        for (int k = 0; k < 3; k++) { tmin[i * 3 + k] = v[k]; tmax[i * 3 + k] = v[3 + k]; cen[i * 3 + k] = v[6 + k]; }
        for (int k = 0; k < 9; k++) plain = plain && std::isfinite(v[k]) && !(v[k] == 0.0f && std::signbit(v[k]));
    }
    crt_bvh_build_info bi;
    std::memset(&bi, 0, sizeof(bi));
    std::vector<Key> keys(n);
    auto make_key = [&](size_t pos, uint32_t tri) {
        keys[pos].c[0] = cen[tri * 3]; keys[pos].c[1] = cen[tri * 3 + 1]; keys[pos].c[2] = cen[tri * 3 + 2];
        keys[pos].idx = tri;
    };
    if (!plain) { // the whole tree on the host
        for (size_t i = 0; i < n; i++) make_key(i, (uint32_t)i);
        root_ = build_node(keys, 0, (int)n, 1);
        bi.n_triangles = (uint32_t)n; bi.n_nodes = (uint32_t)nodes_.size(); bi.host_ranges = 1; bi.host_triangles = (uint32_t)n;
        bi.levels = max_depth_;
    } else {
        std::vector<uint32_t> perm(n);
        std::vector<crt_bvh_node> flat(2 * n + 1);
        std::vector<crt_bvh_host_range> ranges;
        int rc = crt_bvh_build_device((uint32_t)n, tmin.data(), tmax.data(), cen.data(), thresh_n, device, perm.data(), flat.data(), (uint32_t)flat.size(), &ranges, &bi);
        if (rc != CRT_OK) throw Error(rc, std::string("BVH device build failed: ") + crt_last_error());
        for (size_t pos = 0; pos < n; pos++) make_key(pos, perm[pos]);
        nodes_.resize(bi.n_nodes);
        for (uint32_t i = 0; i < bi.n_nodes; i++) {
            BVHNode& o = nodes_[i];
            const crt_bvh_node& f = flat[i];
            o.lc = f.lc; o.rc = f.rc; o.n = f.n; o.it = f.it;
            o.AA = mk(f.aa[0], f.aa[1], f.aa[2]); o.BB = mk(f.bb[0], f.bb[1], f.bb[2]);
        }
        max_depth_ = bi.levels;
        const auto h0 = std::chrono::steady_clock::now();
        for (const crt_bvh_host_range& hr : ranges) { // subtrees with equal sort keys: the reference's own recursion from here on
            std::vector<BVHNode> whole;
            whole.swap(nodes_);
            build_node(keys, (int)hr.l, (int)hr.r, hr.depth);
            whole.swap(nodes_); // `whole` now holds the subtree in its own post-order numbering
            for (size_t k = 0; k < whole.size(); k++) {
                BVHNode nd = whole[k];
                if (nd.lc >= 0) nd.lc += (int)hr.first_node;
                if (nd.rc >= 0) nd.rc += (int)hr.first_node;
                nodes_[hr.first_node + k] = nd;
            }
        }
        bi.host_build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - h0).count();
        root_ = (int)nodes_.size() - 1;
    }
    std::vector<Triangle> sorted;
    sorted.reserve(n);
    for (const Key& k : keys) sorted.push_back(triangles[k.idx]);
    triangles.swap(sorted);
    bi.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (!plain) bi.host_build_ms = bi.total_ms;
    if (info) *info = bi;
}

int BVH::build_node(std::vector<Key>& keys, int l, int r, unsigned depth)
{
    if (l >= r) return -1;
    if (depth > max_depth_) max_depth_ = depth;
    BVHNode node;
    node.AA = mk(FLT_MAX, FLT_MAX, FLT_MAX);
    node.BB = mk(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int i = l; i < r; i++) { // BVH.h:43-52
        const Triangle& t = triangles_[keys[i].idx];
        Vec3 lo = t.get_min(), hi = t.get_max();
        node.AA.x = std::min(lo.x, node.AA.x); node.AA.y = std::min(lo.y, node.AA.y); node.AA.z = std::min(lo.z, node.AA.z);
        node.BB.x = std::max(hi.x, node.BB.x); node.BB.y = std::max(hi.y, node.BB.y); node.BB.z = std::max(hi.z, node.BB.z);
    }
    node.it = l;
    node.n = (unsigned)(r - l);
    if (node.n <= thresh_n_) { // BVH.h:57-61
        nodes_.push_back(node);
        return (int)nodes_.size() - 1;
    }
    Vec3 d = sub(node.BB, node.AA);
    int axis = -1; // BVH.h:65-76: ties x >= y >= z
    if (d.x >= d.y && d.x >= d.z) axis = 0;
    else if (d.y >= d.x && d.y >= d.z) axis = 1;
    else if (d.z >= d.x && d.z >= d.y) axis = 2;
    if (axis >= 0)
        std::sort(keys.begin() + l, keys.begin() + r, [axis](const Key& a, const Key& b) { return a.c[axis] < b.c[axis]; });
    int mid = (l + r) / 2;
    node.lc = build_node(keys, l, mid, depth + 1);
    node.rc = build_node(keys, mid, r, depth + 1);
    nodes_.push_back(node); // post-order
    return (int)nodes_.size() - 1;
}

// ------------------------------------------------------------------- Scene --
Scene::Scene(unsigned width, unsigned height) : width_(width), height_(height) {}
Scene::~Scene() { free(); }

void Scene::add_light_obj(Object& obj)
{
    triangles_.insert(triangles_.end(), obj.get_triangles().begin(), obj.get_triangles().end()); // Scene.h:38-42
    light_objs_.push_back(obj);
    objects_.push_back(std::make_pair(true, obj));
    flat_valid_ = false;
}
void Scene::add_normal_obj(Object& obj)
{
    triangles_.insert(triangles_.end(), obj.get_triangles().begin(), obj.get_triangles().end()); // Scene.h:44-48
    objects_.push_back(std::make_pair(false, obj));
    flat_valid_ = false;
}
void Scene::set_BVH(unsigned thresh_n)
{
    if (triangles_.empty()) throw Error(CRT_ERR_INVALID_ARG, "Scene::set_BVH on an empty scene");
    delete bvh_;
    bvh_ = nullptr;
    bvh_ = new BVH(thresh_n, triangles_); // Scene.h:50-54
    flat_valid_ = false;
}
void Scene::set_BVH_device(unsigned thresh_n, int device, crt_bvh_build_info* info)
{
    if (triangles_.empty()) throw Error(CRT_ERR_INVALID_ARG, "Scene::set_BVH on an empty scene");
    delete bvh_;
    bvh_ = nullptr;
    bvh_ = new BVH(thresh_n, triangles_, device, info);
    flat_valid_ = false;
}
void Scene::free()
{
    delete bvh_;
    bvh_ = nullptr;
    flat_valid_ = false;
}
BVH& Scene::get_bvh()
{
    if (!bvh_) throw Error(CRT_ERR_INVALID_ARG, "Scene::get_bvh before set_BVH");
    return *bvh_;
}

const crt_scene_desc& Scene::flat()
{
    if (flat_valid_) return desc_;
    if (!bvh_) throw Error(CRT_ERR_INVALID_ARG, "Scene::flat before set_BVH");
    f_nodes_.clear(); f_tris_.clear(); f_light_tris_.clear(); f_mats_.clear(); f_lights_.clear();
    std::vector<Material> uniq;
    std::map<std::array<uint32_t, 8>, int32_t> seen; // textured shapes carry one material per triangle: no linear search
    auto mat_id = [&](const Material& m) -> int32_t {
        const Vec3 kd = m.get_kd(), ke = m.get_ke();
        const float v[7] = {kd.x, kd.y, kd.z, ke.x, ke.y, ke.z, m.get_ns()};
        std::array<uint32_t, 8> key;
        std::memcpy(key.data(), v, sizeof(v));
        key[7] = (uint32_t)m.get_mode();
        auto it = seen.find(key);
        if (it != seen.end() && uniq[it->second].same_as(m)) return it->second;
        uniq.push_back(m);
        seen[key] = (int32_t)uniq.size() - 1;
        return (int32_t)uniq.size() - 1;
    };
    auto flat_tri = [&](const Triangle& t) {
        crt_triangle f;
        Vec3 a = t.get_v1(), b = t.get_v2(), c = t.get_v3(), n = t.get_normal();
        f.v1[0] = a.x; f.v1[1] = a.y; f.v1[2] = a.z;
        f.v2[0] = b.x; f.v2[1] = b.y; f.v2[2] = b.z;
        f.v3[0] = c.x; f.v3[1] = c.y; f.v3[2] = c.z;
        f.normal[0] = n.x; f.normal[1] = n.y; f.normal[2] = n.z;
        f.area = t.get_area();
        f.area_of_obj = t.get_area_of_obj();
        f.material = mat_id(t.get_material());
        return f;
    };
    for (const BVHNode& n : bvh_->get_nodes()) {
        crt_bvh_node f;
        f.lc = n.lc; f.rc = n.rc; f.n = n.n; f.it = n.it;
        f.aa[0] = n.AA.x; f.aa[1] = n.AA.y; f.aa[2] = n.AA.z;
        f.bb[0] = n.BB.x; f.bb[1] = n.BB.y; f.bb[2] = n.BB.z;
        f_nodes_.push_back(f);
    }
    for (const Triangle& t : triangles_) f_tris_.push_back(flat_tri(t));
    for (const Object& o : light_objs_) { // DeviceLights keeps per-light, unsorted copies (DeviceLights.cuh:12-25)
        crt_light l;
        l.first_tri = (uint32_t)f_light_tris_.size();
        l.count = (uint32_t)o.get_triangles().size();
        for (const Triangle& t : o.get_triangles()) f_light_tris_.push_back(flat_tri(t));
        f_lights_.push_back(l);
    }
    for (const Material& m : uniq) {
        crt_material f;
        Vec3 kd = m.get_kd(), ke = m.get_ke();
        f.kd[0] = kd.x; f.kd[1] = kd.y; f.kd[2] = kd.z;
        f.ke[0] = ke.x; f.ke[1] = ke.y; f.ke[2] = ke.z;
        f.ns = m.get_ns();
        f.mode = (int32_t)m.get_mode();
        f.has_emit = m.has_emission() ? 1 : 0;
        f_mats_.push_back(f);
    }
    desc_.nodes = f_nodes_.data(); desc_.n_nodes = (uint32_t)f_nodes_.size(); desc_.root = bvh_->get_root_index();
    desc_.tris = f_tris_.data(); desc_.n_tris = (uint32_t)f_tris_.size();
    desc_.materials = f_mats_.data(); desc_.n_materials = (uint32_t)f_mats_.size();
    desc_.light_tris = f_light_tris_.data(); desc_.n_light_tris = (uint32_t)f_light_tris_.size();
    desc_.lights = f_lights_.data(); desc_.n_lights = (uint32_t)f_lights_.size();
    flat_valid_ = true;
    return desc_;
}

// ------------------------------------------------------------------ Loader --
// reference: include/OBJLoader.h:61-203
void Loader::read_OBJ(const char* obj_path, const char* mtl_dir)
{
    shapes_.clear(); vertices_.clear(); n_normals_ = 0;
    std::ifstream obj(obj_path);
    if (!obj.is_open()) throw Error(CRT_ERR_IO, std::string("unable to open OBJ file: ") + obj_path);
    std::map<std::string, std::vector<size_t>> users; // material id -> shapes using it
    std::string mtl_path, line, tok;
    while (std::getline(obj, line)) {
        std::istringstream ls(line);
        std::string prefix;
        ls >> prefix;
        if (prefix == "v") {
            Vec3 p;
            ls >> p.x >> p.y >> p.z;
            vertices_.push_back(p);
        } else if (prefix == "vt") {
            float u = 0.0f, v = 0.0f;
            ls >> u >> v;
            textures_.push_back(u); textures_.push_back(v);
        } else if (prefix == "vn") {
            n_normals_++; // values are never used downstream (Triangle.h:27-28), only their count matters
        } else if (prefix == "f") {
            std::vector<uint64_t> idx;
            while (ls >> tok) {
                uint64_t v = 0;
                try { v = std::stoull(tok.substr(0, tok.find('/'))); }
                catch (const std::exception&) { throw Error(CRT_ERR_PARSE, "bad face index '" + tok + "' in " + obj_path); }
                idx.push_back(v > 0 ? v - 1 : vertices_.size() + v); // OBJLoader.h:105-106
            }
            if (!shapes_.empty()) { // faces before the first usemtl are dropped (OBJLoader.h:120-123)
                if (idx.size() < 3) throw Error(CRT_ERR_PARSE, std::string("face with fewer than 3 vertices in ") + obj_path);
                Shape& s = shapes_.back();
                s.faces.push_back(idx[0]); s.faces.push_back(idx[1]); s.faces.push_back(idx[2]); // Loader.h:62-64
            }
        } else if (prefix == "mtllib") {
            std::string name;
            ls >> name;
            mtl_path = std::string(mtl_dir) + "/" + name;
        } else if (prefix == "usemtl") {
            std::string id;
            ls >> id;
            users[id].push_back(shapes_.size());
            Shape s;
            s.material_id = id;
            shapes_.push_back(s);
        }
    }
    std::ifstream mtl(mtl_path);
    if (!mtl.is_open()) throw Error(CRT_ERR_IO, "unable to open MTL file: " + mtl_path);
    std::vector<size_t> cur;
    while (std::getline(mtl, line)) { // OBJLoader.h:154-201
        std::istringstream ls(line);
        std::string prefix;
        ls >> prefix;
        if (prefix == "newmtl") {
            std::string id;
            ls >> id;
            cur = users[id];
        } else if (prefix == "Kd" || prefix == "Ke") {
            float v[3] = {0, 0, 0};
            ls >> v[0] >> v[1] >> v[2];
            for (size_t s : cur) std::memcpy(prefix == "Kd" ? shapes_[s].kd : shapes_[s].ke, v, sizeof(v));
        } else if (prefix == "Ns") {
            float ns = 1.0f;
            ls >> ns;
            for (size_t s : cur) shapes_[s].ns = ns;
        } else if (prefix == "map_Kd") {
            std::string name;
            ls >> name;
            for (size_t s : cur) { shapes_[s].has_map_kd = true; shapes_[s].map_kd = std::string(mtl_dir) + "/" + name; }
        } // Ks is parsed and discarded by the reference (Loader.h:45,107)
    }
}

const Loader::Texture& Loader::texture(const std::string& path) const
{
    auto it = tex_cache_.find(path);
    if (it != tex_cache_.end()) return it->second;
    crtimg::Image img;
    const std::string err = crtimg::load(path, img);
    if (!err.empty()) throw Error(err.rfind("cannot open", 0) == 0 ? CRT_ERR_IO : CRT_ERR_UNSUPPORTED, "map_Kd: " + err);
    Texture t;
    t.x = img.width; t.y = img.height; t.comp = img.comp;
    t.px.swap(img.px);
    return tex_cache_.emplace(path, std::move(t)).first->second;
}

// reference: include/Loader.h:40-124
void Loader::load_object(uint64_t index, std::vector<Triangle>& triangles, std::vector<Triangle>& light_triangles) const
{
    triangles.clear();
    light_triangles.clear();
    if (index >= shapes_.size()) throw Error(CRT_ERR_INVALID_ARG, "Loader::load_object: shape index out of range");
    const Shape& s = shapes_[index];
    // Loader.h:55-59: stbi_load(map_kd, &height, &width, &channel, 0) -- the reference receives the image's x in `height`
    // and its y in `width`, and indexes with those names below; restated literally
    const Texture* tex = s.has_map_kd ? &texture(s.map_kd) : nullptr;
    const int width = tex ? tex->y : 0, height = tex ? tex->x : 0, channel = tex ? tex->comp : 0;
    Material m(mk(s.kd[0], s.kd[1], s.kd[2]), mk(s.ke[0], s.ke[1], s.ke[2]), s.ns, s.ns > 1 ? SPECULAR : DIFFUSE); // Loader.h:107
    for (size_t f = 0; f + 2 < s.faces.size(); f += 3) {
        uint64_t a = s.faces[f], b = s.faces[f + 1], c = s.faces[f + 2];
        if (a >= vertices_.size() || b >= vertices_.size() || c >= vertices_.size())
            throw Error(CRT_ERR_PARSE, "face references a vertex that does not exist");
        // Loader.h:70-72 indexes normals[] with the vertex index: the reference reads out of bounds otherwise
        if (a >= n_normals_ || b >= n_normals_ || c >= n_normals_)
            throw Error(CRT_ERR_PARSE, "OBJ must carry one vn per v (reference Loader.h:70-72 indexes normals by vertex index)");
        if (tex) {
            // Loader.h:81-103: kd = mean of the texels at the three vertices; textures[] is indexed by the VERTEX index too
            if (2 * a + 1 >= textures_.size() || 2 * b + 1 >= textures_.size() || 2 * c + 1 >= textures_.size())
                throw Error(CRT_ERR_PARSE, "textured OBJ must carry one vt per v (reference Loader.h:81-83 indexes textures by vertex index)");
            Vec3 k[3];
            const uint64_t vi[3] = {a, b, c};
            for (int q = 0; q < 3; q++) {
                float intpart;
                const float tu = textures_[2 * vi[q]], tv = textures_[2 * vi[q] + 1];
                const int u = (int)(std::modf(std::modf(tu, &intpart) + 1, &intpart) * (width - 1));  // Loader.h:86,92,98
                const int v = (int)(std::modf(std::modf(tv, &intpart) + 1, &intpart) * (height - 1)); // Loader.h:87,93,99
                const long long offset = ((long long)v * width + u) * channel;                           // Loader.h:88
                if (offset < 0 || (size_t)offset + 2 >= tex->px.size())
                    throw Error(CRT_ERR_PARSE, "texture lookup outside the image (the reference indexes with width and height swapped, Loader.h:58): " + s.map_kd);
                // Eigen::Vector3f(tex[o], tex[o+1], tex[o+2]) / 255.  (Loader.h:89; evaluated eagerly)
                k[q] = mk((float)tex->px[offset] / 255.0f, (float)tex->px[offset + 1] / 255.0f, (float)tex->px[offset + 2] / 255.0f);
            }
            // kd = (kd_1 + kd_2 + kd_3) / 3  (Loader.h:103)
            const Vec3 kd = mk(((k[0].x + k[1].x) + k[2].x) / 3.0f, ((k[0].y + k[1].y) + k[2].y) / 3.0f, ((k[0].z + k[1].z) + k[2].z) / 3.0f);
            m = Material(kd, mk(s.ke[0], s.ke[1], s.ke[2]), s.ns, s.ns > 1 ? SPECULAR : DIFFUSE);
        }
        Triangle t(vertices_[a], vertices_[b], vertices_[c], m);
        (m.has_emission() ? light_triangles : triangles).push_back(t); // Loader.h:119-122
    }
}

// ------------------------------------------------------------------ Camera --
void get_inverse_view_matrix(const float eye[3], const float lookat[3], const float up[3], float out[9])
{
    Vec3 f = unit(sub(mk(lookat[0], lookat[1], lookat[2]), mk(eye[0], eye[1], eye[2]))); // Camera.h:12
    Vec3 r = unit(cross3(mk(up[0], up[1], up[2]), f));                                   // Camera.h:15
    Vec3 u = unit(cross3(f, r));                                                         // Camera.h:18
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
    out[3] = u.x; out[4] = u.y; out[5] = u.z;
    out[6] = f.x; out[7] = f.y; out[8] = f.z;
}

// -------------------------------------------------------------------- JSON --
namespace {
struct JValue {
    enum Kind { NUL, NUM, STR, ARR, OBJ, BOOL } kind = NUL;
    double num = 0;
    std::string str;
    std::vector<JValue> arr;
    std::vector<std::pair<std::string, JValue>> obj;
    const JValue& at(const std::string& k) const
    {
        for (const auto& kv : obj)
            if (kv.first == k) return kv.second;
        throw Error(CRT_ERR_PARSE, "config: missing key '" + k + "'");
    }
    double number() const
    {
        if (kind != NUM) throw Error(CRT_ERR_PARSE, "config: expected a number");
        return num;
    }
};
struct JParser {
    const std::string& s;
    size_t p = 0;
    explicit JParser(const std::string& src) : s(src) {}
    void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) p++; }
    [[noreturn]] void fail(const char* what) { throw Error(CRT_ERR_PARSE, std::string("config JSON: ") + what + " at offset " + std::to_string(p)); }
    JValue value()
    {
        ws();
        if (p >= s.size()) fail("unexpected end");
        JValue v;
        char c = s[p];
        if (c == '{') {
            v.kind = JValue::OBJ; p++; ws();
            if (p < s.size() && s[p] == '}') { p++; return v; }
            for (;;) {
                ws();
                JValue k = value();
                if (k.kind != JValue::STR) fail("object key must be a string");
                ws();
                if (p >= s.size() || s[p] != ':') fail("expected ':'");
                p++;
                v.obj.push_back(std::make_pair(k.str, value()));
                ws();
                if (p < s.size() && s[p] == ',') { p++; continue; }
                if (p < s.size() && s[p] == '}') { p++; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            v.kind = JValue::ARR; p++; ws();
            if (p < s.size() && s[p] == ']') { p++; return v; }
            for (;;) {
                v.arr.push_back(value());
                ws();
                if (p < s.size() && s[p] == ',') { p++; continue; }
                if (p < s.size() && s[p] == ']') { p++; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v.kind = JValue::STR; p++;
            while (p < s.size() && s[p] != '"') {
                if (s[p] == '\\' && p + 1 < s.size()) {
                    char e = s[p + 1];
                    v.str.push_back(e == 'n' ? '\n' : e == 't' ? '\t' : e);
                    p += 2;
                } else v.str.push_back(s[p++]);
            }
            if (p >= s.size()) fail("unterminated string");
            p++;
        } else if (c == 't' && s.compare(p, 4, "true") == 0) { v.kind = JValue::BOOL; v.num = 1; p += 4; }
        else if (c == 'f' && s.compare(p, 5, "false") == 0) { v.kind = JValue::BOOL; v.num = 0; p += 5; }
        else if (c == 'n' && s.compare(p, 4, "null") == 0) { p += 4; }
        else {
            const char* b = s.c_str() + p;
            char* e = nullptr;
            v.num = std::strtod(b, &e);
            if (e == b) fail("unexpected character");
            v.kind = JValue::NUM;
            p += (size_t)(e - b);
        }
        return v;
    }
};
void copy_path(char (&dst)[512], const std::string& src)
{
    if (src.size() >= sizeof(dst)) throw Error(CRT_ERR_INVALID_ARG, "config: path longer than 511 bytes");
    std::memset(dst, 0, sizeof(dst));
    std::memcpy(dst, src.data(), src.size());
}
} // namespace

// reference: src/main.cu:67-90
crt_task load_task(const std::string& config_path, TaskObjs* all_objs)
{
    std::ifstream f(config_path);
    if (!f.is_open()) throw Error(CRT_ERR_IO, "unable to open config: " + config_path);
    std::stringstream ss;
    ss << f.rdbuf();
    std::string text = ss.str();
    JParser jp(text);
    JValue j = jp.value();
    if (j.kind != JValue::OBJ) throw Error(CRT_ERR_PARSE, "config: top level must be an object");
    crt_task t;
    std::memset(&t, 0, sizeof(t));
    const JValue& objs = j.at("OBJ_paths");
    if (objs.kind != JValue::ARR || objs.arr.empty()) throw Error(CRT_ERR_PARSE, "config: OBJ_paths must be a non-empty array");
    t.n_objs = (uint32_t)objs.arr.size();
    if (all_objs) all_objs->clear();
    for (uint32_t i = 0; i < t.n_objs; i++) {
        const std::string& o = objs.arr[i].at("OBJ_path").str;
        const std::string& m = objs.arr[i].at("MTL_dir").str;
        if (i < 8) { copy_path(t.obj_path[i], o); copy_path(t.mtl_dir[i], m); }   // (the POD's slots; the rest through all_objs / crt_task_obj)
        if (all_objs) all_objs->push_back(std::make_pair(o, m));
    }
    auto vec = [&](const char* key, float out[3]) {
        const JValue& v = j.at(key);
        out[0] = (float)v.at("x").number(); out[1] = (float)v.at("y").number(); out[2] = (float)v.at("z").number();
    };
    vec("eye_pos", t.eye_pos);
    vec("lookat", t.lookat);
    vec("up", t.up);
    t.fov_y = (float)j.at("fov_y").number();
    t.width = (uint32_t)j.at("width").number();
    t.height = (uint32_t)j.at("height").number();
    t.bvh_thresh_n = (uint32_t)j.at("bvh_thresh_n").number();
    t.p_rr = (float)j.at("P_RR").number();
    t.spp = (uint32_t)j.at("spp").number();
    t.light_sample_n = (uint32_t)j.at("light_sample_n").number();
    return t;
}

// reference: src/main.cu:122-145
void load_task_scene(const crt_task& task, Scene& scene, const std::string& base_dir, const TaskObjs* all_objs)
{
    if (task.n_objs > 8 && !all_objs)
        throw Error(CRT_ERR_INVALID_ARG, "load_task_scene: a task of more than 8 OBJ files needs the list load_task returns");
    // (a list is indexed up to n_objs below: one that belongs to another task -- or an empty one -- must not be read past its end)
    if (all_objs && all_objs->size() != task.n_objs)
        throw Error(CRT_ERR_INVALID_ARG, "load_task_scene: the OBJ list has " + std::to_string(all_objs->size()) + " entries, the task " + std::to_string(task.n_objs));
    auto resolve = [&](const char* p) {
        std::string s(p);
        if (!s.empty() && s[0] == '/') return s;
        return base_dir.empty() ? s : base_dir + "/" + s;
    };
    for (uint32_t i = 0; i < task.n_objs; i++) {
        Loader loader;
        std::string obj = resolve(all_objs ? (*all_objs)[i].first.c_str() : task.obj_path[i]), mtl = resolve(all_objs ? (*all_objs)[i].second.c_str() : task.mtl_dir[i]);
        loader.read_OBJ(obj.c_str(), mtl.c_str());
        std::vector<Triangle> tris, light_tris;
        for (uint64_t s = 0; s < loader.size(); s++) {
            loader.load_object(s, tris, light_tris);
            if (!tris.empty()) { Object o(tris); scene.add_normal_obj(o); }
            if (!light_tris.empty()) { Object o(light_tris); scene.add_light_obj(o); }
        }
    }
}

// ------------------------------------------------------------------ Render --
Render::Render(Scene* scene, unsigned spp, float P_RR, unsigned light_sample_n, int device)
    : scene_(scene), spp_(spp), light_sample_n_(light_sample_n), P_RR_(P_RR)
{
    if (!scene) throw Error(CRT_ERR_INVALID_ARG, "Render: null scene");
    const crt_scene_desc& d = scene->flat();
    int rc = crt_scene_create(&d, device, &device_scene_);
    if (rc != CRT_OK) throw Error(rc, std::string("Render: crt_scene_create failed: ") + crt_last_error());
    frame_buffer_.assign((size_t)3 * scene->get_pixels(), 0);
    mean_buffer_.assign((size_t)3 * scene->get_pixels(), 0.0f);
}
// Several devices (the reference stops at device 0, src/main.cu:92-105): one replica of the scene per entry of `devices`, the
// frame sharded by interleaved pixel tiles and gathered with one RCCL all-gather (include/crt.h, crt_multi).
Render::Render(Scene* scene, unsigned spp, float P_RR, unsigned light_sample_n, const std::vector<int>& devices, uint32_t gather)
    : scene_(scene), spp_(spp), light_sample_n_(light_sample_n), P_RR_(P_RR)
{
    if (!scene) throw Error(CRT_ERR_INVALID_ARG, "Render: null scene");
    if (devices.empty()) throw Error(CRT_ERR_INVALID_ARG, "Render: empty device list");
    const crt_scene_desc& d = scene->flat();
    int rc = crt_multi_create(&d, devices.data(), (uint32_t)devices.size(), gather, &multi_);
    if (rc != CRT_OK) throw Error(rc, std::string("Render: crt_multi_create failed: ") + crt_last_error());
    rank_stats_.resize(devices.size());
    frame_buffer_.assign((size_t)3 * scene->get_pixels(), 0);
    mean_buffer_.assign((size_t)3 * scene->get_pixels(), 0.0f);
}
Render::~Render() { free(); }

void Render::free()
{
    if (device_scene_) { crt_scene_destroy(device_scene_); device_scene_ = nullptr; }
    if (multi_) { crt_multi_destroy(multi_); multi_ = nullptr; }
}
void Render::set_width(const unsigned& w)
{
    scene_->set_width(w);
    frame_buffer_.assign((size_t)3 * scene_->get_pixels(), 0);
    mean_buffer_.assign((size_t)3 * scene_->get_pixels(), 0.0f);
}
void Render::set_height(const unsigned& h)
{
    scene_->set_height(h);
    frame_buffer_.assign((size_t)3 * scene_->get_pixels(), 0);
    mean_buffer_.assign((size_t)3 * scene_->get_pixels(), 0.0f);
}

void Render::run_view(const float eye_pos[3], const float inv_view_mat[9], float fovY)
{
    if (!device_scene_ && !multi_) throw Error(CRT_ERR_INVALID_ARG, "Render::run_view after free()");
    crt_camera cam;
    std::memcpy(cam.eye, eye_pos, sizeof(cam.eye));
    std::memcpy(cam.inv_view, inv_view_mat, sizeof(cam.inv_view));
    cam.fov_y = fovY;
    crt_params p;
    std::memset(&p, 0, sizeof(p));
    p.width = scene_->get_width(); p.height = scene_->get_height();
    p.spp = spp_; p.p_rr = P_RR_; p.light_sample_n = (int32_t)light_sample_n_;
    p.seed = seed_; p.rank = 0; p.world = 1; p.traversal = traversal_; p.flags = flags_ & (CRT_FLAG_TRACE_ALL | CRT_FLAG_BOUNDED_RADIANCE);
    int rc;
    if (multi_) {
        rc = crt_multi_render(multi_, &cam, &p, frame_buffer_.data(), mean_buffer_.data(), rank_stats_.data(), &multi_info_);
        if (rc == CRT_OK) { // totals over the ranks; times of the slowest one
            stats_ = rank_stats_[0];
            for (size_t r = 1; r < rank_stats_.size(); r++) {
                const crt_stats& o = rank_stats_[r];
                stats_.paths += o.paths; stats_.rays += o.rays; stats_.shadow_rays += o.shadow_rays; stats_.probe_rays += o.probe_rays;
                stats_.rays_untraced += o.rays_untraced;
                stats_.kernel_ms = std::max(stats_.kernel_ms, o.kernel_ms); stats_.total_ms = std::max(stats_.total_ms, o.total_ms);
            }
        }
    } else {
        rc = crt_render(device_scene_, &cam, &p, frame_buffer_.data(), mean_buffer_.data(), &stats_);
    }
    if (rc != CRT_OK) throw Error(rc, std::string("Render::run_view failed: ") + crt_last_error());
}

void Render::save_frame_buffer(const char* save_path) const
{
    int rc = crt_write_png(save_path, scene_->get_width(), scene_->get_height(), frame_buffer_.data());
    if (rc != CRT_OK) throw Error(rc, std::string("save_frame_buffer failed: ") + crt_last_error());
}

} // namespace crt

// =========================================================== C ABI (host) ==
namespace {
thread_local std::string g_last_error;
}
extern "C" void crt_set_last_error_(const char* msg) { g_last_error = msg ? msg : ""; }

struct crt_host_scene {
    crt::Scene scene;
    crt_host_scene(uint32_t w, uint32_t h) : scene(w, h) {}
};

#define CRT_HOST_TRY(...)                                                                   \
    try { __VA_ARGS__; return CRT_OK; }                                                            \
    catch (const crt::Error& e) { g_last_error = e.what(); return e.status; }               \
    catch (const std::bad_alloc&) { g_last_error = "out of host memory"; return CRT_ERR_OOM; } \
    catch (const std::exception& e) { g_last_error = e.what(); return CRT_ERR_INVALID_ARG; }

extern "C" {

const char* crt_strerror(int status)
{
    switch (status) {
    case CRT_OK: return "ok";
    case CRT_ERR_INVALID_ARG: return "invalid argument";
    case CRT_ERR_NO_DEVICE: return "no HIP device";
    case CRT_ERR_HIP: return "HIP runtime error";
    case CRT_ERR_UNSUPPORTED: return "unsupported";
    case CRT_ERR_IO: return "I/O error";
    case CRT_ERR_PARSE: return "parse error";
    case CRT_ERR_OOM: return "out of memory";
    default: return "unknown status";
    }
}
const char* crt_last_error(void) { return g_last_error.c_str(); }
int crt_abi_version(void) { return CRT_ABI_VERSION; }

int crt_host_scene_create(uint32_t width, uint32_t height, crt_host_scene** out)
{
    if (!out || width == 0 || height == 0) { g_last_error = "crt_host_scene_create: bad arguments"; return CRT_ERR_INVALID_ARG; }
    CRT_HOST_TRY(*out = new crt_host_scene(width, height));
}
int crt_host_scene_destroy(crt_host_scene* s)
{
    delete s;
    return CRT_OK;
}
int crt_host_scene_add_obj(crt_host_scene* s, const char* obj_path, const char* mtl_dir)
{
    if (!s || !obj_path || !mtl_dir) { g_last_error = "crt_host_scene_add_obj: null argument"; return CRT_ERR_INVALID_ARG; }
    CRT_HOST_TRY({
        crt::Loader loader;
        loader.read_OBJ(obj_path, mtl_dir);
        std::vector<crt::Triangle> tris, light_tris;
        for (uint64_t i = 0; i < loader.size(); i++) {
            loader.load_object(i, tris, light_tris);
            if (!tris.empty()) { crt::Object o(tris); s->scene.add_normal_obj(o); }
            if (!light_tris.empty()) { crt::Object o(light_tris); s->scene.add_light_obj(o); }
        }
    });
}
int crt_host_scene_set_bvh(crt_host_scene* s, uint32_t thresh_n)
{
    if (!s) { g_last_error = "crt_host_scene_set_bvh: null scene"; return CRT_ERR_INVALID_ARG; }
    CRT_HOST_TRY(s->scene.set_BVH(thresh_n));
}
int crt_host_scene_set_bvh_device(crt_host_scene* s, uint32_t thresh_n, int device, crt_bvh_build_info* info)
{
    if (!s) { g_last_error = "crt_host_scene_set_bvh_device: null scene"; return CRT_ERR_INVALID_ARG; }
    CRT_HOST_TRY(s->scene.set_BVH_device(thresh_n, device, info));
}
int crt_host_scene_desc(const crt_host_scene* s, crt_scene_desc* out)
{
    if (!s || !out) { g_last_error = "crt_host_scene_desc: null argument"; return CRT_ERR_INVALID_ARG; }
    CRT_HOST_TRY(*out = const_cast<crt_host_scene*>(s)->scene.flat());
}
int crt_host_scene_num_objects(const crt_host_scene* s, uint32_t* n)
{
    if (!s || !n) { g_last_error = "crt_host_scene_num_objects: null argument"; return CRT_ERR_INVALID_ARG; }
    *n = (uint32_t)s->scene.get_objects().size();
    return CRT_OK;
}
int crt_host_scene_object(const crt_host_scene* s, uint32_t index, float* area, int32_t* is_light, uint32_t* n_tris)
{
    if (!s || index >= s->scene.get_objects().size()) { g_last_error = "crt_host_scene_object: bad index"; return CRT_ERR_INVALID_ARG; }
    const auto& o = s->scene.get_objects()[index];
    if (area) *area = o.second.get_area();
    if (is_light) *is_light = o.first ? 1 : 0;
    if (n_tris) *n_tris = (uint32_t)o.second.get_triangles().size();
    return CRT_OK;
}
int crt_inverse_view(const float eye[3], const float lookat[3], const float up[3], float out[9])
{
    if (!eye || !lookat || !up || !out) { g_last_error = "crt_inverse_view: null argument"; return CRT_ERR_INVALID_ARG; }
    crt::get_inverse_view_matrix(eye, lookat, up, out);
    return CRT_OK;
}
int crt_image_load(const char* path, int32_t* x, int32_t* y, int32_t* comp, uint8_t* out, uint64_t cap)
{
    if (!path || !x || !y || !comp) { g_last_error = "crt_image_load: null argument"; return CRT_ERR_INVALID_ARG; }
    try {
        crtimg::Image img;
        const std::string err = crtimg::load(path, img);
        if (!err.empty()) { g_last_error = "crt_image_load: " + err; return err.rfind("cannot open", 0) == 0 ? CRT_ERR_IO : CRT_ERR_UNSUPPORTED; }
        *x = img.width; *y = img.height; *comp = img.comp;
        if (out) {
            if (cap < img.px.size()) { g_last_error = "crt_image_load: buffer too small"; return CRT_ERR_INVALID_ARG; }
            std::memcpy(out, img.px.data(), img.px.size());
        }
        return CRT_OK;
    } catch (const std::bad_alloc&) { // (a file may honestly announce more pixels than the host has memory for)
        g_last_error = "crt_image_load: out of host memory";
        return CRT_ERR_OOM;
    }
}
int crt_task_load(const char* path, crt_task* out)
{
    if (!path || !out) { g_last_error = "crt_task_load: null argument"; return CRT_ERR_INVALID_ARG; }
    CRT_HOST_TRY(*out = crt::load_task(path));
}
int crt_task_obj(const char* path, uint32_t index, char* obj_path, char* mtl_dir, uint32_t cap)
{
    if (!path || !obj_path || !mtl_dir) { g_last_error = "crt_task_obj: null argument"; return CRT_ERR_INVALID_ARG; }
    CRT_HOST_TRY({
        crt::TaskObjs all;
        (void)crt::load_task(path, &all);
        if (index >= all.size()) throw crt::Error(CRT_ERR_INVALID_ARG, "crt_task_obj: index beyond the file's OBJ_paths");
        if (all[index].first.size() + 1 > cap || all[index].second.size() + 1 > cap) throw crt::Error(CRT_ERR_INVALID_ARG, "crt_task_obj: buffer too small");
        std::memcpy(obj_path, all[index].first.c_str(), all[index].first.size() + 1);
        std::memcpy(mtl_dir, all[index].second.c_str(), all[index].second.size() + 1);
    });
}

// ---- PNG (stored deflate blocks; replaces stbi_write_png, Render.cuh:489-493) ----
static uint32_t crc_table[256];
static bool crc_ready = false;
static uint32_t crc32_update(uint32_t c, const uint8_t* p, size_t n)
{
    if (!crc_ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t k = i;
            for (int j = 0; j < 8; j++) k = (k & 1) ? 0xEDB88320u ^ (k >> 1) : k >> 1;
            crc_table[i] = k;
        }
        crc_ready = true;
    }
    for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return c;
}
static void put32(std::vector<uint8_t>& v, uint32_t x)
{
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
static void chunk(std::vector<uint8_t>& png, const char* type, const std::vector<uint8_t>& data)
{
    put32(png, (uint32_t)data.size());
    size_t start = png.size();
    png.insert(png.end(), type, type + 4);
    png.insert(png.end(), data.begin(), data.end());
    uint32_t c = crc32_update(0xffffffffu, png.data() + start, png.size() - start) ^ 0xffffffffu;
    put32(png, c);
}
int crt_write_png(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb)
{
    if (!path || !rgb || width == 0 || height == 0) { g_last_error = "crt_write_png: bad arguments"; return CRT_ERR_INVALID_ARG; }
    std::vector<uint8_t> raw;
    raw.reserve((size_t)height * (3 * width + 1));
    for (uint32_t y = 0; y < height; y++) {
        raw.push_back(0); // filter: none
        raw.insert(raw.end(), rgb + (size_t)y * width * 3, rgb + (size_t)(y + 1) * width * 3);
    }
    std::vector<uint8_t> z;
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    size_t pos = 0;
    while (pos < raw.size() || raw.empty()) {
        size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n >= raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xff)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xff)); z.push_back((uint8_t)((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        for (size_t i = 0; i < n; i++) { a = (a + raw[pos + i]) % 65521u; b = (b + a) % 65521u; }
        pos += n;
        if (raw.empty()) break;
    }
    put32(z, (b << 16) | a);
    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> ihdr;
    put32(ihdr, width); put32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(png, "IHDR", ihdr);
    chunk(png, "IDAT", z);
    chunk(png, "IEND", std::vector<uint8_t>());
    FILE* f = std::fopen(path, "wb");
    if (!f) { g_last_error = std::string("cannot open for writing: ") + path; return CRT_ERR_IO; }
    size_t w = std::fwrite(png.data(), 1, png.size(), f);
    std::fclose(f);
    if (w != png.size()) { g_last_error = std::string("short write: ") + path; return CRT_ERR_IO; }
    return CRT_OK;
}

} // extern "C"
