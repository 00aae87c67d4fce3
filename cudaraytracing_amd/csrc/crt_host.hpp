// cudaraytracing_amd/csrc/crt_host.hpp
//
// Host side of the path tracer, above the C ABI of include/crt.h: the C++
// classes a user of the reference knows -- Material, Triangle, Object, BVH,
// Scene, Loader, Task, Render and get_inverse_view_matrix -- re-implemented
// for a flat, upload-ready data model (structure-of-arrays friendly PODs,
// de-duplicated material table, index-permutation BVH build).
//
// Same names, argument meaning and results as the reference classes
// (include/Scene.h, Object.h, Triangle.h, Material.h, BVH.h, Loader.h,
// OBJLoader.h, Camera.h, Render.cuh:357-557); error behaviour is stricter:
// failures throw crt::Error / return a crt_status instead of printing and
// continuing.
#ifndef CRT_HOST_HPP
#define CRT_HOST_HPP

#include "../../include/crt.h"

#include <cstdint>
#include <stdexcept>
#include <map>
#include <string>
#include <vector>

namespace crt {

struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string& what) : std::runtime_error(what), status(st) {}
};

struct Vec3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
};

enum Illum { DIFFUSE = 0, SPECULAR = 1 }; // reference: include/Material.h:7-10

// reference: include/Material.h:11-77 (ks / ka are never read downstream)
class Material {
public:
    Material();
    Material(Vec3 kd, Vec3 ke, float ns, Illum mode);
    Vec3 get_kd() const { return kd_; }
    Vec3 get_ke() const { return ke_; }
    float get_ns() const { return ns_; }
    bool has_emission() const { return has_emit_; }
    Illum get_mode() const { return mode_; }
    bool same_as(const Material& o) const;

private:
    Vec3 kd_, ke_;
    float ns_;
    bool has_emit_;
    Illum mode_;
};

// reference: include/Triangle.h:9-132
class Triangle {
public:
    Triangle(Vec3 v1, Vec3 v2, Vec3 v3, const Material& m);
    Vec3 get_v1() const { return v1_; }
    Vec3 get_v2() const { return v2_; }
    Vec3 get_v3() const { return v3_; }
    Vec3 get_normal() const { return normal_; }
    Vec3 get_center() const { return center_; }
    float get_area() const { return area_; }
    float get_area_of_obj() const { return area_of_obj_; }
    void set_area_of_obj(float a) { area_of_obj_ = a; }
    const Material& get_material() const { return material_; }
    Vec3 get_min() const { return lo_; }
    Vec3 get_max() const { return hi_; }

private:
    Vec3 v1_, v2_, v3_, center_, normal_, lo_, hi_;
    Material material_;
    float area_ = 0.0f, area_of_obj_ = 0.0f;
};

// reference: include/Object.h:7-31
class Object {
public:
    explicit Object(const std::vector<Triangle>& ts);
    std::vector<Triangle>& get_triangles() { return triangles_; }
    const std::vector<Triangle>& get_triangles() const { return triangles_; }
    float get_area() const { return area_; }

private:
    std::vector<Triangle> triangles_;
    float area_ = 0.0f;
};

// reference: include/BVH.h:9-20
struct BVHNode {
    int lc = -1, rc = -1;
    unsigned n = 0;
    int it = -1;
    Vec3 AA, BB;
};

// reference: include/BVH.h:22-120.  Reorders `triangles` in place, emits
// nodes in post-order (root last).
class BVH {
public:
    BVH(unsigned thresh_n, std::vector<Triangle>& triangles);
    // the same tree built on GPU `device` (csrc/crt_bvh_build.hip), byte-identical node and triangle arrays; ranges with equal
    // sort keys are finished by build_node on the host
    BVH(unsigned thresh_n, std::vector<Triangle>& triangles, int device, crt_bvh_build_info* info);
    int get_root_index() const { return root_; }
    size_t get_nodes_size() const { return nodes_.size(); }
    const std::vector<BVHNode>& get_nodes() const { return nodes_; }
    const std::vector<Triangle>& get_triangles() const { return triangles_; }
    size_t get_triangles_size() const { return triangles_.size(); }
    unsigned max_depth() const { return max_depth_; }

private:
    struct Key { float c[3]; uint32_t idx; };
    int build_node(std::vector<Key>& keys, int l, int r, unsigned depth);
    unsigned thresh_n_;
    int root_ = -1;
    unsigned max_depth_ = 0;
    std::vector<Triangle>& triangles_;
    std::vector<BVHNode> nodes_;
};

// reference: include/Scene.h:16-102
class Scene {
public:
    Scene(unsigned width, unsigned height);
    ~Scene();
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    void add_light_obj(Object& obj);
    void add_normal_obj(Object& obj);
    void set_BVH(unsigned thresh_n);
    void set_BVH_device(unsigned thresh_n, int device, crt_bvh_build_info* info = nullptr);
    void free();
    unsigned get_height() const { return height_; }
    unsigned get_width() const { return width_; }
    unsigned get_pixels() const { return width_ * height_; }
    void set_height(unsigned h) { height_ = h; }
    void set_width(unsigned w) { width_ = w; }
    BVH& get_bvh();
    bool has_bvh() const { return bvh_ != nullptr; }
    std::vector<Object>& get_light_objs() { return light_objs_; }
    std::vector<Triangle>& get_triangles() { return triangles_; }
    // every object in creation order (normal and light interleaved) with a light flag
    const std::vector<std::pair<bool, Object>>& get_objects() const { return objects_; }

    // Flat, upload-ready view (valid until the scene changes); builds the
    // de-duplicated material table on first use after set_BVH.
    const crt_scene_desc& flat();

private:
    unsigned width_, height_;
    BVH* bvh_ = nullptr;
    std::vector<Object> light_objs_;
    std::vector<std::pair<bool, Object>> objects_;
    std::vector<Triangle> triangles_;
    // flat storage
    bool flat_valid_ = false;
    std::vector<crt_bvh_node> f_nodes_;
    std::vector<crt_triangle> f_tris_, f_light_tris_;
    std::vector<crt_material> f_mats_;
    std::vector<crt_light> f_lights_;
    crt_scene_desc desc_{};
};

// reference: include/Loader.h:13-131 + include/OBJLoader.h:12-229
class Loader {
public:
    void read_OBJ(const char* obj_path, const char* mtl_dir);
    void load_object(uint64_t index, std::vector<Triangle>& triangles, std::vector<Triangle>& light_triangles) const;
    uint64_t size() const { return shapes_.size(); }

private:
    struct Shape {
        std::string material_id;
        std::vector<uint64_t> faces; // 3 vertex indices per face (only the first three of an `f` line are kept)
        float kd[3] = {0, 0, 0}, ke[3] = {0, 0, 0};
        float ns = 1.0f;
        bool has_map_kd = false;
        std::string map_kd;          // mtl_dir + "/" + file name (OBJLoader.h:184-193)
    };
    struct Texture {                 // what stbi_load returns (Loader.h:58): x, y, components, 8-bit samples
        int x = 0, y = 0, comp = 0;
        std::vector<uint8_t> px;
    };
    const Texture& texture(const std::string& path) const;
    std::vector<Shape> shapes_;
    std::vector<Vec3> vertices_;
    std::vector<float> textures_;    // u, v per `vt` line (OBJLoader.h:88-93)
    size_t n_normals_ = 0;
    mutable std::map<std::string, Texture> tex_cache_;
};

// reference: include/Camera.h:9-36; out is column-major
void get_inverse_view_matrix(const float eye[3], const float lookat[3], const float up[3], float out[9]);

// reference: src/main.cu:40-90.  all_objs (optional): every (OBJ_path, MTL_dir) pair of the file -- the crt_task POD holds the first eight
typedef std::vector<std::pair<std::string, std::string>> TaskObjs;
crt_task load_task(const std::string& config_path, TaskObjs* all_objs = nullptr);
// loads every OBJ of a task into `scene` the way render_view does (src/main.cu:122-145); all_objs as returned by load_task when the
// file names more than eight
void load_task_scene(const crt_task& task, Scene& scene, const std::string& base_dir, const TaskObjs* all_objs = nullptr);

// reference: include/Render.cuh:357-557.  Drives the device layer through the C ABI.
class Render {
public:
    Render(Scene* scene, unsigned spp = 16, float P_RR = 0.8f, unsigned light_sample_n = 1, int device = 0);
    // one rank per entry of `devices` (crt_multi: tile shards, one RCCL all-gather); gather = CRT_GATHER_*
    Render(Scene* scene, unsigned spp, float P_RR, unsigned light_sample_n, const std::vector<int>& devices, uint32_t gather = CRT_GATHER_AUTO);
    ~Render();
    Render(const Render&) = delete;
    Render& operator=(const Render&) = delete;
    // run_view(eye_pos, inv_view_mat, fovY): inv_view column-major, fovY in radians
    void run_view(const float eye_pos[3], const float inv_view_mat[9], float fovY);
    void free();
    void save_frame_buffer(const char* save_path) const;
    unsigned char* get_frame_buffer() const { return const_cast<unsigned char*>(frame_buffer_.data()); }
    const float* get_mean_buffer() const { return mean_buffer_.data(); }
    void set_spp(const int& spp) { spp_ = (unsigned)spp; }
    void set_P_RR(const float& p) { P_RR_ = p; }
    void set_light_sample_n(const int& n) { light_sample_n_ = (unsigned)n; }
    void set_seed(uint64_t s) { seed_ = s; }
    void set_traversal(uint32_t t) { traversal_ = t; }
    void set_flags(uint32_t f) { flags_ = f; } // CRT_FLAG_* of crt_params that a caller may choose: CRT_FLAG_TRACE_ALL, CRT_FLAG_BOUNDED_RADIANCE
    void set_width(const unsigned& w);
    void set_height(const unsigned& h);
    const crt_stats& last_stats() const { return stats_; }
    // multi-device renders: what the exchange ran on (n_ranks == 0 for a single-device Render) and the ranks' own statistics
    const crt_multi_info& last_multi_info() const { return multi_info_; }
    const std::vector<crt_stats>& last_rank_stats() const { return rank_stats_; }

private:
    Scene* scene_;
    unsigned spp_, light_sample_n_;
    float P_RR_;
    uint64_t seed_ = 0;
    uint32_t traversal_ = CRT_TRAVERSAL_EXACT;
    uint32_t flags_ = 0;
    crt_scene* device_scene_ = nullptr;
    crt_multi* multi_ = nullptr;
    crt_multi_info multi_info_{};
    std::vector<crt_stats> rank_stats_;
    std::vector<unsigned char> frame_buffer_;
    std::vector<float> mean_buffer_;
    crt_stats stats_{};
};

} // namespace crt
#endif
