// tests/shim_decls/Scene.h -- TEST-ONLY declarations of the reference interfaces that the shim of INTEGRATION.md section 2 calls
// (reference: include/Scene.h:28-101, include/BVH.h:9-21,86-119, include/Triangle.h:58-131, include/Material.h:7-76,
// include/Object.h:27): names and signatures only, so that tests/test_host_layer.py can put the documented binding through a
// compiler against include/crt.h.  The reference's own headers cannot be used for that (Global.h pulls curand_kernel.h).
#ifndef SHIM_DECLS_SCENE_H
#define SHIM_DECLS_SCENE_H
#include <cstddef>
#include <vector>
#include "Eigen/Dense"

enum Illum { DIFFUSE, SPECULAR };
class Material {
public:
    Eigen::Vector3f get_kd() const;
    Eigen::Vector3f get_ks() const;
    Eigen::Vector3f get_ka() const;
    Eigen::Vector3f get_ke() const;
    float get_ns() const;
    bool has_emission() const;
    Illum get_mode() const;
};
class Triangle {
public:
    Eigen::Vector3f get_v1() const;
    Eigen::Vector3f get_v2() const;
    Eigen::Vector3f get_v3() const;
    Eigen::Vector3f get_normal() const;
    float get_area() const;
    float get_area_of_obj() const;
    Material get_material() const;
};
class Object {
public:
    std::vector<Triangle>& get_triangles();
};
struct BVHNode {
    int lc, rc;
    unsigned int n;
    int it;
    Eigen::Vector3f AA, BB;
};
class BVH {
public:
    int get_root_index() const;
    const std::vector<BVHNode>& get_nodes();
    const std::vector<Triangle>& get_triangles();
};
class Scene {
public:
    unsigned int get_height() const;
    unsigned int get_width() const;
    unsigned int get_pixels() const;
    BVH& get_bvh();
    std::vector<Object>& get_light_objs();
};
#endif
