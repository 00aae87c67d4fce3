// cudaraytracing_amd/csrc/crt_bvh_build.h -- internal interface between the host layer (crt_host.cpp: BVH) and the device
// builder (crt_bvh_build.hip).  The C ABI entry is crt_host_scene_set_bvh_device (include/crt.h).
#ifndef CRT_BVH_BUILD_H
#define CRT_BVH_BUILD_H

#include "../../include/crt.h"

#include <vector>

// a range [l, r) of the triangle order whose subtree the host builder has to finish (equal sort keys inside it, see
// crt_bvh_build.hip); its nodes occupy [first_node, first_node + count(r - l)) of the post-order array; depth = level of its root (root = 1)
struct crt_bvh_host_range {
    uint32_t l, r, first_node, depth;
};

// tmin / tmax / centroid: 3 floats per triangle (Triangle::get_min / get_max / get_center).  out_perm[pos] = index of the
// triangle that stands at position pos of the BVH order; out_nodes = the post-order node array (nodes of host ranges are left
// zeroed).  Byte-identical to BVH::build_node wherever no host range is reported.
int crt_bvh_build_device(uint32_t n, const float* tmin, const float* tmax, const float* centroid, uint32_t thresh, int device, uint32_t* out_perm,
                         crt_bvh_node* out_nodes, uint32_t n_nodes_cap, std::vector<crt_bvh_host_range>* host_ranges, crt_bvh_build_info* info);

#endif
