// rtu_device.h — HBM layout of an uploaded scene and the kernel launch interface.
//
// Layout choices (see DESIGN.md "Data layout in HBM"):
//  * scene-graph nodes, materials, lights: tiny, wave-uniform -> read through the
//    scalar cache (s_load), one struct per item, as on the host (rtu_scene.h);
//  * BVH nodes: 32 B, sibling pairs 64-B aligned -> an inner visit is one
//    64-byte line (4 x dwordx4);
//  * triangles: gathered ONCE at upload into leaf order, 48 B per triangle
//    {A.xyz,N.x | B.xyz,N.y | C.xyz,N.z} so a triangle test is three aligned
//    16-byte loads and no index chasing; N is the normalised face normal the
//    reference recomputes per test (objFunctions.cpp:263), evaluated here once
//    with the same operations;
//  * per-vertex normals / texcoords stay indexed (touched only on an accepted hit).
#ifndef RTU_DEVICE_H_INCLUDED
#define RTU_DEVICE_H_INCLUDED

#include "rtu_render.h"
#include "rtu_vec.h"

struct DevMesh {
    const float4*   bvh;        // 2 float4 per RtuBvhNode: {bmin.xyz, index} {bmax.xyz, count}
    const float4*   tri;        // 3 float4 per element slot (leaf order)
    const uint32_t* elements;   // element slot -> face id
    const uint32_t* f;          // face -> 3 vertex ids
    const float*    v;
    const uint32_t* fn;
    const float*    vn;
    float    bmin[3], bmax[3];
    uint32_t n_bvh_nodes, n_elements;
};

struct DevNode {                // one scene-graph node (wave-uniform data)
    float   tm[9], itm[9], pos[3];
    int32_t parent, obj_type, mesh_id, material_id, depth;
    int32_t chain[RTU_MAX_NODE_DEPTH];  // chain[d] = ancestor at depth d (chain[depth] == self)
};

struct DevScene {
    const DevNode*     nodes;
    const RtuMaterial* materials;
    const RtuLight*    lights;
    const DevMesh*     meshes;
    uint32_t n_nodes, n_lights;
    float    background[3];     // background.Sample(...) for an untextured / NULL-map background
    float    environment[3];    // environment.SampleEnvironment(...) likewise
};

// Shade() recursion frames live in an HBM arena, one column per thread:
// arena[(level * RTU_FRAME_FIELDS + field) * n_threads + thread]
#define RTU_FRAME_FIELDS 17

struct KernelArgs {
    DevScene   scene;
    RtuFrameDesc frame;
    float4*    out;             // shard rows * width
    float*     arena;
    unsigned long long* counters;  // 11 x u64 (RtuStats order) or nullptr
    uint32_t   tiles_x;         // ceil(width / 8)
    uint32_t   n_threads;       // gridDim.x * 64
};

// Launches the variant matching (stack_depth, stats). Returns hipError_t as int.
int rtu_launch_render(const KernelArgs& args, uint32_t n_blocks, uint32_t bvh_stack_needed, bool stats,
                      hipStream_t stream);

#endif
