// host_internal.h — shared declarations of librtu_host.so's translation units.
#ifndef RTU_HOST_INTERNAL_H
#define RTU_HOST_INTERNAL_H

#include "rtu_host.h"

#include <string>
#include <vector>

namespace rtu {

void set_error(const std::string& msg);

struct MeshData {
    std::vector<float>      v, vn, vt;
    std::vector<uint32_t>   f, fn, ft;
    std::vector<RtuBvhNode> bvh;
    std::vector<uint32_t>   elements;
    uint32_t bvh_depth = 0;
    float    bound_min[3] = {1, 1, 1};   // cyTriMesh.h:128 "not ready" box
    float    bound_max[3] = {0, 0, 0};
};

struct TextureData {
    RtuTexture hdr{};               // rgb pointer is fixed up by Scene::rebuild_desc
    std::vector<uint8_t> rgb;
};

// An owned flattened scene; `desc` always points into the vectors below.
struct Scene {
    std::vector<TextureData> textures;
    std::vector<RtuTexture>  texture_descs;
    std::vector<RtuTexMap>   material_maps;   // empty or 4 per material
    RtuTexMap background_map{}, environment_map{};
    std::vector<RtuNode>     nodes;
    std::vector<RtuMaterial> materials;
    std::vector<RtuLight>    lights;
    std::vector<MeshData>    meshes;
    std::vector<RtuMesh>     mesh_descs;
    RtuCamera   camera{};
    RtuEnvColor background{};
    RtuEnvColor environment{};
    RtuSceneDesc desc{};

    static Scene* from_desc(const RtuSceneDesc& d);
    void rebuild_desc();
};

}  // namespace rtu

extern "C" RtuScene* rtu_scene_wrap(rtu::Scene* s);

#endif
