// scene_blob.cpp — owned scenes (RtuScene) and their blob serialisation.
//
// A blob is the on-disk form of an RtuSceneDesc: the format of the golden
// fixtures under tests/golden/ and the way a flattened scene reaches a machine
// that has no scene files. Layout (little endian, every section 16-byte
// aligned):
//   BlobHeader | nodes | materials | lights | per mesh: MeshHeader v f vn fn vt ft bvh elements
//   [ | per texture: TexHeader rgb | material maps | background map | environment map ]   (textured scenes only)
#include "host_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace rtu {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }

namespace {

const char kMagic[8] = {'R', 'T', 'U', 'S', 'C', 'N', '0', '1'};

struct BlobHeader {
    char        magic[8];
    uint32_t    n_nodes, n_materials, n_lights, n_meshes;
    RtuCamera   camera;
    RtuEnvColor background, environment;
    uint32_t    n_textures;      // 0 in the blobs of untextured scenes (the field was padding): no texture section
    uint32_t    has_maps;        // 1: material maps (4 per material) + background + environment maps follow the textures
    uint32_t    pad[2];
};

struct TexHeader {
    int32_t type, width, height, reserved;
    float   color1[3], color2[3];
    uint32_t pad[2];
};

struct MeshHeader {
    uint32_t nv, nf, nvn, nvt, n_bvh_nodes, n_elements, bvh_depth, reserved;
    float    bound_min[3], bound_max[3];
    uint32_t pad[2];
};

static_assert(sizeof(RtuNode) == 128, "RtuNode layout");
static_assert(sizeof(RtuMaterial) == 96, "RtuMaterial layout");
static_assert(sizeof(RtuLight) == 32, "RtuLight layout");
static_assert(sizeof(RtuBvhNode) == 32, "RtuBvhNode layout");
static_assert(sizeof(BlobHeader) % 16 == 0, "BlobHeader alignment");
static_assert(sizeof(MeshHeader) % 16 == 0, "MeshHeader alignment");
static_assert(sizeof(TexHeader) % 16 == 0, "TexHeader alignment");
static_assert(sizeof(RtuTexMap) == 96, "RtuTexMap layout");

struct Writer {
    std::vector<unsigned char> buf;
    void put(const void* p, size_t n) {
        const unsigned char* c = static_cast<const unsigned char*>(p);
        buf.insert(buf.end(), c, c + n);
        while (buf.size() % 16) buf.push_back(0);
    }
};

struct Reader {
    const unsigned char* p;
    size_t size, off;
    bool take(void* dst, size_t n) {
        if (off + n > size) return false;
        if (n) memcpy(dst, p + off, n);
        off += n;
        off = (off + 15) & ~size_t(15);
        return true;
    }
};

}  // namespace

Scene* Scene::from_desc(const RtuSceneDesc& d) {
    Scene* s = new Scene;
    s->nodes.assign(d.nodes, d.nodes + d.n_nodes);
    s->materials.assign(d.materials, d.materials + d.n_materials);
    s->lights.assign(d.lights, d.lights + d.n_lights);
    s->camera = d.camera;
    s->background = d.background;
    s->environment = d.environment;
    s->textures.resize(d.n_textures);
    for (uint32_t i = 0; i < d.n_textures; i++) {
        s->textures[i].hdr = d.textures[i];
        const size_t n = d.textures[i].type == RTU_TEX_FILE ? size_t(d.textures[i].width) * d.textures[i].height * 3 : 0;
        if (n && d.textures[i].rgb) s->textures[i].rgb.assign(d.textures[i].rgb, d.textures[i].rgb + n);
    }
    if (d.material_maps) s->material_maps.assign(d.material_maps, d.material_maps + size_t(d.n_materials) * 4);
    s->background_map = d.background_map;
    s->environment_map = d.environment_map;
    s->meshes.resize(d.n_meshes);
    for (uint32_t i = 0; i < d.n_meshes; i++) {
        const RtuMesh& m = d.meshes[i];
        MeshData& o = s->meshes[i];
        o.v.assign(m.v, m.v + size_t(m.nv) * 3);
        o.f.assign(m.f, m.f + size_t(m.nf) * 3);
        o.vn.assign(m.vn, m.vn + size_t(m.nvn) * 3);
        if (m.fn && m.nvn) o.fn.assign(m.fn, m.fn + size_t(m.nf) * 3);
        if (m.vt && m.nvt) o.vt.assign(m.vt, m.vt + size_t(m.nvt) * 3);
        if (m.ft && m.nvt) o.ft.assign(m.ft, m.ft + size_t(m.nf) * 3);
        o.bvh.assign(m.bvh, m.bvh + m.n_bvh_nodes);
        o.elements.assign(m.elements, m.elements + m.n_elements);
        o.bvh_depth = m.bvh_depth;
        memcpy(o.bound_min, m.bound_min, sizeof o.bound_min);
        memcpy(o.bound_max, m.bound_max, sizeof o.bound_max);
    }
    s->rebuild_desc();
    return s;
}

void Scene::rebuild_desc() {
    mesh_descs.resize(meshes.size());
    for (size_t i = 0; i < meshes.size(); i++) {
        MeshData& o = meshes[i];
        RtuMesh& m = mesh_descs[i];
        memset(&m, 0, sizeof m);
        m.nv = uint32_t(o.v.size() / 3);
        m.nf = uint32_t(o.f.size() / 3);
        m.nvn = uint32_t(o.vn.size() / 3);
        m.nvt = uint32_t(o.vt.size() / 3);
        m.n_bvh_nodes = uint32_t(o.bvh.size());
        m.n_elements = uint32_t(o.elements.size());
        m.bvh_depth = o.bvh_depth;
        memcpy(m.bound_min, o.bound_min, sizeof m.bound_min);
        memcpy(m.bound_max, o.bound_max, sizeof m.bound_max);
        m.v = o.v.data();
        m.f = o.f.data();
        m.vn = o.vn.data();
        m.fn = o.fn.empty() ? nullptr : o.fn.data();
        m.vt = o.vt.empty() ? nullptr : o.vt.data();
        m.ft = o.ft.empty() ? nullptr : o.ft.data();
        m.bvh = o.bvh.data();
        m.elements = o.elements.data();
    }
    memset(&desc, 0, sizeof desc);
    desc.n_nodes = uint32_t(nodes.size());
    desc.n_materials = uint32_t(materials.size());
    desc.n_lights = uint32_t(lights.size());
    desc.n_meshes = uint32_t(meshes.size());
    desc.nodes = nodes.data();
    desc.materials = materials.data();
    desc.lights = lights.data();
    desc.meshes = mesh_descs.data();
    desc.camera = camera;
    desc.background = background;
    desc.environment = environment;
    texture_descs.resize(textures.size());
    for (size_t i = 0; i < textures.size(); i++) {
        texture_descs[i] = textures[i].hdr;
        texture_descs[i].rgb = textures[i].rgb.empty() ? nullptr : textures[i].rgb.data();
    }
    desc.n_textures = uint32_t(textures.size());
    desc.textures = texture_descs.empty() ? nullptr : texture_descs.data();
    desc.material_maps = material_maps.size() == materials.size() * 4 && !material_maps.empty() ? material_maps.data() : nullptr;
    desc.background_map = background_map;
    desc.environment_map = environment_map;
}

static void* to_blob(const RtuSceneDesc& d, size_t* size_out) {
    Writer w;
    BlobHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, kMagic, 8);
    h.n_nodes = d.n_nodes;
    h.n_materials = d.n_materials;
    h.n_lights = d.n_lights;
    h.n_meshes = d.n_meshes;
    h.camera = d.camera;
    h.background = d.background;
    h.environment = d.environment;
    const bool textured = d.n_textures > 0;  // maps without a texture are folded away by the flatteners
    h.n_textures = d.n_textures;
    h.has_maps = textured ? 1u : 0u;
    w.put(&h, sizeof h);
    w.put(d.nodes, sizeof(RtuNode) * d.n_nodes);
    w.put(d.materials, sizeof(RtuMaterial) * d.n_materials);
    w.put(d.lights, sizeof(RtuLight) * d.n_lights);
    for (uint32_t i = 0; i < d.n_meshes; i++) {
        const RtuMesh& m = d.meshes[i];
        MeshHeader mh;
        memset(&mh, 0, sizeof mh);
        mh.nv = m.nv; mh.nf = m.nf; mh.nvn = m.nvn; mh.nvt = (m.vt && m.ft) ? m.nvt : 0;
        mh.n_bvh_nodes = m.n_bvh_nodes; mh.n_elements = m.n_elements; mh.bvh_depth = m.bvh_depth;
        memcpy(mh.bound_min, m.bound_min, sizeof mh.bound_min);
        memcpy(mh.bound_max, m.bound_max, sizeof mh.bound_max);
        w.put(&mh, sizeof mh);
        w.put(m.v, sizeof(float) * 3 * m.nv);
        w.put(m.f, sizeof(uint32_t) * 3 * m.nf);
        w.put(m.vn, sizeof(float) * 3 * m.nvn);
        w.put(m.fn, (m.fn && m.nvn) ? sizeof(uint32_t) * 3 * m.nf : 0);
        w.put(m.vt, mh.nvt ? sizeof(float) * 3 * m.nvt : 0);
        w.put(m.ft, mh.nvt ? sizeof(uint32_t) * 3 * m.nf : 0);
        w.put(m.bvh, sizeof(RtuBvhNode) * m.n_bvh_nodes);
        w.put(m.elements, sizeof(uint32_t) * m.n_elements);
    }
    if (textured) {
        for (uint32_t i = 0; i < d.n_textures; i++) {
            const RtuTexture& t = d.textures[i];
            TexHeader th;
            memset(&th, 0, sizeof th);
            th.type = t.type; th.width = t.width; th.height = t.height;
            memcpy(th.color1, t.color1, sizeof th.color1);
            memcpy(th.color2, t.color2, sizeof th.color2);
            w.put(&th, sizeof th);
            w.put(t.rgb, (t.type == RTU_TEX_FILE && t.rgb) ? size_t(t.width) * t.height * 3 : 0);
        }
        std::vector<RtuTexMap> none(size_t(d.n_materials) * 4);
        memset(none.data(), 0, none.size() * sizeof(RtuTexMap));
        w.put(d.material_maps ? d.material_maps : none.data(), sizeof(RtuTexMap) * d.n_materials * 4);
        w.put(&d.background_map, sizeof(RtuTexMap));
        w.put(&d.environment_map, sizeof(RtuTexMap));
    }
    void* out = malloc(w.buf.size());
    if (!out) return nullptr;
    memcpy(out, w.buf.data(), w.buf.size());
    *size_out = w.buf.size();
    return out;
}

static Scene* from_blob(const void* blob, size_t size) {
    Reader r{static_cast<const unsigned char*>(blob), size, 0};
    BlobHeader h;
    if (!r.take(&h, sizeof h) || memcmp(h.magic, kMagic, 8) != 0) {
        set_error("scene blob: bad magic or truncated header");
        return nullptr;
    }
    // A corrupt count must not drive a huge allocation: every element needs bytes in the blob.
    if (size_t(h.n_nodes) * sizeof(RtuNode) > size || size_t(h.n_materials) * sizeof(RtuMaterial) > size ||
        size_t(h.n_lights) * sizeof(RtuLight) > size || size_t(h.n_meshes) * sizeof(MeshHeader) > size) {
        set_error("scene blob: counts exceed blob size");
        return nullptr;
    }
    Scene* s = new Scene;
    s->camera = h.camera;
    s->background = h.background;
    s->environment = h.environment;
    s->nodes.resize(h.n_nodes);
    s->materials.resize(h.n_materials);
    s->lights.resize(h.n_lights);
    bool ok = r.take(s->nodes.data(), sizeof(RtuNode) * h.n_nodes) &&
              r.take(s->materials.data(), sizeof(RtuMaterial) * h.n_materials) &&
              r.take(s->lights.data(), sizeof(RtuLight) * h.n_lights);
    s->meshes.resize(h.n_meshes);
    for (uint32_t i = 0; ok && i < h.n_meshes; i++) {
        MeshHeader mh;
        ok = r.take(&mh, sizeof mh);
        if (!ok) break;
        size_t need = (size_t(mh.nv) + mh.nvn + mh.nvt) * 12 + size_t(mh.nf) * 12 +
                      size_t(mh.n_bvh_nodes) * sizeof(RtuBvhNode) + size_t(mh.n_elements) * 4;
        if (need > size) { ok = false; break; }
        MeshData& o = s->meshes[i];
        o.v.resize(size_t(mh.nv) * 3);
        o.f.resize(size_t(mh.nf) * 3);
        o.vn.resize(size_t(mh.nvn) * 3);
        o.fn.resize(mh.nvn ? size_t(mh.nf) * 3 : 0);
        o.vt.resize(size_t(mh.nvt) * 3);
        o.ft.resize(mh.nvt ? size_t(mh.nf) * 3 : 0);
        o.bvh.resize(mh.n_bvh_nodes);
        o.elements.resize(mh.n_elements);
        o.bvh_depth = mh.bvh_depth;
        memcpy(o.bound_min, mh.bound_min, sizeof o.bound_min);
        memcpy(o.bound_max, mh.bound_max, sizeof o.bound_max);
        ok = r.take(o.v.data(), o.v.size() * 4) && r.take(o.f.data(), o.f.size() * 4) &&
             r.take(o.vn.data(), o.vn.size() * 4) && r.take(o.fn.data(), o.fn.size() * 4) &&
             r.take(o.vt.data(), o.vt.size() * 4) && r.take(o.ft.data(), o.ft.size() * 4) &&
             r.take(o.bvh.data(), o.bvh.size() * sizeof(RtuBvhNode)) &&
             r.take(o.elements.data(), o.elements.size() * 4);
    }
    if (ok && h.has_maps) {
        if (size_t(h.n_textures) * sizeof(TexHeader) > size) ok = false;
        s->textures.resize(ok ? h.n_textures : 0);
        for (uint32_t i = 0; ok && i < h.n_textures; i++) {
            TexHeader th;
            ok = r.take(&th, sizeof th);
            if (!ok) break;
            TextureData& t = s->textures[i];
            t.hdr.type = th.type; t.hdr.width = th.width; t.hdr.height = th.height;
            memcpy(t.hdr.color1, th.color1, sizeof th.color1);
            memcpy(t.hdr.color2, th.color2, sizeof th.color2);
            size_t n = 0;
            if (th.type == RTU_TEX_FILE) {
                if (th.width < 0 || th.height < 0 || size_t(th.width) * size_t(th.height) * 3 > size) { ok = false; break; }
                n = size_t(th.width) * th.height * 3;
            }
            t.rgb.resize(n);
            ok = r.take(t.rgb.data(), n);
        }
        s->material_maps.resize(size_t(h.n_materials) * 4);
        ok = ok && r.take(s->material_maps.data(), sizeof(RtuTexMap) * s->material_maps.size()) &&
             r.take(&s->background_map, sizeof(RtuTexMap)) && r.take(&s->environment_map, sizeof(RtuTexMap));
        for (size_t i = 0; ok && i < s->material_maps.size(); i++)
            if (s->material_maps[i].present && s->material_maps[i].texture >= (int32_t)h.n_textures) ok = false;
    }
    if (!ok) {
        set_error("scene blob: truncated");
        delete s;
        return nullptr;
    }
    s->rebuild_desc();
    return s;
}

}  // namespace rtu

struct RtuScene {
    rtu::Scene* impl;
};

extern "C" {

const char* rtu_host_last_error(void) { return rtu::g_last_error.c_str(); }

RtuScene* rtu_scene_wrap(rtu::Scene* s) {
    if (!s) return nullptr;
    RtuScene* h = new RtuScene;
    h->impl = s;
    return h;
}

RtuScene* rtu_scene_clone(const RtuSceneDesc* desc) {
    if (!desc) return nullptr;
    return rtu_scene_wrap(rtu::Scene::from_desc(*desc));
}

RtuScene* rtu_scene_load_blob(const void* blob, size_t size) {
    if (!blob) return nullptr;
    return rtu_scene_wrap(rtu::from_blob(blob, size));
}

RtuScene* rtu_scene_load_blob_file(const char* path) {
    FILE* fp = fopen(path, "rb");
    if (!fp) {
        rtu::set_error(std::string("cannot open ") + path);
        return nullptr;
    }
    std::vector<unsigned char> buf;
    unsigned char tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, fp)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(fp);
    return rtu_scene_load_blob(buf.data(), buf.size());
}

void* rtu_scene_to_blob(const RtuSceneDesc* desc, size_t* size_out) {
    if (!desc || !size_out) return nullptr;
    return rtu::to_blob(*desc, size_out);
}

int rtu_scene_save_blob_file(const RtuSceneDesc* desc, const char* path) {
    size_t n = 0;
    void* b = rtu_scene_to_blob(desc, &n);
    if (!b) return -1;
    FILE* fp = fopen(path, "wb");
    if (!fp) {
        free(b);
        rtu::set_error(std::string("cannot create ") + path);
        return -2;
    }
    size_t w = fwrite(b, 1, n, fp);
    fclose(fp);
    free(b);
    return w == n ? 0 : -3;
}

void rtu_blob_free(void* blob) { free(blob); }

const RtuSceneDesc* rtu_scene_desc(const RtuScene* scene) { return scene ? &scene->impl->desc : nullptr; }

void rtu_scene_set_resolution(RtuScene* scene, int width, int height) {
    if (!scene) return;
    scene->impl->camera.img_width = width;
    scene->impl->camera.img_height = height;
    scene->impl->desc.camera = scene->impl->camera;
}

void rtu_scene_free(RtuScene* scene) {
    if (!scene) return;
    delete scene->impl;
    delete scene;
}

}  // extern "C"
