// driver.cpp — build-owned driver around the UNMODIFIED reference sources.
//
// TEST INFRASTRUCTURE ONLY. This file is compiled in the authoring container
// (where /root/reference exists) into oracle/_ref/ref_render by
// oracle/ref_harness/Makefile. Nothing here ships in the product path and the
// reference sources are never copied: they are #included / compiled from where
// they lie. See SURVEY.md Appendix A for the recipe this follows.
//
// What it does, for one scene:
//   1. LoadScene() (ExternalLibrary/xmlload.cpp:64) on a path-remapped copy of the XML;
//   2. flattens the reference's in-memory scene graph into an RtuSceneDesc and
//      writes it as a blob (the golden INPUT for every other implementation);
//   3. renders "recipe W" (SURVEY.md §8c): one ray through each pixel centre,
//      Trace() (RenderFunctions.cpp:181) + Material::Shade(...,5) exactly as
//      Render() does per sample (RenderFunctions.cpp:96-103,134), z = hInfo.z;
//   4. writes z.f32, rgb.f32 (linear, pre-gamma), Result.png / ZBuffer.png via
//      the reference's own RenderImage (scene.h:590-654) and stats.json.
//
// Include order mirrors main.cpp:1-10 minus viewport.cpp (GLUT is absent here).
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "rtu_host.h"

// scene.h:47-53 defines function-like min/max macros that break std::min/std::max
// in objFunctions.cpp under libstdc++; make them object-like no-ops instead.
using std::max;
using std::min;
#define min min
#define max max

// The reference's material / light parameters have setters but no getters; the
// flattener has to read them.
#define private public
#define protected public
#include "ExternalLibrary/scene.h"
#include "ExternalLibrary/objects.h"
#include "ExternalLibrary/materials.h"
#include "ExternalLibrary/lights.h"
#include "ExternalLibrary/texture.h"
#undef private
#undef protected

extern RenderImage renderImage;     // normally declared by viewport.cpp:41
extern TexturedColor environment;   // normally declared by xmlload.cpp:28 before RenderFunctions.cpp
extern Camera camera;

#include "ExternalLibrary/xmlload.cpp"
#include "ExternalLibrary/lodepng.cpp"
#include "ExternalLibrary/cyPhotonMap.h"
#include "RenderFunctions.cpp"

// main.cpp:17-27
RenderImage renderImage;
Camera camera;
Sphere theSphere;
Plane thePlane;
Node rootNode;
MaterialList materials;
LightList lights;
ObjFileList objList;
TexturedColor background;
TexturedColor environment;
TextureList textureList;

// ---- ray counters: only cross-TU calls are wrapped = root-level secondary /
// shadow rays issued by mtlFunctions.o / lightFunctions.o.
static std::atomic<long long> g_secondary{0}, g_shadow{0};
extern "C" {
bool __real__Z5TraceRK3RayP4NodeR7HitInfo(const Ray&, Node*, HitInfo&);
bool __real__Z11ShadowTraceRK3RayP4NodeR7HitInfo(const Ray&, Node*, HitInfo&);
bool __wrap__Z5TraceRK3RayP4NodeR7HitInfo(const Ray& r, Node* n, HitInfo& h) {
    g_secondary.fetch_add(1, std::memory_order_relaxed);
    return Trace(r, n, h);
}
bool __wrap__Z11ShadowTraceRK3RayP4NodeR7HitInfo(const Ray& r, Node* n, HitInfo& h) {
    g_shadow.fetch_add(1, std::memory_order_relaxed);
    return ShadowTrace(r, n, h);
}
}

// ---- flatten the reference scene graph ------------------------------------
struct Flat {
    std::vector<RtuNode> nodes;
    std::vector<RtuMaterial> mats;
    std::vector<RtuLight> lts;
    std::vector<const TriObj*> meshObjs;
    struct MeshStore {
        std::vector<float> v, vn, vt;
        std::vector<uint32_t> f, fn, ft, elements;
        std::vector<RtuBvhNode> bvh;
        uint32_t depth;
    };
    std::vector<MeshStore> meshStore;
    std::vector<RtuMesh> meshes;
    std::vector<std::vector<uint8_t>> texData;
    std::vector<RtuTexture> textures;
    std::vector<RtuTexMap> maps;
};

static int MaterialIndex(const Material* m) {
    if (!m) return -1;
    for (size_t i = 0; i < materials.size(); i++)
        if (materials[i] == m) return (int)i;
    return -1;
}

static void ConvertBVH(const cyBVHTriMesh& bvh, unsigned int id, uint32_t level, Flat::MeshStore& out) {
    if (out.bvh.size() <= id) {
        RtuBvhNode z;
        memset(&z, 0, sizeof z);
        out.bvh.resize(id + 1, z);
    }
    const float* b = bvh.GetNodeBounds(id);
    RtuBvhNode n;
    n.bmin[0] = b[0]; n.bmin[1] = b[1]; n.bmin[2] = b[2];
    n.bmax[0] = b[3]; n.bmax[1] = b[4]; n.bmax[2] = b[5];
    if (level > out.depth) out.depth = level;
    if (bvh.IsLeafNode(id)) {
        n.count = bvh.GetNodeElementCount(id);
        // element offset: pointer difference to the start of the element list
        n.index = (uint32_t)(bvh.GetNodeElements(id) - bvh.elements);
        out.bvh[id] = n;
    } else {
        n.count = 0;
        n.index = bvh.GetFirstChildNode(id);
        out.bvh[id] = n;
        ConvertBVH(bvh, bvh.GetFirstChildNode(id), level + 1, out);
        ConvertBVH(bvh, bvh.GetSecondChildNode(id), level + 1, out);
    }
}

static int MeshIndex(Flat& F, const TriObj* t) {
    for (size_t i = 0; i < F.meshObjs.size(); i++)
        if (F.meshObjs[i] == t) return (int)i;
    F.meshObjs.push_back(t);
    F.meshStore.emplace_back();
    Flat::MeshStore& s = F.meshStore.back();
    s.depth = 0;
    for (unsigned i = 0; i < t->NV(); i++) { s.v.push_back(t->V(i).x); s.v.push_back(t->V(i).y); s.v.push_back(t->V(i).z); }
    for (unsigned i = 0; i < t->NVN(); i++) { s.vn.push_back(t->VN(i).x); s.vn.push_back(t->VN(i).y); s.vn.push_back(t->VN(i).z); }
    for (unsigned i = 0; i < t->NVT(); i++) { s.vt.push_back(t->VT(i).x); s.vt.push_back(t->VT(i).y); s.vt.push_back(t->VT(i).z); }
    for (unsigned i = 0; i < t->NF(); i++) {
        for (int k = 0; k < 3; k++) s.f.push_back(t->F(i).v[k]);
        if (t->NVN()) for (int k = 0; k < 3; k++) s.fn.push_back(t->FN(i).v[k]);
        if (t->NVT()) for (int k = 0; k < 3; k++) s.ft.push_back(t->FT(i).v[k]);
    }
    const cyBVHTriMesh& bvh = t->bvh;  // the tree TriObj::IntersectRay walks (objects.h:63)
    for (unsigned i = 0; i < t->NF(); i++) s.elements.push_back(bvh.elements[i]);
    ConvertBVH(bvh, bvh.GetRootNodeID(), 1, s);
    return (int)F.meshObjs.size() - 1;
}

static void FlattenNode(Flat& F, const Node* n, int parent, int depth) {
    int me = (int)F.nodes.size();
    RtuNode o;
    memset(&o, 0, sizeof o);
    for (int i = 0; i < 9; i++) { o.tm[i] = n->GetTransform().data[i]; o.itm[i] = n->GetInverseTransform().data[i]; }
    o.pos[0] = n->GetPosition().x; o.pos[1] = n->GetPosition().y; o.pos[2] = n->GetPosition().z;
    o.parent = parent;
    o.depth = depth;
    o.mesh_id = -1;
    o.material_id = MaterialIndex(n->GetMaterial());
    const Object* obj = n->GetNodeObj();
    if (!obj) o.obj_type = RTU_OBJ_NONE;
    else if (obj == &theSphere) o.obj_type = RTU_OBJ_SPHERE;
    else if (obj == &thePlane) o.obj_type = RTU_OBJ_PLANE;
    else {
        const TriObj* t = dynamic_cast<const TriObj*>(obj);
        if (!t) { fprintf(stderr, "unknown object type\n"); exit(2); }
        o.obj_type = RTU_OBJ_TRIMESH;
        o.mesh_id = MeshIndex(F, t);
    }
    F.nodes.push_back(o);
    for (int i = 0; i < n->GetNumChild(); i++) FlattenNode(F, n->GetChild(i), me, depth + 1);
    F.nodes[me].subtree_end = (int)F.nodes.size();
}

static RtuEnvColor FlattenEnv(const TexturedColor& t) {
    RtuEnvColor e;
    memset(&e, 0, sizeof e);
    e.color[0] = t.GetColor().r; e.color[1] = t.GetColor().g; e.color[2] = t.GetColor().b;
    e.has_map = t.GetTexture() ? 1 : 0;
    e.map_is_null = (t.GetTexture() && t.GetTexture()->texture == NULL) ? 1 : 0;
    return e;
}

// textureList (xmlload.cpp:29) -> RtuTexture[], in list order
static void FlattenTextures(Flat& F) {
    F.texData.resize(textureList.list.size());
    for (size_t i = 0; i < textureList.list.size(); i++) {
        const Texture* t = textureList.list[i]->GetObj();
        RtuTexture o;
        memset(&o, 0, sizeof o);
        if (const TextureChecker* c = dynamic_cast<const TextureChecker*>(t)) {
            o.type = RTU_TEX_CHECKER;
            o.color1[0] = c->color1.r; o.color1[1] = c->color1.g; o.color1[2] = c->color1.b;
            o.color2[0] = c->color2.r; o.color2[1] = c->color2.g; o.color2[2] = c->color2.b;
        } else if (const TextureFile* f = dynamic_cast<const TextureFile*>(t)) {
            o.type = RTU_TEX_FILE;
            o.width = f->width; o.height = f->height;
            F.texData[i].resize((size_t)f->width * f->height * 3);
            for (size_t k = 0; k < f->data.size(); k++) {
                F.texData[i][3 * k] = f->data[k].r; F.texData[i][3 * k + 1] = f->data[k].g; F.texData[i][3 * k + 2] = f->data[k].b;
            }
        } else { fprintf(stderr, "unknown texture class\n"); exit(2); }
        F.textures.push_back(o);
    }
    for (size_t i = 0; i < F.textures.size(); i++) F.textures[i].rgb = F.texData[i].empty() ? NULL : F.texData[i].data();
}
static RtuTexMap FlattenMap(const TexturedColor& t) {
    RtuTexMap m;
    memset(&m, 0, sizeof m);
    m.texture = -1;
    const TextureMap* tm = t.GetTexture();
    if (!tm) return m;
    m.present = 1;
    for (size_t i = 0; i < textureList.list.size(); i++)
        if (textureList.list[i]->GetObj() == tm->texture) m.texture = (int32_t)i;
    for (int k = 0; k < 9; k++) { m.tm[k] = tm->GetTransform().data[k]; m.itm[k] = tm->GetInverseTransform().data[k]; }
    m.pos[0] = tm->GetPosition().x; m.pos[1] = tm->GetPosition().y; m.pos[2] = tm->GetPosition().z;
    return m;
}

static void Flatten(Flat& F, RtuSceneDesc& d) {
    for (size_t i = 0; i < materials.size(); i++) {
        RtuMaterial m;
        memset(&m, 0, sizeof m);
        const MtlBlinn* b = dynamic_cast<const MtlBlinn*>(materials[i]);
        // MultiMtl::Shade forwards to mtls[hInfo.mtlID] and mtlID is never written (scene.h:159,162): sub-material 0
        if (const MultiMtl* mm = dynamic_cast<const MultiMtl*>(materials[i]))
            b = mm->mtls.empty() ? NULL : dynamic_cast<const MtlBlinn*>(mm->mtls[0]);
        if (!b) { fprintf(stderr, "non-Blinn material: outside the flattened format\n"); exit(2); }
        auto put = [](float* dst, const Color& c) { dst[0] = c.r; dst[1] = c.g; dst[2] = c.b; };
        put(m.diffuse, b->diffuse.GetColor());
        put(m.specular, b->specular.GetColor());
        put(m.reflection, b->reflection.GetColor());
        put(m.refraction, b->refraction.GetColor());
        put(m.emission, b->emission.GetColor());
        put(m.absorption, b->absorption);
        m.glossiness = b->glossiness;
        m.ior = b->ior;
        m.reflection_glossiness = b->reflectionGlossiness;
        m.refraction_glossiness = b->refractionGlossiness;
        // like the host loader: a map whose texture failed to load multiplies the colour by black (scene.h:382,421)
        const TexturedColor* tcs[5] = {&b->diffuse, &b->specular, &b->reflection, &b->refraction, &b->emission};
        float* dst[5] = {m.diffuse, m.specular, m.reflection, m.refraction, m.emission};
        for (int k = 0; k < 5; k++)
            if (tcs[k]->GetTexture() && !tcs[k]->GetTexture()->texture)
                for (int j = 0; j < 3; j++) dst[k][j] = dst[k][j] * 0.0f;
        F.mats.push_back(m);
        for (int k = 0; k < 4; k++) {
            RtuTexMap tm = FlattenMap(*tcs[k]);
            if (tm.present && tm.texture < 0) { memset(&tm, 0, sizeof tm); tm.texture = -1; }
            F.maps.push_back(tm);
        }
    }
    FlattenTextures(F);
    for (size_t i = 0; i < lights.size(); i++) {
        RtuLight l;
        memset(&l, 0, sizeof l);
        if (const AmbientLight* a = dynamic_cast<const AmbientLight*>(lights[i])) {
            l.type = RTU_LIGHT_AMBIENT;
            l.intensity[0] = a->intensity.r; l.intensity[1] = a->intensity.g; l.intensity[2] = a->intensity.b;
        } else if (const DirectLight* dl = dynamic_cast<const DirectLight*>(lights[i])) {
            l.type = RTU_LIGHT_DIRECT;
            l.intensity[0] = dl->intensity.r; l.intensity[1] = dl->intensity.g; l.intensity[2] = dl->intensity.b;
            l.vec[0] = dl->direction.x; l.vec[1] = dl->direction.y; l.vec[2] = dl->direction.z;
        } else if (const PointLight* p = dynamic_cast<const PointLight*>(lights[i])) {
            l.type = RTU_LIGHT_POINT;
            l.intensity[0] = p->intensity.r; l.intensity[1] = p->intensity.g; l.intensity[2] = p->intensity.b;
            l.vec[0] = p->position.x; l.vec[1] = p->position.y; l.vec[2] = p->position.z;
            l.size = p->size;
        } else { fprintf(stderr, "unknown light type\n"); exit(2); }
        F.lts.push_back(l);
    }
    FlattenNode(F, &rootNode, -1, 0);
    for (size_t i = 0; i < F.meshObjs.size(); i++) {
        const TriObj* t = F.meshObjs[i];
        Flat::MeshStore& s = F.meshStore[i];
        RtuMesh m;
        memset(&m, 0, sizeof m);
        m.nv = t->NV(); m.nf = t->NF(); m.nvn = t->NVN(); m.nvt = t->NVT();
        m.n_bvh_nodes = (uint32_t)s.bvh.size();
        m.n_elements = (uint32_t)s.elements.size();
        m.bvh_depth = s.depth;
        m.bound_min[0] = t->GetBoundMin().x; m.bound_min[1] = t->GetBoundMin().y; m.bound_min[2] = t->GetBoundMin().z;
        m.bound_max[0] = t->GetBoundMax().x; m.bound_max[1] = t->GetBoundMax().y; m.bound_max[2] = t->GetBoundMax().z;
        m.v = s.v.data(); m.f = s.f.data(); m.vn = s.vn.data();
        m.fn = s.fn.empty() ? NULL : s.fn.data();
        m.vt = s.vt.empty() ? NULL : s.vt.data();
        m.ft = s.ft.empty() ? NULL : s.ft.data();
        m.bvh = s.bvh.data(); m.elements = s.elements.data();
        F.meshes.push_back(m);
    }
    memset(&d, 0, sizeof d);
    d.n_nodes = (uint32_t)F.nodes.size(); d.nodes = F.nodes.data();
    d.n_materials = (uint32_t)F.mats.size(); d.materials = F.mats.data();
    d.n_lights = (uint32_t)F.lts.size(); d.lights = F.lts.data();
    d.n_meshes = (uint32_t)F.meshes.size(); d.meshes = F.meshes.data();
    d.camera.pos[0] = camera.pos.x; d.camera.pos[1] = camera.pos.y; d.camera.pos[2] = camera.pos.z;
    d.camera.dir[0] = camera.dir.x; d.camera.dir[1] = camera.dir.y; d.camera.dir[2] = camera.dir.z;
    d.camera.up[0] = camera.up.x; d.camera.up[1] = camera.up.y; d.camera.up[2] = camera.up.z;
    d.camera.fov = camera.fov; d.camera.focaldist = camera.focaldist; d.camera.dof = camera.dof;
    d.camera.img_width = camera.imgWidth; d.camera.img_height = camera.imgHeight;
    d.background = FlattenEnv(background);
    d.environment = FlattenEnv(environment);
    bool any = false;
    for (size_t i = 0; i < F.maps.size(); i++) any = any || F.maps[i].present;
    d.n_textures = (uint32_t)F.textures.size();
    d.textures = F.textures.empty() ? NULL : F.textures.data();
    d.material_maps = any ? F.maps.data() : NULL;
    d.background_map = FlattenMap(background);
    d.environment_map = FlattenMap(environment);
    if (d.background_map.texture < 0) { memset(&d.background_map, 0, sizeof(RtuTexMap)); d.background_map.texture = -1; }
    if (d.environment_map.texture < 0) { memset(&d.environment_map, 0, sizeof(RtuTexMap)); d.environment_map.texture = -1; }
}

// ---- recipe W --------------------------------------------------------------
static void RenderRows(int y0, int y1, Point3 org, float* z, float* rgb, std::atomic<long long>* hits) {
    const int W = camera.imgWidth, H = camera.imgHeight;
    long long nh = 0;
    for (int y = y0; y < y1; y++) {
        for (int x = 0; x < W; x++) {
            Point3 cp = CalculateCurrentPoint(x, y, 0.5f, 0.5f, org);
            Ray ray = Ray(camera.pos, (cp - camera.pos).GetNormalized());
            HitInfo h;
            bool hit = Trace(ray, &rootNode, h);
            Color c;
            if (hit) {
                nh++;
                const Material* mtl = h.node->GetMaterial();
                c = mtl ? mtl->Shade(ray, h, lights, 5) : Color(1, 1, 1);
            } else {
                c = background.Sample(Point3((float)x / camera.imgWidth, (float)y / camera.imgHeight, 0));
            }
            int i = x + W * y;
            z[i] = h.z;
            rgb[3 * i + 0] = c.r; rgb[3 * i + 1] = c.g; rgb[3 * i + 2] = c.b;
            // RenderFunctions.cpp:155-160
            Color g = c;
            g.r = pow(g.r, 1 / 2.2);
            g.g = pow(g.g, 1 / 2.2);
            g.b = pow(g.b, 1 / 2.2);
            renderImage.GetPixels()[i] = Color24(g);
            renderImage.GetZBuffer()[i] = h.z;
            renderImage.IncrementNumRenderPixel(1);
        }
    }
    hits->fetch_add(nh);
    (void)H;
}

// ---- recipe S ("next" row f1) -----------------------------------------------
// The sample loop of Render() (RenderFunctions.cpp:73-152) with spp samples instead of the
// hard-coded 1024 and direct lighting only, every sample traced and shaded in turn. Every rand()
// call of the reference (lightFunctions.cpp:47-48, RenderFunctions.cpp:291-293 via
// mtlFunctions.cpp:164,226,276) and of this loop is redirected at link time (-Wl,--wrap=rand) to a
// counter-based stream: the n-th call while sample `index` of pixel `p` is evaluated returns
// rand31(sample_key(p, index), n) — the "sequential" stream of oracle/rtu_oracle.cpp, so that
// restatement can be compared with this build bit for bit.
static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
static thread_local uint32_t t_key = 0, t_counter = 0;
extern "C" int __wrap_rand(void) {
    return (int)(mix32(t_key ^ mix32((t_counter++) * 0x9e3779b9U + 0x85ebca6bU)) >> 1);
}

// gi: recipe P — the Monte-Carlo gather of Render() (RenderFunctions.cpp:129-135) before the direct Shade.
static void RenderRowsSampled(int y0, int y1, int spp, bool gi, float* z, float* rgb, std::atomic<long long>* hits) {
    const int W = camera.imgWidth;
    long long nh = 0;
    float pixelIncrement = 1.0 / spp;
    for (int y = y0; y < y1; y++) {
        for (int x = 0; x < W; x++) {
            Color pixelValuesSum = Color(0.0, 0.0, 0.0);
            float zSum = 0.0;
            int numOfHits = 0;
            for (int index = 0; index < spp; index++) {
                t_key = mix32(mix32((uint32_t)(x + W * y) + 0x68bc21ebU) ^ ((uint32_t)index * 0x9e3779b9U + 1U));
                t_counter = 0;
                Point3 imgOrigin = CalculateImageOrigin(camera.focaldist);
                float currentOffset = index * pixelIncrement;
                float offsetX = Halton(index, 4);
                float offsetY = Halton(index, 5);
                float sampleX = static_cast<float>(rand()) / static_cast<float>(RAND_MAX);
                float sampleTheta = static_cast<float>(rand()) / (static_cast<float>(RAND_MAX / (2 * M_PI)));
                float camOffsetX = sqrt(sampleX * camera.dof * camera.dof) * cos(sampleTheta);
                float camOffsetY = sqrt(sampleX * camera.dof * camera.dof) * sin(sampleTheta);
                Point3 sampledPosition = camera.pos + camera.up * camOffsetY +
                                         camera.dir.GetNormalized().Cross(camera.up.GetNormalized()).GetNormalized() * camOffsetX;
                Point3 currentPoint = CalculateCurrentPoint(x, y, currentOffset + offsetX, currentOffset + offsetY, imgOrigin);
                Ray ray = Ray(sampledPosition, (currentPoint - sampledPosition).GetNormalized());
                HitInfo h;
                Color c;
                if (Trace(ray, &rootNode, h)) {
                    nh++;
                    zSum += h.z;
                    numOfHits++;
                    const Material* mtl = h.node->GetMaterial();
                    if (gi && mtl) {
                        LightList monteCarloList;
                        MonteCarlo(monteCarloList, h, x, y, monteCarloBounces, monteCarloSampleSize);
                        c = mtl->Shade(ray, h, monteCarloList, 5);
                        c += mtl->Shade(ray, h, lights, 5);
                    } else {
                        c = mtl ? mtl->Shade(ray, h, lights, 5) : Color(1, 1, 1);
                    }
                } else {
                    c = background.Sample(Point3((float)x / camera.imgWidth, (float)y / camera.imgHeight, 0));
                }
                pixelValuesSum += c;
            }
            pixelValuesSum /= (float)spp;
            int i = x + W * y;
            z[i] = numOfHits ? zSum / (float)numOfHits : BIGFLOAT;
            rgb[3 * i + 0] = pixelValuesSum.r; rgb[3 * i + 1] = pixelValuesSum.g; rgb[3 * i + 2] = pixelValuesSum.b;
            Color g = pixelValuesSum;
            g.r = pow(g.r, 1 / 2.2);
            g.g = pow(g.g, 1 / 2.2);
            g.b = pow(g.b, 1 / 2.2);
            renderImage.GetPixels()[i] = Color24(g);
            renderImage.GetZBuffer()[i] = z[i];
            renderImage.IncrementNumRenderPixel(1);
        }
    }
    hits->fetch_add(nh);
}

static bool WriteFile(const std::string& path, const void* p, size_t n) {
    FILE* fp = fopen(path.c_str(), "wb");
    if (!fp) return false;
    bool ok = fwrite(p, 1, n, fp) == n;
    fclose(fp);
    return ok;
}

int main(int argc, char** argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s scene.xml width height outdir [threads] [--scene-only | --spp N | --paths N]\n", argv[0]);
        return 1;
    }
    const char* xml = argv[1];
    int W = atoi(argv[2]), H = atoi(argv[3]);
    std::string out = argv[4];
    int threads = argc > 5 ? atoi(argv[5]) : 1;
    bool sceneOnly = argc > 6 && strcmp(argv[6], "--scene-only") == 0;
    int spp = (argc > 7 && (strcmp(argv[6], "--spp") == 0 || strcmp(argv[6], "--paths") == 0)) ? atoi(argv[7]) : 0;  // 0: recipe W
    bool gi = argc > 7 && strcmp(argv[6], "--paths") == 0;  // recipe P
    if (threads < 1) threads = 1;

    if (!LoadScene(xml)) return 3;
    if (W > 0 && H > 0) { camera.imgWidth = W; camera.imgHeight = H; }
    W = camera.imgWidth; H = camera.imgHeight;
    renderImage.Init(W, H);

    Flat F;
    RtuSceneDesc desc;
    Flatten(F, desc);
    if (rtu_scene_save_blob_file(&desc, (out + "/scene.rtus").c_str()) != 0) {
        fprintf(stderr, "cannot write scene blob\n");
        return 4;
    }
    if (sceneOnly) return 0;

    std::vector<float> z((size_t)W * H), rgb((size_t)W * H * 3);
    std::atomic<long long> hits{0};
    srand(1);
    Point3 org = CalculateImageOrigin(camera.focaldist);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) {
        int y0 = (int)((long long)H * t / threads), y1 = (int)((long long)H * (t + 1) / threads);
        if (spp > 0) th.emplace_back(RenderRowsSampled, y0, y1, spp, gi, z.data(), rgb.data(), &hits);
        else th.emplace_back(RenderRows, y0, y1, org, z.data(), rgb.data(), &hits);
    }
    for (auto& t : th) t.join();
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    WriteFile(out + "/z.f32", z.data(), z.size() * 4);
    WriteFile(out + "/rgb.f32", rgb.data(), rgb.size() * 4);
    renderImage.SaveImage((out + "/Result.png").c_str());
    renderImage.ComputeZBufferImage();
    renderImage.SaveZImage((out + "/ZBuffer.png").c_str());
    WriteFile(out + "/result.u8", renderImage.GetPixels(), (size_t)W * H * 3);
    WriteFile(out + "/zbuffer.u8", renderImage.GetZBufferImage(), (size_t)W * H);

    FILE* fp = fopen((out + "/stats.json").c_str(), "w");
    fprintf(fp,
            "{\"width\": %d, \"height\": %d, \"spp\": %d, \"threads\": %d, \"seconds\": %.6f, \"primary\": %lld, "
            "\"primary_hits\": %lld, \"secondary\": %lld, \"shadow\": %lld}\n",
            W, H, spp, threads, sec, (long long)W * H * (spp > 0 ? spp : 1), hits.load(), g_secondary.load(), g_shadow.load());
    fclose(fp);
    printf("recipe %s %dx%d: %.3f s, hits %lld, secondary %lld, shadow %lld\n", gi ? "P" : spp > 0 ? "S" : "W", W, H, sec, hits.load(),
           g_secondary.load(), g_shadow.load());
    return 0;
}
