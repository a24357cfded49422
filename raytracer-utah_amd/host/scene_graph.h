// scene_graph.h — host-side mirror of the reference's plugin surface
// (ExternalLibrary/scene.h, objects.h, materials.h, lights.h). The class and
// member names follow the reference so a maintainer can map one onto the other;
// the implementation is this project's own. The render path itself never walks
// these objects: Flatten() turns them into the pointer-free RtuSceneDesc that
// crosses the C-ABI.
#ifndef RTU_SCENE_GRAPH_H
#define RTU_SCENE_GRAPH_H

#include "host_internal.h"

#include <cmath>
#include <memory>
#include <string>
#include <vector>

namespace rtu {

struct Point3 {
    float x = 0, y = 0, z = 0;
    Point3() {}
    Point3(float X, float Y, float Z) : x(X), y(Y), z(Z) {}
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
inline Point3 operator+(Point3 a, Point3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Point3 operator-(Point3 a, Point3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Point3 operator*(Point3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Point3 operator/(Point3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float Dot(Point3 a, Point3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }   // cyPoint.h:296,348
inline Point3 Cross(Point3 a, Point3 b) {                                                 // cyPoint.h:346
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float Length(Point3 a) { return sqrtf(Dot(a, a)); }
inline Point3 GetNormalized(Point3 a) { return a / Length(a); }                           // cyPoint.h:295

struct Color {
    float r = 0, g = 0, b = 0;
    Color() {}
    Color(float R, float G, float B) : r(R), g(G), b(B) {}
};

// cyMatrix.h Matrix3: column-major data[9] (:290-294)
struct Matrix3 {
    float data[9];
    void Zero() { for (float& d : data) d = 0; }
    void SetIdentity() { Zero(); data[0] = data[4] = data[8] = 1; }
    Matrix3 operator*(const Matrix3& right) const;    // :528-541
    Point3 operator*(const Point3& p) const;          // :543-547
    void GetInverse(Matrix3& inverse) const;          // :612-633 (adjugate / det)
    void SetRotation(const Point3& axis, float angle);  // :412-430
};

// scene.h:223-262
class Transformation {
public:
    Transformation() { InitTransform(); }
    const Matrix3& GetTransform() const { return tm; }
    const Point3& GetPosition() const { return pos; }
    const Matrix3& GetInverseTransform() const { return itm; }
    void Translate(Point3 p) { pos = pos + p; }
    void Rotate(Point3 axis, float degree);
    void Scale(float sx, float sy, float sz);
    void Transform(const Matrix3& m);
    void InitTransform() { pos = Point3(0, 0, 0); tm.SetIdentity(); itm.SetIdentity(); }
private:
    Matrix3 tm;
    Point3  pos;
    Matrix3 itm;
};

// scene.h:268-275, objects.h
class Object {
public:
    virtual ~Object() {}
    virtual int Type() const = 0;  // RTU_OBJ_*; replaces the IntersectRay virtual on the device
};
class Sphere : public Object { public: int Type() const override { return RTU_OBJ_SPHERE; } };
class Plane : public Object { public: int Type() const override { return RTU_OBJ_PLANE; } };

// cy::TriMesh::Mtl (cyTriMesh.h:72-103): a material of a .mtl library that the .obj uses (usemtl) — the fields
// LoadNode turns into a MtlBlinn (xmlload.cpp:209-232)
struct ObjMtl {
    std::string used_name;   // as written after usemtl
    std::string name;        // as written after newmtl (empty if no library defines it: the defaults below stay)
    float Ka[3] = {0, 0, 0}, Kd[3] = {1, 1, 1}, Ks[3] = {0, 0, 0}, Tf[3] = {0, 0, 0};
    float Ns = 0, Ni = 1;
    int   illum = 2;
    bool  has_map_Kd = false, has_map_Ks = false;
    std::string map_Kd, map_Ks;
    uint32_t first_face = 0, face_count = 0;   // while reading: the first face line under it, faces read under it
    uint32_t cumulative_face_count = 0;        // mcfc: faces of materials 0..this one after regrouping
};

// objects.h:45-66 = cyTriMesh + cyBVHTriMesh
class TriObj : public Object {
public:
    int Type() const override { return RTU_OBJ_TRIMESH; }
    // objects.h:52-60: LoadFromFileObj, ComputeNormals if none, ComputeBoundingBox, bvh.SetMesh(this,4)
    bool Load(const char* filename, bool loadMtl);
    unsigned NM() const { return (unsigned)mtls.size(); }
    MeshData data;
    std::vector<ObjMtl> mtls;   // loadMtl only
    std::string error;
};

// cyTriMesh.h:263-547; mtls_out (may be NULL): the materials the file uses, filled from its .mtl libraries
bool LoadObjFile(const char* filename, bool loadMtl, MeshData& out, std::string& err, std::vector<ObjMtl>* mtls_out = nullptr);
void ComputeNormals(MeshData& m);                                                        // cyTriMesh.h:248-261
void ComputeBoundingBox(MeshData& m);                                                    // cyTriMesh.h:226-246
void BuildBVH(MeshData& m, unsigned maxElementsPerNode);                                 // cyBVH.h:122-142,242-328

// Texture (scene.h:308-365): TextureFile (texture.h:14-31) or TextureChecker (:34-46); the sampling
// itself happens on the device / in the oracle, here they are data.
struct Texture {
    int type = RTU_TEX_CHECKER;
    std::string name;                 // file textures are shared by name (textureList.Find, xmlload.cpp:538)
    int width = 0, height = 0;
    std::vector<uint8_t> rgb;         // width*height*3
    Color color1 = Color(0, 0, 0), color2 = Color(1, 1, 1);  // TextureChecker(), texture.h:37
};
bool LoadTextureFile(const char* filename, Texture& out);  // TextureFile::Load (texture.cpp:56-90): .png / .ppm

// TextureMap (scene.h:375-397): a texture (may be NULL) under its own transformation
class TextureMap : public Transformation {
public:
    const Texture* texture = nullptr;
};

// TexturedColor (scene.h:405-433)
struct TexturedColor {
    Color color;
    std::unique_ptr<TextureMap> map;
    void SetColor(const Color& c) { color = c; }
};

// scene.h:298-314, materials.h:19-57
class Material {
public:
    virtual ~Material() {}
    std::string name;
};
class MtlBlinn : public Material {
public:
    TexturedColor diffuse, specular, reflection, refraction, emission;
    float glossiness = 20.0f;
    Color absorption;
    float ior = 1;
    float reflectionGlossiness = 0, refractionGlossiness = 0;
    MtlBlinn() {
        diffuse.color = Color(0.5f, 0.5f, 0.5f);
        specular.color = Color(0.7f, 0.7f, 0.7f);
    }
};

// materials.h:61-82. HitInfo::mtlID is initialised to 0 and written by nobody (scene.h:159,162), so Shade() always
// forwards to the FIRST sub-material: that is what Flatten() emits for a node bound to a MultiMtl.
class MultiMtl : public Material {
public:
    std::vector<std::unique_ptr<MtlBlinn>> mtls;
    void AppendMaterial(MtlBlinn* m) { mtls.emplace_back(m); }
};

// scene.h:280-293, lights.h
class Light {
public:
    virtual ~Light() {}
    virtual int Type() const = 0;
    std::string name;
    Color intensity;
};
class AmbientLight : public Light { public: int Type() const override { return RTU_LIGHT_AMBIENT; } };
class DirectLight : public Light {
public:
    Point3 direction{0, 0, 1};
    void SetDirection(Point3 d) { direction = GetNormalized(d); }  // lights.h:53
    int Type() const override { return RTU_LIGHT_DIRECT; }
};
class PointLight : public Light {
public:
    Point3 position;
    float  size = 0;
    int Type() const override { return RTU_LIGHT_POINT; }
};

// scene.h:437-513
class Node : public Transformation {
public:
    std::string name;
    std::vector<std::unique_ptr<Node>> child;
    Object*   obj = nullptr;   // not owned (scene.h:442)
    Material* mtl = nullptr;
    int  GetNumChild() const { return (int)child.size(); }
    Node* AppendChild() { child.emplace_back(new Node); return child.back().get(); }
};

// scene.h:517-535
struct Camera {
    Point3 pos, dir, up;
    float fov, focaldist, dof;
    int imgWidth, imgHeight;
    void Init() {
        pos = Point3(0, 0, 0); dir = Point3(0, 0, -1); up = Point3(0, 1, 0);
        fov = 40; focaldist = 1; dof = 0; imgWidth = 200; imgHeight = 150;
    }
};

// The reference keeps these as process globals (main.cpp:17-27); here they are
// one object so several scenes can coexist.
struct SceneGraph {
    Node   rootNode;
    Camera camera;
    Sphere theSphere;
    Plane  thePlane;
    std::vector<std::unique_ptr<Material>> materials;
    std::vector<std::unique_ptr<Light>>    lights;
    std::vector<std::pair<std::string, std::unique_ptr<TriObj>>> objList;
    TexturedColor background, environment;
    std::vector<std::unique_ptr<Texture>> textureList;  // xmlload.cpp:45
    std::string error;   // why the scene cannot be flattened (e.g. a textured material), empty if fine
};

// int LoadScene(const char*) (xmlload.cpp:64-131): true on success.
bool LoadScene(const char* filename, const std::string& remap_from, const std::string& remap_to, SceneGraph& sg,
               std::string& err);
// Scene graph -> flattened, owned scene (pre-order nodes, meshes deduplicated).
Scene* Flatten(const SceneGraph& sg, std::string& err);

}  // namespace rtu
#endif
