// load_scene.cpp — the input side of the boundary: scene XML -> scene graph
// (the job of ExternalLibrary/xmlload.cpp:64-555) -> flattened RtuSceneDesc.
//
// The VALUES must equal the reference's bit for bit (node tm/itm/pos, camera
// frame, light and material parameters), so attribute defaults, the order in
// which transforms are applied and every float operation follow SURVEY.md
// Appendix E. The XML reader below is a small DOM parser written for this
// project (the reference uses TinyXML).
#include "scene_graph.h"

#include <strings.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace rtu {

// ---------------------------------------------------------------------------
// Matrix3 (cyMatrix.h)
Matrix3 Matrix3::operator*(const Matrix3& right) const {  // :528-541
    Matrix3 r;
    for (int i = 0; i < 9; i += 3)
        for (int j = 0; j < 3; j++) {
            float a = data[j] * right.data[i];
            float b = data[3 + j] * right.data[i + 1];
            float c = data[6 + j] * right.data[i + 2];
            r.data[i + j] = a + b + c;
        }
    return r;
}
Point3 Matrix3::operator*(const Point3& p) const {  // :543-547
    return Point3(p.x * data[0] + p.y * data[3] + p.z * data[6], p.x * data[1] + p.y * data[4] + p.z * data[7],
                  p.x * data[2] + p.y * data[5] + p.z * data[8]);
}
void Matrix3::GetInverse(Matrix3& inv) const {  // :612-633
    inv.data[0] = (data[4] * data[8] - data[5] * data[7]);
    inv.data[1] = (data[2] * data[7] - data[1] * data[8]);
    inv.data[2] = (data[1] * data[5] - data[2] * data[4]);
    inv.data[3] = (data[5] * data[6] - data[3] * data[8]);
    inv.data[4] = (data[0] * data[8] - data[2] * data[6]);
    inv.data[5] = (data[2] * data[3] - data[0] * data[5]);
    inv.data[6] = (data[3] * data[7] - data[4] * data[6]);
    inv.data[7] = (data[1] * data[6] - data[0] * data[7]);
    inv.data[8] = (data[0] * data[4] - data[1] * data[3]);
    float det = data[0] * inv.data[0] + data[1] * inv.data[3] + data[2] * inv.data[6];
    for (float& d : inv.data) d /= det;
}
void Matrix3::SetRotation(const Point3& axis, float angle) {  // :412-430
    const float sinAngle = sinf(angle), cosAngle = cosf(angle);
    const float t = 1.0f - cosAngle;
    const float tx = t * axis.x, ty = t * axis.y, tz = t * axis.z;
    const float txy = tx * axis.y, txz = tx * axis.z, tyz = ty * axis.z;
    const float sx = sinAngle * axis.x, sy = sinAngle * axis.y, sz = sinAngle * axis.z;
    data[0] = tx * axis.x + cosAngle; data[1] = txy + sz;               data[2] = txz - sy;
    data[3] = txy - sz;               data[4] = ty * axis.y + cosAngle; data[5] = tyz + sx;
    data[6] = txz + sy;               data[7] = tyz - sx;               data[8] = tz * axis.z + cosAngle;
}

// Transformation (scene.h:244-247)
void Transformation::Rotate(Point3 axis, float degree) {
    Matrix3 m;
    m.SetRotation(axis, degree * (float)M_PI / 180.0f);
    Transform(m);
}
void Transformation::Scale(float sx, float sy, float sz) {
    Matrix3 m;
    m.Zero();
    m.data[0] = sx; m.data[4] = sy; m.data[8] = sz;
    Transform(m);
}
void Transformation::Transform(const Matrix3& m) {
    tm = m * tm;
    pos = m * pos;
    tm.GetInverse(itm);
}

// ---------------------------------------------------------------------------
// Minimal XML DOM: elements, attributes, comments, declarations; text is ignored.
namespace {

struct XmlElement {
    std::string name;
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<std::unique_ptr<XmlElement>> children;
    const char* attr(const char* key) const {  // attribute names are case sensitive (TinyXML)
        for (auto& a : attrs)
            if (a.first == key) return a.second.c_str();
        return nullptr;
    }
    const XmlElement* first(const char* key) const {  // exact-name child, as FirstChildElement(name)
        for (auto& c : children)
            if (c->name == key) return c.get();
        return nullptr;
    }
};

struct XmlParser {
    const std::string& s;
    size_t i = 0;
    std::string err;
    explicit XmlParser(const std::string& src) : s(src) {}
    void skip_ws() { while (i < s.size() && isspace((unsigned char)s[i])) i++; }
    bool starts(const char* t) const { return s.compare(i, strlen(t), t) == 0; }
    static std::string decode(const std::string& v) {
        std::string o;
        for (size_t k = 0; k < v.size(); k++) {
            if (v[k] != '&') { o += v[k]; continue; }
            static const std::pair<const char*, char> ents[] = {{"&amp;", '&'}, {"&lt;", '<'}, {"&gt;", '>'}, {"&quot;", '"'}, {"&apos;", '\''}};
            bool done = false;
            for (auto& e : ents)
                if (v.compare(k, strlen(e.first), e.first) == 0) { o += e.second; k += strlen(e.first) - 1; done = true; break; }
            if (!done) o += v[k];
        }
        return o;
    }
    // skips comments / declarations / text up to the next element start; false at EOF or a closing tag
    bool next_element_start() {
        for (;;) {
            while (i < s.size() && s[i] != '<') i++;
            if (i >= s.size()) return false;
            if (starts("<!--")) {
                size_t e = s.find("-->", i + 4);
                if (e == std::string::npos) { err = "unterminated comment"; return false; }
                i = e + 3;
            } else if (starts("<?")) {
                size_t e = s.find("?>", i + 2);
                if (e == std::string::npos) { err = "unterminated declaration"; return false; }
                i = e + 2;
            } else if (starts("<!")) {
                size_t e = s.find('>', i);
                if (e == std::string::npos) { err = "unterminated <!"; return false; }
                i = e + 1;
            } else if (starts("</")) {
                return false;
            } else return true;
        }
    }
    std::unique_ptr<XmlElement> element() {
        // s[i] == '<'
        i++;
        std::unique_ptr<XmlElement> e(new XmlElement);
        while (i < s.size() && !isspace((unsigned char)s[i]) && s[i] != '>' && s[i] != '/') e->name += s[i++];
        for (;;) {
            skip_ws();
            if (i >= s.size()) { err = "unexpected end inside <" + e->name; return nullptr; }
            if (s[i] == '/') {
                if (i + 1 < s.size() && s[i + 1] == '>') { i += 2; return e; }
                err = "stray '/' in <" + e->name;
                return nullptr;
            }
            if (s[i] == '>') { i++; break; }
            std::string key;
            while (i < s.size() && !isspace((unsigned char)s[i]) && s[i] != '=' && s[i] != '>' && s[i] != '/') key += s[i++];
            skip_ws();
            if (i >= s.size() || s[i] != '=') { err = "attribute without value in <" + e->name; return nullptr; }
            i++;
            skip_ws();
            if (i >= s.size() || (s[i] != '"' && s[i] != '\'')) { err = "unquoted attribute in <" + e->name; return nullptr; }
            char q = s[i++];
            size_t end = s.find(q, i);
            if (end == std::string::npos) { err = "unterminated attribute in <" + e->name; return nullptr; }
            e->attrs.emplace_back(key, decode(s.substr(i, end - i)));
            i = end + 1;
        }
        // children until the matching close tag
        for (;;) {
            if (next_element_start()) {
                auto c = element();
                if (!c) return nullptr;
                e->children.push_back(std::move(c));
            } else {
                if (!err.empty()) return nullptr;
                if (i < s.size() && starts("</")) {
                    size_t end = s.find('>', i);
                    if (end == std::string::npos) { err = "unterminated close tag"; return nullptr; }
                    i = end + 1;
                    return e;
                }
                err = "missing </" + e->name + ">";
                return nullptr;
            }
        }
    }
};

inline bool same(const std::string& a, const char* b) { return strcasecmp(a.c_str(), b) == 0; }  // COMPARE, xmlload.cpp:33-37

// ReadFloat / ReadVector / ReadColor (xmlload.cpp:452-495): start from the default,
// override per component (parsed as double -> float), then multiply by "value".
void ReadFloat(const XmlElement& e, float& f, const char* name = "value") {
    double d = (double)f;
    if (const char* a = e.attr(name)) {
        double t;
        if (sscanf(a, "%lf", &t) == 1) d = t;
    }
    f = (float)d;
}
void ReadVector(const XmlElement& e, Point3& v) {
    ReadFloat(e, v.x, "x");
    ReadFloat(e, v.y, "y");
    ReadFloat(e, v.z, "z");
    float f = 1;
    ReadFloat(e, f);
    v = v * f;
}
void ReadColor(const XmlElement& e, Color& c) {
    ReadFloat(e, c.r, "r");
    ReadFloat(e, c.g, "g");
    ReadFloat(e, c.b, "b");
    float f = 1;
    ReadFloat(e, f);
    c.r *= f; c.g *= f; c.b *= f;
}

struct Loader {
    SceneGraph& sg;
    std::string remap_from, remap_to;
    std::vector<std::pair<Node*, std::string>> nodeMtlList;

    std::string remap(const std::string& p) const {
        if (!remap_from.empty() && p.compare(0, remap_from.size(), remap_from) == 0) return remap_to + p.substr(remap_from.size());
        return p;
    }

    // ReadTexture (xmlload.cpp:499-555)
    void ReadTexture(const XmlElement& e, TexturedColor& tc) {
        const char* texName = e.attr("texture");
        if (!texName) return;
        const Texture* tex = nullptr;
        if (same(texName, "checkerboard")) {
            std::unique_ptr<Texture> c(new Texture);
            c->type = RTU_TEX_CHECKER;
            c->name = texName;
            for (auto& ch : e.children) {
                if (same(ch->name, "color1")) { Color col(0, 0, 0); ReadColor(*ch, col); c->color1 = col; }
                else if (same(ch->name, "color2")) { Color col(0, 0, 0); ReadColor(*ch, col); c->color2 = col; }
            }
            tex = c.get();
            sg.textureList.push_back(std::move(c));
        } else {
            for (auto& t : sg.textureList)
                if (t->type == RTU_TEX_FILE && t->name == texName) tex = t.get();
            if (!tex) {
                std::unique_ptr<Texture> f(new Texture);
                f->type = RTU_TEX_FILE;
                f->name = texName;
                if (LoadTextureFile(remap(texName).c_str(), *f)) {  // a file that fails to load leaves TextureMap(NULL): black
                    tex = f.get();
                    sg.textureList.push_back(std::move(f));
                }
            }
        }
        tc.map.reset(new TextureMap);
        tc.map->texture = tex;
        LoadTransform(*tc.map, e);
    }

    // LoadTransform (xmlload.cpp:264-290): in XML child order
    void LoadTransform(Transformation& t, const XmlElement& e) {
        for (auto& c : e.children) {
            if (same(c->name, "scale")) {
                Point3 s(1, 1, 1);
                ReadVector(*c, s);
                t.Scale(s.x, s.y, s.z);
            } else if (same(c->name, "rotate")) {
                Point3 s(0, 0, 0);
                ReadVector(*c, s);
                s = GetNormalized(s);  // s.Normalize(), :274
                float a = 0;           // uninitialised in the reference when "angle" is absent
                ReadFloat(*c, a, "angle");
                t.Rotate(s, a);
            } else if (same(c->name, "translate")) {
                Point3 p(0, 0, 0);
                ReadVector(*c, p);
                t.Translate(p);
            }
        }
    }

    // LoadNode (xmlload.cpp:167-260)
    void LoadNode(Node* parent, const XmlElement& e) {
        Node* node = parent->AppendChild();
        const char* name = e.attr("name");
        node->name = name ? name : "";
        const char* mtlName = e.attr("material");
        if (mtlName) nodeMtlList.emplace_back(node, mtlName);
        const char* type = e.attr("type");
        if (type) {
            if (same(type, "sphere")) node->obj = &sg.theSphere;
            else if (same(type, "plane")) node->obj = &sg.thePlane;
            else if (same(type, "obj")) {
                std::string key = node->name;
                TriObj* obj = nullptr;
                for (auto& o : sg.objList)
                    if (o.first == key) obj = o.second.get();
                if (!obj) {
                    std::unique_ptr<TriObj> t(new TriObj);
                    if (!t->Load(remap(key).c_str(), mtlName == nullptr)) {
                        // the reference prints the error and leaves the node without an object (:205-207); a face that
                        // points outside its vertex lists (where the reference reads out of bounds) fails the scene
                        if (sg.error.empty() && t->error.find("does not exist") != std::string::npos) sg.error = t->error;
                    } else {
                        obj = t.get();
                        if (t->NM() > 0) MakeMultiMtl(*t, node, key);
                        sg.objList.emplace_back(key, std::move(t));
                    }
                }
                node->obj = obj;
            }
        }
        for (auto& c : e.children)
            if (same(c->name, "object")) LoadNode(node, *c);
        LoadTransform(*node, e);  // after the children, :253-258
    }

    // File texture by name, shared through textureList (Texture* ReadTexture(const char*), xmlload.cpp:535-555)
    const Texture* ReadTextureFile(const char* texName) {
        for (auto& t : sg.textureList)
            if (t->name == texName) return t.get();
        std::unique_ptr<Texture> f(new Texture);
        f->type = RTU_TEX_FILE;
        f->name = texName;
        if (!LoadTextureFile(remap(texName).c_str(), *f)) return nullptr;  // stays TextureMap(NULL): black
        sg.textureList.push_back(std::move(f));
        return sg.textureList.back().get();
    }

    // The .obj of a node without material= brought its own materials (usemtl / .mtl): one MtlBlinn per material,
    // together a MultiMtl named like the file, bound to the node by that name (xmlload.cpp:208-241) — unless a
    // material of that name already exists. Reproduced as written, including :222 (map_Ks lands on the DIFFUSE
    // texture, replacing map_Kd's) and the unused gloss of :226.
    void MakeMultiMtl(const TriObj& obj, Node* node, const std::string& name) {
        for (auto& m : sg.materials)
            if (m->name == name) return;
        std::unique_ptr<MultiMtl> mm(new MultiMtl);
        auto set_map = [&](TexturedColor& tc, const std::string& file) {
            tc.map.reset(new TextureMap);
            tc.map->texture = ReadTextureFile(file.c_str());
        };
        for (const ObjMtl& mtl : obj.mtls) {
            MtlBlinn* m = new MtlBlinn;
            m->diffuse.SetColor(Color(mtl.Kd[0], mtl.Kd[1], mtl.Kd[2]));
            m->specular.SetColor(Color(mtl.Ks[0], mtl.Ks[1], mtl.Ks[2]));
            m->glossiness = mtl.Ns;
            m->ior = mtl.Ni;
            if (mtl.has_map_Kd) set_map(m->diffuse, mtl.map_Kd);
            if (mtl.has_map_Ks) set_map(m->diffuse, mtl.map_Ks);
            if (mtl.illum > 2 && mtl.illum <= 7) {
                m->reflection.SetColor(Color(mtl.Ks[0], mtl.Ks[1], mtl.Ks[2]));
                if (mtl.has_map_Ks) set_map(m->reflection, mtl.map_Ks);
                if (mtl.illum >= 6) m->refraction.SetColor(Color(-(mtl.Tf[0] - 1), -(mtl.Tf[1] - 1), -(mtl.Tf[2] - 1)));  // 1 - Color is -(c - 1), cyColor.h:56
            }
            mm->AppendMaterial(m);
        }
        mm->name = name;
        sg.materials.emplace_back(std::move(mm));
        nodeMtlList.emplace_back(node, name);
    }

    // LoadMaterial (xmlload.cpp:294-370)
    void LoadMaterial(const XmlElement& e) {
        const char* type = e.attr("type");
        if (!type || !same(type, "blinn")) return;
        std::unique_ptr<MtlBlinn> m(new MtlBlinn);
        for (auto& c : e.children) {
            Color col(1, 1, 1);
            float f = 1;
            if (same(c->name, "diffuse")) { ReadColor(*c, col); m->diffuse.SetColor(col); ReadTexture(*c, m->diffuse); }
            else if (same(c->name, "specular")) { ReadColor(*c, col); m->specular.SetColor(col); ReadTexture(*c, m->specular); }
            else if (same(c->name, "glossiness")) { ReadFloat(*c, f); m->glossiness = f; }
            else if (same(c->name, "emission")) { ReadColor(*c, col); m->emission.SetColor(col); ReadTexture(*c, m->emission); }
            else if (same(c->name, "reflection")) {
                ReadColor(*c, col); m->reflection.SetColor(col); ReadTexture(*c, m->reflection);
                f = 0; ReadFloat(*c, f, "glossiness"); m->reflectionGlossiness = f;
            } else if (same(c->name, "refraction")) {
                ReadColor(*c, col); m->refraction.SetColor(col);
                ReadFloat(*c, f, "index"); m->ior = f;
                ReadTexture(*c, m->refraction);
                f = 0; ReadFloat(*c, f, "glossiness"); m->refractionGlossiness = f;
            } else if (same(c->name, "absorption")) { ReadColor(*c, col); m->absorption = col; }
        }
        const char* name = e.attr("name");
        m->name = name ? name : "";
        sg.materials.emplace_back(std::move(m));
    }

    // LoadLight (xmlload.cpp:374-448)
    void LoadLight(const XmlElement& e) {
        const char* type = e.attr("type");
        if (!type) return;
        std::unique_ptr<Light> light;
        if (same(type, "ambient")) {
            light.reset(new AmbientLight);
            for (auto& c : e.children)
                if (same(c->name, "intensity")) { Color col(1, 1, 1); ReadColor(*c, col); light->intensity = col; }
        } else if (same(type, "direct")) {
            DirectLight* l = new DirectLight;
            light.reset(l);
            for (auto& c : e.children) {
                if (same(c->name, "intensity")) { Color col(1, 1, 1); ReadColor(*c, col); l->intensity = col; }
                else if (same(c->name, "direction")) { Point3 v(1, 1, 1); ReadVector(*c, v); l->SetDirection(v); }
            }
        } else if (same(type, "point")) {
            PointLight* l = new PointLight;
            light.reset(l);
            for (auto& c : e.children) {
                if (same(c->name, "intensity")) { Color col(1, 1, 1); ReadColor(*c, col); l->intensity = col; }
                else if (same(c->name, "position")) { Point3 v(0, 0, 0); ReadVector(*c, v); l->position = v; }
                else if (same(c->name, "size")) { float f = 0; ReadFloat(*c, f); l->size = f; }
            }
        } else return;
        const char* name = e.attr("name");
        light->name = name ? name : "";
        sg.lights.emplace_back(std::move(light));
    }
};

}  // namespace

bool LoadScene(const char* filename, const std::string& remap_from, const std::string& remap_to, SceneGraph& sg, std::string& err) {
    std::ifstream in(filename, std::ios::binary);
    if (!in) {
        err = std::string("Failed to load the file \"") + filename + "\"";
        return false;
    }
    std::stringstream ss;
    ss << in.rdbuf();
    std::string text = ss.str();
    XmlParser P(text);
    std::unique_ptr<XmlElement> xml;
    while (P.next_element_start()) {  // doc.FirstChildElement("xml")
        auto e = P.element();
        if (!e) { err = "XML parse error: " + P.err; return false; }
        if (e->name == "xml") { xml = std::move(e); break; }
    }
    if (!P.err.empty()) { err = "XML parse error: " + P.err; return false; }
    if (!xml) { err = "No \"xml\" tag found."; return false; }
    const XmlElement* scene = xml->first("scene");
    if (!scene) { err = "No \"scene\" tag found."; return false; }
    const XmlElement* cam = xml->first("camera");
    if (!cam) { err = "No \"camera\" tag found."; return false; }

    Loader L{sg, remap_from, remap_to, {}};
    for (auto& c : scene->children) {  // LoadScene(TiXmlElement*), xmlload.cpp:139-163
        if (same(c->name, "background")) {
            Color col(1, 1, 1); ReadColor(*c, col); sg.background.SetColor(col); L.ReadTexture(*c, sg.background);
        } else if (same(c->name, "environment")) {
            Color col(1, 1, 1); ReadColor(*c, col); sg.environment.SetColor(col); L.ReadTexture(*c, sg.environment);
        } else if (same(c->name, "object")) L.LoadNode(&sg.rootNode, *c);
        else if (same(c->name, "material")) L.LoadMaterial(*c);
        else if (same(c->name, "light")) L.LoadLight(*c);
    }

    for (auto& nm : L.nodeMtlList)  // xmlload.cpp:100-106: first material with that exact name
        for (auto& m : sg.materials)
            if (m->name == nm.second) { nm.first->mtl = m.get(); break; }

    // camera, xmlload.cpp:108-126
    Camera& camera = sg.camera;
    camera.Init();
    camera.dir = camera.dir + camera.pos;
    for (auto& c : cam->children) {
        if (same(c->name, "position")) ReadVector(*c, camera.pos);
        else if (same(c->name, "target")) ReadVector(*c, camera.dir);
        else if (same(c->name, "up")) ReadVector(*c, camera.up);
        else if (same(c->name, "fov")) ReadFloat(*c, camera.fov);
        else if (same(c->name, "focaldist")) ReadFloat(*c, camera.focaldist);
        else if (same(c->name, "dof")) ReadFloat(*c, camera.dof);
        else if (same(c->name, "width")) { if (const char* a = c->attr("value")) sscanf(a, "%d", &camera.imgWidth); }
        else if (same(c->name, "height")) { if (const char* a = c->attr("value")) sscanf(a, "%d", &camera.imgHeight); }
    }
    camera.dir = camera.dir - camera.pos;
    camera.dir = GetNormalized(camera.dir);
    Point3 x = Cross(camera.dir, camera.up);
    camera.up = GetNormalized(Cross(x, camera.dir));
    return true;
}

namespace {

void flatten_node(const SceneGraph& sg, const Node* n, int parent, int depth, Scene& out, std::map<const TriObj*, int>& meshIds) {
    int me = (int)out.nodes.size();
    RtuNode o;
    memset(&o, 0, sizeof o);
    memcpy(o.tm, n->GetTransform().data, sizeof o.tm);
    memcpy(o.itm, n->GetInverseTransform().data, sizeof o.itm);
    o.pos[0] = n->GetPosition().x; o.pos[1] = n->GetPosition().y; o.pos[2] = n->GetPosition().z;
    o.parent = parent;
    o.depth = depth;
    o.mesh_id = -1;
    o.material_id = -1;
    if (n->mtl)
        for (size_t i = 0; i < sg.materials.size(); i++)
            if (sg.materials[i].get() == n->mtl) { o.material_id = (int)i; break; }
    o.obj_type = n->obj ? n->obj->Type() : RTU_OBJ_NONE;
    if (o.obj_type == RTU_OBJ_TRIMESH) {
        const TriObj* t = static_cast<const TriObj*>(n->obj);
        if (t->data.f.empty()) {
            o.obj_type = RTU_OBJ_NONE;  // empty mesh: its (empty) bounding box rejects every ray (objFunctions.cpp:337)
        } else {
            auto it = meshIds.find(t);
            if (it == meshIds.end()) {
                it = meshIds.emplace(t, (int)out.meshes.size()).first;
                out.meshes.push_back(t->data);
            }
            o.mesh_id = it->second;
        }
    }
    out.nodes.push_back(o);
    for (auto& c : n->child) flatten_node(sg, c.get(), me, depth + 1, out, meshIds);
    out.nodes[me].subtree_end = (int)out.nodes.size();
}

RtuEnvColor flatten_env(const TexturedColor& t) {
    RtuEnvColor e;
    memset(&e, 0, sizeof e);
    e.color[0] = t.color.r; e.color[1] = t.color.g; e.color[2] = t.color.b;
    e.has_map = t.map ? 1 : 0;
    e.map_is_null = (t.map && !t.map->texture) ? 1 : 0;
    return e;
}

// TextureMap -> RtuTexMap; textures are numbered in textureList order
RtuTexMap flatten_map(const SceneGraph& sg, const TexturedColor& t) {
    RtuTexMap m;
    memset(&m, 0, sizeof m);
    m.texture = -1;
    if (!t.map) return m;
    m.present = 1;
    for (size_t i = 0; i < sg.textureList.size(); i++)
        if (sg.textureList[i].get() == t.map->texture) m.texture = (int32_t)i;
    const Matrix3& tm = t.map->GetTransform();
    const Matrix3& itm = t.map->GetInverseTransform();
    const Point3 pos = t.map->GetPosition();
    for (int k = 0; k < 9; k++) { m.tm[k] = tm.data[k]; m.itm[k] = itm.data[k]; }
    m.pos[0] = pos.x; m.pos[1] = pos.y; m.pos[2] = pos.z;
    return m;
}

}  // namespace

Scene* Flatten(const SceneGraph& sg, std::string& err) {
    if (!sg.error.empty()) {
        err = sg.error;
        return nullptr;
    }
    std::unique_ptr<Scene> s(new Scene);
    std::vector<RtuTexMap> maps;
    bool any_map = false;
    for (auto& mp : sg.materials) {
        const MtlBlinn* b = dynamic_cast<const MtlBlinn*>(mp.get());
        if (const MultiMtl* mm = dynamic_cast<const MultiMtl*>(mp.get()))
            b = mm->mtls.empty() ? nullptr : mm->mtls[0].get();  // hInfo.mtlID is always 0 (scene.h:159,162; materials.h:66)
        RtuMaterial m;
        memset(&m, 0, sizeof m);
        if (!b) { err = "unsupported material class"; return nullptr; }
        auto put = [](float* d, const Color& c) { d[0] = c.r; d[1] = c.g; d[2] = c.b; };
        put(m.diffuse, b->diffuse.color);
        put(m.specular, b->specular.color);
        put(m.reflection, b->reflection.color);
        put(m.refraction, b->refraction.color);
        put(m.emission, b->emission.color);
        put(m.absorption, b->absorption);
        m.glossiness = b->glossiness;
        m.ior = b->ior;
        m.reflection_glossiness = b->reflectionGlossiness;
        m.refraction_glossiness = b->refractionGlossiness;
        // a material map whose texture failed to load multiplies the colour by black (scene.h:382,421):
        // folded into the colour, so that such a scene stays an untextured one
        const TexturedColor* tcs[5] = {&b->diffuse, &b->specular, &b->reflection, &b->refraction, &b->emission};
        float* dst[5] = {m.diffuse, m.specular, m.reflection, m.refraction, m.emission};
        for (int k = 0; k < 5; k++)
            if (tcs[k]->map && !tcs[k]->map->texture)
                for (int j = 0; j < 3; j++) dst[k][j] = dst[k][j] * 0.0f;
        s->materials.push_back(m);
        for (int k = 0; k < 4; k++) {
            RtuTexMap tm = flatten_map(sg, *tcs[k]);
            if (tm.present && tm.texture < 0) memset(&tm, 0, sizeof tm), tm.texture = -1;  // folded above
            any_map = any_map || tm.present;
            maps.push_back(tm);
        }
    }
    if (any_map) s->material_maps = maps;
    for (auto& t : sg.textureList) {
        TextureData td;
        memset(&td.hdr, 0, sizeof td.hdr);
        td.hdr.type = t->type; td.hdr.width = t->width; td.hdr.height = t->height;
        if (t->type == RTU_TEX_CHECKER) {
            td.hdr.color1[0] = t->color1.r; td.hdr.color1[1] = t->color1.g; td.hdr.color1[2] = t->color1.b;
            td.hdr.color2[0] = t->color2.r; td.hdr.color2[1] = t->color2.g; td.hdr.color2[2] = t->color2.b;
        }
        td.rgb = t->rgb;
        s->textures.push_back(std::move(td));
    }
    for (auto& lp : sg.lights) {
        RtuLight l;
        memset(&l, 0, sizeof l);
        l.type = lp->Type();
        l.intensity[0] = lp->intensity.r; l.intensity[1] = lp->intensity.g; l.intensity[2] = lp->intensity.b;
        if (const DirectLight* d = dynamic_cast<const DirectLight*>(lp.get())) {
            l.vec[0] = d->direction.x; l.vec[1] = d->direction.y; l.vec[2] = d->direction.z;
        } else if (const PointLight* p = dynamic_cast<const PointLight*>(lp.get())) {
            l.vec[0] = p->position.x; l.vec[1] = p->position.y; l.vec[2] = p->position.z;
            l.size = p->size;
        }
        s->lights.push_back(l);
    }
    std::map<const TriObj*, int> meshIds;
    flatten_node(sg, &sg.rootNode, -1, 0, *s, meshIds);
    const Camera& c = sg.camera;
    RtuCamera& o = s->camera;
    o.pos[0] = c.pos.x; o.pos[1] = c.pos.y; o.pos[2] = c.pos.z;
    o.dir[0] = c.dir.x; o.dir[1] = c.dir.y; o.dir[2] = c.dir.z;
    o.up[0] = c.up.x; o.up[1] = c.up.y; o.up[2] = c.up.z;
    o.fov = c.fov; o.focaldist = c.focaldist; o.dof = c.dof;
    o.img_width = c.imgWidth; o.img_height = c.imgHeight;
    s->background = flatten_env(sg.background);
    s->environment = flatten_env(sg.environment);
    // a map whose texture is NULL is fully described by RtuEnvColor (has_map, map_is_null): no RtuTexMap
    auto real_map = [&](const TexturedColor& t) {
        RtuTexMap m = flatten_map(sg, t);
        if (m.texture < 0) { memset(&m, 0, sizeof m); m.texture = -1; }
        return m;
    };
    s->background_map = real_map(sg.background);
    s->environment_map = real_map(sg.environment);
    s->rebuild_desc();
    return s.release();
}

}  // namespace rtu

extern "C" RtuScene* rtu_scene_load_xml(const char* xml_path, const char* remap_from, const char* remap_to) {
    if (!xml_path) {
        rtu::set_error("xml_path is NULL");
        return nullptr;
    }
    rtu::SceneGraph sg;
    std::string err;
    if (!rtu::LoadScene(xml_path, remap_from ? remap_from : "", remap_to ? remap_to : "", sg, err)) {
        rtu::set_error(err);
        return nullptr;
    }
    rtu::Scene* s = rtu::Flatten(sg, err);
    if (!s) {
        rtu::set_error(err);
        return nullptr;
    }
    return rtu_scene_wrap(s);
}
