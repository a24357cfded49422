// mesh.cpp — OBJ reader, vertex normals, bounding box and the BVH build that the
// reference performs in TriObj::Load (ExternalLibrary/objects.h:52-60). The
// tree must be THE SAME tree cy::BVH builds (same split rule, same element
// order), because the traversal order decides which of two exactly-equal hits
// wins and because the traversal counters are compared
// with the reference's.
#include "scene_graph.h"

#include <cctype>
#include <cstdio>
#include <cstring>

namespace rtu {

namespace {

// One logical line: leading blanks and '#' comment lines skipped, runs of white
// space collapsed to one blank, at most 1023 characters (cyTriMesh.h:273-303).
struct LineReader {
    FILE* fp;
    char  data[1024];
    int read() {
        int c = fgetc(fp);
        while (!feof(fp)) {
            while (isspace(c) && (!feof(fp) || c != '\0')) c = fgetc(fp);
            if (c == '#') {
                while (!feof(fp) && c != '\n' && c != '\r' && c != '\0') c = fgetc(fp);
            } else break;
        }
        int i = 0;
        bool inspace = false;
        while (i < 1024 - 1) {
            if (feof(fp) || c == '\n' || c == '\r' || c == '\0') break;
            if (isspace(c)) inspace = true;
            else {
                if (inspace) data[i++] = ' ';
                inspace = false;
                data[i++] = (char)c;
            }
            c = fgetc(fp);
        }
        data[i] = '\0';
        return i;
    }
    bool is(const char* cmd) const {
        int i = 0;
        for (; cmd[i]; i++)
            if (cmd[i] != data[i]) return false;
        return data[i] == '\0' || data[i] == ' ';
    }
    void vertex(float v[3]) const {  // ReadVertex: missing components stay 0
        v[0] = v[1] = v[2] = 0;
        sscanf(data + 2, "%f %f %f", &v[0], &v[1], &v[2]);
    }
};

}  // namespace

// cy::TriMesh::LoadFromFileObj (cyTriMesh.h:263-450): v / vt / vn / f; polygons are
// fan-triangulated keeping the first vertex; 1-based and negative indices.
bool LoadObjFile(const char* filename, bool loadMtl, MeshData& out, std::string& err) {
    FILE* fp = fopen(filename, "r");
    if (!fp) {
        err = std::string("ERROR: Cannot open file ") + filename;
        return false;
    }
    LineReader L{fp, {0}};
    std::vector<float> v, vt, vn;
    std::vector<uint32_t> f, ft, fn;
    bool hasTextures = false, hasNormals = false;
    bool usesMtl = false;
    while (int rb = L.read()) {
        if (L.is("v")) {
            float p[3]; L.vertex(p); v.insert(v.end(), p, p + 3);
        } else if (L.is("vt")) {
            float p[3]; L.vertex(p); vt.insert(vt.end(), p, p + 3); hasTextures = true;
        } else if (L.is("vn")) {
            float p[3]; L.vertex(p); vn.insert(vn.end(), p, p + 3); hasNormals = true;
        } else if (L.is("f")) {
            int facevert = -1;
            bool inspace = true, negative = false;
            int type = 0;
            uint32_t index = 0;
            uint32_t face[3] = {0, 0, 0}, tface[3] = {0, 0, 0}, nface[3] = {0, 0, 0};
            auto emit = [&]() {
                f.insert(f.end(), face, face + 3);
                if (hasTextures) ft.insert(ft.end(), tface, tface + 3);
                if (hasNormals) fn.insert(fn.end(), nface, nface + 3);
            };
            for (int i = 2; i < rb; i++) {
                char ch = L.data[i];
                if (ch == ' ') { inspace = true; continue; }
                if (inspace) {  // first character of a new vertex token
                    inspace = false; negative = false; type = 0; index = 0;
                    if (facevert < 2) facevert++;
                    else {  // 4th, 5th ... vertex: close the previous triangle, keep v0 and the last vertex
                        emit();
                        face[1] = face[2]; tface[1] = tface[2]; nface[1] = nface[2];
                    }
                }
                if (ch == '/') { type++; index = 0; }
                if (ch == '-') negative = true;
                if (ch >= '0' && ch <= '9') {
                    index = index * 10 + (uint32_t)(ch - '0');
                    switch (type) {
                        case 0: face[facevert] = negative ? (uint32_t)(v.size() / 3) - index : index - 1; break;
                        case 1: tface[facevert] = negative ? (uint32_t)(vt.size() / 3) - index : index - 1; hasTextures = true; break;
                        case 2: nface[facevert] = negative ? (uint32_t)(vn.size() / 3) - index : index - 1; hasNormals = true; break;
                    }
                }
            }
            emit();
        } else if (loadMtl && (L.is("usemtl") || L.is("mtllib"))) {
            usesMtl = true;
        }
        if (feof(fp)) break;
    }
    fclose(fp);
    if (usesMtl) {
        // With loadMtl the reference regroups faces by material and builds a MultiMtl
        // (cyTriMesh.h:461-487, xmlload.cpp:209-243): not on the in-scope path yet.
        err = std::string("OBJ with .mtl materials is not supported yet: ") + filename;
        return false;
    }
    out = MeshData();
    if (f.empty()) return true;  // "No faces found" (:452): the mesh stays empty
    // ft / fn exist only if EVERY face carried them (the reference would index garbage otherwise)
    if (ft.size() != f.size()) { ft.clear(); vt.clear(); }
    if (fn.size() != f.size()) { fn.clear(); vn.clear(); }
    out.v.swap(v); out.f.swap(f); out.vt.swap(vt); out.ft.swap(ft); out.vn.swap(vn); out.fn.swap(fn);
    return true;
}

// cy::TriMesh::ComputeNormals (cyTriMesh.h:248-261): area-weighted vertex normals
void ComputeNormals(MeshData& m) {
    size_t nv = m.v.size() / 3, nf = m.f.size() / 3;
    m.vn.assign(nv * 3, 0.0f);
    m.fn = m.f;
    auto V = [&](uint32_t i) { return Point3(m.v[3 * i], m.v[3 * i + 1], m.v[3 * i + 2]); };
    for (size_t i = 0; i < nf; i++) {
        const uint32_t* fv = &m.f[3 * i];
        Point3 N = Cross(V(fv[1]) - V(fv[0]), V(fv[2]) - V(fv[0]));
        for (int k = 0; k < 3; k++) {
            m.vn[3 * fv[k] + 0] += N.x; m.vn[3 * fv[k] + 1] += N.y; m.vn[3 * fv[k] + 2] += N.z;
        }
    }
    for (size_t i = 0; i < nv; i++) {
        Point3 n = GetNormalized(Point3(m.vn[3 * i], m.vn[3 * i + 1], m.vn[3 * i + 2]));
        m.vn[3 * i] = n.x; m.vn[3 * i + 1] = n.y; m.vn[3 * i + 2] = n.z;
    }
}

// cy::TriMesh::ComputeBoundingBox (cyTriMesh.h:226-246)
void ComputeBoundingBox(MeshData& m) {
    size_t nv = m.v.size() / 3;
    if (nv == 0) {
        m.bound_min[0] = m.bound_min[1] = m.bound_min[2] = 1;
        m.bound_max[0] = m.bound_max[1] = m.bound_max[2] = 0;
        return;
    }
    for (int k = 0; k < 3; k++) m.bound_min[k] = m.bound_max[k] = m.v[k];
    for (size_t i = 1; i < nv; i++)
        for (int k = 0; k < 3; k++) {
            float c = m.v[3 * i + k];
            if (m.bound_min[k] > c) m.bound_min[k] = c;
            if (m.bound_max[k] < c) m.bound_max[k] = c;
        }
}

// ---------------------------------------------------------------------------
// cy::BVH::Build for a triangle mesh (cyBVH.h:122-142, 242-328; BVHTriMesh :339-379)
namespace {

const unsigned kMaxElementCount = 8;  // CY_BVH_MAX_ELEMENT_COUNT, cyBVH.h:44-48

struct Box6 {
    float b[6];
    Box6() { b[0] = b[1] = b[2] = 1e30f; b[3] = b[4] = b[5] = -1e30f; }
    void add(const Box6& o) {
        for (int i = 0; i < 3; i++) {
            if (b[i] > o.b[i]) b[i] = o.b[i];
            if (b[i + 3] < o.b[i + 3]) b[i + 3] = o.b[i + 3];
        }
    }
};

struct Builder {
    const MeshData& m;
    std::vector<uint32_t>& elements;
    unsigned maxPerNode;
    struct Temp {
        int child1 = -1, child2 = -1;
        Box6 box;
        uint32_t count = 0, offset = 0;
    };
    std::vector<Temp> temps;

    Box6 elementBounds(uint32_t i) const {  // BVHTriMesh::GetElementBounds :356-367
        Box6 r;
        const uint32_t* fv = &m.f[3 * i];
        for (int k = 0; k < 3; k++) r.b[k] = r.b[k + 3] = m.v[3 * fv[0] + k];
        for (int j = 1; j < 3; j++)
            for (int k = 0; k < 3; k++) {
                float c = m.v[3 * fv[j] + k];
                if (r.b[k] > c) r.b[k] = c;
                if (r.b[k + 3] < c) r.b[k + 3] = c;
            }
        return r;
    }
    float elementCenter(uint32_t i, int dim) const {  // :370-374
        const uint32_t* fv = &m.f[3 * i];
        return (m.v[3 * fv[0] + dim] + m.v[3 * fv[1] + dim] + m.v[3 * fv[2] + dim]) / 3.0f;
    }

    // MeanSplit (:290-326): midpoint of the widest axis, then the other two axes
    uint32_t meanSplit(uint32_t count, uint32_t* el, const float* box) const {
        if (count <= maxPerNode) return 0;
        float d[3] = {box[3] - box[0], box[4] - box[1], box[5] - box[2]};
        unsigned sd[3];
        sd[0] = d[0] >= d[1] ? (d[0] >= d[2] ? 0 : 2) : (d[1] >= d[2] ? 1 : 2);
        sd[1] = (sd[0] + 1) % 3;
        sd[2] = (sd[0] + 2) % 3;
        if (d[sd[1]] < d[sd[2]]) { unsigned t = sd[1]; sd[1] = sd[2]; sd[2] = t; }
        for (int s = 0; s < 3; s++) {
            unsigned dim = sd[s];
            float splitPos = 0.5f * (box[dim] + box[dim + 3]);
            uint32_t i = 0, j = count;
            while (i < j) {
                if (elementCenter(el[i], (int)dim) <= splitPos) i++;
                else {
                    j--;
                    uint32_t t = el[i]; el[i] = el[j]; el[j] = t;
                }
            }
            if (i < count && i > 0) return i;
        }
        return 0;
    }

    // SplitTempNode (:242-276)
    void split(int t) {
        uint32_t count = temps[t].count, offset = temps[t].offset;
        uint32_t* el = &elements[offset];
        Box6 box = temps[t].box;
        uint32_t c1 = meanSplit(count, el, box.b);
        if (c1 == 0 || c1 >= count) {
            if (count > kMaxElementCount) c1 = count / 2;
            else return;  // leaf
        }
        Box6 b1, b2;
        for (uint32_t i = 0; i < c1; i++) b1.add(elementBounds(el[i]));
        for (uint32_t i = c1; i < count; i++) b2.add(elementBounds(el[i]));
        Temp a, b;
        a.box = b1; a.count = c1; a.offset = offset;
        b.box = b2; b.count = count - c1; b.offset = offset + c1;
        int ia = (int)temps.size();
        temps.push_back(a);
        int ib = (int)temps.size();
        temps.push_back(b);
        temps[t].child1 = ia;
        temps[t].child2 = ib;
        split(ia);
        split(ib);
    }

    // ConvertTempData (:279-288): children of a node are adjacent, ids grow depth-first
    uint32_t convert(std::vector<RtuBvhNode>& nodes, uint32_t id, int t, uint32_t childIndex, uint32_t level, uint32_t& depth) {
        const Temp& T = temps[t];
        RtuBvhNode n;
        for (int k = 0; k < 3; k++) { n.bmin[k] = T.box.b[k]; n.bmax[k] = T.box.b[k + 3]; }
        if (level > depth) depth = level;
        if (T.child1 < 0) {
            n.index = T.offset;
            n.count = T.count;
            nodes[id] = n;
            return childIndex;
        }
        n.index = childIndex;
        n.count = 0;
        nodes[id] = n;
        uint32_t next = convert(nodes, childIndex, T.child1, childIndex + 2, level + 1, depth);
        return convert(nodes, childIndex + 1, T.child2, next, level + 1, depth);
    }
};

}  // namespace

void BuildBVH(MeshData& m, unsigned maxElementsPerNode) {
    m.bvh.clear();
    m.elements.clear();
    m.bvh_depth = 0;
    uint32_t n = (uint32_t)(m.f.size() / 3);
    if (n == 0) return;
    if (maxElementsPerNode > kMaxElementCount) maxElementsPerNode = kMaxElementCount;
    m.elements.resize(n);
    for (uint32_t i = 0; i < n; i++) m.elements[i] = i;
    Builder B{m, m.elements, maxElementsPerNode, {}};
    Box6 box;
    for (uint32_t i = 0; i < n; i++) box.add(B.elementBounds(i));
    Builder::Temp root;
    root.box = box; root.count = n; root.offset = 0;
    B.temps.reserve(2 * (size_t)n + 2);
    B.temps.push_back(root);
    B.split(0);
    RtuBvhNode zero;
    memset(&zero, 0, sizeof zero);
    m.bvh.assign(B.temps.size() + 1, zero);  // node 0 is unused, the root is node 1 (cyBVH.h:76,199)
    uint32_t depth = 0;
    B.convert(m.bvh, 1, 0, 2, 1, depth);
    m.bvh_depth = depth;
}

bool TriObj::Load(const char* filename, bool loadMtl) {
    if (!LoadObjFile(filename, loadMtl, data, error)) return false;
    if (data.vn.empty()) ComputeNormals(data);   // objects.h:56
    ComputeBoundingBox(data);                    // objects.h:57
    BuildBVH(data, 4);                           // objects.h:58
    return true;
}

}  // namespace rtu
