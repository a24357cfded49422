// mesh.cpp — what TriObj::Load does in the reference (ExternalLibrary/objects.h:52-60): read the
// .obj (and, for a node without material=, its .mtl libraries), vertex normals when the file
// has none, the bounding box, and the BVH.
//
// The VALUES must be the reference's — the element order and the tree decide which of two
// exactly-equal hits wins and the traversal counters are compared with the
// reference's; tests/test_host.py::test_loader_matches_reference_scene_values holds this file to
// byte identity with the scene the compiled reference builds. The code is this project's own:
// the file is parsed from memory by a small line cursor, the tree is written straight into its
// final numbering (no intermediate node list).
#include "scene_graph.h"

#include <cctype>
#include <cstdio>
#include <cstring>
#include <functional>

namespace rtu {

namespace {

// ---- text ------------------------------------------------------------------------------------
// Logical lines of an .obj / .mtl file as cy::TriMesh sees them (cyTriMesh.h:273-303): blank lines
// and lines starting with '#' vanish, white space inside a line collapses to single blanks, a line
// holds at most 1023 characters (what is left over reads as the next line), a NUL byte ends a line.
class LineCursor {
public:
    explicit LineCursor(std::string text) : text_(std::move(text)) {}

    // Next non-empty logical line into `line`; false at the end of the text (or at a line that is empty
    // after all — a NUL byte — where the reference's read loop stops too).
    bool next(std::string& line) {
        line.clear();
        for (;;) {  // blank space and comment lines before the line
            while (pos_ < text_.size() && isspace((unsigned char)text_[pos_])) pos_++;
            if (pos_ >= text_.size() || text_[pos_] != '#') break;
            while (pos_ < text_.size() && !ends_line(text_[pos_])) pos_++;
            if (pos_ < text_.size()) pos_++;
        }
        bool gap = false;
        while (pos_ < text_.size() && line.size() < kMaxLine) {
            const char c = text_[pos_];
            if (ends_line(c)) { pos_++; break; }  // the terminator is consumed with its line
            pos_++;
            if (isspace((unsigned char)c)) { gap = true; continue; }
            if (gap) {
                gap = false;
                line.push_back(' ');
                if (line.size() == kMaxLine) { pos_--; break; }  // the character itself opens the next line
            }
            line.push_back(c);
        }
        return !line.empty();
    }
    bool at_end() const { return pos_ >= text_.size(); }

private:
    static constexpr size_t kMaxLine = 1023;
    static bool ends_line(char c) { return c == '\n' || c == '\r' || c == '\0'; }
    std::string text_;
    size_t pos_ = 0;
};

bool read_file(const char* filename, std::string& out) {
    FILE* fp = fopen(filename, "rb");
    if (!fp) return false;
    char chunk[65536];
    size_t n;
    out.clear();
    while ((n = fread(chunk, 1, sizeof chunk, fp)) > 0) out.append(chunk, n);
    fclose(fp);
    return true;
}

// `line` is the command `cmd` (followed by a blank or nothing): Buffer::IsCommand
bool is_command(const std::string& line, const char* cmd) {
    const size_t n = strlen(cmd);
    return line.compare(0, n, cmd) == 0 && (line.size() == n || line[n] == ' ');
}
// the text after the command word and its blank ("usemtl NAME" -> "NAME"); empty when there is none
std::string argument(const std::string& line, size_t from) { return from < line.size() ? line.substr(from) : std::string(); }

// up to three floats after the two-character command position (ReadVertex / ReadFloat3 read from data+2)
int scan3(const std::string& line, float out[3]) {
    out[0] = out[1] = out[2] = 0;
    if (line.size() <= 2) return 0;
    const int n = sscanf(line.c_str() + 2, "%f %f %f", &out[0], &out[1], &out[2]);
    return n < 0 ? 0 : n;
}

// One vertex reference of a face line: "v", "v/vt", "v//vn", "v/vt/vn" with 1-based or negative (relative)
// indices. As in the reference's character loop (cyTriMesh.h:421-431) a '-' anywhere marks every field of
// the reference from there on as relative, characters that are neither digits, '/' nor '-' are skipped and a
// field without digits leaves its slot untouched.
struct FaceRef {
    bool     has[3] = {false, false, false};
    uint32_t value[3] = {0, 0, 0};  // the number as written
    bool     relative[3] = {false, false, false};
};
FaceRef parse_face_ref(const char* s, size_t n) {
    FaceRef r;
    int field = 0;
    bool rel = false;
    uint32_t number = 0;
    for (size_t i = 0; i < n; i++) {
        const char c = s[i];
        if (c == '/') { field++; number = 0; }
        else if (c == '-') rel = true;
        else if (c >= '0' && c <= '9') {
            number = number * 10u + (uint32_t)(c - '0');
            if (field < 3) { r.has[field] = true; r.value[field] = number; r.relative[field] = rel; }
        }
    }
    return r;
}

// ---- .mtl ------------------------------------------------------------------------------------
// cyTriMesh.h:505-543: every library named by a mtllib line is read; only materials the .obj uses exist
void read_mtl_libraries(const char* obj_filename, const std::vector<std::string>& libs, std::vector<ObjMtl>& mtls) {
    std::string dir;
    if (const char* slash = strrchr(obj_filename, '\\')) dir.assign(obj_filename, (size_t)(slash - obj_filename) + 1);
    else if (const char* fwd = strrchr(obj_filename, '/')) dir.assign(obj_filename, (size_t)(fwd - obj_filename) + 1);
    for (const std::string& lib : libs) {
        std::string text;
        if (!read_file((dir + lib).c_str(), text)) continue;  // "ERROR: Cannot open file", and on with the next one
        LineCursor cur(std::move(text));
        std::string line;
        ObjMtl* m = nullptr;
        auto after_blanks = [&](size_t from) {  // Buffer::Copy: skip control characters / blanks, keep the rest
            while (from < line.size() && (unsigned char)line[from] <= ' ') from++;
            return argument(line, from);
        };
        while (cur.next(line)) {
            if (is_command(line, "newmtl")) {
                m = nullptr;
                const std::string name = argument(line, 7);
                for (ObjMtl& k : mtls)
                    if (k.used_name == name) { m = &k; break; }
                if (m) m->name = after_blanks(7);
                continue;
            }
            if (!m) continue;
            float f3[3];
            auto colour = [&](float dst[3]) {  // ReadFloat3: one number means grey
                const int n = scan3(line, f3);
                if (n == 1) f3[1] = f3[2] = f3[0];
                dst[0] = f3[0]; dst[1] = f3[1]; dst[2] = f3[2];
            };
            if (is_command(line, "Ka")) colour(m->Ka);
            else if (is_command(line, "Kd")) colour(m->Kd);
            else if (is_command(line, "Ks")) colour(m->Ks);
            else if (is_command(line, "Tf")) colour(m->Tf);
            else if (is_command(line, "Ns")) { if (line.size() > 2) sscanf(line.c_str() + 2, "%f", &m->Ns); }
            else if (is_command(line, "Ni")) { if (line.size() > 2) sscanf(line.c_str() + 2, "%f", &m->Ni); }
            else if (is_command(line, "illum")) { if (line.size() > 5) sscanf(line.c_str() + 5, "%d", &m->illum); }
            else if (is_command(line, "map_Kd")) { m->map_Kd = after_blanks(7); m->has_map_Kd = true; }
            else if (is_command(line, "map_Ks")) { m->map_Ks = after_blanks(7); m->has_map_Ks = true; }
            // map_Ka / map_Ns / map_d / bump / disp: stored by the reference, read by nobody (xmlload.cpp:209-232)
        }
    }
}

}  // namespace

// cy::TriMesh::LoadFromFileObj (cyTriMesh.h:263-547): v / vt / vn / f, polygons fan-triangulated around
// their first vertex; with loadMtl also usemtl / mtllib: the faces are regrouped material by material
// (:469-491) and the materials the file uses are filled from its .mtl libraries.
bool LoadObjFile(const char* filename, bool loadMtl, MeshData& out, std::string& err, std::vector<ObjMtl>* mtls_out) {
    std::string text;
    if (!read_file(filename, text)) {
        err = std::string("ERROR: Cannot open file ") + filename;
        return false;
    }
    LineCursor cur(std::move(text));
    std::vector<float> v, vt, vn;
    std::vector<uint32_t> f, ft, fn;
    std::vector<int> face_mtl;          // per triangle: the material current when its line was read, -1 none
    std::vector<ObjMtl> mtls;
    std::vector<std::string> libs;
    int current_mtl = -1;
    bool with_vt = false, with_vn = false;  // sticky from the first vt / vn line or face field on (:378-432)
    std::string line;
    while (cur.next(line)) {
        float p[3];
        if (is_command(line, "v")) { scan3(line, p); v.insert(v.end(), p, p + 3); }
        else if (is_command(line, "vt")) { scan3(line, p); vt.insert(vt.end(), p, p + 3); with_vt = true; }
        else if (is_command(line, "vn")) { scan3(line, p); vn.insert(vn.end(), p, p + 3); with_vn = true; }
        else if (is_command(line, "f")) {
            // corners[0] stays, corners[1..2] slide along the polygon: triangle (0, k-1, k) for every k >= 2
            uint32_t corner[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // [corner][v, vt, vn]
            const size_t before = f.size() / 3;
            auto emit_triangle = [&]() {
                for (int c = 0; c < 3; c++) f.push_back(corner[c][0]);
                if (with_vt) for (int c = 0; c < 3; c++) ft.push_back(corner[c][1]);
                if (with_vn) for (int c = 0; c < 3; c++) fn.push_back(corner[c][2]);
                face_mtl.push_back(current_mtl);
            };
            int seen = 0;
            size_t i = 2;
            while (i < line.size()) {
                if (line[i] == ' ') { i++; continue; }
                size_t e = i;
                while (e < line.size() && line[e] != ' ') e++;
                int c = seen < 3 ? seen : 2;
                if (seen >= 3) {  // a further corner: the triangle so far is complete, its last corner becomes the middle one
                    emit_triangle();
                    for (int k = 0; k < 3; k++) corner[1][k] = corner[2][k];
                }
                seen++;
                const FaceRef r = parse_face_ref(line.data() + i, e - i);
                const uint32_t counts[3] = {(uint32_t)(v.size() / 3), (uint32_t)(vt.size() / 3), (uint32_t)(vn.size() / 3)};
                for (int k = 0; k < 3; k++) {
                    if (!r.has[k]) continue;
                    corner[c][k] = r.relative[k] ? counts[k] - r.value[k] : r.value[k] - 1u;  // unsigned wrap-around like the reference; checked below
                    if (k == 1) with_vt = true;
                    if (k == 2) with_vn = true;
                }
                i = e;
            }
            emit_triangle();  // (also for a line with fewer than three corners: missing ones are vertex 0, :388-391)
            if (current_mtl >= 0) mtls[(size_t)current_mtl].face_count += (uint32_t)(f.size() / 3 - before);
        } else if (loadMtl) {
            if (is_command(line, "usemtl")) {
                const std::string name = argument(line, 7);
                if (name.empty()) current_mtl = mtls.empty() ? -1 : 0;  // MtlList::CreateMtl("") answers 0
                else {
                    current_mtl = -1;
                    for (size_t k = 0; k < mtls.size(); k++)
                        if (mtls[k].used_name == name) current_mtl = (int)k;
                    if (current_mtl < 0) {
                        ObjMtl m;
                        m.used_name = name;
                        m.first_face = (uint32_t)(f.size() / 3);
                        mtls.push_back(m);
                        current_mtl = (int)mtls.size() - 1;
                    }
                }
            }
            if (is_command(line, "mtllib")) libs.push_back(argument(line, 7));
        }
    }
    out = MeshData();
    if (mtls_out) mtls_out->clear();
    if (f.empty()) return true;  // "No faces found" (:452): the mesh stays empty
    const size_t nf = f.size() / 3;
    // ft / fn exist when the file has texture vertices / normals (SetNumTexVerts / SetNumNormals, :458-459) AND every
    // face carries them — a face list shorter than f would be read past its end by the reference
    if (vt.empty() || ft.size() != f.size()) { ft.clear(); vt.clear(); }
    if (vn.empty() || fn.size() != f.size()) { fn.clear(); vn.clear(); }
    // The reference indexes its arrays with whatever the file says. Here a face that points outside the
    // vertex / texture-vertex / normal lists is an error before anything is computed from it.
    auto in_range = [&](const std::vector<uint32_t>& idx, size_t count, const char* what) {
        for (uint32_t i : idx)
            if (i >= count) {
                err = std::string("ERROR: ") + filename + ": face references a " + what + " that does not exist";
                return false;
            }
        return true;
    };
    if (!in_range(f, v.size() / 3, "vertex") || !in_range(ft, vt.size() / 3, "texture vertex") || !in_range(fn, vn.size() / 3, "normal")) return false;

    if (!mtls.empty()) {
        // :469-491: material by material, the faces read while it was current (searched from the first face that
        // used it); then the faces without material
        std::vector<uint32_t> order;
        order.reserve(nf);
        for (size_t m = 0; m < mtls.size(); m++) {
            uint32_t taken = 0;
            for (size_t i = mtls[m].first_face; taken < mtls[m].face_count && i < nf; i++)
                if (face_mtl[i] == (int)m) { order.push_back((uint32_t)i); taken++; }
            mtls[m].cumulative_face_count = (uint32_t)order.size();
        }
        if (order.size() < nf)
            for (size_t i = 0; i < nf; i++)
                if (face_mtl[i] < 0) order.push_back((uint32_t)i);
        auto regroup = [&](std::vector<uint32_t>& idx) {
            if (idx.empty()) return;
            std::vector<uint32_t> g(idx.size(), 0u);  // (a face no pass picked would be uninitialised in the reference)
            for (size_t k = 0; k < order.size(); k++)
                for (int c = 0; c < 3; c++) g[3 * k + c] = idx[3 * (size_t)order[k] + c];
            idx.swap(g);
        };
        regroup(f); regroup(ft); regroup(fn);
    }
    if (loadMtl) read_mtl_libraries(filename, libs, mtls);
    out.v.swap(v); out.f.swap(f); out.vt.swap(vt); out.ft.swap(ft); out.vn.swap(vn); out.fn.swap(fn);
    if (mtls_out) mtls_out->swap(mtls);
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

// ---------------------------------------------------------------------------------------------
// The reference's BVH (cy::BVH::Build, cyBVH.h:122-142 and 242-328, over BVHTriMesh :339-379).
// What has to come out the same: a node is split at the MIDPOINT of its box on the widest axis —
// elements whose centre (mean of the three vertices) is <= the midpoint go left; if that leaves a side
// empty the next-widest axes are tried, then the list is halved if it still holds more than 8; a list
// of <= maxPerNode elements is a leaf. The in-place partition decides the element order, node ids are
// handed out pair by pair in depth-first order (the root is 1, node 0 stays unused).
namespace {

struct Bounds {
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    void include(const Bounds& o) {
        for (int k = 0; k < 3; k++) {
            if (lo[k] > o.lo[k]) lo[k] = o.lo[k];
            if (hi[k] < o.hi[k]) hi[k] = o.hi[k];
        }
    }
};

// Elements failing `keeps_left` are exchanged with the last not-yet-classified one, so what arrives from the back is
// classified next (the permutation cyBVH.h:313-321 produces; std::partition promises no particular one).
template <class Pred>
uint32_t partition_from_back(uint32_t* el, uint32_t count, Pred keeps_left) {
    uint32_t left = 0, unclassified_end = count;
    while (left < unclassified_end) {
        if (keeps_left(el[left])) { left++; continue; }
        unclassified_end--;
        std::swap(el[left], el[unclassified_end]);
    }
    return left;
}

class TreeWriter {
public:
    TreeWriter(MeshData& mesh, unsigned max_per_node) : m_(mesh), max_per_node_(max_per_node) {}

    Bounds triangle_bounds(uint32_t face) const {  // GetElementBounds, :356-367
        Bounds b;
        for (int c = 0; c < 3; c++) {
            const float* p = &m_.v[3 * (size_t)m_.f[3 * (size_t)face + c]];
            for (int k = 0; k < 3; k++) {
                if (c == 0 || b.lo[k] > p[k]) b.lo[k] = p[k];
                if (c == 0 || b.hi[k] < p[k]) b.hi[k] = p[k];
            }
        }
        return b;
    }

    // Writes node `id` for elements [first, first + count) and everything below it.
    void write(uint32_t id, uint32_t first, uint32_t count, const Bounds& box, uint32_t level) {
        if (m_.bvh.size() <= id) m_.bvh.resize((size_t)id + 1, RtuBvhNode{});
        if (level > m_.bvh_depth) m_.bvh_depth = level;
        RtuBvhNode n{};
        for (int k = 0; k < 3; k++) { n.bmin[k] = box.lo[k]; n.bmax[k] = box.hi[k]; }
        uint32_t* el = &m_.elements[first];
        uint32_t left = split(el, count, box);
        if (left == 0 || left >= count) {
            if (count <= kHardLeafLimit) {  // a leaf
                n.index = first;
                n.count = count;
                m_.bvh[id] = n;
                return;
            }
            left = count / 2;  // nothing separates them: halve the list as it stands
        }
        Bounds lb, rb;
        for (uint32_t i = 0; i < left; i++) lb.include(triangle_bounds(el[i]));
        for (uint32_t i = left; i < count; i++) rb.include(triangle_bounds(el[i]));
        const uint32_t pair = next_id_;  // siblings are neighbours; the left subtree takes its ids before the right one
        next_id_ += 2;
        n.index = pair;
        n.count = 0;
        m_.bvh[id] = n;
        write(pair, first, left, lb, level + 1);
        write(pair + 1, first + left, count - left, rb, level + 1);
    }
    uint32_t nodes_used() const { return next_id_; }

private:
    static constexpr uint32_t kHardLeafLimit = 8;  // CY_BVH_MAX_ELEMENT_COUNT, cyBVH.h:44-48

    float centre(uint32_t face, int axis) const {  // GetElementCenter, :370-374
        const uint32_t* fv = &m_.f[3 * (size_t)face];
        return (m_.v[3 * (size_t)fv[0] + axis] + m_.v[3 * (size_t)fv[1] + axis] + m_.v[3 * (size_t)fv[2] + axis]) / 3.0f;
    }

    // MeanSplit (:290-326): how many elements go left, 0 when the node is (or has to be) a leaf
    uint32_t split(uint32_t* el, uint32_t count, const Bounds& box) const {
        if (count <= max_per_node_) return 0;
        const float ext[3] = {box.hi[0] - box.lo[0], box.hi[1] - box.lo[1], box.hi[2] - box.lo[2]};
        // widest axis first (x before y before z among equals), then the two others in cyclic order unless the
        // second of them is strictly wider
        int first = ext[0] >= ext[1] ? (ext[0] >= ext[2] ? 0 : 2) : (ext[1] >= ext[2] ? 1 : 2);
        int axes[3] = {first, (first + 1) % 3, (first + 2) % 3};
        if (ext[axes[1]] < ext[axes[2]]) std::swap(axes[1], axes[2]);
        for (int axis : axes) {
            const float mid = 0.5f * (box.lo[axis] + box.hi[axis]);
            const uint32_t left = partition_from_back(el, count, [&](uint32_t face) { return centre(face, axis) <= mid; });
            if (left > 0 && left < count) return left;
        }
        return 0;
    }

    MeshData& m_;
    unsigned  max_per_node_;
    uint32_t  next_id_ = 2;
};

}  // namespace

void BuildBVH(MeshData& m, unsigned maxElementsPerNode) {
    m.bvh.clear();
    m.elements.clear();
    m.bvh_depth = 0;
    const uint32_t n = (uint32_t)(m.f.size() / 3);
    if (n == 0) return;
    if (maxElementsPerNode > 8) maxElementsPerNode = 8;
    m.elements.resize(n);
    for (uint32_t i = 0; i < n; i++) m.elements[i] = i;
    TreeWriter w(m, maxElementsPerNode);
    Bounds all;
    for (uint32_t i = 0; i < n; i++) all.include(w.triangle_bounds(i));
    m.bvh.assign(2, RtuBvhNode{});  // node 0 is unused, the root is node 1 (cyBVH.h:76,199)
    w.write(1, 0, n, all, 1);
    m.bvh.resize(w.nodes_used(), RtuBvhNode{});
}

bool TriObj::Load(const char* filename, bool loadMtl) {
    if (!LoadObjFile(filename, loadMtl, data, error, &mtls)) return false;
    if (data.vn.empty()) ComputeNormals(data);   // objects.h:56
    ComputeBoundingBox(data);                    // objects.h:57
    BuildBVH(data, 4);                           // objects.h:58
    return true;
}

}  // namespace rtu
