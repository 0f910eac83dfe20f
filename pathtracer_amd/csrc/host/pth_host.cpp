// pth_host: C++ mirror of the reference's host-side scene ingest (include/pth_host.h).
// Every f32 expression follows the reference's Rust operation by operation; compile with
// -ffp-contract=off so no multiply-add is fused (Rust never fuses on its own).
#include "../../../include/pth_host.h"

#include <algorithm>
#include <cctype>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

namespace {

// ---- scene model (src/scene/material.rs:9-39, src/scene/shape.rs:104-176, src/scene/scene.rs:72-174) ----
struct Material {
    float base_color[3] = {0.8f, 0.8f, 0.8f};
    float metallic = 0.0f, roughness = 0.5f;
    float emission[3] = {0.0f, 0.0f, 0.0f};
    float emission_strength = 0.0f, ior = 1.5f, transmission = 0.0f;
    int32_t texture_id = -1;
    bool is_emissive() const {  // material.rs:97-100
        return emission_strength > 0.0f && (emission[0] > 0.0f || emission[1] > 0.0f || emission[2] > 0.0f);
    }
};
struct Shape {
    std::string name;  // empty = none
    uint32_t type = 0;
    bool negative = false;
    float position[3] = {0, 0, 0}, normal[3] = {0, 1, 0};
    float radius = 1.0f, radius2 = 0.0f, height = 0.0f;
    float rotation[3] = {0, 0, 0}, v0[3] = {0, 0, 0}, v1[3] = {0, 0, 0}, v2[3] = {0, 0, 0};
    float power = 8.0f;
    uint32_t max_iterations = 12;
    std::string texture;  // empty = none
    bool has_texture_scale = false;
    float texture_scale = 1.0f;
    float uv0[2] = {0, 0}, uv1[2] = {0, 0}, uv2[2] = {0, 0};
    Material material;
};
struct CameraConfig {
    float position[3] = {0, 0, 0}, rotation[3] = {0, 0, 0};
    float fov = 60.0f, exposure = 1.0f;
    uint32_t max_bounces = 16;
    float firefly_clamp = 100.0f;
    float skybox_color[3] = {0.5f, 0.7f, 1.0f};
    float skybox_brightness = 0.3f;
    uint32_t tone_mapper = 0, fractal_march_steps = 256;
};
struct ModelRef {
    std::string path;
    float position[3] = {0, 0, 0};
    float scale = 1.0f;
    Material material;
};

const char* TYPE_NAMES[17] = {"sphere", "plane", "cube", "cylinder", "cone", "torus", "disc", "triangle", "skybox",
                              "mandelbulb", "julia", "ellipsoid", "paraboloid", "hyperboloid", "mebius", "pyramid", "tetrahedron"};

struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 mk(float x, float y, float z) { return V3{x, y, z}; }
inline V3 mk(const float* p) { return V3{p[0], p[1], p[2]}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 vmin(V3 a, V3 b) { return {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }

struct Aabb { V3 mn, mx; };
const float INF = std::numeric_limits<float>::infinity();
const Aabb AABB_EMPTY = {{INF, INF, INF}, {-INF, -INF, -INF}};
inline Aabb aunion(Aabb a, Aabb b) { return {vmin(a.mn, b.mn), vmax(a.mx, b.mx)}; }
inline float surface_area(const Aabb& a) {  // aabb.rs:46-49
    V3 d = a.mx - a.mn;
    return 2.0f * (d.x * d.y + d.y * d.z + d.z * d.x);
}
inline V3 center(const Aabb& a) { return (a.mn + a.mx) * 0.5f; }
const float AABB_EPS = 0.0001f;  // constants.rs:14
Aabb pad(Aabb a) {               // aabb.rs:69-81
    for (int ax = 0; ax < 3; ax++)
        if (a.mx[ax] - a.mn[ax] < AABB_EPS) {
            a.mn.at(ax) -= AABB_EPS;
            a.mx.at(ax) += AABB_EPS;
        }
    return a;
}
Aabb shape_aabb(const Shape& s) {  // aabb.rs:105-174
    V3 pos = mk(s.position);
    float r = s.radius, r2 = s.radius2, h = s.height;
    switch (s.type) {
        case 0: case 2: case 16: { V3 e = mk(r, r, r); return {pos - e, pos + e}; }
        case 3: { V3 e = mk(r, h * 0.5f, r); return {pos - e, pos + e}; }
        case 4: case 12: case 15: return {pos - mk(r, 0.0f, r), pos + mk(r, h, r)};
        case 5: { float ext = r + r2; V3 e = mk(ext, r2, ext); return {pos - e, pos + e}; }
        case 6: { V3 e = mk(r, r, r); return pad({pos - e, pos + e}); }
        case 7: {
            V3 a = mk(s.v0), b = mk(s.v1), c = mk(s.v2);
            return pad({vmin(vmin(a, b), c), vmax(vmax(a, b), c)});
        }
        case 9: case 10: case 14: { float k = r * 1.5f; V3 e = mk(k, k, k); return {pos - e, pos + e}; }
        case 11: { V3 e = mk(r, std::fmax(h, r), std::fmax(r2, r)); return {pos - e, pos + e}; }
        case 13: { float hh = h * 0.5f; V3 e = mk(r + hh, hh, r + hh); return {pos - e, pos + e}; }
        default: { V3 big = mk(1e6f, 1e6f, 1e6f); return {-big, big}; }
    }
}

// ---- SAH BVH (src/accel/bvh.rs) -----------------------------------------------------------------------
const int NUM_BINS = 12, LEAF_MAX = 4;  // constants.rs:10-11
struct BuildNode { Aabb bounds; int64_t left, right; size_t first, count; };
struct BvhBuilder {
    const std::vector<Aabb>& aabbs;
    std::vector<V3> cents;
    std::vector<uint32_t> indices;
    std::vector<BuildNode> nodes;
    explicit BvhBuilder(const std::vector<Aabb>& a) : aabbs(a) {
        cents.resize(a.size());
        indices.resize(a.size());
        for (size_t i = 0; i < a.size(); i++) { cents[i] = center(a[i]); indices[i] = (uint32_t)i; }
        nodes.reserve(2 * a.size());
    }
    void find_best_split(size_t start, size_t end, const Aabb& pb, int& best_axis, float& best_split) {  // bvh.rs:111-172
        float best_cost = INF;
        best_axis = 0;
        best_split = 0.0f;
        for (int axis = 0; axis < 3; axis++) {
            float mn = pb.mn[axis], mx = pb.mx[axis];
            float extent = mx - mn;
            if (std::fabs(extent) < 1e-8f) continue;
            Aabb bb[NUM_BINS];
            uint32_t bc[NUM_BINS];
            for (int b = 0; b < NUM_BINS; b++) { bb[b] = AABB_EMPTY; bc[b] = 0; }
            float inv_extent = (float)NUM_BINS / extent;
            for (size_t k = start; k < end; k++) {
                uint32_t idx = indices[k];
                float v = (cents[idx][axis] - mn) * inv_extent;
                int b = !(v > 0.0f) ? 0 : (v >= (float)NUM_BINS ? NUM_BINS - 1 : (int)v);  // saturating `as usize`, then min
                bb[b] = aunion(bb[b], aabbs[idx]);
                bc[b] += 1;
            }
            float right_area[NUM_BINS - 1];
            uint32_t right_count[NUM_BINS - 1];
            Aabb rb = AABB_EMPTY;
            uint32_t rc = 0;
            for (int i = NUM_BINS - 1; i >= 1; i--) {
                rb = aunion(rb, bb[i]);
                rc += bc[i];
                right_area[i - 1] = surface_area(rb);
                right_count[i - 1] = rc;
            }
            Aabb lb = AABB_EMPTY;
            uint32_t lc = 0;
            float bin_width = extent / (float)NUM_BINS;
            for (int i = 0; i < NUM_BINS - 1; i++) {
                lb = aunion(lb, bb[i]);
                lc += bc[i];
                if (lc == 0 || right_count[i] == 0) continue;
                float cost = (float)lc * surface_area(lb) + (float)right_count[i] * right_area[i];
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    best_split = mn + (float)(i + 1) * bin_width;
                }
            }
        }
    }
    size_t partition(size_t start, size_t end, int axis, float split) {  // bvh.rs:174-186
        size_t lo = start, hi = end;
        while (lo < hi) {
            if (cents[indices[lo]][axis] < split) lo++;
            else { hi--; std::swap(indices[lo], indices[hi]); }
        }
        return lo;
    }
    // bvh.rs:59-109, with an explicit work stack instead of recursion (10 M primitives); nodes are
    // numbered in the same pre-order.
    void build() {
        struct Job { size_t start, end; int64_t parent; bool is_right; };
        std::vector<Job> jobs;
        jobs.push_back({0, aabbs.size(), -1, false});
        while (!jobs.empty()) {
            Job j = jobs.back();
            jobs.pop_back();
            Aabb bounds = AABB_EMPTY;
            for (size_t k = j.start; k < j.end; k++) bounds = aunion(bounds, aabbs[indices[k]]);
            int64_t ni = (int64_t)nodes.size();
            if (j.parent >= 0) { if (j.is_right) nodes[j.parent].right = ni; else nodes[j.parent].left = ni; }
            size_t count = j.end - j.start;
            if (count <= (size_t)LEAF_MAX) {
                nodes.push_back({bounds, -1, -1, j.start, count});
                continue;
            }
            int axis;
            float split;
            find_best_split(j.start, j.end, bounds, axis, split);
            size_t raw_mid = partition(j.start, j.end, axis, split);
            size_t mid = (raw_mid == j.start || raw_mid == j.end) ? (j.start + j.end) / 2 : raw_mid;
            nodes.push_back({bounds, -1, -1, 0, 0});
            jobs.push_back({mid, j.end, ni, true});     // right is built after the whole left subtree
            jobs.push_back({j.start, mid, ni, false});
        }
    }
};

// ---- YAML subset ---------------------------------------------------------------------------------------
struct YNode {
    enum Kind { Null, Scalar, Seq, Map } kind = Null;
    std::string s;
    std::vector<YNode> seq;
    std::vector<std::pair<std::string, YNode>> map;
    const YNode* get(const std::string& k) const {
        for (auto& kv : map) if (kv.first == k) return &kv.second;
        return nullptr;
    }
};
struct YLine { int indent; std::string text; };
std::string trim(const std::string& s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? "" : s.substr(a, b - a + 1);
}
std::string unquote(const std::string& s) {
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
    return s;
}
YNode parse_value(const std::string& v) {
    YNode n;
    std::string t = trim(v);
    if (t.empty() || t == "~" || t == "null") return n;
    if (t == "{}") { n.kind = YNode::Map; return n; }  // an empty flow mapping: what serde writes for a struct with every field skipped
    if (t.front() == '[') {
        n.kind = YNode::Seq;
        size_t e = t.rfind(']');
        std::string inner = t.substr(1, e == std::string::npos ? std::string::npos : e - 1);
        std::stringstream ss(inner);
        std::string item;
        while (std::getline(ss, item, ',')) {
            std::string it = trim(item);
            if (!it.empty()) n.seq.push_back(parse_value(it));
        }
        return n;
    }
    n.kind = YNode::Scalar;
    n.s = unquote(t);
    return n;
}
// Position of the ':' that separates key and value (followed by space or end), or npos.
size_t key_colon(const std::string& t) {
    if (t.empty() || t.front() == '[' || t.front() == '"' || t.front() == '\'') return std::string::npos;
    for (size_t i = 0; i < t.size(); i++)
        if (t[i] == ':' && (i + 1 == t.size() || t[i + 1] == ' ')) return i;
    return std::string::npos;
}
YNode parse_block(std::vector<YLine>& L, size_t& i, int indent) {
    YNode n;
    if (i >= L.size()) return n;
    bool is_seq = L[i].text[0] == '-' && (L[i].text.size() == 1 || L[i].text[1] == ' ');
    if (is_seq) {
        n.kind = YNode::Seq;
        while (i < L.size() && L[i].indent == indent && L[i].text[0] == '-' && (L[i].text.size() == 1 || L[i].text[1] == ' ')) {
            std::string rest = L[i].text.size() > 1 ? L[i].text.substr(2) : "";
            size_t lead = rest.find_first_not_of(' ');
            std::string r = trim(rest);
            if (r.empty()) {
                i++;
                if (i < L.size() && L[i].indent > indent) n.seq.push_back(parse_block(L, i, L[i].indent));
                else n.seq.push_back(YNode());
            } else if (key_colon(r) != std::string::npos) {
                int item_indent = indent + 2 + (int)(lead == std::string::npos ? 0 : lead);
                L[i].indent = item_indent;
                L[i].text = r;
                n.seq.push_back(parse_block(L, i, item_indent));
            } else {
                n.seq.push_back(parse_value(r));
                i++;
            }
        }
    } else {
        n.kind = YNode::Map;
        while (i < L.size() && L[i].indent == indent && !(L[i].text[0] == '-' && (L[i].text.size() == 1 || L[i].text[1] == ' '))) {
            size_t c = key_colon(L[i].text);
            if (c == std::string::npos) { i++; continue; }
            std::string key = unquote(trim(L[i].text.substr(0, c)));
            std::string val = trim(L[i].text.substr(c + 1));
            if (val.empty()) {
                i++;
                if (i < L.size() && (L[i].indent > indent || (L[i].indent == indent && L[i].text[0] == '-')))
                    n.map.push_back({key, parse_block(L, i, L[i].indent)});
                else
                    n.map.push_back({key, YNode()});
            } else {
                n.map.push_back({key, parse_value(val)});
                i++;
            }
        }
    }
    return n;
}
bool parse_yaml_file(const std::string& path, YNode& out, std::string& err) {
    std::ifstream f(path);
    if (!f) { err = "cannot read " + path; return false; }
    std::vector<YLine> lines;
    std::string raw;
    while (std::getline(f, raw)) {
        bool q1 = false, q2 = false;
        size_t cut = std::string::npos;
        for (size_t i = 0; i < raw.size(); i++) {
            if (raw[i] == '"' && !q1) q2 = !q2;
            else if (raw[i] == '\'' && !q2) q1 = !q1;
            else if (raw[i] == '#' && !q1 && !q2 && (i == 0 || raw[i - 1] == ' ' || raw[i - 1] == '\t')) { cut = i; break; }
        }
        if (cut != std::string::npos) raw = raw.substr(0, cut);
        size_t a = raw.find_first_not_of(' ');
        if (a == std::string::npos) continue;
        std::string t = trim(raw);
        if (t.empty() || t == "---") continue;
        lines.push_back({(int)a, t});
    }
    size_t i = 0;
    out = lines.empty() ? YNode() : parse_block(lines, i, lines[0].indent);
    return true;
}
// serde: YAML number -> f64 -> `as f32`.
bool y_f32(const YNode* n, float& v) {
    if (!n || n->kind != YNode::Scalar) return false;
    if (n->s == ".inf" || n->s == "+.inf") { v = INF; return true; }   // YAML 1.2 core schema spellings
    if (n->s == "-.inf") { v = -INF; return true; }
    if (n->s == ".nan") { v = std::numeric_limits<float>::quiet_NaN(); return true; }
    char* e = nullptr;
    double d = std::strtod(n->s.c_str(), &e);
    if (e == n->s.c_str()) return false;
    v = (float)d;
    return true;
}
bool y_u32(const YNode* n, uint32_t& v) { float f; if (!y_f32(n, f)) return false; v = (uint32_t)f; return true; }
bool y_vec(const YNode* n, float* v, int k) {
    if (!n || n->kind != YNode::Seq || (int)n->seq.size() < k) return false;
    for (int i = 0; i < k; i++) if (!y_f32(&n->seq[i], v[i])) return false;
    return true;
}
void read_material(const YNode* n, Material& m) {
    if (!n || n->kind != YNode::Map) return;
    y_vec(n->get("base_color"), m.base_color, 3);
    y_f32(n->get("metallic"), m.metallic);
    y_f32(n->get("roughness"), m.roughness);
    y_vec(n->get("emission"), m.emission, 3);
    y_f32(n->get("emission_strength"), m.emission_strength);
    y_f32(n->get("ior"), m.ior);
    y_f32(n->get("transmission"), m.transmission);
    float t;
    if (y_f32(n->get("texture_id"), t)) m.texture_id = (int32_t)t;
}
bool file_exists(const std::string& p) { std::ifstream f(p); return (bool)f; }
std::string dir_of(const std::string& p) { size_t s = p.find_last_of('/'); return s == std::string::npos ? "." : p.substr(0, s); }
// constants.rs:91-108 with `root` standing for the working directory.
std::string resolve_path(const std::string& scene_dir, const std::string& rel, const std::string& root) {
    if (!root.empty() && file_exists(root + "/" + rel)) return root + "/" + rel;
    if (file_exists(rel)) return rel;
    if (file_exists(scene_dir + "/" + rel)) return scene_dir + "/" + rel;
    return rel;
}

// ---- OBJ (src/model/obj_loader.rs:48-255 over tobj 4.0.3) ----------------------------------------------------
// tobj is a Cargo.lock pin, not part of the reference checkout: `load_obj(path, &GPU_LOAD_OPTIONS)` (triangulate,
// single_index) is restated from its published parser.  What the reference's scenes depend on: a numeric token that
// Rust's f32::from_str / isize::from_str rejects fails the whole load, and the reference then logs the error and
// renders the scene WITHOUT that model (app/state.rs:122-140; resources/models/aelita/aelita.obj has such a token);
// `vt` reads two values and ignores a third; a mesh ends at `o` / `g` and where `usemtl` changes the material;
// vertices are merged per (v, vt, vn) triple in first-seen order and a mesh's texcoords hold only the vertices that
// have one; a failed MTL file leaves the model without any materials.
struct ObjMtl { bool has_kd = false, has_ka = false, has_ks = false, has_ns = false, has_d = false, has_ni = false;
                float kd[3], ka[3], ks[3], ns, d, ni; std::string map_kd; };
// f32::from_str: [sign] (inf | infinity | nan | digits[.digits] | .digits) [e [sign] digits], nothing before or after;
// strtof rounds the decimal value once to binary32, as Rust does.
bool rust_f32(const std::string& t, float& out) {
    size_t i = 0, n = t.size();
    if (i < n && (t[i] == '+' || t[i] == '-')) i++;
    std::string body;
    for (size_t k = i; k < n; k++) body += (char)std::tolower((unsigned char)t[k]);
    if (body == "inf" || body == "infinity" || body == "nan") { out = std::strtof(t.c_str(), nullptr); return true; }
    size_t d0 = i;
    while (i < n && std::isdigit((unsigned char)t[i])) i++;
    size_t int_digits = i - d0, frac_digits = 0;
    if (i < n && t[i] == '.') {
        i++;
        size_t f0 = i;
        while (i < n && std::isdigit((unsigned char)t[i])) i++;
        frac_digits = i - f0;
    }
    if (int_digits + frac_digits == 0) return false;
    if (i < n && (t[i] == 'e' || t[i] == 'E')) {
        i++;
        if (i < n && (t[i] == '+' || t[i] == '-')) i++;
        size_t e0 = i;
        while (i < n && std::isdigit((unsigned char)t[i])) i++;
        if (i == e0) return false;
    }
    if (i != n) return false;
    out = std::strtof(t.c_str(), nullptr);
    return true;
}
bool rust_isize(const std::string& t, long long& out) {
    size_t i = 0, n = t.size();
    if (i < n && (t[i] == '+' || t[i] == '-')) i++;
    if (i == n) return false;
    for (size_t k = i; k < n; k++) if (!std::isdigit((unsigned char)t[k])) return false;
    out = std::strtoll(t.c_str(), nullptr, 10);
    return true;
}
std::vector<std::string> split_ws(const std::string& line) {
    std::stringstream ss(line);
    std::vector<std::string> t;
    std::string w;
    while (ss >> w) t.push_back(w);
    return t;
}
// tobj parse_floatn: the first n words after the keyword, all valid, exactly n of them found.
bool floatn(const std::vector<std::string>& t, size_t first, size_t n, std::vector<float>& dst) {
    std::vector<float> v;
    for (size_t k = first; k < t.size() && v.size() < n; k++) {
        float x;
        if (!rust_f32(t[k], x)) return false;
        v.push_back(x);
    }
    if (v.size() != n) return false;
    dst.insert(dst.end(), v.begin(), v.end());
    return true;
}
std::string after_keyword(const std::string& line) {  // the rest of the line after its first space, trimmed (names may hold spaces)
    std::string s = trim(line);
    size_t sp = s.find(' ');
    return sp == std::string::npos ? std::string() : trim(s.substr(sp + 1));
}
bool valid_utf8(const std::string& s) {
    size_t i = 0, n = s.size();
    while (i < n) {
        unsigned char c = (unsigned char)s[i];
        size_t len = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 0;
        if (len == 0 || i + len > n) return false;
        for (size_t k = 1; k < len; k++) if (((unsigned char)s[i + k] >> 6) != 2) return false;
        i += len;
    }
    return true;
}
// tobj load_mtl.  false where tobj returns an error (the caller then drops ALL materials of the model).
bool parse_mtl(const std::string& path, std::vector<ObjMtl>& mats, std::map<std::string, size_t>& index) {
    std::ifstream f(path);
    if (!f) return false;
    std::string line;
    ObjMtl* cur = nullptr;
    size_t base = mats.size();
    std::vector<ObjMtl> local;
    std::map<std::string, size_t> local_index;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (!valid_utf8(line)) return false;
        std::vector<std::string> t = split_ws(line);
        if (t.empty() || t[0] == "#") continue;
        std::vector<float> v;
        if (t[0] == "newmtl") {
            std::string name = after_keyword(line);
            if (name.empty()) return false;
            local_index[name] = base + local.size();
            local.push_back(ObjMtl());
            cur = &local.back();
        } else if (t[0] == "Kd" || t[0] == "Ka" || t[0] == "Ks") {
            if (!floatn(t, 1, 3, v)) return false;
            if (cur) {
                if (t[0] == "Kd") { cur->has_kd = true; memcpy(cur->kd, v.data(), 12); }
                else if (t[0] == "Ka") { cur->has_ka = true; memcpy(cur->ka, v.data(), 12); }
                else { cur->has_ks = true; memcpy(cur->ks, v.data(), 12); }
            }
        } else if (t[0] == "Ns" || t[0] == "Ni" || t[0] == "d") {
            if (!floatn(t, 1, 1, v)) return false;
            if (cur) {
                if (t[0] == "Ns") { cur->has_ns = true; cur->ns = v[0]; }
                else if (t[0] == "Ni") { cur->has_ni = true; cur->ni = v[0]; }
                else { cur->has_d = true; cur->d = v[0]; }
            }
        } else if (t[0] == "map_Kd") {
            std::string tex = after_keyword(line);
            if (tex.empty()) return false;
            if (cur) cur->map_kd = tex;
        }
    }
    mats.insert(mats.end(), local.begin(), local.end());
    for (auto& kv : local_index) index[kv.first] = kv.second;
    return true;
}
Material obj_material_to_pbr(const ObjMtl& o, const Material& fallback) {  // obj_loader.rs:178-214
    Material m = fallback;
    if (o.has_kd) memcpy(m.base_color, o.kd, 12);
    else if (o.has_ka) memcpy(m.base_color, o.ka, 12);
    if (o.has_ks) { float inten = (o.ks[0] + o.ks[1] + o.ks[2]) / 3.0f; m.metallic = std::fmin(std::fmax(inten, 0.0f), 1.0f); }
    if (o.has_ns) { float r = 1.0f - std::sqrt(o.ns / 1000.0f); m.roughness = std::fmin(std::fmax(r, 0.04f), 1.0f); }
    if (o.has_d && o.d < 1.0f) m.transmission = 1.0f - o.d;
    if (o.has_ni && o.ni > 0.0f) m.ior = o.ni;
    return m;
}
struct ObjVert { long long v, vt, vn; bool operator<(const ObjVert& o) const { return std::tie(v, vt, vn) < std::tie(o.v, o.vt, o.vn); } };
struct ObjMesh { std::vector<float> positions, texcoords; std::vector<uint32_t> indices; bool has_mat = false; size_t mat = 0; };
const long long OBJ_ABSENT = std::numeric_limits<long long>::min();
// tobj::load_obj with GPU_LOAD_OPTIONS.  false where it returns Err.
bool tobj_load(const std::string& path, std::vector<ObjMesh>& meshes, std::vector<ObjMtl>& materials, bool& mtl_ok, std::string& err) {
    std::ifstream f(path);
    if (!f) { err = "Failed to load OBJ: " + path; return false; }
    std::vector<float> pos, col, tex, nrm;
    std::vector<std::vector<ObjVert>> faces;
    std::map<std::string, size_t> mat_index;
    bool have_mat = false;
    size_t mat_id = 0;
    mtl_ok = true;
    auto fail_ = [&](const char* what) { err = std::string("Failed to load OBJ: ") + path + ": " + what; return false; };
    auto export_mesh = [&]() -> bool {
        ObjMesh m;
        m.has_mat = have_mat; m.mat = mat_id;
        std::map<ObjVert, uint32_t> index_map;
        auto add = [&](const ObjVert& vert) -> bool {
            auto it = index_map.find(vert);
            if (it != index_map.end()) { m.indices.push_back(it->second); return true; }
            if (vert.v == OBJ_ABSENT || vert.v < 0 || (size_t)vert.v * 3 + 2 >= pos.size()) return false;
            m.positions.insert(m.positions.end(), pos.begin() + vert.v * 3, pos.begin() + vert.v * 3 + 3);
            if (!tex.empty() && vert.vt != OBJ_ABSENT) {
                if (vert.vt < 0 || (size_t)vert.vt * 2 + 1 >= tex.size()) return false;
                m.texcoords.insert(m.texcoords.end(), tex.begin() + vert.vt * 2, tex.begin() + vert.vt * 2 + 2);
            }
            if (!nrm.empty() && vert.vn != OBJ_ABSENT && (vert.vn < 0 || (size_t)vert.vn * 3 + 2 >= nrm.size())) return false;
            if (!col.empty() && (size_t)vert.v * 3 + 2 >= col.size()) return false;
            uint32_t i = (uint32_t)index_map.size();
            index_map[vert] = i;
            m.indices.push_back(i);
            return true;
        };
        for (auto& fc : faces) {
            if (fc.size() == 1) { if (!add(fc[0]) || !add(fc[0]) || !add(fc[0])) return false; }
            else if (fc.size() == 2) { if (!add(fc[0]) || !add(fc[1]) || !add(fc[1])) return false; }
            else for (size_t k = 1; k + 1 < fc.size(); k++) if (!add(fc[0]) || !add(fc[k]) || !add(fc[k + 1])) return false;
        }
        faces.clear();
        meshes.push_back(std::move(m));
        return true;
    };
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (!valid_utf8(line)) return fail_("not UTF-8");
        std::vector<std::string> t = split_ws(line);
        if (t.empty() || t[0] == "#") continue;
        if (t[0] == "v") {
            if (!floatn(t, 1, 3, pos)) return fail_("vertex position");
            for (size_t k = 4; k < t.size() && k < 7; k++) {  // vertex colours, kept as far as they parse
                float c;
                if (!rust_f32(t[k], c)) break;
                col.push_back(c);
            }
        } else if (t[0] == "vt") {
            if (!floatn(t, 1, 2, tex)) return fail_("texture coordinate");
        } else if (t[0] == "vn") {
            if (!floatn(t, 1, 3, nrm)) return fail_("normal");
        } else if (t[0] == "f" || t[0] == "l") {
            const long long sizes[3] = {(long long)(pos.size() / 3), (long long)(tex.size() / 2), (long long)(nrm.size() / 3)};
            std::vector<ObjVert> face;
            for (size_t k = 1; k < t.size(); k++) {
                long long part[3] = {OBJ_ABSENT, OBJ_ABSENT, OBJ_ABSENT};
                size_t start = 0, i = 0;
                for (;;) {
                    size_t slash = t[k].find('/', start);
                    std::string p = t[k].substr(start, slash == std::string::npos ? std::string::npos : slash - start);
                    if (i >= 3) return fail_("face");
                    if (!p.empty()) {
                        long long x;
                        if (!rust_isize(p, x)) return fail_("face index");
                        part[i] = x < 0 ? sizes[i] + x : x - 1;  // 0 -> -1: out of bounds when used
                    }
                    i++;
                    if (slash == std::string::npos) break;
                    start = slash + 1;
                }
                face.push_back({part[0], part[1], part[2]});
            }
            faces.push_back(face);
        } else if (t[0] == "o" || t[0] == "g") {
            if (!faces.empty() && !export_mesh()) return fail_("face vertex out of bounds");
        } else if (t[0] == "mtllib") {
            if (!parse_mtl(dir_of(path) + "/" + after_keyword(line), materials, mat_index)) mtl_ok = false;
        } else if (t[0] == "usemtl") {
            std::string name = after_keyword(line);
            if (name.empty()) return fail_("usemtl without a name");
            auto it = mat_index.find(name);
            bool new_have = it != mat_index.end();
            size_t new_id = new_have ? it->second : 0;
            if ((new_have != have_mat || (new_have && new_id != mat_id)) && !faces.empty() && !export_mesh()) return fail_("face vertex out of bounds");
            have_mat = new_have; mat_id = new_id;
        }
    }
    if (!export_mesh()) return fail_("face vertex out of bounds");
    return true;
}
bool load_obj(const std::string& path, const float* position, float scale, const Material& default_material,
              std::vector<Shape>& out, std::string& err) {
    std::vector<ObjMesh> meshes;
    std::vector<ObjMtl> materials;
    bool mtl_ok = true;
    if (!tobj_load(path, meshes, materials, mtl_ok, err)) return false;
    if (!mtl_ok) materials.clear();  // resolve_materials, obj_loader.rs:61-78
    auto rv = [&](const ObjMesh& m, uint32_t i) { return mk(m.positions[i * 3] * scale, m.positions[i * 3 + 1] * scale, m.positions[i * 3 + 2] * scale); };  // obj_loader.rs:239-245
    float fmax_ = std::numeric_limits<float>::max();
    V3 bb_min = mk(fmax_, fmax_, fmax_), bb_max = mk(-fmax_, -fmax_, -fmax_);
    for (auto& m : meshes)
        for (uint32_t i : m.indices) { V3 v = rv(m, i); bb_min = vmin(bb_min, v); bb_max = vmax(bb_max, v); }
    V3 ctr = (bb_min + bb_max) * 0.5f;
    V3 offset = mk(position) - ctr;
    for (auto& m : meshes) {
        Material mat = default_material;
        std::string tex;
        if (m.has_mat && m.mat < materials.size()) {
            mat = obj_material_to_pbr(materials[m.mat], default_material);
            tex = materials[m.mat].map_kd;
            if (!tex.empty() && !file_exists(tex) && file_exists(dir_of(path) + "/" + tex)) tex = dir_of(path) + "/" + tex;  // resolve_texture_path, obj_loader.rs:216-237
        }
        const bool has_uv = !m.texcoords.empty();
        for (size_t k = 0; k + 2 < m.indices.size(); k += 3) {
            Shape s;
            s.type = 7;
            s.radius = 0.0f;
            s.power = 0.0f;
            s.max_iterations = 0;
            float* vs[3] = {s.v0, s.v1, s.v2};
            float* us[3] = {s.uv0, s.uv1, s.uv2};
            for (int j = 0; j < 3; j++) {
                const uint32_t i = m.indices[k + j];
                V3 v = rv(m, i) + offset;
                vs[j][0] = v.x; vs[j][1] = v.y; vs[j][2] = v.z;
                const size_t base = (size_t)i * 2;  // read_uv, obj_loader.rs:247-255: by the merged vertex's index, V flipped
                if (has_uv && base + 1 < m.texcoords.size()) { us[j][0] = m.texcoords[base]; us[j][1] = 1.0f - m.texcoords[base + 1]; }
            }
            s.texture = tex;
            s.material = mat;
            out.push_back(s);
        }
    }
    return true;
}

// ---- glam 0.29.3 scalar quaternion helpers (Cargo.lock pin; not in the reference checkout) ------------------
struct Q { float x, y, z, w; };
Q qmul(Q a, Q b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
            a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
V3 qrot(Q q, V3 v) {
    V3 b = mk(q.x, q.y, q.z);
    float b2 = b.x * b.x + b.y * b.y + b.z * b.z;
    float s1 = q.w * q.w - b2;
    float vb = (v.x * b.x + v.y * b.y + v.z * b.z) * 2.0f;
    V3 cr = mk(b.y * v.z - v.y * b.z, b.z * v.x - v.z * b.x, b.x * v.y - v.x * b.y);
    float w2 = q.w * 2.0f;
    return mk(v.x * s1 + b.x * vb + cr.x * w2, v.y * s1 + b.y * vb + cr.y * w2, v.z * s1 + b.z * vb + cr.z * w2);
}

uint32_t pcg(uint32_t input) {
    uint32_t state = input * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}

}  // namespace

struct pth_scene {
    CameraConfig camera;
    std::vector<Shape> shapes;
    std::vector<std::string> tex_paths;  // first-seen order
    struct Tex { bool provided = false; uint32_t w = 0, h = 0; std::vector<uint32_t> px; };
    std::vector<Tex> tex;
    bool built = false;
    std::vector<ptx_shape> figs;
    std::vector<ptx_material> mats;
    std::vector<ptx_bvh_node> nodes;
    std::vector<uint32_t> prims, lights, infinite, tex_pixels;
    std::vector<ptx_texture_info> tex_infos;
    std::string err;
};

namespace {
void collect_textures(pth_scene* s) {
    std::vector<std::string> paths;
    for (auto& sh : s->shapes)
        if (!sh.texture.empty() && std::find(paths.begin(), paths.end(), sh.texture) == paths.end()) paths.push_back(sh.texture);
    if (paths != s->tex_paths) {
        std::vector<pth_scene::Tex> nt(paths.size());
        for (size_t i = 0; i < paths.size(); i++)
            for (size_t j = 0; j < s->tex_paths.size(); j++)
                if (s->tex_paths[j] == paths[i]) nt[i] = s->tex[j];
        s->tex_paths = paths;
        s->tex = nt;
    }
}
}  // namespace

extern "C" {

uint16_t pth_f32_to_f16_bits(float val) {  // shape.rs:290-322
    uint32_t bits;
    memcpy(&bits, &val, 4);
    uint32_t sign = (bits >> 16) & 0x8000u;
    int exponent = (int)((bits >> 23) & 0xFFu);
    uint32_t mantissa = bits & 0x007FFFFFu;
    if (exponent == 0) return (uint16_t)sign;
    if (exponent == 0xFF) return (uint16_t)(mantissa != 0 ? (sign | 0x7E00u) : (sign | 0x7C00u));
    int new_exp = exponent - 127 + 15;
    if (new_exp >= 31) return (uint16_t)(sign | 0x7C00u);
    if (new_exp <= 0) {
        if (new_exp < -10) return (uint16_t)sign;
        uint32_t m = (mantissa | 0x00800000u) >> (1 - new_exp + 13);
        return (uint16_t)(sign | m);
    }
    return (uint16_t)(sign | ((uint32_t)new_exp << 10) | (mantissa >> 13));
}

pth_scene* pth_scene_new(void) {
    pth_scene* s = new pth_scene();
    float p[3] = {0.0f, 2.0f, -10.0f};  // constants.rs:20
    memcpy(s->camera.position, p, 12);
    return s;
}
void pth_scene_free(pth_scene* s) { delete s; }
const char* pth_last_error(const pth_scene* s) { return s ? s->err.c_str() : "null scene"; }

pth_scene* pth_scene_load(const char* yaml_path, const char* resource_root, char* err, size_t err_len) {
    auto bail = [&](const std::string& m) -> pth_scene* {
        if (err && err_len) snprintf(err, err_len, "%s", m.c_str());
        return nullptr;
    };
    if (!yaml_path) return bail("null path");
    YNode doc;
    std::string e;
    if (!parse_yaml_file(yaml_path, doc, e)) return bail(e);
    pth_scene* s = pth_scene_new();
    std::string root = resource_root ? resource_root : "";
    std::string scene_dir = dir_of(yaml_path);
    if (const YNode* c = doc.get("camera"); c && c->kind == YNode::Map) {
        CameraConfig cam;  // a present `camera:` map starts from the per-field serde defaults (position 0)
        y_vec(c->get("position"), cam.position, 3);
        y_vec(c->get("rotation"), cam.rotation, 3);
        y_f32(c->get("fov"), cam.fov);
        y_f32(c->get("exposure"), cam.exposure);
        y_u32(c->get("max_bounces"), cam.max_bounces);
        y_f32(c->get("firefly_clamp"), cam.firefly_clamp);
        y_vec(c->get("skybox_color"), cam.skybox_color, 3);
        y_f32(c->get("skybox_brightness"), cam.skybox_brightness);
        y_u32(c->get("tone_mapper"), cam.tone_mapper);
        y_u32(c->get("fractal_march_steps"), cam.fractal_march_steps);
        s->camera = cam;
    }
    const YNode* shapes = doc.get("shapes");
    if (!shapes) shapes = doc.get("figures");  // serde alias, scene.rs:169
    if (shapes && shapes->kind == YNode::Seq) {
        for (auto& n : shapes->seq) {
            if (n.kind != YNode::Map) continue;
            Shape sh;
            const YNode* ty = n.get("type");
            int tid = -1;
            if (ty) for (int i = 0; i < 17; i++) if (ty->s == TYPE_NAMES[i]) tid = i;
            if (tid < 0) { pth_scene_free(s); return bail("unknown or missing shape type '" + (ty ? ty->s : "") + "'"); }
            sh.type = (uint32_t)tid;
            if (const YNode* nm = n.get("name"); nm && nm->kind == YNode::Scalar) sh.name = nm->s;
            if (const YNode* ng = n.get("negative")) sh.negative = ng->s == "true";
            y_vec(n.get("position"), sh.position, 3);
            y_vec(n.get("normal"), sh.normal, 3);
            y_f32(n.get("radius"), sh.radius);
            y_f32(n.get("radius2"), sh.radius2);
            y_f32(n.get("height"), sh.height);
            y_vec(n.get("rotation"), sh.rotation, 3);
            y_vec(n.get("v0"), sh.v0, 3);
            y_vec(n.get("v1"), sh.v1, 3);
            y_vec(n.get("v2"), sh.v2, 3);
            y_f32(n.get("power"), sh.power);
            y_u32(n.get("max_iterations"), sh.max_iterations);
            if (const YNode* tx = n.get("texture"); tx && tx->kind == YNode::Scalar) sh.texture = resolve_path(scene_dir, tx->s, root);
            if (y_f32(n.get("texture_scale"), sh.texture_scale)) sh.has_texture_scale = true;
            read_material(n.get("material"), sh.material);
            s->shapes.push_back(sh);
        }
    }
    if (const YNode* models = doc.get("models"); models && models->kind == YNode::Seq) {
        for (auto& n : models->seq) {
            if (n.kind != YNode::Map) continue;
            ModelRef m;
            if (const YNode* p = n.get("path")) m.path = resolve_path(scene_dir, p->s, root);
            y_vec(n.get("position"), m.position, 3);
            y_f32(n.get("scale"), m.scale);
            read_material(n.get("material"), m.material);
            std::string oe;
            std::vector<Shape> tris;
            if (load_obj(m.path, m.position, m.scale, m.material, tris, oe)) s->shapes.insert(s->shapes.end(), tris.begin(), tris.end());
            // state.rs:138: a model that fails to load is logged and skipped
        }
    }
    collect_textures(s);
    return s;
}

// ---- scene export (src/scene/exporter.rs:11-70, src/app/scene_ops.rs:115-124) ----------------------------------------
namespace {
// An f32 as the shortest decimal that reads back to the same value, always with a fraction or an exponent
// (serde_yml prints floats through ryu: 1.0, 0.8, 1e-5), YAML spellings for the non-finite values.
std::string yaml_f32(float v) {
    if (v != v) return ".nan";
    if (v == INF) return ".inf";
    if (v == -INF) return "-.inf";
    char buf[48];
    auto r = std::to_chars(buf, buf + sizeof(buf), v);
    std::string t(buf, r.ptr);
    if (t.find_first_of(".en") == std::string::npos) t += ".0";
    return t;
}
std::string yaml_vec3(const float* v) { return "[" + yaml_f32(v[0]) + ", " + yaml_f32(v[1]) + ", " + yaml_f32(v[2]) + "]"; }
// A plain scalar when that is unambiguous, else double-quoted.
std::string yaml_str(const std::string& v) {
    bool plain = !v.empty() && v.find_first_of(":#[]{},&*!|>'\"%@`\n\t") == std::string::npos && v.front() != ' ' && v.back() != ' ' &&
                 v.front() != '-' && v.front() != '?';
    if (plain) {
        char* e = nullptr;
        (void)std::strtod(v.c_str(), &e);
        if (*e == '\0' || v == "true" || v == "false" || v == "null" || v == "~" || v == "yes" || v == "no") plain = false;
    }
    if (plain) return v;
    std::string q = "\"";
    for (char ch : v) {
        if (ch == '"' || ch == '\\') q += '\\';
        if (ch == '\n') { q += "\\n"; continue; }
        q += ch;
    }
    return q + "\"";
}
bool zero3(const float* v) { return v[0] == 0.0f && v[1] == 0.0f && v[2] == 0.0f; }
bool eq3(const float* v, float a, float b, float c) { return v[0] == a && v[1] == b && v[2] == c; }
bool material_is_default(const Material& m) {  // material.rs:102-104
    return eq3(m.base_color, 0.8f, 0.8f, 0.8f) && m.metallic == 0.0f && m.roughness == 0.5f && zero3(m.emission) &&
           m.emission_strength == 0.0f && m.ior == 1.5f && m.transmission == 0.0f && m.texture_id == -1;
}
}  // namespace

// AppState::save_scene + exporter::save_scene: the camera block and every shape (model triangles included, as the
// reference saves its expanded shape list with `models: []`), fields at their serde default left out
// (scene.rs:72-141, shape.rs:104-176, material.rs:9-39), numeric arrays in flow style (collapse_block_arrays).
// Per-vertex UVs are not written (shape.rs:169-175: skip_serializing).
int pth_scene_save(const pth_scene* s, const char* yaml_path) {
    if (!s || !yaml_path) return -1;
    std::ofstream f(yaml_path);
    if (!f) return -2;
    const CameraConfig& c = s->camera;
    const CameraConfig d;
    std::string cam;
    if (!zero3(c.position)) cam += "  position: " + yaml_vec3(c.position) + "\n";
    if (!zero3(c.rotation)) cam += "  rotation: " + yaml_vec3(c.rotation) + "\n";
    if (c.fov != d.fov) cam += "  fov: " + yaml_f32(c.fov) + "\n";
    if (c.exposure != d.exposure) cam += "  exposure: " + yaml_f32(c.exposure) + "\n";
    if (c.max_bounces != d.max_bounces) cam += "  max_bounces: " + std::to_string(c.max_bounces) + "\n";
    if (c.firefly_clamp != d.firefly_clamp) cam += "  firefly_clamp: " + yaml_f32(c.firefly_clamp) + "\n";
    if (!eq3(c.skybox_color, d.skybox_color[0], d.skybox_color[1], d.skybox_color[2])) cam += "  skybox_color: " + yaml_vec3(c.skybox_color) + "\n";
    if (c.skybox_brightness != d.skybox_brightness) cam += "  skybox_brightness: " + yaml_f32(c.skybox_brightness) + "\n";
    if (c.tone_mapper != d.tone_mapper) cam += "  tone_mapper: " + std::to_string(c.tone_mapper) + "\n";
    if (c.fractal_march_steps != d.fractal_march_steps) cam += "  fractal_march_steps: " + std::to_string(c.fractal_march_steps) + "\n";
    f << (cam.empty() ? "camera: {}\n" : "camera:\n" + cam);
    if (s->shapes.empty()) f << "shapes: []\n";
    else f << "shapes:\n";
    for (const Shape& sh : s->shapes) {
        std::string o;
        auto field = [&](const std::string& line) { o += (o.empty() ? "- " : "  ") + line + "\n"; };
        if (!sh.name.empty()) field("name: " + yaml_str(sh.name));
        field(std::string("type: ") + (sh.type < 17u ? TYPE_NAMES[sh.type] : "sphere"));
        if (sh.negative) field("negative: true");
        if (!zero3(sh.position)) field("position: " + yaml_vec3(sh.position));
        if (!eq3(sh.normal, 0.0f, 1.0f, 0.0f)) field("normal: " + yaml_vec3(sh.normal));
        if (sh.radius != 1.0f) field("radius: " + yaml_f32(sh.radius));
        if (sh.radius2 != 0.0f) field("radius2: " + yaml_f32(sh.radius2));
        if (sh.height != 0.0f) field("height: " + yaml_f32(sh.height));
        if (!zero3(sh.rotation)) field("rotation: " + yaml_vec3(sh.rotation));
        if (!zero3(sh.v0)) field("v0: " + yaml_vec3(sh.v0));
        if (!zero3(sh.v1)) field("v1: " + yaml_vec3(sh.v1));
        if (!zero3(sh.v2)) field("v2: " + yaml_vec3(sh.v2));
        if (sh.power != 8.0f) field("power: " + yaml_f32(sh.power));
        if (sh.max_iterations != 12u) field("max_iterations: " + std::to_string(sh.max_iterations));
        if (!sh.texture.empty()) field("texture: " + yaml_str(sh.texture));
        if (sh.has_texture_scale) field("texture_scale: " + yaml_f32(sh.texture_scale));
        const Material& m = sh.material;
        if (!material_is_default(m)) {
            field("material:");
            if (!eq3(m.base_color, 0.8f, 0.8f, 0.8f)) field("  base_color: " + yaml_vec3(m.base_color));
            if (m.metallic != 0.0f) field("  metallic: " + yaml_f32(m.metallic));
            if (m.roughness != 0.5f) field("  roughness: " + yaml_f32(m.roughness));
            if (!zero3(m.emission)) field("  emission: " + yaml_vec3(m.emission));
            if (m.emission_strength != 0.0f) field("  emission_strength: " + yaml_f32(m.emission_strength));
            if (m.ior != 1.5f) field("  ior: " + yaml_f32(m.ior));
            if (m.transmission != 0.0f) field("  transmission: " + yaml_f32(m.transmission));
            if (m.texture_id != -1) field("  texture_id: " + std::to_string(m.texture_id));
        }
        f << o;
    }
    f.flush();
    return f ? 0 : -2;
}

int pth_scene_append_shapes(pth_scene* s, const ptx_shape* figs, const ptx_material* mats, uint32_t n) {
    if (!s || (n && (!figs || !mats))) return -1;
    for (uint32_t i = 0; i < n; i++) {
        Shape sh;
        const ptx_shape& f = figs[i];
        if (f.shape_type > 16u) { s->err = "shape_type out of range"; return -1; }
        sh.type = f.shape_type;
        sh.negative = f.csg_op == 1u;
        memcpy(sh.position, f.position, 12);
        memcpy(sh.normal, f.normal, 12);
        sh.radius = f.radius; sh.radius2 = f.radius2; sh.height = f.height;
        memcpy(sh.rotation, f.rotation, 12);
        memcpy(sh.v0, f.v0, 12); memcpy(sh.v1, f.v1, 12); memcpy(sh.v2, f.v2, 12);
        if (f.shape_type == 9u || f.shape_type == 10u) { sh.power = f.v0[0]; sh.max_iterations = (uint32_t)f.v0[1]; }
        sh.has_texture_scale = f.texture_scale != 1.0f;  // the record holds texture_scale.unwrap_or(1.0): 1.0 reads back as None
        sh.texture_scale = f.texture_scale;
        {   // binary16 -> f32 is exact and the build's truncating f32 -> binary16 restores the same bits
            auto h2f = [](uint32_t h) {
                uint32_t sign = (h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu, bits;
                if (e == 0) { float v = (float)m * 5.9604644775390625e-8f; memcpy(&bits, &v, 4); bits |= sign; }
                else if (e == 31) bits = sign | 0x7F800000u | (m << 13);
                else bits = sign | ((e + 112u) << 23) | (m << 13);
                float o; memcpy(&o, &bits, 4); return o;
            };
            sh.uv0[0] = h2f(f._pad2 & 0xFFFFu); sh.uv0[1] = h2f(f._pad2 >> 16);
            sh.uv1[0] = h2f(f._pad3 & 0xFFFFu); sh.uv1[1] = h2f(f._pad3 >> 16);
            sh.uv2[0] = h2f(f._pad4 & 0xFFFFu); sh.uv2[1] = h2f(f._pad4 >> 16);
        }
        const ptx_material& m = mats[i];
        memcpy(sh.material.base_color, m.base_color, 12);
        memcpy(sh.material.emission, m.emission, 12);
        sh.material.metallic = m.metallic; sh.material.roughness = m.roughness;
        sh.material.emission_strength = m.emission_strength; sh.material.ior = m.ior;
        sh.material.transmission = m.transmission; sh.material.texture_id = m.texture_id;
        s->shapes.push_back(sh);
    }
    s->built = false;
    return 0;
}

int pth_scene_set_shape_texture(pth_scene* s, uint32_t shape_index, const char* texture_path) {
    if (!s || shape_index >= s->shapes.size() || !texture_path) return -1;
    s->shapes[shape_index].texture = texture_path;
    collect_textures(s);
    s->built = false;
    return 0;
}
uint32_t pth_scene_shape_count(const pth_scene* s) { return s ? (uint32_t)s->shapes.size() : 0; }
int pth_scene_set_camera(pth_scene* s, const float position[3], const float rotation[3], float fov) {
    if (!s || !position || !rotation) return -1;
    memcpy(s->camera.position, position, 12);
    memcpy(s->camera.rotation, rotation, 12);
    s->camera.fov = fov;
    return 0;
}
int pth_scene_set_sky(pth_scene* s, const float skybox_color[3], float skybox_brightness) {
    if (!s || !skybox_color) return -1;
    memcpy(s->camera.skybox_color, skybox_color, 12);
    s->camera.skybox_brightness = skybox_brightness;
    return 0;
}

int pth_scene_append_heightfield(pth_scene* s, uint32_t quads, float extent, float y0, float amp, uint32_t seed, uint32_t glass_every) {
    if (!s || quads == 0 || quads > 8192u) return -1;
    const uint32_t nv = quads + 1;
    auto lattice = [&](int32_t ix, int32_t iz, uint32_t oct) {
        uint32_t h = pcg((uint32_t)ix * 73856093u ^ (uint32_t)iz * 19349663u ^ (seed + oct * 0x9E3779B9u));
        return (float)(h >> 8) * (1.0f / 16777216.0f);
    };
    auto noise = [&](float x, float z, uint32_t oct) {
        float fx = std::floor(x), fz = std::floor(z);
        int32_t ix = (int32_t)fx, iz = (int32_t)fz;
        float tx = x - fx, tz = z - fz;
        float sx = tx * tx * (3.0f - 2.0f * tx), sz = tz * tz * (3.0f - 2.0f * tz);
        float a = lattice(ix, iz, oct), b = lattice(ix + 1, iz, oct), c = lattice(ix, iz + 1, oct), d = lattice(ix + 1, iz + 1, oct);
        float ab = a + (b - a) * sx, cd = c + (d - c) * sx;
        return ab + (cd - ab) * sz;
    };
    std::vector<float> hgt((size_t)nv * nv);
    float step = 2.0f * extent / (float)quads;
    for (uint32_t j = 0; j < nv; j++)
        for (uint32_t i = 0; i < nv; i++) {
            float x = -extent + (float)i * step, z = -extent + (float)j * step;
            float f = 0.0f, a = 0.5f, fr = 0.35f;
            for (uint32_t o = 0; o < 4; o++) { f += a * noise(x * fr, z * fr, o); a *= 0.5f; fr *= 2.0f; }
            hgt[(size_t)j * nv + i] = y0 + amp * f;
        }
    s->shapes.reserve(s->shapes.size() + (size_t)quads * quads * 2);
    uint32_t tri = 0;
    for (uint32_t j = 0; j < quads; j++)
        for (uint32_t i = 0; i < quads; i++) {
            float x0 = -extent + (float)i * step, x1 = -extent + (float)(i + 1) * step;
            float z0 = -extent + (float)j * step, z1 = -extent + (float)(j + 1) * step;
            float h00 = hgt[(size_t)j * nv + i], h10 = hgt[(size_t)j * nv + i + 1];
            float h01 = hgt[(size_t)(j + 1) * nv + i], h11 = hgt[(size_t)(j + 1) * nv + i + 1];
            float P[4][3] = {{x0, h00, z0}, {x1, h10, z0}, {x0, h01, z1}, {x1, h11, z1}};
            int order[2][3] = {{0, 2, 1}, {1, 2, 3}};
            for (int t = 0; t < 2; t++, tri++) {
                Shape sh;
                sh.type = 7;
                sh.radius = 0.0f;
                sh.power = 0.0f;
                sh.max_iterations = 0;
                memcpy(sh.v0, P[order[t][0]], 12);
                memcpy(sh.v1, P[order[t][1]], 12);
                memcpy(sh.v2, P[order[t][2]], 12);
                uint32_t h = pcg(tri ^ seed);
                float cr = 0.35f + 0.5f * (float)(h & 0xFFu) * (1.0f / 255.0f);
                float cg = 0.35f + 0.5f * (float)((h >> 8) & 0xFFu) * (1.0f / 255.0f);
                float cb = 0.35f + 0.5f * (float)((h >> 16) & 0xFFu) * (1.0f / 255.0f);
                float bc[3] = {cr, cg, cb};
                memcpy(sh.material.base_color, bc, 12);
                sh.material.roughness = 0.8f;
                if (glass_every && tri % glass_every == glass_every - 1) {
                    float w[3] = {1.0f, 1.0f, 1.0f};
                    memcpy(sh.material.base_color, w, 12);
                    sh.material.transmission = 1.0f;
                    sh.material.ior = 1.5f;
                    sh.material.roughness = 0.0f;
                }
                s->shapes.push_back(sh);
            }
        }
    s->built = false;
    return 0;
}

uint32_t pth_scene_texture_count(const pth_scene* s) { return s ? (uint32_t)s->tex_paths.size() : 0; }
const char* pth_scene_texture_path(const pth_scene* s, uint32_t i) { return s && i < s->tex_paths.size() ? s->tex_paths[i].c_str() : nullptr; }
int pth_scene_provide_texture(pth_scene* s, uint32_t i, const uint8_t* rgba8, uint32_t width, uint32_t height) {
    if (!s || i >= s->tex.size() || !rgba8 || width == 0 || height == 0) return -1;
    pth_scene::Tex& t = s->tex[i];
    t.provided = true;
    t.w = width; t.h = height;
    t.px.resize((size_t)width * height);
    for (size_t k = 0; k < t.px.size(); k++)  // texture_atlas.rs:81-84: 0xAABBGGRR
        t.px[k] = ((uint32_t)rgba8[4 * k + 3] << 24) | ((uint32_t)rgba8[4 * k + 2] << 16) | ((uint32_t)rgba8[4 * k + 1] << 8) | rgba8[4 * k];
    s->built = false;
    return 0;
}

// Bvh::build over plain arrays (bvh.rs:37-57), the host form of pth_bvh_fn.
int pth_bvh_build(void* /*user*/, const float* aabbs, uint32_t n, ptx_bvh_node* nodes_out, uint32_t* n_nodes_out, uint32_t* prim_indices_out) {
    if (!nodes_out || !n_nodes_out || (n > 0 && (!aabbs || !prim_indices_out))) return -1;
    if (n == 0) {  // bvh.rs:38-43
        memset(nodes_out, 0, sizeof(ptx_bvh_node));
        *n_nodes_out = 1;
        return 0;
    }
    std::vector<Aabb> boxes(n);
    for (uint32_t k = 0; k < n; k++) boxes[k] = {mk(aabbs + (size_t)k * 6), mk(aabbs + (size_t)k * 6 + 3)};
    BvhBuilder b(boxes);
    b.build();
    // bvh.rs:188-213: the build order is already the pre-order of `flatten`, so node i of the build list is
    // node i of the flat array and the right child index is the build index.
    for (size_t i = 0; i < b.nodes.size(); i++) {
        const BuildNode& bn = b.nodes[i];
        ptx_bvh_node& o = nodes_out[i];
        o.aabb_min[0] = bn.bounds.mn.x; o.aabb_min[1] = bn.bounds.mn.y; o.aabb_min[2] = bn.bounds.mn.z;
        o.aabb_max[0] = bn.bounds.mx.x; o.aabb_max[1] = bn.bounds.mx.y; o.aabb_max[2] = bn.bounds.mx.z;
        if (bn.count > 0) { o.left_or_prim = (uint32_t)bn.first; o.prim_count = (uint32_t)bn.count; }
        else { o.left_or_prim = (uint32_t)bn.right; o.prim_count = 0; }
    }
    *n_nodes_out = (uint32_t)b.nodes.size();
    memcpy(prim_indices_out, b.indices.data(), (size_t)n * 4);
    return 0;
}

int pth_scene_build(pth_scene* s) { return pth_scene_build_with(s, pth_bvh_build, nullptr); }

int pth_scene_build_with(pth_scene* s, pth_bvh_fn bvh, void* user) {
    if (!s || !bvh) return -1;
    collect_textures(s);
    // texture_atlas.rs:26-79 + state.rs:328-348
    s->tex_pixels.assign(1, 0xFF808080u);
    s->tex_infos.assign(1, ptx_texture_info{1, 1, 0, 0});
    std::map<std::string, int32_t> cache;
    for (size_t i = 0; i < s->tex_paths.size(); i++) {
        if (!s->tex[i].provided) continue;
        ptx_texture_info ti{s->tex[i].w, s->tex[i].h, (uint32_t)s->tex_pixels.size(), 0};
        cache[s->tex_paths[i]] = (int32_t)s->tex_infos.size();
        s->tex_infos.push_back(ti);
        s->tex_pixels.insert(s->tex_pixels.end(), s->tex[i].px.begin(), s->tex[i].px.end());
    }
    // state.rs:350-377 (records) and 543-567 (finite / plane split, boxes).  Every shape is independent; large scenes
    // are packed by several threads, each over a contiguous slice, the per-slice index lists joined in slice order.
    size_t n = s->shapes.size();
    s->figs.resize(n);
    s->mats.resize(n);
    size_t n_slices = n < 65536 ? 1 : std::min<size_t>(16, std::max<unsigned>(1u, std::thread::hardware_concurrency()));
    if (const char* e = getenv("PTH_BUILD_THREADS")) n_slices = std::min<size_t>(64, std::max(1, atoi(e)));  // tests, measurements
    struct Slice { std::vector<uint32_t> lights, infinite, finite; };
    std::vector<Slice> slices(n_slices);
    auto pack = [&](size_t si) {
        Slice& out = slices[si];
        const size_t i0 = n * si / n_slices, i1 = n * (si + 1) / n_slices;
        for (size_t i = i0; i < i1; i++) {
            const Shape& sh = s->shapes[i];
            ptx_material m{};  // material.rs:121-134
            memcpy(m.base_color, sh.material.base_color, 12);
            m.metallic = sh.material.metallic;
            memcpy(m.emission, sh.material.emission, 12);
            m.roughness = std::fmax(sh.material.roughness, 0.04f);
            m.emission_strength = sh.material.emission_strength;
            m.ior = sh.material.ior;
            m.transmission = sh.material.transmission;
            m.texture_id = sh.material.texture_id;
            if (!sh.texture.empty()) {
                auto it = cache.find(sh.texture);
                if (it != cache.end()) m.texture_id = it->second;
            }
            s->mats[i] = m;
            ptx_shape g{};  // shape.rs:251-278
            g.shape_type = sh.type;
            g.material_idx = (uint32_t)i;
            g.radius = sh.radius; g.radius2 = sh.radius2; g.height = sh.height;
            memcpy(g.position, sh.position, 12);
            {   // glam Vec3::normalize_or_zero
                float x = sh.normal[0], y = sh.normal[1], z = sh.normal[2];
                float d = x * x + y * y + z * z;
                float rcp = 1.0f / std::sqrt(d);
                if (std::isfinite(rcp) && rcp > 0.0f) { g.normal[0] = x * rcp; g.normal[1] = y * rcp; g.normal[2] = z * rcp; }
                else { g.normal[0] = g.normal[1] = g.normal[2] = 0.0f; }
            }
            g.csg_op = sh.negative ? 1u : 0u;
            memcpy(g.rotation, sh.rotation, 12);
            g.texture_scale = sh.has_texture_scale ? sh.texture_scale : 1.0f;
            if (sh.type == 9u || sh.type == 10u) { g.v0[0] = sh.power; g.v0[1] = (float)sh.max_iterations; g.v0[2] = 0.0f; }
            else memcpy(g.v0, sh.v0, 12);
            memcpy(g.v1, sh.v1, 12);
            memcpy(g.v2, sh.v2, 12);
            g._pad2 = (uint32_t)pth_f32_to_f16_bits(sh.uv0[0]) | ((uint32_t)pth_f32_to_f16_bits(sh.uv0[1]) << 16);
            g._pad3 = (uint32_t)pth_f32_to_f16_bits(sh.uv1[0]) | ((uint32_t)pth_f32_to_f16_bits(sh.uv1[1]) << 16);
            g._pad4 = (uint32_t)pth_f32_to_f16_bits(sh.uv2[0]) | ((uint32_t)pth_f32_to_f16_bits(sh.uv2[1]) << 16);
            s->figs[i] = g;
            if (sh.material.is_emissive()) out.lights.push_back((uint32_t)i);
            if (sh.type == 1u) out.infinite.push_back((uint32_t)i);
            else if (sh.type != 8u) out.finite.push_back((uint32_t)i);
        }
    };
    auto run_slices = [&](auto&& fn) {
        if (n_slices == 1) { fn(0); return; }
        std::vector<std::thread> th;
        for (size_t si = 0; si < n_slices; si++) th.emplace_back(fn, si);
        for (auto& t : th) t.join();
    };
    run_slices(pack);
    std::vector<uint32_t> finite;
    s->lights.clear();
    s->infinite.clear();
    for (const Slice& sl : slices) {
        s->lights.insert(s->lights.end(), sl.lights.begin(), sl.lights.end());
        s->infinite.insert(s->infinite.end(), sl.infinite.begin(), sl.infinite.end());
        finite.insert(finite.end(), sl.finite.begin(), sl.finite.end());
    }
    std::vector<float> boxes(finite.size() * 6);
    run_slices([&](size_t si) {
        const size_t k0 = finite.size() * si / n_slices, k1 = finite.size() * (si + 1) / n_slices;
        for (size_t k = k0; k < k1; k++) {
            Aabb b = shape_aabb(s->shapes[finite[k]]);
            float* o = boxes.data() + k * 6;
            o[0] = b.mn.x; o[1] = b.mn.y; o[2] = b.mn.z; o[3] = b.mx.x; o[4] = b.mx.y; o[5] = b.mx.z;
        }
    });
    s->nodes.assign(std::max<size_t>(1, 2 * finite.size()), ptx_bvh_node{});
    s->prims.assign(finite.size(), 0u);
    uint32_t n_nodes = 0;
    if (int rc = bvh(user, boxes.data(), (uint32_t)finite.size(), s->nodes.data(), &n_nodes, s->prims.data())) {
        s->err = "the BVH builder failed with code " + std::to_string(rc);
        return -2;
    }
    s->nodes.resize(n_nodes);
    for (size_t k = 0; k < s->prims.size(); k++) s->prims[k] = finite[s->prims[k]];  // state.rs:559-565: leaf indices -> figure indices
    s->built = true;
    return 0;
}

#define PTH_GETTER(name, type, field)                                        \
    const type* name(const pth_scene* s, uint32_t* n) {                      \
        if (!s || !s->built) { if (n) *n = 0; return nullptr; }              \
        if (n) *n = (uint32_t)s->field.size();                               \
        return s->field.data();                                              \
    }
PTH_GETTER(pth_scene_figures, ptx_shape, figs)
PTH_GETTER(pth_scene_materials, ptx_material, mats)
PTH_GETTER(pth_scene_bvh_nodes, ptx_bvh_node, nodes)
PTH_GETTER(pth_scene_bvh_prims, uint32_t, prims)
PTH_GETTER(pth_scene_lights, uint32_t, lights)
PTH_GETTER(pth_scene_infinite, uint32_t, infinite)
PTH_GETTER(pth_scene_tex_pixels, uint32_t, tex_pixels)
PTH_GETTER(pth_scene_tex_infos, ptx_texture_info, tex_infos)

int pth_scene_camera(const pth_scene* s, uint32_t width, uint32_t height, uint32_t frame_index, uint32_t sample_count, ptx_camera* out) {
    if (!s || !out || width == 0 || height == 0) return -1;
    const CameraConfig& c = s->camera;
    // camera.rs:29-35: yaw = rotation[1], pitch = rotation[0]; f32::to_radians = x * (PI / 180) in f32.
    const float deg = 3.14159265358979323846f / 180.0f;
    float yaw = c.rotation[1] * deg, pitch = c.rotation[0] * deg;
    // f32 sin/cos/tan taken as the rounded double-precision libm value.
    float hy = yaw * 0.5f, hp = pitch * 0.5f;
    float sy = (float)std::sin((double)hy), cy = (float)std::cos((double)hy);
    float sx = (float)std::sin((double)hp), cx = (float)std::cos((double)hp);
    Q q = qmul(qmul(Q{0.0f, sy, 0.0f, cy}, Q{sx, 0.0f, 0.0f, cx}), Q{0.0f, 0.0f, 0.0f, 1.0f});  // EulerRot::YXZ with roll 0
    V3 fwd = qrot(q, mk(0.0f, 0.0f, 1.0f)), right = qrot(q, mk(1.0f, 0.0f, 0.0f)), up = qrot(q, mk(0.0f, 1.0f, 0.0f));
    memset(out, 0, sizeof(*out));
    memcpy(out->position, c.position, 12);
    float half = c.fov * deg * 0.5f;
    float tn = (float)std::tan((double)half);
    out->focal_length = 1.0f / tn;
    out->right[0] = right.x; out->right[1] = right.y; out->right[2] = right.z;
    out->up[0] = up.x; out->up[1] = up.y; out->up[2] = up.z;
    out->forward[0] = fwd.x; out->forward[1] = fwd.y; out->forward[2] = fwd.z;
    out->aspect = (float)width / (float)height;
    out->exposure = c.exposure;
    out->frame_index = frame_index;
    out->width = width; out->height = height;
    out->sample_count = sample_count;
    out->max_bounces = c.max_bounces;
    out->tone_mapper = c.tone_mapper;
    out->fractal_march_steps = c.fractal_march_steps;
    out->firefly_clamp = c.firefly_clamp;
    out->skybox_brightness = c.skybox_brightness;
    memcpy(out->skybox_color, c.skybox_color, 12);
    return 0;
}

int pth_scene_set_render_settings(pth_scene* s, uint32_t max_bounces, uint32_t fractal_march_steps, uint32_t tone_mapper,
                                  float firefly_clamp, float exposure) {
    if (!s) return -1;
    if (max_bounces != 0xFFFFFFFFu) s->camera.max_bounces = max_bounces;
    if (fractal_march_steps != 0xFFFFFFFFu) s->camera.fractal_march_steps = fractal_march_steps;
    if (tone_mapper != 0xFFFFFFFFu) s->camera.tone_mapper = tone_mapper;
    if (firefly_clamp == firefly_clamp) s->camera.firefly_clamp = firefly_clamp;
    if (exposure == exposure) s->camera.exposure = exposure;
    return 0;
}

void pth_accumulator_init(pth_accumulator* a) { if (a) { a->sample_count = 0; a->dirty = 1; a->frame_index = 0; } }
void pth_accumulator_reset(pth_accumulator* a) { if (a) { a->sample_count = 0; a->dirty = 1; } }  // frame_index keeps running
int pth_accumulator_advance(pth_accumulator* a, uint32_t* frame_for_sample) {
    if (!a) return 0;
    a->sample_count += 1;  // accumulator.rs:31-36
    int needs_clear = a->dirty ? 1 : 0;
    a->dirty = 0;
    if (frame_for_sample) *frame_for_sample = a->frame_index;
    a->frame_index += 1;   // rendering.rs:82: wrapping_add(1)
    return needs_clear;
}
void pth_build_post_params(uint32_t width, uint32_t height, const uint32_t* effects, uint32_t n_effects, uint32_t oil_radius,
                           uint32_t comic_levels, uint32_t out_params[16]) {
    for (int i = 0; i < 16; i++) out_params[i] = 0u;
    out_params[0] = width;
    out_params[1] = height;
    const uint32_t count = n_effects < 8u ? n_effects : 8u;
    out_params[2] = count;
    out_params[3] = oil_radius;
    for (uint32_t i = 0; i < count; i++) out_params[4 + i] = effects[i];
    out_params[12] = comic_levels;
}

uint32_t pth_dispatch_size(uint32_t extent, uint32_t workgroup) { return workgroup ? (extent + workgroup - 1) / workgroup : 0; }

}  // extern "C"
