#include "scene_io.hpp"
#include "decoders.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

namespace evplp {

// Corner triples (indices into `face`) of a triangulation of the polygon face[0 .. n): a fan when every corner is convex in the polygon's own
// plane (or the polygon is degenerate), ear clipping otherwise.  Orientation is kept.
static void triangulate_polygon(const std::vector<float> &pos, const std::vector<std::pair<int, int>> &face, std::vector<int> &out) {
    const int n = (int)face.size();
    out.clear();
    if (n < 3) return;
    auto fan = [&]() { out.clear(); for (int k = 1; k + 1 < n; k++) { out.push_back(0); out.push_back(k); out.push_back(k + 1); } };
    if (n == 3) { fan(); return; }
    auto P = [&](int k, int c) { return (double)pos[(size_t)3 * (size_t)face[(size_t)k].first + (size_t)c]; };
    double nx = 0, ny = 0, nz = 0;                      // Newell normal
    for (int k = 0; k < n; k++) {
        const int j = (k + 1) % n;
        nx += (P(k, 1) - P(j, 1)) * (P(k, 2) + P(j, 2)); ny += (P(k, 2) - P(j, 2)) * (P(k, 0) + P(j, 0)); nz += (P(k, 0) - P(j, 0)) * (P(k, 1) + P(j, 1));
    }
    const double ax = std::fabs(nx), ay = std::fabs(ny), az = std::fabs(nz);
    if (!(ax + ay + az > 0.0)) { fan(); return; }       // degenerate (or NaN): nothing to decide
    // drop the dominant axis; keep the winding counter-clockwise in the projection
    const int drop = ax >= ay && ax >= az ? 0 : ay >= az ? 1 : 2, ua = (drop + 1) % 3, va = (drop + 2) % 3;
    const double sgn = (drop == 0 ? nx : drop == 1 ? ny : nz) > 0 ? 1.0 : -1.0;
    std::vector<double> u((size_t)n), v((size_t)n);
    for (int k = 0; k < n; k++) { u[(size_t)k] = P(k, ua); v[(size_t)k] = sgn * P(k, va); }
    auto cross = [&](int a, int b, int c) { return (u[(size_t)b] - u[(size_t)a]) * (v[(size_t)c] - v[(size_t)a]) - (v[(size_t)b] - v[(size_t)a]) * (u[(size_t)c] - u[(size_t)a]); };
    bool convex = true;
    for (int k = 0; k < n && convex; k++) if (cross((k + n - 1) % n, k, (k + 1) % n) < 0.0) convex = false;
    if (convex) { fan(); return; }
    std::vector<int> ring((size_t)n);
    for (int k = 0; k < n; k++) ring[(size_t)k] = k;
    auto inside = [&](int a, int b, int c, int q) {     // strictly inside or on the boundary of the (counter-clockwise) triangle, corners excluded by the caller
        return cross(a, b, q) >= 0.0 && cross(b, c, q) >= 0.0 && cross(c, a, q) >= 0.0;
    };
    int guard = 0;
    while (ring.size() > 3 && guard++ < 4 * n * n) {
        bool clipped = false;
        const int m = (int)ring.size();
        for (int k = 0; k < m; k++) {
            const int a = ring[(size_t)((k + m - 1) % m)], b = ring[(size_t)k], c = ring[(size_t)((k + 1) % m)];
            if (cross(a, b, c) <= 0.0) continue;         // reflex or flat corner: not an ear
            bool empty = true;
            for (int q : ring) if (q != a && q != b && q != c && inside(a, b, c, q)) { empty = false; break; }
            if (!empty) continue;
            out.push_back(a); out.push_back(b); out.push_back(c);
            ring.erase(ring.begin() + k);
            clipped = true;
            break;
        }
        if (!clipped) {                                   // self-intersecting or collinear leftovers: close with a fan of what remains
            for (size_t k = 1; k + 1 < ring.size(); k++) { out.push_back(ring[0]); out.push_back(ring[k]); out.push_back(ring[k + 1]); }
            return;
        }
    }
    if (ring.size() == 3) { out.push_back(ring[0]); out.push_back(ring[1]); out.push_back(ring[2]); }
}


std::string dirname_of(const std::string &path) {
    size_t i = path.find_last_of("/\\");
    return i == std::string::npos ? std::string(".") : path.substr(0, i);
}
std::string join_path(const std::string &dir, const std::string &rel) {
    if (!rel.empty() && (rel[0] == '/' || (rel.size() > 1 && rel[1] == ':'))) return rel;  // absolute (main.cpp:52)
    return dir + "/" + rel;
}
std::string read_text_file(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::stringstream ss; ss << f.rdbuf();
    return ss.str();
}

namespace {

struct MtlEntry {
    float kd[3] = { 0.6f, 0.6f, 0.6f }, ks[3] = { 0, 0, 0 }; float ns = 0.f;   // Assimp DefaultMaterial: grey 0.6
    std::string map_kd, map_ks, map_ns;
};

// RtTexture(filepath, gamma = 1) (rtcommon.h:139-194): JPEG / PNG decoded to 8-bit RGB (decoders.hpp), texel =
// byte / 255 (pow(x, 1.0) is the identity), alpha 0, rows flipped vertically as
// stbi_set_flip_vertically_on_load(1) does (:32).  The reference indexes 4-channel files with the wrong stride
// (:178-189 reads data[i * 4 + j] from a 3-byte-per-pixel buffer, out of bounds for the last quarter); this build
// reads them as the RGB they are.  PFM and binary PPM are accepted in addition (build-only, for generated scenes).
TextureData load_texture(const std::string &path) {
    TextureData t; t.path = path;
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open texture " + path);
    unsigned char head[8] = { 0 };
    size_t got = std::fread(head, 1, 8, f);
    if (is_jpeg(head, got) || is_png(head, got)) {
        std::fclose(f);
        DecodedImage img = decode_image_file(path);
        t.w = img.w; t.h = img.h; t.rgba.assign((size_t)img.w * img.h * 4, 0.f);
        for (int y = 0; y < img.h; y++) for (int x = 0; x < img.w; x++) for (int k = 0; k < 3; k++)
            t.rgba[4 * ((size_t)(img.h - 1 - y) * img.w + x) + k] = (float)img.rgb[3 * ((size_t)y * img.w + x) + k] / 255.0f;
        return t;
    }
    std::rewind(f);
    char magic[3] = { 0, 0, 0 };
    if (std::fscanf(f, "%2s", magic) != 1) { std::fclose(f); throw std::runtime_error("bad texture header " + path); }
    if (!std::strcmp(magic, "PF")) {
        int w, h; float scale;
        if (std::fscanf(f, "%d %d %f", &w, &h, &scale) != 3) { std::fclose(f); throw std::runtime_error("bad PFM header " + path); }
        std::fgetc(f);
        std::vector<float> rgb((size_t)w * h * 3);
        if (std::fread(rgb.data(), sizeof(float), rgb.size(), f) != rgb.size()) { std::fclose(f); throw std::runtime_error("short PFM " + path); }
        t.w = w; t.h = h; t.rgba.assign((size_t)w * h * 4, 0.f);
        // PFM rows are bottom-to-top already == flipped image rows in GL texture order
        for (size_t i = 0; i < (size_t)w * h; i++) for (int k = 0; k < 3; k++) t.rgba[4 * i + k] = rgb[3 * i + k];
    } else if (!std::strcmp(magic, "P6")) {
        int w, h, maxv;
        if (std::fscanf(f, "%d %d %d", &w, &h, &maxv) != 3 || maxv != 255) { std::fclose(f); throw std::runtime_error("bad PPM header " + path); }
        std::fgetc(f);
        std::vector<unsigned char> rgb((size_t)w * h * 3);
        if (std::fread(rgb.data(), 1, rgb.size(), f) != rgb.size()) { std::fclose(f); throw std::runtime_error("short PPM " + path); }
        t.w = w; t.h = h; t.rgba.assign((size_t)w * h * 4, 0.f);
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) for (int k = 0; k < 3; k++)
            t.rgba[4 * ((size_t)(h - 1 - y) * w + x) + k] = (float)rgb[3 * ((size_t)y * w + x) + k] / 255.0f;   // u8/255, gamma 1.0
    } else { std::fclose(f); throw std::runtime_error("unsupported texture format (JPEG, PNG, PFM or binary PPM expected): " + path); }
    std::fclose(f);
    return t;
}

int texture_id(HostScene &scene, const std::string &path) {
    for (size_t i = 0; i < scene.textures.size(); i++) if (scene.textures[i].path == path) return (int)i;   // gTexturesMap rtcommon.h:33-51
    scene.textures.push_back(load_texture(path));
    return (int)scene.textures.size() - 1;
}

void parse_mtl(const std::string &path, std::map<std::string, MtlEntry> &out, std::vector<std::string> &order) {
    std::ifstream f(path);
    if (!f) return;   // Assimp tolerates a missing MTL: everything gets the default material
    std::string line, cur;
    while (std::getline(f, line)) {
        std::istringstream ss(line); std::string tok; ss >> tok;
        if (tok == "newmtl") { ss >> cur; out[cur] = MtlEntry(); order.push_back(cur); }
        else if (cur.empty()) continue;
        else if (tok == "Kd") ss >> out[cur].kd[0] >> out[cur].kd[1] >> out[cur].kd[2];
        else if (tok == "Ks") ss >> out[cur].ks[0] >> out[cur].ks[1] >> out[cur].ks[2];
        else if (tok == "Ns") ss >> out[cur].ns;
        else if (tok == "map_Kd" || tok == "map_Ks" || tok == "map_Ns") {
            // the file name is the last token (options such as "-s 1 1 1" may precede it); backslashes of Windows exporters
            std::string name, t; while (ss >> t) name = t;
            for (char &ch : name) if (ch == '\\') ch = '/';
            (tok == "map_Kd" ? out[cur].map_kd : tok == "map_Ks" ? out[cur].map_ks : out[cur].map_ns) = name;
        }
    }
}

struct ObjResult { std::vector<MeshData> meshes; std::vector<evplp_material> materials; };

// Wavefront OBJ -> one mesh per material (Assimp: one aiMesh per material group; Triangulate +
// JoinIdenticalVertices, rtcommon.h:650-653).  Vertex normals are not loaded: no device program of
// the reference reads them.
ObjResult read_obj(HostScene &scene, const std::string &obj_path, bool want_materials) {
    std::ifstream f(obj_path);
    if (!f) throw std::runtime_error("Impossible to load the scene: " + obj_path);   // rtcommon.h:658-661
    const std::string dir = dirname_of(obj_path);
    std::vector<float> pos, tex;
    std::map<std::string, MtlEntry> mtl; std::vector<std::string> mtl_order;
    // material slot 0 = DefaultMaterial (rtcommon.h:745)
    std::vector<std::string> mat_names = { "" };
    std::vector<MeshData> meshes(1);
    std::vector<std::map<std::pair<int, int>, int32_t>> dedup(1);
    int cur = 0;
    std::string line;
    std::vector<std::pair<int, int>> face;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        const char *s = line.c_str();
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            float x, y, z; if (std::sscanf(s + 2, "%f %f %f", &x, &y, &z) == 3) { pos.push_back(x); pos.push_back(y); pos.push_back(z); }
        } else if (s[0] == 'v' && s[1] == 't') {
            float u = 0, v = 0; std::sscanf(s + 3, "%f %f", &u, &v); tex.push_back(u); tex.push_back(v);
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            face.clear();
            const char *p = s + 2;
            while (*p) {
                while (*p == ' ' || *p == '\t' || *p == '\r') p++;
                if (!*p) break;
                char *end; long vi = std::strtol(p, &end, 10); long ti = 0;
                if (end == p) break;
                p = end;
                if (*p == '/') { p++; if (*p != '/') { ti = std::strtol(p, &end, 10); p = end; } if (*p == '/') { p++; std::strtol(p, &end, 10); p = end; } }
                const long nv = (long)(pos.size() / 3), nt = (long)(tex.size() / 2);
                const long vl = vi < 0 ? nv + vi : vi - 1;
                long tl = ti == 0 ? -1 : (ti < 0 ? nt + ti : ti - 1);
                if (vl < 0 || vl >= nv) throw std::runtime_error("OBJ face references a missing vertex: " + obj_path);
                if (tl < 0 || tl >= nt) tl = -1;                          // (a texture coordinate that does not exist: none, rtcommon.h:701-705)
                const int v = (int)vl, t = (int)tl;
                face.emplace_back(v, t);
            }
            MeshData &m = meshes[cur];
            auto &dd = dedup[cur];
            auto index_of = [&](const std::pair<int, int> &k) {
                auto it = dd.find(k);
                if (it != dd.end()) return it->second;
                int32_t id = (int32_t)(m.verts.size() / 3);
                m.verts.insert(m.verts.end(), pos.begin() + 3 * k.first, pos.begin() + 3 * k.first + 3);
                if (k.second >= 0 && (size_t)(2 * k.second + 1) < tex.size()) { m.uvs.push_back(tex[2 * k.second]); m.uvs.push_back(tex[2 * k.second + 1]); }
                else { m.uvs.push_back(0.f); m.uvs.push_back(0.f); }   // rtcommon.h:701-705
                dd[k] = id; return id;
            };
            // aiProcess_Triangulate (rtcommon.h:650-653).  A CONVEX polygon is a fan from its first corner -- what Assimp's triangulation step gives
            // a convex quad, and the same surface as any other triangulation of a planar polygon.  A polygon with a reflex corner is ear-clipped
            // in the plane of its Newell normal (round 6; a fan would cover area outside it): Assimp 3.3.0 clips ears too, though not necessarily
            // in this order -- the surface is the same, the triangle set may differ (the library is a binary the reference links; not restated).
            std::vector<int> order;
            triangulate_polygon(pos, face, order);
            for (size_t k = 0; k + 2 < order.size(); k += 3) {
                m.idx.push_back(index_of(face[(size_t)order[k]])); m.idx.push_back(index_of(face[(size_t)order[k + 1]])); m.idx.push_back(index_of(face[(size_t)order[k + 2]]));
            }
        } else if (!line.compare(0, 6, "usemtl")) {
            std::istringstream ss(line.substr(6)); std::string name; ss >> name;
            int found = -1;
            for (size_t i = 0; i < mat_names.size(); i++) if (mat_names[i] == name) found = (int)i;
            if (found < 0) { mat_names.push_back(name); meshes.emplace_back(); dedup.emplace_back(); found = (int)mat_names.size() - 1; }
            cur = found;
        } else if (!line.compare(0, 6, "mtllib")) {
            std::istringstream ss(line.substr(6)); std::string name; ss >> name;
            parse_mtl(join_path(dir, name), mtl, mtl_order);
        }
    }
    ObjResult r;
    for (size_t i = 0; i < meshes.size(); i++) {
        if (meshes[i].idx.empty()) continue;
        MeshData m = std::move(meshes[i]);
        if (want_materials) {
            MtlEntry e; auto it = mtl.find(mat_names[i]); if (it != mtl.end()) e = it->second;
            evplp_material em; std::memset(&em, 0, sizeof(em));
            std::memcpy(em.kd, e.kd, 12); std::memcpy(em.ks, e.ks, 12);
            // A constant Ns reaches the renderer unchanged: Assimp's OBJ importer scales it by 4 and the
            // reference divides the constant by 4 again (rtcommon.h:55-64).
            em.ns = e.ns;
            em.tex_kd = e.map_kd.empty() ? -1 : texture_id(scene, join_path(dir, e.map_kd));
            em.tex_ks = e.map_ks.empty() ? -1 : texture_id(scene, join_path(dir, e.map_ks));
            em.tex_ns = e.map_ns.empty() ? -1 : texture_id(scene, join_path(dir, e.map_ns));
            m.material = (int32_t)r.materials.size();
            r.materials.push_back(em);
        }
        r.meshes.push_back(std::move(m));
    }
    if (r.meshes.empty()) {
        std::ifstream head(obj_path); std::string first; std::getline(head, first);
        if (first.compare(0, 22, "version https://git-lf") == 0)
            throw std::runtime_error("Impossible to load the scene: " + obj_path + " is a Git-LFS pointer, not the mesh (fetch the asset with git lfs pull)");
        throw std::runtime_error("OBJ has no faces: " + obj_path);
    }
    return r;
}

} // namespace

void add_obj(HostScene &scene, const std::string &obj_path) {
    ObjResult r = read_obj(scene, obj_path, true);
    const int32_t mat_offset = (int32_t)scene.materials.size();   // rtcommon.h:663
    for (auto &m : r.materials) scene.materials.push_back(m);
    for (auto &m : r.meshes) { m.material += mat_offset; scene.meshes.push_back(std::move(m)); }
}

void add_arealight(HostScene &scene, const std::string &obj_path, const float intensity[4]) {
    if (scene.light_mesh >= 0) throw std::runtime_error("only one area light is supported (rtcommon.h:770-774)");
    ObjResult r = read_obj(scene, obj_path, false);
    if (r.meshes.size() != 1) throw std::runtime_error("the area-light OBJ must contain exactly one mesh (rtcommon.h:794-795): " + obj_path);
    // overrideMaterial: the mesh gets a placeholder; evplp_set_arealight installs the emitter material
    evplp_material black; std::memset(&black, 0, sizeof(black)); black.tex_kd = black.tex_ks = black.tex_ns = -1;
    r.meshes[0].material = (int32_t)scene.materials.size();
    scene.materials.push_back(black);
    scene.meshes.push_back(std::move(r.meshes[0]));
    scene.light_mesh = (int32_t)scene.meshes.size() - 1;
    std::memcpy(scene.light_intensity, intensity, 16);
}

MeshData load_single_mesh_obj(const std::string &obj_path) {
    HostScene scratch;
    ObjResult r = read_obj(scratch, obj_path, false);
    if (r.meshes.size() != 1) throw std::runtime_error("the OBJ must contain exactly one mesh (rtcomphoton.h:635): " + obj_path);
    return std::move(r.meshes[0]);
}

static void vec3_from(const Json &j, float out[3], const char *what) {
    if (!j.is_array() || j.size() != 3) throw JsonError(std::string(what) + ": expected an array of 3 numbers");
    for (int k = 0; k < 3; k++) out[k] = j.at((size_t)k).as_float(what);
}

evplp_camera camera_from_json(const Json &j, float aspect) {
    evplp_camera c; std::memset(&c, 0, sizeof(c));
    const float deg = 0.01745329251994329576923690768489f;   // glm::radians
    if (j.has("fovy")) c.fovy = j.at("fovy").as_float("fovy") * deg;                                           // rtcommon.h:551-555
    else if (j.has("fovx")) c.fovy = 2.0f * std::atan2(std::tan(j.at("fovx").as_float("fovx") * deg * 0.5f), aspect);   // :556-560
    else throw JsonError("camera: forgot fov (fovx or fovy)");                                                  // :563
    vec3_from(j.at("origin"), c.origin, "camera.origin");
    vec3_from(j.at("direction"), c.lookat, "camera.direction");   // a look-AT point (:567, :588)
    vec3_from(j.at("up"), c.up, "camera.up");
    c.aspect = aspect;
    return c;
}

HostScene load_scene(const Json &root, const std::string &json_path) {
    HostScene s;
    const std::string dir = dirname_of(json_path);
    s.res_x = (int32_t)root.at("resX").as_int("resX"); s.res_y = (int32_t)root.at("resY").as_int("resY");
    const Json &list = root.at("scene");                                     // main.cpp:46-58
    if (!list.is_array()) throw JsonError("scene: expected an array of OBJ paths");
    for (size_t i = 0; i < list.size(); i++) add_obj(s, join_path(dir, list.at(i).as_string("scene[]")));
    const Json &al = root.at("arealight");                                   // main.cpp:60-67
    const Json &I = al.at("intensity");
    if (!I.is_array() || I.size() != 4) throw JsonError("arealight.intensity: expected 4 numbers");
    float inten[4]; for (int k = 0; k < 4; k++) inten[k] = I.at((size_t)k).as_float("arealight.intensity");
    add_arealight(s, join_path(dir, al.at("obj").as_string("arealight.obj")), inten);
    float aspect = (float)s.res_x / (float)s.res_y;                          // main.cpp:70
    if (root.has("camera")) { s.camera = camera_from_json(root.at("camera"), aspect); s.has_camera = true; }
    else if (root.has("stablecamera")) { s.camera = camera_from_json(root.at("stablecamera"), aspect); s.has_camera = true; }
    else throw JsonError("missing required key \"camera\" (or \"stablecamera\")");
    return s;
}

int upload_scene(evplp_context *ctx, const HostScene &scene) {
    int rc;
    std::vector<int> tex_ids;
    for (const TextureData &t : scene.textures) {
        rc = evplp_add_texture(ctx, t.w, t.h, t.rgba.data());
        if (rc < 0) return rc;
        tex_ids.push_back(rc);
    }
    std::vector<int> mat_ids;
    for (evplp_material m : scene.materials) {
        if (m.tex_kd >= 0) m.tex_kd = tex_ids[m.tex_kd];
        if (m.tex_ks >= 0) m.tex_ks = tex_ids[m.tex_ks];
        if (m.tex_ns >= 0) m.tex_ns = tex_ids[m.tex_ns];
        rc = evplp_add_material(ctx, &m);
        if (rc < 0) return rc;
        mat_ids.push_back(rc);
    }
    int light = -1;
    for (size_t i = 0; i < scene.meshes.size(); i++) {
        const MeshData &m = scene.meshes[i];
        rc = evplp_add_mesh(ctx, m.verts.data(), m.uvs.data(), (int32_t)(m.verts.size() / 3), m.idx.data(), (int32_t)(m.idx.size() / 3), mat_ids[m.material]);
        if (rc < 0) return rc;
        if ((int)i == scene.light_mesh) light = rc;
    }
    if (light < 0) return EVPLP_ERR_INVALID;
    if ((rc = evplp_set_arealight(ctx, light, scene.light_intensity)) < 0) return rc;
    if ((rc = evplp_set_camera(ctx, &scene.camera)) < 0) return rc;
    return evplp_build_accel(ctx);
}

} // namespace evplp
