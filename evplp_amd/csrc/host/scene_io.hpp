// Host scene model + loaders: the RtScene contract (rt/rtcommon.h:816-819) without GL/OptiX
// members, filled by an OBJ/MTL reader that replaces the Assimp import of
// RtScene::addObject / addAreaLight (rt/rtcommon.h:644-798) and by LoadScene (main.cpp:42-85).
#pragma once
#include "../../../include/evplp.h"
#include "json.hpp"

#include <string>
#include <vector>

namespace evplp {

struct MeshData {
    std::vector<float> verts;    // float3 per vertex      (RtMesh::mVertices  rtcommon.h:460)
    std::vector<float> uvs;      // float2 per vertex      (RtMesh::mTexCoords :462)
    std::vector<int32_t> idx;    // int3 per triangle      (RtMesh::mTriIndices :463)
    int32_t material = 0;        //                        (RtMesh::mMatIndex :467)
};
struct TextureData { int32_t w = 0, h = 0; std::vector<float> rgba; std::string path; };

struct HostScene {
    std::vector<MeshData> meshes;
    std::vector<evplp_material> materials;
    std::vector<TextureData> textures;
    int32_t light_mesh = -1;
    float light_intensity[4] = { 0, 0, 0, 0 };   // JSON arealight.intensity, unscaled
    evplp_camera camera{};
    bool has_camera = false;
    int32_t res_x = 0, res_y = 0;
};

// RtScene::addObject (rtcommon.h:644-757): appends one mesh per material group of the OBJ.
// Throws JsonError/std::runtime_error with a message on failure.
void add_obj(HostScene &scene, const std::string &obj_path);
// RtScene::addAreaLight (rtcommon.h:772-798): the OBJ must yield exactly one mesh.
void add_arealight(HostScene &scene, const std::string &obj_path, const float intensity[4]);
// TriangleMesh::LoadMeshes of setupPhotonSplatIcosohedron (rtcomphoton.h:632-644): the OBJ must yield exactly one mesh (:635)
MeshData load_single_mesh_obj(const std::string &obj_path);
// RtStableCamera (rtcommon.h:548-571): origin / direction (= look-at point) / up / fovx|fovy (degrees)
evplp_camera camera_from_json(const Json &j, float aspect);
// LoadScene (main.cpp:42-85)
HostScene load_scene(const Json &root, const std::string &json_path);
// uploads through the C ABI (textures, materials, meshes, light, camera) and builds the accel
int upload_scene(evplp_context *ctx, const HostScene &scene);

// stores `text` as the context's last error (defined in context.cpp; the host side has no access to evplp_context otherwise)
void set_context_error(evplp_context *ctx, const char *text);

std::string dirname_of(const std::string &path);
std::string join_path(const std::string &dir, const std::string &rel);
std::string read_text_file(const std::string &path);

} // namespace evplp
