// evplp-render: the command-line driver, restating reflectcuts/main.cpp:87-124 (one argument: the
// scene JSON; default ../scene/conference/conference_ours.json).  Extras: --device N,
// --set '{"key": value}' (merged over the photonfam block), --synth DIR NAME TRIS (write the
// procedural stand-in scene).
#include "../../../include/evplp.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

int main(int argc, const char *argv[]) {
    std::string json = "../scene/conference/conference_ours.json";   // main.cpp:95-98
    std::string overrides; int device = 0;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--device") && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--set") && i + 1 < argc) overrides = argv[++i];
        else if (!std::strcmp(argv[i], "--synth") && i + 3 < argc) {
            int rc = evplp_synth_scene(argv[i + 1], argv[i + 2], std::atoi(argv[i + 3]), 1234u, 1024, 1024);
            if (rc < 0) { std::fprintf(stderr, "evplp_synth_scene failed (%d)\n", rc); return 1; }
            std::printf("wrote %s/%s.json (%d triangles)\n", argv[i + 1], argv[i + 2], rc);
            return 0;
        } else json = argv[i];
    }
    char err[1024] = "";
    int rc = evplp_render_json(json.c_str(), overrides.empty() ? nullptr : overrides.c_str(), device, err, sizeof(err));
    if (rc != EVPLP_OK) { std::fprintf(stderr, "evplp-render: %s (status %d)\n", err, rc); return 1; }
    return 0;
}
