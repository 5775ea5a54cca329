// ref_obj_harness.cpp -- TEST INFRASTRUCTURE, built only where the reference checkout exists (oracle/Makefile -> oracle/_ref/,
// never committed, never shipped to the GPU box).  Links the REFERENCE's own OBJ reader (src/load_obj.cpp, compiled from
// where it lies) and applies the triangle fan of the reference's front-end, which lives in main.cpp next to SDL code and
// cannot be compiled here: the loop below follows main.cpp:246-275 statement by statement.
// tests/golden/make_golden_obj.py uses it to produce tests/golden/obj_golden.npz.
#include <string>
#include <vector>

#include "load_obj.h"
#include "prims.h"

using namespace hagrid;

extern "C" int ref_load_model(const char* file_name, float* out, int cap) {
    ObjLoader::File obj_file;                                   // main.cpp:247-250
    ObjLoader::MaterialLib mtl_lib;
    if (!ObjLoader::load_scene(std::string(file_name), obj_file, mtl_lib)) return -1;
    int count = 0;
    for (auto& object : obj_file.objects) {                     // main.cpp:252-272
        for (auto& group : object.groups) {
            for (auto& face : group.faces) {
                auto v0 = obj_file.vertices[face.indices[0].v];
                for (int i = 0; i < face.index_count - 2; i++) {
                    auto v1 = obj_file.vertices[face.indices[i + 1].v];
                    auto v2 = obj_file.vertices[face.indices[i + 2].v];
                    auto e1 = v0 - v1;
                    auto e2 = v2 - v0;
                    auto n = cross(e1, e2);
                    const Tri tri = { v0, n.x, e1, n.y, e2, n.z };
                    if (count < cap) {
                        const float* f = reinterpret_cast<const float*>(&tri);
                        for (int k = 0; k < 12; k++) out[12 * count + k] = f[k];
                    }
                    count++;
                }
            }
        }
    }
    return count;
}
