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

// The material library the REFERENCE's load_mtl (src/load_obj.cpp:241-361) reads from a file, as text: "ok=<0|1>" and, per material
// in the order of the names, its fields (floats as their bit patterns).  tests/golden/make_golden_obj.py stores the text,
// tests/cpp/obj_dump.cpp prints the same form for include/hagrid/load_obj.h.
#include <algorithm>
#include <cstdio>
#include <cstring>

static void put_floats(std::string& s, const char* key, const float* f, int n) {
    s += key; s += "=";
    for (int i = 0; i < n; i++) { unsigned u; std::memcpy(&u, f + i, 4); char b[16]; std::snprintf(b, sizeof(b), "%s%08x", i ? " " : "", u); s += b; }
    s += "\n";
}

extern "C" int ref_load_mtl(const char* file_name, char* out, int cap) {
    ObjLoader::MaterialLib lib;
    const bool ok = ObjLoader::load_mtl(std::string(file_name), lib);
    std::vector<std::string> names;
    for (auto& kv : lib) names.push_back(kv.first);
    std::sort(names.begin(), names.end());
    std::string s = ok ? "ok=1\n" : "ok=0\n";
    for (auto& n : names) {
        const ObjLoader::Material& m = lib[n];
        s += "name=" + n + "\n";
        put_floats(s, "ka", &m.ka.x, 3); put_floats(s, "kd", &m.kd.x, 3); put_floats(s, "ks", &m.ks.x, 3); put_floats(s, "ke", &m.ke.x, 3);
        put_floats(s, "ns", &m.ns, 1); put_floats(s, "ni", &m.ni, 1); put_floats(s, "tf", &m.tf.x, 3); put_floats(s, "tr", &m.tr, 1); put_floats(s, "d", &m.d, 1);
        s += "illum=" + std::to_string(m.illum) + "\n";
        s += "map_ka=" + m.map_ka + "\nmap_kd=" + m.map_kd + "\nmap_ks=" + m.map_ks + "\nmap_ke=" + m.map_ke + "\nmap_bump=" + m.map_bump + "\nmap_d=" + m.map_d + "\n";
    }
    if (int(s.size()) + 1 > cap) return -1;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return int(s.size());
}
