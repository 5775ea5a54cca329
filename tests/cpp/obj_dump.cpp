// obj_dump -- loads an OBJ file with include/hagrid/load_obj.h and writes "<count or -1>\n" followed by the raw Tri records
// to stdout (tests/test_obj_loader.py compares them with the reference reader's, tests/golden/obj_golden.npz).
// obj_dump --mtl FILE: the material library load_mtl reads, in the text form of oracle/ref_obj_harness.cpp (ref_load_mtl).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "hagrid/load_obj.h"

static void put_floats(std::string& s, const char* key, const float* f, int n) {
    s += key; s += "=";
    for (int i = 0; i < n; i++) { unsigned u; std::memcpy(&u, f + i, 4); char b[16]; std::snprintf(b, sizeof(b), "%s%08x", i ? " " : "", u); s += b; }
    s += "\n";
}

int main(int argc, char** argv) {
    if (argc == 3 && !std::strcmp(argv[1], "--mtl")) {
        hagrid::ObjLoader::MaterialLib lib;
        const bool ok = hagrid::ObjLoader::load_mtl(argv[2], lib);
        std::vector<std::string> names;
        for (auto& kv : lib) names.push_back(kv.first);
        std::sort(names.begin(), names.end());
        std::string s = ok ? "ok=1\n" : "ok=0\n";
        for (auto& n : names) {
            const hagrid::ObjLoader::Material& m = lib[n];
            s += "name=" + n + "\n";
            put_floats(s, "ka", &m.ka.x, 3); put_floats(s, "kd", &m.kd.x, 3); put_floats(s, "ks", &m.ks.x, 3); put_floats(s, "ke", &m.ke.x, 3);
            put_floats(s, "ns", &m.ns, 1); put_floats(s, "ni", &m.ni, 1); put_floats(s, "tf", &m.tf.x, 3); put_floats(s, "tr", &m.tr, 1); put_floats(s, "d", &m.d, 1);
            s += "illum=" + std::to_string(m.illum) + "\n";
            s += "map_ka=" + m.map_ka + "\nmap_kd=" + m.map_kd + "\nmap_ks=" + m.map_ks + "\nmap_ke=" + m.map_ke + "\nmap_bump=" + m.map_bump + "\nmap_d=" + m.map_d + "\n";
        }
        std::fwrite(s.data(), 1, s.size(), stdout);
        return 0;
    }
    if (argc != 2) return 2;
    std::vector<hagrid::Tri> tris;
    const bool ok = hagrid::load_obj_triangles(argv[1], tris);
    std::printf("%d\n", ok ? int(tris.size()) : -1);
    if (ok && !tris.empty()) std::fwrite(tris.data(), sizeof(hagrid::Tri), tris.size(), stdout);
    return 0;
}
