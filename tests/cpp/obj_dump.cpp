// obj_dump -- loads an OBJ file with include/hagrid/load_obj.h and writes "<count or -1>\n" followed by the raw Tri records
// to stdout (tests/test_obj_loader.py compares them with the reference reader's, tests/golden/obj_golden.npz).
#include <cstdio>
#include <vector>

#include "hagrid/load_obj.h"

int main(int argc, char** argv) {
    if (argc != 2) return 2;
    std::vector<hagrid::Tri> tris;
    const bool ok = hagrid::load_obj_triangles(argv[1], tris);
    std::printf("%d\n", ok ? int(tris.size()) : -1);
    if (ok && !tris.empty()) std::fwrite(tris.data(), sizeof(hagrid::Tri), tris.size(), stdout);
    return 0;
}
