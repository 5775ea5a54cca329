// load_obj.h -- Wavefront OBJ reader with the interface and the observable behaviour of the reference's ObjLoader
// (src/load_obj.h:13-95, src/load_obj.cpp:78-239) and the triangle fan of its front-end (src/main.cpp:246-275).
//
// Host-only and outside the GPU hot path: it exists so that real scenes reach build_grid exactly as
// the reference would hand them over.  Header-only; written against the behaviour, not the text, of the reference, and
// pinned to it: tests/golden/obj_golden.npz holds the triangle arrays the reference's own load_obj.cpp + fan produce for the
// fixtures in tests/golden/obj/ (generated in the authoring container by tests/golden/make_golden_obj.py), and
// tests/test_obj_loader.py requires this reader to reproduce them bit for bit.
//
// Behaviour that is part of the contract (each item is what load_obj.cpp does, line numbers there):
//   * element 0 of vertices / normals / texcoords is a dummy, indices are 1-based, a negative index counts back from the
//     end of the list as it stands when the face is read (:94-97, :175-179);
//   * lines are read into a 1024-byte buffer: a longer line ends the reading silently (:101-102);
//   * leading white space, empty lines and lines starting with '#' are skipped; trailing white space (CR included) is cut (:104-112);
//   * "v", "vn", "vt" take their numbers with strtof, missing numbers read as 0 (:115-148); another "v?" is an error;
//   * a face keeps at most Face::max_indices = 8 corners, the rest of the line is ignored (:155-168); fewer than three
//     corners, a vertex index <= 0 or a negative normal / texture index after the conversion is an error and drops the face;
//   * "g name" opens a group, "o name" an object with one group (:200-208) -- both need an argument: a bare "g" is an unknown
//     command; "usemtl", "mtllib" are recorded, "s" is ignored, anything else is an error;
//   * errors do not stop the reading, but load_obj returns false if there was any (:238) and the front-end then refuses
//     the scene (main.cpp:249-250).
// Deviation: a vertex index beyond the end of the list is an error here (the reference stores it and reads out of bounds
// later, main.cpp:256).  load_mtl (load_obj.cpp:241-361) is pinned the same way: tests/golden/mtl_golden.npz.
#ifndef HAGRID_LOAD_OBJ_H
#define HAGRID_LOAD_OBJ_H

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "prims.h"
#include "vec.h"

namespace hagrid {

class ObjLoader {
public:
    struct Index { int v, n, t; };

    struct Face {
        static constexpr int max_indices = 8;
        Index indices[max_indices];
        int index_count;
        int material;
    };

    struct Group { std::vector<Face> faces; };
    struct Object { std::vector<Group> groups; };

    struct Material {
        vec3 ka = vec3(0.0f), kd = vec3(0.0f), ks = vec3(0.0f), ke = vec3(0.0f);
        float ns = 0.0f, ni = 0.0f;
        vec3 tf = vec3(0.0f);
        float tr = 0.0f, d = 0.0f;
        int illum = 0;
        std::string map_ka, map_kd, map_ks, map_ke, map_bump, map_d;
    };

    struct File {
        std::vector<Object>      objects;
        std::vector<vec3>        vertices;
        std::vector<vec3>        normals;
        std::vector<vec2>        texcoords;
        std::vector<std::string> materials;
        std::vector<std::string> mtl_libs;
    };

    struct Path {
        Path() {}
        Path(const char* p) : Path(std::string(p)) {}
        Path(const std::string& p) : path(p) {
            for (char& c : path) if (c == '\\') c = '/';
            const size_t cut = path.rfind('/');
            base = cut == std::string::npos ? std::string(".") : path.substr(0, cut);
            file = cut == std::string::npos ? path : path.substr(cut + 1);
        }
        operator const std::string&() const { return path; }
        std::string path, base, file;
    };

    typedef std::unordered_map<std::string, Material> MaterialLib;

    static bool load_obj(const std::string& path, File& file) {
        std::ifstream in(path, std::ios::binary);
        if (!in) return false;
        std::string text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());

        file.objects.emplace_back();
        file.objects.back().groups.emplace_back();
        file.materials.emplace_back("");
        file.vertices.emplace_back();
        file.normals.emplace_back();
        file.texcoords.emplace_back();
        int material = 0, errors = 0;

        const size_t line_capacity = 1023;          // characters that fit the reference's buffer
        size_t at = 0;
        while (at < text.size()) {
            size_t end = text.find('\n', at);
            const bool last = end == std::string::npos;
            if (last) end = text.size();
            if (end - at > line_capacity) break;     // the reference's getline fails here and the reading ends
            std::string line = text.substr(at, end - at);
            at = last ? end : end + 1;
            const size_t nul = line.find('\0');
            if (nul != std::string::npos) line.resize(nul);

            Cursor c(line);
            c.skip_space();
            if (c.done() || c.peek() == '#') continue;
            c.trim_right();

            const char k0 = c.peek(), k1 = c.peek(1);
            if (k0 == 'v') {
                if (k1 == ' ' || k1 == '\t') {
                    c.advance(1);
                    vec3 v; v.x = c.number(); v.y = c.number(); v.z = c.number();
                    file.vertices.push_back(v);
                } else if (k1 == 'n') {
                    c.advance(2);
                    vec3 n; n.x = c.number(); n.y = c.number(); n.z = c.number();
                    file.normals.push_back(n);
                } else if (k1 == 't') {
                    c.advance(2);
                    vec2 t; t.x = c.number(); t.y = c.number();
                    file.texcoords.push_back(t);
                } else {
                    errors++;
                }
            } else if (k0 == 'f' && is_space(k1)) {
                c.advance(2);
                Face f;
                f.index_count = 0;
                f.material = material;
                while (f.index_count < Face::max_indices && c.corner(f.indices[f.index_count])) f.index_count++;
                bool ok = f.index_count >= 3;
                for (int i = 0; ok && i < f.index_count; i++) {
                    Index& x = f.indices[i];
                    if (x.v < 0) x.v += int(file.vertices.size());
                    if (x.t < 0) x.t += int(file.texcoords.size());
                    if (x.n < 0) x.n += int(file.normals.size());
                }
                for (int i = 0; ok && i < f.index_count; i++) {
                    const Index& x = f.indices[i];
                    ok = x.v > 0 && x.t >= 0 && x.n >= 0 && x.v < int(file.vertices.size());
                }
                if (ok) file.objects.back().groups.back().faces.push_back(f);
                else errors++;
            } else if (k0 == 'g' && is_space(k1)) {
                file.objects.back().groups.emplace_back();
            } else if (k0 == 'o' && is_space(k1)) {
                file.objects.emplace_back();
                file.objects.back().groups.emplace_back();
            } else if (c.keyword("usemtl")) {
                const std::string name = c.word();
                const auto it = std::find(file.materials.begin(), file.materials.end(), name);
                material = int(it - file.materials.begin());
                if (it == file.materials.end()) file.materials.push_back(name);
            } else if (c.keyword("mtllib")) {
                file.mtl_libs.push_back(c.word());
            } else if (k0 == 's' && is_space(k1)) {
                // smoothing groups carry nothing the grid needs
            } else {
                errors++;
            }
        }
        return errors == 0;
    }

    /// Material library reader with the observable behaviour of load_obj.cpp:241-361 (materials take no part in the grid path;
    /// the viewer and load_scene's callers get what the reference would give them):
    ///   * a material exists from its first ATTRIBUTE on (a "newmtl" without attributes leaves no entry), zero-initialised;
    ///     attributes before any "newmtl" belong to the material named "";
    ///   * "newmtl name" for a name the library already holds counts as an error but still selects that material;
    ///   * Ka Kd Ks Ke Tf take three numbers, Ns Ni Tr d one, illum one (read as a float, stored as int); missing numbers are 0;
    ///   * map_Ka map_Kd map_Ks map_Ke map_bump bump map_d take the rest of the line; anything else is an error;
    ///   * errors do not stop the reading; the return value says whether there was none.  Lines as in load_obj (1024-byte buffer).
    static bool load_mtl(const std::string& path, MaterialLib& mtl_lib) {
        std::ifstream in(path, std::ios::binary);
        if (!in) return false;
        std::string text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        int errors = 0;
        std::string name;
        const size_t line_capacity = 1023;
        size_t at = 0;
        while (at < text.size()) {
            size_t end = text.find('\n', at);
            const bool last = end == std::string::npos;
            if (last) end = text.size();
            if (end - at > line_capacity) break;
            std::string line = text.substr(at, end - at);
            at = last ? end : end + 1;
            const size_t nul = line.find('\0');
            if (nul != std::string::npos) line.resize(nul);

            Cursor c(line);
            c.skip_space();
            if (c.done() || c.peek() == '#') continue;
            c.trim_right();
            auto three = [&](vec3& v) { v.x = c.number(); v.y = c.number(); v.z = c.number(); };
            const char k0 = c.peek(), k1 = c.peek(1);
            const bool arg2 = is_space(c.peek(2));                          // a two-letter command followed by white space
            if (c.keyword("newmtl")) {
                c.advance(1);
                name = c.word();
                if (mtl_lib.find(name) != mtl_lib.end()) errors++;
            } else if (k0 == 'K') {
                if (!arg2 || (k1 != 'a' && k1 != 'd' && k1 != 's' && k1 != 'e')) { errors++; continue; }
                Material& m = mtl_lib[name];
                c.advance(3);
                three(k1 == 'a' ? m.ka : (k1 == 'd' ? m.kd : (k1 == 's' ? m.ks : m.ke)));
            } else if (k0 == 'N') {
                if (!arg2 || (k1 != 's' && k1 != 'i')) { errors++; continue; }
                Material& m = mtl_lib[name];
                c.advance(3);
                (k1 == 's' ? m.ns : m.ni) = c.number();
            } else if (k0 == 'T') {
                if (!arg2 || (k1 != 'f' && k1 != 'r')) { errors++; continue; }
                Material& m = mtl_lib[name];
                c.advance(3);
                if (k1 == 'f') three(m.tf); else m.tr = c.number();
            } else if (k0 == 'd' && is_space(k1)) {
                Material& m = mtl_lib[name];
                c.advance(2);
                m.d = c.number();
            } else if (c.keyword("illum")) {
                Material& m = mtl_lib[name];
                c.advance(1);
                m.illum = int(c.number());
            } else if (c.keyword("map_Ka")) { mtl_lib[name].map_ka = c.rest(); }
            else if (c.keyword("map_Kd")) { mtl_lib[name].map_kd = c.rest(); }
            else if (c.keyword("map_Ks")) { mtl_lib[name].map_ks = c.rest(); }
            else if (c.keyword("map_Ke")) { mtl_lib[name].map_ke = c.rest(); }
            else if (c.keyword("map_bump")) { mtl_lib[name].map_bump = c.rest(); }
            else if (c.keyword("bump")) { mtl_lib[name].map_bump = c.rest(); }
            else if (c.keyword("map_d")) { mtl_lib[name].map_d = c.rest(); }
            else errors++;
        }
        return errors == 0;
    }

    static bool load_scene(const Path& path, File& file, MaterialLib& mtl_lib) {
        if (!load_obj(path, file)) return false;
        for (auto& lib : file.mtl_libs) load_mtl(path.base + "/" + lib, mtl_lib);
        return true;
    }

private:
    static bool is_space(char ch) { return std::isspace(static_cast<unsigned char>(ch)) != 0; }

    // a position in one line; reading past the end yields '\0'
    struct Cursor {
        std::string s;
        size_t p = 0;
        explicit Cursor(const std::string& line) : s(line) {}
        bool done() const { return p >= s.size(); }
        char peek(size_t ahead = 0) const { return p + ahead < s.size() ? s[p + ahead] : '\0'; }
        void advance(size_t n) { p = std::min(p + n, s.size()); }
        void skip_space() { while (!done() && is_space(s[p])) p++; }
        void trim_right() {                        // never removes the first character of the command
            while (s.size() > p + 1 && is_space(s.back())) s.pop_back();
        }
        float number() {                           // strtof from here; no number: 0 and the position stays
            const char* b = s.c_str() + p;
            char* e = nullptr;
            const float v = std::strtof(b, &e);
            p += size_t(e - b);
            return v;
        }
        long integer() {
            const char* b = s.c_str() + p;
            char* e = nullptr;
            const long v = std::strtol(b, &e, 10);
            p += size_t(e - b);
            return v;
        }
        bool corner(Index& x) {                    // v, v/t, v//n or v/t/n, white space allowed around the slashes
            skip_space();
            const char ch = peek();
            if (!(std::isdigit(static_cast<unsigned char>(ch)) || ch == '-')) return false;
            x.v = int(integer()); x.t = 0; x.n = 0;
            skip_space();
            if (peek() == '/') {
                advance(1);
                if (peek() != '/') x.t = int(integer());
                skip_space();
                if (peek() == '/') { advance(1); x.n = int(integer()); }
            }
            return true;
        }
        bool keyword(const char* w) {              // the word followed by white space; moves behind the word
            const size_t n = std::strlen(w);
            if (s.compare(p, n, w) != 0 || !is_space(peek(n))) return false;
            advance(n);
            return true;
        }
        std::string rest() {                       // behind the separator that follows a keyword: leading white space cut, up to the line's end
            advance(1);
            skip_space();
            return s.substr(p);
        }
        std::string word() {                       // the next run of non-space characters
            skip_space();
            const size_t b = p;
            while (!done() && !is_space(s[p])) p++;
            return s.substr(b, p - b);
        }
    };
};

/// The triangles of an OBJ scene as the reference's front-end builds them (its static load_model, main.cpp:246-275; named
/// differently here because main.cpp defines its own next to `using namespace hagrid`): every face becomes a fan
/// around its first corner, Tri = {v0, n.x, e1 = v0 - v1, n.y, e2 = v2 - v0, n.z} with n = cross(e1, e2).
inline bool load_obj_triangles(const std::string& file_name, std::vector<Tri>& tris) {
    ObjLoader::File obj;
    ObjLoader::MaterialLib materials;
    if (!ObjLoader::load_scene(file_name, obj, materials)) return false;
    for (const auto& object : obj.objects)
        for (const auto& group : object.groups)
            for (const auto& face : group.faces) {
                const vec3 v0 = obj.vertices[face.indices[0].v];
                for (int i = 1; i + 1 < face.index_count; i++) {
                    const vec3 v1 = obj.vertices[face.indices[i].v], v2 = obj.vertices[face.indices[i + 1].v];
                    const vec3 e1 = v0 - v1, e2 = v2 - v0, n = cross(e1, e2);
                    tris.push_back(Tri(v0, n.x, e1, n.y, e2, n.z));
                }
            }
    return true;
}

} // namespace hagrid

#endif // HAGRID_LOAD_OBJ_H
