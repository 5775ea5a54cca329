"""Writes the OBJ fixtures of tests/golden/obj/ and, from the REFERENCE's own reader, their triangle arrays.

Run in the authoring container only (needs /root/reference: oracle/Makefile compiles the reference's src/load_obj.cpp,
from where it lies, together with oracle/ref_obj_harness.cpp -- the fan of main.cpp:246-275 -- into
oracle/_ref/libhagrid_ref_obj.so):

    make -C oracle && python tests/golden/make_golden_obj.py

Output: tests/golden/obj/*.obj and *.mtl (fixtures written by this script, not taken from anywhere), tests/golden/mtl_golden.npz
(per .mtl fixture the material library the reference's load_mtl reads, as the text ref_load_mtl prints) and tests/golden/obj_golden.npz
with, per fixture, `<name>_ok` (did the reference accept the file) and `<name>_tris` (float32 [n, 12]).  tests/test_obj_loader.py
compares include/hagrid/load_obj.h against them bit for bit.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OBJ = os.path.join(HERE, "obj")
ROOT = os.path.dirname(os.path.dirname(HERE))


def grid_vertices(n, scale=0.37):
    return [(scale * ((i * 7) % 11) - 1.0, 0.25 * ((i * 5) % 13) + 0.125 * i, 1.5 - 0.0625 * ((i * 3) % 17)) for i in range(n)]


def vlines(vs):
    return ["v %.7g %.7g %.7g" % v for v in vs]


def fixtures():
    f = {}
    vs = grid_vertices(12)
    f["tris"] = "\n".join(["# plain triangles"] + vlines(vs) + ["f 1 2 3", "f 4 5 6", "f 7 9 8", "f 10 12 11", "f 1 12 6"]) + "\n"
    f["quads"] = "\n".join(vlines(vs) + ["vn 0 0 1", "vn 0 1 0", "vt 0.5 0.5", "vt 0.25 0.75"] +
                           ["f 1/1/1 2/2/1 3/1/2 4/2/2", "f 5/1/1 6/1/1 7/1/1 8/1/1", "f 9 10 11 12", "f 12/2 11/1 2/2 1/1"]) + "\n"
    vs16 = grid_vertices(16, 0.21)
    f["ngon8"] = "\n".join(vlines(vs16) + ["f 1 2 3 4 5 6 7 8", "f 16 15 14 13 12", "f 9 10 11 12 13 14 15"]) + "\n"
    # the reader keeps eight corners: the ninth and later corners of a face are dropped
    f["ngon9"] = "\n".join(vlines(vs16) + ["f 1 2 3 4 5 6 7 8 9", "f 16 15 14 13 12 11 10 9 8 7 6 5", "f 2 4 6"]) + "\n"
    f["negative"] = "\n".join(vlines(vs[:4]) + ["f -4 -3 -2 -1", "f -1 -2 -4"] + vlines(vs[4:8]) + ["f -4 -3 1 2", "f -8 -1 -5", "f 1 -1 4"]) + "\n"
    f["forms"] = "\r\n".join(["# CRLF line ends, tabs, blanks around the slashes, groups, objects, materials", "mtllib scene.mtl other.mtl", "o first thing"] +
                             vlines(vs[:6]) + ["vn 1 0 0", "vt 0 1", "  \t g   panel  ", "usemtl red", "s 1", "f 1//1 2//1 3//1", "f\t1/1\t2/1\t3/1\t4/1",
                                               "f 1 / 1 / 1   2 / 1 / 1   5/ 1 /1", "o second", "usemtl blue", "s off", "   f 4 5 6   ", "g a b c", "usemtl red", "f 6 5 4 3 2 1", ""]) + "\r\n"
    f["missing_numbers"] = "\n".join(["v 1 2", "v\t3", "v 0.5 -0.5 0.25", "v  1e-3   2E2  -3.5e+1", "v 4 5 6 7 8", "vn 1", "vt", "f 1 2 3", "f 3 4 5"]) + "\n"
    f["errors"] = "\n".join(vlines(vs[:6]) + ["f 1 2 3", "f 1 2", "l 1 2", "g", "f 4 5 6", "vp 0.5", "f 2 4 6"]) + "\n"       # every error is counted, the file is refused
    f["zero_index"] = "\n".join(vlines(vs[:4]) + ["f 1 2 3", "f 0 1 2", "f 2 3 4"]) + "\n"
    f["bad_minus"] = "\n".join(vlines(vs[:4]) + ["f 1 2 3", "f 1 - 2 3", "f 2 3 4"]) + "\n"
    long_comment = "# " + "x" * 1100
    f["long_line"] = "\n".join(vlines(vs[:6]) + ["f 1 2 3", "f 2 3 4", long_comment, "f 4 5 6", "f 1 3 5"]) + "\n"                # the reading ends at the long line
    pad = "f 1 2 3"
    f["line_1023"] = "\n".join(vlines(vs[:6]) + [pad + " " * (1023 - len(pad)), "f 4 5 6", "f 1 3 5" + " " * 1017, "f 2 4 6"]) + "\n"   # 1023 characters fit, 1024 do not
    f["no_final_newline"] = "\n".join(vlines(vs[:5]) + ["f 1 2 3", "f 3 4 5"])
    f["empty"] = "# nothing\n\n"
    return f


def mtl_fixtures():
    """Material libraries for load_mtl (load_obj.cpp:241-361): every command, attributes before any newmtl, a material without
    attributes (leaves no entry), a redefinition (an error that still selects the material), missing numbers, texture names with
    blanks, CRLF, unknown / malformed commands, the 1024-byte line buffer."""
    f = {}
    f["plain"] = "\n".join(["# two materials", "newmtl red", "Ka 0.1 0.2 0.3", "Kd 1 0 0", "Ks 0.5 0.5 0.5", "Ke 0 0 0.25", "Ns 96.078431", "Ni 1.45",
                            "Tf 1 1 1", "Tr 0.25", "d 0.75", "illum 2", "map_Ka amb.png", "map_Kd tex/red diffuse.png", "map_Ks spec.png", "map_Ke glow.png",
                            "map_bump bump.png", "map_d alpha.png", "", "newmtl blue", "Kd 0 0 1", "bump other_bump.png", "illum 7.9"]) + "\n"
    f["forms"] = "\r\n".join(["Kd 0.25 0.5 0.75", "  \t newmtl   first   ignored words  ", "Ka 1", "Kd\t0.5\t0.25", "Ns", "newmtl empty_one", "newmtl second",
                              "d\t0.5", "map_Kd    spaced name.png   ", "newmtl first", "Ks 1 2 3", "illum -3"]) + "\r\n"
    f["errors"] = "\n".join(["newmtl a", "Kd 1 1 1", "Kx 1 2 3", "Ka", "Nq 3", "Ns5", "Tx 1", "dissolve 1", "illumination 2", "map_Ka", "map_Kz x.png", "foo", "newmtl a", "Kd 0 1 0", "d 1"]) + "\n"
    f["long_line"] = "\n".join(["newmtl a", "Kd 1 2 3", "# " + "x" * 1100, "newmtl b", "Kd 3 2 1"]) + "\n"
    f["no_final_newline"] = "newmtl last\nKd 0.125 0.25 0.5"
    f["empty"] = "# nothing\n\n"
    return f


def main():
    os.makedirs(OBJ, exist_ok=True)
    R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libhagrid_ref_obj.so"))
    R.ref_load_model.restype = C.c_int
    R.ref_load_model.argtypes = [C.c_char_p, C.c_void_p, C.c_int]
    out = {}
    for name, text in fixtures().items():
        path = os.path.join(OBJ, name + ".obj")
        with open(path, "wb") as fh:
            fh.write(text.encode("ascii"))
        buf = np.zeros((4096, 12), np.float32)
        n = R.ref_load_model(path.encode(), buf.ctypes.data, buf.shape[0])
        out[name + "_ok"] = np.array(n >= 0)
        out[name + "_tris"] = buf[:max(n, 0)].copy()
        print(f"{name:18s} reference: {'refused' if n < 0 else str(n) + ' triangles'}")
    np.savez_compressed(os.path.join(HERE, "obj_golden.npz"), **out)
    R.ref_load_mtl.restype = C.c_int
    R.ref_load_mtl.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    out = {}
    for name, text in mtl_fixtures().items():
        path = os.path.join(OBJ, name + ".mtl")
        with open(path, "wb") as fh:
            fh.write(text.encode("ascii"))
        buf = C.create_string_buffer(1 << 16)
        n = R.ref_load_mtl(path.encode(), buf, len(buf))
        assert n >= 0
        out[name] = np.frombuffer(buf.raw[:n], dtype=np.uint8).copy()
        print(f"{name:18s} reference: {buf.raw[:5].decode()} {buf.raw[:n].count(b'name=')} material(s)")
    buf = C.create_string_buffer(1 << 16)
    n = R.ref_load_mtl(os.path.join(OBJ, "does_not_exist.mtl").encode(), buf, len(buf))
    out["missing_file"] = np.frombuffer(buf.raw[:n], dtype=np.uint8).copy()
    np.savez_compressed(os.path.join(HERE, "mtl_golden.npz"), **out)


if __name__ == "__main__":
    main()
