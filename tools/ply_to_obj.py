#!/usr/bin/env python3
"""Convert a binary little-endian PLY (x y z nx ny nz rgba + triangle faces) to the OBJ layout the
reference's sibling pair toy.ply / toy.obj uses: one `vn` + `v` pair per vertex, faces `a//a b//b c//c`
in file order, `mtllib` + `usemtl` header.  Floats are written with %.9g so that strtof() returns the
exact binary32 value stored in the PLY.

Used once to make tests/scenes/bunny.obj from the reference's resources/models/bunny.ply, because the
reference's default scene resources/models/bunny.obj is missing from its snapshot.

    python tools/ply_to_obj.py /root/reference/resources/models/bunny.ply tests/scenes/bunny.obj bunny.mtl buffy-blue
"""
import sys

import numpy as np


def read_ply(path):
    with open(path, "rb") as f:
        header = []
        while True:
            line = f.readline().decode("ascii").strip()
            header.append(line)
            if line == "end_header":
                break
        assert "format binary_little_endian 1.0" in header, header
        nv = nf = 0
        vprops = []
        section = None
        for h in header:
            t = h.split()
            if t[:2] == ["element", "vertex"]:
                nv, section = int(t[2]), "v"
            elif t[:2] == ["element", "face"]:
                nf, section = int(t[2]), "f"
            elif t[0] == "property" and section == "v":
                vprops.append((t[2], {"float": "<f4", "uchar": "u1", "int": "<i4"}[t[1]]))
        vdt = np.dtype(vprops)
        verts = np.frombuffer(f.read(nv * vdt.itemsize), dtype=vdt, count=nv)
        fdt = np.dtype([("n", "u1"), ("idx", "<i4", (3,))])
        faces = np.frombuffer(f.read(nf * fdt.itemsize), dtype=fdt, count=nf)
        assert (faces["n"] == 3).all()
    return verts, faces["idx"].copy()


def main():
    src, dst, mtl, mat = sys.argv[1:5]
    verts, faces = read_ply(src)
    with open(dst, "w") as out:
        out.write("# converted from %s by tools/ply_to_obj.py\n" % src.split("/")[-1])
        out.write("# Vertices: %d\n# Faces: %d\n" % (len(verts), len(faces)))
        out.write("mtllib %s\n\n" % mtl)
        for v in verts:
            out.write("vn %.9g %.9g %.9g\n" % (v["nx"], v["ny"], v["nz"]))
            out.write("v %.9g %.9g %.9g\n" % (v["x"], v["y"], v["z"]))
        out.write("usemtl %s\n" % mat)
        for a, b, c in faces + 1:
            out.write("f %d//%d %d//%d %d//%d\n" % (a, a, b, b, c, c))
    print("wrote", dst, len(verts), "vertices", len(faces), "faces")


if __name__ == "__main__":
    main()
