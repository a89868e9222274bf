"""Generates tests/golden/obj_cases.npz from the REFERENCE's own vendored tinyobjloader (oracle/_ref/libref_tinyobj.so, built
by `make -C oracle ref` from /root/reference/tiny_obj_loader.h where it lies).  Run in the build container only:

    python tests/golden/make_obj_golden.py

Each case is an OBJ text; the fixture stores the text and what TriangleMesh::LoadObj gets back from
tinyobj::ObjReader (trianglemesh.cpp:160-205): attrib.vertices / normals / texcoords as float32 bit patterns and the
concatenated {vertex, normal, texcoord} index triplets of all shapes -- or the fact that ParseFromFile failed (the
reference then calls exit(1)).  tests/test_obj.py requires agpt_obj_parse to reproduce them byte for byte."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def _rng_numbers(seed, n):
    """Decimal spellings that stress tryParseDouble (tiny_obj_loader.h:836-960): long fractions (its pow_lut / pow path),
    exponents (ldexp(m * 5^e, e)), signs, leading dots, trailing garbage."""
    r = np.random.RandomState(seed)
    out = []
    for i in range(n):
        k = r.randint(8)
        if k == 0:
            out.append("%d" % r.randint(-1000, 1000))
        elif k == 1:
            out.append("%.*f" % (r.randint(1, 18), r.uniform(-100, 100)))
        elif k == 2:
            out.append("%.*e" % (r.randint(0, 12), r.uniform(-1, 1) * 10.0 ** r.randint(-30, 30)))
        elif k == 3:
            out.append(("%.6f" % r.uniform(0, 1))[1:])          # ".123456"
        elif k == 4:
            out.append("+%.3fE%+d" % (r.uniform(0, 10), r.randint(-12, 12)))
        elif k == 5:
            out.append(("-%.*f" % (r.randint(1, 10), r.uniform(0, 1))).replace("0.", ".", 1))   # "-.5234"
        elif k == 6:
            out.append("%d." % r.randint(0, 99))                  # "5."
        else:
            out.append("%.9g" % np.float32(r.uniform(-1e3, 1e3)))
    return out


def cases():
    c = []
    c.append(("triangle_v_vt_vn", "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nvn 0 0 1\nf 1/1/1 2/2/1 3/3/1\n"))
    c.append(("quad_and_ngons_convex",
              "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 2 0.5 0\nv 0.5 2 0\n"
              "f 1 2 3 4\nf 1 2 5 3 6 4\nf 4 3 2 1\nf 1 2 5 3 4\n"))
    c.append(("concave_polygons",
              "v 0 0 0\nv 2 0 0\nv 2 2 0\nv 1 0.5 0\nv 0 2 0\n"          # arrow head: vertex 4 is reflex
              "v 0 0 1\nv 3 0 1\nv 3 1 1\nv 1 1 1\nv 1 3 1\nv 0 3 1\n"     # L shape
              "f 1 2 3 4 5\nf 6 7 8 9 10 11\nf 11 10 9 8 7 6\n"
              "v 0 0 0\nv 0 0 4\nv 0 4 4\nv 0 2 1\nv 0 4 0\n"              # concave in the yz plane (axes pick)
              "f 12 13 14 15 16\n"))
    c.append(("negative_and_positive_refs",
              "v 0 0 0\nv 1 0 0\nv 1 1 0\nvn 0 0 1\nvt 0.5 0.5\nf -3/-1/-1 -2/1/-1 -1/-1/1\n"
              "v 0 1 0\nvt .25 .75\nvn 0 1 0\nf -4/-2/-2 -3/-2/2 -1/-1/-1 \nf 1/1/1 2/2/1 4/2/2\n"))
    c.append(("mixed_ref_forms__ignore_normals",      # v//vn next to plain v: normal_index -1 -> valid for LoadObj only with ignore_normals
              "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//-1\nf 1 3 4\nf -4//1 -2 -1//1 2\n"))
    c.append(("groups_objects_and_flush_timing",
              "o first\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\ng a b\nusemtl none\ns 1\nf 1 3 4\n"
              "g\nf 5 6 7 8 9\n"                              # forward references: vertices come AFTER the face ...
              "v 0 0 1\nv 2 0 1\nv 2 2 1\nv 1 .5 1\nv 0 2 1\n"  # ... but before the group is flushed (triangulated) at EOF
              "s off\nl 1 2 3\np 4\n"))
    c.append(("forward_refs_flushed_too_early",
              "f 1 2 3 4 5\ng later\nv 0 0 0\nv 2 0 0\nv 2 2 0\nv 1 .5 0\nv 0 2 0\nf 1 2 3\n"))
    c.append(("line_endings_comments_whitespace",
              "# comment\r\n\r\n  v   0\t0  0 \r\n\tv 1 0 0\rv 0 1 0\n   # another\nvt 0 0 0\nvt 1 1\n"
              "f   1/1   2/2\t3/1  \r\nf 1 2\nf 3\nf\n"))
    c.append(("degenerate_and_colinear_polygons",
              "v 0 0 0\nv 1 0 0\nv 2 0 0\nv 3 0 0\nv 3 1 0\nv 0 1 0\nv 0 0 0\n"
              "f 1 2 3 4\nf 1 2 3 4 5 6\nf 1 7 1 7\nf 1 1 1 1 1\nf 2 3 4 5 6 1\n"))
    c.append(("vertex_colors_and_extra_components", "v 0 0 0 1 0 0\nv 1 0 0 0.5\nv 0 1 0 1 1\nvn 0 0 1 9\nvt 0 0 0\nvt 1 0 .5 7\nf 1/1/1 2/2/1 3/1/1\n"))
    nums = _rng_numbers(7, 3 * 400)
    txt = "".join("v %s %s %s\n" % tuple(nums[3 * i:3 * i + 3]) for i in range(400))
    txt += "".join("vn %s %s %s\n" % tuple(_rng_numbers(8 + i, 3)) for i in range(60))
    txt += "".join("vt %s %s\n" % tuple(_rng_numbers(100 + i, 2)) for i in range(60))
    txt += "".join("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (i + 1, i % 60 + 1, i % 60 + 1, i + 2, (i + 1) % 60 + 1, 1, i + 3, 2, 2) for i in range(0, 396, 3))
    c.append(("number_spellings", txt))
    c.append(("number_edge_spellings",
              "v 1e 2e+ -3.e-\nv abc +def .\nv 1.2.3 4..5 6e7e8\nv -0 +0 -0.0\nv 1e400 -1e400 1e-400\n"
              "v 00012.5000 123456789012345678901234567890 0.000000000000000000000000000001\n"
              "v 3.4028235e38 3.4028236e38 1.17549435e-38\nv 1.401298464e-45 0.7e-45 16777217\n"
              "v 0.1 0.2 0.3\nv 1/2 3,4 5;6\nv +.5 -.5e1 .e1\nv 1E5 1e+05 1e-05\nv 9007199254740993 0.30000000000000004 2.5e-1\n"
              "v\nv 1\nv 1 2\nvt\nvn\n"      # a bare `v' / `vt' / `vn' is not a statement (no IS_SPACE after the keyword)
              "f 1 2 3\nf 4 5 6\nf 7 8 9\nf 10 11 12\nf 13 14 15\n"))
    c.append(("zero_index_fails", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 0\n"))
    c.append(("malformed_index_fails", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 x\n"))
    c.append(("zero_index_in_line_fails", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\nl 1 0\n"))
    c.append(("index_spellings_plain", "v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nf +1 2abc 3.9\nf  -4  -3xyz  -1.5e3\n"))
    c.append(("index_spellings_full", "v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvt 0 0\nvn 0 0 1\nvn 1\nvt 7\nf 1/1/1 2/1x/1 4/+2/2junk\nf 1/2/1/9 2/2/2/ 3/1/1//\n"))
    c.append(("trailing_slash_fails", "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nf 1/ 2/ 3/\n"))
    return c


def main():
    so = os.path.join(ROOT, "oracle", "_ref", "libref_tinyobj.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
    L = C.CDLL(so)
    fpp, ipp = C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_int))
    L.ref_tinyobj_load.argtypes = [C.c_char_p, fpp, C.POINTER(C.c_int), fpp, C.POINTER(C.c_int), fpp, C.POINTER(C.c_int), ipp,
                                   C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]
    L.ref_tinyobj_free.argtypes = [C.c_void_p]
    out = {}
    names = []
    for name, text in cases():
        with tempfile.NamedTemporaryFile("wb", suffix=".obj", delete=False) as f:
            f.write(text.encode("latin-1"))
            path = f.name
        v, n, t = C.POINTER(C.c_float)(), C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
        ix = C.POINTER(C.c_int)()
        nv, nn, nt, ni, ns = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        msg = C.create_string_buffer(4096)
        ok = L.ref_tinyobj_load(path.encode(), C.byref(v), C.byref(nv), C.byref(n), C.byref(nn), C.byref(t), C.byref(nt), C.byref(ix),
                                C.byref(ni), C.byref(ns), msg, 4096)
        os.unlink(path)
        names.append(name)
        out[name + "/text"] = np.frombuffer(text.encode("latin-1"), np.uint8)
        out[name + "/ok"] = np.int32(ok)
        if ok:
            out[name + "/vertices"] = np.ctypeslib.as_array(v, (3 * nv.value,)).view(np.uint32).copy() if nv.value else np.zeros(0, np.uint32)
            out[name + "/normals"] = np.ctypeslib.as_array(n, (3 * nn.value,)).view(np.uint32).copy() if nn.value else np.zeros(0, np.uint32)
            out[name + "/texcoords"] = np.ctypeslib.as_array(t, (2 * nt.value,)).view(np.uint32).copy() if nt.value else np.zeros(0, np.uint32)
            out[name + "/indices"] = np.ctypeslib.as_array(ix, (3 * ni.value,)).astype(np.int32).copy() if ni.value else np.zeros(0, np.int32)
            for p in (v, n, t, ix):
                L.ref_tinyobj_free(p)
        print("%-40s ok=%d  v=%d vn=%d vt=%d tri=%d shapes=%d  %s" % (name, ok, nv.value, nn.value, nt.value, ni.value // 3, ns.value,
                                                                      msg.value.decode(errors="replace").strip().replace("\n", " | ")[:90]))
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "obj_cases.npz"), **out)


if __name__ == "__main__":
    main()
