"""Generates tests/golden/hdr_cases.npz from the REFERENCE's own vendored stb_image v2.27 (oracle/_ref/libref_stb_hdr.so,
built by `make -C oracle ref` from /root/reference/lib/stb_image.h where it lies).  Run in the build container only:

    python tests/golden/make_hdr_golden.py

Each case is the byte content of a Radiance .hdr file (the reference ships none: the files are synthesised here by a small
RGBE writer -- run-length and flat scanlines, both signatures, header variants, malformed inputs on which stb fails
deterministically); the fixture stores the bytes and what HDRTexture's call stbi_loadf(filename, &w, &h, &n, 0)
(texture.h:44) returns: width, height, n and the float32 pixels as bit patterns, or the fact that it returned NULL.
tests/test_hdr.py requires agpt_hdr_parse to reproduce them byte for byte.

Not in the fixture: files that END EARLY.  stb then converts uninitialised stack bytes (flat data) or spins for ever on
zero-length runs (run-length data); agpt_hdr_parse fails with AGPT_ERR_INVALID there, tested without a reference value."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def to_rgbe(rgb):
    """float RGB [H, W, 3] -> uint8 RGBE [H, W, 4] (the usual shared-exponent rounding; any bytes would do for the reader)."""
    rgb = np.asarray(rgb, np.float64)
    m = rgb.max(axis=-1)
    e = np.zeros(m.shape, np.int64)
    nz = m > 1e-32
    mant, ex = np.frexp(m[nz])
    e[nz] = ex + 128
    scale = np.zeros(m.shape)
    scale[nz] = mant * 256.0 / m[nz]
    out = np.zeros(rgb.shape[:-1] + (4,), np.uint8)
    out[..., :3] = np.clip(rgb * scale[..., None], 0, 255).astype(np.uint8)
    out[..., 3] = np.clip(e, 0, 255).astype(np.uint8)
    out[~nz] = 0
    return out


def rle_channel(row, literal_only=False, r=None):
    """One channel of a scanline as (count > 128: run of count - 128 | count <= 128: count literals) records."""
    out = bytearray()
    i, n = 0, len(row)
    while i < n:
        run = 1
        while i + run < n and run < 127 and row[i + run] == row[i]:
            run += 1
        if run >= 3 and not literal_only:
            out += bytes([128 + run, row[i]])
            i += run
            continue
        j = i
        while j < n and j - i < 128:
            k = 1
            while j + k < n and k < 3 and row[j + k] == row[j]:
                k += 1
            if k >= 3 and not literal_only:
                break
            j += 1
        if r is not None and j - i > 1:
            j = i + int(r.randint(1, j - i + 1))     # ragged literal lengths, and zero-length records in between
            if r.randint(4) == 0:
                out += bytes([0])
        out += bytes([j - i]) + bytes(row[i:j])
        i = j
    return bytes(out)


def rle_scanlines(q, literal_only=False, r=None):
    h, w, _ = q.shape
    out = bytearray()
    for y in range(h):
        out += bytes([2, 2, w >> 8, w & 255])
        for c in range(4):
            out += rle_channel(q[y, :, c].tolist(), literal_only, r)
    return bytes(out)


def hdr(q, sig=b"#?RADIANCE", lines=(b"FORMAT=32-bit_rle_rgbe",), dims=None, flat=False, literal_only=False, r=None):
    h, w, _ = q.shape
    head = sig + b"\n" + b"".join(l + b"\n" for l in lines) + b"\n" + (dims if dims is not None else b"-Y %d +X %d" % (h, w)) + b"\n"
    return head + (q.tobytes() if flat else rle_scanlines(q, literal_only, r))


def cases():
    r = np.random.RandomState(20260306)
    sky = lambda h, w: to_rgbe(np.concatenate([r.uniform(0, 2, (h // 2, w, 3)), np.full((h - h // 2, w, 3), .25)], 0) *
                               (10.0 ** r.uniform(-3, 3, (h, 1, 1))))
    noise = lambda h, w: r.randint(0, 256, (h, w, 4)).astype(np.uint8)      # every exponent byte incl. 0 (black) and 255
    c = []
    c.append(("rle_64x32_sky", hdr(sky(32, 64))))
    c.append(("rle_8x3_minimum_rle_width", hdr(sky(3, 8))))
    c.append(("rle_noise_every_exponent", hdr(noise(16, 256))))
    c.append(("rle_literals_only", hdr(noise(4, 40), literal_only=True)))
    c.append(("rle_ragged_records_and_zero_counts", hdr(sky(8, 300), r=r)))
    c.append(("rle_runs_of_127", hdr(np.tile(np.array([[[200, 100, 50, 130]]], np.uint8), (2, 1000, 1)))))
    c.append(("rle_wide_4096x2", hdr(sky(2, 4096))))
    c.append(("flat_7x5_narrow", hdr(noise(5, 7), flat=True)))                 # width < 8: always flat
    c.append(("flat_1x1", hdr(noise(1, 1), flat=True)))
    q = noise(6, 20)
    q[0, 0] = (130, 7, 9, 128)                                                 # first byte != 2: the whole image is flat
    c.append(("flat_in_an_rle_width", hdr(q, flat=True)))
    q = noise(6, 20)
    q[0, 0] = (2, 2, 200, 128)                                                 # (2, 2, len with bit 7): flat as well
    c.append(("flat_starting_2_2_high_len", hdr(q, flat=True)))
    # two run-length rows, then a row that is not: stb restarts at the top -- pixel 0 = that row's first four bytes, then flat
    # pixels from pixel 1.  The file carries exactly the bytes that needs.
    a, b = noise(2, 12), noise(5, 12)
    b[0, 0] = (9, 9, 9, 140)
    body = rle_scanlines(a) + b.tobytes()[:4] + r.randint(0, 256, (5 * 12 - 1) * 4).astype(np.uint8).tobytes()
    c.append(("rle_then_flat_restart", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 5 +X 12\n" + body))
    c.append(("sig_rgbe", hdr(sky(4, 16), sig=b"#?RGBE")))
    c.append(("header_comments_exposure_and_long_line",
              hdr(sky(4, 16), lines=(b"# made by a test", b"EXPOSURE=2.5", b"X" * 3000, b"FORMAT=32-bit_rle_rgbe", b"GAMMA=1"))))
    c.append(("header_format_after_nul", hdr(sky(2, 9), lines=(b"FORMAT=32-bit_rle_rgbe\0junk",))))   # strcmp stops at the NUL
    c.append(("dims_extra_spaces", hdr(sky(3, 10), dims=b"-Y   3   +X   10  ")))
    c.append(("dims_plus_signs", hdr(sky(3, 10), dims=b"-Y +3 +X +10")))
    c.append(("trailing_bytes_ignored", hdr(sky(3, 10)) + b"trailing"))
    # failures
    c.append(("fail_signature_crlf", b"#?RADIANCE\r\n" + hdr(sky(2, 8))[11:]))
    c.append(("fail_not_an_image", b"not an image at all\n" * 4))
    c.append(("fail_signature_without_newline", b"#?RADIANCE 1.0\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 1\n" + bytes(4)))
    c.append(("fail_no_format", hdr(sky(2, 8), lines=(b"EXPOSURE=1",))))
    c.append(("fail_format_xyze", hdr(sky(2, 8), lines=(b"FORMAT=32-bit_rle_xyze",))))
    c.append(("fail_layout_plus_y", hdr(sky(2, 8), dims=b"+Y 2 +X 8")))
    c.append(("fail_layout_x_first", hdr(sky(2, 8), dims=b"+X 8 -Y 2")))
    c.append(("fail_layout_minus_x", hdr(sky(2, 8), dims=b"-Y 2 -X 8")))
    c.append(("fail_too_large", hdr(noise(1, 1), dims=b"-Y 20000000 +X 8", flat=True)))
    q = sky(2, 16)
    bad = bytearray(hdr(q))
    bad[bad.index(b"\n\n") + 2 + len(b"-Y 2 +X 16\n") + 3] = 17               # scanline says 17 pixels
    c.append(("fail_scanline_length", bytes(bad)))
    head = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 16\n" + bytes([2, 2, 0, 16])
    c.append(("fail_run_too_long", head + bytes([128 + 17, 5]) + bytes(64)))
    c.append(("fail_literals_too_long", head + bytes([10]) + bytes(10) + bytes([7]) + bytes(64)))
    return c


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_stb_hdr.so"))
    L.ref_stbi_loadf.restype = C.POINTER(C.c_float)
    L.ref_stbi_loadf.argtypes = [C.c_char_p] + [C.POINTER(C.c_int)] * 3
    L.ref_stbi_free.argtypes = [C.POINTER(C.c_float)]
    L.ref_stbi_failure_reason.restype = C.c_char_p
    out, names = {}, []
    for name, data in cases():
        with tempfile.NamedTemporaryFile("wb", suffix=".hdr", delete=False) as f:
            f.write(data)
            path = f.name
        w, h, n = C.c_int(0), C.c_int(0), C.c_int(0)
        p = L.ref_stbi_loadf(path.encode(), C.byref(w), C.byref(h), C.byref(n))
        os.unlink(path)
        ok = bool(p)
        assert ok != name.startswith("fail_"), (name, L.ref_stbi_failure_reason())
        names.append(name)
        out[name + "/bytes"] = np.frombuffer(data, np.uint8)
        out[name + "/ok"] = np.int32(ok)
        if ok:
            assert n.value == 3
            px = np.ctypeslib.as_array(p, (h.value * w.value * 3,)).copy()
            L.ref_stbi_free(p)
            out[name + "/size"] = np.array([w.value, h.value], np.int32)
            out[name + "/pixels"] = px.view(np.uint32)
            print("%-42s %5d x %-4d mean %.4g" % (name, w.value, h.value, px.mean()))
        else:
            print("%-42s stb: %s" % (name, L.ref_stbi_failure_reason().decode()))
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "hdr_cases.npz"), **out)


if __name__ == "__main__":
    main()
