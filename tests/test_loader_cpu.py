"""CPU tests of the host-side ingest behind the C ABI (no GPU work): cloud
readers for the three formats the reference accepts, metric scaling, the voxel
grid against the oracle, config parsing through a failing run."""
import os
import struct

import numpy as np
import pytest

REF_PLY = "/root/reference/example/calib_1/pointcloud/cloud_1.ply"
REF_PCD = "/root/reference/example/example_data/pointcloud/pos_9.pcd"


def write_ply_binary(path, pts, big_endian=False, extra_element=True):
    e = ">" if big_endian else "<"
    hdr = ["ply", "format binary_%s_endian 1.0" % ("big" if big_endian else "little"), "comment test",
           "element vertex %d" % len(pts), "property float x", "property float y", "property float z"]
    if extra_element:
        hdr += ["element camera 1", "property float view_px", "property int viewportx"]
    hdr += ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode())
        f.write(pts.astype(e + "f4").tobytes())
        if extra_element:
            f.write(struct.pack(e + "fi", 1.0, 7))


def write_ply_ascii(path, pts):
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty double x\nproperty double y\nproperty double z\n"
                "property uchar red\nend_header\n" % len(pts))
        for p in pts:
            f.write("%.9g %.9g %.9g 255\n" % tuple(p))


def write_pcd_binary(path, pts):
    # the field table of the reference's example_data clouds: rgb _ x y z _ (32-byte records)
    n = len(pts)
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS rgb _ x y z _\nSIZE 4 1 4 4 4 1\n"
           "TYPE F U F F F U\nCOUNT 1 12 1 1 1 4\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary\n" % (n, n))
    rec = np.zeros((n, 32), dtype=np.uint8)
    rec[:, 16:28] = pts.astype("<f4").view(np.uint8).reshape(n, 12)
    with open(path, "wb") as f:
        f.write(hdr.encode())
        f.write(rec.tobytes())


def write_pcd_ascii(path, pts):
    n = len(pts)
    with open(path, "w") as f:
        f.write("VERSION .7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH %d\nHEIGHT 1\nPOINTS %d\nDATA ascii\n" % (n, n))
        for p in pts:
            f.write("%.9g %.9g %.9g\n" % tuple(p))


def lzf_compress(data: bytes) -> bytes:
    """A small LZF encoder for the tests (format of liblzf): runs of a repeated byte become back
    references of distance 1 (lengths up to 264, so both length encodings occur), everything else literals."""
    out = bytearray()
    lit = bytearray()

    def flush():
        for k in range(0, len(lit), 32):
            piece = lit[k:k + 32]
            out.append(len(piece) - 1)
            out.extend(piece)
        lit.clear()

    i, n = 0, len(data)
    while i < n:
        run = 1
        while i + run < n and data[i + run] == data[i] and run < 265:
            run += 1
        if run >= 4 and (lit or out):                 # a reference needs one byte of history
            flush()
            length = run - 1                          # the first byte goes out as a literal below
            lit.append(data[i])
            flush()
            ln = length - 2
            if ln < 7:
                out.append(ln << 5)
            else:
                out.append(7 << 5)
                out.append(ln - 7)
            out.append(0)                             # distance 1
            i += run
        else:
            lit.append(data[i])
            i += 1
    flush()
    return bytes(out)


def write_pcd_compressed(path, pts):
    # field order as PCL writes an XYZRGB cloud; body = the fields one after the other (SoA), LZF-compressed
    n = len(pts)
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F F\n"
           "COUNT 1 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary_compressed\n" % (n, n))
    p = pts.astype("<f4")
    body = p[:, 0].tobytes() + p[:, 1].tobytes() + p[:, 2].tobytes() + np.zeros(n, "<f4").tobytes()
    comp = lzf_compress(body)
    with open(path, "wb") as f:
        f.write(hdr.encode())
        f.write(struct.pack("<II", len(comp), len(body)))
        f.write(comp)


@pytest.fixture()
def pts():
    rng = np.random.default_rng(12)
    return rng.uniform(-0.4, 0.9, (1000, 3)).astype(np.float32)


def test_ply_readers(ccal, pts, tmp_path):
    p = str(tmp_path / "cloud_1.ply")
    write_ply_binary(p, pts)
    np.testing.assert_array_equal(ccal.read_point_cloud(p), pts)
    write_ply_binary(p, pts, big_endian=True, extra_element=False)
    np.testing.assert_array_equal(ccal.read_point_cloud(p), pts)
    write_ply_ascii(p, pts)
    np.testing.assert_array_equal(ccal.read_point_cloud(p), pts)
    # metric scaling: float -> double divide -> float (helpers.cpp:159-165)
    write_ply_binary(p, pts)
    got = ccal.read_point_cloud(p, "mm")
    np.testing.assert_array_equal(got, (pts.astype(np.float64) / 1000.0).astype(np.float32))


def test_pcd_readers(ccal, pts, tmp_path):
    p = str(tmp_path / "pos_3.pcd")
    write_pcd_binary(p, pts)
    np.testing.assert_array_equal(ccal.read_point_cloud(p), pts)
    write_pcd_ascii(p, pts)
    np.testing.assert_array_equal(ccal.read_point_cloud(p), pts)
    got = ccal.read_point_cloud(p, "cm")
    np.testing.assert_array_equal(got, (pts.astype(np.float64) / 100.0).astype(np.float32))
    # binary_compressed (PCL's usual save format): LZF stream over the field-major body
    q = pts.copy()
    q[100:400, 1] = 0.25                                     # long runs: back references of both length forms
    q[500:520, 2] = 0.0
    write_pcd_compressed(p, q)
    np.testing.assert_array_equal(ccal.read_point_cloud(p), q)
    raw = open(p, "rb").read()
    open(p, "wb").write(raw[:-40])                           # truncated stream
    with pytest.raises(ccal.CcalError):
        ccal.read_point_cloud(p)
    body = raw.index(b"DATA binary_compressed\n") + 23
    open(p, "wb").write(raw[:body + 8] + bytes([0xE0, 0xFF, 0xFF]) + raw[body + 11:])   # a reference before the start
    with pytest.raises(ccal.CcalError):
        ccal.read_point_cloud(p)
    hdr_bad = raw.replace(b"POINTS %d" % len(q), b"POINTS %d" % (len(q) + 1))
    open(p, "wb").write(hdr_bad)                             # POINTS != WIDTH x HEIGHT
    with pytest.raises(ccal.CcalError):
        ccal.read_point_cloud(p)


def test_xyz_reader(ccal, pts, tmp_path):
    p = str(tmp_path / "scan_2.xyz")
    with open(p, "w") as f:
        f.write("header line is skipped\n")
        for i, q in enumerate(pts[:100]):
            f.write("%d %.6f,%.6f,%.6f extra\n" % (i, q[0] * 1000, q[1] * 1000, q[2] * 1000))
        f.write("\nignored after the first empty line\n")
    got = ccal.read_point_cloud(p, "mm")
    want = np.array([[np.float32(float(np.float32(float("%.6f" % (v * 1000)))) / 1000.0) for v in q] for q in pts[:100]],
                    dtype=np.float32)
    np.testing.assert_array_equal(got, want)


def test_ascii_numbers_are_strtof_values(ccal, tmp_path):
    """The text readers take a decimal to the float strtof gives (the reference reads with stof and PCL's
    stream extraction): short and long digit strings, exponents, values next to float midpoints (where
    rounding through double would go wrong), denormals, overflow, signed zeros, inf/nan, hex floats."""
    import ctypes
    libc = ctypes.CDLL(None)
    libc.strtof.restype = ctypes.c_float
    libc.strtof.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
    rng = np.random.default_rng(99)
    toks = ["0", "-0", "-0.0", "+1", "1.", ".5", "-.25e1", "1e0", "1E+2", "1e-3", "3.4028235e38", "3.4028236e38", "1e39",
            "1e-45", "7e-46", "1.17549435e-38", "1.1754942e-38", "123456789012345678901234567890", "0.1", "0.3",
            "16777217", "16777217.0000000001", "33554434.000", "9007199254740993", "1.00000005960464477539",
            "1.00000005960464477540", "1.0000000596046447753906250", "0.000000000000000000000012345", "12345e22", "12345e23",
            "inf", "-inf", "nan", "0x1.8p1", "1e", "1e+", "5e-324", "1.5abc", "00012.50"]
    # midpoints between adjacent floats and their decimal neighbours
    for _ in range(200):
        f = np.float32(rng.uniform(-2, 2) * 10.0 ** rng.integers(-6, 7))
        g = np.nextafter(f, np.float32(np.inf), dtype=np.float32)
        mid = (float(f) + float(g)) / 2.0
        toks += [repr(mid), "%.17g" % mid, "%.20g" % np.nextafter(mid, np.inf), "%.12g" % mid, "%.9g" % f, "%.6e" % f]
    for _ in range(3000):
        digits = int(rng.integers(1, 18))
        m = int(rng.integers(0, 10 ** digits))
        frac = int(rng.integers(0, digits + 1))
        t = str(m).rjust(frac + 1, "0")
        t = (t[:-frac] + "." + t[-frac:]) if frac else t
        if rng.random() < 0.3:
            t += "e%d" % int(rng.integers(-30, 30))
        toks.append(("-" if rng.random() < 0.3 else "") + t)
    toks += [toks[-1]] * ((3 - len(toks) % 3) % 3)
    want = np.array([libc.strtof(t.encode(), None) for t in toks], dtype=np.float32).reshape(-1, 3)
    n = len(want)
    rows = [" ".join(toks[3 * i:3 * i + 3]) for i in range(n)]
    ply = str(tmp_path / "n_1.ply")
    with open(ply, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nend_header\n" % n)
        f.write("\n".join(rows) + "\n")
    pcd = str(tmp_path / "n_1.pcd")
    with open(pcd, "w") as f:
        f.write("VERSION .7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH %d\nHEIGHT 1\nPOINTS %d\nDATA ascii\n" % (n, n))
        f.write("\r\n".join(rows) + "\n")                       # CRLF line ends too
    for path in (ply, pcd):
        got = ccal.read_point_cloud(path)
        np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


def test_unsupported_and_missing_files(ccal, tmp_path):
    with pytest.raises(ccal.CcalError):
        ccal.read_point_cloud(str(tmp_path / "nope_1.ply"))
    bad = tmp_path / "cloud_1.obj"
    bad.write_text("v 0 0 0\n")
    with pytest.raises(ccal.CcalError):
        ccal.read_point_cloud(str(bad))


@pytest.mark.skipif(not os.path.exists(REF_PLY), reason="reference example data not present")
def test_reference_example_files(ccal):
    raw = open(REF_PLY, "rb").read()
    h = raw.index(b"end_header\n") + 11
    want = np.frombuffer(raw[h:h + 12 * 211566], dtype="<f4").reshape(-1, 3)
    np.testing.assert_array_equal(ccal.read_point_cloud(REF_PLY), want)
    pcd = ccal.read_point_cloud(REF_PCD)
    assert pcd.shape == (112220, 3)
    b = open(REF_PCD, "rb").read()
    body = b[b.index(b"DATA binary\n") + 12:]
    rec = np.frombuffer(body[:32 * 112220], dtype=np.uint8).reshape(-1, 32)
    want = rec[:, 16:28].copy().view("<f4").reshape(-1, 3)
    np.testing.assert_array_equal(pcd, want)


def test_voxel_grid_matches_oracle(ccal, oracle, calib1):
    rng = np.random.default_rng(5)
    pts = rng.uniform(-0.2, 0.3, (50000, 3)).astype(np.float32)
    np.testing.assert_array_equal(ccal.voxel_grid(pts, 0.02), oracle.voxel_grid(pts, 0.02))
    sub = calib1["sub_0"]
    np.testing.assert_array_equal(ccal.voxel_grid(sub, 0.02), oracle.voxel_grid(sub, 0.02))
    assert ccal.voxel_grid(np.zeros((0, 3), np.float32)).shape == (0, 3)


def test_depth_cut_compares_in_double(ccal):
    """`pts.z<=0.8` (:167-169) promotes the float z to double: 0.8f = 0.800000011920929 is dropped."""
    z = np.array([0.5, np.float32(0.8), np.nextafter(np.float32(0.8), np.float32(0)), 0.9, np.nan, -1.0, 0.8000001],
                 dtype=np.float32)
    pts = np.stack([np.arange(len(z), dtype=np.float32), np.zeros(len(z), np.float32), z], axis=1)
    got = ccal.filter_z(pts)
    np.testing.assert_array_equal(got[:, 0], [0, 2, 5])
    # 800 mm read with the mm metric lands exactly on 0.8f
    assert np.float32(np.float64(np.float32(800.0)) / 1000.0) == np.float32(0.8)
    assert ccal.filter_z(np.zeros((0, 3), np.float32)).shape == (0, 3)


def test_pose_parser_matches_oracle(ccal, oracle):
    """The product's getTransVector (:70-98) against the oracle's, degree branch included: std::stof
    precision, int truncation of degrees (towards zero), pi = 3.141592, metric scaling."""
    cases = [("0.0204504,0.111258,0.0505291,-3.15376,-0.0280023,1.58043", "m", "rad"),
             ("20.4504, 111.258, 50.5291, -180.7, -1.6, 90.55", "mm", "degree"),
             (" 2.04504,11.1258,5.05291,179.999,-0.999,0.5", "cm", "degree"),
             ("1,2,3,4.9,-4.9,359.2", "m", "degree"),
             ("1e-3,2e-3,-3e-3,1.5e0,-2.25,3", "m", "radian"),
             ("1,2,3,4,5,6", "mm", "rad")]
    for csv, metric, angle in cases:
        scale = {"m": 1.0, "cm": 100.0, "mm": 1000.0}[metric]
        want = oracle.parse_pose_csv(csv, scale=scale, degree=(angle == "degree"))
        np.testing.assert_array_equal(ccal.parse_pose(csv, metric, angle), want)
    got = ccal.parse_pose("0,0,0,4.9,-4.9,359.2", "m", "degree")
    np.testing.assert_array_equal(got[3:], [4 * 3.141592 / 180, -4 * 3.141592 / 180, 359 * 3.141592 / 180])
    with pytest.raises(ccal.CcalError):
        ccal.parse_pose("1,2,3", "m", "rad")


def test_run_config_reports_bad_inputs(ccal, tmp_path):
    res, err = str(tmp_path / "results.txt"), str(tmp_path / "errors.txt")
    with pytest.raises(ccal.CcalError, match="cannot open config"):
        ccal.run_config(str(tmp_path / "missing.xml"), res, err)
    cfg = tmp_path / "c.xml"
    cfg.write_text("<?xml version='1.0'?><data><camera><clouds></clouds></camera></data>")
    with pytest.raises(ccal.CcalError, match="no data.camera.clouds"):
        ccal.run_config(str(cfg), res, err)
    # the reference writes the "Results:" header before reading anything (:571-574)
    assert open(res).read().startswith("Results: ")


def test_write_config_round_trip(ccal, tmp_path):
    src = tmp_path / "in.xml"
    src.write_text("""<?xml version="1.0"?>
<data>
    <camera>
        <transformations>
            <approximate_transformation>
                <angle>degree</angle>
                <metric>mm</metric>
                <value>1,2,3,4,5,6</value>
            </approximate_transformation>
        </transformations>
        <metric>m</metric>
    </camera>
</data>
""")
    pose = np.array([0.0204504123456789, 0.111258, -0.0505291, -3.15376, 1e-9, 1.58043])
    dst = tmp_path / "out.xml"
    ccal.write_config(str(src), str(dst), pose)
    text = dst.read_text()
    assert "<angle>rad</angle>" in text and "<metric>m</metric>" in text and "degree" not in text
    val = text[text.index("<value>") + 7:text.index("</value>")]
    np.testing.assert_array_equal(np.array([float(v) for v in val.split(",")]), pose)   # 17 digits round-trip
    assert text.count("<camera>") == 1 and text.strip().endswith("</data>")
    with pytest.raises(ccal.CcalError):
        ccal.write_config(str(tmp_path / "missing.xml"), str(dst), pose)
