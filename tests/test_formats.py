"""N4: MAVMAP's text formats either side of bundle adjustment (mavmap_amd/formats.py).

The reference holds no sample files and its readers need boost / Eigen / OpenCV (not in this image), so the rules are
pinned here by hand-written files that exercise every branch of reference src/util/io.cc:12-143, :190-296 and the
writers src/sfm/sequential_mapper.cc:1485-1643; helper maths against src/base3d/projection_test.cc:15-28."""
import math

import numpy as np
import pytest

from mavmap_amd import _abi as A
from mavmap_amd import formats as F
from mavmap_amd import synth

IMAGEDATA = """# NAME, ROLL, PITCH, YAW, LAT, LON, ALT, LOCAL_HEIGHT, TX, TY, TZ, CAM_IDX, CAM_MODEL, CAM_PARAMS[]
img0001, 0.01, -0.02, 1.5, 47.1, 8.2, 520.5, 50, 0.0, 0.0, 50.0, 1, pinhole, 600, 600, 376, 240

img0002, 0.0, 0.0, 1.6, 47.1, 8.2, 520.5, 50, 11.0, 0.5, 50.2
 img0003,0,0,1.6,47.1,8.2,520.5,50,22,1,50.1, 2, OpenCV, 600, 601, 376, 240, -0.1, 0.02, 1e-3, -1e-3
img0004, 0, 0, 1.6, 47.1, 8.2, 520.5, 50, 33, 1, 50.1
"""


def test_read_image_data_rules(tmp_path):
    p = tmp_path / "imagedata.txt"
    p.write_text(IMAGEDATA)
    im = F.read_image_data(str(p), "/data/", "pre_", "_suf", ".jpg")
    assert [i["name"] for i in im] == ["img0001", "img0002", " img0003", "img0004"]        # the name is not trimmed
    assert im[0]["path"] == "/data/pre_img0001_suf.jpg"
    assert (im[0]["roll"], im[0]["pitch"], im[0]["yaw"], im[0]["local_height"], im[0]["tz"]) == (0.01, -0.02, 1.5, 50.0, 50.0)
    assert im[0]["camera_idx"] == 1 and im[0]["camera_model"] == "PINHOLE" and im[0]["camera_params"] == [600, 600, 376, 240]
    # no camera fields: the previous line's camera
    assert im[1]["camera_idx"] == 1 and im[1]["camera_params"] == im[0]["camera_params"] and im[1]["tx"] == 11.0
    assert im[2]["camera_idx"] == 2 and im[2]["camera_model"] == "OPENCV" and len(im[2]["camera_params"]) == 8
    assert im[3]["camera_idx"] == 2 and im[3]["camera_model"] == "OPENCV"
    assert F.camera_params_with_code(im[2])[-1] == float(A.MODEL_OPENCV) and len(F.camera_params_with_code(im[2])) == 9


@pytest.mark.parametrize("text,exc,msg", [
    ("a, 0,0,0, 0,0,0, 0, 0,0,0\n", F.DomainError, "first image"),
    ("a, 0,0,0, 0,0,0, 0, 0,0,0, 1, PINHOLE, 1,2,3,4\nb, 0,0,0, 0,0,0, 0, 0,0,0, 1, PINHOLE, 1,2,3,4\n", F.DomainError, "same index"),
    ("a, 0,0,0, 0,0,0, 0, 0,0,0, 1, PINHOLE, 1,2,3\n", F.DomainError, "at least specify 4"),
    ("a, 0,0,0, 0,0,0, 0, 0,0,0, 1\n", F.DomainError, "No camera model"),
    ("a, 0,0,0, 0,0,0, 0, 0,0,0,\n", ValueError, "lexical"),          # trailing comma: CAM_IDX = "" does not cast
    ("a, 0,0,x, 0,0,0, 0, 0,0,0, 1, PINHOLE, 1,2,3,4\n", ValueError, "lexical"),
    ("a, 0,0,0, 0,0,0, 0, 0,0\n", ValueError, "lexical"),
    ("a, 0,0,0, 0,0,0, 0, 0,0,0, 1.5, PINHOLE, 1,2,3,4\n", ValueError, "lexical"),   # lexical_cast<int>("1.5")
])
def test_read_image_data_rejects_what_the_reference_rejects(tmp_path, text, exc, msg):
    p = tmp_path / "bad.txt"
    p.write_text(text)
    with pytest.raises(exc, match=msg):
        F.read_image_data(str(p))


GCP = """#gcp1, 10.5, 20.25, 3
0, 100.5, 200
3, 110, 210.5

##gcp2, 1, 2, 3
1, 5, 6
#gcp3, 4, 5, 6
2, 7, 8
4, 9, 10
"""


def test_read_control_point_data_rules(tmp_path):
    p = tmp_path / "gcp.txt"
    p.write_text(GCP)
    cps = F.read_control_point_data(str(p))
    assert [c["name"] for c in cps] == ["gcp1", "gcp2", "gcp3"]
    assert cps[0]["xyz"] == [10.5, 20.25, 3.0] and not cps[0]["fixed"] and cps[0]["points2D"] == [(0, (100.5, 200.0)), (3, (110.0, 210.5))]
    assert cps[1]["fixed"] and cps[1]["points2D"] == [(1, (5.0, 6.0))]
    assert cps[2]["points2D"] == [(2, (7.0, 8.0)), (4, (9.0, 10.0))]      # last point: appended because its name differs
    # the reference's control flow: a single control point that is not closed by an empty line is lost,
    # a closed one is kept; a control point without observations is an error
    p.write_text("#only, 1, 2, 3\n0, 1, 2\n")
    assert F.read_control_point_data(str(p)) == []
    p.write_text("#only, 1, 2, 3\n0, 1, 2\n\n")
    assert [c["name"] for c in F.read_control_point_data(str(p))] == ["only"]
    p.write_text("#a, 1, 2, 3\n#b, 1, 2, 3\n0, 1, 2\n\n")
    with pytest.raises(F.DomainError):
        F.read_control_point_data(str(p))


def test_euler_helpers_round_trip_like_the_reference_test():
    # src/base3d/projection_test.cc:15-28 (tolerance of the single-precision atan2f)
    for rx, ry, rz in [(0.1, 0.2, 0.3), (-0.5, 0.4, 2.0), (0.0, 0.0, 0.0), (1.0, -1.2, -2.5)]:
        R = F.rot_mat_from_euler_angles(rx, ry, rz)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-14) and abs(np.linalg.det(R) - 1) < 1e-14
        got = F.euler_angles_from_rot_mat(R)
        assert np.allclose(got, (rx, ry, rz), atol=1e-6)
        assert all(v == float(np.float32(v)) for v in got)               # values are floats widened to double
    # imagedata pose -> world->camera pose -> written exterior parameters reproduce the file's numbers
    im = dict(roll=0.02, pitch=-0.03, yaw=1.2, tx=12.5, ty=-3.25, tz=48.0)
    rvec, tvec = F.image_pose(im)
    rx, ry, rz, tx, ty, tz = F.extract_exterior_params(rvec, tvec)
    assert np.allclose((rx, ry, rz), (0.02, -0.03, 1.2), atol=1e-6) and np.allclose((tx, ty, tz), (12.5, -3.25, 48.0), atol=1e-10)


def test_number_format_is_setprecision_12():
    assert F._g(1.0) == "1" and F._g(0.1 + 0.2) == "0.3" and F._g(123456789.123456) == "123456789.123"
    assert F._g(1e-7) == "1e-07" and F._g(-1.5e20) == "-1.5e+20" and F._g(float("nan")) == "nan" and F._g(-1) == "-1"


def _scene_files(tmp_path, p, poses, intr, points, errors, tag):
    """What the mapper writes after BA, from a flat problem: image-data + point-cloud-data."""
    images = [dict(name=f"img{i:04d}", lat=47.0, lon=8.0, alt=500.0, local_height=50.0, camera_idx=int(p.image_camera[i]) + 1,
                   camera_model=F.MODEL_NAME[int(p.camera_model[p.image_camera[i]])]) for i in range(p.num_images)]
    cps = [list(intr[p.image_camera[i]][:A.MODEL_NUM_PARAMS[int(p.camera_model[p.image_camera[i]])]]) +
           [float(p.camera_model[p.image_camera[i]])] for i in range(p.num_images)]
    f_img, f_pts = tmp_path / f"image-data-{tag}.txt", tmp_path / f"point-cloud-data-{tag}.txt"
    F.write_image_data(str(f_img), images, [(poses[i, :3], poses[i, 3:]) for i in range(p.num_images)], cps)
    F.write_point_cloud_data(str(f_pts), points, np.bincount(p.obs_point, minlength=p.num_points), errors)
    return f_img, f_pts


@pytest.mark.gpu
def test_written_files_of_a_gpu_solve_equal_the_oracles(mavba, oracle, tmp_path):
    """BA on the device, the mapper's output files written from the result, read back: the same numbers (12 significant
    digits) as from the oracle's result; the image-data file is valid imagedata input again."""
    from tests.conftest import global_opts
    p = synth.make_config("C3", scale=0.01, seed=12)
    g, o = p.copy(), p.copy()
    eg = np.full(p.num_points, np.nan)
    mavba.bundle_adjustment(g, global_opts(), point3D_errors=eg)
    _, eo = oracle.solve(o, oracle.options(**global_opts()), want_point_errors=True)
    fg = _scene_files(tmp_path, p, g.poses, g.intrinsics, g.points, eg, "gpu")
    fo = _scene_files(tmp_path, p, o.poses, o.intrinsics, o.points, eo, "oracle")
    for a, b in zip(fg, fo):
        ra, rb = F.read_table(str(a)), F.read_table(str(b))
        assert len(ra) == len(rb) > 0
        for x, y in zip(ra, rb):
            assert len(x) == len(y)
            for u, v in zip(x, y):
                try:
                    fu, fv = float(u), float(v)
                except ValueError:
                    assert u == v
                    continue
                assert (math.isnan(fu) and math.isnan(fv)) or abs(fu - fv) <= 2e-6 * max(abs(fv), 1e-3), (u, v)
    # the writer repeats the camera on every line, so with shared cameras the reference's own reader rejects the file
    with pytest.raises(F.DomainError, match="same index"):
        F.read_image_data(str(fg[0]))
    row = F.read_table(str(fg[0]))[3]
    back = dict(zip(("roll", "pitch", "yaw", "lat", "lon", "alt", "local_height", "tx", "ty", "tz"), map(float, row[1:11])))
    rvec, tvec = F.image_pose(back)
    assert np.allclose(np.concatenate([rvec, tvec]), g.poses[3], atol=2e-5)   # (angles went through float32)
    assert row[12] in ("PINHOLE", "OPENCV") and len(row) == 13 + A.MODEL_NUM_PARAMS[F.MODEL_CODE[row[12]]]
    pts = F.read_table(str(fg[1]))
    assert len(pts) == p.num_points and pts[0][3] == "nan" and int(pts[5][6]) == int(np.sum(p.obs_point == 5))
