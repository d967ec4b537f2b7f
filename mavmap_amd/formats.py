"""MAVMAP's on-disk text formats either side of bundle adjustment, restated without
boost / Eigen / OpenCV so that real MAVMAP data can be run through the stand-alone BA (bench.py --problem, tests).

    read_image_data            imagedata.txt            reference src/util/io.cc:12-143
    write_image_data           image-data-*.txt         reference src/sfm/sequential_mapper.cc:1485-1536
    write_point_cloud_data     point-cloud-data-*.txt   reference src/sfm/sequential_mapper.cc:1539-1643
    read_control_point_data    GCP file                 reference src/util/io.cc:190-296
    image_pose / extract_exterior_params                reference src/base2d/image.cc:30-47, src/base3d/projection.cc:26-104

The parsers keep the reference's rules, including the ones that look like accidents (the image name is not trimmed,
a trailing comma after TZ is an error, a GCP file with a single control point yields nothing, Euler angles go
through single-precision atan2f): files that the reference reads read the same here, files it rejects are rejected.
Numbers are written like `std::setprecision(12)` on a default-format stream writes them ("%.12g").
"""
import math

import numpy as np

from . import _abi as A

MODEL_CODE = {"PINHOLE": A.MODEL_PINHOLE, "OPENCV": A.MODEL_OPENCV, "CATA": A.MODEL_CATA}
MODEL_NAME = {v: k for k, v in MODEL_CODE.items()}


class DomainError(ValueError):
    """std::domain_error of the reference's readers."""


def _num(item, kind=float):
    """boost::trim + boost::lexical_cast: the WHOLE trimmed token must be a number."""
    t = item.strip(" \t\r\n\v\f")
    try:
        if kind is int:
            if not t or t.lstrip("+-") != t.lstrip("+-").strip() or not t.lstrip("+-").isdigit():
                raise ValueError(t)
            return int(t)
        if not t or t.lower().lstrip("+-") in ("infinity",) or any(c in t for c in " \t_"):
            raise ValueError(t)
        return float(t)
    except ValueError:
        raise ValueError(f"bad lexical cast: {item!r}") from None


def read_image_data(path, root_path="", prefix="", suffix="", ext=""):
    """imagedata.txt -> list of dicts. One line per image:
       NAME, ROLL, PITCH, YAW, LAT, LON, ALT, LOCAL_HEIGHT, TX, TY, TZ [, CAM_IDX, CAM_MODEL, CAM_PARAMS...]
    A line without camera fields reuses the previous line's camera (io.cc:86-97).

    Deliberately STRICTER than the reference on malformed lines: a line with fewer than 11 fields raises (the
    reference's std::getline on the exhausted stream leaves the previous token in place, so it silently casts that
    token again and accepts the line), and a trailing comma after TZ is an (empty, hence rejected) CAM_IDX rather than
    a repeat of TZ. Every well-formed file reads the same."""
    images, camera_idxs = [], set()
    with open(path) as fh:
        for line in fh.read().split("\n"):
            if len(line) == 0 or line[0] == "#":
                continue
            items = line.split(",")
            if len(items) < 11:
                raise ValueError(f"bad lexical cast: line has {len(items)} fields: {line!r}")
            im = dict(name=items[0])  # (not trimmed, io.cc:37-38)
            im["path"] = root_path + prefix + im["name"] + suffix + ext
            for k, key in enumerate(("roll", "pitch", "yaw", "lat", "lon", "alt", "local_height", "tx", "ty", "tz")):
                im[key] = _num(items[1 + k])
            if len(items) == 11:
                if not images:
                    raise DomainError("You must specify a camera model for the first image.")
                im["camera_idx"] = images[-1]["camera_idx"]
                im["camera_model"] = images[-1]["camera_model"]
                im["camera_params"] = list(images[-1]["camera_params"])
            else:
                im["camera_idx"] = _num(items[11], int)
                if im["camera_idx"] in camera_idxs:
                    raise DomainError("Two cameras with the same index have been defined.")
                camera_idxs.add(im["camera_idx"])
                # (a line that ends right after CAM_IDX leaves the model empty -> "No camera model specified.")
                im["camera_model"] = items[12].strip(" \t\r\n\v\f").upper() if len(items) > 12 else ""
                im["camera_params"] = [_num(t) for t in items[13:]]
            if im["camera_model"] == "":
                raise DomainError("No camera model specified.")
            if len(im["camera_params"]) < 4:
                raise DomainError("You must at least specify 4 parameters for a camera model: focal_length (fx, fy) "
                                  "and principal point (cx, cy)")
            images.append(im)
    return images


def rot_mat_from_euler_angles(rx, ry, rz):
    """R = Rz Ry Rx (projection.cc:40-53)."""
    cx, sx, cy, sy, cz, sz = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry), math.cos(rz), math.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def euler_angles_from_rot_mat(R):
    """projection.cc:26-37 - through atan2f, i.e. SINGLE precision, NaN -> 0."""
    f = np.float32
    rx = float(np.arctan2(f(R[2, 1]), f(R[2, 2])))
    ry = float(np.arctan2(f(-R[2, 0]), f(math.sqrt(R[2, 1] * R[2, 1] + R[2, 2] * R[2, 2]))))
    rz = float(np.arctan2(f(R[1, 0]), f(R[0, 0])))
    return tuple(0.0 if math.isnan(v) else v for v in (rx, ry, rz))


def _rodrigues(rvec):
    th = float(np.linalg.norm(rvec))
    if th == 0.0:
        return np.eye(3)
    k = np.asarray(rvec, float) / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * (K @ K)


def _log_so3(R):
    from .synth import log_so3
    return log_so3(R)


def image_pose(image):
    """The world->camera pose (rvec, tvec) the mapper starts an image from: Image::proj_matrix() = the INVERSE of
    [R(roll, pitch, yaw) | (tx, ty, tz)] (image.cc:30-47): imagedata.txt holds camera->world orientation and position."""
    R = rot_mat_from_euler_angles(image["roll"], image["pitch"], image["yaw"])
    t = np.array([image["tx"], image["ty"], image["tz"]])
    Rw = R.T
    return _log_so3(Rw), -Rw @ t


def extract_exterior_params(rvec, tvec):
    """(rx, ry, rz, tx, ty, tz) of the inverted pose, as written to image-data-*.txt (projection.cc:90-104)."""
    R = _rodrigues(np.asarray(rvec, float))
    Ri = R.T
    ti = -Ri @ np.asarray(tvec, float)
    return euler_angles_from_rot_mat(Ri) + (float(ti[0]), float(ti[1]), float(ti[2]))


def _g(x):
    """operator<< of a double at std::setprecision(12), default float field."""
    x = float(x)
    if math.isnan(x):
        return "-nan" if math.copysign(1.0, x) < 0 else "nan"
    if math.isinf(x):
        return "inf" if x > 0 else "-inf"
    return "%.12g" % x


def write_image_data(path, images, poses, camera_params):
    """image-data-*.txt: one line per image that has a pose. `images`: the dicts of read_image_data (name, lat, lon,
    alt, local_height, camera_idx, camera_model are copied through), `poses[i]` = (rvec, tvec) or None for images the
    mapper never registered (skipped, sequential_mapper.cc:1497-1502), `camera_params[i]` = FeatureManager's vector
    for the image's camera: K values and the model code LAST (the code is not written)."""
    with open(path, "w") as fh:
        fh.write("# BASENAME, ROLL, PITCH, YAW, LAT, LON, ALT, LOCAL_HEIGHT, TX, TY, TZ, CAM_IDX, CAM_MODEL, CAM_PARAMS[]\n")
        for im, pose, cp in zip(images, poses, camera_params):
            if pose is None:
                continue
            rx, ry, rz, tx, ty, tz = extract_exterior_params(pose[0], pose[1])
            fields = [im["name"], _g(rx), _g(ry), _g(rz), _g(im["lat"]), _g(im["lon"]), _g(im["alt"]), _g(im["local_height"]),
                      _g(tx), _g(ty), _g(tz), str(int(im["camera_idx"])), im["camera_model"]]
            fields += [_g(v) for v in list(cp)[:-1]]
            fh.write(", ".join(fields) + "\n")
        fh.write("\n")


def write_point_cloud_data(path, points3D, track_len, errors, colors=None):
    """point-cloud-data-*.txt: X, Y, Z, MEAN_R, MEAN_G, MEAN_B, TRACK_LEN, MEAN_RESIDUAL. `errors[i]` NaN / None = the
    point has no point3D error -> -1 (sequential_mapper.cc:1603-1609). Without images there are no colours: the
    reference divides a zero colour sum by zero observations there, i.e. writes nan."""
    with open(path, "w") as fh:
        fh.write("# X, Y, Z, MEAN_R, MEAN_G, MEAN_B, TRACK_LEN, MEAN_RESIDUAL\n")
        for i, X in enumerate(points3D):
            e = errors[i] if errors is not None else None
            e = -1.0 if e is None or (isinstance(e, float) and math.isnan(e)) or np.isnan(e) else e
            c = colors[i] if colors is not None else (float("nan"),) * 3
            fh.write(", ".join([_g(X[0]), _g(X[1]), _g(X[2]), _g(c[0]), _g(c[1]), _g(c[2]), str(int(track_len[i])), _g(e)]) + "\n")
        fh.write("\n")


def read_table(path):
    """Rows of a written image-data / point-cloud-data file (comment and empty lines skipped), fields trimmed."""
    rows = []
    with open(path) as fh:
        for line in fh.read().split("\n"):
            if line and line[0] != "#":
                rows.append([t.strip() for t in line.split(",")])
    return rows


def _init_gcp(line):
    fixed = len(line) > 1 and line[1] == "#"
    items = (line[2:] if fixed else line[1:]).split(",")
    if len(items) < 4:
        raise ValueError(f"bad lexical cast: {line!r}")
    return dict(name=items[0].strip(" \t\r\n\v\f"), xyz=[_num(items[1]), _num(items[2]), _num(items[3])], fixed=fixed, points2D=[])


def _gcp_observation(line, cp):
    items = line.split(",")
    if len(items) < 3:
        raise ValueError(f"bad lexical cast: {line!r}")
    idx = _num(items[0], int)
    if idx < 0:
        raise ValueError(f"bad lexical cast: {items[0]!r}")
    cp["points2D"].append((idx, (_num(items[1]), _num(items[2]))))


def read_control_point_data(path):
    """GCP file -> list of dicts(name, xyz, fixed, points2D=[(image_idx, (x, y))]).
        #NAME, X, Y, Z          a control point (## = its coordinates are held fixed in BA)
        IMAGE_IDX, PX, PY       one observation per line; an empty line or the next '#' closes the point
    The control flow is the reference's (io.cc:249-296): the last point of a file that does not end in an empty line is
    appended only if some point came before it and its name differs from that point's."""
    with open(path) as fh:
        lines = fh.read().split("\n")
    if lines and lines[-1] == "":
        lines.pop()  # (std::getline does not produce a final empty line for a trailing newline)
    out, cp, i, n = [], None, 0, len(lines)

    def obs(line):
        if cp is None:
            raise ValueError("observation before the first control point")
        _gcp_observation(line, cp)

    while i < n:
        line = lines[i]; i += 1
        if len(line) == 0:
            continue
        if line[0] == "#":
            cp = _init_gcp(line)
        else:
            obs(line)
        while i < n:
            line = lines[i]; i += 1
            if len(line) == 0 or line[0] == "#":
                if cp is None or len(cp["points2D"]) == 0:
                    raise DomainError("control_point must have at least two points2D.")
                out.append(dict(cp, points2D=list(cp["points2D"])))
                if len(line) > 0 and line[0] == "#":
                    cp = _init_gcp(line)
                break
            obs(line)
    if out and cp is not None and out[-1]["name"] != cp["name"]:
        out.append(dict(cp, points2D=list(cp["points2D"])))
    return out


def camera_params_with_code(image):
    """FeatureManager.camera_params entry of an image's camera: the file's parameters + the model code as a double
    (reference src/sfm/sequential_mapper.cc:954-973)."""
    code = MODEL_CODE.get(image["camera_model"])
    if code is None:
        raise DomainError("Camera model does not exist.")
    return list(image["camera_params"]) + [float(code)]  # (the parameter count is not checked there either)
