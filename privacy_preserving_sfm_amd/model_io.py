"""Text model IO of the line-feature reconstruction format: cameras.txt / images.txt /
points3D.txt as the reference reads and writes them (src/base/reconstruction.cc:721-1095; images carry
LINES2D[] as (A, B, C, is_aligned, POINT3D_ID), points carry TRACK[] as (IMAGE_ID, line_idx)).  Lets real reference
outputs be used as fixtures for the BundleAdjuster-shaped driver.  Host-only, no device code."""
import os

import numpy as np

from .bundle_adjustment import Camera, FeatureLine, Image, Point3D, Reconstruction, kInvalidPoint3DId

MODEL_NAMES = ["SIMPLE_PINHOLE", "PINHOLE", "SIMPLE_RADIAL", "RADIAL", "OPENCV", "OPENCV_FISHEYE", "FULL_OPENCV", "FOV", "SIMPLE_RADIAL_FISHEYE",
               "RADIAL_FISHEYE", "THIN_PRISM_FISHEYE"]          # base/camera_models.h:189-349, id = index


def _data_lines(path):
    with open(path) as f:
        for line in f:
            yield line.strip()


def read_cameras_text(path):                                    # reconstruction.cc:721-765
    cams = {}
    for line in _data_lines(path):
        if not line or line[0] == "#":
            continue
        it = line.split(" ")
        cid, model = int(it[0]), MODEL_NAMES.index(it[1])
        cam = Camera(cid, model, [float(v) for v in it[4:]])    # Camera checks the parameter count (VerifyParams)
        cam.width, cam.height = int(it[2]), int(it[3])
        cams[cid] = cam
    return cams


def read_images_text(path):                                     # :767-879
    images = {}
    it = _data_lines(path)
    for line in it:
        if not line or line[0] == "#":
            continue
        f = line.split(" ")
        iid = int(f[0])
        img = Image(iid, int(f[8]), [float(v) for v in f[1:5]], [float(v) for v in f[5:8]])
        img.name = f[9] if len(f) > 9 else ""
        img.NormalizeQvec()
        try:
            second = next(it)
        except StopIteration:
            break
        lines = []
        if second:
            g = second.split(" ")
            assert len(g) % 5 == 0, "LINES2D[] as (A, B, C, is_aligned, POINT3D_ID)"
            for k in range(0, len(g), 5):
                d = np.array([np.float32(g[k]), np.float32(g[k + 1]), np.float32(g[k + 2])], dtype=np.float64)   # std::stof: float32 on purpose
                assert g[k + 3] in ("0", "1")
                pid = kInvalidPoint3DId if g[k + 4] == "-1" else int(g[k + 4])
                lines.append(FeatureLine(d / np.linalg.norm(d[:2]), g[k + 3] == "1", pid))
        img.lines = lines
        images[iid] = img
    return images


def read_points3d_text(path):                                   # :881-960
    pts = {}
    for line in _data_lines(path):
        if not line or line[0] == "#":
            continue
        f = line.split(" ")
        pid = int(f[0])
        p = Point3D([float(f[1]), float(f[2]), float(f[3])])
        p.color = (int(f[4]) & 255, int(f[5]) & 255, int(f[6]) & 255)
        p.error = float(f[7])
        rest = [v for v in f[8:] if v != ""]
        p.track = [(int(rest[k]), int(rest[k + 1])) for k in range(0, len(rest) - 1, 2)]
        pts[pid] = p
    return pts


def read_text(path):
    """Reconstruction::ReadText: <path>/cameras.txt, images.txt, points3D.txt"""
    rec = Reconstruction()
    rec.cameras = read_cameras_text(os.path.join(path, "cameras.txt"))
    rec.images = read_images_text(os.path.join(path, "images.txt"))
    rec.points3D = read_points3d_text(os.path.join(path, "points3D.txt"))
    return rec


def _num(v):
    return repr(float(v)) if not float(v).is_integer() else "%.17g" % float(v)     # precision(17): round-trips a double


def write_text(rec, path):                                      # :962-1095
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "cameras.txt"), "w") as f:
        f.write("# Camera list with one line of data per camera:\n#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]\n# Number of cameras: %d\n" % len(rec.cameras))
        for cid, cam in rec.cameras.items():
            f.write(" ".join([str(cid), MODEL_NAMES[cam.model_id], str(getattr(cam, "width", 0)), str(getattr(cam, "height", 0))] + ["%.17g" % v for v in cam.params]) + "\n")
    nobs = sum(sum(1 for l in im.lines if l.HasPoint3D()) for im in rec.images.values())
    with open(os.path.join(path, "images.txt"), "w") as f:
        f.write("# Image list with two lines of data per image:\n#   IMAGE_ID, QW, QX, QY, QZ, TX, TY, TZ, CAMERA_ID, NAME\n#   LINES2D[] as (A, B, C, is_aligned, POINT3D_ID)\n")
        f.write("# Number of images: %d, mean observations per image: %.17g\n" % (len(rec.images), nobs / max(len(rec.images), 1)))
        for iid, im in rec.images.items():
            q = np.asarray(im.qvec, dtype=np.float64)
            n = np.linalg.norm(q)
            q = np.array([1.0, 0, 0, 0]) if n == 0 else q / n
            f.write(" ".join([str(iid)] + ["%.17g" % v for v in q] + ["%.17g" % v for v in im.tvec] + [str(im.camera_id), getattr(im, "name", "")]) + "\n")
            items = []
            for l in im.lines:
                d = l.Line()
                items += ["%.17g" % d[0], "%.17g" % d[1], "%.17g" % d[2], "1" if l.IsAligned() else "0", str(l.point3D_id) if l.HasPoint3D() else "-1"]
            f.write(" ".join(items) + "\n")
    ntrack = sum(len(p.track) for p in rec.points3D.values())
    with open(os.path.join(path, "points3D.txt"), "w") as f:
        f.write("# 3D point list with one line of data per point:\n#   POINT3D_ID, X, Y, Z, R, G, B, ERROR, TRACK[] as (IMAGE_ID, line_idx)\n")
        f.write("# Number of points: %d, mean track length: %.17g\n" % (len(rec.points3D), ntrack / max(len(rec.points3D), 1)))
        for pid, p in rec.points3D.items():
            col = getattr(p, "color", (0, 0, 0))
            head = [str(pid)] + ["%.17g" % v for v in p.xyz] + [str(int(c)) for c in col] + ["%.17g" % getattr(p, "error", -1.0)]
            tr = []
            for (iid, idx) in p.track:
                tr += [str(iid), str(idx)]
            f.write(" ".join(head) + " " + " ".join(tr) + "\n")
