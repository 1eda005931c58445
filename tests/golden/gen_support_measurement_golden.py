"""Generates tests/golden/support_measurement_{vectors,vectors_expected,scene}.* — run in the BUILD container only.

The expected values come from the reference's OWN InlierSupportMeasurer (src/optim/support_measurement.cc:36-60),
compiled where it lies by `make -C oracle _ref` into oracle/_ref/support_measurement (never copied, git-ignored).

* support_measurement_vectors.txt           input: thresholds + crafted residual vectors (hex floats)
* support_measurement_vectors_expected.txt  what the reference prints for them (Evaluate, Compare matrix, sequential winner)
* support_measurement_scene.json            a seeded P6L scene (lines, points, 3x4 models as hex floats), the residual
                                            vectors of the models (oracle restatement of estimators/utils.cc:40-89 — the
                                            device must reproduce them bit for bit), and the reference's Evaluate / Compare /
                                            winner for those vectors at five thresholds (0, two finite, DBL_MAX, inf)

    python tests/golden/gen_support_measurement_golden.py
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "support_measurement")
DBL_MAX = float(np.finfo(np.float64).max)


def hexes(v):
    return " ".join(float(x).hex() for x in np.asarray(v, dtype=np.float64).ravel())


def write_input(path, thresholds, vectors):
    with open(path, "w") as f:
        f.write("T %d %s\n" % (len(thresholds), hexes(thresholds)))
        for v in vectors:
            f.write(("V %d %s" % (len(v), hexes(v))).rstrip() + "\n")


def run_reference(path):
    return subprocess.run([REF_BIN, path], capture_output=True, text=True, check=True, timeout=120).stdout


def crafted_vectors():
    rng = np.random.default_rng(20260930)
    vs = []
    vs.append(rng.uniform(0, 2e-4, 1000))                              # 0 random around the thresholds
    vs.append(rng.uniform(0, 1e-5, 1000))                              # 1 all inliers at 1.44e-4
    vs.append(rng.uniform(1.0, 2.0, 1000))                             # 2 none
    v = rng.uniform(0, 1e-4, 500); vs.append(v)                        # 3 ties in count with 4, 5: same values,
    vs.append(v[::-1].copy())                                          # 4 ... reversed (another sequential sum)
    w = v.copy(); w[7] = np.nextafter(w[7], 1.0); vs.append(w)         # 5 ... one entry one ulp larger
    v = rng.uniform(0, 1e-4, 300); v[::3] = DBL_MAX; vs.append(v)      # 6 DBL_MAX entries (points behind the camera)
    vs.append(np.full(17, DBL_MAX))                                    # 7 only DBL_MAX
    vs.append(np.array([]))                                            # 8 empty
    vs.append(np.array([1.44e-4, np.nextafter(1.44e-4, 1.0), np.nextafter(1.44e-4, 0.0), 0.0, -0.0]))  # 9 at the boundary (<=)
    vs.append(10.0 ** rng.uniform(-300, 300, 2000))                    # 10 extreme ranges: the sum's rounding order matters
    v = rng.uniform(0, 1e-4, 4096) * (1.0 + 1e-9 * np.arange(4096)); vs.append(v)   # 11 long: sequential != tree sum
    return vs


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref", "libppsfm_oracle.so"])
    thresholds = [0.0, 1.44e-4, 5e-5, DBL_MAX, float("inf")]
    vin = os.path.join(HERE, "support_measurement_vectors.txt")
    write_input(vin, thresholds, crafted_vectors())
    open(os.path.join(HERE, "support_measurement_vectors_expected.txt"), "w").write(run_reference(vin))

    # the geometric case: residual vectors of 3x4 models on a seeded scene
    import oracle_lib as orc
    from privacy_preserving_sfm_amd import synthetic
    orc.build()
    sc = synthetic.make_ransac_scene(600, outlier_ratio=0.4, noise_px=0.5, seed=4242)
    rng = np.random.default_rng(77)
    P = sc["gt_pose"]
    models = [P.copy()]
    for s in (1e-6, 1e-4, 1e-3, 1e-2):                                 # perturbed poses: decreasing support
        for _ in range(2):
            models.append(P + rng.normal(0, s, (3, 4)))
    models.append(P * 2.0)                                             # the same pose scaled: same residuals up to rounding -> near ties
    models.append(-P)                                                  # every point behind the camera: DBL_MAX everywhere
    Q = P.copy(); Q[2, 3] -= 4.0; models.append(Q)                     # part of the points behind the camera
    models.append(models[1].copy())                                    # an exact duplicate: Compare is false both ways
    models = np.array(models)
    res = np.array([orc.line_residuals(sc["lines"], sc["points"], m) for m in models])
    max_residual = sc["max_error"] ** 2
    th = [0.0, max_residual, 0.25 * max_residual, DBL_MAX, float("inf")]
    tmp = os.path.join(HERE, "_scene_vectors.tmp")
    write_input(tmp, th, list(res))
    out = run_reference(tmp)
    os.remove(tmp)
    samples = np.zeros((64, 6), dtype=np.uint32)
    orc.lib().orc_sampler(0, 600, 6, 64, samples.ctypes.data_as(orc.C.POINTER(orc.C.c_uint32)))
    json.dump(dict(
        comment="generated by gen_support_measurement_golden.py; expected = stdout of oracle/_ref/support_measurement "
                "(reference src/optim/support_measurement.cc compiled in place)",
        n=600, lines=[float(x).hex() for x in sc["lines"].ravel()], points=[float(x).hex() for x in sc["points"].ravel()],
        models=[float(x).hex() for x in models.ravel()], thresholds=[float(x).hex() for x in th],
        residuals=[float(x).hex() for x in res.ravel()], samples=[int(x) for x in samples.ravel()],
        expected=out), open(os.path.join(HERE, "support_measurement_scene.json"), "w"))
    print("wrote fixtures:", len(out.splitlines()), "reference lines for the scene")


if __name__ == "__main__":
    main()
