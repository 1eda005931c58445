"""Readers for tests/golden/support_measurement_* (expected values = the reference's own InlierSupportMeasurer,
src/optim/support_measurement.cc:36-60, compiled in place as oracle/_ref/support_measurement;
generator tests/golden/gen_support_measurement_golden.py)."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _floats(tokens):
    return np.array([float.fromhex(t) for t in tokens], dtype=np.float64)


def read_vectors(path=os.path.join(GOLD, "support_measurement_vectors.txt")):
    thresholds, vectors = None, []
    for line in open(path):
        tok = line.split()
        if not tok:
            continue
        v = _floats(tok[2:])
        assert len(v) == int(tok[1])
        if tok[0] == "T":
            thresholds = v
        else:
            vectors.append(v)
    return thresholds, vectors


def parse_expected(text, num_vectors, num_thresholds):
    """-> evaluate[t][i] = (count, sum), compare[t] = bool matrix [i][j], winner[t]"""
    ev = [[None] * num_vectors for _ in range(num_thresholds)]
    cmp_rows = [[] for _ in range(num_thresholds)]
    win = [None] * num_thresholds
    for line in text.splitlines():
        tok = line.split()
        if tok[0] == "E":
            ev[int(tok[2])][int(tok[1])] = (int(tok[3]), float.fromhex(tok[4]))
        elif tok[0] == "C":
            cmp_rows[int(tok[1])].append([t == "1" for t in tok[2:]])
        elif tok[0] == "W":
            win[int(tok[1])] = int(tok[2])
    cmp = [np.array(r, dtype=bool) for r in cmp_rows]
    assert all(c.shape == (num_vectors, num_vectors) for c in cmp) and None not in win
    return ev, cmp, win


def read_scene(path=os.path.join(GOLD, "support_measurement_scene.json")):
    d = json.load(open(path))
    n = d["n"]
    sc = dict(n=n, lines=_floats(d["lines"]).reshape(n, 3), points=_floats(d["points"]).reshape(n, 3),
              models=_floats(d["models"]).reshape(-1, 3, 4), thresholds=_floats(d["thresholds"]),
              samples=np.array(d["samples"], dtype=np.uint32).reshape(-1, 6))
    sc["residuals"] = _floats(d["residuals"]).reshape(len(sc["models"]), n)
    sc["expected"] = parse_expected(d["expected"], len(sc["models"]), len(sc["thresholds"]))
    return sc


def same_bits(a, b):
    return np.float64(a).view(np.uint64) == np.float64(b).view(np.uint64)
