"""dense Cholesky solve (K3b) at n = 1500 / 3000 / 6000 in both launch structures (PPSFM_CHOL_MODE)"""
import os, sys
import numpy as np
from privacy_preserving_sfm_amd import device
rng = np.random.default_rng(0)
sizes = [int(a) for a in sys.argv[1:]] or [3000, 1500, 6000]
for n in sizes:
    B = rng.normal(size=(n, n)); A = B @ B.T + n * np.eye(n); b = rng.normal(size=n)
    for mode in ("columns", "tasks"):
        os.environ["PPSFM_CHOL_MODE"] = mode
        x, ms = device.dense_cholesky_solve(A, b, repeat=10)
        r = np.linalg.norm(A @ x - b) / np.linalg.norm(b)
        print(f"n={n} {mode:8s}: {ms:.3f} ms per solve, {n**3/3/ms/1e9:.2f} TFLOP/s, rel resid {r:.2e}", flush=True)
