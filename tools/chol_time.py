import numpy as np, time
from privacy_preserving_sfm_amd import device
rng = np.random.default_rng(0)
for n in (3000, 1500, 6000):
    B = rng.normal(size=(n, n)); A = B @ B.T + n * np.eye(n); b = rng.normal(size=n)
    x, ms = device.dense_cholesky_solve(A, b, repeat=5)
    r = np.linalg.norm(A @ x - b) / np.linalg.norm(b)
    print(f"n={n}: {ms:.3f} ms per solve, {n**3/3/ms/1e9:.2f} TFLOP/s, rel resid {r:.2e}")
