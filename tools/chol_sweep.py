"""one-off validation sweep of the dense Cholesky solve over many sizes (every launch-structure variant: odd / even block
counts, deferred pairs, padding): relative residual and agreement of graph replays"""
import numpy as np, sys
from privacy_preserving_sfm_amd import device
rng = np.random.default_rng(0)
worst = 0.0
sizes = list(range(1, 200, 13)) + list(range(200, 6400, 197)) + [2880, 2944, 3008, 3072, 5952, 6016, 6080]
for n in sizes:
    B = rng.normal(size=(n, min(n, 80))); A = B @ B.T + n * np.diag(rng.uniform(0.5, 2.0, n)); b = rng.normal(size=n)
    x, _ = device.dense_cholesky_solve(A, b)
    x2, _ = device.dense_cholesky_solve(A, b, repeat=2)
    r = np.linalg.norm(A @ x - b) / np.linalg.norm(b)
    worst = max(worst, r)
    assert r < 1e-12, (n, r)
    assert np.array_equal(x, x2), n
print("sizes %d, worst relative residual %.2e" % (len(sizes), worst))
