import numpy as np
from privacy_preserving_sfm_amd import device
rng = np.random.default_rng(0)
n = 400
B = rng.normal(size=(n, n)); A = B @ B.T + n * np.eye(n); b = rng.normal(size=n)
xr = np.linalg.solve(A, b)
L = np.linalg.cholesky(A); y = np.linalg.solve(L, b)
for rep in (1, 2, 2, 3):
    x, ms = device.dense_cholesky_solve(A, b, repeat=rep)
    print("rep", rep, "max err per block", [float(np.abs(x[s:s+64] - xr[s:s+64]).max()) for s in range(0, n, 64)])
# hypotheses for the wrong blocks: x computed with some x_k taken as zero?
x, ms = device.dense_cholesky_solve(A, b, repeat=2)
T = (n + 1 + 63) // 64
N = T * 64
Lp = np.eye(N); Lp[:n, :n] = L
yp = np.zeros(N); yp[:n] = y
def backsub(drop):
    xx = np.zeros(N)
    for j in range(T - 1, -1, -1):
        acc = yp[64*j:64*j+64].copy()
        for k in range(T - 1, j, -1):
            if (j, k) in drop: continue
            acc -= Lp[64*k:64*k+64, 64*j:64*j+64].T @ xx[64*k:64*k+64]
        xx[64*j:64*j+64] = np.linalg.solve(Lp[64*j:64*j+64, 64*j:64*j+64].T, acc)
    return xx[:n]
print("check full", np.abs(backsub(set()) - xr).max())
j = T - 3
for k in range(T - 1, j, -1):
    print("block", j, "if update from", k, "dropped: err", np.abs(backsub({(j, k)})[64*j:64*j+64] - x[64*j:64*j+64]).max())
print("x block", j, x[64*j:64*j+4], "ref", xr[64*j:64*j+4])
