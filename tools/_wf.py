import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from privacy_preserving_sfm_amd.device import dense_cholesky_solve
rng = np.random.default_rng(0)
for n in (1500, 3000, 4030, 5000):
    B = rng.normal(size=(n, 64)); A = B @ B.T + n * np.eye(n); b = rng.normal(size=n)
    out = []
    for mode, wf in (("columns", 6), ("tasks", 6), ("tasks", 9), ("tasks", 12), ("tasks", 16), ("tasks", 24), ("tasks", 99)):
        os.environ["PPSFM_CHOL_MODE"] = mode; os.environ["PPSFM_CHOL_WHOLE_FROM"] = str(wf)
        _, ms = dense_cholesky_solve(A, b, repeat=5)
        out.append("%s/%d %.3f" % (mode[0], wf, ms))
    print(n, "  ".join(out), flush=True)
