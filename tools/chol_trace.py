"""per-launch durations of one dense Cholesky solve (run under rocprofv3 --kernel-trace, then summarised by
tools/chol_trace_summary.py): which launches are bound by the chain workgroup and which by the bulk tiles"""
import sys
import numpy as np
from privacy_preserving_sfm_amd import device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(0)
B = rng.normal(size=(n, 64)); A = B @ B.T + n * np.eye(n); b = rng.normal(size=n)
device.dense_cholesky_solve(A, b, repeat=3)
