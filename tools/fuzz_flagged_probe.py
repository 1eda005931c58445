"""The cases tools/fuzz_reduced_system.py flags, put through the GPU suite's acceptance rule (tests/test_gpu_fuzz.py _check_case: parameters within 1e-5 or within
20 x the oracle's own movement under a 1e-12 perturbation of its input).   gpurun -- python tools/fuzz_flagged_probe.py <seed> <case> [<case> ...]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as orc
import fuzz_scenes
import test_gpu_fuzz as T
from privacy_preserving_sfm_amd.device import camera_num_params
orc.build()
seed = int(sys.argv[1])
for case in map(int, sys.argv[2:]):
    sc, m = fuzz_scenes.reduced_system_case(seed, case, camera_num_params)
    try:
        drift, explained, _, _ = T._check_case(orc, sc, m, (case, m))
        print("seed %d case %3d: drift %.2e, oracle's own spread %s -> accepted" % (seed, case, drift, "%.2e" % explained if explained is not None else "-"), flush=True)
    except AssertionError as e:
        print("seed %d case %3d: REFUSED %s" % (seed, case, str(e)[:600]), flush=True)
