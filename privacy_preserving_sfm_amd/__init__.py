"""MI355X-native line-feature bundle adjustment + P6L RANSAC hot path.

PyTorch-ROCm wheels bundle their own HIP runtime (libamdhip64, same soname as /opt/rocm's).  Whichever copy is
loaded first serves the whole process; torch does not find the GPU if the system copy got there first.  Since
multi-GPU runs use torch.distributed for RCCL plumbing, torch is imported here BEFORE libppsfm_hip.so so that
one runtime — torch's — is shared.  torch is optional: without it the library simply uses /opt/rocm's runtime.
"""
try:  # noqa: SIM105
    import torch as _torch  # noqa: F401
except Exception:  # pragma: no cover - torch is plumbing, not a requirement of the C ABI
    _torch = None
