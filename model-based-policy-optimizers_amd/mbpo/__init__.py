"""mbpo — MI355X-native drop-in for the MBPO inner-loop hot path of lasgroup/Model-based-policy-optimizers.

Same package name and public names as the reference (`mbpo.systems`, `mbpo.optimizers`), so switching is a
matter of putting `model-based-policy-optimizers_amd/` on sys.path instead of the reference checkout.
Compute runs in libmbpo_hip.so (hand-written gfx950 HIP kernels behind a C-ABI, include/mbpo_hip.h).
"""
__version__ = "0.1.0"
