"""ctypes binding of libmbpo_hip.so (include/mbpo_hip.h).

The product path has NO CPU fallback: importing this module without the built library, or calling an op
with a non-CUDA(HIP) tensor, raises.  torch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional, Sequence

import torch

MBPO_MAX_LAYERS = 8
ACT_IDS = {"swish": 0, "silu": 0, "relu": 1, "tanh": 2}

SYS_PENDULUM, SYS_ENSEMBLE = 0, 1
SYS_GENERIC = 100      # host-side only: a user-defined System, stepped outside the fused kernel (ops.generic_rollout)
ENS_MEAN, ENS_TS1, ENS_TSINF = 0, 1, 2
REWARD_PENDULUM, REWARD_QUADRATIC = 0, 1

_LIB_PATH = Path(__file__).resolve().parent / "_lib" / "libmbpo_hip.so"


class P2pDesc(C.Structure):
    """mbpo_p2p_desc"""
    _fields_ = [("world", C.c_int32), ("rank", C.c_int32), ("n_max", C.c_int64), ("regions", C.c_void_p * 16)]


class MbpoHipError(RuntimeError):
    pass


class MlpDesc(C.Structure):
    _fields_ = [
        ("params", C.c_void_p),
        ("net_stride", C.c_int64),
        ("n_nets", C.c_int32),
        ("n_layers", C.c_int32),
        ("dims", C.c_int32 * (MBPO_MAX_LAYERS + 1)),
        ("activation", C.c_int32),
    ]


class RolloutDesc(C.Structure):
    _fields_ = [
        ("policy", MlpDesc),
        ("dynamics", MlpDesc),
        ("x_dim", C.c_int32),
        ("u_dim", C.c_int32),
        ("n_envs", C.c_int64),
        ("n_steps", C.c_int32),
        ("episode_length", C.c_int32),
        ("action_repeat", C.c_int32),
        ("system_kind", C.c_int32),
        ("ens_mode", C.c_int32),
        ("ens_predict_delta", C.c_int32),
        ("ens_sample_noise", C.c_int32),
        ("ens_min_std", C.c_float),
        ("reward_kind", C.c_int32),
        ("reward_params", C.c_void_p),
        ("sys_params", C.c_void_p),
        ("norm_mean", C.c_void_p),
        ("norm_std", C.c_void_p),
        ("deterministic", C.c_int32),
        ("ppo_extras", C.c_int32),
        ("env_major", C.c_int32),
        ("action_clip", C.c_float),
        ("actions", C.c_void_p),
        ("policy_noise", C.c_void_p),
        ("model_noise", C.c_void_p),
        ("member_idx", C.c_void_p),
        ("seed", C.c_uint64),
        ("offset", C.c_uint64),
        ("rng_dev", C.c_void_p),
        ("obs", C.c_void_p),
        ("first_obs", C.c_void_p),
        ("steps", C.c_void_p),
        ("done", C.c_void_p),
        ("transitions", C.c_void_p),
        ("row_len", C.c_int32),
    ]


class EpisodeStepDesc(C.Structure):
    """mbpo_episode_step_desc"""
    _fields_ = [("x_dim", C.c_int32), ("u_dim", C.c_int32), ("n_envs", C.c_int64), ("episode_length", C.c_int32),
                ("action_repeat", C.c_int32), ("ppo_extras", C.c_int32), ("env_major", C.c_int32), ("step_index", C.c_int32),
                ("n_steps", C.c_int32), ("action", C.c_void_p), ("raw_action", C.c_void_p), ("log_prob", C.c_void_p),
                ("reward", C.c_void_p), ("x_next", C.c_void_p), ("sys_done", C.c_void_p), ("first_obs", C.c_void_p),
                ("obs", C.c_void_p), ("steps", C.c_void_p), ("done", C.c_void_p), ("transitions", C.c_void_p), ("row_len", C.c_int32)]


class EnsTrainDesc(C.Structure):
    """mbpo_ens_train_desc"""
    _fields_ = [("x_dim", C.c_int32), ("u_dim", C.c_int32), ("dynamics", MlpDesc), ("rows", C.c_void_p), ("row_len", C.c_int32),
                ("next_obs_off", C.c_int32), ("idx", C.c_void_p), ("batch", C.c_int64), ("predict_delta", C.c_int32),
                ("min_std", C.c_float), ("grads", C.c_void_p), ("metrics", C.c_void_p), ("workspace", C.c_void_p)]


class SacDesc(C.Structure):
    _fields_ = [
        ("x_dim", C.c_int32), ("u_dim", C.c_int32),
        ("policy_layers", C.c_int32), ("policy_dims", C.c_int32 * (MBPO_MAX_LAYERS + 1)),
        ("q_layers", C.c_int32), ("q_dims", C.c_int32 * (MBPO_MAX_LAYERS + 1)),
        ("policy_activation", C.c_int32), ("q_activation", C.c_int32),
        ("params", C.c_void_p), ("target_q", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p),
        ("step_count", C.c_void_p), ("grads", C.c_void_p),
        ("workspace", C.c_void_p), ("metrics", C.c_void_p), ("metrics_accum", C.c_void_p),
        ("batch", C.c_void_p), ("batch_size", C.c_int32), ("row_len", C.c_int32),
        ("norm_mean", C.c_void_p), ("norm_std", C.c_void_p),
        ("noise_alpha", C.c_void_p), ("noise_critic", C.c_void_p), ("noise_actor", C.c_void_p),
        ("seed", C.c_uint64), ("offset", C.c_uint64), ("rng_dev", C.c_void_p),
        ("discounting", C.c_float), ("reward_scaling", C.c_float), ("target_entropy", C.c_float), ("tau", C.c_float),
        ("lr_policy", C.c_float), ("lr_q", C.c_float), ("lr_alpha", C.c_float),
        ("wd_policy", C.c_float), ("wd_q", C.c_float), ("wd_alpha", C.c_float), ("max_grad_norm", C.c_float),
        ("grad_scale", C.c_float),
        ("non_equidistant_time", C.c_int32), ("continuous_discounting", C.c_float), ("min_time_between_switches", C.c_float),
        ("max_time_between_switches", C.c_float), ("env_dt", C.c_float),
    ]


class PpoDesc(C.Structure):
    _fields_ = [
        ("x_dim", C.c_int32), ("u_dim", C.c_int32),
        ("policy_layers", C.c_int32), ("policy_dims", C.c_int32 * (MBPO_MAX_LAYERS + 1)),
        ("value_layers", C.c_int32), ("value_dims", C.c_int32 * (MBPO_MAX_LAYERS + 1)),
        ("policy_activation", C.c_int32), ("value_activation", C.c_int32),
        ("params", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("step_count", C.c_void_p), ("grads", C.c_void_p),
        ("workspace", C.c_void_p), ("metrics", C.c_void_p), ("metrics_accum", C.c_void_p),
        ("data", C.c_void_p), ("batch_size", C.c_int32), ("unroll_length", C.c_int32), ("row_len", C.c_int32),
        ("norm_mean", C.c_void_p), ("norm_std", C.c_void_p), ("entropy_noise", C.c_void_p),
        ("seed", C.c_uint64), ("offset", C.c_uint64), ("rng_dev", C.c_void_p),
        ("entropy_cost", C.c_float), ("discounting", C.c_float), ("reward_scaling", C.c_float), ("gae_lambda", C.c_float),
        ("clipping_epsilon", C.c_float), ("normalize_advantage", C.c_int32),
        ("lr", C.c_float), ("wd", C.c_float), ("grad_scale", C.c_float),
    ]


class BpttDesc(C.Structure):
    _fields_ = [
        ("x_dim", C.c_int32), ("u_dim", C.c_int32), ("horizon", C.c_int32),
        ("actor_layers", C.c_int32), ("actor_dims", C.c_int32 * (MBPO_MAX_LAYERS + 1)),
        ("critic_layers", C.c_int32), ("critic_dims", C.c_int32 * (MBPO_MAX_LAYERS + 1)),
        ("actor_activation", C.c_int32), ("critic_activation", C.c_int32),
        ("init_stddev", C.c_float),
        ("actor_params", C.c_void_p), ("target_critic_params", C.c_void_p),
        ("system_kind", C.c_int32), ("dynamics", MlpDesc), ("ens_predict_delta", C.c_int32),
        ("reward_kind", C.c_int32), ("reward_params", C.c_void_p), ("sys_params", C.c_void_p),
        ("state_mean", C.c_void_p), ("state_std", C.c_void_p), ("reward_mean_std", C.c_void_p),
        ("init_states", C.c_void_p), ("n", C.c_int64),
        ("act_noise", C.c_void_p), ("seed", C.c_uint64), ("offset", C.c_uint64), ("rng_dev", C.c_void_p),
        ("discount", C.c_float), ("lambda_", C.c_float), ("ent_coef", C.c_float),
        ("transitions", C.c_void_p), ("lambda_values", C.c_void_p), ("grads", C.c_void_p), ("metrics", C.c_void_p),
        ("workspace", C.c_void_p),
    ]


_lib: Optional[C.CDLL] = None


def lib_path() -> Path:
    return Path(os.environ.get("MBPO_HIP_LIB", str(_LIB_PATH)))


def load() -> C.CDLL:
    """Load the library once.  Raises MbpoHipError if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not p.exists():
        raise MbpoHipError(
            f"{p} not found: build it with `python model-based-policy-optimizers_amd/build.py` "
            "(needs hipcc; this framework has no CPU fallback)")
    lib = C.CDLL(str(p))
    lib.mbpo_version.restype = C.c_int
    lib.mbpo_last_error.restype = C.c_char_p
    lib.mbpo_ensemble_mlp_forward.restype = C.c_int
    lib.mbpo_ensemble_mlp_forward.argtypes = [C.POINTER(MlpDesc), C.c_void_p, C.c_int32, C.c_void_p, C.c_int64,
                                              C.c_void_p]
    lib.mbpo_model_rollout.restype = C.c_int
    lib.mbpo_model_rollout.argtypes = [C.POINTER(RolloutDesc), C.c_void_p]
    _bind_optional(lib)
    _lib = lib
    return lib


def _bind_optional(lib: C.CDLL) -> None:
    """Signatures of the remaining entry points (bound when present so partial builds still load)."""
    i32, i64, u64, f32, vp = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_void_p
    sigs = {
        # name: argtypes (see include/mbpo_hip.h)
        "mbpo_replay_insert": [vp, i64, i32, vp, vp, i64, vp],
        "mbpo_replay_gather": [vp, i64, i32, vp, vp, i64, vp, vp],
        "mbpo_replay_sample": [vp, i64, i32, vp, u64, u64, vp, i64, vp, vp, vp],
        "mbpo_running_stats_reduce": [vp, i64, i32, i32, i32, vp, vp, vp, i32, vp],
        "mbpo_running_stats_apply": [vp, vp, i32, f32, f32, vp],
        "mbpo_running_stats_update": [vp, i64, i32, i32, i32, vp, vp, vp, f32, f32, vp],
        "mbpo_gae_scan": [vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, f32, i32, vp],
        "mbpo_gae_scan_discounts": [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, i32, vp],
        "mbpo_lambda_return_scan": [vp, vp, vp, i64, i32, f32, f32, i32, vp],
        "mbpo_critic_grads": [vp, i32, i32, vp, i32, vp, i32, vp, vp, i64, vp, vp, vp, vp, vp, vp],
        "mbpo_adamw_step": [vp, vp, vp, vp, vp, i64, f32, f32, f32, i32, vp, f32, vp, vp, vp],
        "mbpo_rng_advance": [vp, u64, vp],
        "mbpo_soft_update": [vp, vp, vp, i64, f32, vp],
        "mbpo_philox_permutation": [u64, u64, vp, i64, vp, vp, vp],
    }
    for name, argtypes in sigs.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = C.c_int
            fn.argtypes = argtypes
    for name in ("mbpo_sac_grads", "mbpo_sac_grad_norms", "mbpo_sac_apply", "mbpo_sac_step", "mbpo_sac_finalize"):
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = C.c_int
            fn.argtypes = [C.POINTER(SacDesc), vp]
    fn = getattr(lib, "mbpo_sac_finalize_advance", None)
    if fn is not None:
        fn.restype = C.c_int
        fn.argtypes = [C.POINTER(SacDesc), vp, u64, vp]
    for name in ("mbpo_ppo_grads", "mbpo_ppo_apply", "mbpo_ppo_step"):
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = C.c_int
            fn.argtypes = [C.POINTER(PpoDesc), vp]
    fn = getattr(lib, "mbpo_ppo_workspace_floats", None)
    if fn is not None:
        fn.restype = C.c_int64
        fn.argtypes = [C.POINTER(PpoDesc)]
    fn = getattr(lib, "mbpo_critic_workspace_floats", None)
    if fn is not None:
        fn.restype = C.c_int64
        fn.argtypes = [i32, i32, vp, i64]
    lib.mbpo_policy_act.restype = C.c_int
    lib.mbpo_policy_act.argtypes = [C.POINTER(MlpDesc), vp, i64, vp, vp, i32, f32, vp, u64, u64, vp, u64, vp, vp, vp, vp, vp]
    lib.mbpo_philox_normal_fill.restype = C.c_int
    lib.mbpo_philox_normal_fill.argtypes = [C.c_uint64, C.c_uint64, vp, C.c_uint32, C.c_uint64, i64, vp, vp]
    lib.mbpo_mlp_vjp_workspace_floats.restype = C.c_int64
    lib.mbpo_mlp_vjp_workspace_floats.argtypes = [C.POINTER(MlpDesc), i64]
    lib.mbpo_mlp_vjp.restype = C.c_int
    lib.mbpo_mlp_vjp.argtypes = [C.POINTER(MlpDesc), vp, i64, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.mbpo_mlp_layered_workspace_floats.restype = C.c_int64
    lib.mbpo_mlp_layered_workspace_floats.argtypes = [C.POINTER(MlpDesc), i64]
    lib.mbpo_mlp_layered_vjp.restype = C.c_int
    lib.mbpo_mlp_layered_vjp.argtypes = [C.POINTER(MlpDesc), vp, i64, vp, vp, vp, vp, vp, vp]
    lib.mbpo_episode_step.restype = C.c_int
    lib.mbpo_episode_step.argtypes = [C.POINTER(EpisodeStepDesc), vp]
    lib.mbpo_running_stats_workspace_floats.restype = C.c_int64
    lib.mbpo_running_stats_workspace_floats.argtypes = [i32]
    lib.mbpo_ens_nll_workspace_floats.restype = C.c_int64
    lib.mbpo_ens_nll_workspace_floats.argtypes = [C.POINTER(EnsTrainDesc)]
    lib.mbpo_ens_nll_grads.restype = C.c_int
    lib.mbpo_ens_nll_grads.argtypes = [C.POINTER(EnsTrainDesc), vp]
    u64 = C.c_uint64
    lib.mbpo_icem_sample.restype = C.c_int
    lib.mbpo_icem_sample.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, u64, u64, vp, vp, vp, vp]
    lib.mbpo_icem_update.restype = C.c_int
    lib.mbpo_icem_update.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, i32, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.mbpo_icem_update_constrained.restype = C.c_int
    lib.mbpo_icem_update_constrained.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, i32, i32, f32, i32, vp, f32, i32, vp, vp, vp, vp, vp, vp,
                                                 vp, vp]
    # one-shot peer-memory all-reduce (csrc/p2p.hip)
    lib.mbpo_p2p_region_bytes.restype = C.c_int64
    lib.mbpo_p2p_region_bytes.argtypes = [i32, i64]
    for name, argtypes in (("mbpo_p2p_alloc", [i64, C.POINTER(C.c_void_p), vp]), ("mbpo_p2p_open", [vp, i32, C.POINTER(C.c_void_p)]),
                           ("mbpo_p2p_close", [vp]), ("mbpo_p2p_free", [vp]),
                           ("mbpo_p2p_all_reduce_sum", [C.POINTER(P2pDesc), vp, i64, vp]),
                           ("mbpo_p2p_status", [C.POINTER(P2pDesc), C.POINTER(C.c_int32)]),
                           ("mbpo_sac_grads_p2p", [C.POINTER(SacDesc), C.POINTER(P2pDesc), vp]),
                           ("mbpo_sac_gather_p2p", [C.POINTER(SacDesc), C.POINTER(P2pDesc), vp]),
                           ("mbpo_sac_grads_exchange_p2p", [C.POINTER(SacDesc), C.POINTER(P2pDesc), vp]),
                           ("mbpo_sac_step_p2p", [C.POINTER(SacDesc), C.POINTER(P2pDesc), vp])):
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = argtypes
    fn = getattr(lib, "mbpo_bptt_actor_grads", None)
    if fn is not None:
        fn.restype = C.c_int
        fn.argtypes = [C.POINTER(BpttDesc), vp]
    fn = getattr(lib, "mbpo_bptt_workspace_floats", None)
    if fn is not None:
        fn.restype = C.c_int64
        fn.argtypes = [C.POINTER(BpttDesc)]
    fn = getattr(lib, "mbpo_sac_grads_phase", None)
    if fn is not None:
        fn.restype = C.c_int
        fn.argtypes = [C.POINTER(SacDesc), i32, vp]
    fn = getattr(lib, "mbpo_sac_workspace_floats", None)
    if fn is not None:
        fn.restype = C.c_int64
        fn.argtypes = [C.POINTER(SacDesc)]
    fn = getattr(lib, "mbpo_sac_control_offset", None)
    if fn is not None:
        fn.restype = C.c_int64
        fn.argtypes = [C.POINTER(SacDesc)]


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().mbpo_last_error().decode("utf-8", "replace")
        raise MbpoHipError(f"{what} failed (rc={rc}): {msg}")


def require_device_tensor(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise MbpoHipError(f"{name}: tensor is on {t.device}; the HIP path needs a GPU tensor (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def current_stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def mlp_desc(params: torch.Tensor, dims: Sequence[int], activation: str = "swish", n_nets: int = 1,
             net_stride: Optional[int] = None) -> MlpDesc:
    """Describe `n_nets` MLPs stored in `params` (flat; see mbpo_hip.h for the layout)."""
    dims = [int(d) for d in dims]
    n_layers = len(dims) - 1
    if not (1 <= n_layers <= MBPO_MAX_LAYERS):
        raise ValueError(f"n_layers={n_layers} outside [1,{MBPO_MAX_LAYERS}]")
    per_net = sum(dims[i] * dims[i + 1] + dims[i + 1] for i in range(n_layers))
    if net_stride is None:
        net_stride = per_net
    if params.numel() < (n_nets - 1) * net_stride + per_net:
        raise ValueError(f"params has {params.numel()} floats; need {(n_nets - 1) * net_stride + per_net}")
    d = MlpDesc()
    d.params = params.data_ptr()
    d.net_stride = net_stride
    d.n_nets = n_nets
    d.n_layers = n_layers
    for i, v in enumerate(dims):
        d.dims[i] = v
    d.activation = ACT_IDS[activation]
    return d
