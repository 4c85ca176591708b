"""mbpo.optimizers — same exports as the reference (mbpo/optimizers/__init__.py:1-6) for the hot path."""
from mbpo.optimizers.base_optimizer import BaseOptimizer
from mbpo.optimizers.policy_optimizers.brax_optimizers import BraxOptimizer, BraxOutput, BraxState, PPOOptimizer, SACOptimizer
from mbpo.optimizers.policy_optimizers.bptt_optimizer import BPTTOptimizer, BPTTState
from mbpo.optimizers.policy_optimizers.sac.sac import SAC
from mbpo.optimizers.trajectory_optimizers.icem_optimizer import iCEMOptimizer, iCemOptimizerState, iCemParams, iCemTO
