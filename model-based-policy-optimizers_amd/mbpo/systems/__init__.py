"""mbpo.systems — the reference's exports (mbpo/systems/__init__.py:1-4), name for name, plus the learned-ensemble System."""
from mbpo.systems.base_systems import System, SystemState, SystemParams
from mbpo.systems.pendulum_system import PendulumSystem, PendulumDynamics, PendulumReward
from mbpo.systems.dynamics.base_dynamics import DynamicsParams, Dynamics, Normal
from mbpo.systems.rewards.base_rewards import RewardParams, Reward
# not in the reference (its learned model would come from the external `bsm` package, setup.py:22)
from mbpo.systems.ensemble_system import EnsembleDynamics, EnsembleDynamicsParams, EnsembleSystem
from mbpo.systems.rewards.pendulum_reward import QuadraticReward
