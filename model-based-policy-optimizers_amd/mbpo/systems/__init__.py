"""mbpo.systems — same exports as the reference (mbpo/systems/__init__.py:1-4) plus the learned-ensemble System."""
from mbpo.systems.base_systems import System, SystemParams, SystemState
from mbpo.systems.dynamics.base_dynamics import Dynamics, Normal
from mbpo.systems.rewards.base_rewards import Reward
from mbpo.systems.pendulum_system import PendulumSystem
from mbpo.systems.ensemble_system import EnsembleDynamics, EnsembleDynamicsParams, EnsembleSystem
from mbpo.systems.rewards.pendulum_reward import PendulumReward, QuadraticReward
