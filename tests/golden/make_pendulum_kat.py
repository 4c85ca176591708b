"""Writes pendulum_kat.json: closed-form known answers for PendulumDynamics.next_state
(mbpo/systems/dynamics/pendulum_dynamics.py:29-63) and PendulumReward (rewards/pendulum_reward.py:27-42).

Each case is evaluated here by plain `math` arithmetic written out from the cited lines for that specific input
(g=9.81, m=l=1, dt=0.05, max_speed=8, max_torque=2, angle_cost=1, control_cost=0.02, target=0) — it does not
import oracle/ or the product, so the fixture is an independent pin for both.
"""
import json
import math
from pathlib import Path

G, DT = 9.81, 0.05
K = 3 * G / 2  # 3g/(2l) = 14.715

cases = []

def add(why, x, u, x_next, reward):
    cases.append({"why": why, "x": x, "u": [u], "x_next": x_next, "reward": reward})

# 1. upright, no torque: fixed point, zero reward
add("theta=0, u=0: thdd=0 -> fixed point; r=0", [1.0, 0.0, 0.0], 0.0, [1.0, 0.0, 0.0], 0.0)
# 2. upright, full torque: thdd = 3*2 = 6; thdot' = .3; th' = .015; r = -0.02
add("theta=0, u=1: thdd=6, thdot'=0.3, th'=0.015; r=-0.02*1", [1.0, 0.0, 0.0], 1.0,
    [math.cos(0.015), math.sin(0.015), 0.3], -0.02)
# 3. u=5 is clipped to 1 in the dynamics but NOT in the reward: r = -0.02*25
add("u=5: dynamics clip to 1 (same x' as case 2); reward uses unclipped u: -0.5", [1.0, 0.0, 0.0], 5.0,
    [math.cos(0.015), math.sin(0.015), 0.3], -0.5)
# 4. hanging down: theta = pi; diff = ((pi+pi) mod 2pi) - pi = -pi -> r = -pi^2 ; thdd = K*sin(pi) ~ 0
add("theta=pi, u=0: r=-pi^2; x' ~ [-1, 0, 0] (sin(pi) rounding only)", [-1.0, 0.0, 0.0], 0.0, [-1.0, 0.0, 0.0], -math.pi ** 2)
# 5. speed clip: thdot = 7.9, u=1 -> 8.2 clipped to 8; th' = 0 + 8*.05 = .4 ; r = -(0.1*7.9^2) - 0.02
add("speed clip at +8: thdot 7.9+0.3 -> 8; th'=0.4; r=-(0.1*62.41)-0.02", [1.0, 0.0, 7.9], 1.0,
    [math.cos(0.4), math.sin(0.4), 8.0], -(0.1 * 7.9 ** 2) - 0.02)
# 6. horizontal: theta = pi/2, u=0: thdd = K; thdot' = K*dt; th' = pi/2 + K*dt*dt
w = K * DT
add("theta=pi/2, u=0: thdd=14.715, thdot'=0.73575, th'=pi/2+0.0367875; r=-(pi/2)^2", [0.0, 1.0, 0.0], 0.0,
    [-math.sin(w * DT), math.cos(w * DT), w], -(math.pi / 2) ** 2)
# 7. generic: theta=-3, thdot=-2, u=-0.5: uc=-1; thdd = K*sin(-3) - 3
thdd = K * math.sin(-3.0) + 3.0 * (-1.0)
thd = -2.0 + thdd * DT
th = -3.0 + thd * DT
add("theta=-3, thdot=-2, u=-0.5: uc=-1, thdd=K*sin(-3)-3; r=-(9+0.4)-0.02*0.25", [math.cos(-3.0), math.sin(-3.0), -2.0], -0.5,
    [math.cos(th), math.sin(th), thd], -(9.0 + 0.1 * 4.0) - 0.02 * 0.25)
# 8. negative speed clip
add("speed clip at -8: thdot -7.95, u=-1 -> -8.25 clipped; th'=-0.4", [1.0, 0.0, -7.95], -1.0,
    [math.cos(-0.4), math.sin(-0.4), -8.0], -(0.1 * 7.95 ** 2) - 0.02)

out = {"_doc": __doc__, "params": {"g": G, "m": 1.0, "l": 1.0, "dt": DT, "max_speed": 8.0, "max_torque": 2.0,
                                    "angle_cost": 1.0, "control_cost": 0.02, "target_angle": 0.0},
       "reset": {"x": [-1.0, 0.0, 0.0], "reward": 0.0, "source": "pendulum_system.py:41-46"},
       "cases": cases}
Path(__file__).with_name("pendulum_kat.json").write_text(json.dumps(out, indent=1))
