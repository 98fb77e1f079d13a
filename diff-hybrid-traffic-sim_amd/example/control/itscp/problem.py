"""Inflow schedules of the itscp problems (reference example/control/itscp/problem.py:5-81): the horizon is cut into
sessions; in each session one axis (north-south or west-east) carries heavy inflow (0.9..1.0) and the other almost
none (0..0.01), alternating from a random first choice.  Same np.random draw order as the reference."""
import numpy as np


def problem(lane_id, num_timestep, num_session):
    per = num_timestep // num_session
    axis = []
    for i in range(num_session):
        if i == 0:
            axis.append("NS" if np.random.random((1)).item() > 0.5 else "WE")
        else:
            axis.append("WE" if axis[-1] == "NS" else "NS")
    schedule = {}
    for id in lane_id:
        cur = []
        for s in range(num_session):
            r = np.random.random((1)).item()
            heavy = (id.loc in ("north", "south")) if axis[s] == "NS" else (id.loc in ("west", "east"))
            r = 0.9 + r * 0.1 if heavy else 0.0 + r * 0.01
            cur.extend([r] * per)
        schedule[id] = cur[:num_timestep]
    return schedule


def problem_1(lane_id, num_timestep):
    return problem(lane_id, num_timestep, 1)


def problem_2(lane_id, num_timestep):
    return problem(lane_id, num_timestep, 2)


def problem_3(lane_id, num_timestep):
    return problem(lane_id, num_timestep, 3)
