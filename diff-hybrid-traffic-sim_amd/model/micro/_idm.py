"""IDM acceleration on the reference's import path (model.micro._idm; reference _idm.py:6-50), evaluated by the
device code of the rollout kernel (dhts_idm_batch), one vehicle per call."""
import torch as th

IDM_DELTA = 4.0


class IDM:

    @staticmethod
    def _solve(a_max, a_pref, v_curr, v_target, pos_delta, vel_delta, min_space, time_pref, delta_time):
        from dhts import device, ops
        row = [float(x) for x in (a_max, a_pref, v_curr, v_target, pos_delta, vel_delta, min_space, time_pref, delta_time)]
        return ops.idm_batch(th.tensor([row], dtype=th.float64, device=device.get()))

    @staticmethod
    def compute_acceleration(a_max, a_pref, v_curr, v_target, pos_delta, vel_delta, min_space, time_pref, delta_time):
        """-> (acc, optimal_spacing, clipped_acceleration, clipped_optimal_spacing)"""
        o = IDM._solve(a_max, a_pref, v_curr, v_target, pos_delta, vel_delta, min_space, time_pref, delta_time)
        return float(o["acc"][0]), float(o["sstar"][0]), bool(o["clipped_acc"][0]), bool(o["clipped_spacing"][0])
