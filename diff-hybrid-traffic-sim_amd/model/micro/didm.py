"""IDM Jacobians on the reference's import path (model.micro.didm; reference didm.py:13-103)."""
from model.micro._idm import IDM


class dIDM(IDM):

    @staticmethod
    def compute_dEgo(a_max, a_pref, v_curr, v_target, pos_delta, vel_delta, min_space, time_pref, optimal_spacing,
                     delta_time, clipped_acceleration, clipped_optimal_spacing):
        """d(p', v')/d(p, v) of the ego vehicle, float32 2x2 tensor.  The clip flags are re-derived on the device."""
        o = IDM._solve(a_max, a_pref, v_curr, v_target, pos_delta, vel_delta, min_space, time_pref, delta_time)
        return o["dEgo"][0].cpu()

    @staticmethod
    def compute_dLeading(a_max, a_pref, v_curr, v_target, pos_delta, vel_delta, min_space, time_pref, optimal_spacing,
                         delta_time, clipped_acceleration, clipped_optimal_spacing):
        o = IDM._solve(a_max, a_pref, v_curr, v_target, pos_delta, vel_delta, min_space, time_pref, delta_time)
        return o["dLeading"][0].cpu()
