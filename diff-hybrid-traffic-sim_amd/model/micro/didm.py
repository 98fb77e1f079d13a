"""IDM Jacobians on the reference's import path (model.micro.didm; reference didm.py:13-103), evaluated by the device code
(dhts_idm_jac_batch) from the CALLER's optimal spacing and clip flags, as the reference's signatures take them: its lane passes the
flags and the spacing of the forward pass (derived from the gap clamped to 1e-5) beside the un-clamped gap (dmicro_lane.py:97)."""
import torch as th

from model.micro._idm import IDM


class dIDM(IDM):

    @staticmethod
    def _jac(a_max, a_pref, v_curr, v_target, pos_delta, vel_delta, min_space, time_pref, optimal_spacing, delta_time,
             clipped_acceleration, clipped_optimal_spacing):
        from dhts import device, ops
        row = [float(x) for x in (a_max, a_pref, v_curr, v_target, pos_delta, vel_delta, min_space, time_pref, optimal_spacing, delta_time,
                                  1.0 if clipped_acceleration else 0.0, 1.0 if clipped_optimal_spacing else 0.0)]
        return ops.idm_jac_batch(th.tensor([row], dtype=th.float64, device=device.get()))

    @staticmethod
    def compute_dEgo(a_max, a_pref, v_curr, v_target, pos_delta, vel_delta, min_space, time_pref, optimal_spacing,
                     delta_time, clipped_acceleration, clipped_optimal_spacing):
        """d(p', v')/d(p, v) of the ego vehicle, float32 2x2 tensor (didm.py:13-56)."""
        return dIDM._jac(a_max, a_pref, v_curr, v_target, pos_delta, vel_delta, min_space, time_pref, optimal_spacing, delta_time,
                         clipped_acceleration, clipped_optimal_spacing)[0][0].cpu()

    @staticmethod
    def compute_dLeading(a_max, a_pref, v_curr, v_target, pos_delta, vel_delta, min_space, time_pref, optimal_spacing,
                         delta_time, clipped_acceleration, clipped_optimal_spacing):
        """d(p', v')/d(p, v) of the leading vehicle (didm.py:59-103)."""
        return dIDM._jac(a_max, a_pref, v_curr, v_target, pos_delta, vel_delta, min_space, time_pref, optimal_spacing, delta_time,
                         clipped_acceleration, clipped_optimal_spacing)[1][0].cpu()
