"""Analytic ARZ Jacobians on the reference's import path (model.macro.darz; reference darz.py:12-233).
Evaluated by the same device code the rollout kernel uses (dhts_arz_interface_batch), one interface per call."""
import numpy as np

from model.macro._arz import ARZ, GAMMA


class dARZ(ARZ):

    @staticmethod
    def compute_dLdR(rs, Q_L, Q_R, u_max, gamma=GAMMA):
        """(dQ_0/dQ_L, dQ_0/dQ_R) as float32 2x2 arrays for the Riemann solution `rs` of (Q_L, Q_R)."""
        if rs.case_ind not in (0, 1, 2):
            raise ValueError("Case index of Riemann solution should be 0 (Q_L), 1 (Q_M) or 2 (Q_C)")
        out = rs._detail if getattr(rs, "_detail", None) is not None else ARZ._solve_pair(Q_L, Q_R, u_max)
        return out["dL"][0].cpu().numpy().astype(np.float32), out["dR"][0].cpu().numpy().astype(np.float32)

    @staticmethod
    def flux_prime(q):
        """Jacobian of the flux (r u, y u) wrt (r, y) at state q, float32 2x2."""
        # an interface whose left state is q and whose right state equals it resolves to Q_0 = q (equal speeds)
        out = ARZ._solve_pair(q, q, q.u_max)
        return out["fp"][0].cpu().numpy().astype(np.float32)
