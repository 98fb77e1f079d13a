"""dMicroLane / dMicroForwardLayer on the reference's import path (road.lane.dmicro_lane; reference
dmicro_lane.py:15-298): the drop-in operator, backed by dhts_micro_step_fwd / dhts_micro_step_bwd."""
import torch as th

from dhts import ops
from road.lane._micro_lane import MicroLane


class dMicroLane(MicroLane):
    """Microscopic lane differentiated with the analytic per-step IDM Jacobians (the tape the kernel writes)."""

    class dLane:
        """One step's tape; `dqs` in the reference's layout [num_vehicle][2][2][2]
        (dqs[a, 0] = d(next a)/d(a), dqs[a, 1] = d(next a)/d(leader of a))."""

        def __init__(self, num_vehicle, tape=None):
            self.num_vehicle = num_vehicle
            self.tape = tape

        @property
        def dqs(self):
            n = self.num_vehicle
            npad = (n + 63) // 64 * 64
            t = self.tape.reshape(2, npad, 4)[:, :n, :]
            return t.permute(1, 0, 2).reshape(n, 2, 2, 2).cpu().numpy()

    def __init__(self, id, lane_length, speed_limit):
        super().__init__(id, lane_length, speed_limit)
        self.d_lane = []

    def clear_gradient(self):
        self.d_lane = []


class dMicroForwardLayer(th.autograd.Function):
    """(lane, p[V+1], s[V+1], delta_time) -> (p'[V], s'[V]); backward -> (None, g_p[V+1], g_s[V+1], None).

    The last input slot is the virtual leader of the head vehicle, so its cotangent reaches the head gap
    (reference dmicro_lane.py:130-153).  A collision (gap < 0) is tolerated like in the reference, which prints
    and zeroes the deltas (_micro_lane.py:151-160); the device records it in the fault word."""

    @staticmethod
    def forward(ctx, lane, p, s, delta_time):
        v = lane.num_vehicle()
        assert p.shape[0] == v + 1 and s.shape[0] == v + 1, "Vehicle number mismatch"
        dev = p.device
        desc = ops.micro_desc(1, v, delta_time)
        params = th.tensor([mv.params() for mv in lane.curr_vehicle], dtype=th.float64, device=dev).t().reshape(6, 1, v).contiguous()
        pd, sd = p.detach(), s.detach()
        # the head gap exactly as compute_state_delta hands it to the IDM (numbers or 0-dim tensors)
        head = th.tensor([[float(lane.head_position_delta), float(lane.head_speed_delta)]], dtype=th.float64, device=dev)
        tape = th.empty(ops.micro_step_tape_numel(desc), dtype=th.float32, device=dev)
        err = ops.new_error_record(dev)
        # (a plain MicroLane in a differentiable episode: the reference steps it in float32 tensor arithmetic, _env.py:484-498)
        tensor_ladder = type(lane).__name__ == "MicroLane" and bool(getattr(lane, "_tensor_ladder", False))
        # (a dMicroLane whose head gap is a tensor: detach_vehicle leaves the gap alone, the reference's head vehicle steps in mixed
        #  float32 / double arithmetic, reference dmicro_lane.py:228-250 with _idm.py:6-50)
        head_tensor = not tensor_ladder and isinstance(lane.head_position_delta, th.Tensor)
        np_, nv_ = ops.micro_step_fwd(desc, pd[:-1].reshape(1, v), sd[:-1].reshape(1, v), params, head, tape=tape, err=err,
                                      tensor_ladder=tensor_ladder, head_tensor=head_tensor)
        code = err.tolist()
        if code[0] == 2:
            print("Collision detected at vehicle %d" % code[3])
            print("Set deltas to 0, but please check traffic flow for unrealistic behavior...")
        lane.d_lane.append(dMicroLane.dLane(v, tape))
        ctx.desc, ctx.tape = desc, tape
        return np_[0], nv_[0]

    @staticmethod
    def backward(ctx, grad_np, grad_ns):
        desc = ctx.desc
        v = desc.capacity
        g_p, g_v, g_virtual = ops.micro_step_bwd(desc, ctx.tape, grad_np.contiguous().reshape(1, v),
                                                 grad_ns.contiguous().reshape(1, v))
        gv = g_virtual[0].float()
        return None, th.cat([g_p[0], gv[0:1]]), th.cat([g_v[0], gv[1:2]]), None
