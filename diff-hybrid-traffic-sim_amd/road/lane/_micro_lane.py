"""MicroLane on the reference's import path (road.lane._micro_lane; reference _micro_lane.py:19-330).

Vehicles stay host-side records (road.vehicle.MicroVehicle) in tail -> head order; their positions and speeds
live in two float32 device arrays that the step operator (road.lane.dmicro_lane.dMicroForwardLayer ->
dhts_micro_step_fwd) reads and writes.  `vehicle.position` / `.speed` are refreshed as 0-dim views of those
arrays after every update, so example-style host code can keep reading them.
"""
import torch as th

from dhts import device
from dmath.operation import sigmoid
from road.lane._base_lane import BaseLane

DEFAULT_HEAD_POSITION_DELTA = 1000
DEFAULT_HEAD_SPEED_DELTA = 0
POSITION_DELTA_EPS = 1e-5


class MicroLane(BaseLane):

    def __init__(self, id, lane_length, speed_limit):
        super().__init__(id, lane_length, speed_limit)
        self.curr_vehicle = []                 # index i follows index i + 1; head = last
        self.acc_info = []
        self.next_vehicle_position = []
        self.next_vehicle_speed = []
        self.head_position_delta = DEFAULT_HEAD_POSITION_DELTA
        self.head_speed_delta = DEFAULT_HEAD_SPEED_DELTA
        self._p = None                         # float32 [V] device arrays (None = rebuild from the vehicle records)
        self._v = None
        self._np = None
        self._nv = None
        self.d_lane = []

    def is_macro(self):
        return False

    def is_micro(self):
        return True

    # ---- vehicle list ---------------------------------------------------------------------------------------
    def num_vehicle(self):
        return len(self.curr_vehicle)

    def add_head_vehicle(self, vehicle):
        self.curr_vehicle.append(vehicle)
        self._p = self._v = None

    def add_tail_vehicle(self, vehicle):
        self.curr_vehicle.insert(0, vehicle)
        self._p = self._v = None

    def add_vehicle(self, vehicle):
        """Sorted insert by position with the reference's overlap asserts (_micro_lane.py:61-113)."""
        pos = float(vehicle.position)
        assert 0 <= pos <= self.length, ""
        at = 0
        for i, v in enumerate(self.curr_vehicle):
            if float(v.position) > pos:
                break
            at = i + 1
        if at > 0:
            b = self.curr_vehicle[at - 1]
            assert pos - float(b.position) > (vehicle.length + b.length) * 0.5, ""
        if at < self.num_vehicle():
            a = self.curr_vehicle[at]
            assert float(a.position) - pos >= (vehicle.length + a.length) * 0.5, ""
        self.curr_vehicle.insert(at, vehicle)
        self._p = self._v = None

    def get_head_vehicle(self):
        assert self.num_vehicle(), ""
        return self.curr_vehicle[-1]

    def get_tail_vehicle(self):
        assert self.num_vehicle(), ""
        return self.curr_vehicle[0]

    def remove_head_vehicle(self):
        v = self.curr_vehicle.pop()
        if self._p is not None:
            self._p, self._v = self._p[:-1], self._v[:-1]
        return v

    # ---- state ------------------------------------------------------------------------------------------------
    def _sync(self):
        if self._p is None or self._p.shape[0] != self.num_vehicle():
            if self.num_vehicle():
                self._p = device.as_f32([device.as_f32(v.position) for v in self.curr_vehicle])
                self._v = device.as_f32([device.as_f32(v.speed) for v in self.curr_vehicle])
            else:
                self._p = th.zeros(0, device=device.get())
                self._v = th.zeros(0, device=device.get())

    def _refresh_records(self):
        for i, mv in enumerate(self.curr_vehicle):
            mv.position = self._p[i]
            mv.speed = self._v[i]

    def set_state_vector(self, position, speed):
        assert len(position) == self.num_vehicle(), "Vehicle number mismatch"
        assert len(speed) == self.num_vehicle(), "Vehicle number mismatch"
        self._p, self._v = device.as_f32(position), device.as_f32(speed)
        self._refresh_records()

    def get_state_vector(self):
        self._sync()
        return self._p, self._v

    def set_next_state_vector(self, position, speed):
        assert len(position) == self.num_vehicle(), "Vehicle number mismatch"
        assert len(speed) == self.num_vehicle(), "Vehicle number mismatch"
        self._np, self._nv = device.as_f32(position), device.as_f32(speed)
        self.next_vehicle_position = self._np
        self.next_vehicle_speed = self._nv

    def get_next_state_vector(self):
        return self._np, self._nv

    def update_state(self):
        if self.num_vehicle() and self._np is not None:
            self._p, self._v = self._np, self._nv
            self._refresh_records()

    def compute_state_delta(self, id):
        """(gap, speed difference) to the leader of vehicle `id`; the head uses the lane's head deltas."""
        if id == self.num_vehicle() - 1:
            return self.head_position_delta, self.head_speed_delta
        mv, lv = self.curr_vehicle[id], self.curr_vehicle[id + 1]
        return abs(lv.position - mv.position) - ((lv.length + mv.length) * 0.5), mv.speed - lv.speed

    def entering_free_space(self):
        if self.num_vehicle():
            return self.curr_vehicle[0].position - 0.5 * self.curr_vehicle[0].length
        return self.length

    def on_this_lane(self, position, differentiable):
        if not isinstance(position, th.Tensor):
            position = th.tensor(position)
        if differentiable:
            return sigmoid(position, constant=16.0) * sigmoid(self.length - position, constant=16.0)
        return float(position >= 0 and position <= self.length)

    def clear(self):
        self.curr_vehicle.clear()
        self.next_vehicle_position = []
        self.next_vehicle_speed = []
        self._p = self._v = self._np = self._nv = None

    # ---- one step ---------------------------------------------------------------------------------------------
    def vectorize_input(self):
        """(p, v) plus the virtual leader slot (p_head + head_position_delta, v_head - head_speed_delta)."""
        cp, cs = self.get_state_vector()
        if len(cp):
            hp = (cp[-1] + device.as_f32(self.head_position_delta)).reshape(1)
            hs = (cs[-1] - device.as_f32(self.head_speed_delta)).reshape(1)
            cp, cs = th.cat([cp, hp]), th.cat([cs, hs])
        return cp, cs

    def forward(self, delta_time):
        from road.lane.dmicro_lane import dMicroForwardLayer
        if self.num_vehicle() == 0:
            self._np, self._nv = self.get_state_vector()
            return
        cp, cs = self.vectorize_input()
        np_, ns = dMicroForwardLayer.apply(self, cp, cs, delta_time)
        self.set_next_state_vector(np_, ns)
