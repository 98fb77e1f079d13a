"""ItscpRoadNetwork on the reference's import path (example.control.itscp._simulator; reference _simulator.py:19-281):
RoadNetwork whose lane boundaries are blended between a green and a red value by differentiable traffic signals.

  macro lane, left ghost : prev_signal * green + (1 - prev_signal) * (r = 0, u = u_max); green = the connected upstream cell
                           (or, for a source lane, r = schedule, u = u_eq(r)); prev_signal = the upstream lane's signal
  macro lane, right ghost: s * green + (1 - s) * (r = 1, u = 0), s = sigmoid(32 (signal - 0.5))
  micro lane, head gap   : blend of the route leader's gap (green) and the distance to the stop line (red), weighted by
                           the signals of the previous / current / next lane of the head vehicle's route
Host glue over the lane mirrors; the lane steps themselves run in the HIP kernels.
"""
import numpy as np
import torch as th

from dmath.operation import sigmoid
from example.common.rms import RunningMean
from model.macro._arz import ARZ
from road.network.road_network import RoadNetwork
from road.vehicle.vehicle import DEFAULT_VEHICLE_LENGTH

MICRO_STOP_DISTANCE = 3.0
MICRO_STOP_GRADIENT = 1e-0


class ItscpRoadNetwork(RoadNetwork):

    def __init__(self, speed_limit):
        super().__init__(speed_limit)
        self.lane_signal = {}
        self.lane_incoming = {}
        self.lane_waiting_micro_vehicle = {}
        self.lane_waiting_micro_route = {}
        self.signal_rms = RunningMean(100_000)
        self.lane_incoming_acc = {}
        # the draw that admits a waiting vehicle onto a micro source lane (reference: np.random.random at call time,
        # _simulator.py:159); a test replays the reference's recorded draws through this hook
        self.random_draw = lambda: np.random.random((1,)).item()

    def interpolate_signal(self, signal, green_value, red_value):
        assert signal >= 0.0 and signal <= 1.0, ""
        return green_value * signal + red_value * (1.0 - signal)

    def setup_macro_boundary(self, id, differentiable):
        lane = self.lane[id]
        assert lane.is_macro(), ""
        # upstream end
        if not lane.has_prev_lane():
            green_r = self.lane_incoming[id]
            green_u = ARZ.compute_u_eq(green_r, self.speed_limit)
            prev_signal = 1.0                      # a source lane always accepts its inflow
        else:
            green_r, green_u = self.get_macro_boundary(id, True, differentiable)
            prev_id = self.macro_route.get_prev_lane(lane.id)
            prev_signal = 0.0 if prev_id == -1 else self.lane_signal[prev_id]
        lane.set_leftmost_cell(self.interpolate_signal(prev_signal, green_r, 0),
                               self.interpolate_signal(prev_signal, green_u, self.speed_limit))
        # downstream end
        green_r, green_u = self.get_macro_boundary(id, False, differentiable)
        signal = self.lane_signal[id]
        signal = sigmoid(signal - 0.5, constant=32) if differentiable else float(signal > 0.5)
        lane.set_rightmost_cell(signal * green_r + (1.0 - signal) * 1.0, signal * green_u + (1.0 - signal) * 0.0)

    def setup_micro_boundary(self, id, differentiable):
        k_sig = 16.0
        lane = self.lane[id]
        assert lane.is_micro(), ""
        if not lane.has_prev_lane():                # stochastic inflow of a source lane
            if lane.entering_free_space() > DEFAULT_VEHICLE_LENGTH * 0.5:
                rand = self.random_draw()
                wv, wr = self.lane_waiting_micro_vehicle[id], self.lane_waiting_micro_route[id]
                if rand < self.lane_incoming[id] and len(wv) and len(wr):
                    self.add_vehicle(wv[-1], wr[-1])
                    self.lane_waiting_micro_vehicle[id] = wv[:-1]
                    self.lane_waiting_micro_route[id] = wr[:-1]
        super().setup_micro_boundary(id, differentiable)
        green_dp, green_dv = lane.head_position_delta, lane.head_speed_delta
        if lane.num_vehicle() == 0:
            return
        hv = lane.get_head_vehicle()
        hr = self.micro_route[hv.id]
        red_dp = lane.length - hv.position - (hv.length * 0.5)
        red_dp = red_dp if float(red_dp) > 0.0 else 0.0
        red_dv = 0.0
        prev_exist, next_exist = hr.prev_lane_id() != -1, hr.next_lane_id() != -1
        prev_score = sigmoid(-hv.position, constant=k_sig) if (differentiable and prev_exist) else 0
        curr_score = (sigmoid(hv.position, constant=k_sig) * sigmoid(lane.length - hv.position, constant=k_sig)) if differentiable else 1
        next_score = sigmoid(hv.position - lane.length, constant=k_sig) if (differentiable and next_exist) else 0
        total = prev_score + curr_score + next_score
        final = 0
        if prev_exist:
            final = final + (prev_score / total) * self.lane_signal[hr.prev_lane_id()]
        final = final + (curr_score / total) * self.lane_signal[hr.curr_lane_id()]
        if next_exist:
            final = final + (next_score / total) * self.lane_signal[hr.next_lane_id()]
        if differentiable:
            with th.no_grad():
                self.signal_rms.update(final.detach().cpu().numpy() if isinstance(final, th.Tensor) else final)
                constant = 32. / np.abs(self.signal_rms.mean())
            final = sigmoid(final - 0.5, constant=constant)
            lane.head_position_delta = green_dp * final + red_dp * (1.0 - final)
            lane.head_speed_delta = green_dv * final + red_dv * (1.0 - final)
        elif ItscpRoadNetwork.is_signal_green(final):
            lane.head_position_delta, lane.head_speed_delta = green_dp, green_dv
        else:
            lane.head_position_delta, lane.head_speed_delta = red_dp, red_dv

    @staticmethod
    def is_signal_green(signal):
        return signal >= 0.5
