"""RoadNetwork on the reference's import path (road.network.road_network; reference road_network.py:17-646).

Step order kept from the reference (road_network.py:79-111): every lane's boundary from the time-n state, every
lane's step, every lane's commit, then the hand-offs in lane-id order.  Each lane step is one call of the HIP
operator; the batched / time-fused API for many independent lanes is dhts.macro_rollout / dhts.micro_rollout.
"""
import numpy as np

from road.lane._macro_lane import MacroLane
from road.lane._micro_lane import DEFAULT_HEAD_POSITION_DELTA, DEFAULT_HEAD_SPEED_DELTA, MicroLane
from road.network.conversion import Conversion
from road.network.route import MacroRoute, MicroRoute
from road.vehicle.micro_vehicle import DEFAULT_VEHICLE_LENGTH, MicroVehicle

MAX_ROUTE_LENGTH = 32


class RoadNetwork:

    def __init__(self, speed_limit):
        self.lane = {}
        self.speed_limit = speed_limit           # uniform over the network
        self.vehicle_length = DEFAULT_VEHICLE_LENGTH
        self.macro_route = MacroRoute()
        self.vehicle = {}
        self.micro_route = {}
        self.num_lane = 0
        self.num_vehicle = 0

    # ---- construction ------------------------------------------------------------------------------------------
    def add_lane(self, lane):
        assert isinstance(lane, (MacroLane, MicroLane)), ""
        lane.speed_limit = self.speed_limit
        lane.id = self.num_lane
        self.num_lane += 1
        self.lane[lane.id] = lane
        return lane.id

    def add_vehicle(self, nv, route):
        assert nv.length == self.vehicle_length, ""
        nv.id = self.num_vehicle
        self.num_vehicle += 1
        self.vehicle[nv.id] = nv
        self.micro_route[nv.id] = route
        lane = self.lane[route.curr_lane_id()]
        assert lane.is_micro()
        lane.add_vehicle(nv)
        return nv.id

    def connect_lane(self, prev_lane_id, next_lane_id):
        a, b = self.lane[prev_lane_id], self.lane[next_lane_id]
        a.add_next_lane(b)
        b.add_prev_lane(a)

    # ---- one step ----------------------------------------------------------------------------------------------
    def forward(self, delta_time, differentiable):
        for lane in self.lane.values():
            self.setup_boundary(lane.id, differentiable)
        for lane in self.lane.values():
            # the reference's PLAIN MicroLane (itscp `micro` mode) holds torch tensors in a differentiable episode and steps in their
            # float32 arithmetic; its dMicroLane (hybrid mode) and every evaluation episode compute with Python floats
            lane._tensor_ladder = bool(differentiable)
            lane.forward(delta_time)
        for lane in self.lane.values():
            lane.update_state()
        self.conversion(delta_time)

    def conversion(self, delta_time):
        for lane in self.lane.values():
            if lane.is_macro():
                self.conversion_macro(lane, delta_time)
            else:
                self.conversion_micro(lane)

    def conversion_macro(self, lane, delta_time):
        nid = self.macro_route.get_next_lane(lane.id)
        if nid == -1:
            return
        nxt = self.lane[nid]
        if nxt.is_macro():
            Conversion.macro_to_macro(self, lane, nxt)
        else:
            Conversion.macro_to_micro(self, lane, nxt, delta_time)

    def conversion_micro(self, lane):
        if not lane.num_vehicle():
            return
        nid = self.micro_route[lane.get_head_vehicle().id].next_lane_id()
        if nid == -1:
            Conversion.micro_to_none(self, lane)
        elif self.lane[nid].is_macro():
            Conversion.micro_to_macro(self, lane)
        else:
            Conversion.micro_to_micro(self, lane)

    def setup_boundary(self, id, differentiable):
        if self.lane[id].is_macro():
            self.setup_macro_boundary(id, differentiable)
        else:
            self.setup_micro_boundary(id, differentiable)

    def get_macro_state_of_micro_lane(self, id, differentiable):
        """(density, speed) a micro lane would have as a macro cell (reference road_network.py:207-297): every vehicle on the
        lane, every vehicle on an upstream micro lane that is routed onto it (at minus its distance to that lane's end) and
        every vehicle on a downstream micro lane that came through it (at the lane's length plus its position) counts with
        the weight lane.on_this_lane(position) -- a product of two sigmoids when differentiable, a 0 / 1 test otherwise;
        density = sum of weight x length / lane length, capped at 1; speed = weighted mean, or the speed limit when the
        weights sum to nothing."""
        lane = self.lane[id]
        assert lane.is_micro(), ""
        seen = [(v, v.position) for v in lane.curr_vehicle]
        for up in lane.prev_lane.values():
            if not up.is_macro():
                seen += [(v, -(up.length - v.position)) for v in up.curr_vehicle if self.micro_route[v.id].next_lane_id() == id]
        for down in lane.next_lane.values():
            if not down.is_macro():
                seen += [(v, lane.length + v.position) for v in down.curr_vehicle if self.micro_route[v.id].prev_lane_id() == id]
        density, speed_sum, weight_sum = 0, 0, 0
        for v, position in seen:
            w = lane.on_this_lane(position, differentiable)
            density = density + w * (v.length / lane.length)
            speed_sum = speed_sum + w * v.speed
            weight_sum = weight_sum + w
        density = min(density, 1.0)
        return density, (speed_sum / weight_sum if weight_sum > 0 else self.speed_limit)

    # ---- macro boundary ------------------------------------------------------------------------------------------
    def get_macro_boundary(self, id, left, differentiable):
        """Ghost (r, u) at one end: the connected macro lane's edge cell (the single neighbour, or this step's
        macro_route choice among several), else the lane's own stored ghost."""
        lane = self.lane[id]
        adj = lane.prev_lane if left else lane.next_lane
        own = lane.get_leftmost_cell() if left else lane.get_rightmost_cell()
        if len(adj) == 0:
            return own.state.q.r, own.state.u
        if len(adj) == 1:
            other = next(iter(adj.values()))
        else:
            other = self.lane[self.macro_route.get_prev_lane(id) if left else self.macro_route.get_next_lane(id)]
        if other.is_macro():
            r, _, u = other.get_state_vector()
            return (r[-1], u[-1]) if left else (r[0], u[0])
        return own.state.q.r, own.state.u

    def setup_macro_boundary(self, id, differentiable):
        lane = self.lane[id]
        assert lane.is_macro(), ""
        lane.set_leftmost_cell(*self.get_macro_boundary(id, True, differentiable))
        lane.set_rightmost_cell(*self.get_macro_boundary(id, False, differentiable))

    def create_random_macro_route(self):
        """Random one-to-one matching of macro lanes to still-unclaimed successors (reference :389-423)."""
        route = MacroRoute()
        for lane_id in np.random.permutation(list(self.lane.keys())):
            lane = self.lane[lane_id]
            if lane.is_micro():
                continue
            for nid in np.random.permutation(list(lane.next_lane.keys())):
                if nid not in route.prev_lane_dict:
                    route.next_lane_dict[lane_id] = nid
                    route.prev_lane_dict[nid] = lane_id
                    break
        return route

    # ---- micro boundary ------------------------------------------------------------------------------------------
    def setup_micro_boundary(self, id, differentiable):
        """Gap of the head vehicle to its leader further along its route (reference :429-580): walk the route; an
        on-route micro successor holding vehicles gives its tail vehicle as the leader; an on-route macro successor
        ends the search with the defaults; gap = remaining lane lengths + leader offset - len/2, clamped at 0."""
        lane = self.lane[id]
        assert lane.is_micro(), ""
        lane.head_position_delta = DEFAULT_HEAD_POSITION_DELTA
        lane.head_speed_delta = DEFAULT_HEAD_SPEED_DELTA
        if lane.num_vehicle() == 0:
            return
        hv = lane.get_head_vehicle()
        route = self.micro_route[hv.id]
        reach = lane.length - hv.position - hv.length * 0.5
        found = None
        k = route.curr_idx
        while k < route.route_length() - 1:
            here, there = self.lane[route.route[k]], self.lane[route.route[k + 1]]
            stop = False
            for cand in here.next_lane.values():
                if cand.id != there.id:
                    continue
                if isinstance(cand, MacroLane):
                    stop = True
                elif isinstance(cand, MicroLane):
                    if cand.num_vehicle():
                        lv = cand.get_tail_vehicle()
                        gap = reach + (lv.position - lv.length * 0.5)
                        gap = gap if float(gap) > 0.0 else 0.0
                        found = (gap, hv.speed - lv.speed)
                        stop = True
                else:
                    raise ValueError()
            if stop:
                break
            reach = reach + there.length
            k += 1
        if found is not None:
            # a single on-route leader has score 1.0, so the weighted average is the pair itself
            lane.head_position_delta = found[0]
            lane.head_speed_delta = found[1]

    # ---- routes / vehicles ---------------------------------------------------------------------------------------
    def create_random_route(self, lane_id):
        """Random walk along next lanes, at most MAX_ROUTE_LENGTH lanes (reference :604-646).  Per visited lane that
        has successors exactly one np.random.randint is drawn; a successor already on the route is skipped by
        rotating through the others, falling back to the first draw when all of them were visited."""
        visited = []
        cur_id = lane_id
        for _ in range(MAX_ROUTE_LENGTH):
            visited.append(cur_id)
            cur = self.lane[cur_id]
            if not cur.has_next_lane():
                break
            keys = list(cur.next_lane.keys())
            first = np.random.randint(0, len(keys))
            pick = first
            while keys[pick] in visited:
                pick = (pick + 1) % len(keys)
                if pick == first:
                    break
            cur_id = keys[pick]
        return MicroRoute(visited)

    def create_default_vehicle_with_random_route(self, lane_id):
        return MicroVehicle.default_micro_vehicle(self.speed_limit), self.create_random_route(lane_id)

    def create_random_vehicle_with_random_route(self, lane_id):
        return MicroVehicle.random_micro_vehicle(self.speed_limit), self.create_random_route(lane_id)
