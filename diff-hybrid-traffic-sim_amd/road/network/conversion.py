"""Hand-offs between lanes after every step (reference road/network/conversion.py:8-215).

Host glue over the lane mirrors (torch ops on the lanes' device arrays); the fused network kernel of a later
round moves this onto the device.  Semantics kept:
  macro -> micro : the last cell's flux r*u*dt charges a per-successor capacitor; once it holds one vehicle length
                   and the successor has that much free space at its entrance, a default vehicle spawns at
                   position 0 with the cell's speed; its ancillary variable `a` carries the capacitor's gradient
                   (value = one vehicle length), and the capacitor keeps the detached remainder.
  micro -> macro : when the head vehicle's position passes the lane end by more than its length it is removed and
                   deposited as density a/len * overlap/dx into the overlapped leading cells of the successor
                   (straight-through clamp to [1e-5, 1 - 1e-5]), which take the vehicle's speed.
  micro -> micro : at position >= lane length the head vehicle becomes the successor's tail vehicle.
  micro -> none  : at position >= lane length the head vehicle leaves the network.
"""
import torch as th

from dhts import device
from model.macro._arz import ARZ
from road.vehicle.micro_vehicle import MicroVehicle


class Conversion:

    @staticmethod
    def macro_to_macro(network, prev_lane, next_lane):
        return None

    @staticmethod
    def macro_to_micro(network, prev_lane, next_lane, delta_time):
        r, _, u = prev_lane.get_state_vector()
        prev_lane.add_flux_capacitor(next_lane.id, r[-1] * u[-1] * delta_time)
        stored = prev_lane.flux_capacitor[next_lane.id]
        level = float(stored.detach()) if isinstance(stored, th.Tensor) else float(stored)
        nv = MicroVehicle.default_micro_vehicle(next_lane.speed_limit)
        if level >= nv.length and float(next_lane.entering_free_space()) >= nv.length * 1.0:
            nv.position = 0
            nv.speed = u[-1]
            nv.a = stored - (level - nv.length)           # value = one vehicle length, gradient = d(capacitor)
            if isinstance(stored, th.Tensor):
                prev_lane.flux_capacitor[next_lane.id] = th.tensor(level - nv.length, dtype=stored.dtype, device=stored.device)
            else:
                prev_lane.flux_capacitor[next_lane.id] = level - nv.length
            network.add_vehicle(nv, network.create_random_route(next_lane.id))

    @staticmethod
    def micro_to_macro(network, prev_lane):
        if not prev_lane.num_vehicle():
            return
        hv = prev_lane.get_head_vehicle()
        next_lane = network.lane[network.micro_route[hv.id].next_lane_id()]
        assert next_lane.is_macro(), ""
        if not float(hv.position) > prev_lane.length + 1.0 * hv.length:
            return
        prev_lane.remove_head_vehicle()
        ln = float(hv.length)
        front = device.as_f32(hv.position) - prev_lane.length          # vehicle front in the successor's coordinate
        rear = front - ln
        dx = next_lane.cell_length
        r, y, u = next_lane.get_state_vector()
        _, _, _ = r, y, u
        q = next_lane._curr.t["q"]
        new_r, new_y, new_u = r, y, u
        for ci in range(next_lane.num_cell):
            c_lo, c_hi = dx * ci, dx * (ci + 1)
            if not (c_hi > float(rear) and c_lo < float(front)):
                break
            hi = front if float(front) > c_hi else th.as_tensor(c_hi, dtype=th.float32, device=front.device)
            lo = rear if float(rear) < c_lo else th.as_tensor(c_lo, dtype=th.float32, device=front.device)
            overlap = dx + ln - (hi - lo)
            n_r = r[ci] + (device.as_f32(hv.a) / ln) * (overlap / dx)
            val = float(n_r)
            if val > 1.0 - 1e-5:                            # straight-through clamp
                n_r = n_r - (val - (1.0 - 1e-5))
            elif val < 1e-5:
                n_r = n_r - (val - 1e-5)
            speed = device.as_f32(hv.speed)
            idx = (th.tensor(ci, device=r.device),)
            new_r = new_r.index_put(idx, n_r.reshape(()))
            new_u = new_u.index_put(idx, speed.reshape(()))
            new_y = new_y.index_put(idx, ARZ.compute_y(n_r, speed, next_lane.speed_limit).reshape(()))
        next_lane._curr.set_all(new_r, new_y, new_u, q)

    @staticmethod
    def micro_to_micro(network, prev_lane):
        if not prev_lane.num_vehicle():
            return
        hv = prev_lane.get_head_vehicle()
        route = network.micro_route[hv.id]
        next_lane = network.lane[route.next_lane_id()]
        assert next_lane.is_micro(), ""
        if float(hv.position) >= prev_lane.length:
            prev_lane.remove_head_vehicle()
            hv.position = hv.position - prev_lane.length
            next_lane.add_tail_vehicle(hv)
            route.increment_curr_idx()

    @staticmethod
    def micro_to_none(network, prev_lane):
        if prev_lane.num_vehicle() and float(prev_lane.get_head_vehicle().position) >= prev_lane.length:
            prev_lane.remove_head_vehicle()
