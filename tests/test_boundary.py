"""The drop-in boundary: libdhts.so loads and exports every symbol include/dhts.h declares; host-side argument
checks; the product never touches the oracle.  No GPU needed (no compute calls)."""
import ctypes as C
import os
import re

import pytest

from conftest import PKG, ROOT


def header_functions():
    txt = open(os.path.join(ROOT, "include", "dhts.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dhts_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from dhts import _lib
    lib = _lib.lib()
    names = header_functions()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), "libdhts.so does not export %s" % n
    # and the binding table covers the header exactly
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_padding():
    from dhts import _lib
    lib = _lib.lib()
    assert lib.dhts_version() == 100
    assert [lib.dhts_padded(n) for n in (1, 64, 65, 512)] == [64, 64, 128, 512]


def test_options_accept_documented_values_only():
    from dhts import _lib
    lib = _lib.lib()
    for v in (0, 1, 5, 8):
        assert lib.dhts_set_option(_lib.OPT_MACRO_FWD_WAVES, v) == 0
    assert lib.dhts_set_option(_lib.OPT_MACRO_FWD_WAVES, 17) == _lib.E_INVALID
    for v in (1, 2, 4, 0):
        assert lib.dhts_set_option(_lib.OPT_MICRO_FWD_WAVES, v) == 0
    assert lib.dhts_set_option(_lib.OPT_MICRO_FWD_WAVES, 3) == _lib.E_INVALID
    for v in (1, 2, 4, 0):
        assert lib.dhts_set_option(_lib.OPT_MACRO_FWD_GROUP, v) == 0
    assert lib.dhts_set_option(_lib.OPT_MACRO_FWD_GROUP, 3) == _lib.E_INVALID
    for v in (1, 2, 0):
        assert lib.dhts_set_option(_lib.OPT_MACRO_FWD_VARIANT, v) == 0
    assert lib.dhts_set_option(_lib.OPT_MACRO_FWD_VARIANT, 3) == _lib.E_INVALID
    for v in (1, 64, 158, 0):
        assert lib.dhts_set_option(_lib.OPT_NETSTEP_LDS_KB, v) == 0
    assert lib.dhts_set_option(_lib.OPT_NETSTEP_LDS_KB, 159) == _lib.E_INVALID
    for v in (256, 512, 1024, 0):
        assert lib.dhts_set_option(_lib.OPT_NETSTEP_BLOCK, v) == 0
    assert lib.dhts_set_option(_lib.OPT_NETSTEP_BLOCK, 128) == _lib.E_INVALID
    assert lib.dhts_set_option(_lib.OPT_MACRO_FWD_ROTATE, 2) == _lib.E_INVALID and lib.dhts_set_option(_lib.OPT_MACRO_FWD_ROTATE, 1) == 0
    for v in (0, 1, 2):
        assert lib.dhts_set_option(_lib.OPT_HYB_PACK, v) == 0
    assert lib.dhts_set_option(_lib.OPT_HYB_PACK, 3) == _lib.E_INVALID
    assert lib.dhts_set_option(_lib.OPT_REWARD_CHAIN, 2) == _lib.E_INVALID and lib.dhts_set_option(_lib.OPT_REWARD_CHAIN, 0) == 0
    assert lib.dhts_set_option(99, 0) == _lib.E_INVALID
    assert lib.dhts_set_option(_lib.OPT_MACRO_FWD_WAVES, 0) == 0       # back to the heuristics


def documented_macro_tape_row():
    """The ROW LAYOUT lines of dhts_macro_tape_bytes' comment in include/dhts.h, parsed: a list of (block, [(ctype, count
    expression in n_cells), ...]).  The geometry tests below, bench.py's tape census and tests/test_gpu_parity.py's decoder
    all stand on these three lines, so a header that drifts from the kernels fails here."""
    txt = open(os.path.join(ROOT, "include", "dhts.h")).read()
    doc = txt[txt.index("ROW LAYOUT"):txt.index("size_t dhts_macro_tape_bytes")]
    blocks = []
    for name in "SHE":
        m = re.search(r"^ \*\s+%s: (.*?)(?:\s{3,}|$)" % name, doc, flags=re.M)
        assert m, "include/dhts.h: no line for block %s of the macro rollout tape" % name
        fields = []
        for f in m.group(1).split(","):
            fm = re.match(r"\s*(float32|uint32|uint16)\s+(\w+)?\s*((?:\[[^\]]+\])*)", f)
            assert fm, "unparsed field %r" % f
            dims = re.findall(r"\[([^\]]+)\]", fm.group(3))
            fields.append((fm.group(1), dims))
        blocks.append((name, fields))
    return blocks


def documented_macro_tape_row_bytes(N):
    size = {"float32": 4, "uint32": 4, "uint16": 2}
    line = lambda nbytes: (nbytes + 127) // 128 * 128      # noqa: E731
    total, per_block = 0, {}
    for name, fields in documented_macro_tape_row():
        nbytes = 0
        for ctype, dims in fields:
            n = size[ctype]
            for d in dims:
                n *= int(eval(d, {"__builtins__": {}}, {"n_cells": N}))
            nbytes += n
        per_block[name] = line(nbytes)
        total += line(nbytes)
    return total, per_block


def test_header_documents_the_tape_the_kernels_write():
    """The header's row description, field by field (it described a layout no kernel implemented)."""
    blocks = dict(documented_macro_tape_row())
    assert blocks["S"] == [("float32", ["n_cells", "3"])]
    assert blocks["H"] == [("uint32", []), ("uint32", []), ("uint16", ["n_cells + 1"])]
    assert blocks["E"] == [("float32", ["n_cells + 1", "2", "4"])]
    txt = open(os.path.join(ROOT, "include", "dhts.h")).read()
    assert "mask[" not in txt and "[n_cells][2] " not in txt


@pytest.mark.parametrize("N", [1, 2, 63, 64, 65, 100, 512, 513, 1000, 4000])
def test_macro_tape_rows_follow_the_documented_geometry(N):
    """dhts_macro_tape_bytes equals the row include/dhts.h documents (parsed from the header's comment, not restated here),
    every block rounded up to whole 128-byte lines."""
    from dhts import _lib
    lib = _lib.lib()
    d = _lib.MacroDesc(3, N, 0.01, 5.0, 30.0)
    row, _ = documented_macro_tape_row_bytes(N)
    assert lib.dhts_macro_tape_bytes(C.byref(d), 7) == 7 * 3 * row


def test_tape_bytes_match_documented_layout():
    from dhts import _lib
    lib = _lib.lib()
    d = _lib.MacroDesc(1024, 512, 0.01, 5.0, 30.0)
    assert lib.dhts_macro_tape_bytes(C.byref(d), 1000) == 1000 * 1024 * (384 + 72 + 1032) * 16  # S 512 x 12 B | header 1034 B -> 9 lines | E 513 x 32 B
    assert lib.dhts_macro_step_tape_bytes(C.byref(d)) == 1024 * 3 * 512 * 16           # 48 B per cell (the reference's dqs)
    m = _lib.MicroDesc(4096, 256, 0.01)
    assert lib.dhts_micro_tape_bytes(C.byref(m), 1000) == 1000 * 4096 * 256 * 12        # 12 B per vehicle-step
    assert lib.dhts_micro_step_tape_bytes(C.byref(m)) == 4096 * 2 * 256 * 16            # 32 B per vehicle (the reference's dqs)
    bad = _lib.MacroDesc(0, 512, 0.01, 5.0, 30.0)
    assert lib.dhts_macro_tape_bytes(C.byref(bad), 10) == 0


def test_invalid_arguments_are_rejected_without_a_gpu():
    from dhts import _lib
    lib = _lib.lib()
    d = _lib.MacroDesc(4, 0, 0.01, 5.0, 30.0)           # zero cells
    assert lib.dhts_macro_rollout_fwd(C.byref(d), 1, *([None] * 13)) == _lib.E_INVALID
    d = _lib.MacroDesc(4, 16, 0.01, 5.0, 30.0)          # NULL state pointers
    assert lib.dhts_macro_rollout_fwd(C.byref(d), 1, *([None] * 13)) == _lib.E_INVALID
    assert lib.dhts_macro_rollout_bwd(C.byref(d), 1, *([None] * 9)) == _lib.E_INVALID
    m = _lib.MicroDesc(4, 5000, 0.01)                   # over capacity
    assert lib.dhts_micro_rollout_fwd(C.byref(m), 1, *([None] * 11)) == _lib.E_INVALID
    assert lib.dhts_macro_state_from_ru(8, 30.0, None, None, None, None, None) == _lib.E_INVALID


def test_network_entry_points_reject_bad_sizes():
    from dhts import _lib
    lib = _lib.lib()
    ok = _lib.NetDesc(4, 40, 236, 300, 1, 60, 5, 1 / 30, 60.0, 0.2, 5.0)
    assert lib.dhts_net_macro_hist_bytes(C.byref(ok)) == 4 * 301 * 4 * 236 * 4
    assert lib.dhts_net_macro_tape_bytes(C.byref(ok)) == 4 * 300 * 3 * 256 * 16
    long_run = _lib.NetDesc(4, 40, 236, 600, 1, 60, 5, 1 / 30, 60.0, 0.2, 5.0)       # 600 * 236 > 100000 samples: the window slides
    assert lib.dhts_net_macro_hist_bytes(C.byref(long_run)) == 4 * 601 * 4 * 236 * 4
    too_wide = _lib.NetDesc(4, 300, 900, 10, 1, 60, 5, 1 / 30, 60.0, 0.2, 5.0)        # cells + lanes > 1024
    assert lib.dhts_net_macro_tape_bytes(C.byref(too_wide)) == 0
    tabs = _lib.NetTables()
    assert lib.dhts_net_macro_rollout_fwd(C.byref(ok), C.byref(tabs), *([None] * 9)) == _lib.E_INVALID
    assert lib.dhts_net_macro_rollout_bwd(C.byref(ok), C.byref(tabs), *([None] * 10)) == _lib.E_INVALID


def test_network_tables_builder():
    """dhts.network.MacroNetworkTables: ghost-source tables of a 3-lane chain with a fork."""
    import numpy as np
    from dhts.network import MacroNetworkTables
    # lane 0 -> {1, 2}; per-step route picks 1 (t = 0) then 2 (t = 1); lane 3 is isolated
    mr = np.array([[1, -1, -1, -1], [2, -1, -1, -1]])
    t = MacroNetworkTables([2, 3, 1, 4], [10.0, 15.0, 5.0, 20.0], [(0, 1), (0, 2)], [1, 0, 0, 2], [0, 0, 0, 0], mr,
                           np.ones((4, 2)))
    assert t.n_cells == 10 and t.lane_off.tolist() == [0, 2, 5, 6] and np.allclose(t.lane_dx, 5.0)
    assert t.left_src.tolist() == [[-1, 0, 0, -1], [-1, 0, 0, -1]]          # single upstream lane: always that lane
    assert t.left_gate.tolist() == [[-2, 0, -1, -2], [-2, -1, 0, -2]]        # gated by the route's choice (-1 = red)
    assert t.right_src.tolist() == [[1, -1, -1, -1], [2, -1, -1, -1]]        # fork follows the route; sinks keep their own ghost
    assert t.is_source.tolist() == [True, False, False, True]


def test_hybrid_tables_builder_and_route_grouping():
    """dhts.network.HybridNetworkTables / group_routes: macro lane 0 -> micro lane 1 -> macro lane 2 (the G7 shape) plus a
    macro fork 3 -> {4, 5}."""
    import numpy as np
    from dhts.network import HybridNetworkTables, group_routes
    T = 2
    mr = -np.ones((T, 6), dtype=np.int64)
    mr[:, 0] = 1                                  # the macro route may point at a micro lane (conversion target)
    mr[0, 3], mr[1, 3] = 4, 5
    t = HybridNetworkTables([1, 0, 1, 1, 1, 1], [4, 9, 2, 3, 3, 3], [20.0, 15.0, 10.0, 15.0, 15.0, 15.0],
                            [(0, 1), (1, 2), (3, 4), (3, 5)], [1, 0, 0, 2, 0, 0], [0, 0, 0, 0, 0, 0], mr, np.ones((6, T)))
    assert t.lane_ncell.tolist() == [4, 0, 2, 3, 3, 3] and t.n_cells == 15 and t.lane_off.tolist() == [0, 4, 4, 6, 9, 12]
    assert t.lane_dx.tolist() == [5.0, 0.0, 5.0, 5.0, 5.0, 5.0]
    assert t.left_src[0].tolist() == [-1, -1, -3, -1, 3, 3]        # lane 2's only upstream lane is micro: own ghost
    assert t.left_gate[0].tolist() == [-2, -1, -1, -2, 3, -1] and t.left_gate[1].tolist() == [-2, -1, -1, -2, -1, 3]
    assert t.right_src[:, 0].tolist() == [-1, -1] and t.right_src[:, 3].tolist() == [4, 5]
    assert t.conv_next[:, 0].tolist() == [1, 1] and (t.conv_next[:, 1] == -1).all()
    assert t.nxt_ptr.tolist() == [0, 1, 2, 2, 4, 4, 4] and t.nxt_idx.tolist() == [1, 2, 4, 5]
    routes, ptr = group_routes([[1, 2, -1], [4, -1, -1], [1, -1, -1]], 6)
    assert routes.tolist() == [[1, 2, -1], [1, -1, -1], [4, -1, -1]]      # stable per first lane
    assert ptr.tolist() == [0, 0, 2, 2, 2, 3, 3]
    t.check_kernel_limits()
    mid = HybridNetworkTables([1] + [0] * 30, [4] + [0] * 30, [20.0] * 31, [(k, k + 1) for k in range(0, 30)],
                              [0] * 31, [0] * 31, -np.ones((T, 31), dtype=np.int64), np.ones((31, T)))
    mid.check_kernel_limits()                     # 30 micro lanes: inside the kernels' 64 since round 3
    big = HybridNetworkTables([1] + [0] * 70, [4] + [0] * 70, [20.0] * 71, [(k, k + 1) for k in range(0, 70)],
                              [0] * 71, [0] * 71, -np.ones((T, 71), dtype=np.int64), np.ones((71, T)))
    with pytest.raises(ValueError, match="64 micro lanes"):
        big.check_kernel_limits()
    # a micro SOURCE lane (no upstream lane: stochastic admission, _simulator.py:153-174) needs its draws
    src = HybridNetworkTables([0, 1], [0, 3], [10.0, 15.0], [(0, 1)], [0, 0], [0, 0], -np.ones((T, 2), dtype=np.int64), np.ones((2, T)))
    assert src.lane_source.tolist() == [1, 0]
    with pytest.raises(ValueError, match="admission draws"):
        src.check_kernel_limits()
    src.set_micro_sources([0.3, 0.7]).check_kernel_limits()


def test_hybrid_entry_points_reject_bad_arguments():
    from dhts import _lib
    lib = _lib.lib()
    ok = _lib.NetDesc(4, 144, 256, 600, 9, 120, 45, 1 / 30, 60.0, 0.2, 5.0)
    tabs = _lib.HybridTables()
    assert lib.dhts_net_hybrid_workspace_bytes(C.byref(ok), C.byref(tabs)) == 0          # no routes
    tabs.n_routes, tabs.route_stride = 8, 32
    assert lib.dhts_net_hybrid_workspace_bytes(C.byref(ok), C.byref(tabs)) > 4 * 600 * 512 * 36
    assert lib.dhts_net_hybrid_rollout_fwd(C.byref(ok), C.byref(tabs), *([None] * 10)) == _lib.E_INVALID
    assert lib.dhts_net_hybrid_rollout_bwd(C.byref(ok), C.byref(tabs), *([None] * 10)) == _lib.E_INVALID
    too_wide = _lib.NetDesc(4, 300, 900, 10, 1, 60, 5, 1 / 30, 60.0, 0.2, 5.0)
    assert lib.dhts_net_hybrid_workspace_bytes(C.byref(too_wide), C.byref(tabs)) == 0


def test_missing_library_fails_loudly(monkeypatch):
    from dhts import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "SO_PATH", os.path.join(PKG, "csrc", "does_not_exist.so"))
    with pytest.raises(_lib.DhtsError, match="no CPU fallback"):
        _lib.lib()


def test_product_does_not_touch_oracle():
    """Nothing under the product package imports, links or mentions the oracle."""
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.lower(), os.path.join(dirpath, f)
    txt = open(os.path.join(ROOT, "include", "dhts.h")).read()
    assert "oracle" not in txt.lower()


def test_cpu_tensors_are_rejected():
    import torch
    from dhts import ops
    with pytest.raises(TypeError, match="CUDA"):
        ops.macro_state_from_ru(torch.zeros(4), torch.zeros(4), 30.0)
