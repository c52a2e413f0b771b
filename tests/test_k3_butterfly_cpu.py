"""Lane-level model of the K3 epilogue's row statistics (csrc/coarse_match.hip: butterfly16 + swap16 + the LDS broadcast):
pins, on the CPU, which lane ends up with which accumulator row's total, that every one of the 32 lanes of a half wave
contributes exactly once, and that the four 16-byte broadcast reads hand every lane the totals of ITS sixteen rows in register
order.  DPP controls as in the kernel: row_mirror 0x140, row_half_mirror 0x141, quad_perm [2,3,0,1] 0x4E, [1,0,3,2] 0xB1."""
import numpy as np


def dpp_source_lane(ctrl, lane):
    row, i = lane & ~15, lane & 15
    if ctrl == 0x140:                      # row_mirror
        return row | (15 - i)
    if ctrl == 0x141:                      # row_half_mirror
        return row | (i & 8) | (7 - (i & 7))
    if ctrl == 0x4E:                       # quad_perm [2,3,0,1]
        return lane ^ 2
    if ctrl == 0xB1:                       # quad_perm [1,0,3,2]
        return lane ^ 1
    raise ValueError(ctrl)


def butterfly16(x, op):
    """x [64 lanes][16 values] -> [64]: csrc/coarse_match.hip::butterfly16, lane by lane."""
    lanes = np.arange(64)
    cur = [x[:, k] for k in range(16)]
    for ctrl, bit in ((0x140, 8), (0x141, 4), (0x4E, 2), (0xB1, 1)):
        h = len(cur) // 2
        src = np.array([dpp_source_lane(ctrl, l) for l in lanes])
        sel = (lanes & bit) != 0
        nxt = []
        for k in range(h):
            keep = np.where(sel, cur[k + h], cur[k])
            send = np.where(sel, cur[k], cur[k + h])
            nxt.append(op(keep, send[src]))
        cur = nxt
    return cur[0]


def swap16_combine(v, op):
    """v_permlane16_swap of (v, v): both lanes of an xor-16 pair see (row-0 value, row-1 value) of their row pair."""
    lanes = np.arange(64)
    a = v[lanes & ~16]
    b = v[lanes | 16]
    return op(a, b)


def mfma32_row(r, half):
    return (r & 3) + 8 * (r >> 2) + 4 * half


def test_row_totals_land_where_the_kernel_reads_them():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((64, 16)) * 8            # lane-local value of register row r (already reduced over the lane's columns)
    for name, op, red in (("max", np.maximum, np.max), ("sum", np.add, np.sum)):
        tot = swap16_combine(butterfly16(x, op), op)
        for lane in range(64):
            half, r_own = lane >> 5, lane & 15
            members = [l for l in range(64) if l >> 5 == half]
            want = red(x[members, r_own])
            assert np.isclose(tot[lane], want, rtol=1e-12, atol=1e-12), (name, lane)
        # LDS broadcast: lanes with bit 4 clear write row mfma32_row(lane & 15, half); every lane reads four float4 at
        # 8 q + 4 half and takes them as its register rows 4 q .. 4 q + 3
        s_m = np.full(32, np.nan)
        for lane in range(64):
            if lane & 16 == 0:
                s_m[mfma32_row(lane & 15, lane >> 5)] = tot[lane]
        assert not np.isnan(s_m).any()
        for lane in range(64):
            half = lane >> 5
            got = np.concatenate([s_m[8 * q + 4 * half: 8 * q + 4 * half + 4] for q in range(4)])
            members = [l for l in range(64) if l >> 5 == half]
            want = np.array([red(x[members, r]) for r in range(16)])
            assert np.allclose(got, want, rtol=1e-12, atol=1e-12), (name, lane)
            # the tile row of register r of this lane is mfma32_row(r, half): the gate read back belongs to that row
            assert [mfma32_row(r, half) for r in range(16)] == [8 * (r >> 2) + 4 * half + (r & 3) for r in range(16)]


def test_every_lane_contributes_exactly_once():
    # indicator inputs: value r of lane l = 1 only for one (l, r): its total must be 1 in lanes r, r + 16 of l's half, 0 elsewhere
    for l0 in (0, 7, 19, 33, 63):
        for r0 in (0, 5, 15):
            x = np.zeros((64, 16))
            x[l0, r0] = 1.0
            tot = swap16_combine(butterfly16(x, np.add), np.add)
            for lane in range(64):
                expect = 1.0 if (lane >> 5 == l0 >> 5 and lane & 15 == r0) else 0.0
                assert tot[lane] == expect, (l0, r0, lane)
