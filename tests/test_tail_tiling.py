"""CPU checks of the tiling behind the D x H zero-skipping folded decoder tail (vqvdb_amd/csrc/vq_tail_rows.h, vq_tail_groups.h).

The kernels never run the MFMAs of a (tile, input row) pair outside the tile's reach box.  That is only bit-identical to the oracle's
tail_apply (which multiplies every position of the planes a voxel depends on) if the composite weights outside the box are EXACTLY zero.
Here, without a GPU: the composite operator up_conv -> PixelShuffle3D -> final is built in numpy from random weights, its support is
compared with the per-axis reach the kernels use, and a Python mirror of the kernels' unit / group schedules is checked to cover every
(voxel, input row) of the support exactly once, with the issue counts the bench line reports."""
import numpy as np


def lo(o):
    c = (max(o - 1, 0)) >> 1
    return max(c - 1, 0)


def hi(o):
    c = (min(o + 1, 7)) >> 1
    return min(c + 1, 3)


def pair_mask(ph):
    return [oh for oh in range(8) if lo(oh) <= ph <= hi(oh)]


def single_tiles(ph):     # tile ids of a single unit: 0 = rows (1,2), 1 = (3,4), 2 = (5,6), 3 = (0,7)
    return [t for t, ok in enumerate((ph <= 2, True, ph >= 1, True)) if ok]


def cells(pair, od0, tid):
    if pair:
        return (od0, tid), (od0 + 1, tid)
    return ((od0, 0), (od0, 7)) if tid == 3 else ((od0, 2 * tid + 1), (od0, 2 * tid + 2))


UNITS = [(False, 0), (True, 1), (True, 3), (True, 5), (False, 7)]            # tail_rows16_k: (pair unit?, first output plane)
GROUPS = [((True, 1), (False, 0)), ((True, 3), None), ((True, 5), (False, 7))]   # tail_groups16_k: pair part, single part


def composite_support(seed=0):
    """|Wc[voxel][position]| summed over channels for random up_conv / final weights, fp64 (oracle tail_build's definition)."""
    rng = np.random.default_rng(seed)
    Wu = rng.standard_normal((256, 64, 3, 3, 3))          # up_conv: out channel oc*8 + s, s = sub-position of the pixel shuffle
    Wf = rng.standard_normal((32, 3, 3, 3))               # final: 32 -> 1
    Wu = Wu.reshape(32, 8, 64, 3, 3, 3)
    mag = np.zeros((8, 8, 8, 4, 4, 4))
    for od in range(8):
        for oh in range(8):
            for ow in range(8):
                for dd in range(3):
                    for dh in range(3):
                        for dw in range(3):
                            z = (od + dd - 1, oh + dh - 1, ow + dw - 1)
                            if min(z) < 0 or max(z) > 7:
                                continue
                            c = tuple(v >> 1 for v in z)
                            s = (z[0] & 1) * 4 + (z[1] & 1) * 2 + (z[2] & 1)
                            g = np.einsum("o,octuv->ctuv", Wf[:, dd, dh, dw], Wu[:, s])          # [ci][td][th][tw]
                            for td in range(3):
                                for th in range(3):
                                    for tw in range(3):
                                        p = (c[0] + td - 1, c[1] + th - 1, c[2] + tw - 1)
                                        if min(p) < 0 or max(p) > 3:
                                            continue
                                        mag[od, oh, ow, p[0], p[1], p[2]] += np.abs(g[:, td, th, tw]).sum()
    return mag


def test_support_of_the_composite_operator_is_the_per_axis_reach():
    mag = composite_support()
    box = np.zeros_like(mag, dtype=bool)
    for od in range(8):
        for oh in range(8):
            for ow in range(8):
                box[od, oh, ow, lo(od):hi(od) + 1, lo(oh):hi(oh) + 1, lo(ow):hi(ow) + 1] = True
    assert np.array_equal(mag > 0, box)                      # exactly the box: nothing outside, and (generic weights) everything inside
    assert int(box.sum()) * 64 == 884736                      # the structurally non-zero MACs per leaf (x 64 channels)
    assert [hi(o) - lo(o) + 1 for o in range(8)] == [2, 3, 3, 4, 4, 3, 3, 2]


def _schedule_rows16():
    """(unit, pd, ph) -> tiles, as tail_rows16_k walks them; yields (cell pair, pd, ph)."""
    for pair, od0 in UNITS:
        for pd in range(lo(od0), hi(od0) + 1):
            for ph in range(4):
                for tid in (pair_mask(ph) if pair else single_tiles(ph)):
                    yield cells(pair, od0, tid), pd, ph


def _schedule_groups16():
    for (pp, ps) in GROUPS:
        od_p = pp[1]
        for pd in range(lo(od_p), hi(od_p) + 1):
            with_single = ps is not None and lo(ps[1]) <= pd <= hi(ps[1])
            for ph in range(4):
                for tid in pair_mask(ph):
                    yield cells(True, od_p, tid), pd, ph
                if with_single:
                    for tid in single_tiles(ph):
                        yield cells(False, ps[1], tid), pd, ph


def _check(schedule, expected_phases):
    need = {(od, oh, pd, ph) for od in range(8) for oh in range(8) for pd in range(lo(od), hi(od) + 1) for ph in range(lo(oh), hi(oh) + 1)}
    seen, tile_rows, wasted, rows = set(), 0, 0, set()
    for (ca, cb), pd, ph in schedule:
        tile_rows += 1
        for (od, oh) in (ca, cb):
            key = (od, oh, pd, ph)
            assert key not in seen
            seen.add(key)
            if key not in need:
                assert lo(od) <= pd <= hi(od)                # a tile never runs a plane one of its cells cannot reach ...
                wasted += 1                                  # ... only rows: the corner tile (oh = 0 with oh = 7)
    assert need <= seen                                      # every (cell, input row) of the support is issued
    assert tile_rows == 296 and wasted == 16                 # 288 exact + the 2 x 8 half-empty corner tile-rows: 1 212 416 MAC / leaf
    assert tile_rows * 64 * 1024 // 16 == 1212416
    return tile_rows


def test_unit_schedule_covers_the_support_once():
    _check(_schedule_rows16(), 224)
    assert sum((hi(od0) - lo(od0) + 1) * 16 for _, od0 in UNITS) == 224       # phases = weight slices of tail_rows16_k


def test_group_schedule_covers_the_support_once():
    _check(_schedule_groups16(), 160)
    assert sum((hi(pp[1]) - lo(pp[1]) + 1) * 16 for pp, _ in GROUPS) == 160   # phases of tail_groups16_k
    # every tile of a pair unit has ONE reach box (both cells): that is why skipping per tile is exact there
    for _, od0 in (u for u in UNITS if u[0]):
        assert (lo(od0), hi(od0)) == (lo(od0 + 1), hi(od0 + 1))
    for tid in range(3):
        (_, a), (_, b) = cells(False, 0, tid)
        assert (lo(a), hi(a)) == (lo(b), hi(b))
