"""The L2-aware tile order of cm_gemm_sf2_cand (csrc/coarse_match.hip, r05) restated in Python: workgroup b runs on XCD b % 8 and
takes tile order index t = (b % 8) * per_xcd + b // 8 (dfsfm_sf::xcd_band_tile); t is mapped to (tm, tn) by row GROUPS of sb tile
rows, column tile by column tile inside a group.  Checked here without a GPU: the mapping is a bijection onto the tile grid for every
shape the tests and workloads use (so no tile is skipped or computed twice -- the kernel's statistics are sums over tiles), and it
has the locality it exists for: the 64 tiles an XCD runs at a time span at most two row groups."""
import pytest


def host_sb(ntm):
    return min(12, max(1, (ntm + 7) // 8))            # coarse_match_impl: g.sb


def tile_of(t_id, ntm, ntn, sb):
    gsz = sb * ntn
    grp, rem = divmod(t_id, gsz)
    rows_g = min(sb, ntm - grp * sb)
    return grp * sb + rem % rows_g, rem // rows_g       # (tm, tn)


@pytest.mark.parametrize("L,S", [(4800, 4800), (10816, 10816), (15000, 15000), (26600, 26600), (192, 192), (4800, 140), (937, 1663),
                                 (128, 128), (129, 5000), (26600, 15000), (3000, 3000)])
def test_traversal_is_a_bijection(L, S):
    ntm, ntn = (L + 127) // 128, (S + 127) // 128
    sb = host_sb(ntm)
    ntiles = ntm * ntn
    grid = (ntiles + 7) // 8 * 8
    per_xcd = grid // 8
    seen = set()
    for b in range(grid):
        t = (b & 7) * per_xcd + (b >> 3)
        if t >= ntiles:
            continue
        tm, tn = tile_of(t, ntm, ntn, sb)
        assert 0 <= tm < ntm and 0 <= tn < ntn, (b, t, tm, tn)
        seen.add((tm, tn))
    assert len(seen) == ntiles


def test_an_xcd_keeps_few_f0_rows_hot():
    """8 pairs of 4800 x 4800: what the XCD's 64 concurrent tiles (32 CUs x 2 workgroups) touch -- at most 2 row groups = 10 tile
    rows of f0 (1.25 MB of a 4-MB L2) against all 38 column tiles of f1 in the row-major order it replaces."""
    ntm = ntn = 38
    sb = host_sb(ntm)
    assert sb == 5
    ntiles = ntm * ntn
    per_xcd = ((ntiles + 7) // 8 * 8) // 8
    for xcd in range(8):
        order = [tile_of(t, ntm, ntn, sb) for t in range(xcd * per_xcd, min(ntiles, (xcd + 1) * per_xcd))]
        for k in range(0, len(order) - 63, 16):
            window = order[k:k + 64]
            rows = {tm for tm, _ in window}
            cols = {tn for _, tn in window}
            assert len({tm // sb for tm in rows}) <= 2 and len(rows) <= 2 * sb
            assert len(cols) <= 64 // min(sb, 3) + 2            # a window walks ~13 column tiles, not all 38
