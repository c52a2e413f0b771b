"""Per-entry parity rules shared by the GPU end-to-end tests and ``__graft_entry__.smoke()``.

north_star: match indices bit-exact, confidences / sub-pixel offsets within 1e-4.  Two kinds of entries can
legitimately differ between two correct fp32 evaluations whose summation order differs (GPU tiles vs CPU loops),
and ONLY those are exempted -- one entry at a time, never as a percentage:

* coarse: an entry whose oracle confidence lies within ``TOL_THR`` of the threshold (the ``conf > thr`` test
  may go either way), or whose confidence is within ``TOL_TIE`` of the competing maximum of its row / column
  (the ``conf == max`` mutual-nearest-neighbour test, coarse_matching.py:171-177, may pick the neighbour);
* refinement: a track whose two best candidate scores (mean std over valid views, fine_matching.py:129-179)
  differ by less than ``TOL_SCORE`` in the oracle (the first-minimum argmin may pick the other candidate).

* coarse, rule "oracle-noise" (opt-in through ``exact=``, used by exactly two tests whose studies are committed): a confidence
  that differs from the fp32 oracle's by more than 1e-4 is accepted iff it lies within 1e-4 of the EXACT value of that entry
  and the fp32 oracle itself is within ``TOL_ORACLE_NOISE`` of that value -- i.e. where the oracle's own fp32 rounding, not the
  product, is what breaks the 1e-4.  The rule never accepts a value that is not within north_star's 1e-4 of the exact confidence.
    - LoFTR at the reference's production frame sizes (``exact=(feat_c0, feat_c1, temperature)``: the float64 dual-softmax of
      the oracle's own fp32 features, ``exact_conf_at``).  tools/studies/loftr_hires_noise_study.py
      (profiles/r04_loftr_hires_noise_study.txt): a confidence is a ratio of sums over L competitors; ATen's fp32 CPU softmax
      over 26 600 entries (1600x1064 frames) is up to 1.15e-4 (mean +2.9e-5, biased high) from the exact value of its own
      inputs, 3.3e-5 at 640x480; the features contribute 1.7e-5.
    - ASpanFormer at 640x480 (``exact=callable`` returning the float64 evaluation of the whole oracle).
      tools/studies/aspan_noise_study.py (profiles/r03_aspan_noise_study.txt): the network amplifies rounding noise ~500x, the
      fp32 oracle sits 8.8e-5 from its own float64 evaluation, another fp32 summation order 2.1e-4 from the oracle.

Every exemption is returned so the caller can print / bound the list.
"""
import numpy as np
import torch

TOL_CONF = 1e-4     # north_star tolerance on confidences
TOL_THR = 1e-5      # |conf - thr| below which the threshold test is undecidable in fp32
TOL_TIE = 1e-6      # |conf - competing max| below which the mutual-max test is undecidable
TOL_PX = 1e-4       # north_star tolerance on refined coordinates / std
TOL_SCORE = 1e-5    # candidate-score gap below which the argmin is undecidable
TOL_ORACLE_NOISE = 2e-4   # bound on the fp32 oracle's own distance from the exact value under the "oracle-noise" rule (measured <= 1.15e-4)


def exact_conf_at(feat0, feat1, temperature, b, i, j, chunk=2048):
    """Dual-softmax confidences (coarse_matching.py:103-116) of the entries (b, i, j) in float64, from features [N,L,C] /
    [N,S,C], without the dense matrix: row log-sum-exps per chunk of rows, column log-sum-exps as running (max, sum)."""
    f0, f1 = feat0.detach().cpu().double(), feat1.detach().cpu().double()
    b, i, j = (torch.as_tensor(_np(x)).long() for x in (b, i, j))
    C = f0.shape[-1]
    out = torch.empty(len(b), dtype=torch.float64)
    for n in torch.unique(b).tolist():
        sel = (b == n).nonzero()[:, 0]
        a, c = f0[n] / C ** 0.5, f1[n] / C ** 0.5
        L, S = a.shape[0], c.shape[0]
        row_lse = torch.empty(L, dtype=torch.float64)
        col_m = torch.full((S,), -float("inf"), dtype=torch.float64)
        col_s = torch.zeros(S, dtype=torch.float64)
        for lo in range(0, L, chunk):
            sim = (a[lo:lo + chunk] @ c.T) / temperature
            row_lse[lo:lo + chunk] = torch.logsumexp(sim, 1)
            m = torch.maximum(col_m, sim.max(0)[0])
            col_s = col_s * torch.exp(col_m - m) + torch.exp(sim - m).sum(0)
            col_m = m
        col_lse = col_m + torch.log(col_s)
        ii, jj = i[sel], j[sel]
        sv = (a[ii] * c[jj]).sum(-1) / temperature
        out[sel] = torch.exp(sv - row_lse[ii]) * torch.exp(sv - col_lse[jj])
    return out.numpy()


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def check_coarse(hip, ref, conf, thr, border=None, exact=None):
    """hip / ref: dicts with b_ids, i_ids, j_ids, mconf (+ optional mkpts*); conf: the ORACLE's dense
    confidence matrix [N,L,S].  Asserts the per-entry rules above; returns the list of exempted entries.
    ``exact`` switches the "oracle-noise" rule on (module docstring): (oracle feat_c0, feat_c1, temperature), or a callable
    (b, i, j arrays) -> exact confidences."""
    tol_conf = TOL_CONF
    conf = _np(conf).astype(np.float64)
    hb, hi_, hj, hc = (_np(hip[k]) for k in ("b_ids", "i_ids", "j_ids", "mconf"))
    rb, ri, rj, rc = (_np(ref[k]) for k in ("b_ids", "i_ids", "j_ids", "mconf"))
    H = {(int(b), int(i)): (int(j), float(c)) for b, i, j, c in zip(hb, hi_, hj, hc)}
    R = {(int(b), int(i)): (int(j), float(c)) for b, i, j, c in zip(rb, ri, rj, rc)}
    assert len(H) == len(hb) and len(R) == len(rb), "duplicate (b, i) rows"
    rowmax, colmax = conf.max(2), conf.max(1)

    def selectable(b, i, j):
        """(b,i,j) passes threshold and mutual-max tests within the undecidable margins."""
        c = conf[b, i, j]
        return c > thr - TOL_THR and c >= rowmax[b, i] - TOL_TIE and c >= colmax[b, j] - TOL_TIE

    def fragile(b, i, j):
        """the oracle's own selection of (b,i,j) hangs on a margin smaller than the tolerances."""
        c = conf[b, i, j]
        row = conf[b, i].copy()
        row[j] = -1.0
        col = conf[b, :, j].copy()
        col[i] = -1.0
        return abs(c - thr) <= TOL_THR or c - row.max() <= TOL_TIE or c - col.max() <= TOL_TIE

    exempt, bad, sums = [], [], []
    for key in sorted(set(H) | set(R)):
        b, i = key
        if key in H and key in R:
            (j, c), (jr, cr) = H[key], R[key]
            if j == jr:
                if abs(c - cr) > tol_conf:
                    (sums if exact is not None else bad).append(("conf", key, j, c, cr))
            elif selectable(b, i, j) and fragile(b, i, jr):
                exempt.append(("tie", key, j, jr))
            else:
                bad.append(("j differs", key, j, jr, conf[b, i, j], conf[b, i, jr]))
        elif key in H:
            j, c = H[key]
            if abs(c - conf[b, i, j]) > tol_conf:
                bad.append(("conf of extra entry", key, j, c, conf[b, i, j]))
            elif selectable(b, i, j) and (abs(conf[b, i, j] - thr) <= TOL_THR or conf[b, i, j] < rowmax[b, i]
                                          or conf[b, i, j] < colmax[b, j]):
                exempt.append(("extra", key, j, conf[b, i, j]))
            else:
                bad.append(("extra entry", key, j, c, conf[b, i, j], rowmax[b, i], colmax[b, j]))
        else:
            j, c = R[key]
            if fragile(b, i, j):
                exempt.append(("missing", key, j, c))
            else:
                bad.append(("missing entry", key, j, c))
    if sums:     # rule "oracle-noise": judge these entries against their exact confidence
        bb, ii, jj = [e[1][0] for e in sums], [e[1][1] for e in sums], [e[2] for e in sums]
        ex = exact(np.asarray(bb), np.asarray(ii), np.asarray(jj)) if callable(exact) else \
            exact_conf_at(exact[0], exact[1], exact[2], bb, ii, jj)
        for (_, key, j, c, cr), ce in zip(sums, ex):
            if abs(c - ce) <= TOL_CONF and abs(cr - ce) <= TOL_ORACLE_NOISE:
                exempt.append(("oracle-noise", key, j, c, cr, float(ce)))
            else:
                bad.append(("conf (also off the exact value)", key, j, c, cr, float(ce)))
    assert not bad, (len(bad), bad[:8])
    return exempt


def check_coarse_rows(hip, ref, exempt):
    """Rows present in both tables: pixel coordinates identical (integer grid x scale), same order."""
    skip = {e[1] for e in exempt}
    hb, hi_ = _np(hip["b_ids"]), _np(hip["i_ids"])
    rb, ri = _np(ref["b_ids"]), _np(ref["i_ids"])
    hk = [(int(b), int(i)) for b, i in zip(hb, hi_)]
    rk = [(int(b), int(i)) for b, i in zip(rb, ri)]
    assert hk == sorted(hk), "rows must be in ascending (b, i) order like torch.where"
    hsel = [n for n, k in enumerate(hk) if k not in skip]
    rsel = [n for n, k in enumerate(rk) if k not in skip]
    assert [hk[n] for n in hsel] == [rk[n] for n in rsel]
    for name in ("mkpts0_f", "mkpts1_f", "mkpts0_c", "mkpts1_c"):
        if name in hip and name in ref:
            assert np.array_equal(_np(hip[name])[hsel], _np(ref[name])[rsel]), name


def check_refine(hip_q, hip_r, hip_std, ref_q, ref_r, ref_std, valid, query_pts, q_scale, cand_score, left,
                 tol_px=TOL_PX):
    """hip_q/ref_q [T,2], hip_r/ref_r [Vq,T,2], hip_std/ref_std [Vq,T], valid [Vq,T] bool, query_pts [T,2]
    (input, original scale), q_scale [T,2] (pixel scale of the query view), cand_score [T, left*left] (oracle).
    A track may pick another candidate only if the oracle's scores of the two are within TOL_SCORE; such tracks
    are listed and excluded from the coordinate comparison (another left point = other heat-maps); every other
    track must agree within tol_px on all three outputs.  Returns the list of flipped tracks."""
    hip_q, ref_q, hip_r, ref_r, hip_std, ref_std = (_np(x).astype(np.float64) for x in
                                                    (hip_q, ref_q, hip_r, ref_r, hip_std, ref_std))
    valid, qp, qs, sc = _np(valid).astype(bool), _np(query_pts).astype(np.float64), _np(q_scale).astype(np.float64), \
        _np(cand_score).astype(np.float64)
    T = hip_q.shape[0]
    r = left // 2

    def cand(q):     # refined query point -> candidate index (build_moved_query, fine_matching.py:221-232)
        off = np.rint((q - qp) / qs + r).astype(np.int64)
        return off[:, 1] * left + off[:, 0]
    ch, cr = cand(hip_q), cand(ref_q)
    flips, bad = [], []
    for t in np.nonzero(ch != cr)[0]:
        gap = abs(sc[t, ch[t]] - sc[t, cr[t]])
        (flips if gap < TOL_SCORE else bad).append((int(t), int(ch[t]), int(cr[t]), float(gap)))
    assert not bad, ("argmin differs on decidable scores", bad[:8])
    same = ch == cr
    dq = np.abs(hip_q - ref_q).max(-1)
    assert dq[same].max(initial=0.0) <= tol_px, dq[same].max()
    m = valid & same[None, :]
    dr = np.abs(hip_r - ref_r).max(-1)
    assert dr[m].max(initial=0.0) <= tol_px, dr[m].max()
    ds = np.abs(hip_std - ref_std)
    assert ds[m].max(initial=0.0) <= tol_px, ds[m].max()
    return flips
