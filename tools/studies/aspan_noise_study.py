"""CPU study (not product code): where does ASpanFormer's confidence noise at 640x480 come from?

Runs HipASpanFormer's HOST logic on the CPU stand-ins (tests/cpu_standins.py emulate the split-GEMM algebra exactly) and
compares the confidences of the oracle's matches with the fp64 evaluation of the oracle, for several arithmetic variants:
  split     every GEMM operand and every stored activation in the 22-bit fp16x2 form (what the GPU does)
  exact     stand-in planes hold fp32 values, weights unrounded: plain fp32 arithmetic in another summation order
  exact:<stage,...>  exact only inside the named stages (backbone, ini, gla0..gla3, fd), split elsewhere
usage: python tools/studies/aspan_noise_study.py variant [variant ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cpu_standins as cs  # noqa: E402
from detectorfreesfm_amd import ops, synth  # noqa: E402
from detectorfreesfm_amd import aspanformer as A  # noqa: E402
from detectorfreesfm_amd.params import aspanformer_param_spec, planted_aspanformer_state_dict  # noqa: E402
from oracle import restate_aspanformer as ra  # noqa: E402

STATE = {"exact": False, "hp": "exact"}     # hp: what an "exact" stage computes in: "exact" (fp32) or "f64"
_orig_split = cs._split


def _split(x):
    if STATE["exact"]:
        return x, torch.zeros_like(x)
    hi, lo = _orig_split(x.float())
    return hi.to(x.dtype), lo.to(x.dtype)


PLANE = torch.float64        # dtype of the stand-in planes and of every buffer the forward allocates as "fp32"


class _TorchProxy:
    """``torch`` for the aspanformer module with float32 -> PLANE, so the forward's own fp32 buffers do not truncate."""

    def __getattr__(self, name):
        return PLANE if name == "float32" else getattr(torch, name)


def patch():
    cs._split = _split
    # planes become fp32 tensors so that exact mode can store unrounded values; in split mode the values written are
    # fp16-representable anyway
    def empty_rows(shape, C, device):
        return ops.SplitAct(torch.zeros((*shape, C), dtype=PLANE, device=device),
                            torch.zeros((*shape, C), dtype=PLANE, device=device), C)

    def empty(N, H, W, C, device):
        cp = (C + 7) // 8 * 8
        return ops.SplitAct(torch.zeros((N, H, W, cp), dtype=PLANE, device=device),
                            torch.zeros((N, H, W, cp), dtype=PLANE, device=device), C)
    ops.SplitAct.empty_rows = staticmethod(empty_rows)
    ops.SplitAct.empty = staticmethod(empty)

    def _rows_split(y):
        hi, lo = cs._split(y)
        return ops.SplitAct(hi.clone(), lo.clone(), y.shape[-1])

    def _put_split(dst, y):
        hi, lo = cs._split(y)
        dst.hi.copy_(hi.reshape(dst.hi.shape))
        dst.lo.copy_(lo.reshape(dst.lo.shape))

    def _to_split(y):
        C = y.shape[-1]
        cp = (C + 7) // 8 * 8
        hi, lo = cs._split(y)
        H = torch.zeros((*y.shape[:3], cp), dtype=PLANE)
        L = torch.zeros((*y.shape[:3], cp), dtype=PLANE)
        H[..., :C], L[..., :C] = hi, lo
        return ops.SplitAct(H, L, C)
    cs._rows_split, cs._put_split, cs._to_split = _rows_split, _put_split, _to_split
    orig_init = ops.PackedDense.__init__

    def init(self, w, bias=None, cin_pad=None, tap_padded=False):
        orig_init(self, w, bias, cin_pad, tap_padded)
        w4 = w if w.dim() == 4 else w[:, :, None, None]
        self._w_exact = w4.detach().to(PLANE)
    ops.PackedDense.__init__ = init
    orig_conv = cs._conv

    def conv(x, pw, stride=1, pad=0, residual=None, relu=False, out=None, out_split=False):
        import torch.nn.functional as F
        if not STATE["exact"]:
            # the split algebra in fp32 on whatever the planes hold (fp16-representable values)
            xs = x if isinstance(x, ops.SplitAct) else None
            if xs is not None:
                x = ops.SplitAct(xs.hi.float(), xs.lo.float(), xs.C)
            else:
                x = x.float()
            res = residual
            if isinstance(res, ops.SplitAct):
                res = res.float().float()
            elif res is not None:
                res = res.float()
            y = orig_conv(x, pw, stride, pad, res, relu, None, False).to(PLANE)
            if out_split:
                return cs._to_split(y)
            if out is None:
                return y.contiguous()
            out.copy_(y.reshape(out.shape))
            return out
        cd = torch.float64 if STATE["hp"] == "f64" else torch.float32
        xv = (x.hi + x.lo / 2048.0) if isinstance(x, ops.SplitAct) else x
        xv = xv[..., :pw._w_exact.shape[1]]
        y = F.conv2d(xv.permute(0, 3, 1, 2).to(cd), pw._w_exact.to(cd), None, stride, pad).permute(0, 2, 3, 1).to(PLANE)
        if pw.bias is not None:
            y = y + pw.bias
        if residual is not None:
            y = y + (residual.float() if isinstance(residual, ops.SplitAct) else residual).reshape(y.shape)
        if relu == 2:
            y = torch.nn.functional.leaky_relu(y, 0.01)
        elif relu:
            y = torch.relu(y)
        if out_split:
            return cs._to_split(y)
        if out is None:
            return y.contiguous()
        out.copy_(y.reshape(out.shape))
        return out
    cs._conv = conv
    ops.SplitAct.float = lambda self: (self.hi + self.lo / 2048.0)[..., :self.C]


def stage_hooks(stages):
    """Switch STATE['exact'] on inside the named stages by wrapping the functions the forward calls at stage boundaries."""
    if stages is None:
        return
    orig_bb = A.backbone_tokens_hip

    def bb(*a, **k):
        STATE["exact"] = "backbone" in stages
        # finer: conv call index inside the backbone -> stem (0), layer1 (1-4), layer2 (5-9), layer3 (10-14), out conv (15)
        names = ["bb_stem"] + ["bb_l1"] * 4 + ["bb_l2"] * 5 + ["bb_l3"] * 5 + ["bb_out"]
        cnt = [0]
        inner = ops.conv2d_nhwc

        def conv(*aa, **kk):
            nm = names[cnt[0]]
            cnt[0] += 1
            prev = STATE["exact"]
            STATE["exact"] = prev or nm in stages
            try:
                return inner(*aa, **kk)
            finally:
                STATE["exact"] = prev
        ops.conv2d_nhwc = conv
        try:
            return orig_bb(*a, **k)
        finally:
            ops.conv2d_nhwc = inner
            STATE["exact"] = False
    A.backbone_tokens_hip = bb
    # the transformer stages are told apart by the packed-weight dict a linear call receives
    orig_linear = cs._linear
    tags = {}

    def linear(x, pw, *a, **k):
        t = tags.get(id(pw))
        STATE["exact"] = t in stages if t else STATE["exact"]
        try:
            return orig_linear(x, pw, *a, **k)
        finally:
            if t:
                STATE["exact"] = False
    cs._linear = linear
    return tags


def main():
    variants = sys.argv[1:] or ["split", "exact"]
    patch()
    A.torch = _TorchProxy()
    import detectorfreesfm_amd.coarse as Cmod
    orig_cm = cs._cm
    cs._cm = lambda f0, f1, *a, **k: orig_cm(ops.SplitAct(f0.hi.float(), f0.lo.float(), f0.C), ops.SplitAct(f1.hi.float(), f1.lo.float(), f1.C), *a, **k)
    cfg = A.aspanformer_coarse_only_config(0.2)
    sd = planted_aspanformer_state_dict(aspanformer_param_spec(cfg), 0)
    data = synth.coarse_pair_batch(1, 480, 640, seed=7)
    with torch.no_grad():
        o = ra.aspanformer_forward(sd, cfg, data, with_fine_backbone=False)
        torch.set_default_dtype(torch.float64)
        o64 = ra.aspanformer_forward({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, cfg,
                                     {k: v.double() for k, v in data.items()}, with_fine_backbone=False)
        torch.set_default_dtype(torch.float32)
    c64 = o64["conf_matrix"][0].numpy()
    c32 = o["conf_matrix"][0].double().numpy()
    oi, oj = o["i_ids"].numpy(), o["j_ids"].numpy()
    print(f"oracle: {len(oi)} matches; fp32-vs-fp64 noise over the whole matrix {np.abs(c32 - c64).max():.2e}, "
          f"over the matches {np.abs(c32[oi, oj] - c64[oi, oj]).max():.2e}", flush=True)
    for v in variants:
        m = A.HipASpanFormer(cfg)
        m.load_state_dict(sd, strict=True)
        m = m.eval()
        kind = v.split(":", 1)[0]
        STATE["hp"] = "f64" if kind == "f64" else "exact"
        stages = set(v.split(":", 1)[1].split(",")) if ":" in v else None
        STATE["exact"] = v in ("exact", "f64")
        with cs.cpu_ops(), torch.no_grad():
            if stages is not None:
                orig_bb = A.backbone_tokens_hip
                tags = stage_hooks(stages)
                P = m._pack()
                for e in P["ini"]:
                    for pw in e.values():
                        if isinstance(pw, ops.PackedDense):
                            tags[id(pw)] = "ini"
                for nm in ("dec", "upm"):
                    tags[id(P[nm])] = "ini"
                for li, e in enumerate(P["gla"]):
                    for k, pw in e.items():
                        if isinstance(pw, ops.PackedDense):
                            tags[id(pw)] = "fd" if k in ("fd0", "fd2") else f"gla{li}"
                from cpu_standins import _linear  # noqa: F401
                ops.linear = cs._linear
                ops.conv2d_nhwc = cs._conv
            d = dict(data)
            m(d)
            if stages is not None:
                A.backbone_tokens_hip = orig_bb
        STATE["exact"] = False
        hi, hj, hc = (d[k].numpy() for k in ("i_ids", "j_ids", "mconf"))
        same = len(hi) == len(oi) and np.array_equal(hi, oi) and np.array_equal(hj, oj)
        dev64 = np.abs(hc - c64[hi, hj])
        dev32 = np.abs(hc - c32[hi, hj])
        print(f"{v:28s} rows identical: {same}; vs fp64 max {dev64.max():.2e} mean {dev64.mean():.2e}; vs fp32 oracle max "
              f"{dev32.max():.2e}, rows beyond 1e-4: {int((dev32 > 1e-4).sum())}", flush=True)


if __name__ == "__main__":
    main()
