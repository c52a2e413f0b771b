"""The fused encoder layer (csrc/encoder_fused.hip: dfsfm_encoder_kv_f32 + dfsfm_encoder_apply_f32) on a real MI355X,
through the C ABI, against the oracle's LoFTREncoderLayer restatement evaluated in float64
(src/MultiviewMatcher/matcher_module/transformer.py:66-95, linear_attention.py:28-60).  Shapes are the refinement head's:
225-token reference windows, (V-1) x 225-token query sequences with per-view masks, the 121-token windows of the second
iteration, row counts that are not multiples of the 32 / 128-token tiles; every intermediate the kernel can dump is checked
as well, so a failure names the stage."""
import numpy as np
import pytest
import torch

from detectorfreesfm_amd import ops
from oracle import restate

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
C, H = 128, 8


def _weights(seed):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape, sc in (("q_proj", (C, C), 0.12), ("k_proj", (C, C), 0.12), ("v_proj", (C, C), 0.12), ("merge", (C, C), 0.12),
                            ("mlp.0", (2 * C, 2 * C), 0.09), ("mlp.2", (C, 2 * C), 0.09)):
        sd[f"l.{name}.weight"] = torch.randn(shape, generator=g) * sc
    for nm in ("norm1", "norm2"):
        sd[f"l.{nm}.weight"] = 1.0 + 0.2 * torch.randn(C, generator=g)
        sd[f"l.{nm}.bias"] = 0.2 * torch.randn(C, generator=g)
    return sd


def _fused(sd):
    d = {k: v.to(DEV) for k, v in sd.items()}
    return ops.EncoderFusedWeights(d["l.q_proj.weight"], d["l.k_proj.weight"], d["l.v_proj.weight"], d["l.merge.weight"],
                                   d["l.mlp.0.weight"], d["l.mlp.2.weight"], (d["l.norm1.weight"], d["l.norm1.bias"]),
                                   (d["l.norm2.weight"], d["l.norm2.bias"]))


def _to_split(t, pad_cols=0):
    """fp32 [N, L, C] -> SplitAct view [N, L, C] (optionally the first half of a wider buffer: a row-strided view)."""
    N, L, _ = t.shape
    buf = ops.SplitAct.empty_rows((N, L), C + pad_cols, DEV)
    buf.hi.zero_()
    buf.lo.zero_()
    view = buf.cols(0, C)
    ops.split_rows(t.to(DEV).contiguous(), None, out_split=view)
    return view


def _stages64(sd, x, src, xm, sm):
    """float64 intermediates of LoFTREncoderLayer.forward in the kernel's dump order."""
    sd = {k: v.double() for k, v in sd.items()}
    N, L, _ = x.shape
    S = src.shape[1]
    q = x @ sd["l.q_proj.weight"].T
    k = src @ sd["l.k_proj.weight"].T
    v = src @ sd["l.v_proj.weight"].T
    msg = restate.linear_attention(q.view(N, L, H, 16), k.view(N, S, H, 16), v.view(N, S, H, 16), xm, sm).reshape(N, L, C)
    m1 = torch.nn.functional.layer_norm(msg @ sd["l.merge.weight"].T, (C,), sd["l.norm1.weight"], sd["l.norm1.bias"])
    o = torch.relu(torch.cat([x, m1], -1) @ sd["l.mlp.0.weight"].T) @ sd["l.mlp.2.weight"].T
    out = x + torch.nn.functional.layer_norm(o, (C,), sd["l.norm2.weight"], sd["l.norm2.bias"])
    return {1: q, 2: msg, 3: m1, 4: o, 5: out}


CASES = [
    # N, L, S, q_group (0 = no mask), kv_group, self?
    (7, 225, 225, 0, 0, True),            # reference windows, self attention
    (5, 900, 900, 225, 225, True),        # query sequences (4 views), self attention with per-view masks on both sides
    (5, 900, 225, 225, 0, False),         # cross: queries attend to the reference window
    (5, 225, 900, 0, 225, False),         # cross: reference attends to the masked query views
    (9, 121, 363, 0, 121, False),         # second refinement iteration: W = 11
    (3, 3375, 225, 225, 0, False),        # 15 query views (max_track_length 16)
    (1, 4800, 4800, 0, 0, True),          # one long sequence: 150 blocks through the 4 waves of one workgroup
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_fused_layer_vs_fp64(built_lib, case):
    N, L, S, qg, kg, is_self = CASES[case]
    sd = _weights(10 + case)
    fw = _fused(sd)
    g = torch.Generator().manual_seed(100 + case)
    x = torch.randn((N, L, C), generator=g)
    src = x if is_self else torch.randn((N, S, C), generator=g)
    xm = sm = xmask = smask = None
    if qg:
        xmask = torch.rand((N, L // qg), generator=g) > 0.3
        xmask[:, 0] = True
        xm = xmask.repeat_interleave(qg, dim=1).double()
    if kg:
        smask = xmask if (is_self and qg) else (torch.rand((N, S // kg), generator=g) > 0.3)
        smask[:, 0] = True
        sm = smask.repeat_interleave(kg, dim=1).double()
    xs = _to_split(x, pad_cols=C)                    # x lives in the first half of a [., 2C] buffer like in the product
    ss = xs if is_self else _to_split(src)
    # the kernels see the 22-bit split values: the reference starts from exactly those
    x64, s64 = xs.float().double().cpu(), ss.float().double().cpu()
    ref = _stages64(sd, x64, s64, xm, sm)
    state = ops.encoder_kv(ss, fw, smask.to(DEV) if smask is not None else None, kg or 1)
    out_s = ops.SplitAct.empty_rows((N, L), C, DEV)
    out32 = torch.empty((N, L, C), dtype=torch.float32, device=DEV)
    worst = {}
    for stage in (1, 2, 3, 4):
        dbg = ops.encoder_apply(xs, fw, state, S, xmask.to(DEV) if xmask is not None else None, qg or 1, out_split=out_s,
                                out=out32, debug_stage=stage)
        r = ref[stage].reshape(-1, C)
        worst[stage] = float((dbg.double().cpu() - r).abs().max() / r.abs().max())
    out = out32.double().cpu()
    worst[5] = float((out - ref[5]).abs().max() / ref[5].abs().max())
    print(f"[fused encoder case {case}: N={N} L={L} S={S}] relative errors q/msg/norm1/mlp/out: "
          + " ".join(f"{worst[k]:.1e}" for k in (1, 2, 3, 4, 5)))
    assert all(worst[k] < 2e-5 for k in worst), worst
    assert torch.equal(out_s.float(), out32)                               # the fp32 form is the exact value of the planes
    # deterministic
    out_b = ops.SplitAct.empty_rows((N, L), C, DEV)
    state_b = ops.encoder_kv(ss, fw, smask.to(DEV) if smask is not None else None, kg or 1)
    ops.encoder_apply(xs, fw, state_b, S, xmask.to(DEV) if xmask is not None else None, qg or 1, out_split=out_b)
    # (the state's fragments are written only in the lanes that hold a head's rows; the other slots are never read)
    img = lambda t: t[:, :16384].view(N, 8, 2, 2, 32, 16)              # [n, fragment (b, t), plane, lane half, lane & 31, bytes]
    for f in range(8):
        rows = slice(0, 16) if f % 2 == 0 else slice(16, 32)
        assert torch.equal(img(state)[:, f, :, :, rows], img(state_b)[:, f, :, :, rows])
    assert torch.equal(state[:, 16384:], state_b[:, 16384:])
    assert torch.equal(out_b.hi, out_s.hi) and torch.equal(out_b.lo, out_s.lo)


def test_fused_layer_equals_unfused_path(built_lib):
    """The five-GEMM path of encoder_layer_split (kept for d_model 256) and the fused kernels agree to fp32 noise, and a
    sequence's result does not depend on what else is in the batch (tracks are independent units)."""
    from detectorfreesfm_amd import coarse
    sd = _weights(3)
    get = lambda name: sd["l." + name].to(DEV)
    w = coarse.EncoderLayerWeights(get, "")
    assert w.fused is not None
    g = torch.Generator().manual_seed(4)
    N, L = 6, 225
    x = torch.randn((N, L, C), generator=g)
    xs = _to_split(x, pad_cols=C)
    full = ops.SplitAct(xs.hi.as_strided((N, L, 2 * C), xs.hi.stride()), xs.lo.as_strided((N, L, 2 * C), xs.lo.stride()), 2 * C)
    out_f = torch.empty((N, L, C), device=DEV)
    coarse.encoder_layer_split(w, full, full.cols(0, C), out_f, None, H, is_self=True)
    fused, w.fused = w.fused, None
    out_u = torch.empty((N, L, C), device=DEV)
    coarse.encoder_layer_split(w, full, full.cols(0, C), out_u, None, H, is_self=True)
    w.fused = fused
    assert ((out_f - out_u).abs().max() / out_u.abs().max()).item() < 2e-5
    sub = ops.SplitAct(full.hi[2:5], full.lo[2:5], 2 * C)
    out_s = torch.empty((3, L, C), device=DEV)
    coarse.encoder_layer_split(w, sub, sub.cols(0, C), out_s, None, H, is_self=True)
    assert torch.equal(out_s, out_f[2:5])


def test_fused_layer_argument_checks(built_lib):
    sd = _weights(1)
    fw = _fused(sd)
    xs = _to_split(torch.zeros((2, 16, C)))
    state = ops.encoder_kv(xs, fw)
    from detectorfreesfm_amd._lib import DfsfmError
    with pytest.raises(DfsfmError):                  # 16-token sequences: a 32-token tile could touch three of them
        ops.encoder_apply(xs, fw, state, 16, out_split=ops.SplitAct.empty_rows((2, 16), C, DEV))
    with pytest.raises(DfsfmError):
        ops.EncoderFusedWeights(torch.zeros(256, 256), torch.zeros(256, 256), torch.zeros(256, 256), torch.zeros(256, 256),
                                torch.zeros(512, 512), torch.zeros(256, 512), (torch.ones(256), torch.zeros(256)),
                                (torch.ones(256), torch.zeros(256)))


def test_fused128_kernels_bit_reproducible_at_full_occupancy(built_lib):
    """encoder_fused.hip keeps packed-fp32 VALU instructions (see tests/test_gpu_encoder256.py::test_fused256_kernels_bit_reproducible_
    at_full_occupancy for why that is asserted rather than assumed): 600 tracks x 900 query tokens -- every CU walks ~16 tiles of the
    persistent grid -- give the same bits run after run and the same bits per track as a 3-track launch."""
    N, L, S = 600, 900, 225
    sd = _weights(4)
    fw = _fused(sd)
    g = torch.Generator().manual_seed(23)
    xs, ss = _to_split(torch.randn((N, L, C), generator=g)), _to_split(torch.randn((N, S, C), generator=g))

    def run(x, s):
        out = ops.SplitAct.empty_rows((x.hi.shape[0], L), C, DEV)
        state = ops.encoder_kv(s, fw)
        ops.encoder_apply(x, fw, state, S, out_split=out)
        return out.hi.clone(), out.lo.clone()
    first = run(xs, ss)
    for _ in range(6):
        again = run(xs, ss)
        assert torch.equal(again[0], first[0]) and torch.equal(again[1], first[1])
    part = run(xs[297:300], ss[297:300])
    assert torch.equal(part[0], first[0][297:300]) and torch.equal(part[1], first[1][297:300])
