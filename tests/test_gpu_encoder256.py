"""The d_model-256 fused encoder layer (csrc/encoder256.hip: dfsfm_encoder256_state_f32 + dfsfm_encoder256_apply_f32) on a
real MI355X, through the C ABI, against the oracle's LoFTREncoderLayer restatement evaluated in float64
(third_party/LoFTR/src/loftr/loftr_module/transformer.py:35-58, linear_attention.py:20-47).  Shapes are the coarse
transformer's: 4800-token sequences (640x480), the ragged 15 000 / 26 600-token grids of the production frame sizes (not
multiples of the 16-token wave tile: tiles straddle sequences), padding masks, small grids; every intermediate the kernel
can dump is checked as well, so a failure names the stage."""
import pytest
import torch

from detectorfreesfm_amd import ops
from oracle import restate

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
C, H, D = 256, 8, 32


def _weights(seed):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape, sc in (("q_proj", (C, C), 0.09), ("k_proj", (C, C), 0.09), ("v_proj", (C, C), 0.09), ("merge", (C, C), 0.09),
                            ("mlp.0", (2 * C, 2 * C), 0.06), ("mlp.2", (C, 2 * C), 0.06)):
        sd[f"l.{name}.weight"] = torch.randn(shape, generator=g) * sc
    for nm in ("norm1", "norm2"):
        sd[f"l.{nm}.weight"] = 1.0 + 0.2 * torch.randn(C, generator=g)
        sd[f"l.{nm}.bias"] = 0.2 * torch.randn(C, generator=g)
    return sd


def _fused(sd):
    d = {k: v.to(DEV) for k, v in sd.items()}
    return ops.Encoder256Weights(d["l.q_proj.weight"], d["l.merge.weight"], d["l.mlp.0.weight"], d["l.mlp.2.weight"],
                                 (d["l.norm1.weight"], d["l.norm1.bias"]), (d["l.norm2.weight"], d["l.norm2.bias"]),
                                 wk=d["l.k_proj.weight"], wv=d["l.v_proj.weight"])


def _to_split(t, pad_cols=0):
    N, L, _ = t.shape
    buf = ops.SplitAct.empty_rows((N, L), C + pad_cols, DEV)
    buf.hi.zero_()
    buf.lo.zero_()
    view = buf.cols(0, C)
    ops.split_rows(t.to(DEV).contiguous(), None, out_split=view)
    return view


def _stages64(sd, x, k, v, xm, sm):
    """float64 intermediates of the query side of LoFTREncoderLayer.forward in the kernel's dump order (k, v given)."""
    sd = {n: w.double() for n, w in sd.items()}
    N, L, _ = x.shape
    S = k.shape[1]
    q = x @ sd["l.q_proj.weight"].T
    msg = restate.linear_attention(q.view(N, L, H, D), k.view(N, S, H, D), v.view(N, S, H, D), xm, sm).reshape(N, L, C)
    m1 = torch.nn.functional.layer_norm(msg @ sd["l.merge.weight"].T, (C,), sd["l.norm1.weight"], sd["l.norm1.bias"])
    o = torch.relu(torch.cat([x, m1], -1) @ sd["l.mlp.0.weight"].T) @ sd["l.mlp.2.weight"].T
    out = x + torch.nn.functional.layer_norm(o, (C,), sd["l.norm2.weight"], sd["l.norm2.bias"])
    return {1: q, 2: msg, 3: m1, 4: o, 5: out}


CASES = [
    # N, L, S, masks?
    (2, 4800, 4800, False),           # BASELINE configs[1] grid
    (3, 300, 192, False),             # small grids, L not a multiple of 16 x 4: partial tiles, tiles straddling sequences
    (2, 937, 1663, True),             # odd lengths with padding masks on both sides
    (1, 26600, 15000, False),         # production grids: 1600x1064 queries attend to a 1200x800 source
    (5, 16, 40, True),                # the minimum sequence length: every wave tile is one sequence
    (4, 23, 23, False),               # L < 32: a 64-row tile spans three sequences, a wave tile at most two
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_fused256_layer_vs_fp64(built_lib, case):
    N, L, S, masked = CASES[case]
    sd = _weights(20 + case)
    fw = _fused(sd)
    g = torch.Generator().manual_seed(200 + case)
    x = torch.randn((N, L, C), generator=g)
    src = torch.randn((N, S, C), generator=g)
    xm = sm = xmask = smask = None
    if masked:
        xmask = torch.rand((N, L), generator=g) > 0.3
        smask = torch.rand((N, S), generator=g) > 0.3
        xmask[:, 0] = smask[:, 0] = True
        xm, sm = xmask.double(), smask.double()
    xs = _to_split(x, pad_cols=C)                    # x lives in the first half of a [., 2C] buffer like in the product
    x64 = xs.float().double().cpu()                  # the kernel sees the 22-bit split values: the reference starts from those
    # source side on the device like the product: k | v projection (split-plane GEMM), then the state
    pkv = ops.PackedDense(torch.cat([sd["l.k_proj.weight"], sd["l.v_proj.weight"]], 0).to(DEV))
    kv = ops.linear(_to_split(src), pkv).view(N, S, 2 * C)
    state = ops.encoder256_state(kv[..., :C], kv[..., C:], smask.to(DEV) if masked else None, 1)
    k64, v64 = kv[..., :C].double().cpu(), kv[..., C:].double().cpu()
    ref = _stages64(sd, x64, k64, v64, xm, sm)
    out_s = ops.SplitAct.empty_rows((N, L), C, DEV)
    out32 = torch.empty((N, L, C), dtype=torch.float32, device=DEV)
    worst = {}
    for stage in (1, 2, 3, 4):
        dbg = ops.encoder256_apply(xs, fw, state, S, xmask.to(DEV) if masked else None, 1, out_split=out_s, out=out32,
                                   debug_stage=stage)
        r = ref[stage].reshape(-1, C)
        worst[stage] = float((dbg.double().cpu() - r).abs().max() / r.abs().max())
    out = out32.double().cpu()
    worst[5] = float((out - ref[5]).abs().max() / ref[5].abs().max())
    print(f"[fused256 case {case}: N={N} L={L} S={S}] relative errors q/msg/norm1/mlp/out: "
          + " ".join(f"{worst[k]:.1e}" for k in (1, 2, 3, 4, 5)))
    assert all(worst[k] < 2e-5 for k in worst), worst
    assert torch.equal(out_s.float(), out32)                               # the fp32 form is the exact value of the planes
    if masked:                                                             # masked queries: message 0 -> the layer still runs
        assert torch.isfinite(out32).all()
    # deterministic, state included
    state_b = ops.encoder256_state(kv[..., :C], kv[..., C:], smask.to(DEV) if masked else None, 1)
    assert torch.equal(state, state_b)
    out_b = ops.SplitAct.empty_rows((N, L), C, DEV)
    ops.encoder256_apply(xs, fw, state_b, S, xmask.to(DEV) if masked else None, 1, out_split=out_b)
    assert torch.equal(out_b.hi, out_s.hi) and torch.equal(out_b.lo, out_s.lo)


def _decode_image(img, N):
    """apply image [N, 33 KB] -> (KV [N, H, d, v], Ksum [N, 256]) as float64"""
    fr = img[:, :32768].contiguous().view(torch.float16).view(N, H, 2, 2, 64, 8).double()      # [n, h, rb, plane, lane, slot]
    val = fr[:, :, :, 0] + fr[:, :, :, 1] / 2048.0
    dec = torch.empty((N, H, D, D), dtype=torch.float64)
    for lane in range(64):
        i, grp = lane & 15, lane >> 4
        for j in range(8):
            d = 16 * (j >> 2) + 4 * grp + (j & 3)
            for rb in range(2):
                dec[:, :, d, 16 * rb + i] = val[:, :, rb, lane, j]
    return dec, img[:, 32768:].contiguous().view(torch.float32).view(N, C).double()


@pytest.mark.parametrize("N,S,masked", [(2, 4800, False), (3, 333, True), (1, 26600, True), (5, 40, False), (16, 4800, False)])
def test_fused_kv_vs_fp64_and_vs_state(built_lib, N, S, masked):
    """dfsfm_encoder256_kv_f32 (k | v projection fused with the partial KV sums) against float64, and against the unfused
    source side (projection GEMM + dfsfm_encoder256_state_f32) it replaces; ragged lengths, masks, many chunks, determinism."""
    sd = _weights(40 + N)
    fw = _fused(sd)
    g = torch.Generator().manual_seed(300 + S)
    src = torch.randn((N, S, C), generator=g)
    smask = None
    if masked:
        smask = torch.rand((N, S), generator=g) > 0.3
        smask[:, 0] = True
    ss = _to_split(src)
    s64 = ss.float().double().cpu()
    img = ops.encoder256_kv(ss, fw, smask.to(DEV) if masked else None, 1)
    KVd, ksd = _decode_image(img.cpu(), N)
    w64 = {n: w.double() for n, w in sd.items()}
    K = restate.elu1(s64 @ w64["l.k_proj.weight"].T).view(N, S, H, D)
    V = (s64 @ w64["l.v_proj.weight"].T).view(N, S, H, D)
    if masked:
        K = K * smask.double()[:, :, None, None]
        V = V * smask.double()[:, :, None, None]
    KV = torch.einsum("nshd,nshv->nhdv", K, V / S)
    e_kv = ((KVd - KV).abs().max() / KV.abs().max()).item()
    e_ks = ((ksd - K.sum(1).reshape(N, C)).abs().max() / K.sum(1).abs().max()).item()
    print(f"[fused kv N={N} S={S}] relative errors KV {e_kv:.1e}, Ksum {e_ks:.1e}")
    assert e_kv < 2e-6 and e_ks < 2e-6
    pkv = ops.PackedDense(torch.cat([sd["l.k_proj.weight"], sd["l.v_proj.weight"]], 0).to(DEV))
    kv = ops.linear(ss, pkv).view(N, S, 2 * C)
    KVu, ksu = _decode_image(ops.encoder256_state(kv[..., :C], kv[..., C:], smask.to(DEV) if masked else None, 1).cpu(), N)
    assert ((KVd - KVu).abs().max() / KV.abs().max()).item() < 2e-6 and ((ksd - ksu).abs().max() / ksu.abs().max()).item() < 2e-6
    assert torch.equal(img, ops.encoder256_kv(ss, fw, smask.to(DEV) if masked else None, 1))      # fixed summation order


def test_state_image_vs_fp64(built_lib):
    """dfsfm_encoder256_state_f32 alone: the image's fragments hold KV_h^T = (sum phi(k)^T v / S)^T in the kernel's k order, the
    tail Ksum -- decoded on the host and compared with float64 sums (chunked partial sums: S = 4800 runs as 34 chunks)."""
    g = torch.Generator().manual_seed(9)
    N, S = 3, 4800
    k = torch.randn((N, S, C), generator=g)
    v = torch.randn((N, S, C), generator=g)
    img = ops.encoder256_state(k.to(DEV), v.to(DEV)).cpu()
    fr = img[:, :32768].contiguous().view(torch.float16).view(N, H, 2, 2, 64, 8).double()      # [n, h, rb, plane, lane, slot]
    val = fr[:, :, :, 0] + fr[:, :, :, 1] / 2048.0                                               # [n, h, rb, lane, slot]
    ksum = img[:, 32768:].contiguous().view(torch.float32).view(N, C).double()
    K = restate.elu1(k.double()).view(N, S, H, D)
    KV = torch.einsum("nshd,nshv->nhdv", K, v.double().view(N, S, H, D) / S)
    dec = torch.empty_like(KV)
    for lane in range(64):
        i, grp = lane & 15, lane >> 4
        for j in range(8):
            d = 16 * (j >> 2) + 4 * grp + (j & 3)
            for rb in range(2):
                dec[:, :, d, 16 * rb + i] = val[:, :, rb, lane, j]
    assert ((dec - KV).abs().max() / KV.abs().max()).item() < 2e-6
    assert ((ksum - K.sum(1).reshape(N, C)).abs().max() / K.sum(1).abs().max()).item() < 2e-6


def test_fused256_equals_unfused_path_and_batch_independence(built_lib):
    """encoder_layer_split: the fused d_model-256 path and the five-GEMM + K1 path agree to fp32 noise (self and cross, with
    masks), and a sequence's result does not depend on what else is in the batch."""
    from detectorfreesfm_amd import coarse
    sd = _weights(3)
    get = lambda name: sd["l." + name].to(DEV)
    w = coarse.EncoderLayerWeights(get, "")
    assert w.fused256 is not None and w.fused is None
    g = torch.Generator().manual_seed(4)
    N, L, S = 4, 1200, 1000
    x = torch.randn((N, L, C), generator=g)
    y = torch.randn((N, S, C), generator=g)
    xm = (torch.rand((N, L), generator=g) > 0.2).to(DEV)
    ym = (torch.rand((N, S), generator=g) > 0.2).to(DEV)
    xs = _to_split(x, pad_cols=C)
    full = ops.SplitAct(xs.hi.as_strided((N, L, 2 * C), xs.hi.stride()), xs.lo.as_strided((N, L, 2 * C), xs.lo.stride()), 2 * C)
    ys = _to_split(y)

    def run(fused, src, is_self, masks, fused_kv=True):
        keep, w.fused256 = w.fused256, (w.fused256 if fused else None)
        keep_kv, coarse.FUSED_KV256 = coarse.FUSED_KV256, fused_kv
        out = torch.empty((N, L, C), device=DEV)
        m = (xm, xm if is_self else ym) if masks else (None, None)
        coarse.encoder_layer_split(w, full, src, out, None, H, m[0], m[1], is_self=is_self)
        w.fused256, coarse.FUSED_KV256 = keep, keep_kv
        return out
    for is_self, masks in ((True, False), (False, False), (False, True), (True, True)):
        src = full.cols(0, C) if is_self else ys
        a, b, c = run(True, src, is_self, masks), run(False, src, is_self, masks), run(True, src, is_self, masks, fused_kv=False)
        assert ((a - b).abs().max() / b.abs().max()).item() < 2e-5, (is_self, masks)
        assert ((a - c).abs().max() / b.abs().max()).item() < 2e-5, (is_self, masks)
    out_f = run(True, full.cols(0, C), True, False)
    sub = ops.SplitAct(full.hi[1:3], full.lo[1:3], 2 * C)
    out_s = torch.empty((2, L, C), device=DEV)
    coarse.encoder_layer_split(w, sub, sub.cols(0, C), out_s, None, H, is_self=True)
    assert torch.equal(out_s, out_f[1:3])


def test_fused256_argument_checks(built_lib):
    from detectorfreesfm_amd._lib import DfsfmError
    sd = _weights(1)
    fw = _fused(sd)
    xs = _to_split(torch.zeros((2, 8, C)))
    k = torch.zeros((2, 8, C), device=DEV)
    state = ops.encoder256_state(k, k)
    with pytest.raises(DfsfmError):                  # 8-token sequences: a wave's 16 tokens could touch three of them
        ops.encoder256_apply(xs, fw, state, 8, out_split=ops.SplitAct.empty_rows((2, 8), C, DEV))
    with pytest.raises(DfsfmError):
        ops.encoder256_apply(xs, fw, state[:1], 8, out_split=ops.SplitAct.empty_rows((2, 8), C, DEV))
    with pytest.raises(DfsfmError):
        ops.Encoder256Weights(torch.zeros(128, 128), torch.zeros(128, 128), torch.zeros(256, 256), torch.zeros(128, 256),
                              (torch.ones(128), torch.zeros(128)), (torch.ones(128), torch.zeros(128)))


def test_fused256_kernels_bit_reproducible_at_full_occupancy(built_lib):
    """encoder256.hip keeps the SLP vectoriser's packed-fp32 instructions (csrc/Makefile bans them only from MFMA kernels whose waves
    can share a SIMD).  r05's reproducer (tools/ubench/pk_vs_mfma.hip) found NO wrong packed result beside a co-resident MFMA wave,
    so the ban is a workaround for a fine_match-specific defect, not a hardware rule -- and the property that defect violated is
    what is asserted, here for the kernels that keep packed math: the cross-layer launch of the bench step (8 x 4800 tokens, every
    CU busy for three rounds) gives the same bits run after run."""
    N, L, S = 8, 4800, 4800
    sd = _weights(3)
    fw = _fused(sd)
    g = torch.Generator().manual_seed(17)
    xs, ss = _to_split(torch.randn((N, L, C), generator=g)), _to_split(torch.randn((N, S, C), generator=g))

    def run(x, s):
        n = x.hi.shape[0]
        out = ops.SplitAct.empty_rows((n, L), C, DEV)
        state = ops.encoder256_kv(s, fw, None, 1)
        ops.encoder256_apply(x, fw, state, S, None, 1, out_split=out)
        return out.hi.clone(), out.lo.clone(), state.clone() if isinstance(state, torch.Tensor) else None
    first = run(xs, ss)
    for _ in range(6):
        again = run(xs, ss)
        assert torch.equal(again[0], first[0]) and torch.equal(again[1], first[1])
        if first[2] is not None:
            assert torch.equal(again[2], first[2])
    # (Batch composition is NOT bit-invariant for this kernel by design: enc256_kv splits a sequence into as many chunks as fill the
    # chip for the launch's N, so the fixed-order sum of the KV partials has another association for N = 1 than for N = 8 -- same
    # values to rounding, tests/test_gpu_e2e.py::test_loftr_batch_equals_singles.  The d_model-128 kernel sums per track in one
    # workgroup and IS batch-invariant: tests/test_gpu_encoder_fused.py.)
    one = run(xs[5:6], ss[5:6])
    err = (one[0][0].float() + one[1][0].float() / 2048.0 - (first[0][5].float() + first[1][5].float() / 2048.0)).abs().max().item()
    assert err < 2e-5, err
