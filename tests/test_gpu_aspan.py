"""ASpanFormer coarse matcher on the GPU: every new kernel of csrc/aspan_ops.hip through the C ABI against
the reference's own operations (torch CPU), then HipASpanFormer end to end against the fixture written by the real module and
against the oracle, with the per-entry parity rules of tests/parity.py."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import parity
from cpu_standins import _span_attention as span_reference
from detectorfreesfm_amd import ops, plugin, synth
from oracle import restate_aspanformer as ra
from oracle.make_golden import aspanformer_cases, aspanformer_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(npz):
    return {k: npz[k].item() for k in npz.files if npz[k].ndim == 0}


@pytest.mark.parametrize("N,H,W,C,k", [(1, 60, 80, 768, 4), (2, 24, 32, 256, 2), (1, 12, 16, 128, 4)])
def test_avgpool_vs_torch(built_lib, N, H, W, C, k):
    g = torch.Generator().manual_seed(H + k)
    wide = torch.randn((N, H, W, C + 64), generator=g)
    x = wide[..., 32:32 + C]                                         # a column slice of a wider token buffer
    ref = F.avg_pool2d(x.permute(0, 3, 1, 2).contiguous(), k, stride=k).permute(0, 2, 3, 1)
    out = ops.avgpool(wide.to(DEV)[..., 32:32 + C], k).cpu()
    assert torch.equal(out, ref)                                     # same summation order, exact division


@pytest.mark.parametrize("L,S,swap", [(300, 300, False), (300, 117, False), (70, 676, True)])
def test_full_attention_vs_reference(built_lib, L, S, swap):
    g = torch.Generator().manual_seed(L + S)
    N = 2 if swap else 1
    q = torch.randn((N, L, 768), generator=g)[..., :256]
    kv = torch.randn((N, S, 768), generator=g)
    k, v = kv[..., 256:512], kv[..., 512:]
    scale = 1.3 / math.sqrt(32)
    outs = []
    for n in range(N):
        o = n ^ 1 if swap else n
        outs.append(ra.full_attention(q[n:n + 1].double().transpose(1, 2).contiguous(), k[o:o + 1].double().transpose(1, 2).contiguous(),
                                      v[o:o + 1].double().transpose(1, 2).contiguous(), 8, temp=1.3).transpose(1, 2))
    ref = torch.cat(outs, 0)
    qd, kvd = q.to(DEV), kv.to(DEV)
    out = ops.full_attention(qd, kvd[..., 256:512], kvd[..., 512:], 8, scale, kv_swap=swap).cpu().double()
    assert (out - ref).abs().max().item() < 2e-6 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("hw,hw_k,s", [((12, 16), (12, 16), 1), ((6, 8), (6, 8), 2), ((30, 40), (24, 52), 2), ((60, 80), (60, 80), 1)])
def test_span_attention_vs_reference(built_lib, hw, hw_k, s):
    """Random maps and flows whose spans reach far outside the key map (zero padding) and whose centres sit on the border."""
    g = torch.Generator().manual_seed(hw[0] * 7 + s)
    h, w = hw
    hk, wk = hw_k
    H0, W0 = h * s, w * s
    q = torch.randn((h * w, 768), generator=g)[:, :256]
    kv = torch.randn((hk * wk, 768), generator=g)
    flow = torch.cat([torch.rand((H0 * W0, 1), generator=g) * (wk * s + 6) - 3, torch.rand((H0 * W0, 1), generator=g) * (hk * s + 6) - 3,
                      torch.randn((H0 * W0, 2), generator=g) * 1.5 - 1.0], 1).contiguous()
    so = torch.tensor([[a - 3.5, b - 3.5] for a in range(8) for b in range(8)])
    ref = span_reference(q.double(), hw, kv[:, 256:512].double(), kv[:, 512:].double(), hw_k, flow.double(), (H0, W0), so.double(),
                         8, [2, 8], 5, 1.0)
    qd, kvd = q.to(DEV), kv.to(DEV)
    out = ops.span_attention(qd, hw, kvd[:, 256:512], kvd[:, 512:], hw_k, flow.to(DEV), (H0, W0), so.to(DEV), 8, [2, 8], 5).cpu().double()
    # a sample that lands within fp32 rounding of a cell edge may take the neighbouring cell: compare in the mean and bound the max
    err = (out - ref).abs()
    assert err.mean().item() < 1e-6 and err.max().item() < 5e-4


def test_span_attention_unaligned_rows_take_the_4_byte_gathers(built_lib):
    """Key / value rows whose row pitch is not a multiple of 16 bytes cannot use the 16-byte gathers: the entry point then runs
    span_attention_kernel<false> (4-byte taps).  Same products and sums per element -> the result must equal the aligned launch
    bit for bit."""
    g = torch.Generator().manual_seed(21)
    hw, s = (12, 16), 2
    H0, W0 = hw[0] * s, hw[1] * s
    q = torch.randn((hw[0] * hw[1], 256), generator=g).to(DEV)
    kv = torch.randn((hw[0] * hw[1], 2 * 256 + 2), generator=g).to(DEV)          # pitch 514 floats: rows are 8-byte aligned only
    flow = torch.cat([torch.rand((H0 * W0, 1), generator=g) * 36 - 2, torch.rand((H0 * W0, 1), generator=g) * 28 - 2,
                      torch.randn((H0 * W0, 2), generator=g) - 1.0], 1).contiguous().to(DEV)
    so = torch.tensor([[a - 3.5, b - 3.5] for a in range(8) for b in range(8)]).to(DEV)
    k_u, v_u = kv[:, :256], kv[:, 258:514]
    assert k_u.stride(0) % 4 != 0
    narrow = ops.span_attention(q, hw, k_u, v_u, hw, flow, (H0, W0), so, 8, [2, 8], 5)
    wide = ops.span_attention(q, hw, k_u.contiguous(), v_u.contiguous(), hw, flow, (H0, W0), so, 8, [2, 8], 5)
    assert torch.equal(narrow, wide)


def test_span_attention_batched_swap(built_lib):
    """Two row-stacked images in one launch, image n sampling the keys / values of image n ^ 1 (how a pair of equal frames runs)."""
    g = torch.Generator().manual_seed(11)
    hw, s = (12, 16), 2
    H0, W0 = hw[0] * s, hw[1] * s
    q = torch.randn((2, hw[0] * hw[1], 768), generator=g)
    flow = torch.cat([torch.rand((2, H0 * W0, 1), generator=g) * 36 - 2, torch.rand((2, H0 * W0, 1), generator=g) * 28 - 2,
                      torch.randn((2, H0 * W0, 2), generator=g) - 1.0], 2).contiguous()
    so = torch.tensor([[a - 3.5, b - 3.5] for a in range(8) for b in range(8)])
    qd = q.to(DEV)
    out = ops.span_attention(qd[..., :256], hw, qd[..., 256:512], qd[..., 512:], hw, flow.to(DEV), (H0, W0), so.to(DEV), 8, [2, 8], 5,
                             kv_swap=True).cpu().double()
    for n in (0, 1):
        ref = span_reference(q[n, :, :256].double(), hw, q[1 - n, :, 256:512].double(), q[1 - n, :, 512:].double(), hw, flow[n].double(),
                             (H0, W0), so.double(), 8, [2, 8], 5, 1.0)
        err = (out[n] - ref).abs()
        assert err.mean().item() < 1e-6 and err.max().item() < 5e-4


@pytest.mark.parametrize("C", [256, 384])
def test_layernorm2d_vs_reference(built_lib, C):
    g = torch.Generator().manual_seed(C)
    rows = 1003
    x = torch.randn((rows, C + 32), generator=g)[:, :C] * 3 + 0.5
    aff, b = torch.randn((C,), generator=g), torch.randn((C,), generator=g)
    res = ops.SplitAct.empty_rows((rows,), 640, DEV)
    r32 = torch.randn((rows, C), generator=g)
    ops.split_rows(r32.to(DEV), out_split=res.cols(0, C))
    xd = x.double()
    ref = aff.double() * (xd - xd.mean(1, keepdim=True)) / (xd.std(1, keepdim=True) + 1e-6) + b.double()
    xg = torch.randn((rows, C + 32)).to(DEV)
    xg[:, :C] = x.to(DEV)
    out = ops.layernorm2d(xg[:, :C], aff.to(DEV), b.to(DEV)).cpu().double()
    assert (out - ref).abs().max().item() < 3e-6 * ref.abs().max().item()
    dst = ops.SplitAct.empty_rows((rows,), 640, DEV)
    ops.layernorm2d(xg[:, :C], aff.to(DEV), b.to(DEV), residual=res.cols(0, C), out_split=dst.cols(128, 128 + C), want_f32=False)
    got = dst.cols(128, 128 + C).float().cpu().double()
    want = res.cols(0, C).float().cpu().double() + ref
    assert (got - want).abs().max().item() < 3e-6 * want.abs().max().item()


@pytest.mark.parametrize("bilinear,scale", [(True, 4), (False, 4), (False, 2), (False, 1)])
def test_upsample_vs_torch(built_lib, bilinear, scale):
    g = torch.Generator().manual_seed(scale)
    x = torch.randn((1, 15, 20, 384), generator=g)
    xs = x[..., 128:384]
    xc = xs.permute(0, 3, 1, 2).contiguous()
    ref = (F.interpolate(xc, scale_factor=scale, mode="bilinear") if bilinear else F.interpolate(xc, scale_factor=scale, mode="nearest"))
    ref = ref.permute(0, 2, 3, 1).reshape(-1, 256)
    xd = x.to(DEV)[..., 128:384]
    out = ops.upsample(xd, scale, bilinear).cpu().reshape(-1, 256)
    assert (out - ref).abs().max().item() < (2e-6 if bilinear else 0.0) + 1e-30
    dst = ops.SplitAct.empty_rows((ref.shape[0],), 768, DEV)
    ops.upsample(xd, scale, bilinear, out_split=dst.cols(256, 512), want_f32=False)
    assert (dst.cols(256, 512).float().cpu() - ref).abs().max().item() < 4e-6


@pytest.mark.parametrize("hw,out", [((100, 140), (96, 128)), ((900, 1200), (896, 1184)), ((37, 53), (32, 32)), ((64, 64), (64, 64))])
def test_resize_bilinear_vs_torch(built_lib, hw, out):
    """The online resize of ASpanFormer.resize_df: F.interpolate(size, bilinear, align_corners=False) on [1,1,H,W] frames."""
    g = torch.Generator().manual_seed(hw[0])
    x = torch.rand((1, 1, *hw), generator=g)
    ref = F.interpolate(x, size=list(out), mode="bilinear", align_corners=False)
    got = ops.resize_bilinear(x.to(DEV), *out).cpu()
    assert got.shape == ref.shape and (got - ref).abs().max().item() < 2e-6
    ref64 = F.interpolate(x.double(), size=list(out), mode="bilinear", align_corners=False)
    assert (got.double() - ref64).abs().max().item() < 3e-4          # fp32 source coordinates (ulp 1.2e-4 at x = 1200) on both fp32 sides


def test_flow_decode_vs_reference(built_lib):
    g = torch.Generator().manual_seed(0)
    x = torch.randn((4800, 64), generator=g) * 3
    ref = torch.cat([torch.sigmoid(x[:, :2].double()) * torch.tensor([80.0, 60.0], dtype=torch.float64), x[:, 2:4].double()], 1)
    out = ops.flow_decode(x.to(DEV), 80, 60).cpu().double()
    assert (out - ref).abs().max().item() < 1e-5


# ---------------------------------------------------------------- end to end
def _aspan(thr=0.2):
    from detectorfreesfm_amd.aspanformer import HipASpanFormer, aspanformer_coarse_only_config
    from detectorfreesfm_amd.params import aspanformer_param_spec, planted_aspanformer_state_dict
    cfg = aspanformer_coarse_only_config(thr)
    sd = planted_aspanformer_state_dict(aspanformer_param_spec(cfg), 0)
    m = HipASpanFormer(cfg)
    m.load_state_dict(sd, strict=True)
    return cfg, sd, m.eval().to(DEV)


def _strict(d, ref, conf, thr, what, max_exempt):
    ex = parity.check_coarse(d, ref, conf, thr)
    parity.check_coarse_rows(d, ref, ex)
    print(f"[{what}] {len(ref['i_ids'])} reference matches, {len(d['i_ids'])} on the GPU, exempted entries: {ex}")
    assert len(ex) <= max_exempt
    return ex


@pytest.mark.parametrize("case", [0, 1, 2])
def test_aspanformer_e2e_golden(built_lib, golden, case):
    """Fixture written by the real ASpanFormer module: equal frames, two frame sizes (cross-size span attention), and a
    100x140 frame that the online resize turns into 96x128 first."""
    gz = golden("aspanformer_e2e")
    c = _case(gz)
    tag, hw0, hw1 = aspanformer_cases()[case]
    cfg, sd, m = _aspan(c["thr"])
    data = aspanformer_inputs(c, hw0, hw1)
    d = synth.to_device(data, DEV)
    m(d)
    ref = {k: gz[f"{tag}_{k}"] for k in ("b_ids", "i_ids", "j_ids", "mconf", "mkpts0_f", "mkpts1_f")}
    _strict(d, ref, gz[f"{tag}_conf_matrix"], c["thr"], f"aspanformer_e2e {tag}", 1)
    assert len(ref["i_ids"]) > 10 and (d["m_bids"] == d["b_ids"]).all()
    pf = d["predict_flow"]
    for i in (0, 1):                                                  # the predicted flows: sigmoid * size, 12 x 16 cells
        assert np.abs(pf[i].cpu().numpy() - gz[f"{tag}_flow{i}"]).max() < 2e-3
    for k in ("offset_bids_left", "offset_lids_left", "offset_bids_right", "offset_lids_right"):
        assert np.array_equal(d[k].cpu().numpy(), gz[f"{tag}_{k}"]), k
    assert np.abs(d["offset_kpts1_f_left"].cpu().numpy() - gz[f"{tag}_offset_kpts1_f_left"]).max() < 2e-2   # x8 pixels


def test_aspanformer_480x640_vs_oracle(built_lib):
    """configs[1] frame size: 4800 tokens per image, levels of 300 / 1200 / 4800 tokens.

    Indices: identical to the oracle's (usual per-entry exemptions).  Confidences: this network amplifies rounding noise
    ~500x (exp of the predicted variances -> sampling spans -> bilinear samples -> softmax; unbiased-std normalisation of
    small messages), and the amplified noise is injected by EVERY layer of the ResNet backbone, not by one stage that could
    be kept exact (tools/studies/aspan_noise_study.py, log in profiles/r03_aspan_noise_study.txt: with the backbone alone
    evaluated in float64 the result is within 9.3e-5 of the fp32 oracle, with anything less it is not -- a plain fp32
    evaluation of the same network in another summation order already deviates 2.1e-4 from the oracle, 8-9 rows beyond
    1e-4, and the oracle itself sits 8.8e-5 from its own float64 evaluation).  The comparison therefore runs under the named
    rule "oracle-noise" of tests/parity.py -- the same rule the LoFTR production-size test uses: a confidence beyond 1e-4 of the
    fp32 oracle must lie within north_star's 1e-4 of the EXACT value (the float64 evaluation of the oracle) while the fp32
    oracle itself is the one that is off; such entries are listed and bounded (at most 0.3 % of the rows; measured r03: 4 of
    3545).  In addition the GPU must be as accurate against the float64 evaluation as the reference's own fp32 evaluation
    (<= 1.25 x the oracle's fp32-vs-fp64 deviation; measured 9.4e-5 vs 8.8e-5).
    The 96x128 fixtures from the real module hold the plain 1e-4 (tests above)."""
    cfg, sd, m = _aspan(0.2)
    data = synth.coarse_pair_batch(1, 480, 640, seed=7)
    d = synth.to_device(data, DEV)
    m(d)
    with torch.no_grad():
        o = ra.aspanformer_forward(sd, cfg, data, with_fine_backbone=False)
        torch.set_default_dtype(torch.float64)
        try:
            o64 = ra.aspanformer_forward({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, cfg,
                                         {k: v.double() for k, v in data.items()}, with_fine_backbone=False)
        finally:
            torch.set_default_dtype(torch.float32)
    assert o["i_ids"].numel() > 300
    c64 = o64["conf_matrix"][0].numpy()
    ex = parity.check_coarse(d, o, o["conf_matrix"], 0.2, exact=lambda b, i, j: c64[i, j])
    parity.check_coarse_rows(d, o, [e for e in ex if e[0] != "oracle-noise"])
    noise = np.abs(o["conf_matrix"][0].double().numpy() - c64).max()
    hi, hj, hc = (d[k].cpu().numpy() for k in ("i_ids", "j_ids", "mconf"))
    dev64 = np.abs(hc - c64[hi, hj])
    R = {int(i): float(c) for i, c in zip(o["i_ids"], o["mconf"])}
    dev32 = np.array([abs(float(c) - R[int(i)]) for i, c in zip(hi, hc) if int(i) in R])
    print(f"[aspanformer 480x640] {len(R)} reference matches, {len(hi)} on the GPU, exempted entries: {ex}; oracle fp32-vs-fp64 "
          f"noise {noise:.2e}; GPU vs fp64 max {dev64.max():.2e}; GPU vs fp32 oracle max {dev32.max():.2e}, "
          f"{int((dev32 > parity.TOL_CONF).sum())} rows beyond 1e-4")
    noisy = [e for e in ex if e[0] == "oracle-noise"]
    assert len(ex) - len(noisy) <= 3 and len(noisy) <= 0.003 * len(dev32) and dev64.max() <= 1.25 * noise
    assert (d["predict_flow"][0].cpu() - o["predict_flow"][0]).abs().max().item() < 2e-2


def test_aspanformer_plugin_surface(built_lib, tmp_path):
    """build_model('aspanformer_hip') from a {'state_dict': ...} checkpoint (coarse_match_worker.py:45-60) -> extract_matches."""
    from detectorfreesfm_amd.params import aspanformer_param_spec, planted_aspanformer_state_dict
    from detectorfreesfm_amd.aspanformer import aspanformer_coarse_only_config
    sd = planted_aspanformer_state_dict(aspanformer_param_spec(aspanformer_coarse_only_config(0.2)), 0)
    ckpt = tmp_path / "aspan.ckpt"
    torch.save({"state_dict": {"matcher." + k: v for k, v in sd.items()}}, ckpt)
    detector, matcher = plugin.build_model({"matcher": "aspanformer_hip", "type": "coarse_only", "match_thr": 0.2, "seed": 666,
                                            "aspanformer_hip": {"weight_path": str(ckpt)}})
    matcher.cuda()
    data = synth.coarse_pair_batch(1, 96, 128, seed=1003)
    d = {k: v.cuda() for k, v in data.items()}
    mk0, mk1, mc = plugin.extract_matches(d, detector=detector, matcher=matcher)
    with torch.no_grad():
        o = ra.aspanformer_forward(sd, matcher.config, data, with_fine_backbone=False)
    _strict(d, o, o["conf_matrix"], 0.2, "aspanformer plugin", 1)
    assert len(mc) > 10 and np.abs(mc - o["mconf"].numpy()).max() <= parity.TOL_CONF
    with pytest.raises(NotImplementedError):
        matcher({"image0": torch.zeros(1, 1, 96, 128, device=DEV), "image1": torch.zeros(1, 1, 96, 128, device=DEV),
                 "mask0": torch.ones(1, 12, 16, device=DEV), "mask1": torch.ones(1, 12, 16, device=DEV)})


def test_aspanformer_scene_cached_tokens(built_lib):
    """plugin.match_scene_cached with the ASpanFormer matcher: backbone once per image and several pairs per transformer pass
    (pairs (0,1), (0,2) travel as one stream of four images), tables equal to the one-pair-per-call forward."""
    cfg, sd, m = _aspan(0.2)
    base = synth.coarse_pair_batch(2, 100, 140, seed=1000)                  # the online resize turns these into 96 x 128
    images = torch.cat([base["image0"], base["image1"][:1]], 0)
    pairs = [(0, 1), (0, 2), (1, 2)]
    assert m.PAIRS_PER_PASS >= 2
    tables = plugin.match_scene_cached(m, images, pairs, batch=2)
    total = 0
    for (i, j) in pairs:
        d = synth.to_device({"image0": images[i:i + 1], "image1": images[j:j + 1]}, DEV)
        m(d)
        ref = torch.cat([d["mkpts0_f"], d["mkpts1_f"], d["mconf"][:, None]], -1).cpu().numpy()
        got = tables[(i, j)]
        assert got.shape == ref.shape and np.array_equal(got[:, :4], ref[:, :4]) and np.abs(got[:, 4] - ref[:, 4]).max() <= 1e-6
        total += len(ref)
    assert total > 30
