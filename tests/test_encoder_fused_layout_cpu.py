"""Host side of the fused encoder layer (csrc/encoder_fused.hip) without a GPU: a LANE-LEVEL model of the two kernels'
data flow -- MFMA 32x32x16 operand / accumulator layouts, the fragment streams ``ops.EncoderFusedWeights`` builds, the
accumulator -> operand chaining rule and the "apply image" hand-over between the kernels -- evaluated in float64 on the
exact fragments the kernels would read, against oracle.restate.encoder_layer.  It pins the packing order and every index
permutation; the arithmetic of the real kernels is checked on the GPU (tests/test_gpu_encoder_fused.py)."""
import numpy as np
import torch

from detectorfreesfm_amd import ops
from oracle import restate

C, H = 128, 8


def dch(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def a_matrix(frag):
    """[64 lanes, 8] fragment as the 32 x 16 A operand: lane (i, kg) slot j -> A[i][8 kg + j]."""
    f = np.asarray(frag, dtype=np.float64).reshape(2, 32, 8)              # [kg, i, j]
    return np.concatenate([f[0], f[1]], axis=1)                            # [32, 16]


def b_matrix(frag):
    """[64 lanes, 8] fragment as the 16 x 32 B operand: lane (j, kg) slot jj -> B[8 kg + jj][j]."""
    return a_matrix(frag).T


def d_lanes(D):
    """32 x 32 result -> accumulator registers [64 lanes, 16]: lane (col, h) reg r = D[dch(r, h)][col]."""
    out = np.empty((64, 16))
    for h in range(2):
        for r in range(16):
            out[32 * h:32 * h + 32, r] = D[dch(r, h), :]
    return out


def frags_from_acc(v):
    """to_frags: accumulator block [64, 16] -> the two operand fragments [64, 8] (registers 0-7 and 8-15)."""
    return v[:, :8], v[:, 8:]


def stream_frag(stream, slab, idx):
    """value (hi + lo / 2048) of fragment pair ``idx`` of slab ``slab``: [64, 8] float64."""
    f = stream[slab * 16 + 2 * idx].double() + stream[slab * 16 + 2 * idx + 1].double() / 2048.0
    return f.numpy()


def x_frag_d_order(x_tile, s):
    """What enc_apply_kernel's xfrags() reads for k-step s: lane (tok, h) slots = channels 16 s + 4 h + {0..3}, 16 s + 8 + 4 h + {0..3}."""
    f = np.empty((64, 8))
    for h in range(2):
        cols = [16 * s + 4 * h + e for e in range(4)] + [16 * s + 8 + 4 * h + e for e in range(4)]
        f[32 * h:32 * h + 32] = x_tile[:, cols]
    return f


def x_frag_natural(x_tile, s):
    """enc_kv_kernel's A fragment: lane (tok, h) slots = channels 16 s + 8 h + j."""
    f = np.empty((64, 8))
    for h in range(2):
        f[32 * h:32 * h + 32] = x_tile[:, 16 * s + 8 * h:16 * s + 8 * h + 8]
    return f


def rows_from_d(v_blocks):
    """accumulator blocks [[64, 16]] * 4 of the TRANSPOSED orientation (lane = token) -> [32 tokens, 128 channels]."""
    out = np.empty((32, C))
    for b, v in enumerate(v_blocks):
        for h in range(2):
            for r in range(16):
                out[:, 32 * b + dch(r, h)] = v[32 * h:32 * h + 32, r]
    return out


def model_kv(fw, src):
    """enc_kv_kernel for one sequence [S, 128] -> (16 apply-image fragments [64, 8], Ksum [128])."""
    S = src.shape[0]
    kv = [np.zeros((32, 32)) for _ in range(4)]
    ksum = np.zeros(C)
    for s0 in range(0, S, 32):
        tile = np.zeros((32, C))
        n = min(32, S - s0)
        tile[:n] = src[s0:s0 + n]
        tm = np.zeros(32)
        tm[:n] = 1.0
        for p in range(4):
            d = [np.zeros((32, 32)), np.zeros((32, 32))]
            for u in range(2):
                for ks in range(4):
                    for b in range(2):
                        d[b] += a_matrix(x_frag_natural(tile, 4 * u + ks)) @ b_matrix(stream_frag(fw.kv_stream, 2 * p + u, ks * 2 + b))
            kl, vl = d_lanes(d[0]), d_lanes(d[1])                       # lane = channel, registers = tokens dch(r, h)
            tok_of = np.array([[dch(r, h) for r in range(16)] for h in range(2)])     # [h, r]
            m = np.concatenate([np.tile(tm[tok_of[0]], (32, 1)), np.tile(tm[tok_of[1]], (32, 1))], 0)   # [64, 16]
            kf = (np.where(kl > 0, kl, np.expm1(kl)) + 1.0) * m
            vf = vl * m / S
            ks_lane = kf.sum(1)
            ksum[32 * p:32 * p + 32] += ks_lane[:32] + ks_lane[32:]
            (k0, k1), (v0, v1) = frags_from_acc(kf), frags_from_acc(vf)
            kv[p] += a_matrix(k0) @ b_matrix(v0) + a_matrix(k1) @ b_matrix(v1)
    frags = []
    for p in range(4):
        lanes = d_lanes(kv[p])                                            # lane = v channel, registers = k channel
        head0 = (np.arange(64) & 31) < 16
        f0, f1 = frags_from_acc(lanes)
        frags += [np.where(head0[:, None], f0, 0.0), np.where(head0[:, None], 0.0, f1)]
    return frags, ksum


def model_apply(fw, x, frags_kv, ksum, S):
    """enc_apply_kernel for one 32-token tile of one sequence."""
    def gemm4(first_slab, bfrag):                   # 4 slabs x (2 k-steps x 4 blocks): k-steps 0..7
        acc = [np.zeros((32, 32)) for _ in range(4)]
        for s in range(4):
            for ks in range(2):
                for b in range(4):
                    acc[b] += a_matrix(stream_frag(fw.apply_stream, first_slab + s, ks * 4 + b)) @ b_matrix(bfrag(2 * s + ks))
        return [d_lanes(a) for a in acc]
    q = gemm4(0, lambda s: x_frag_d_order(x, s))
    q_rows = rows_from_d(q)
    phi = [np.where(v > 0, v, np.expm1(v)) + 1.0 for v in q]
    a_fr = {}
    Z = {}
    for b in range(4):
        ks_l = np.empty((64, 16))
        for h in range(2):
            for r in range(16):
                ks_l[32 * h:32 * h + 32, r] = ksum[32 * b + dch(r, h)]
        z0 = (phi[b][:, :8] * ks_l[:, :8]).sum(1)
        z1 = (phi[b][:, 8:] * ks_l[:, 8:]).sum(1)
        z0 = z0 + np.roll(z0, 32)
        z1 = z1 + np.roll(z1, 32)
        Z[2 * b], Z[2 * b + 1] = 1.0 / (z0 + 1e-6), 1.0 / (z1 + 1e-6)
        a_fr[2 * b], a_fr[2 * b + 1] = frags_from_acc(phi[b])
    msg = []
    for b in range(4):
        acc = np.zeros((32, 32))
        for t in range(2):
            acc += a_matrix(frags_kv[2 * b + t]) @ b_matrix(a_fr[2 * b + t])
        v = d_lanes(acc)
        zz = np.concatenate([np.tile(Z[2 * b][:, None], (1, 8)), np.tile(Z[2 * b + 1][:, None], (1, 8))], 1)
        msg.append(v * zz * S)
    msg_rows = rows_from_d(msg)
    m_fr = {}
    for b in range(4):
        m_fr[2 * b], m_fr[2 * b + 1] = frags_from_acc(msg[b])
    merged = rows_from_d(gemm4(4, lambda s: m_fr[s]))
    m1 = torch.nn.functional.layer_norm(torch.from_numpy(merged), (C,), fw.n1[0].double(), fw.n1[1].double()).numpy()
    # back to accumulator layout to exercise the chaining into mlp.0
    m1_blocks = []
    for b in range(4):
        v = np.empty((64, 16))
        for h in range(2):
            for r in range(16):
                v[32 * h:32 * h + 32, r] = m1[:, 32 * b + dch(r, h)]
        m1_blocks.append(v)
    n_fr = {}
    for b in range(4):
        n_fr[2 * b], n_fr[2 * b + 1] = frags_from_acc(m1_blocks[b])
    o = [np.zeros((32, 32)) for _ in range(4)]
    slab = 8
    for hc in range(4):
        hacc = [np.zeros((32, 32)), np.zeros((32, 32))]
        for u in range(4):
            for ks in range(4):
                kstep = 4 * u + ks
                bf = x_frag_d_order(x, kstep) if kstep < 8 else n_fr[kstep - 8]
                for b in range(2):
                    hacc[b] += a_matrix(stream_frag(fw.apply_stream, slab + u, ks * 2 + b)) @ b_matrix(bf)
        slab += 4
        h_fr = {}
        for b in range(2):
            h_fr[2 * b], h_fr[2 * b + 1] = frags_from_acc(np.maximum(d_lanes(hacc[b]), 0.0))
        for v2 in range(2):
            for ks in range(2):
                for b in range(4):
                    o[b] += a_matrix(stream_frag(fw.apply_stream, slab + v2, ks * 4 + b)) @ b_matrix(h_fr[2 * v2 + ks])
        slab += 2
    assert slab == 32
    o_rows = rows_from_d([d_lanes(a) for a in o])
    out = x + torch.nn.functional.layer_norm(torch.from_numpy(o_rows), (C,), fw.n2[0].double(), fw.n2[1].double()).numpy()
    return q_rows, msg_rows, m1, o_rows, out


def test_fragment_streams_and_chaining_reproduce_the_layer():
    g = torch.Generator().manual_seed(5)
    sd = {}
    for name, shape in (("q_proj", (C, C)), ("k_proj", (C, C)), ("v_proj", (C, C)), ("merge", (C, C)),
                        ("mlp.0", (2 * C, 2 * C)), ("mlp.2", (C, 2 * C))):
        # values exactly representable in the split form, so the float64 model has nothing to round
        w = (torch.randn(shape, generator=g) * 0.08).half().float()
        sd[f"l.{name}.weight"] = w
    for nm in ("norm1", "norm2"):
        sd[f"l.{nm}.weight"] = 1.0 + 0.1 * torch.randn(C, generator=g)
        sd[f"l.{nm}.bias"] = 0.1 * torch.randn(C, generator=g)
    fw = ops.EncoderFusedWeights(sd["l.q_proj.weight"], sd["l.k_proj.weight"], sd["l.v_proj.weight"], sd["l.merge.weight"],
                                 sd["l.mlp.0.weight"], sd["l.mlp.2.weight"], (sd["l.norm1.weight"], sd["l.norm1.bias"]),
                                 (sd["l.norm2.weight"], sd["l.norm2.bias"]))
    assert fw.kv_stream.shape == (128, 64, 8) and fw.apply_stream.shape == (512, 64, 8)
    S, L = 45, 32                                                        # a ragged source (two blocks, 13 tokens in the second)
    src = torch.randn((1, S, C), generator=g).half().double()
    x = torch.randn((1, L, C), generator=g).half().double()
    frags_kv, ksum = model_kv(fw, src[0].numpy())
    q, msg, m1, o, out = model_apply(fw, x[0].numpy(), frags_kv, ksum, S)
    sd64 = {k: v.double() for k, v in sd.items()}
    ref = restate.encoder_layer(sd64, "l.", x, src, H)[0].numpy()
    # intermediate stages of the reference, for a sharper diagnosis than the final output alone
    qr = (x[0] @ sd64["l.q_proj.weight"].T).numpy()
    assert np.abs(q - qr).max() < 1e-10
    kk = restate.elu1((src[0] @ sd64["l.k_proj.weight"].T).view(S, H, 16))
    vv = (src[0] @ sd64["l.v_proj.weight"].T).view(S, H, 16)
    assert np.abs(ksum - kk.sum(0).reshape(-1).numpy()).max() < 1e-10
    m_ref = restate.linear_attention(torch.from_numpy(qr).view(1, L, H, 16), (src[0] @ sd64["l.k_proj.weight"].T).view(1, S, H, 16),
                                     vv.view(1, S, H, 16)).reshape(L, C).numpy()
    assert np.abs(msg - m_ref).max() < 1e-9
    assert np.abs(out - ref).max() < 1e-9


def test_kslot_orders():
    nat, dord = ops._kslots(3, False), ops._kslots(3, True)
    assert nat.tolist() == [[48 + j for j in range(8)], [56 + j for j in range(8)]]
    assert dord[0].tolist() == [48, 49, 50, 51, 56, 57, 58, 59] and dord[1].tolist() == [52, 53, 54, 55, 60, 61, 62, 63]
    assert sorted(dord.reshape(-1).tolist()) == list(range(48, 64))
