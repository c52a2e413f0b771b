"""Host side of the d_model-256 fused encoder layer (csrc/encoder256.hip) without a GPU: a LANE-LEVEL model of the kernel's
data flow -- v_mfma_f32_16x16x32_f16 operand / accumulator layouts, the 128-slab fragment stream ``ops.Encoder256Weights``
builds, the accumulator -> operand chaining rule (two 16-channel blocks = one 32-wide k-step), the staging-tile reads and
the "apply image" ``enc256_image_kernel`` writes -- evaluated in float64 on the exact fragments the kernel would read, in the
kernel's slab order, against oracle.restate.encoder_layer.  It pins the packing order and every index permutation; the
arithmetic of the real kernels is checked on the GPU (tests/test_gpu_encoder256.py)."""
import numpy as np
import torch

from detectorfreesfm_amd import ops
from oracle import restate

C, H, D = 256, 8, 32


def a_matrix(frag):
    """[64 lanes, 8] fragment as the 16 x 32 A operand: lane (i, g) slot j -> A[i][8 g + j] (the hardware's k position; any
    bijection works as long as A and B use the same one)."""
    f = np.asarray(frag, dtype=np.float64).reshape(4, 16, 8)              # [g, i, j]
    return np.concatenate([f[g] for g in range(4)], axis=1)                # [16, 32]


def b_matrix(frag):
    """[64 lanes, 8] fragment as the 32 x 16 B operand: lane (n, g) slot j -> B[8 g + j][n]."""
    return a_matrix(frag).T


def d_lanes(Dm):
    """16 x 16 result -> accumulator registers [64 lanes, 4]: lane (n, g) reg r = D[4 g + r][n]."""
    out = np.empty((64, 4))
    for g in range(4):
        for r in range(4):
            out[16 * g:16 * g + 16, r] = Dm[4 * g + r, :]
    return out


def to_frag16(a, b):
    """two consecutive accumulator blocks [64, 4] -> the operand fragment of their k-step [64, 8]"""
    return np.concatenate([a, b], axis=1)


def stream_frag(stream, slab, idx):
    f = stream[slab * 16 + 2 * idx].double() + stream[slab * 16 + 2 * idx + 1].double() / 2048.0
    return f.numpy()


def x_frag(x_tile, s):
    """enc256_apply_kernel's xfrag(): lane (n, g) slots j < 4: channels 32 s + 4 g + j; j >= 4: 32 s + 16 + 4 g + (j - 4)."""
    f = np.empty((64, 8))
    for g in range(4):
        cols = [32 * s + 4 * g + e for e in range(4)] + [32 * s + 16 + 4 * g + e for e in range(4)]
        f[16 * g:16 * g + 16] = x_tile[:, cols]
    return f


def rows_from_blocks(blocks):
    """16 accumulator blocks [[64, 4]] (lane = token) -> [16 tokens, 256 channels]: block B lane (n, g) reg r = channel 16 B + 4 g + r."""
    out = np.empty((16, C))
    for B, v in enumerate(blocks):
        for g in range(4):
            for r in range(4):
                out[:, 16 * B + 4 * g + r] = v[16 * g:16 * g + 16, r]
    return out


def blocks_from_rows(rows):
    out = []
    for B in range(16):
        v = np.empty((64, 4))
        for g in range(4):
            for r in range(4):
                v[16 * g:16 * g + 16, r] = rows[:, 16 * B + 4 * g + r]
        out.append(v)
    return out


def image_frags(KV):
    """enc256_image_kernel: KV [H, d, v] -> fragment (h, rb): lane (i, g) slot j = KV[h][16 (j >> 2) + 4 g + (j & 3)][16 rb + i]."""
    frags = {}
    for h in range(H):
        for rb in range(2):
            f = np.empty((64, 8))
            for lane in range(64):
                i, g = lane & 15, lane >> 4
                for j in range(8):
                    f[lane, j] = KV[h, 16 * (j >> 2) + 4 * g + (j & 3), 16 * rb + i]
            frags[2 * h + rb] = f
    return frags


def model_apply(fw, x, frags_kv, ksum, S, n1, n2):
    """enc256_apply_kernel for one wave tile (16 tokens of one sequence), slab by slab."""
    slab = [0]

    def gemm16(bfrag):                              # 16 slabs: k-step ks x row half nb2 (8 blocks each)
        acc = [np.zeros((16, 16)) for _ in range(16)]
        for ks in range(8):
            for nb2 in range(2):
                for b in range(8):
                    acc[8 * nb2 + b] += a_matrix(stream_frag(fw.stream, slab[0], b)) @ b_matrix(bfrag(ks))
                slab[0] += 1
        return [d_lanes(a) for a in acc]
    q = gemm16(lambda s: x_frag(x, s))
    q_rows = rows_from_blocks(q)
    phi = [np.where(v > 0, v, np.expm1(v)) + 1.0 for v in q]
    ks_l = blocks_from_rows(np.tile(ksum[None, :], (16, 1)))                # Ksum[16 B + 4 g + r] per lane, like the f32x4 reads
    a_fr, Z = {}, {}
    for h in range(H):
        z = (phi[2 * h] * ks_l[2 * h]).sum(1) + (phi[2 * h + 1] * ks_l[2 * h + 1]).sum(1)
        z = z.reshape(4, 16).sum(0)                                          # xor 16, xor 32: all four lane groups
        Z[h] = 1.0 / (np.tile(z, 4) + 1e-6)
        a_fr[h] = to_frag16(phi[2 * h], phi[2 * h + 1])
    msg = []
    for f in range(16):                                                      # block f = head f >> 1, row block f & 1
        acc = a_matrix(frags_kv[f]) @ b_matrix(a_fr[f >> 1])
        msg.append(d_lanes(acc) * Z[f >> 1][:, None] * S)
    msg_rows = rows_from_blocks(msg)
    m_fr = {h: to_frag16(msg[2 * h], msg[2 * h + 1]) for h in range(H)}
    merged = rows_from_blocks(gemm16(lambda s: m_fr[s]))
    m1 = torch.nn.functional.layer_norm(torch.from_numpy(merged), (C,), n1[0].double(), n1[1].double()).numpy()
    m1_blocks = blocks_from_rows(m1)
    n_fr = {s: to_frag16(m1_blocks[2 * s], m1_blocks[2 * s + 1]) for s in range(8)}
    o = [np.zeros((16, 16)) for _ in range(16)]
    for hc in range(8):
        hacc = [np.zeros((16, 16)) for _ in range(4)]
        for u in range(8):                                                   # k-steps 2u, 2u+1 of [x | norm1]
            for ks in range(2):
                kstep = 2 * u + ks
                bf = x_frag(x, kstep) if kstep < 8 else n_fr[kstep - 8]
                for b in range(4):
                    hacc[b] += a_matrix(stream_frag(fw.stream, slab[0], ks * 4 + b)) @ b_matrix(bf)
            slab[0] += 1
        hl = [np.maximum(d_lanes(a), 0.0) for a in hacc]
        h_fr = [to_frag16(hl[0], hl[1]), to_frag16(hl[2], hl[3])]
        for t in range(2):
            for nb2 in range(2):
                for b in range(8):
                    o[8 * nb2 + b] += a_matrix(stream_frag(fw.stream, slab[0], b)) @ b_matrix(h_fr[t])
                slab[0] += 1
    assert slab[0] == 128
    o_rows = rows_from_blocks([d_lanes(a) for a in o])
    out = x + torch.nn.functional.layer_norm(torch.from_numpy(o_rows), (C,), n2[0].double(), n2[1].double()).numpy()
    return q_rows, msg_rows, m1, o_rows, out


def model_kv(fw, src):
    """enc256_kv_kernel for one chunk [S, 256]: classic orientation (lane = channel, registers = tokens 4 g + r of a 16-token
    block), the kv stream's slabs in order (head h: slabs 4 h .. 4 h + 3 = k-steps 2 u, 2 u + 1 x blocks [k0, k1, v0, v1]), then the
    KV MFMAs whose A / B fragments are the accumulators with slots 4-7 zero.  Returns K1's partial layout (KV[h][d][v], Ksum[h][d])."""
    S = src.shape[0]
    KV = np.zeros((H, D, D))
    ksum = np.zeros((H, D))

    def x_frag_nat(tile, s):                       # lane (token, g) slots = channels 32 s + 8 g + j
        f = np.empty((64, 8))
        for g in range(4):
            f[16 * g:16 * g + 16] = tile[:, 32 * s + 8 * g:32 * s + 8 * g + 8]
        return f
    for s0 in range(0, S, 16):
        tile = np.zeros((16, C))
        n = min(16, S - s0)
        tile[:n] = src[s0:s0 + n]
        tm = np.zeros(16)
        tm[:n] = 1.0
        tm_l = np.stack([np.repeat(tm[4 * g + r], 16) for g in range(4) for r in range(4)]).reshape(4, 4, 16)   # [g, r, lane&15]
        tm_l = np.concatenate([tm_l[g].T for g in range(4)], 0)                                                  # [64 lanes, 4]
        for h in range(H):
            d = [np.zeros((16, 16)) for _ in range(4)]
            for u in range(4):
                for ks in range(2):
                    for b in range(4):
                        # D[token][channel] += A(x)[token][k] B(W)[k][channel]
                        d[b] += a_matrix(x_frag_nat(tile, 2 * u + ks)) @ b_matrix(stream_frag(fw.kv_stream, 4 * h + u, ks * 4 + b))
            acc = [d_lanes(m) for m in d]           # lane (n = channel, g), reg r = D[token 4 g + r][channel n]
            kf = [(np.where(acc[b] > 0, acc[b], np.expm1(acc[b])) + 1.0) * tm_l for b in range(2)]
            vf = [acc[2 + b] * tm_l / S for b in range(2)]
            z = np.zeros((64, 4))
            for kb in range(2):
                ks_l = kf[kb].sum(1)                # per lane: its channel, its 4 tokens
                ksum[h, 16 * kb:16 * kb + 16] += ks_l.reshape(4, 16).sum(0)
                for vb in range(2):
                    blk = a_matrix(to_frag16(kf[kb], z)) @ b_matrix(to_frag16(vf[vb], z))     # [16 d, 16 v]
                    KV[h, 16 * kb:16 * kb + 16, 16 * vb:16 * vb + 16] += blk
    return KV, ksum


def _layer(seed):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in (("q_proj", (C, C)), ("k_proj", (C, C)), ("v_proj", (C, C)), ("merge", (C, C)),
                        ("mlp.0", (2 * C, 2 * C)), ("mlp.2", (C, 2 * C))):
        sd[f"l.{name}.weight"] = (torch.randn(shape, generator=g) * 0.06).half().float()     # exactly representable in split form
    for nm in ("norm1", "norm2"):
        sd[f"l.{nm}.weight"] = 1.0 + 0.1 * torch.randn(C, generator=g)
        sd[f"l.{nm}.bias"] = 0.1 * torch.randn(C, generator=g)
    return sd, g


def test_fragment_stream_and_chaining_reproduce_the_layer():
    sd, g = _layer(5)
    n1 = (sd["l.norm1.weight"], sd["l.norm1.bias"])
    n2 = (sd["l.norm2.weight"], sd["l.norm2.bias"])
    fw = ops.Encoder256Weights(sd["l.q_proj.weight"], sd["l.merge.weight"], sd["l.mlp.0.weight"], sd["l.mlp.2.weight"], n1, n2,
                               wk=sd["l.k_proj.weight"], wv=sd["l.v_proj.weight"])
    assert fw.stream.shape == (2048, 64, 8) and fw.kv_stream.shape == (512, 64, 8)
    S, L = 45, 16
    src = torch.randn((1, S, C), generator=g).half().double()
    x = torch.randn((1, L, C), generator=g).half().double()
    sd64 = {k: v.double() for k, v in sd.items()}
    # source side as the product computes it: k | v projection, then KV[h][d][v] = sum_s phi(k)[s,h,d] v[s,h,v] / S, Ksum
    k = restate.elu1((src[0] @ sd64["l.k_proj.weight"].T).view(S, H, D))
    v = (src[0] @ sd64["l.v_proj.weight"].T).view(S, H, D)
    KV = torch.einsum("shd,shv->hdv", k, v / S).numpy()
    ksum = k.sum(0).reshape(-1).numpy()
    # the fused source-side kernel reproduces both from the tokens and the kv stream (ragged: 45 = 2 blocks of 16 + 13 tokens)
    KV_m, ksum_m = model_kv(fw, src[0].numpy())
    assert np.abs(KV_m - KV).max() < 1e-12 and np.abs(ksum_m.reshape(-1) - ksum).max() < 1e-10
    q, msg, m1, o, out = model_apply(fw, x[0].numpy(), image_frags(KV_m), ksum_m.reshape(-1), S, n1, n2)
    ref = restate.encoder_layer(sd64, "l.", x, src, H)[0].numpy()
    qr = (x[0] @ sd64["l.q_proj.weight"].T).numpy()
    assert np.abs(q - qr).max() < 1e-10
    m_ref = restate.linear_attention(torch.from_numpy(qr).view(1, L, H, D), (src[0] @ sd64["l.k_proj.weight"].T).view(1, S, H, D),
                                     v.view(1, S, H, D)).reshape(L, C).numpy()
    assert np.abs(msg - m_ref).max() < 1e-9
    assert np.abs(out - ref).max() < 1e-9


def test_kslot16_order_and_staging_swizzle():
    ks = ops._kslots16(3)
    assert ks[0].tolist() == [96, 97, 98, 99, 112, 113, 114, 115] and ks[3].tolist() == [108, 109, 110, 111, 124, 125, 126, 127]
    assert sorted(ks.reshape(-1).tolist()) == list(range(96, 128))
    # staging tile of a wave: 16 rows x 512 B per plane, 16-byte chunk c of row t at t * 512 + ((c ^ (t & 15)) << 4).  The
    # 8-byte fragment reads of lanes 0-31 (tokens 0-15 x lane groups 0, 1: same logical chunk) must hit 64 distinct banks
    def off(t, c):
        return t * 512 + ((c ^ (t & 15)) << 4)
    for s in range(8):
        for gpair in range(2):
            banks = []
            for grp in (2 * gpair, 2 * gpair + 1):
                for tok in range(16):
                    a = off(tok, 4 * s + (grp >> 1)) + 8 * (grp & 1)
                    banks += [(a // 4) % 64, (a // 4 + 1) % 64]
            assert len(set(banks)) == 64
    # the swizzle is a bijection of a row's 32 chunks
    assert all(sorted((c ^ (t & 15)) for c in range(32)) == list(range(32)) for t in range(16))
