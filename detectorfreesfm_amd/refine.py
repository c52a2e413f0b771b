"""MI355X multiview refinement head behind the reference's ``refinement_models`` /
``MultiviewMatcher`` plugin surface.

``HipMultiviewMatcher`` is a drop-in for ``MultiviewMatcher(config, test=True).eval()``
(src/MultiviewMatcher/MultiviewMatcher.py:17-405) as built by
src/post_optimization/matcher_model/multiview_match_worker.py:16-56: same config dict
(``model.multiview_refinement``), same 72-tensor ``state_dict`` layout (``backbone.*``,
``fine_transformer.*``), same in-place ``forward(data)`` contract (``query_points_refined``,
``reference_points_refined``, ``std`` -- read by ``extract_results`` :59-82).

Hand-written HIP: RoIAlign patch extraction (K8, with the ImageNet normalisation fused, NHWC
output in (track, view) order), the S2DNet patch CNN (K9) and every nn.Linear (K10) on the
fp16x2-split MFMA implicit-GEMM kernel with fused bias/BN/ReLU/residual, max pooling, linear
attention (K1, D=16), LayerNorm + residual, and the fused fine correlation / softmax expectation /
candidate argmin / refined keypoints (K11+K12).  No library (MIOpen / hipBLASLt) path exists in this package.

Output-identical work the reference wastes is skipped ("dead work"):
* only the centre (W+4)^2 of relu1_2 feeds adaptation layer 0 (the reference convolves the full
  35x35 map and crops to WxW afterwards, s2dnet.py:164-193);
* the bicubic align_corners upsample of adaptation layer 1 is evaluated only at the WxW centre, by a separable
  resampling kernel fed with PyTorch's own interpolation coefficients;
* padded view slots (image index -1) are never cropped or convolved: they are masked everywhere
  downstream, the reference feeds them a copy of the last patch (MultiviewMatcher.py:253-266).
"""

import torch
import torch.nn.functional as F      # _bicubic_rows: PyTorch's own interpolation coefficients, on the host

from . import ops
from .coarse import EncoderLayerWeights, encoder_layer_fused128, encoder_layer_split
from .params import ParamModule, multiview_param_spec

IMAGENET_MEAN = (0.485, 0.456, 0.406)   # S2DNet.mean / .std, backbone/S2DNet/s2dnet.py:66-67
IMAGENET_STD = (0.229, 0.224, 0.225)


def _bicubic_rows(n_in: int, n_out: int, lo: int, hi: int) -> torch.Tensor:
    """Rows [lo,hi) of the (n_out x n_in) matrix of nn.Upsample(mode='bicubic', align_corners=True)
    along one axis, obtained from PyTorch's own kernel applied to the identity."""
    eye = torch.eye(n_in, dtype=torch.float32).view(1, n_in, n_in, 1)          # [1, k, in, 1]
    up = F.interpolate(eye, size=(n_out, 1), mode="bicubic", align_corners=True)  # [1, k, out, 1]
    return up[0, :, lo:hi, 0].t().contiguous()                                    # [hi-lo, k]


class HipMultiviewMatcher(ParamModule):
    def __init__(self, config: dict, test: bool = True, max_backbone_patches: int = 16384, **_unused):
        super().__init__()
        # tap-reuse ("same") schedule for the stride-1 3x3 / 5x5 convs; False packs them for the flattened-K kernel (set before
        # the first forward; tests/test_gpu_e2e.py::test_flattened_k_conv_schedule_equals_same_schedule)
        self.same_conv = True
        # conv1_1 -> conv1_2 -> pool / centre window as ONE launch (ops.s2d_front); False: the three separate layers
        # (tests/test_gpu_kernels.py::test_s2d_front_*, tests/test_gpu_e2e.py::test_refine_fused_front_equals_three_launches)
        self.fused_front = True
        self.direct_features = True     # r06: the backbone's features go to the transformer as split planes (see forward)
        if not test:
            raise NotImplementedError("training path is out of scope; build with test=True")
        bb = config["backbone"]
        s2d = bb["s2dnet"]
        mt = config["multiview_transform"]
        mtest = config["multiview_matching_test"]
        ok = (bb["type"] == "S2DNet" and s2d["num_layers"] == 2 and s2d["combine"] and
              s2d["substitute_pooling_layers"] and s2d["zoomin_strategy"] == "post" and
              config["n_matching_steps"] == 1 and not config["enable_multiview_scale_align"] and
              mt["sparse"] and not mt["enable_rescaled_crop"] and mt["attention"] == "linear" and
              mt["attention_type"] == "multiview" and mt["norm_method"] == "layernorm" and
              mt["rezero"] is None and not mt["final_proj"] and mt["type"] == "LoFTR" and
              # what the fused K11/K12 kernel hard-codes (fine_matching.py:36-98,129-179,258-285)
              mtest["type"] == "s2d" and mtest["best_left_strategy"] == "smallest_mean_std" and
              mtest["s2d"]["type"] == "heatmap" and mtest["s2d"]["obtain_offset_method"] == "argsoftmax")
        if not ok:
            raise NotImplementedError("HipMultiviewMatcher implements the shipped refinement configuration "
                                      "(hydra_training_configs/experiment/multiview_refinement_matching.yaml)")
        W, crop = mt["window_size"], mt["crop_size"]
        if not (crop // 2 - W // 2 - 2 >= 0 and crop // 2 + W // 2 + 3 <= crop):
            # the dead-work shortcut evaluates adaptation layer 0 on the centre (W+4)^2 crop with a pad-0 5x5; it
            # equals the reference's pad-2 convolution of the full map only while that halo stays inside the patch
            raise NotImplementedError(f"window_size {W} too close to crop_size {crop}: need crop >= window + 6")
        self.config = config
        # patches per S2DNet pass (bounds the activation buffers; results do not depend on it:
        # tests/test_gpu_e2e.py::test_refine_backbone_patch_chunks_are_invisible)
        self.max_backbone_patches = int(max_backbone_patches)
        self.register_spec(multiview_param_spec(config))
        self.register_buffer("_mean", torch.tensor(IMAGENET_MEAN), persistent=False)
        self.register_buffer("_std", torch.tensor(IMAGENET_STD), persistent=False)
        self._packed = None

    def load_state_dict(self, state_dict, *args, **kwargs):
        self._packed = None
        return super().load_state_dict(state_dict, *args, **kwargs)

    def _apply(self, fn, *a, **kw):
        self._packed = None
        return super()._apply(fn, *a, **kw)

    # -- weight packing -----------------------------------------------------------------------
    def _pack(self):
        g = self.p
        P = {"enc": {i: (g(f"backbone.encoder.{i}.weight"), g(f"backbone.encoder.{i}.bias"))
                     for i in (0, 2, 5, 7, 10, 12, 14)}}
        for i in (0, 1):
            q = f"backbone.adaptation_layers.adap_layer_{i}."
            w5, b5 = g(q + "2.weight"), g(q + "2.bias")
            s = g(q + "3.weight").double() / torch.sqrt(g(q + "3.running_var").double() + 1e-5)
            # conv(+bias) -> eval BN folded: w*s, (b - mean)*s + beta -- in float64, split from the exact products (coarse._fold_bn)
            P[f"adap{i}"] = (g(q + "0.weight"), g(q + "0.bias"), (w5.double() * s[:, None, None, None]).contiguous(),
                             ((b5.double() - g(q + "3.running_mean").double()) * s + g(q + "3.bias").double()).contiguous())
        mt = self.config["multiview_transform"]
        n_layers = len(mt["layer_names"]) * mt["layer_iter_n"]
        P["layers"] = [EncoderLayerWeights(g, f"fine_transformer.layers.{i}.") for i in range(n_layers)]

        def pk(w, b, split_in=True, same=False):    # same: stride-1 'same' conv -> activation-reuse kernel
            return ops.PackedDense(w, b, cin_pad=(w.shape[1] + 7) // 8 * 8 if split_in else None,
                                   tap_padded=same and split_in and self.same_conv)
        H = {"enc": {i: pk(*P["enc"][i], split_in=(i != 0), same=True) for i in P["enc"]}}   # conv1_1 reads fp32 patches
        # conv1_1 -> conv1_2 -> {centre window, pool} as one launch (csrc/s2d_front.hip)
        H["front"] = ops.S2dFrontWeights(*P["enc"][0], *P["enc"][2])
        for i in (0, 1):
            a = P[f"adap{i}"]
            H[f"adap{i}"] = (pk(a[0], a[1]), pk(a[2], a[3], same=(i == 1)))   # adap1's 5x5 is pad 2, adap0's pad 0
        P["hip"] = H
        self._packed = P
        return P

    # -- K9 on the hand-written NHWC kernels: out [m, W*W, C] tokens written straight into `dst` ------
    def _s2dnet_hip(self, x, P, W, dst):
        """x [m,crop,crop,3] normalised NHWC patches; dst [m, W*W, od] receives adap0 + bicubic(adap1)."""
        H = P["hip"]
        m, crop = x.shape[0], x.shape[1]
        c, r = crop // 2, W // 2
        S = dict(relu=True, out_split=True)             # conv -> ReLU -> split planes for the next conv
        if self.fused_front and crop == ops.SUPPORTED_S2D_FRONT_PATCH and x.is_contiguous():
            if ops.range_check_active():                # the fused launch keeps relu1_1 in LDS: show it to the range guard
                ops.conv2d_nhwc(x, H["enc"][0], 1, 1, **S)
            f0, t = ops.s2d_front(x, H["front"], c - r - 2, c + r + 3)             # centre (W+4)^2 of relu1_2, pooled relu1_2
        else:
            x = ops.conv2d_nhwc(x, H["enc"][0], 1, 1, **S)
            x = ops.conv2d_nhwc(x, H["enc"][2], 1, 1, **S)                        # relu1_2 [m,35,35,64]
            f0 = x.crop(c - r - 2, c + r + 3, c - r - 2, c + r + 3)                 # centre (W+4)^2 view
            t = ops.maxpool3x3s2_nhwc(x)
        t = ops.conv2d_nhwc(ops.conv2d_nhwc(t, H["enc"][5], 1, 1, **S), H["enc"][7], 1, 1, **S)
        t = ops.maxpool3x3s2_nhwc(t)
        for i in (10, 12, 14):
            t = ops.conv2d_nhwc(t, H["enc"][i], 1, 1, **S)                         # relu3_3 [m,9,9,256]
        y1 = ops.conv2d_nhwc(ops.conv2d_nhwc(t, H["adap1"][0], 1, 0, **S), H["adap1"][1], 1, 2)
        h4 = y1.shape[1]
        key = (h4, crop, W)
        if P.get("bicubic_rows_key") != key:
            P["bicubic_rows"] = _bicubic_rows(h4, crop, c - r, c + r + 1).to(y1.device)   # [W, h4]: torch's own coefficients
            P["bicubic_rows_key"] = key
        up = ops.resample_separable(y1, P["bicubic_rows"], P["bicubic_rows"])     # [m, W*W, od]
        a0 = ops.conv2d_nhwc(f0, H["adap0"][0], 1, 0, **S)                        # [m,W+4,W+4,64]
        if isinstance(dst, ops.SplitAct):                                          # + hypercolumn sum, fused
            ops.conv2d_nhwc(a0, H["adap0"][1], 1, 0, residual=up, out_split=dst)   # ... as split planes for the transformer
        else:
            ops.conv2d_nhwc(a0, H["adap0"][1], 1, 0, residual=up, out=dst)
        return dst

    @torch.no_grad()
    @ops.first_call_range_sweep
    def forward(self, data: dict, chunk_track: int = 1000, chunk_backbone_img: bool = True):
        """Updates ``data`` in place like MultiviewMatcher.forward(test) (MultiviewMatcher.py:59-405)."""
        P = self._packed or self._pack()
        cfg = self.config
        mt = cfg["multiview_transform"]
        W, crop = mt["window_size"], mt["crop_size"]
        left = cfg["multiview_matching_test"]["left_point_movement_window_size"]
        if left is None:
            raise NotImplementedError("left_point_movement_window_size=None (training config)")
        if "keypoint_relocalized_offset" in data:
            # the reference overrides best_index with it (fine_matching.py:150-158); silently ignoring it would diverge
            raise NotImplementedError("data['keypoint_relocalized_offset'] (keypoint relocalisation) is not supported")
        WW = W * W
        C = mt["d_model"]
        images = data["images"]
        if isinstance(images, torch.Tensor):
            images = [images[:, i] for i in range(images.shape[1])]
        n_img = len(images)
        dev = images[0].device
        if images[0].shape[0] != 1:
            raise NotImplementedError("batch size 1 only (as the reference, fine_preprocess.py:40)")
        data["W"] = W

        fine_res = float(cfg["backbone"]["resolution"][-1])
        scales = torch.full((1, n_img, 2), fine_res, device=dev)
        if "scales" in data:
            scales = scales * data["scales"][:, :, [1, 0]]
        ref_coarse = data["reference_points_coarse"].contiguous()                     # [1,V-1,T,2]
        pts = torch.cat([data["query_points"][:, None], ref_coarse], dim=1)          # [1,V,T,2]
        img_idxs = torch.cat([data["query_img_idxs"][:, None], data["reference_img_idxs"]], dim=1)
        _, V, T = img_idxs.shape
        pt_scales = scales.view(-1, 2)[img_idxs.view(-1)].view(1, V, T, 2).contiguous()   # -1 -> last image (:103)
        pts = pts / pt_scales

        # ---- view-count grouping (MultiviewMatcher.py:117-133); one host sync ---------------------
        tvm = data["track_valid_mask"]                                                 # [1,V-1,T]
        keys, counts = torch.unique(tvm.sum(-2).max(0)[0], sorted=True, return_counts=True)
        keys, counts = keys.flip(0).tolist(), counts.flip(0).tolist()
        max_view_tracks = 16 * chunk_track
        groups, i = [], 0
        while i < len(keys):
            vv = int(keys[i]) + 1
            if vv * counts[i] <= max_view_tracks:
                groups.append((vv, counts[i]))
                i += 1
            else:
                groups.append((vv, max_view_tracks // vv))
                counts[i] -= max_view_tracks // vv

        # ---- K8 crop + K9 backbone, valid view slots only; features land in [T,V,WW,C] -------------
        flat_idx = img_idxs.reshape(-1)                                                # (v t) order
        flat_pts = pts.reshape(-1, 2)
        order = torch.argsort(flat_idx, stable=True)
        per_img = torch.bincount(flat_idx.clamp(min=-1) + 1, minlength=n_img + 1).tolist()   # one sync
        n_pad, per_img = per_img[0], per_img[1:]
        order = order[n_pad:]                                                          # valid, grouped by image
        M = order.numel()
        r = crop // 2
        boxes = torch.cat([flat_pts - r, flat_pts + r], dim=-1)[order].contiguous()     # fine_preprocess.py:101-104
        names = list(mt["layer_names"]) * mt["layer_iter_n"]
        # r06: when every (track, view) slot is valid and the transformer runs on the fused kernels, the backbone writes its features
        # as split planes in the order the transformer reads them -- reference views of all tracks first, then (track, view >= 1) --
        # and the first encoder layer takes them as they are: no fp32 feature tensor, no split_rows pass (0.4 ms of a 2000-track bag)
        direct = (self.direct_features and n_pad == 0 and all(g_[0] == V for g_ in groups) and sum(g_[1] for g_ in groups) == T
                  and V > 1 and bool(mt["enable"])
                  and bool(names) and all(w.fused is not None for w in P["layers"]) and mt["nhead"] == 8 and WW >= 32
                  and not ops.range_check_active())
        if direct:
            vv_, tt_ = order // T, order % T
            slot = torch.where(vv_ == 0, tt_, T + tt_ * (V - 1) + vv_ - 1)
        else:
            slot = (order % T) * V + order // T                                        # (v t) -> (t v)
        # patch m of the compact list goes to position rank(slot): the CNN then emits features
        # already in (track, view) order -- no gather afterwards (MultiviewMatcher.py:264-270)
        by_slot = torch.argsort(slot)
        pos = torch.empty_like(by_slot)
        pos[by_slot] = torch.arange(M, device=dev)
        patches = torch.empty((M, crop, crop, 3), dtype=torch.float32, device=dev)
        start = 0
        for ii in range(n_img):
            n = per_img[ii]
            if n == 0:
                continue
            ops.roi_align(images[ii], boxes[start:start + n], crop, crop, out_slot=pos[start:start + n],
                          mean=self._mean, std=self._std, out=patches, channels_last=True)
            start += n
        dense = n_pad == 0                      # every (track, view) slot is valid: write in place
        if direct:
            feats = comp = ops.SplitAct.empty_rows((T * V, WW), C, dev)
        else:
            feats = torch.empty((T * V, WW, C), dtype=torch.float32, device=dev) if dense else \
                torch.zeros((T * V, WW, C), dtype=torch.float32, device=dev)
            comp = feats if dense else torch.empty((M, WW, C), dtype=torch.float32, device=dev)
        for s in range(0, M, self.max_backbone_patches):
            e = min(M, s + self.max_backbone_patches)
            self._s2dnet_hip(patches[s:e], P, W, comp[s:e])
        if not dense:
            feats.index_copy_(0, slot[by_slot], comp)
            del comp
        del patches
        if not direct:
            feats = feats.view(T, V, WW, C)

        # ---- K10 transformer + K11/K12 fine matching per view-count group ---------------------------
        movable = data["query_movable_mask"][0].contiguous() if "query_movable_mask" in data else None
        qpts = data["query_points"][0].contiguous()                                    # original scale
        tmask = tvm[0].transpose(0, 1).contiguous()                                    # [T,V-1]
        q_out = torch.empty((T, 2), dtype=torch.float32, device=dev)
        r_out = torch.zeros((1, V - 1, T, 2), dtype=torch.float32, device=dev)
        s_out = torch.zeros((1, V - 1, T), dtype=torch.float32, device=dev)
        nhead = mt["nhead"]
        i = 0
        for cv, nt in groups:
            sl = slice(i, i + nt)
            Vq = cv - 1
            if Vq == 0:      # tracks without any valid reference view: nothing to refine
                q_out[sl] = qpts[sl]
                i += nt
                continue
            qm = tmask[sl, :Vq].contiguous()
            if mt["enable"] and names:
                # split planes [., ., 2C] = [x | norm1(message)], ping-pong; the last layer writes dense [., ., C] planes
                rs = [ops.SplitAct.empty_rows((nt, WW), 2 * C, dev) for _ in range(2)]
                qs = [ops.SplitAct.empty_rows((nt, Vq * WW), 2 * C, dev) for _ in range(2)]
                if direct:      # the backbone's split planes: [T reference sequences | T query sequences of Vq views]
                    r_in = ops.SplitAct(feats.hi[:T][sl], feats.lo[:T][sl], C)
                    q_in = ops.SplitAct(feats.hi[T:].view(T, Vq * WW, C)[sl], feats.lo[T:].view(T, Vq * WW, C)[sl], C)
                else:
                    ops.split_rows(feats[sl, 0], None, out_split=rs[0].cols(0, C))            # strided [nt, WW, C] blocks: no copy
                    ops.split_rows(feats[sl, 1:cv].reshape(nt, Vq * WW, C), None, out_split=qs[0].cols(0, C))
                for li, (w, name) in enumerate(zip(P["layers"], names)):   # matcher_module/transformer.py:158-172
                    last = li == len(names) - 1
                    if last:     # the final features stay split planes (dense [., ., C]): the fine-matching kernel streams them
                        ref = ors = ops.SplitAct.empty_rows((nt, WW), C, dev)
                        qry = oqs = ops.SplitAct.empty_rows((nt, Vq * WW), C, dev)
                    else:
                        ors, oqs = rs[1].cols(0, C), qs[1].cols(0, C)
                    if name not in ("self", "cross"):
                        raise NotImplementedError(name)
                    if direct and li == 0:                # x = the backbone's planes (row stride C): the fused kernels directly
                        if name == "self":
                            encoder_layer_fused128(w, r_in, r_in, ors)
                            encoder_layer_fused128(w, q_in, q_in, oqs, qm, qm, WW, WW)
                        else:
                            encoder_layer_fused128(w, q_in, r_in, oqs, qm, None, WW, 1)
                            encoder_layer_fused128(w, r_in, q_in, ors, None, qm, 1, WW)
                    elif name == "self":
                        encoder_layer_split(w, rs[0], rs[0].cols(0, C), None, ors, nhead, is_self=True)
                        encoder_layer_split(w, qs[0], qs[0].cols(0, C), None, oqs, nhead, qm, qm, WW, WW, is_self=True)
                    else:                                 # cross: both sides from the PRE-update tensors (:163)
                        encoder_layer_split(w, qs[0], rs[0].cols(0, C), None, oqs, nhead, qm, None, WW, 1)
                        encoder_layer_split(w, rs[0], qs[0].cols(0, C), None, ors, nhead, None, qm, 1, WW)
                    rs.reverse(); qs.reverse()
                qry = ops.SplitAct(qry.hi.view(nt, Vq, WW, C), qry.lo.view(nt, Vq, WW, C), C)
            else:
                ref = feats[sl, 0].contiguous()
                qry = feats[sl, 1:cv].reshape(nt, Vq, WW, C)
            m = ops.fine_match(ref, qry, qm, None if movable is None else movable[sl],
                               W, left, qpts[sl], pt_scales[0, 0, sl], ref_coarse[0, :, sl],
                               pt_scales[0, 1:, sl])
            q_out[sl] = m["query_refined"]
            r_out[0, :Vq, sl] = m["ref_refined"].transpose(0, 1)
            s_out[0, :Vq, sl] = m["std"].transpose(0, 1)
            i += nt

        data["query_points_refined"] = q_out[None]
        data["fine_local_heatmap_pred"] = None     # the heat-map is never materialised
        if "reference_points_refined" in data:
            data["reference_points_refined"].append(r_out)
            data["std"].append(s_out)
        else:
            data["reference_points_refined"] = [r_out]
            data["std"] = [s_out]
        return None
