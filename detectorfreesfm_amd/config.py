"""Default configurations of the two hot-path plugins, as plain (lower-cased) dicts.

These are the *values* the reference's plugin builders hand to the models:

* coarse:  ``lower_config(get_cfg_defaults())['loftr']`` merged with
  third_party/LoFTR/configs/loftr/outdoor/loftr_ds_coarse_only.py, then
  ``match_coarse.thr = match_thr`` and ``coarse.temp_bug_fix = False``
  (src/coarse_match/coarse_match_worker.py:27-35; defaults third_party/LoFTR/src/config/default.py:4-45).
* refine:  ``OmegaConf.load(cfg_path)['model']['multiview_refinement']``
  (hydra_training_configs/experiment/multiview_refinement_matching.yaml:21-91) with the
  window rewrite of src/post_optimization/matcher_model/multiview_match_worker.py:20-34.
"""
import copy


def loftr_coarse_only_config(match_thr: float = 0.2) -> dict:
    return {
        "backbone_type": "ResNetFPN",
        "resolution": (8, 2),
        "fine_window_size": 5,
        "fine_concat_coarse_feat": True,
        "resnetfpn": {"initial_dim": 128, "block_dims": [128, 196, 256]},
        "coarse": {"d_model": 256, "d_ffn": 256, "nhead": 8,
                   "layer_names": ["self", "cross"] * 4, "attention": "linear",
                   "temp_bug_fix": False},
        "match_coarse": {"thr": match_thr, "border_rm": 2, "match_type": "dual_softmax",
                         "dsmax_temperature": 0.1, "skh_iters": 3, "skh_init_bin_score": 1.0,
                         "skh_prefilter": False, "train_coarse_percent": 0.3,
                         "train_pad_num_gt_min": 200, "sparse_spvs": True},
        "fine": {"enable": False, "d_model": 128, "d_ffn": 128, "nhead": 8,
                 "layer_names": ["self", "cross"], "attention": "linear"},
    }


_MULTIVIEW_REFINEMENT = {
    "n_matching_steps": 1,
    "enable_multiview_scale_align": False,
    "backbone": {
        "type": "S2DNet",
        "resolution": [4, 1],
        "s2dnet": {"name": "s2dnet", "num_layers": 2, "window_size": 15, "checkpointing": None,
                   "output_dim": 128, "pretrained": None, "substitute_pooling_layers": True,
                   "combine": True, "zoomin_strategy": "post"},
        "pretrained": None,
        "pretrained_fix": False,
    },
    "use_fine_backbone_as_coarse": False,
    "interpol_type": "bilinear",
    "multiview_transform": {
        "sparse": True, "crop_size": 35, "window_size": 15, "enable_rescaled_crop": False,
        "enable": True, "type": "LoFTR", "d_model": 128, "nhead": 8,
        "layer_names": ["self", "cross"], "layer_iter_n": 2, "dropout": 0.0,
        "attention": "linear", "norm_method": "layernorm", "attention_type": "multiview",
        "kernel_fn": "elu + 1", "d_kernel": 16, "redraw_interval": 2, "rezero": None,
        "final_proj": False,
    },
    "multiview_matching_train": {
        "enable": True, "type": "s2d", "detector": "OnGrid", "window_size": 15,
        "left_point_movement_window_size": None, "best_left_strategy": "smallest_mean_std",
        "s2d": {"type": "heatmap", "obtain_offset_method": "argsoftmax"},
    },
    "multiview_matching_test": {
        "enable": True, "type": "s2d", "detector": "OnGrid", "window_size": 15,
        "left_point_movement_window_size": 7, "best_left_strategy": "smallest_mean_std",
        "s2d": {"type": "heatmap", "obtain_offset_method": "argsoftmax"},
    },
}


def multiview_refinement_config(rewindow_size_factor=None) -> dict:
    """Shipped refinement config; ``rewindow_size_factor`` applies the per-iteration window
    shrink of multiview_match_worker.py:20-34 (iter0: W=15,left=7; factor 2: W=11,left=3)."""
    cfg = copy.deepcopy(_MULTIVIEW_REFINEMENT)
    if rewindow_size_factor is not None:
        w = cfg["multiview_transform"]["window_size"]
        w = max(7, ((w // 2) - rewindow_size_factor) * 2 + 1)
        cfg["backbone"]["s2dnet"]["window_size"] = w
        cfg["multiview_transform"]["window_size"] = w
        cfg["multiview_matching_test"]["window_size"] = w
        lw = cfg["multiview_matching_test"]["left_point_movement_window_size"]
        if lw is not None:
            lw = max(3, ((lw // 2) - rewindow_size_factor) * 2 + 1)
            cfg["multiview_matching_test"]["left_point_movement_window_size"] = lw
    return cfg
