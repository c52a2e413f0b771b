"""TEST INFRASTRUCTURE ONLY -- loader for the *real* reference (zju3dv/DetectorFreeSfM).

This module imports the reference's own Python modules from ``/root/reference``
UNCHANGED, so that (a) ``oracle/restate.py`` can be validated against the code it
restates and (b) ``oracle/make_golden.py`` can generate the fixtures committed under
``tests/golden/``.  ``/root/reference`` only exists in the build container, never on the
GPU box, so nothing here may be imported by ``-m gpu`` tests, ``smoke()`` or ``bench.py``.

The reference depends on packages that are not installed here.
None of their arithmetic is on the coarse path; on the refinement path two kornia
helpers and the un-vendored ``roi_align`` extension are executed.  We register minimal
stand-ins in ``sys.modules`` *before* importing the reference:

* ``yacs.config.CfgNode``          -- attribute dict (third_party/LoFTR/src/config/default.py:1)
* ``kornia...dsnt.spatial_expectation2d``, ``kornia.utils.grid.create_meshgrid``
                                   -- executed at src/MultiviewMatcher/utils/fine_matching.py:274-275
* ``loguru.logger``                -- no-op logger
* ``omegaconf.OmegaConf``          -- merge/set_struct/set_readonly on plain dicts
                                      (src/MultiviewMatcher/backbone/S2DNet/base_model.py:21-26)
* ``torchvision.models.vgg16``     -- architecture only (s2dnet.py:86-87)
* ``timm.models.registry.register_model`` -- identity decorator (backbone/resnet.py:7)
* ``roi_align.roi_align.RoIAlign`` -- NOT the reference's code: the un-vendored
                                      longcw/RoIAlign.pytorch submodule; stand-in =
                                      ``oracle.restate.roi_align_crop`` (PARITY UNPINNED for
                                      this stage, see oracle/restate.py header)
* ``src`` package shim             -- bypasses src/__init__.py (imports ray/hloc/natsort)
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("DFSFM_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "third_party", "LoFTR", "src", "loftr"))


class _AttrDict(dict):
    """yacs.CfgNode stand-in: dict with attribute access (enough for default.py)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        import copy
        return copy.deepcopy(self)


class _DictConf(dict):
    """omegaconf node stand-in: nested attribute dict."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = _DictConf(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def install_stubs():
    import torch
    import torch.nn as nn
    import PIL.Image  # noqa: F401  (base_model.py:39 annotates with PIL.Image)

    if "yacs" not in sys.modules:
        yacs = _mod("yacs")
        yc = _mod("yacs.config")
        yc.CfgNode = _AttrDict
        yacs.config = yc

    if "loguru" not in sys.modules:
        lg = _mod("loguru")

        class _L:
            def __getattr__(self, _):
                return lambda *a, **k: None
        lg.logger = _L()

    if "kornia" not in sys.modules:
        k = _mod("kornia")
        kg = _mod("kornia.geometry")
        ks = _mod("kornia.geometry.subpix")
        kd = _mod("kornia.geometry.subpix.dsnt")
        ku = _mod("kornia.utils")
        kug = _mod("kornia.utils.grid")

        def create_meshgrid(h, w, normalized_coordinates=True, device=None):
            # kornia 0.4.1 semantics: [1,h,w,2], last dim (x,y), linspace(-1,1) when normalised
            xs = torch.linspace(0, w - 1, w, device=device)
            ys = torch.linspace(0, h - 1, h, device=device)
            if normalized_coordinates:
                xs = (xs / (w - 1) - 0.5) * 2
                ys = (ys / (h - 1) - 0.5) * 2
            gy, gx = torch.meshgrid(ys, xs, indexing="ij")
            return torch.stack([gx, gy], dim=-1)[None]

        def spatial_expectation2d(heatmap, normalized_coordinates=True):
            b, n, h, w = heatmap.shape
            grid = create_meshgrid(h, w, normalized_coordinates, heatmap.device).to(heatmap.dtype)
            px = grid[..., 0].reshape(-1)
            py = grid[..., 1].reshape(-1)
            flat = heatmap.reshape(b, n, -1)
            ex = torch.sum(px * flat, -1, keepdim=True)
            ey = torch.sum(py * flat, -1, keepdim=True)
            return torch.cat([ex, ey], -1)

        kd.spatial_expectation2d = spatial_expectation2d
        ks.dsnt = kd
        kg.subpix = ks
        k.geometry = kg
        kug.create_meshgrid = create_meshgrid
        ku.grid = kug
        ku.create_meshgrid = create_meshgrid
        k.utils = ku

    if "omegaconf" not in sys.modules:
        oc = _mod("omegaconf")

        class OmegaConf:
            @staticmethod
            def merge(*cfgs):
                out = {}
                for c in cfgs:
                    out.update(dict(c))
                return _DictConf(out)

            @staticmethod
            def set_struct(c, v):
                return None

            @staticmethod
            def set_readonly(c, v):
                return None

            @staticmethod
            def create(d):
                return _DictConf(d)
        oc.OmegaConf = OmegaConf

    if "torchvision" not in sys.modules:
        tv = _mod("torchvision")
        tvm = _mod("torchvision.models")
        tvt = _mod("torchvision.transforms")
        tvf = _mod("torchvision.transforms.functional")
        tv.transforms = tvt
        tvt.functional = tvf

        def vgg16(pretrained=False, **kw):
            cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
            layers, c_in = [], 3
            for v in cfg:
                if v == "M":
                    layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
                else:
                    layers += [nn.Conv2d(c_in, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                    c_in = v
            m = nn.Module()
            m.features = nn.Sequential(*layers)
            return m
        tvm.vgg16 = vgg16
        tv.models = tvm

    if "timm" not in sys.modules:
        t = _mod("timm")
        tm = _mod("timm.models")
        tr = _mod("timm.models.registry")
        tr.register_model = lambda f: f
        tm.registry = tr
        t.models = tm

    if "roi_align" not in sys.modules:
        ra = _mod("roi_align")
        rr = _mod("roi_align.roi_align")
        from oracle.restate import roi_align_crop

        class RoIAlign(nn.Module):
            def __init__(self, crop_height, crop_width, extrapolation_value=0, transform_fpcoor=True):
                super().__init__()
                self.crop_height, self.crop_width = crop_height, crop_width
                self.extrapolation_value = extrapolation_value
                self.transform_fpcoor = transform_fpcoor

            def forward(self, featuremap, boxes, box_ind):
                assert not self.transform_fpcoor
                return roi_align_crop(featuremap, boxes, box_ind.reshape(-1), self.crop_height,
                                      self.crop_width, self.extrapolation_value)
        rr.RoIAlign = RoIAlign
        ra.roi_align = rr

    # `src` package shim: real src/__init__.py pulls ray/hloc/natsort
    if "src" not in sys.modules or not hasattr(sys.modules["src"], "__path__"):
        src = _mod("src")
        src.__path__ = [os.path.join(REFERENCE_ROOT, "src")]
        su = _mod("src.utils")
        su.__path__ = [os.path.join(REFERENCE_ROOT, "src", "utils")]
        sp = _mod("src.utils.profiler")

        class PassThroughProfiler:
            def record_function(self, name):
                import contextlib
                return contextlib.nullcontext()
            profile = record_function
        sp.PassThroughProfiler = PassThroughProfiler
        su.profiler = sp
        src.utils = su


def _ensure_path():
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def import_loftr():
    """Return the reference's ``LoFTR`` class and its lower-cased default config dict."""
    _ensure_path()
    install_stubs()
    from third_party.LoFTR.src.loftr.loftr import LoFTR
    default = importlib.import_module("third_party.LoFTR.src.config.default")

    def lower(c):
        if not isinstance(c, dict):
            return list(c) if isinstance(c, tuple) else c
        return {k.lower(): lower(v) for k, v in c.items()}
    cfg = lower(default._CN)["loftr"]
    return LoFTR, cfg


def import_multiview_matcher():
    """Return the reference's ``MultiviewMatcher`` class."""
    _ensure_path()
    install_stubs()
    from src.MultiviewMatcher.MultiviewMatcher import MultiviewMatcher
    return MultiviewMatcher


def import_match_table_consumers():
    """Return the reference's own ``Match2Kpts``, ``keypoint_worker``, ``update_matches`` and
    ``transform_keypoints`` (src/coarse_match/utils/merge_kpts.py:19-61, coarse_match_worker.py:151-270).

    ``coarse_match_worker`` imports ray / pytorch_lightning / the dataset stack at module level, none of which
    the three functions use.  Their *unchanged source text* is compiled from the reference file (ast, no
    edits) into a namespace that provides what they reference: numpy, the real ``agg_groupby_2d`` and
    identity stand-ins for ``tqdm`` / ``logger``."""
    import ast
    import numpy as np
    _ensure_path()
    install_stubs()
    mk = importlib.import_module("src.coarse_match.utils.merge_kpts")
    path = os.path.join(REFERENCE_ROOT, "src", "coarse_match", "coarse_match_worker.py")
    with open(path) as fh:
        tree = ast.parse(fh.read())
    wanted = ("keypoint_worker", "update_matches", "transform_keypoints")
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    for n in body:
        n.decorator_list = []
    ns = {"np": np, "agg_groupby_2d": mk.agg_groupby_2d, "tqdm": (lambda it, *a, **k: it),
          "logger": sys.modules["loguru"].logger}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return mk.Match2Kpts, ns["keypoint_worker"], ns["update_matches"], ns["transform_keypoints"]


def _compile_defs(rel_path, names, ns):
    """Compile the UNCHANGED source of the named top-level functions / classes of a reference file into ``ns``."""
    import ast
    path = os.path.join(REFERENCE_ROOT, rel_path)
    with open(path) as fh:
        tree = ast.parse(fh.read())
    body = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert len(body) == len(names), (rel_path, names, [n.name for n in body])
    for n in body:
        n.decorator_list = []
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


def import_matching_data():
    """Return the reference's own ``MatchingMultiviewData`` (with ``FeatureTrackStatus``) and ``UpdatedQueryPts``
    (src/post_optimization/data_construct/construct_matching_data.py:10-476,
    src/post_optimization/matcher_model/multiview_match_worker.py:85-108).

    Their modules import cv2 / ray / pytorch_lightning at module level, none of which these classes use: the classes'
    unchanged source text is compiled (ast, no edits) into a namespace that holds what they reference -- numpy, torch,
    ``Dataset``, ``time``, a no-op ``logger`` and the reference's own ``chunks_balance`` / geometry helpers, compiled
    the same way from src/utils/ray_utils.py and src/post_optimization/utils/geometry_utils.py."""
    import time
    import numpy as np
    import torch
    from torch.utils.data.dataset import Dataset
    _ensure_path()
    install_stubs()
    ns = {"np": np, "torch": torch, "Dataset": Dataset, "time": time.time, "logger": sys.modules["loguru"].logger}
    _compile_defs(os.path.join("src", "utils", "ray_utils.py"), ("chunks_balance",), ns)
    _compile_defs(os.path.join("src", "post_optimization", "utils", "geometry_utils.py"),
                  ("convert_pose2T", "convert_T2pose", "project_point_cloud_to_image", "transform_point_cloud_to_camera"), ns)
    _compile_defs(os.path.join("src", "post_optimization", "data_construct", "construct_matching_data.py"),
                  ("FeatureTrackStatus", "MatchingMultiviewData"), ns)
    ns2 = {"np": np, "torch": torch}
    _compile_defs(os.path.join("src", "post_optimization", "matcher_model", "multiview_match_worker.py"),
                  ("UpdatedQueryPts",), ns2)
    return ns["MatchingMultiviewData"], ns2["UpdatedQueryPts"]


def import_matchformer():
    """Return the reference's ``Matchformer`` class (third_party/MatchFormer/model/matchformer.py) and its lower-cased
    coarse_only config (third_party/MatchFormer/config/matchformer_coarse_only.py values).  ``timm.models.layers``
    (DropPath with p = 0, to_2tuple, trunc_normal_) gets identity-level stand-ins; none of them computes on the
    inference path."""
    import torch
    _ensure_path()
    install_stubs()
    if "timm.models.layers" not in sys.modules:
        tl = _mod("timm.models.layers")

        class DropPath(torch.nn.Module):
            def __init__(self, drop_prob=0.0):
                super().__init__()

            def forward(self, x):
                return x
        tl.DropPath = DropPath
        tl.to_2tuple = lambda x: x if isinstance(x, tuple) else (x, x)
        tl.trunc_normal_ = lambda t, std=1.0: torch.nn.init.trunc_normal_(t, std=std)
        sys.modules["timm.models"].layers = tl
    from third_party.MatchFormer.model.matchformer import Matchformer
    return Matchformer


def import_image_readers(frames):
    """Return the reference's own ``read_grayscale`` / ``read_rgb`` (src/dataset/utils.py:80-177, with the helpers they
    call: process_resize, pad_bottom_right, grayscale2tensor, rgb2tensor, mask2tensor, resize_image), compiled unchanged.

    The module imports cv2 / h5py / albumentations at the top, none installed here.  The readers use cv2 for exactly one
    thing, the decode: ``frames`` maps a path to an already decoded uint8 array ([H,W] gray or [H,W,3] RGB) and the cv2
    stand-in hands it out (as BGR for IMREAD_COLOR, so that the readers' own cvtColor(BGR2RGB) restores it).  Everything
    after the decode -- the PIL LANCZOS resize included -- is the reference's code running on the real Pillow."""
    import numpy as np
    import PIL
    import PIL.Image
    import torch
    _ensure_path()
    install_stubs()

    class _Cv2:
        IMREAD_GRAYSCALE, IMREAD_COLOR, COLOR_BGR2RGB = 0, 1, 4

        @staticmethod
        def imread(path, flag):
            a = frames[path]
            if flag == _Cv2.IMREAD_COLOR:
                assert a.ndim == 3
                return np.ascontiguousarray(a[:, :, ::-1])
            assert a.ndim == 2
            return a

        @staticmethod
        def cvtColor(a, code):
            assert code == _Cv2.COLOR_BGR2RGB
            return np.ascontiguousarray(a[:, :, ::-1])

    ns = {"np": np, "torch": torch, "PIL": PIL, "cv2": _Cv2, "logger": sys.modules["loguru"].logger}
    _compile_defs(os.path.join("src", "dataset", "utils.py"),
                  ("process_resize", "pad_bottom_right", "grayscale2tensor", "rgb2tensor", "mask2tensor", "read_rgb",
                   "read_grayscale", "resize_image"), ns)
    return ns["read_grayscale"], ns["read_rgb"]


def aspanformer_coarse_only_config(match_thr=0.4):
    """Lower-cased ``aspan`` node of third_party/aspantransformer/src/config/default.py:5-51 with the overrides of
    configs/aspan/outdoor/aspan_test_coarse_only.py:7-12 and the threshold override of coarse_match_worker.py:53."""
    _ensure_path()
    install_stubs()
    cfgmod = importlib.import_module("third_party.aspantransformer.src.config.default")
    cfg = cfgmod.get_cfg_defaults()

    def lower(c):
        return {k.lower(): lower(v) for k, v in c.items()} if isinstance(c, dict) else c
    a = lower(cfg)["aspan"]
    a["coarse"].update(coarsest_level=[36, 36], train_res=[832, 832], test_res=[1152, 1152])
    a["match_coarse"].update(match_type="dual_softmax", thr=match_thr, train_coarse_percent=0.3)
    a["fine"]["enable"] = False
    return a


class cpu_cuda_calls:
    """ASpanFormer hard-codes ``.cuda()`` on three small constant tensors inside forward (aspan_module/transformer.py:128,
    attention.py:103, aspanformer.py:129).  This container has no GPU: inside this context ``Tensor.cuda`` is the identity,
    which leaves every value the module computes unchanged."""

    def __enter__(self):
        import torch
        self._saved = torch.Tensor.cuda
        torch.Tensor.cuda = lambda t, *a, **k: t

    def __exit__(self, *exc):
        import torch
        torch.Tensor.cuda = self._saved


def import_aspanformer():
    """Return the reference's ``ASpanFormer`` class (third_party/aspantransformer/src/ASpanFormer/aspanformer.py), imported
    unchanged.  ``torchvision.transforms.Resize`` (online resize of frames whose sides are not multiples of 32,
    aspanformer.py:131-139) is NOT available here: the stand-in restates torchvision 0.9.1's tensor path (bilinear
    ``F.interpolate``, align_corners=False) -- builder-written, like the kornia stand-ins; the fixture case that uses it is
    labelled "resized"."""
    _ensure_path()
    install_stubs()
    tvt = sys.modules["torchvision.transforms"]
    if not hasattr(tvt, "Resize"):
        import torch.nn.functional as F

        class Resize:                      # NOT the reference's code: torchvision 0.9.1 functional_tensor.resize, restated
            def __init__(self, size):
                self.size = list(size)

            def forward(self, img):
                return F.interpolate(img, size=self.size, mode="bilinear", align_corners=False)
        tvt.Resize = Resize
    from third_party.aspantransformer.src.ASpanFormer.aspanformer import ASpanFormer
    return ASpanFormer
