"""detectorfreesfm_amd -- MI355X (gfx950) native dense-matching hot path of Detector-Free SfM.

Two drop-in plugins behind the reference's own Python surfaces, with the hot kernels in
hand-written HIP (``csrc/`` -> ``libdfsfm_hip.so``, C ABI in ``include/dfsfm_hip.h``):

* ``HipLoFTR``             coarse matcher (LoFTR coarse_only)
* ``HipMatchformer``       coarse matcher (MatchFormer-LA large, coarse_only)
* ``HipASpanFormer``       coarse matcher (ASpanFormer, coarse_only, online_resize)
* ``HipMultiviewMatcher``  multiview refinement head
"""
from .aspanformer import HipASpanFormer, aspanformer_coarse_only_config
from .coarse import HipLoFTR
from .matchformer import HipMatchformer, matchformer_coarse_only_config
from .refine import HipMultiviewMatcher
from .config import loftr_coarse_only_config, multiview_refinement_config

__all__ = ["HipLoFTR", "HipASpanFormer", "aspanformer_coarse_only_config", "HipMatchformer", "HipMultiviewMatcher", "matchformer_coarse_only_config", "loftr_coarse_only_config", "multiview_refinement_config"]
