"""C-ABI drop-in boundary: the library loads and exports every symbol include/dfsfm_hip.h declares.
No compute is launched here (no GPU needed)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dfsfm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dfsfm_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(built_lib):
    from detectorfreesfm_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 14
    bound = {name for name, _, _ in _lib.SIGNATURES}
    assert set(declared) == bound, (set(declared) ^ bound)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert built_lib.dfsfm_version() == 1


def test_argument_checks_do_not_launch(built_lib):
    L = built_lib
    # null pointers -> BADARG, before any HIP call
    assert L.dfsfm_linear_attention_f32(None, None, None, None, 1, None, 1, None, 1, 8, 8, 8, 32, 256, 256, 256,
                                        256, 1e-6, None, None, 0, None, 0, None) == -1
    assert L.dfsfm_coarse_match_f32(None, None, 1, 16, 16, 256, 0.1, 0.2, 2, 4, 4, 4, 4, None, None, 8.0, None,
                                    None, None, None, None, None, None, None, 0, None) == -1
    assert L.dfsfm_roi_align_f32(None, 1, 3, 8, 8, None, None, None, 4, 35, 35, 0.0, None, None, None, 0, None) == -1
    assert L.dfsfm_fine_match_f32(None, None, None, None, 4, 2, 15, 7, 128, None, None, None, None, 0, 0, None,
                                  None, None, None, None, None, None) == -1
    assert L.dfsfm_fine_match_split(None, None, None, None, None, None, 4, 2, 15, 7, 128, None, None, None, None, 0, 0, None,
                                    None, None, None, None, None, None) == -1
    assert L.dfsfm_fine_match_split(None, None, None, None, None, None, 0, 2, 15, 7, 128, None, None, None, None, 0, 0, None,
                                    None, None, None, None, None, None) == 0
    # empty work lists are a no-op success
    assert L.dfsfm_roi_align_f32(None, 1, 3, 8, 8, None, None, None, 0, 35, 35, 0.0, None, None, None, 0, None) == 0


def test_workspace_queries(built_lib):
    L = built_lib
    assert L.dfsfm_linear_attention_workspace(1, 4800, 8, 32) > 0
    assert L.dfsfm_linear_attention_workspace(2000, 900, 8, 16) == ((2000 * 8 * 272 * 4 + 255) // 256) * 256
    assert L.dfsfm_linear_attention_workspace(1, 4800, 8, 64) > 0         # generic-D kernels (MatchFormer stage 4)
    assert L.dfsfm_linear_attention_workspace(1, 4800, 8, 48) == 0        # unsupported head dim
    w1 = L.dfsfm_coarse_match_workspace(1, 4800, 4800)
    w8 = L.dfsfm_coarse_match_workspace(8, 4800, 4800)
    assert 0 < w1 < w8 <= 8 * w1 + 4096


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from detectorfreesfm_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(_lib.DfsfmError):
        _lib.lib()


def test_ops_refuse_cpu_tensors(built_lib):
    import pytest
    import torch
    from detectorfreesfm_amd import _lib, ops
    q = torch.zeros(1, 8, 8, 32)
    with pytest.raises(_lib.DfsfmError):
        ops.linear_attention(q, q, q)
    with pytest.raises(_lib.DfsfmError):
        ops.coarse_match(torch.zeros(1, 16, 256), torch.zeros(1, 16, 256), (4, 4), (4, 4), 0.2, 2, 0.1)
