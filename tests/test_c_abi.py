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


def test_isa_gate_pattern_and_stamps():
    """csrc/Makefile's ISA gate (the packed-fp32 op_sel hazard, profiles/r06_fine_match_bisect.txt): its pattern flags the forms the
    reproducer shows misreading beside in-flight MFMAs -- op_sel set for src1 / src2 -- and none of the clean ones, and every
    translation unit with MFMAs went through it in this build (the .isa_ok stamp sits next to its ISA listing)."""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "detectorfreesfm_amd", "csrc")
    mk = open(os.path.join(root, "Makefile")).read()
    m = re.search(r'grep -E "([^"]+)" \$\(ROOT\)/build/\$\*\.s > /dev/null', mk)
    assert m, "the gate's grep is gone from csrc/Makefile"
    pat = re.compile(m.group(1))
    bad = ["v_pk_mul_f32 v[152:153], v[202:203], v[152:153] op_sel:[0,1]",
           "v_pk_add_f32 v[78:79], v[78:79], v[78:79] op_sel:[0,1] op_sel_hi:[1,0]",
           "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0]",
           "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,0,1]",
           "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[1,1,0] op_sel_hi:[0,1,1]"]
    good = ["v_pk_mul_f32 v[138:139], v[84:85], v[86:87] op_sel_hi:[1,0]",
            "v_pk_fma_f32 v[82:83], v[80:81], s[90:91], v[82:83] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]",
            "v_pk_fma_f32 v[0:1], v[2:3], s[4:5], v[0:1] op_sel:[1,0,0]",
            "v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]",
            "v_pk_add_f32 v[154:155], v[154:155], v[202:203]",
            "v_pk_add_f16 v1, v2, v3 op_sel:[0,1]"]
    for line in bad:
        assert pat.search(line), line
    for line in good:
        assert not pat.search(line), line
    srcs = re.search(r"^MFMA_SRCS\s*:=\s*(.+)$", mk, re.M).group(1)
    nos = re.search(r"^NOSLP_SRCS\s*:=\s*(.+)$", mk, re.M).group(1).split()
    units = [u for u in srcs.replace("$(NOSLP_SRCS)", " ".join(nos)).split()]
    assert set(units) >= {"fine_match", "coarse_match", "conv_gemm", "linear_attention", "s2d_front", "encoder_fused", "encoder256"}
    for u in units:
        isa = os.path.join(root, "build", u + ".s")
        assert os.path.exists(os.path.join(root, "build", u + ".isa_ok")) and os.path.exists(isa), u
        text = open(isa).read()
        assert "v_mfma" in text and not pat.search(text), u
    # every translation unit that holds MFMAs is on the list
    with_mfma = {f[:-4] for f in os.listdir(root) if f.endswith(".hip") and re.search(r"__builtin_amdgcn_mfma|v_mfma_f32", open(os.path.join(root, f)).read())}
    with_mfma |= {"conv_gemm", "coarse_match"} if re.search(r"__builtin_amdgcn_mfma|v_mfma_f32", open(os.path.join(root, "sf_gemm.h")).read()) else set()
    assert with_mfma <= set(units), with_mfma - set(units)
