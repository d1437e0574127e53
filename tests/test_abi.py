"""CPU: the C-ABI library loads without a GPU and exports every symbol include/jyutvoice_hip.h declares; the ctypes
binding lists exactly those symbols; calls fail loudly (no CPU fallback) when no device is present."""
import ctypes
import os
import re

import pytest
import torch

from conftest import REPO

HEADER = os.path.join(REPO, "include", "jyutvoice_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jv_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from jyutvoice_amd import build
    if not os.path.exists(build.LIB):
        build.build(verbose=False)
    from jyutvoice_amd import _lib
    return _lib.load()


def test_header_symbols_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in the header but not exported"


def test_binding_matches_header(lib):
    from jyutvoice_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback(lib):
    from jyutvoice_amd._lib import JvError, check
    h = ctypes.c_void_p()
    rc = lib.jv_create(ctypes.byref(h), 0, 1, 64, 16)
    assert rc != 0 and not h.value
    with pytest.raises(JvError, match="no HIP device|no CPU fallback"):
        check(rc)
    import jyutvoice_amd
    from jyutvoice_amd import synth
    tts, hift = jyutvoice_amd.build_default("cuda:0")
    with pytest.raises(RuntimeError):
        tts.load_state_dict(synth.tts_state_dict())      # needs the GPU: must not silently fall back
    with pytest.raises(RuntimeError, match="no CPU path|GPU only"):
        jyutvoice_amd.build_default("cpu")[0].load_state_dict(synth.tts_state_dict())


def test_product_does_not_import_oracle():
    """the oracle is test infrastructure: nothing under jyutvoice_amd/ (or infer.py) may import it"""
    bad = []
    roots = [os.path.join(REPO, "jyutvoice_amd"), os.path.join(REPO, "infer.py")]
    for root in roots:
        files = [root] if root.endswith(".py") else [os.path.join(d, f) for d, _, fs in os.walk(root) for f in fs if f.endswith(".py")]
        for f in files:
            if re.search(r"^\s*(from|import)\s+oracle\b", open(f).read(), flags=re.M):
                bad.append(f)
    assert not bad, bad


def test_h3_scale_for_bound_host_logic(lib):
    """the power of two the library derives from a proven bound (registry.hip; host only, no device needed): a power of
    two, bound * s <= 60000 < 2 * bound * s within its clamp, and 0 -- 'keep this layer on bf16x6' -- for unusable bounds"""
    import math
    f = lib.jv_h3_scale_for_bound
    for bound in (1e-3, 0.5, 1.0, 21.7, 280.0, 5.9e4, 6.0e4, 6.5e4, 3.0e7, 1e11):
        s = f(bound)
        assert s > 0 and math.log2(s) == int(math.log2(s)), (bound, s)
        assert bound * s <= 60000.0
        if 2.0 ** -24 < s < 2.0 ** 24:
            assert 2.0 * bound * s > 60000.0, (bound, s)
    assert f(21.7) == 2048.0 and f(280.0) == 128.0        # LayerNorm-fed and attention-output bounds of the synthetic weights
    for bad in (0.0, -1.0, float("nan"), float("inf"), 1e31):
        assert f(bad) == 0.0
