"""What pins the restatements to the reference proper (not to each other): tests/golden/reference_structure.json is
produced by tests/golden/make_reference_fixtures.py from /root/reference itself — `_make_divisible` EXECUTED from the
reference source (deeplabv3p.py:157-164), every structural literal (block tables, OS-dependent strides and rates,
BatchNorm epsilons / momenta, signature defaults, CRF parameters) EXTRACTED from it with `ast`.  The two oracles and the
product's own graph must agree with that file.  (The arithmetic of the layers lives in TensorFlow and stays unpinned —
DESIGN.md §4; the same script regenerates the numeric golden vectors on a box where Keras 2.2.4 / TF 1.13 import.)"""
import json
import os

import numpy as np
import pytest

import dl3_amd  # noqa: F401
from dl3_amd import deeplabv3p as P
from dl3_amd import graph as G
from dl3_amd import subpixel as S
from dl3_amd import utils as U
from oracle import dl3_oracle as O
from oracle import torch_ref as T

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_structure.json")) as f:
    REF = json.load(f)


def test_make_divisible_executed_from_the_reference():
    cases = REF["make_divisible"]["cases"]
    assert len(cases) > 300
    for v, d, want in cases:
        assert O._make_divisible(v, d) == want, (v, d)
        assert P._make_divisible(v, d) == want, (v, d)


def test_mobilenetv2_block_table():
    blocks = REF["inverted_res_blocks"]
    assert [b["block_id"] for b in blocks] == list(range(17))
    assert dict(map(tuple, REF["inverted_res_block_signature"]))["rate"] == 1
    ref = [(b["block_id"], b["filters"], b["stride"], b["expansion"], b["skip_connection"], b["rate"]) for b in blocks]
    assert [tuple(r) for r in O.MNV2_BLOCKS] == ref
    assert [(bid, f, s, e, sk, r) for f, s, e, bid, sk, r in T._MNV2_CALLS] == ref
    assert [(i,) + (f, s, e, sk, r) for i, (f, s, e, sk, r) in enumerate(P._MNV2)] == ref
    assert all(b["alpha"] == {"expr": "alpha"} for b in blocks)


def _os_constants():
    by = {}
    rows = [r for r in REF["os_constants"] if r[1] != "OS"]
    # source order: the `if OS == 8` branch first (deeplabv3p.py:273-277), then the else branch (:278-282)
    for (line, name, val), os_ in zip(rows, [8] * 4 + [16] * 4):
        by.setdefault(os_, {})[name] = val
    return by


def test_signatures_and_defaults():
    sig = dict(map(tuple, REF["Deeplabv3_signature"]))
    import inspect
    got = inspect.signature(P.Deeplabv3).parameters
    assert list(got) == [n for n, _ in REF["Deeplabv3_signature"]]
    for n, v in sig.items():
        d = got[n].default
        assert (tuple(d) if isinstance(d, (tuple, list)) else d) == (tuple(v) if isinstance(v, list) else v), n
    got = inspect.signature(S.Subpixel.__init__).parameters
    for n, v in REF["Subpixel_init_signature"]:
        if v != "<required>" and not isinstance(v, dict) and n in got:
            d = got[n].default
            assert (tuple(d) if isinstance(d, (tuple, list)) else d) == (tuple(v) if isinstance(v, list) else v), n
    assert U.SegModel.epochs == REF["SegModel_class_attrs"]["epochs"]
    assert U.SegModel.batch_size == REF["SegModel_class_attrs"]["batch_size"]
    got = inspect.signature(U.SegModel.create_seg_model).parameters
    assert list(got) == [n for n, _ in REF["create_seg_model_signature"]]
    for n, v in REF["create_seg_model_signature"]:
        if v != "<required>":
            assert got[n].default == v, n
    # dense-CRF hook (utils.py:74-91): the literal parameters of the pydensecrf calls
    crf = {c["fn"]: c["kwargs"] for c in REF["do_crf_calls"]}
    assert crf["unary_from_labels"]["gt_prob"] == U.CRF_PARAMS["gt_prob"]
    assert tuple(crf["addPairwiseGaussian"]["sxy"]) == tuple(U.CRF_PARAMS["gaussian_sxy"])
    assert crf["addPairwiseGaussian"]["compat"] == U.CRF_PARAMS["gaussian_compat"]
    assert crf["addPairwiseBilateral"]["sxy"] == U.CRF_PARAMS["bilateral_sxy"]
    assert crf["addPairwiseBilateral"]["srgb"] == U.CRF_PARAMS["bilateral_srgb"]
    assert crf["addPairwiseBilateral"]["compat"] == U.CRF_PARAMS["bilateral_compat"]
    assert crf["inference"]["args"] == [U.CRF_PARAMS["iterations"]] if "inference" in crf and "args" in crf["inference"] else True


@pytest.mark.parametrize("OS", [8, 16])
def test_xception_graph_against_the_reference_literals(OS):
    k = _os_constants()[OS]
    assert O.XCEPTION_OS[OS] == (k["entry_block3_stride"], k["middle_block_rate"], tuple(k["exit_block_rates"]),
                                 tuple(k["atrous_rates"]))
    assert T._XCEPTION_OS[OS] == O.XCEPTION_OS[OS]
    assert REF["middle_flow_repeats"] == [3, 16]  # range(3) separable convs per block, range(16) middle-flow units
    G.clear_session()
    m = P.Deeplabv3(weights=None, input_shape=(64, 64, 3), classes=21, backbone="xception", OS=OS)
    L = {l.name: l for l in m.layers}
    blocks = {}
    for b in REF["xception_blocks"]:
        prefix = b["args"][2]
        blocks[prefix if isinstance(prefix, str) else "middle_flow_unit"] = b
    rate_of = {"middle_block_rate": k["middle_block_rate"], "exit_block_rates[0]": k["exit_block_rates"][0],
               "exit_block_rates[1]": k["exit_block_rates"][1]}

    def check(prefix, b):
        depth = b["args"][1]
        kw = b["kwargs"]
        stride = kw["stride"] if not isinstance(kw["stride"], dict) else k[kw["stride"]["expr"]]
        rate = kw.get("rate", 1)
        rate = rate_of[rate["expr"]] if isinstance(rate, dict) else rate
        for i in range(3):
            dwl = L["%s_separable_conv%d_depthwise" % (prefix, i + 1)]
            pwl = L["%s_separable_conv%d_pointwise" % (prefix, i + 1)]
            assert dwl.cfg["rate"] == rate and dwl.cfg["stride"] == (stride if i == 2 else 1), dwl.name
            assert pwl.cfg["filters"] == depth[i], pwl.name
            for bn in ("%s_separable_conv%d_depthwise_BN" % (prefix, i + 1), "%s_separable_conv%d_pointwise_BN" % (prefix, i + 1)):
                assert L[bn].cfg["eps"] == 1e-3 and L[bn].cfg["momentum"] == 0.99  # SepConv_BN default epsilon, Keras momentum
        if kw["skip_connection_type"] == "conv":
            sc = L[prefix + "_shortcut"]
            assert sc.cfg["filters"] == depth[-1] and sc.cfg["stride"] == stride and sc.cfg["k"] == 1
        else:
            assert prefix + "_shortcut" not in L

    for prefix, b in blocks.items():
        if prefix == "middle_flow_unit":
            for i in range(16):
                check("middle_flow_unit_%d" % (i + 1), b)
        else:
            check(prefix, b)
    assert dict(map(tuple, REF["sepconv_signature"]))["epsilon"] == 1e-3
    for i, c in enumerate([c for c in REF["sepconv_calls"] if isinstance(c["args"][2], str) and c["args"][2].startswith("aspp")]):
        name = c["args"][2]
        assert L[name + "_depthwise"].cfg["rate"] == k["atrous_rates"][i]
        assert L[name + "_pointwise"].cfg["filters"] == c["args"][1] == 256
        assert L[name + "_depthwise_BN"].cfg["eps"] == c["kwargs"]["epsilon"] == 1e-5
    for c in REF["BatchNormalization_calls"]:
        name = c["kwargs"]["name"]
        if isinstance(name, str) and name in L:
            assert L[name].cfg["eps"] == c["kwargs"].get("epsilon", 1e-3), name
            assert L[name].cfg["momentum"] == c["kwargs"].get("momentum", 0.99), name
    drop = [l for l in m.layers if l.kind == "Dropout"]
    assert len(drop) == 1 and drop[0].cfg["rate"] == REF["Dropout_calls"][0]["args"][0] == 0.1


def test_mobilenetv2_graph_against_the_reference_literals():
    G.clear_session()
    m = P.Deeplabv3(weights=None, input_shape=(64, 64, 3), classes=21, backbone="mobilenetv2", OS=16)
    L = {l.name: l for l in m.layers}
    bn_kw = {c["kwargs"]["name"]["expr"].split("'")[1]: c["kwargs"] for c in REF["BatchNormalization_calls"]
             if isinstance(c["kwargs"]["name"], dict) and c["kwargs"]["name"]["expr"].startswith("prefix + '") and
             "momentum" in c["kwargs"]}
    assert set(bn_kw) == {"expand_BN", "depthwise_BN", "project_BN"}
    for b in REF["inverted_res_blocks"]:
        prefix = "expanded_conv_%d_" % b["block_id"] if b["block_id"] else "expanded_conv_"
        dwl = L[prefix + "depthwise"]
        assert dwl.cfg["stride"] == b["stride"] and dwl.cfg["rate"] == b["rate"], prefix
        assert L[prefix + "project"].cfg["filters"] == P._make_divisible(int(b["filters"] * 1.0), 8)
        assert ((prefix + "add") in L) == b["skip_connection"], prefix
        assert ((prefix + "expand") in L) == (b["block_id"] != 0)
        for suffix, kw in bn_kw.items():
            if prefix + suffix in L:
                assert L[prefix + suffix].cfg["eps"] == kw["epsilon"] and L[prefix + suffix].cfg["momentum"] == kw["momentum"]
    for c in REF["BatchNormalization_calls"]:
        name = c["kwargs"]["name"]
        if isinstance(name, str) and name in L:
            assert L[name].cfg["eps"] == c["kwargs"].get("epsilon", 1e-3), name
            assert L[name].cfg["momentum"] == c["kwargs"].get("momentum", 0.99), name
    # every string literal of the reference that names a layer of this backbone exists in the product graph
    for s in ("Conv", "Conv_BN", "image_pooling", "image_pooling_BN", "aspp0", "aspp0_BN", "aspp0_activation",
              "concat_projection", "concat_projection_BN", "logits_semantic"):
        assert s in REF["string_literals"] and s in L, s
