"""`-m "not gpu"`: the CPU oracle (oracle/bcos_oracle.py) against the golden fixtures recorded from the reference
(tests/golden/make_golden.py).  These pin the oracle; the GPU parity tests then compare the HIP path with it."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import bcos_oracle as O


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


@pytest.fixture(scope="module")
def layers(golden_dir):
    data = np.load(os.path.join(golden_dir, "layers.npz"))
    meta = json.load(open(os.path.join(golden_dir, "layers.json")))
    return data, meta


def _t(data, key):
    return torch.from_numpy(data[key]) if key in data.files else None


def test_oracle_matches_recorded_reference_report(golden_dir):
    """Differences measured between oracle and live reference when the fixtures were generated."""
    rep = json.load(open(os.path.join(golden_dir, "oracle_vs_reference.json")))
    for k, v in rep.items():
        if k.startswith("layer/"):
            assert v["y"][0] <= 1e-6 and v["gx"][0] <= 1e-6, k
    assert rep["r18/oracle_logits"][0] <= 1e-6
    assert rep["r18/oracle_weights"][0] <= 1e-5
    assert rep["r18/oracle_argmax_equal"] is True
    assert rep["r18/completeness_residual_max"] <= 1e-5


def test_conv_layer_cases(layers):
    data, meta = layers
    for c in meta["conv"]:
        n = c["name"]
        x = _t(data, f"{n}/x").requires_grad_(True)
        y = O.bcos_conv2d(x, _t(data, f"{n}/weight"), _t(data, f"{n}/bias"), c["s"], c["p"], c["d"], c["groups"],
                          c["b"], c["max_out"], detach=True, normalize_weight=(c["kind"] == "bcos"))
        (gx,) = torch.autograd.grad(y, x, _t(data, f"{n}/gy"))
        assert rel(y, _t(data, f"{n}/y")) <= 1e-6, n
        assert rel(gx, _t(data, f"{n}/gx")) <= 1e-6, n


def test_linear_layer_cases(layers):
    data, meta = layers
    for c in meta["linear"]:
        n = c["name"]
        x = _t(data, f"{n}/x").requires_grad_(True)
        y = O.bcos_linear(x, _t(data, f"{n}/weight"), _t(data, f"{n}/bias"), c["b"], c["max_out"], detach=True,
                          normalize_weight=(c["kind"] == "bcos"))
        (gx,) = torch.autograd.grad(y, x, _t(data, f"{n}/gy"))
        assert rel(y, _t(data, f"{n}/y")) <= 1e-6, n
        assert rel(gx, _t(data, f"{n}/gx")) <= 1e-6, n


def test_invariants(golden_dir):
    d = np.load(os.path.join(golden_dir, "invariants.npz"))
    x = torch.from_numpy(d["pn/x"])
    assert rel(O.patch_norm(x, 3, 2, 1, 2, 12), d["pn/norm"]) <= 1e-6
    # fast == slow patch norm (reference bcosconv2d.py:233-250), incl. groups
    slow = O.patch_norm_slow(x, (12, 4, 3, 3), (2, 2), (1, 1), (1, 1), 2)
    assert rel(slow, d["pn/norm"]) <= 1e-5
    y = O.bn_uncentered_eval(torch.from_numpy(d["bnu/x"]), torch.from_numpy(d["bnu/running_var"]),
                             torch.from_numpy(d["bnu/weight"]), torch.from_numpy(d["bnu/bias"]))
    assert rel(y, d["bnu/y"]) <= 1e-6
    assert rel(y, d["bnu/y_standard_bn"]) <= 1e-5          # the BnUncV2 fold reproduces the centred BN
    w, b = O.bn_uncentered_fold(torch.from_numpy(d["bnu/src_weight"]), torch.from_numpy(d["bnu/src_bias"]),
                                torch.from_numpy(d["bnu/running_mean"]), torch.from_numpy(d["bnu/running_var"]), 1e-5)
    assert rel(b, d["bnu/bias"]) <= 1e-6
    wb, wa = torch.from_numpy(d["addch/w_before"]), torch.from_numpy(d["addch/w_after"])
    assert torch.equal(wa, torch.cat([wb, -wb], 1) / 2)


def test_localisation_oracle_against_reference_fixture(golden_dir):
    """N2: the oracle's grid-pointing-game arithmetic against the shares produced by the reference's own statements
    (lifted from interpretability/analyses/localisation.py and executed by make_golden.py)."""
    data = np.load(os.path.join(golden_dir, "localisation.npz"))
    rep = json.load(open(os.path.join(golden_dir, "oracle_vs_reference.json")))
    assert rep["loc/oracle_multi_image"][0] == 0.0 and rep["loc/oracle_attributions"][0] <= 1e-6
    att = torch.from_numpy(data["attributions"])
    for smooth, neg in ((0, False), (15, False), (15, True)):
        contribs, metric = O.localisation_fractions(att.clone(), 112, smooth=smooth, neg=neg)
        gold = data[f"fractions_s{smooth}_neg{int(neg)}"]
        assert rel(contribs, gold) <= 1e-6
        assert torch.equal(metric, torch.diagonal(contribs))
    # tensor layout of make_multi_image: image i = a*g + b -> grid row b, column a
    imgs = torch.arange(4.0).view(4, 1, 1, 1).expand(4, 1, 2, 2).contiguous()
    assert O.make_multi_image(imgs)[0, 0].tolist() == [[0, 0, 2, 2], [0, 0, 2, 2], [1, 1, 3, 3], [1, 1, 3, 3]]


def test_oracle_properties():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 8, 7, 7, generator=g)
    w = torch.randn(12, 8, 3, 3, generator=g)
    # scale invariance of unit-norm layers
    y1 = O.bcos_conv2d(x, w, normalize_weight=True, padding=1)
    y2 = O.bcos_conv2d(x, 4.2 * w, normalize_weight=True, padding=1)
    assert rel(y2, y1) <= 1e-6
    # 1x1 conv == linear up to epsilon placement
    w1 = torch.randn(10, 8, 1, 1, generator=g)
    yc = O.bcos_conv2d(x, w1)
    yl = O.bcos_linear(x.permute(0, 2, 3, 1), w1.view(10, 8)).permute(0, 3, 1, 2)
    assert rel(yc, yl) <= 1e-5
    # explanation mode: the output equals the gradient contracted with the input (dynamic linearity, no bias)
    xr = x.clone().requires_grad_(True)
    y = O.bcos_conv2d(xr, w, padding=1, detach=True)
    gy = torch.randn(y.shape, generator=g)
    (gx,) = torch.autograd.grad(y, xr, gy)
    assert abs(float((gx * x).sum() - (gy * y).sum())) <= 1e-3 * float((gy * y).abs().sum())
    # MaxOut takes the max over consecutive filters
    wm = torch.randn(12, 8, 3, 3, generator=g)
    lin = torch.nn.functional.conv2d(x, wm, padding=1)
    ym = O.bcos_conv2d(x, wm, padding=1, max_out=2, b=1)
    assert torch.equal(ym, torch.maximum(lin[:, 0::2], lin[:, 1::2]))


def _golden_net(golden_dir, stem):
    from bcos_hip import synth
    meta = json.load(open(os.path.join(golden_dir, stem + ".json")))
    data = np.load(os.path.join(golden_dir, stem + ".npz"))
    net = synth.build_bcosified_resnet(meta["arch"], seed=meta["weight_seed"])
    record = {k: torch.from_numpy(data["calib/" + k]) for k in meta["calib_order"]}
    synth.apply_calibration(net, record)
    return net, meta, data


def test_resnet18_end_to_end_oracle(golden_dir):
    """Config 1 (B-cosified ResNet-18, 8 x 224^2): product-built weights == reference-built weights, and the oracle
    reproduces the reference's logits / class indices / contribution maps / W(x)."""
    from bcos_hip import synth
    net, meta, data = _golden_net(golden_dir, "resnet18_e2e")
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    for k, (s1, s2) in meta["state_checksum"].items():
        v = sd[k].double()
        assert abs(float(v.sum()) - s1) <= 1e-6 * max(1.0, s2), k
        assert abs(float(v.abs().sum()) - s2) <= 1e-6 * max(1.0, s2), k
    x = synth.synthetic_images(meta["n_images"], seed=meta["image_seed"])
    torch.set_num_threads(8)
    out = O.explain_batch(lambda xx, detach: O.resnet_logits(sd, xx, meta["arch"], detach=detach), x)
    assert rel(out["logits"], data["logits"]) <= 1e-5
    assert np.array_equal(out["prediction"].numpy(), data["prediction"])
    assert rel(out["contribution_map"], data["contribution_map"]) <= 1e-4
    assert rel(out["dynamic_linear_weights"][:2], data["weights_01"]) <= 1e-4
    # completeness: sum (x - mean) W(x) = logit - logit_bias
    mean = torch.tensor(O.IMAGENET_MEAN_ADDINVERSE).view(1, 6, 1, 1)
    lhs = ((x - mean) * out["dynamic_linear_weights"]).sum((1, 2, 3))
    rhs = out["logits"][torch.arange(8), out["prediction"]] + math.log(999)
    assert float((lhs - rhs).abs().max()) <= 1e-4
    rgba = O.gradient_to_image(x[0], torch.from_numpy(data["weights_01"][0]))
    assert float(np.abs(rgba - data["rgba_0"]).max()) <= 1e-5


def test_reference_self_floor_rederived_from_the_oracle(golden_dir):
    """`r50/reference_self_floor_weights` (the reference against itself with oneDNN on / off, make_golden.py:612-618) bounds the
    free-gate maps of the ResNet-50 GPU test at 3 x its value.  The number is not taken on trust from the generator's JSON: the
    oracle under the same two convolution back ends, on whatever host runs this suite, disagrees with itself on W(x) at the same
    level (ReLU gates with ~1e-13 pre-activations open differently under another summation order) while its logits
    and class indices do not move; with oneDNN on it reproduces the recorded W(x)."""
    from bcos_hip import synth
    net, meta, data = _golden_net(golden_dir, "resnet50_small")
    recorded = json.load(open(os.path.join(golden_dir, "oracle_vs_reference.json")))["r50/reference_self_floor_weights"][0]
    x = synth.synthetic_images(4, seed=meta["image_seed"])[:2]
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    fwd = lambda xx, detach: O.resnet_logits(sd, xx, meta["arch"], detach=detach)      # noqa: E731
    a = O.explain_batch(fwd, x)
    with torch.backends.mkldnn.flags(enabled=False):
        b = O.explain_batch(fwd, x)
    assert rel(a["logits"], b["logits"]) <= 1e-4 and torch.equal(a["prediction"], b["prediction"])
    live = rel(a["dynamic_linear_weights"], b["dynamic_linear_weights"])
    assert recorded / 30 <= live <= 30 * recorded, (live, recorded)
    assert min(rel(a["dynamic_linear_weights"], data["weights_01"]), rel(b["dynamic_linear_weights"], data["weights_01"])) <= 3 * recorded
