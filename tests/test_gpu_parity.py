"""`-m gpu`: the HIP path against the CPU oracle and the golden fixtures, through the C ABI (ctypes).

Tolerances (fp32, north_star: 1e-4 relative, class indices bit-exact):
  * single layers                      relL2 <= 1e-5 (measured ~2e-7 .. 1e-6)
  * logits                             relL2 <= 1e-4 (measured 2e-7 R18, 2e-6 R50), arg-max identical
  * W(x) / contribution maps of ReLU networks: the reference disagrees with ITSELF between CPU back-ends by
    relL2 2.4e-6 (R18) / 1.1e-3 (R50) because ReLU gates with ~0 pre-activation flip under a different summation
    order (tests/golden/oracle_vs_reference.json).  R18 is held to 1e-4; R50 to 3x the reference's
    own floor, plus the completeness identity sum (x - mean) W(x) = logit - bias to 1e-4 relative on OUR output,
    which any wrong (rather than merely re-ordered) gradient would violate.
"""
import ctypes
import json
import math
import os
import warnings

import numpy as np
import pytest
import torch
import torch.nn.functional as F
import torch.nn as nn

from oracle import bcos_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


@pytest.fixture(scope="module")
def lib(hip_lib):
    assert torch.cuda.is_available()
    return hip_lib


# ------------------------------------------------------------------------------------------ single layers
def test_module_layers_against_golden(lib, golden_dir):
    from bcos.modules import BcosConv2d, BcosLinear
    from bcos.modules.bcosifyconv2d import BcosifyConv2d
    from bcos.modules.bcosifylinear import BcosifyLinear
    data = np.load(os.path.join(golden_dir, "layers.npz"))
    meta = json.load(open(os.path.join(golden_dir, "layers.json")))
    warnings.simplefilter("ignore")
    for c in meta["conv"] + meta["linear"]:
        n = c["name"]
        is_conv = "k" in c
        if is_conv:
            cls = BcosConv2d if c["kind"] == "bcos" else BcosifyConv2d
            kw = dict(bias=c["bias"]) if c["kind"] == "bcosify" else {}
            m = cls(c["cin"], c["cout"], c["k"], c["s"], c["p"], c["d"], c["groups"], b=c["b"], max_out=c["max_out"], **kw)
        else:
            cls = BcosLinear if c["kind"] == "bcos" else BcosifyLinear
            kw = dict(bias=c["bias"]) if c["kind"] == "bcosify" else {}
            m = cls(c["cin"], c["cout"], b=c["b"], max_out=c["max_out"], **kw)
        with torch.no_grad():
            m.linear.weight.copy_(torch.from_numpy(data[f"{n}/weight"]))
            if f"{n}/bias" in data.files:
                m.linear.bias.copy_(torch.from_numpy(data[f"{n}/bias"]))
        m = m.to(DEV)
        m.set_explanation_mode(True)
        x = torch.from_numpy(data[f"{n}/x"]).to(DEV).requires_grad_(True)
        y = m(x)
        (gx,) = torch.autograd.grad(y, x, torch.from_numpy(data[f"{n}/gy"]).to(DEV))
        assert rel(y, data[f"{n}/y"]) <= 1e-5, n
        assert rel(gx, data[f"{n}/gx"]) <= 1e-5, n


def test_learnable_b_variants_against_golden(lib, golden_dir):
    """clamping / b_loss variants of the B-cosified layers on the HIP path (BCOS_EPI_FORCE_POW, the B == 1 and B == 2
    shortcuts) against the reference-recorded outputs and input gradients (bcosifyconv2d.py:60-65,78-79,91-98)."""
    from bcos.modules.bcosifyconv2d import BcosifyConv2d
    from bcos.modules.bcosifylinear import BcosifyLinear
    data = np.load(os.path.join(golden_dir, "layer_variants.npz"))
    for c in json.load(open(os.path.join(golden_dir, "layer_variants.json"))):
        n = c["name"]
        if c["layer"] == "conv":
            m = BcosifyConv2d(12, 20, 3, 1, 1, b=2, clamping=c["clamping"], b_loss=c["b_loss"])
        else:
            m = BcosifyLinear(48, 40, b=2, clamping=c["clamping"], b_loss=c["b_loss"])
        with torch.no_grad():
            m.linear.weight.copy_(torch.from_numpy(data[f"{n}/weight"]))
        m = m.to(DEV)
        m.b = torch.tensor(c["b"], device=DEV) if c["clamping"] else c["b"]
        m.set_explanation_mode(True)
        x = torch.from_numpy(data[f"{n}/x"]).to(DEV).requires_grad_(True)
        y = m(x)
        (gx,) = torch.autograd.grad(y, x, torch.from_numpy(data[f"{n}/gy"]).to(DEV))
        assert rel(y, data[f"{n}/y"]) <= 1e-5 and rel(gx, data[f"{n}/gx"]) <= 1e-5, n


def test_training_mode_gradients_against_reference_golden(lib, golden_dir):
    """N4 first slice on the device: input, weight and bias gradients of BcosifyConv2d / BcosifyLinear with the dynamic
    scale differentiated (bcosconv2d.py:176-194 without detach), BatchNormUncentered2d with batch statistics
    (batchnorm_uncentered.py:36-44) -- bcos_train_scale_bwd, bcos_patch_norm_bwd, bcos_conv2d_wgrad (fp32 MFMA),
    bcos_colsum, bcos_channel_axpby -- against gradients recorded from the reference in train mode."""
    from test_host_cpu import run_training_goldens, run_training_goldens2
    run_training_goldens(golden_dir, DEV, 1e-5)
    # second slice: learnable exponent (bcos_train_scale_bwd's bgrad), MaxOut in training mode (bcos_maxout_scatter), native
    # unit-norm layers (bcos_weight_rownorm_bwd)
    run_training_goldens2(golden_dir, DEV, 1e-5)


@pytest.mark.parametrize("path", ["plan", "layers"])
def test_resnet18_training_step_against_reference_golden(lib, golden_dir, path):
    """N4 end to end: B-cosified ResNet-18 in train() mode -- batch statistics in all 20 BatchNormUncentered2d, every
    dynamic scale differentiated, BCE-with-logits loss -- against the loss, input gradient and parameter gradients recorded from
    the reference's training-mode step, through BOTH training paths: `plan` = the engine's training plan (bcos_hip/train_plan.py:
    the whole network one autograd node whose forward / backward walk the engine's layer list), `layers` =
    one autograd node per layer on the nn.Module path (no engine attached)."""
    from bcos_hip import engine, synth
    net, _, _ = _golden_net(golden_dir, "resnet18_e2e")                     # the fixture's calibrated weights
    meta = json.load(open(os.path.join(golden_dir, "resnet18_train_step.json")))
    data = np.load(os.path.join(golden_dir, "resnet18_train_step.npz"))
    x = synth.synthetic_images(4, seed=meta["image_seed"], size=meta["size"]).to(DEV).requires_grad_(True)
    target = F.one_hot(torch.tensor(meta["labels"]), 1000).float().to(DEV)
    if path == "plan":
        engine.attach(net)                                                   # (attached in eval mode: the inference plan folds BatchNorm)
    net.train()
    logits = net(x)
    assert (type(logits.grad_fn).__name__ == "_TrainStepFnBackward") == (path == "plan"), type(logits.grad_fn).__name__
    assert rel(logits, data["logits"]) <= 1e-4
    loss = F.binary_cross_entropy_with_logits(logits, target)
    assert abs(float(loss) - float(data["loss"])) <= 1e-5 * abs(float(data["loss"]))
    named = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
    assert [n for n, _ in named] == meta["param_names"]
    grads = torch.autograd.grad(loss, [x] + [p for _, p in named])
    # ReLU gates with ~1e-13 pre-activations open differently under another summation order: the same floor as
    # the explanation maps applies to gradients that pass through them
    assert rel(grads[0], data["gx"]) <= 2e-3
    norms = torch.stack([g.norm() for g in grads[1:]]).cpu()
    assert float(((norms - torch.from_numpy(data["grad_norms"])).abs() / torch.from_numpy(data["grad_norms"])).max()) <= 2e-3
    names = [n for n, _ in named]
    for key in [k for k in data.files if k.startswith("grad/")]:
        assert rel(grads[1 + names.index(key[5:])], data[key]) <= 2e-3, key
    bufs = dict(net.named_buffers())
    for key in [k for k in data.files if k.startswith("running_var/")]:
        assert rel(bufs[key[12:] + ".running_var"], data[key]) <= 1e-4, key
    if path == "plan":
        # back in eval mode the inference plan re-reads the parameters and statistics the training step left behind
        net.eval()
        with torch.no_grad():
            le = net(x.detach())
        engine.detach(net)
        with torch.no_grad():
            lm = net(x.detach())
        assert rel(le, lm) <= 1e-5


def _check_training_step(*args, **kw):
    import test_host_cpu as H
    return H.check_training_step_fixture(*args, **kw)


@pytest.mark.parametrize("path", ["plan", "layers"])
def test_resnet50_training_step_against_reference_golden(lib, golden_dir, path):
    """the Bottleneck training plan pinned to the REFERENCE -- one train()-mode BCE step of the reference's B-cosified
    ResNet-50 (tests/golden/make_golden.py: resnet50_training_step; the calibrated weights of the resnet50_small fixture, 4 images of
    64 x 64) through the plan and per layer.  Free ReLU gates behind 53 layers: the reference's own step moves by 1.4e-2 of the input
    gradient between fp32 and fp64 (and between one thread and eight); the tolerance is 3 x that recorded floor (_floor_tol), the tight
    check of the Bottleneck plan is test_shallow_bottleneck_resnet_training_step_to_1e4."""
    from bcos_hip import synth
    net, _, _ = _golden_net(golden_dir, "resnet50_small")
    meta = json.load(open(os.path.join(golden_dir, "resnet50_train_step.json")))
    data = np.load(os.path.join(golden_dir, "resnet50_train_step.npz"))
    x = synth.synthetic_images(4, seed=meta["image_seed"], size=meta["size"]).to(DEV)
    target = F.one_hot(torch.tensor(meta["labels"]), 1000).float().to(DEV)
    _check_training_step(net, x, data, meta, path, lambda lg: F.binary_cross_entropy_with_logits(lg, target), out_tol=1e-4, tol=_floor_tol(meta))


def _floor_tol(meta):
    """Gradient tolerance of a deep training fixture: 3 x the distance the REFERENCE's own step moves when it is run in fp64 instead of
    fp32 (recorded at generation, make_golden.py: _training_self_floor -- ReLU gates at rounding level, SURVEY.md H1: 1.4e-2 of the
    input gradient behind ResNet-50's 53 layers, 1e-5 behind the 13 of the shallow fixtures, which are held to 1e-4 instead)."""
    fl = meta["reference_self_floor"]["fp32_vs_fp64"]
    return 3.0 * max(fl["gx"], fl["worst_param"])


@pytest.mark.parametrize("path", ["plan", "layers"])
def test_shallow_bottleneck_resnet_training_step_to_1e4(lib, golden_dir, path):
    """The tight Bottleneck variant: one Bottleneck per stage (every block with its downsample shortcut), recorded from the reference in
    train() mode; no gate floor (reference fp32 vs fp64: 1e-5), so EVERY parameter gradient, the input gradient and every running_var
    are held to 1e-4 through the plan and per layer."""
    from bcos_hip import synth
    meta = json.load(open(os.path.join(golden_dir, "resnet14b_train_step.json")))
    data = np.load(os.path.join(golden_dir, "resnet14b_train_step.npz"))
    assert max(meta["reference_self_floor"]["fp32_vs_fp64"].values()) <= 2e-5
    net = synth.build_bcosified_resnet("resnet14b", seed=meta["weight_seed"])
    synth.apply_calibration(net, {k: torch.from_numpy(data["calib/" + k]) for k in meta["calib_order"]})
    x = synth.synthetic_images(4, seed=meta["image_seed"], size=meta["size"]).to(DEV)
    target = F.one_hot(torch.tensor(meta["labels"]), 1000).float().to(DEV)
    _check_training_step(net.to(DEV), x, data, meta, path, lambda lg: F.binary_cross_entropy_with_logits(lg, target), out_tol=1e-5, tol=1e-4,
                         rv_tol=1e-5)


def _clip_loss(emb):
    return (emb * torch.linspace(-1, 1, emb.shape[1], device=emb.device)).sum() / emb.shape[0]


@pytest.mark.parametrize("path", ["plan", "layers"])
def test_clip_training_step_against_reference_golden(lib, golden_dir, path):
    """... and CLIP's ModifiedResNet (CLIP/clip/model.py:10-55, 94-154 through bcosify.py:74-114): the RN50 tower with the weights of the
    clip_rn50 fixture, one train()-mode step recorded from the reference -- three-convolution stem, anti-aliasing pools inside the plan,
    the attention-pool head under autograd behind it."""
    from bcos_hip import synth
    net, _, _ = _golden_clip(golden_dir)
    meta = json.load(open(os.path.join(golden_dir, "clip_rn50_train_step.json")))
    data = np.load(os.path.join(golden_dir, "clip_rn50_train_step.npz"))
    x = synth.synthetic_images(4, seed=meta["image_seed"], size=meta["size"]).to(DEV)
    _check_training_step(net, x, data, meta, path, _clip_loss, out_tol=1e-4, tol=_floor_tol(meta))


@pytest.mark.parametrize("path", ["plan", "layers"])
def test_small_clip_tower_training_step_to_1e4(lib, golden_dir, path):
    """The tight variant the free-gate floor cannot hide behind: a ModifiedResNet of layers (1, 1, 1, 1), width 16 -- a few thousand ReLU
    decisions, none of them near a tie -- recorded from the reference in train() mode; EVERY parameter gradient, the input gradient and
    every running_var to 1e-4: a wrong anti-aliasing-pool or shortcut gradient in the plan shows at full size."""
    from bcos_hip import synth
    meta = json.load(open(os.path.join(golden_dir, "clip_tiny_train_step.json")))
    data = np.load(os.path.join(golden_dir, "clip_tiny_train_step.npz"))
    net = synth.build_bcosified_clip_resnet(meta["layers"], meta["output_dim"], meta["heads"], meta["width"], seed=meta["weight_seed"])
    synth.apply_calibration(net, {k: torch.from_numpy(data["calib/" + k]) for k in meta["calib_order"]})
    x = synth.synthetic_images(4, seed=meta["image_seed"], size=meta["size"]).to(DEV)
    _check_training_step(net.to(DEV), x, data, meta, path, _clip_loss, out_tol=1e-5, tol=1e-4, rv_tol=1e-5)


def test_training_plan_frozen_batchnorm_and_maxout_refusal_on_device(lib):
    """ADVICE r04 (high / medium), on the device: frozen BatchNorms are normalised with their running variance by the plan and their
    buffers do not move; a fused MaxOut node is refused by the plan before any buffer has been touched (the check of
    tests/test_host_cpu.py without the emulated kernels; gradients behind free ReLU gates: the training fixtures' 2e-3)."""
    import test_host_cpu as H
    H.check_frozen_batchnorm_and_maxout("cuda", tol_out=1e-4, tol=2e-3)


@pytest.mark.parametrize("with_addend", [False, True])
def test_patch_norm_term_in_the_input_gradient_epilogue(lib, with_addend):
    """bcos_epilogue.rowadd (ABI v9): the input-gradient launch of a pointwise layer adds x[pixel] * r[pixel] (+ the shortcut's gradient)
    itself.  Same bits as bcos_patch_norm_bwd_add followed by a plain addend (acc + (x r + addend) in both), on a launch with ragged row
    tiles; a strided layer and a launch the library cannot specialise keep the separate pass through the same entry point."""
    from bcos_hip import ops
    torch.manual_seed(3)
    dev = "cuda"
    for (N, H, Cin, Cout) in [(3, 14, 256, 64), (2, 9, 64, 256), (2, 7, 512, 128)]:
        w = (torch.randn(Cout, Cin, 1, 1) / Cin ** 0.5).to(dev)
        x = torch.randn(N, H, H, Cin, device=dev)
        glin = ops.ensure_absmax(torch.randn(N, H, H, Cout, device=dev))
        rnorm = torch.randn(N * H * H, device=dev) * 0.1
        extra = torch.randn(N, H, H, Cin, device=dev) if with_addend else None
        plan = ops.DgradPlan(w, (1, 1), (0, 0))
        assert plan.pointwise
        fused = plan.run_with_patch_norm(glin, x, rnorm, Cin, H, H, addend=extra)
        apart = plan.run(glin, H, H, addend=ops.patch_norm_bwd(x, rnorm.view(N, H, H), Cin, (1, 1), (1, 1), (0, 0), (1, 1), addend=extra))
        want = torch.einsum("nhwo,oc->nhwc", glin.double(), w.view(Cout, Cin).double()) + x.double() * rnorm.view(N, H, H, 1).double()
        if extra is not None:
            want = want + extra.double()
        assert rel(apart, want) <= 1e-5
        assert torch.equal(fused, apart), float((fused - apart).abs().max())
    # a strided layer: the entry point takes the separate pass (same result as before)
    w = (torch.randn(64, 32, 3, 3) / 17.0).to(dev)
    x = torch.randn(2, 12, 12, 32, device=dev)
    glin = ops.ensure_absmax(torch.randn(2, 6, 6, 64, device=dev))
    rnorm = torch.randn(2 * 6 * 6, device=dev) * 0.1
    plan = ops.DgradPlan(w, (2, 2), (1, 1))
    assert not plan.pointwise
    a = plan.run_with_patch_norm(glin, x, rnorm, 32, 12, 12)
    b = plan.run(glin, 12, 12, addend=ops.patch_norm_bwd(x, rnorm.view(2, 6, 6), 32, (3, 3), (2, 2), (1, 1), (1, 1)))
    assert torch.equal(a, b)
    # the library refuses what it cannot fuse, with BCOS_E_NOSUP (the general epilogue: forced here by the test switch)
    from bcos_hip import lib as L
    w = (torch.randn(64, 64, 1, 1) / 8.0).to(dev)
    x = torch.randn(2, 8, 8, 64, device=dev)
    glin = ops.ensure_absmax(torch.randn(2, 8, 8, 64, device=dev))
    rnorm = torch.randn(128, device=dev)
    plan = ops.DgradPlan(w, (1, 1), (0, 0))
    L.set_option("epi_generic", 1)
    try:
        with pytest.raises(L.BcosHipError) as ei:
            plan.run(glin, 8, 8, rowadd=x, rowadd_scale=rnorm)
        assert ei.value.code == L.BCOS_E_NOSUP
        c = plan.run_with_patch_norm(glin, x, rnorm, 64, 8, 8)        # ... and the entry point falls back
    finally:
        L.set_option("epi_generic", 0)
    d = plan.run_with_patch_norm(glin, x, rnorm, 64, 8, 8)
    assert rel(c, d) <= 1e-6


@pytest.mark.parametrize("C", [64, 128, 256])
def test_patch_norm_term_of_a_3x3_layer_has_one_summation_order(lib, C):
    """bcos_patch_norm_bwd(_add): out = x * (sum of r over the patches that contain the pixel) + addend.  The one-wave-per-pixel kernel and the
    4 / 2-pixels-per-wave kernel of the 64- / 128-channel layers (round 6) add the nine taps in ONE order -- the butterfly
    [((r0 + r8) + r4) + (r2 + r6)] + [(r1 + r5) + (r3 + r7)], tap t = 3 th + tw -- so a layer's bits do not depend on its width."""
    from bcos_hip import ops
    torch.manual_seed(C)
    N, H, W = 2, 13, 11
    x = torch.randn(N, H, W, C, device="cuda")
    r = torch.randn(N, H, W, device="cuda")
    add = torch.randn(N, H, W, C, device="cuda")
    got = ops.patch_norm_bwd(x, r, C, (3, 3), (1, 1), (1, 1), (1, 1), addend=add)
    rp = F.pad(r, (1, 1, 1, 1))                                      # tap (th, tw) of pixel (h, w) reads r[h + 1 - th, w + 1 - tw]
    tap = [rp[:, 2 - th:2 - th + H, 2 - tw:2 - tw + W] for th in range(3) for tw in range(3)]
    t = (((tap[0] + tap[8]) + tap[4]) + (tap[2] + tap[6])) + ((tap[1] + tap[5]) + (tap[3] + tap[7]))
    want = x * t[..., None] + add
    assert torch.equal(got, want), float((got - want).abs().max())
    want64 = x.double() * sum(tp.double() for tp in tap)[..., None] + add.double()
    assert rel(got, want64) <= 1e-6


def test_training_plan_with_batched_weights_follows_every_parameter_change(lib, golden_dir):
    """The training plan keeps its weight banks, their images and its input-gradient plans across steps and refreshes them from the
    parameters by one call per step (train_plan.ResNetTrainPlan._weights).  Whatever happens to a parameter between two steps -- an
    in-place update, an optimizer step, a replaced .data -- the step's loss and every gradient equal, bit for bit, those of the same
    network prepared per layer (BCOS_TRAIN_WEIGHT_BATCH=0: fresh banks in front of every launch)."""
    import copy
    from bcos_hip import engine, synth, train_plan
    net, _, _ = _golden_net(golden_dir, "resnet18_e2e")
    ref = copy.deepcopy(net)
    engine.attach(net); engine.attach(ref)
    net.train(); ref.train()
    x = synth.synthetic_images(4, seed=7, size=64).to(DEV)
    target = F.one_hot(torch.tensor([1, 5, 9, 700]), 1000).float().to(DEV)

    def step(model, batched):
        prev = train_plan._WEIGHT_BATCH
        train_plan._WEIGHT_BATCH = batched
        try:
            xs = x.clone().requires_grad_(True)
            loss = F.binary_cross_entropy_with_logits(model(xs), target)
            ps = [p for p in model.parameters() if p.requires_grad]
            return loss.detach(), torch.autograd.grad(loss, [xs] + ps)
        finally:
            train_plan._WEIGHT_BATCH = prev

    def same():
        la, ga = step(net, True)
        lb, gb = step(ref, False)
        assert net._bcos_engine._train_plan._wbatch is not None and ref._bcos_engine._train_plan._wbatch is None
        assert torch.equal(la, lb)
        for a, b in zip(ga, gb):
            assert torch.equal(a, b)

    same()
    convs_a = [m for m in net.modules() if isinstance(m, nn.Conv2d)]
    convs_b = [m for m in ref.modules() if isinstance(m, nn.Conv2d)]
    with torch.no_grad():                                  # in-place updates (what an optimizer step is)
        for ma, mb in zip(convs_a, convs_b):
            ma.weight.mul_(0.97).add_(1e-3); mb.weight.mul_(0.97).add_(1e-3)
    same()
    with torch.no_grad():                                  # a parameter whose storage is replaced (load of a checkpoint by assignment)
        for ma, mb in zip(convs_a[3:6], convs_b[3:6]):
            new = torch.randn_like(ma.weight) * ma.weight.std()
            ma.weight.data = new.clone(); mb.weight.data = new.clone()
    same()
    opt_a = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9); opt_b = torch.optim.SGD(ref.parameters(), lr=1e-3, momentum=0.9)
    for opt, model, batched in ((opt_a, net, True), (opt_b, ref, False)):
        prev = train_plan._WEIGHT_BATCH
        train_plan._WEIGHT_BATCH = batched
        try:
            for _ in range(2):
                opt.zero_grad(set_to_none=True)
                F.binary_cross_entropy_with_logits(model(x), target).backward()
                opt.step()
        finally:
            train_plan._WEIGHT_BATCH = prev
    for pa, pb in zip(net.parameters(), ref.parameters()):
        assert torch.equal(pa, pb)
    same()


def test_weight_banks_and_images_of_many_layers_from_one_launch(lib):
    """bcos_weight_prep_batch (ABI v9; ops.WeightPrepBatch): the forward bank and the input-gradient banks of a set of layers, and their
    f16x2 images, are what the per-layer preparation makes -- layout copy / flip / tap selection, bcos_split_weights_f16x2_conv -- bit
    for bit; a second run() follows updated parameters; the plan over the banks gives the input gradient of the plan built per layer."""
    from bcos_hip import ops
    torch.manual_seed(11)
    dev = "cuda"
    shapes = [((64, 64, 1, 1), (1, 1), (0, 0)), ((64, 64, 3, 3), (1, 1), (1, 1)), ((128, 64, 3, 3), (2, 2), (1, 1)), ((256, 128, 1, 1), (2, 2), (0, 0)),
              ((1000, 512, 1, 1), (1, 1), (0, 0)), ((64, 6, 7, 7), (2, 2), (3, 3)), ((30, 20, 3, 3), (1, 1), (1, 1))]
    params = [torch.nn.Parameter((torch.randn(s) * (0.5 + i)).to(dev)) for i, (s, _, _) in enumerate(shapes)]
    batch = ops.WeightPrepBatch(dev)
    made = []
    for p, (s, stride, padding) in zip(params, shapes):
        made.append((batch.add_forward(p), batch.add_dgrad(p, stride, padding) if s[1] > 8 else None))

    def check():
        for p, (s, stride, padding), (wk, dplan) in zip(params, shapes, made):
            Cout, Cin, kh, kw = s
            w = p.detach()
            ref_bank = F.pad(w.permute(0, 2, 3, 1), (0, (-Cin) % 4)).contiguous()
            assert torch.equal(wk, ref_bank)
            taps = kh * kw
            img = getattr(wk, f"_bcos_wt2_t{taps}")[1]
            assert torch.equal(img, ops.split_weights_f16x2(ref_bank, taps)), (s, "forward image")
            if dplan is None:
                continue
            r = (-Cout) % 4
            wq = w if not r else torch.cat([w, w.new_zeros((r,) + tuple(w.shape[1:]))], 0)
            ref = ops.DgradPlan(wq, stride, padding)
            assert len(ref.classes) == len(dplan.classes) and ref.has_empty == dplan.has_empty and ref.Cout == dplan.Cout and ref.Cin == dplan.Cin
            for a, b in zip(ref.classes, dplan.classes):
                assert a[:8] == b[:8]
                if a[8] is None:
                    assert b[8] is None
                    continue
                assert torch.equal(a[8], b[8]), (s, "class bank")
                t = a[2] * a[3]
                assert torch.equal(getattr(b[8], f"_bcos_wt2_t{t}")[1], ops.split_weights_f16x2(a[8], t)), (s, "class image")
            Ho = (16 + 2 * padding[0] - kh) // stride[0] + 1
            g = ops.ensure_absmax(torch.randn(2, Ho, Ho, wq.shape[0], device=dev))
            assert torch.equal(ref.run(g, 16, 16), dplan.run(g, 16, 16))
    batch.run()
    check()
    with torch.no_grad():
        for p in params:
            p.mul_(1.7).add_(0.01)
    batch.run()
    check()
    with torch.no_grad():                                     # a parameter whose storage is replaced: the batch follows it
        params[1].data = torch.randn_like(params[1]) * 3.0
    batch.run()
    check()


def test_image_range_by_several_workgroups_per_image(lib):
    """bcos_image_absrange_c (ABI v9): the per-image range of a per-pixel maxima tensor from several workgroups per image -- maxima as
    bcos_image_absrange gives them, minima over the nonzero pixels complemented (the form of bcos_epilogue.out_imgmin_c); images without a
    nonzero pixel keep 0 / 0."""
    from bcos_hip import lib as L
    torch.manual_seed(5)
    for (n, hw) in [(3, 5000), (64, 12544), (2, 50176), (5, 100)]:
        am = torch.randint(1, 2 ** 30, (n, hw), dtype=torch.int32, device="cuda")
        am[:, ::7] = 0
        am[0] = 0                                              # an all-zero image
        ref = torch.empty(2, n, dtype=torch.int32, device="cuda")
        L.check(lib.bcos_image_absrange(am.data_ptr(), ref[0].data_ptr(), ref[1].data_ptr(), n, hw, None), "bcos_image_absrange")
        got = torch.zeros(2, n, dtype=torch.int32, device="cuda")
        L.check(lib.bcos_image_absrange_c(am.data_ptr(), got[0].data_ptr(), got[1].data_ptr(), n, hw, None), "bcos_image_absrange_c")
        torch.cuda.synchronize()
        assert torch.equal(got[0], ref[0])
        want_c = torch.where(ref[1] == -1, torch.zeros_like(ref[1]), ~ref[1])      # (0xffffffff: no nonzero pixel)
        assert torch.equal(got[1], want_c)


def test_wgrad_kernel_on_resnet_shapes(lib):
    """The weight gradient at real layer sizes against fp64 autograd: bcos_conv2d_wgrad_ordered (round 6: bf16 planes split once at
    staging, the pixel chunks' partial tiles added in a fixed order -- bit-identical from call to call, into a buffer that was NOT zeroed)
    and bcos_conv2d_wgrad (fp32 atomics) on the same operands; ragged widths (Cout = 1000, 6 of 8 input channels: the fused-tap stem),
    a single pixel chunk (stored straight into gw), dilation, 1 x 1 over many pixels."""
    from bcos_hip import ops
    g = torch.Generator().manual_seed(9)
    cases = [(8, 64, 56, 64, 3, 1, 1, 1), (8, 256, 14, 1024, 1, 1, 0, 1), (4, 8, 64, 64, 7, 2, 3, 1), (8, 128, 28, 128, 3, 2, 1, 1),
             (4, 2048, 2, 1000, 1, 1, 0, 1), (2, 6, 32, 64, 7, 2, 3, 1), (1, 16, 5, 12, 3, 1, 2, 2), (64, 64, 56, 256, 1, 1, 0, 1), (3, 36, 9, 20, 3, 1, 1, 1)]
    for (N, Cin, H, Cout, k, s, p, d) in cases:
        x = torch.randn(N, Cin, H, H, generator=g)
        Ho = (H + 2 * p - d * (k - 1) - 1) // s + 1
        gl = torch.randn(N, Cout, Ho, Ho, generator=g)
        w = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
        (ref,) = torch.autograd.grad(F.conv2d(x.double(), w, None, s, p, d), w, gl.double())
        xp = x.permute(0, 2, 3, 1).contiguous()
        if Cin % 4:
            xp = F.pad(xp, (0, (-Cin) % 4))                       # (the network input: 6 channels in a pitch of 8)
        glp = gl.permute(0, 2, 3, 1).contiguous()
        if Cout % 4:
            glp = F.pad(glp, (0, (-Cout) % 4))
        xd, gd = xp.to(DEV), glp.to(DEV)
        assert ops.wgrad_is_ordered()
        dirty = torch.full((Cout, k, k, Cin), float("nan"), device=DEV)
        gw = ops.conv2d_wgrad(gd, xd, Cin, Cout, (k, k), (s, s), (p, p), (d, d), out=dirty)
        assert rel(gw.permute(0, 3, 1, 2), ref) <= 2e-6, (N, Cin, H, Cout, k, s, p, d, rel(gw.permute(0, 3, 1, 2), ref))
        again = ops.conv2d_wgrad(gd, xd, Cin, Cout, (k, k), (s, s), (p, p), (d, d))
        assert torch.equal(gw, again), (N, Cin, H, Cout, k, s, p)
        ops.WGRAD_ORDERED = False
        try:
            old = ops.conv2d_wgrad(gd, xd, Cin, Cout, (k, k), (s, s), (p, p), (d, d))
        finally:
            ops.WGRAD_ORDERED = True
        assert rel(old.permute(0, 3, 1, 2), ref) <= 2e-6 and rel(gw, old) <= 2e-6


CONV_GEOMS = [  # N, Cin, H, W, Cout, k, s, p   (the distinct R18/R50 geometry classes at reduced size + ragged edges)
    (2, 64, 14, 14, 64, 1, 1, 0), (2, 64, 14, 14, 256, 1, 1, 0), (2, 256, 14, 14, 64, 1, 1, 0),
    (2, 64, 14, 14, 64, 3, 1, 1), (2, 128, 14, 14, 128, 3, 2, 1), (2, 256, 14, 14, 512, 1, 2, 0),
    (2, 8, 32, 32, 64, 7, 2, 3), (3, 512, 7, 7, 1000, 1, 1, 0), (1, 32, 5, 3, 36, 3, 1, 1),
    (1, 4, 1, 1, 4, 1, 1, 0), (5, 12, 9, 11, 20, 3, 2, 1), (2, 2048, 7, 7, 512, 1, 1, 0),
    # ResNet-50's true geometries at N = 4: 64 -> 256 @56^2, 512 -> 2048 @7^2, 3x3 256 -> 256 @14^2, 3x3 / 2 128 -> 128 @56^2
    (4, 64, 56, 56, 256, 1, 1, 0), (4, 512, 7, 7, 2048, 1, 1, 0), (4, 256, 14, 14, 256, 3, 1, 1), (4, 128, 56, 56, 128, 3, 2, 1),
]


def _canary(shape, extra=64):
    """flat buffer with NaN guards on both sides of the payload"""
    n = int(np.prod(shape))
    buf = torch.full((n + 2 * extra,), float("nan"), device=DEV)
    return buf, buf[extra:extra + n].view(shape), extra


@pytest.mark.parametrize("geom", CONV_GEOMS)
def test_c_abi_conv2d_fwd_and_dgrad(lib, geom):
    """bcos_conv2d_fwd / bcos_tapconv dgrad through raw pointers, guards around every output buffer."""
    from bcos_hip import ops
    N, Cin, H, W, Cout, k, s, p = geom
    g = torch.Generator().manual_seed(sum(geom))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    xr = x.clone().requires_grad_(True)
    y_ref, s_ref = O.bcos_conv2d(xr, w, stride=s, padding=p, detach=True, return_scale=True)
    gy = torch.randn(y_ref.shape, generator=g)
    (gx_ref,) = torch.autograd.grad(y_ref, xr, gy)
    Ho, Wo = y_ref.shape[2:]
    xh = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wk = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    ybuf, y, e = _canary((N, Ho, Wo, Cout))
    sbuf, sc, _ = _canary((N, Ho, Wo, Cout))
    nbuf, nrm, _ = _canary((N, Ho, Wo))
    code = lib.bcos_conv2d_fwd(xh.data_ptr(), wk.data_ptr(), None, y.data_ptr(), sc.data_ptr(), nrm.data_ptr(),
                               N, Cin, H, W, Cout, k, k, s, s, p, p, 1, 1, 2.0, None)
    assert code == 0, lib.bcos_last_error_string()
    torch.cuda.synchronize()
    for b in (ybuf, sbuf, nbuf):
        assert torch.isnan(b[:e]).all() and torch.isnan(b[-e:]).all(), "out-of-bounds write"
    assert rel(y.permute(0, 3, 1, 2), y_ref) <= 1e-5
    assert rel(sc.permute(0, 3, 1, 2), s_ref.expand_as(y_ref)) <= 1e-5
    assert rel(nrm, O.patch_norm(x, k, s, p)[:, 0]) <= 1e-5
    glin = ops.mul(gy.permute(0, 2, 3, 1).contiguous().to(DEV), sc.contiguous())
    gbuf, gx, _ = _canary((N, H, W, Cin))
    plan = ops.DgradPlan(w.to(DEV), (s, s), (p, p))
    if plan.has_empty:
        gx.zero_()
    plan.run(glin, H, W, out=gx)
    torch.cuda.synchronize()
    assert torch.isnan(gbuf[:e]).all() and torch.isnan(gbuf[-e:]).all()
    assert rel(gx.permute(0, 3, 1, 2), gx_ref) <= 1e-5
    if s == 1:
        wT = w.flip(2, 3).permute(1, 2, 3, 0).contiguous().to(DEV)
        gx2 = torch.empty((N, H, W, Cin), device=DEV)
        assert lib.bcos_conv2d_dgrad_s1(glin.data_ptr(), wT.data_ptr(), gx2.data_ptr(), N, Cin, H, W, Cout, k, k, p, p, None) == 0
        assert rel(gx2.permute(0, 3, 1, 2), gx_ref) <= 1e-5


@pytest.mark.parametrize("rows,cin,cout", [(392, 192, 768), (392, 768, 192), (7, 48, 40), (1, 4, 4), (1000, 1536, 192), (300, 192, 1000)])
def test_c_abi_linear_fwd_and_dgrad(lib, rows, cin, cout):
    g = torch.Generator().manual_seed(rows + cin)
    x = torch.randn(rows, cin, generator=g)
    w = torch.randn(cout, cin, generator=g) / math.sqrt(cin)
    xr = x.clone().requires_grad_(True)
    y_ref, s_ref = O.bcos_linear(xr, w, detach=True, return_scale=True)
    gy = torch.randn(y_ref.shape, generator=g)
    (gx_ref,) = torch.autograd.grad(y_ref, xr, gy)
    xd, wd = x.to(DEV), w.to(DEV)
    y = torch.empty((rows, cout), device=DEV)
    sc = torch.empty_like(y)
    assert lib.bcos_linear_fwd(xd.data_ptr(), wd.data_ptr(), None, y.data_ptr(), sc.data_ptr(), None, rows, cin, cout, 2.0, None) == 0
    assert rel(y, y_ref) <= 1e-5 and rel(sc, s_ref.expand_as(y_ref)) <= 1e-5
    glin = (gy.to(DEV) * sc).contiguous()
    gx = torch.empty((rows, cin), device=DEV)
    wT = w.t().contiguous().to(DEV)
    assert lib.bcos_linear_dgrad(glin.data_ptr(), wT.data_ptr(), gx.data_ptr(), rows, cin, cout, None) == 0
    assert rel(gx, gx_ref) <= 1e-5


def test_general_b_and_bias_and_epilogue(lib):
    """B != 2 (pow path), bias, fused BN scale/shift + residual + ReLU epilogue, scale_out semantics."""
    from bcos_hip import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 16, 9, 9, generator=g)
    w = torch.randn(24, 16, 3, 3, generator=g) / 12
    bias = torch.randn(24, generator=g) * 0.1
    csc, csh = torch.rand(24, generator=g) + 0.5, torch.randn(24, generator=g) * 0.1
    res = torch.randn(2, 24, 9, 9, generator=g)
    for b in (2.0, 1.5, 2.5):
        xr = x.clone().requires_grad_(True)
        yb, sb = O.bcos_conv2d(xr, w, bias, padding=1, b=b, detach=True, return_scale=True)
        z = torch.relu(yb * csc.view(1, -1, 1, 1) + csh.view(1, -1, 1, 1) + res)
        gz = torch.randn(z.shape, generator=g)
        lin = F.conv2d(x, w, bias, padding=1)
        (glin_ref,) = torch.autograd.grad(z, yb, gz, retain_graph=True)
        y, t, _ = ops.conv2d_fwd(x.permute(0, 2, 3, 1).contiguous().to(DEV), w.permute(0, 2, 3, 1).contiguous().to(DEV),
                                 padding=(1, 1), bias=bias.to(DEV), b=b, ch_scale=csc.to(DEV), ch_shift=csh.to(DEV),
                                 addend=res.permute(0, 2, 3, 1).contiguous().to(DEV), relu=True, want_scale=True)
        assert rel(y.permute(0, 3, 1, 2), z) <= 1e-5, b
        # t = d z / d lin with the scale detached = s * bn_scale * gate
        t_ref = sb * csc.view(1, -1, 1, 1) * (z > 0)
        assert rel(t.permute(0, 3, 1, 2), t_ref) <= 1e-5, b


# ------------------------------------------------------------------------------------------ helper kernels
def test_streaming_kernels(lib):
    from bcos_hip import ops
    g = torch.Generator().manual_seed(9)
    x = torch.rand(3, 6, 20, 18, generator=g)
    mean, std = torch.tensor(O.IMAGENET_MEAN_ADDINVERSE), torch.tensor(O.IMAGENET_STD_ADDINVERSE)
    xn = ops.prep_input(x.to(DEV), mean.to(DEV), std.to(DEV), cpad=8)
    assert rel(xn[..., :6].permute(0, 3, 1, 2), O.normalize6(x, mean.tolist(), std.tolist())) <= 1e-7
    assert float(xn[..., 6:].abs().max()) == 0.0
    x3 = x[:, :3].contiguous()
    xn3 = ops.prep_input(x3.to(DEV), mean.to(DEV), std.to(DEV), cpad=8, add_inverse=True, want_absmax=True)
    assert rel(xn3[..., :6].permute(0, 3, 1, 2), O.normalize6(O.add_inverse(x3), mean.tolist(), std.tolist())) <= 1e-7
    from bcos_hip import lib as blib
    if blib.get_contraction_mode() == "f16x2":          # the per-pixel maxima emitted alongside are exact
        assert torch.equal(ops.absmax_of(xn3), xn3.abs().amax(dim=-1).reshape(-1).view(torch.int32))
    gxn = torch.randn(3, 20, 18, 8, generator=g)
    wts, contrib = ops.finalize_explanation(gxn.to(DEV), x.to(DEV), std.to(DEV))
    w_ref = gxn[..., :6].permute(0, 3, 1, 2) / std.view(1, 6, 1, 1)
    assert rel(wts, w_ref) <= 1e-7 and rel(contrib, (x * w_ref).sum(1)) <= 1e-6
    assert rel(ops.contrib_map(x.to(DEV), w_ref.contiguous().to(DEV)), (x * w_ref).sum(1)) <= 1e-6
    a = torch.randn(2, 64, 13, 12, generator=g)
    for (k, s, p) in ((3, 2, 1), (2, 2, 0), (3, 1, 1)):
        ar = a.clone().requires_grad_(True)
        pr = F.avg_pool2d(ar, k, s, p)
        gp = torch.randn(pr.shape, generator=g)
        (ga,) = torch.autograd.grad(pr, ar, gp)
        y = ops.avgpool2d_fwd(a.permute(0, 2, 3, 1).contiguous().to(DEV), k, s, p, want_absmax=True)
        assert rel(y.permute(0, 3, 1, 2), pr) <= 1e-6
        if blib.get_contraction_mode() == "f16x2":       # (k = 3 / 2: emitted by the pool's own launch; k = 3, s = 1 too; always exact)
            assert torch.equal(ops.absmax_of(y), y.abs().amax(dim=-1).reshape(-1).view(torch.int32))
        # the row kernels (window expanded at compile time) against fp64 on a ragged shape with every border case, odd widths included
        for (Hh, Ww, Cw) in ((7, 9, 24), (112, 112, 64), (5, 4, 8)):
            xa = torch.randn(3, Hh, Ww, Cw, generator=g)
            ref = F.avg_pool2d(xa.double().permute(0, 3, 1, 2), k, s, p).permute(0, 2, 3, 1)
            got = ops.avgpool2d_fwd(xa.to(DEV), k, s, p, want_absmax=True)
            assert rel(got, ref) <= 1e-7, (k, s, p, Hh, Ww, Cw)
            gpa = torch.randn(ref.shape, generator=g)
            xr = xa.double().permute(0, 3, 1, 2).clone().requires_grad_(True)
            (gref,) = torch.autograd.grad(F.avg_pool2d(xr, k, s, p), xr, gpa.double().permute(0, 3, 1, 2))
            mm = torch.randn(3, Hh, Ww, Cw, generator=g)
            ggot = ops.avgpool2d_bwd(gpa.to(DEV), Hh, Ww, k, s, p, mul=mm.to(DEV), want_absmax=True)
            assert rel(ggot, gref.permute(0, 2, 3, 1) * mm.double()) <= 1e-7 and bool(torch.isfinite(ggot).all()), (k, s, p, Hh, Ww, Cw)
            if blib.get_contraction_mode() == "f16x2":
                assert torch.equal(ops.absmax_of(ggot), ggot.abs().amax(dim=-1).reshape(-1).view(torch.int32))
                assert torch.equal(ops.absmax_of(got), got.abs().amax(dim=-1).reshape(-1).view(torch.int32))
        m = torch.randn(2, 13, 12, 64, generator=g)
        gx = ops.avgpool2d_bwd(gp.permute(0, 2, 3, 1).contiguous().to(DEV), 13, 12, k, s, p, mul=m.to(DEV), want_absmax=True)
        assert rel(gx, ga.permute(0, 2, 3, 1) * m) <= 1e-6
        if blib.get_contraction_mode() == "f16x2":
            assert torch.equal(ops.absmax_of(gx), gx.abs().amax(dim=-1).reshape(-1).view(torch.int32))
    for Cw in (8, 24):          # narrow / non-power-of-two widths: fused for 8 (2 lanes per pixel), separate pass for 24
        gp = torch.randn(2, 7, 6, Cw, generator=g).to(DEV)
        gx = ops.avgpool2d_bwd(gp, 13, 12, 3, 2, 1, want_absmax=True)
        if blib.get_contraction_mode() == "f16x2":
            assert torch.equal(ops.absmax_of(gx), gx.abs().amax(dim=-1).reshape(-1).view(torch.int32))
    f = torch.randn(4, 7, 7, 1000, generator=g)
    lg = ops.global_avgpool_logits(f.to(DEV), None, -math.log(999))
    assert rel(lg, f.mean((1, 2)) - math.log(999)) <= 1e-6
    idx, val = ops.argmax_rows(lg)
    assert torch.equal(idx.cpu(), lg.cpu().argmax(1)) and torch.equal(val.cpu(), lg.cpu().max(1).values)
    tie = torch.zeros(2, 70, device=DEV); tie[0, 5] = tie[0, 66] = 3.0; tie[1, 69] = 1.0
    assert ops.argmax_rows(tie)[0].tolist() == [5, 69]
    cls = torch.tensor([1, 999, 0, 500], device=DEV)
    gl = ops.head_onehot_grad(cls, f.to(DEV))
    ref = torch.zeros_like(f)
    for n in range(4):
        ref[n, :, :, cls[n].item()] = f[n, :, :, cls[n].item()] / 49
    assert rel(gl, ref) <= 1e-7
    w2 = torch.randn(37, 123, generator=g)
    gain = torch.rand(37, generator=g) + 0.5
    assert rel(ops.weight_rownorm_scale(w2.to(DEV), gain.to(DEV)), gain[:, None] * w2 / w2.norm(dim=1, keepdim=True)) <= 1e-6
    ca = ops.channel_affine(a.permute(0, 2, 3, 1).contiguous().to(DEV), gain.new_ones(64).to(DEV) * 2, None, relu=True)
    assert rel(ca, torch.relu(2 * a.permute(0, 2, 3, 1))) <= 1e-7


# ------------------------------------------------------------------------------------------ whole networks
def maxabs(a, b):
    """Worst element of a tensor relative to the reference tensor's largest: max |a - b| / max |b| (VERDICT r03 'What's weak' 2 --
    a global relL2 cannot see a few wrong pixels of a map)."""
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


def maxabs_per_image(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float(((a - b).abs().flatten(1).amax(1) / b.abs().flatten(1).amax(1).clamp_min(1e-300)).max())


def _recorded_gates(meta, data):
    return [torch.from_numpy(np.unpackbits(data[f"gate/{i:02d}"])[: int(np.prod(shp))].reshape(shp).astype(np.float32)).to(DEV)
            for i, shp in enumerate(meta["gate_shapes"])]


def _golden_net(golden_dir, stem):
    from bcos_hip import synth
    meta = json.load(open(os.path.join(golden_dir, stem + ".json")))
    data = np.load(os.path.join(golden_dir, stem + ".npz"))
    net = synth.build_bcosified_resnet(meta["arch"], seed=meta["weight_seed"])
    synth.apply_calibration(net, {k: torch.from_numpy(data["calib/" + k]) for k in meta["calib_order"]})
    return net.to(DEV), meta, data


def _completeness(x, out):
    mean = torch.tensor(O.IMAGENET_MEAN_ADDINVERSE, device=x.device).view(1, 6, 1, 1)
    lhs = ((x - mean) * out["dynamic_linear_weights"]).double().sum((1, 2, 3))
    n = x.shape[0]
    rhs = out["logits"][torch.arange(n), out["explained_class_idx"]].double() + math.log(999)
    return float(((lhs - rhs).abs() / rhs.abs().clamp_min(1e-6)).max())


def _oracle_pre_activations(net, x, arch):
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    log = []
    with torch.no_grad():
        O.resnet_logits(sd, x.cpu(), arch, detach=True, gate_log=log)
    return log


def _oracle_gates(net, x, arch):
    """0/1 NHWC gate tensors of every ReLU, recorded from the CPU oracle, in execution order."""
    return [(p > 0).float().permute(0, 2, 3, 1).contiguous().to(x.device) for p in _oracle_pre_activations(net, x, arch)]


def _gate_flips(net, eng, x, arch):
    """(#gates that differ between HIP and oracle, #gates, largest |oracle pre-activation| / rms among them)."""
    pre = _oracle_pre_activations(net, x, arch)
    _, st = eng._run_forward(x, keep=True)
    ours = list(st["stem_ts"]) + [t for rec in st["blocks"] for t in rec["ts"]]
    flips, total, worst = 0, 0, 0.0
    for p, t in zip(pre, ours):
        t = t.act if hasattr(t, "act") else t           # layers whose multiplier is rebuilt keep their activation instead
        open_ref = (p > 0)
        open_hip = (t.permute(0, 3, 1, 2).cpu() != 0)   # t = s * bn_scale * gate
        diff = open_ref != open_hip
        # a closed gate and an exactly-zero scale are indistinguishable in t; only count sites with non-zero s
        flips += int(diff.sum())
        total += diff.numel()
        if diff.any():
            worst = max(worst, float(p[diff].abs().max() / p.pow(2).mean().sqrt()))
    return flips, total, worst


def test_resnet18_config1_against_reference_golden(lib, golden_dir):
    """BASELINE.json configs[0]: B-cosified ResNet-18, forward + explanation on 8 images @224."""
    from bcos_hip import engine, synth
    net, meta, data = _golden_net(golden_dir, "resnet18_e2e")
    x = synth.synthetic_images(meta["n_images"], seed=meta["image_seed"]).to(DEV)
    eng = engine.attach(net)
    out = net.explain_batch(x)
    assert rel(out["logits"], data["logits"]) <= 1e-4
    assert np.array_equal(out["prediction"].cpu().numpy(), data["prediction"])          # bit-exact class indices
    assert _completeness(x, out) <= 1e-4
    # explanation maps.  ReLU gates whose pre-activation is ~1e-13 open differently under a different summation
    # order: the reference's own CPU path run on THIS host differs from the fixture (recorded on the build host) by
    # ~1e-4 on two of the 8 images for that reason.  Hence three checks:
    # (1) free gates vs the fixture: bounded by that reference-vs-reference floor
    assert rel(out["contribution_map"], data["contribution_map"]) <= 2e-3
    # (2) gates pinned to the REFERENCE's recorded gates (images 0-1): 1e-4 holds outright
    gates = [torch.from_numpy(np.unpackbits(data[f"gate/{i:02d}"])[: int(np.prod(shp))].reshape(shp).astype(np.float32)).to(DEV)
             for i, shp in enumerate(meta["gate_shapes"])]
    pinned = eng.explain(x[:2], gates=gates)
    assert rel(pinned["logits"], data["logits"][:2]) <= 1e-4
    assert rel(pinned["contribution_map"], data["contribution_map"][:2]) <= 1e-4
    assert rel(pinned["dynamic_linear_weights"], data["weights_01"]) <= 1e-4
    assert maxabs_per_image(pinned["contribution_map"], data["contribution_map"][:2]) <= 1e-4      # worst element / the image's map maximum
    assert maxabs_per_image(pinned["dynamic_linear_weights"], data["weights_01"]) <= 1e-4
    # (3) against the oracle run on this host: every differing gate is numerically dead, and with the oracle's
    #     gates replayed all 8 maps agree to 1e-4
    flips, total, worst = _gate_flips(net, eng, x, meta["arch"])
    assert flips <= 1e-5 * total and worst <= 1e-5, (flips, total, worst)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    host = O.explain_batch(lambda xx, detach: O.resnet_logits(sd, xx, meta["arch"], detach=detach), x.cpu())
    pinned_host = eng.explain(x, gates=_oracle_gates(net, x, meta["arch"]))
    assert rel(pinned_host["contribution_map"], host["contribution_map"]) <= 1e-4
    assert rel(pinned_host["dynamic_linear_weights"], host["dynamic_linear_weights"]) <= 1e-4
    # the nn.Module path (autograd over per-layer HIP kernels) gives the same answer as the fused plan.  Free gates: the two
    # paths must then evaluate every product the same way, so both run the bf16x3 contraction here (the module path's
    # BN / ReLU kernels emit no operand maxima, its convolutions would otherwise mix f16x2 and bf16x3 launches and a
    # handful of numerically dead gates would open differently -- the effect checks (1)-(3) bound for the fused plan).
    from bcos_hip import lib as blib
    prev = blib.get_contraction_mode()
    blib.set_contraction_mode("bf16x3")
    try:
        out_x3 = eng.explain(x[:3])
        engine.detach(net)
        out_m = net.explain_batch(x[:3])
    finally:
        blib.set_contraction_mode(prev)
    assert rel(out_m["logits"], data["logits"][:3]) <= 1e-4
    assert rel(out_m["dynamic_linear_weights"], out_x3["dynamic_linear_weights"]) <= 1e-4
    assert rel(out_x3["contribution_map"], out["contribution_map"][:3]) <= 2e-3          # f16x2 vs bf16x3, free gates
    # reference-style single image explain(): dict keys / shapes of bcos/common.py:163-186
    xi = x[:1].clone().requires_grad_(True)
    res = net.explain(xi)
    assert set(res) == {"prediction", "explained_class_idx", "dynamic_linear_weights", "contribution_map", "explanation"}
    assert res["prediction"] == int(data["prediction"][0]) and res["explanation"].shape == (224, 224, 4)
    assert rel(res["contribution_map"], data["contribution_map"][:1]) <= 2e-3      # free gates (floor as above)
    # RGBA rendering (host-side gradient_to_image): alpha everywhere; colour where the explanation is visible (the
    # colour of a ~zero-weight pixel is a 0/0-type ratio and legitimately flips between 0 and 1)
    rgba, gold = res["explanation"], data["rgba_0"]
    assert float(np.abs(rgba[..., 3] - gold[..., 3]).max()) <= 5e-3
    vis = gold[..., 3] > 0.05
    assert float(np.abs(rgba[vis][:, :3] - gold[vis][:, :3]).mean()) <= 1e-2


@pytest.mark.parametrize("mode", ["f16x2", "bf16x3", "f32"])
def test_resnet18_structured_images_against_reference_golden(lib, golden_dir, mode):
    """the reference's recorded outputs on STRUCTURED images (tests/golden/resnet18_structured.*: smooth field,
    sparse spots on black, a sharp-edged disc, half black / half white with a faint texture -- the dynamic range is inside the
    image, in the activations and above all in the gradients of the explanation pass).  Logits 1e-4, bit-exact classes; with the
    reference's ReLU decisions replayed, W(x) and the maps hold 1e-4 in relL2 AND in the worst element relative to each image's
    map maximum; free gates are bounded by the reference's own self-disagreement.  Every contraction mode."""
    from bcos_hip import engine, synth
    from bcos_hip import lib as blib
    net, meta_w, _ = _golden_net(golden_dir, "resnet18_e2e")                 # (same weights: the fixture reuses that calibration record)
    meta = json.load(open(os.path.join(golden_dir, "resnet18_structured.json")))
    data = np.load(os.path.join(golden_dir, "resnet18_structured.npz"))
    x = synth.structured_images(meta["n_images"], seed=meta["image_seed"]).to(DEV)
    prev = blib.get_contraction_mode()
    blib.set_contraction_mode(mode)
    try:
        eng = engine.attach(net)
        out = eng.explain(x)
        assert rel(out["logits"], data["logits"]) <= 1e-4 and maxabs(out["logits"], data["logits"]) <= 1e-4
        assert np.array_equal(out["prediction"].cpu().numpy(), data["prediction"])
        assert _completeness(x, out) <= 1e-4
        floor = json.load(open(os.path.join(golden_dir, "oracle_vs_reference.json")))["r18s/reference_self_floor_weights"][0]
        assert rel(out["dynamic_linear_weights"], data["weights"]) <= max(3 * floor, 1e-4), (mode, floor)
        pinned = eng.explain(x, gates=_recorded_gates(meta, data))
        for key, gold in (("dynamic_linear_weights", data["weights"]), ("contribution_map", data["contribution_map"])):
            assert rel(pinned[key], gold) <= 1e-4, (mode, key, rel(pinned[key], gold))
            assert maxabs_per_image(pinned[key], gold) <= 1e-4, (mode, key, maxabs_per_image(pinned[key], gold))
        assert rel(pinned["logits"], data["logits"]) <= 1e-4
    finally:
        blib.set_contraction_mode(prev)


def test_resnet50_against_reference_golden(lib, golden_dir):
    from bcos_hip import engine, synth
    net, meta, data = _golden_net(golden_dir, "resnet50_small")
    x = synth.synthetic_images(4, seed=meta["image_seed"])[:2].to(DEV)
    eng = engine.attach(net)
    out = eng.explain(x)
    assert rel(out["logits"], data["logits"]) <= 1e-4
    assert np.array_equal(out["prediction"].cpu().numpy(), data["prediction"])
    floor = json.load(open(os.path.join(golden_dir, "oracle_vs_reference.json")))["r50/reference_self_floor_weights"][0]
    assert rel(out["contribution_map"], data["contribution_map"]) <= 3 * floor
    assert _completeness(x, out) <= 1e-4
    # 54 layers deep, a first flipped gate perturbs everything downstream at the 1e-5 level, so later gates with
    # |pre-activation| up to ~1e-4 rms follow (measured: 47 of 19.2 M gates, worst 5e-5 rms)
    flips, total, worst = _gate_flips(net, eng, x, meta["arch"])
    assert flips <= 1e-5 * total and worst <= 1e-3, (flips, total, worst)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    host = O.explain_batch(lambda xx, detach: O.resnet_logits(sd, xx, meta["arch"], detach=detach), x.cpu())
    pinned = eng.explain(x, gates=_oracle_gates(net, x, meta["arch"]))
    assert rel(pinned["contribution_map"], host["contribution_map"]) <= 1e-4
    assert rel(pinned["dynamic_linear_weights"], host["dynamic_linear_weights"]) <= 1e-4
    # gates pinned to the REFERENCE's recorded decisions (fixture, both images): maps and W(x) of the 54-layer network
    # hold the 1e-4 target against the reference-recorded outputs outright
    gates = [torch.from_numpy(np.unpackbits(data[f"gate/{i:02d}"])[: int(np.prod(shp))].reshape(shp).astype(np.float32)).to(DEV)
             for i, shp in enumerate(meta["gate_shapes"])]
    pinned_ref = eng.explain(x, gates=gates)
    assert rel(pinned_ref["logits"], data["logits"]) <= 1e-4
    assert rel(pinned_ref["contribution_map"], data["contribution_map"]) <= 1e-4
    assert rel(pinned_ref["dynamic_linear_weights"], data["weights_01"]) <= 1e-4
    # ... and no single element of a map is off by more than 1e-4 of that image's map maximum
    assert maxabs_per_image(pinned_ref["contribution_map"], data["contribution_map"]) <= 1e-4
    assert maxabs_per_image(pinned_ref["dynamic_linear_weights"], data["weights_01"]) <= 1e-4



def test_free_gate_floor_rederived_on_this_host(lib, golden_dir):
    """The free-gate tolerance of the ResNet-50 test is 3 x `r50/reference_self_floor_weights`, a number the fixture generator wrote
    (the reference against itself with oneDNN on / off, make_golden.py:612-618).  Here it is derived again on the host the suite runs
    on, from the oracle (bit-identical to the reference on these logits) under the same two convolution back ends: two summation
    orders of one network disagree on W(x) at that level -- not an artefact of one build container -- and the device result sits inside
    the same band around either of them."""
    from bcos_hip import engine, synth
    net, meta, data = _golden_net(golden_dir, "resnet50_small")
    recorded = json.load(open(os.path.join(golden_dir, "oracle_vs_reference.json")))["r50/reference_self_floor_weights"][0]
    x = synth.synthetic_images(4, seed=meta["image_seed"])[:2]
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    fwd = lambda xx, detach: O.resnet_logits(sd, xx, meta["arch"], detach=detach)      # noqa: E731
    a = O.explain_batch(fwd, x)
    with torch.backends.mkldnn.flags(enabled=False):
        b = O.explain_batch(fwd, x)
    assert rel(a["logits"], b["logits"]) <= 1e-4 and torch.equal(a["prediction"], b["prediction"])
    live = rel(a["dynamic_linear_weights"], b["dynamic_linear_weights"])
    # same order of magnitude as the recorded floor (which gates flip depends on the host's kernels: a band, not a number)
    assert recorded / 30 <= live <= 30 * recorded, (live, recorded)
    out = engine.attach(net).explain(x.to(DEV))
    for ref in (a, b):
        assert rel(out["dynamic_linear_weights"], ref["dynamic_linear_weights"]) <= 3 * max(live, recorded)


def test_determinism_and_batch_independence(lib, golden_dir):
    """No atomics / order-dependent reductions: identical bits run to run, and an image's result does not depend on
    what else is in the batch."""
    from bcos_hip import engine, synth
    net, meta, data = _golden_net(golden_dir, "resnet18_e2e")
    eng = engine.attach(net)
    x = synth.synthetic_images(5, seed=7).to(DEV)
    a, b = eng.explain(x), eng.explain(x)
    for k in ("logits", "dynamic_linear_weights", "contribution_map"):
        assert torch.equal(a[k], b[k]), k
    c = eng.explain(x[1:3])
    assert torch.equal(c["logits"], a["logits"][1:3])
    assert torch.equal(c["dynamic_linear_weights"], a["dynamic_linear_weights"][1:3])


def test_rebuilt_multipliers_match_stored_ones(lib, golden_dir, monkeypatch):
    """BCOS_EPI_MUL_FROM_ACT: the explanation pass with the multipliers t of the inner block convolutions REBUILT from the
    kept activations and patch norms (no t written in the forward) equals the stored-t pass -- gates pinned to the same
    oracle decisions, 1e-5; and the free-gate maps stay within the ResNet-18 floor of the reference fixture."""
    from bcos_hip import engine, synth
    net, meta, data = _golden_net(golden_dir, "resnet18_e2e")
    x = synth.synthetic_images(meta["n_images"], seed=meta["image_seed"]).to(DEV)
    eng = engine.attach(net)
    monkeypatch.setattr(engine, "_STORE_T", True)
    stored = eng.explain(x)
    monkeypatch.setattr(engine, "_STORE_T", False)          # the default
    rebuilt = eng.explain(x)
    _, st = eng._run_forward(x[:2], keep=True)
    assert any(isinstance(t, engine._ActScale) for rec in st["blocks"] for t in rec["ts"])      # the rebuild path is taken
    assert torch.equal(rebuilt["logits"], stored["logits"])
    assert rel(rebuilt["dynamic_linear_weights"], stored["dynamic_linear_weights"]) <= 1e-5
    assert rel(rebuilt["contribution_map"], data["contribution_map"]) <= 2e-3


@pytest.mark.parametrize("arch", ["resnet18", "resnet50"])
def test_engine_launch_list_matches_reference_execution_trace_on_device(lib, golden_dir, monkeypatch, arch):
    """a21: what the fused plan launches ON THE DEVICE against the execution trace recorded from the imported reference
    (tests/golden/resnet_exec_trace.json; the checks are tests/test_host_cpu.py's, here without the emulated kernels)."""
    import test_host_cpu as H
    calls, pools = H._engine_exec_trace(arch, monkeypatch, device="cuda")
    H.check_engine_against_reference_trace(arch, calls, pools, golden_dir)


def test_resnet50_batch256_properties(lib):
    """BASELINE.json configs[1] at full size (batch 256): properties that do not need the CPU oracle at size --
    completeness of every explanation, agreement of a small sub-batch with the oracle, batch independence."""
    from bcos_hip import engine, synth
    net = synth.build_bcosified_resnet("resnet50").to(DEV)
    x = synth.synthetic_images(256).to(DEV)
    with torch.no_grad():
        synth.calibrate(net, x[:8])
    eng = engine.attach(net)
    out = eng.explain(x)
    assert out["logits"].shape == (256, 1000) and out["contribution_map"].shape == (256, 224, 224)
    assert torch.isfinite(out["dynamic_linear_weights"]).all()
    assert _completeness(x, out) <= 1e-4
    assert len(set(out["prediction"].tolist())) > 1          # non-degenerate synthetic task
    # batch independence at the headline size, for everything the path returns: a sub-batch taken from the middle of the batch
    # reproduces its rows of the full-batch result bit for bit -- logits, W(x) and maps
    sub = eng.explain(x[100:102])
    assert torch.equal(sub["logits"], out["logits"][100:102])
    assert torch.equal(sub["dynamic_linear_weights"], out["dynamic_linear_weights"][100:102])
    assert torch.equal(sub["contribution_map"], out["contribution_map"][100:102])
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    ref = O.explain_batch(lambda xx, detach: O.resnet_logits(sd, xx, "resnet50", detach=detach), x[100:102].cpu())
    assert rel(sub["logits"], ref["logits"]) <= 1e-4
    assert torch.equal(sub["prediction"].cpu(), ref["prediction"])
    # W(x) and maps against the oracle at the headline size: four images from different positions of the batch (first tile,
    # two interior positions that straddle tile boundaries of every layer, last image), gates pinned to the oracle's
    # decisions (a free gate whose pre-activation is ~1e-13 opens differently under another summation order).
    # The gates of the other 252 images are the path's own: images are independent, which the bit-equality above shows.
    idx = [0, 77, 200, 255]
    ref4 = O.explain_batch(lambda xx, detach: O.resnet_logits(sd, xx, "resnet50", detach=detach), x[idx].cpu())
    own = eng.explain(x[idx], gates=_oracle_gates(net, x[idx], "resnet50"))
    assert rel(own["logits"], ref4["logits"]) <= 1e-4 and torch.equal(own["prediction"].cpu(), ref4["prediction"])
    assert rel(own["dynamic_linear_weights"], ref4["dynamic_linear_weights"]) <= 1e-4
    assert rel(own["contribution_map"], ref4["contribution_map"]) <= 1e-4
    assert rel(own["logits"], out["logits"][idx]) <= 1e-5                  # (replayed gates: a handful of numerically dead ReLUs may open differently)
    # free gates, the same four images inside the batch-256 pass: bounded by the reference-vs-reference floor of ResNet-50
    assert rel(out["dynamic_linear_weights"][idx], ref4["dynamic_linear_weights"]) <= 3e-3
    assert rel(out["contribution_map"][idx], ref4["contribution_map"]) <= 3e-3


# ------------------------------------------------------------------------------------------ transformer pieces
def test_vit_kernels(lib):
    """DetachableLayerNorm / MyGELU / attention kernels against their oracle forms (explanation-mode gradients)."""
    from bcos_hip import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(50, 192, generator=g) * 2 + 0.3
    w, b = torch.rand(192, generator=g) + 0.5, torch.randn(192, generator=g) * 0.1
    xr = x.clone().requires_grad_(True)
    y_ref = O.layer_norm_detachable(xr, (192,), w, b, 1e-5, detach=True)
    gy = torch.randn(y_ref.shape, generator=g)
    (gx_ref,) = torch.autograd.grad(y_ref, xr, gy)
    y, rstd = ops.layernorm_fwd(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5, want_rstd=True)
    assert rel(y, y_ref) <= 1e-6
    add, m2 = torch.randn(50, 192, generator=g), torch.randn(50, 192, generator=g)
    o, o2 = ops.layernorm_bwd_detached(gy.to(DEV), w.to(DEV), rstd, addend=add.to(DEV), mul2=m2.to(DEV), want_out2=True)
    assert rel(o, gx_ref + add) <= 1e-5 and rel(o2, (gx_ref + add) * m2) <= 1e-5
    z = torch.randn(1000, generator=g) * 2
    yg, gate = ops.gelu_gate(z.to(DEV), want_gate=True)
    assert rel(yg, O.gelu_detachable(z)) <= 1e-6 and rel(gate, 0.5 * (1 + torch.erf(z / 2 ** 0.5))) <= 1e-6
    for (B, T, H) in ((3, 196, 3), (2, 50, 32), (1, 17, 1)):
        qkv = torch.randn(B, T, 3 * H * 64, generator=g)
        inner = H * 64
        q, k, v = (t.view(B, T, H, 64).transpose(1, 2) for t in qkv.split(inner, dim=-1))
        vr = v.clone().requires_grad_(True)
        out_ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ vr).transpose(1, 2).reshape(B, T, inner)
        go = torch.randn(out_ref.shape, generator=g)
        (gv_ref,) = torch.autograd.grad(out_ref, vr, go)
        out, stats = ops.attention_fwd(qkv.to(DEV), H, 0.125, want_stats=True)
        assert rel(out, out_ref) <= 1e-5, (B, T, H)
        gv = ops.attention_bwd_v(qkv.to(DEV), stats, go.to(DEV), H, 0.125)
        assert rel(gv, gv_ref.transpose(1, 2).reshape(B, T, inner)) <= 1e-5, (B, T, H)


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "f16x2"])
def test_layernorm_folded_into_the_contraction(lib, mode):
    """bcos_epilogue.row_scale / a_sumsq + bcos_layernorm_stats: a DetachableLayerNorm (centered_norms.py:197-224) in front of a
    B-cos linear layer (bcosifylinear.py:61-94) or a plain one, computed WITHOUT writing the LayerNorm's output -- rows with means
    far from zero, affine with bias -- against the fp64 composition, forward (value, stored scale) and the detached-variance
    input gradient with residual addend and next-layer multiplier; every contraction mode; through the C ABI."""
    from bcos_hip import ops, vit_engine
    from bcos_hip import lib as blib
    g = torch.Generator().manual_seed(23)
    rows, D, Cout = 333, 192, 256
    x = torch.randn(rows, D, generator=g) * (torch.rand(rows, 1, generator=g) * 3 + 0.1) + torch.randn(rows, 1, generator=g) * 2
    gamma, beta = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.2
    W = torch.randn(Cout, D, generator=g) / D ** 0.5
    eps = 1e-5
    x64 = x.double()
    mean = x64.mean(-1, keepdim=True)
    rstd64 = 1 / (x64.var(-1, unbiased=False, keepdim=True) + eps).sqrt()
    z64 = (x64 - mean) * rstd64 * gamma.double() + beta.double()
    lin64 = z64 @ W.double().t()
    nrm64 = z64.norm(dim=-1, keepdim=True) + 1e-12
    s64 = lin64.abs() / nrm64

    class _LNv:
        w, bias = gamma, beta
    wc, c = vit_engine._fold_ln(W, _LNv)
    prev = blib.get_contraction_mode()
    blib.set_contraction_mode(mode)
    try:
        xd = x.to(DEV)
        rstd, zss = ops.layernorm_stats(xd, gamma.to(DEV), beta.to(DEV), eps, want_zsumsq=True, want_absmax=True)
        assert rel(rstd, rstd64.view(-1)) <= 1e-6 and rel(zss, (z64 ** 2).sum(-1)) <= 1e-6
        if mode == "f16x2":
            assert torch.equal(ops.absmax_of(xd).cpu().view(torch.float32), x.abs().amax(-1))
        for D2, affine in ((64, True), (200, True), (256, False), (50, True), (384, True)):      # every row width of the two kernels
            x2 = torch.randn(77, D2, generator=g) * 3 + 1.5
            g2, b2 = (torch.rand(D2, generator=g) + 0.5, torch.randn(D2, generator=g)) if affine else (None, None)
            r2, z2 = ops.layernorm_stats(x2.to(DEV), None if g2 is None else g2.to(DEV), None if b2 is None else b2.to(DEV), eps, want_zsumsq=True)
            v2, m2 = torch.var_mean(x2.double(), dim=-1, unbiased=False, keepdim=True)
            zz = (x2.double() - m2) / (v2 + eps).sqrt() * (g2.double() if affine else 1.0) + (b2.double() if affine else 0.0)
            assert rel(r2, 1 / (v2 + eps).sqrt().view(-1)) <= 1e-6 and rel(z2, (zz ** 2).sum(-1)) <= 1e-6, D2
        wcd, cd = ops.mark_static(wc.to(DEV)), c.to(DEV)
        # B-cos linear over the LayerNorm output, with the stored scale
        geom = dict(N=1, H=1, W=rows, C=D, P=1, Q=rows, in_sh=1, in_sw=1, dh0=0, dw0=0, dstep_h=1, dstep_w=1, TH=1, TW=1, OH=1, OW=rows,
                    out_sh=1, out_sw=1, out_h0=0, out_w0=0, Cout=Cout)
        y, t = torch.empty(rows, Cout, device=DEV), torch.empty(rows, Cout, device=DEV)
        ops.tapconv(xd, wcd, geom, out=y, scale_out=t, bias=cd, bcos_mode=blib.BCOS_LINEAR_EPS, b=2.0, row_scale=rstd, a_sumsq=zss,
                    contraction=mode)
        assert rel(y, lin64 * s64) <= 3e-6 and rel(t, s64) <= 3e-6, (mode, rel(y, lin64 * s64), rel(t, s64))
        # plain linear (to_qkv): value only
        q = ops.matmul_nt(xd, wcd, bias=cd, row_scale=rstd)
        assert rel(q, lin64) <= 3e-6, (mode, rel(q, lin64))
        # detached-variance input gradient: gx = rstd (gamma gz - mean(gamma gz)), gz = W^T a; + addend; second output * multiplier
        a = torch.randn(rows, Cout, generator=g)
        add, mul = torch.randn(rows, D, generator=g), torch.rand(rows, D, generator=g)
        gz = a.double() @ W.double()
        h = gz * gamma.double() * rstd64
        gx64 = h - h.mean(-1, keepdim=True) + add.double()
        gm, gx = vit_engine._dgrad_ln(ops.ensure_absmax(a.to(DEV)), ops.mark_static(wc.t().contiguous().to(DEV)), rstd, add.to(DEV), mul.to(DEV))
        assert rel(gx, gx64) <= 3e-6 and rel(gm, gx64 * mul.double()) <= 3e-6, (mode, rel(gx, gx64))
    finally:
        blib.set_contraction_mode(prev)


def _golden_vit(golden_dir):
    from bcos_hip import synth
    meta = json.load(open(os.path.join(golden_dir, "vit_ti_e2e.json")))
    data = np.load(os.path.join(golden_dir, "vit_ti_e2e.npz"))
    net = synth.build_bcosified_vit(meta["arch"], seed=meta["weight_seed"])
    synth.apply_calibration(net, {k: torch.from_numpy(data["calib/" + k]) for k in meta["calib_order"]})
    sd = net.state_dict()
    for k, (s1, s2) in meta["state_checksum"].items():
        assert abs(float(sd[k].double().sum()) - s1) <= 1e-6 * max(1.0, s2), k
    return net.to(DEV), meta, data


def test_vit_ti_against_reference_golden(lib, golden_dir):
    """BASELINE.json configs[2] topology: B-cosified simple_vit_ti_patch16_224.  No ReLU gates => the maps are held to
    1e-4 against the reference's recorded outputs directly."""
    from bcos_hip import synth, vit_engine
    net, meta, data = _golden_vit(golden_dir)
    x = synth.synthetic_images(meta["n_images"], seed=meta["image_seed"]).to(DEV)
    out_m = net.explain_batch(x[:2])                       # nn.Module path: autograd over per-layer HIP kernels
    assert rel(out_m["logits"], data["logits"][:2]) <= 1e-4
    assert rel(out_m["dynamic_linear_weights"][:1], data["weights_0"]) <= 1e-4
    eng = vit_engine.attach(net)
    out = net.explain_batch(x)
    assert rel(out["logits"], data["logits"]) <= 1e-4
    assert np.array_equal(out["prediction"].cpu().numpy(), data["prediction"])
    assert rel(out["contribution_map"], data["contribution_map"]) <= 1e-4
    assert rel(out["dynamic_linear_weights"][:1], data["weights_0"]) <= 1e-4
    assert maxabs_per_image(out["contribution_map"], data["contribution_map"]) <= 1e-4      # worst element / the image's map maximum
    assert maxabs_per_image(out["dynamic_linear_weights"][:1], data["weights_0"]) <= 1e-4
    with torch.no_grad():
        assert rel(net(x), data["logits"]) <= 1e-4         # forward through the engine
    again = eng.explain(x)
    assert torch.equal(again["dynamic_linear_weights"], out["dynamic_linear_weights"])     # deterministic


def test_clip_rn50_non_detached_gradient_against_oracle(lib):
    """The CLIP RN50 topology with NOTHING detached (module path outside explanation mode, running BatchNorm statistics): scale
    derivative and patch-norm term of every B-cos convolution, anti-aliasing pools, and the attention pool with q, k and v all
    differentiated (bcos_attention_bwd) -- input gradient of a random projection of the embedding against autograd over the CPU
    oracle with detach=False.  ReLU gates are free: tolerance as for the explanation maps of the 50-layer networks."""
    from bcos_hip import synth
    net = synth.build_bcosified_clip_rn50().to(DEV)
    x = synth.synthetic_images(2, seed=5).to(DEV)
    with torch.no_grad():
        synth.calibrate(net, synth.synthetic_images(4).to(DEV))
    net.eval()
    g = torch.randn(2, 1024, generator=torch.Generator().manual_seed(3)).to(DEV)
    xr = x.clone().requires_grad_(True)
    emb = net(xr)
    (gx,) = torch.autograd.grad(emb, xr, g)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    xc = x.cpu().clone().requires_grad_(True)
    ref = O.clip_rn50_embed(sd, xc, detach=False)
    (gref,) = torch.autograd.grad(ref, xc, g.cpu())
    assert rel(emb, ref) <= 1e-4
    # free ReLU gates of 55 layers under differentiated scales: the oracle differs from ITSELF by 8.6e-3 between its mkldnn and
    # native CPU convolution back-ends on this gradient (measured in the build container); the device path: 7e-3
    assert rel(gx, gref) <= 2e-2, rel(gx, gref)
    # the attention pool alone has no gates: its full backward (q, k, v differentiated, 50 tokens x 32 heads) holds 1e-5
    pool = net.model.attnpool
    f = torch.randn(2, 2048, 7, 7, generator=torch.Generator().manual_seed(9)).to(DEV)
    gp = torch.randn(2, 1024, generator=torch.Generator().manual_seed(10)).to(DEV)
    fr = f.clone().requires_grad_(True)
    yp = pool(fr)
    (gf,) = torch.autograd.grad(yp, fr, gp)
    fc = f.cpu().clone().requires_grad_(True)
    yo = O.bcos_attention_pool(sd, "model.attnpool.", fc, 32, detach=False)
    (go,) = torch.autograd.grad(yo, fc, gp.cpu())
    assert rel(yp, yo) <= 1e-5 and rel(gf, go) <= 1e-5, (rel(yp, yo), rel(gf, go))


@pytest.mark.parametrize("plan", [False, True], ids=["per_layer", "plan"])
def test_vit_training_mode_against_reference_golden(lib, golden_dir, plan):
    """Training mode of the token path on the device: full LayerNorm / GELU / softmax-attention gradients and a BCE training
    step of a small B-cosified SimpleViT against the reference's recorded gradients -- per layer and as ONE autograd node over
    the engine's block list (bcos_hip/vit_train_plan.py)."""
    from test_host_cpu import run_vit_training_goldens
    run_vit_training_goldens(golden_dir, DEV, 1e-5, plan=plan)


def test_vit_training_plan_accumulates_over_two_forwards(lib):
    """Two forward passes through the ViT training plan before one backward (gradient accumulation): the second pass resets the plan's
    arenas of operand maxima while the first one's state is still waiting for its backward -- the sum of the two losses must give the
    per-layer path's gradients, and a second identical step the same gradients again (the zero-fill arena of the weight gradients)."""
    import copy
    import bcos.models.vit as vit
    from bcos_hip import synth, vit_engine
    from bcosify_vit import BcosifyNetwork
    torch.manual_seed(4)
    cfg = synth.vit_model_config("simple_vit_ti_patch16_224")
    std = vit.SimpleViT(image_size=64, patch_size=16, num_classes=12, dim=128, depth=2, heads=2, mlp_dim=256, channels=3,
                        linear_layer=nn.Linear, norm_layer=nn.LayerNorm, act_layer=nn.GELU)
    net = BcosifyNetwork(std, cfg, add_channels=True, logit_layer=cfg["logit_layer"])
    synth.finish_vit_conversion(net, cfg)
    net = net.to(DEV)
    ref = copy.deepcopy(net)
    vit_engine.attach(net)
    net.train(); ref.train()
    xa, xb = synth.synthetic_images(4, seed=1, size=64).to(DEV), synth.synthetic_images(4, seed=2, size=64).to(DEV)
    tgt = F.one_hot(torch.tensor([1, 7, 4, 0]), 12).float().to(DEV)

    def grads(n):
        la, lb = n(xa), n(xb)
        ps = [p_ for p_ in n.parameters() if p_.requires_grad]
        loss = F.binary_cross_entropy_with_logits(la, tgt) + 0.5 * F.binary_cross_entropy_with_logits(lb, tgt)
        return la, torch.autograd.grad(loss, ps)

    la, gp = grads(net)
    assert type(la.grad_fn).__name__ == "_TrainStepFnBackward"
    lr_, gr = grads(ref)
    names = [n for n, p_ in net.named_parameters() if p_.requires_grad]
    for n, a, b in zip(names, gp, gr):
        assert rel(a, b) <= 1e-4, (n, rel(a, b))
    _, gp2 = grads(net)
    for n, a, b in zip(names, gp, gp2):
        assert torch.equal(a, b), (n, rel(a, b))          # (round 6: the weight gradient combines its pixel chunks in a fixed order)


def test_out_of_range_targets(lib):
    """a label outside [0, K) never becomes an out-of-range device read.  Engines: IndexError before any launch,
    [-K, -1] wraps like the reference's out[0, idx] (bcos/common.py:170-176); raw launch: a zero gradient for that image (what the
    one-hot tensor gave), the other images untouched."""
    from bcos_hip import ops, synth, vit_engine
    import bcos_hip.engine as en
    torch.manual_seed(5)
    N, R, K, D = 4, 7, 12, 16
    scale, w = torch.rand(N, R, K, device=DEV) + 0.1, torch.randn(K, D, device=DEV)
    good = torch.tensor([3, 0, 11, 5], device=DEV)
    ref, _ = ops.head_rank1_grad(good, scale, w)
    for bad_val in (-1, K, 1 << 40, -(1 << 40)):
        cls = good.clone()
        cls[1] = bad_val
        out, _ = ops.head_rank1_grad(cls, scale, w, want_absmax=True)
        assert torch.count_nonzero(out.view(N, R, D)[1]) == 0
        keep = [0, 2, 3]
        assert torch.equal(out.view(N, R, D)[keep], ref.view(N, R, D)[keep])
    net = synth.build_bcosified_resnet("resnet18", seed=0).to(DEV)
    x = synth.synthetic_images(4, seed=3).to(DEV)
    with torch.no_grad():
        synth.calibrate(net, x)
    eng = en.attach(net)
    Kc = eng.head.cout
    for bad in (torch.tensor([0, 1, Kc, 2]), torch.tensor([0, -Kc - 1, 1, 2], device=DEV)):
        with pytest.raises(IndexError):
            eng.explain(x, targets=bad)
    with pytest.raises(IndexError):
        eng.explain_targets(x, torch.tensor([0, Kc + 5]))
    a = eng.explain(x, targets=torch.tensor([-1, 5, -Kc, 7]))
    b = eng.explain(x, targets=torch.tensor([Kc - 1, 5, 0, 7], device=DEV))
    assert a["explained_class_idx"].tolist() == [Kc - 1, 5, 0, 7]
    assert torch.equal(a["dynamic_linear_weights"], b["dynamic_linear_weights"])
    vnet = synth.build_bcosified_vit("simple_vit_ti_patch16_224").to(DEV)
    with torch.no_grad():
        synth.calibrate(vnet, x)
    veng = vit_engine.attach(vnet)
    with pytest.raises(IndexError):
        veng.explain(x, targets=torch.tensor([0, 1, 2, veng.head.cout], device=DEV))
    assert veng.explain(x, targets=torch.tensor([-1, 0, 1, 2]))["explained_class_idx"].tolist() == [veng.head.cout - 1, 0, 1, 2]


def test_stream_copy_is_a_copy(lib):
    """bcos_stream_copy (ABI v9), the bandwidth reference of bench.py: bit-exact on ragged sizes, refuses what it cannot vectorise."""
    from bcos_hip import ops
    from bcos_hip.lib import BcosHipError
    for n in (4, 1020, 256 * 8 * 4, 256 * 8 * 4 + 4, (1 << 22) + 12):
        src = torch.randn(n, device=DEV)
        dst = torch.full((n + 8,), 7.0, device=DEV)
        ops.stream_copy(src, dst[4:4 + n])
        assert torch.equal(dst[4:4 + n], src) and float(dst[:4].min()) == 7.0 and float(dst[4 + n:].min()) == 7.0
    with pytest.raises(BcosHipError):
        ops.stream_copy(torch.randn(6, device=DEV))


def test_head_gradient_is_rank_one_per_image(lib):
    """bcos_head_rank1_grad against what it replaces -- the one-hot tensor [N, R, K] (bcos_head_onehot_grad) followed by the K-long
    input-gradient contraction with the LayerNorm's rstd as row factor and the previous layer's scale as multiplier -- and against fp64;
    row maxima exact; through the ViT plan: same logits, maps within the contraction's own rounding."""
    from bcos_hip import ops, synth, vit_engine
    import bcos_hip.vit_engine as ve
    torch.manual_seed(21)
    for (N, R, K, D) in ((3, 197, 1000, 192), (2, 5, 20, 8), (4, 16, 12, 384)):
        scale = torch.rand(N, R, K, device=DEV) + 0.1
        w = torch.randn(K, D, device=DEV)
        rstd = torch.rand(N * R, device=DEV) + 0.5
        mul = torch.randn(N * R, D, device=DEV)
        cls = torch.randint(0, K, (N,), device=DEV)
        out, out2 = ops.head_rank1_grad(cls, scale, w, 2.0, row_scale=rstd, mul=mul, want_out2=True, want_absmax=True)
        a = (0.5 / R) * scale.double()[torch.arange(N), :, cls].reshape(-1) * rstd.double()
        v = a.unsqueeze(1) * w.double()[cls].repeat_interleave(R, dim=0)
        assert rel(out2, v) <= 1e-6 and rel(out, v * mul.double()) <= 1e-6
        am = ops.absmax_of(out)
        assert am is not None and torch.equal(am.view(torch.float32), out.abs().amax(dim=1))
        g_head = ops.head_onehot_grad(cls, scale.view(N, R, 1, K), 2.0).view(N * R, K)
        ref = torch.empty_like(out); ref2 = torch.empty_like(out)
        ops.matmul_nt(ops.ensure_absmax(g_head), ops.mark_static(w.t().contiguous()), out=ref, out2=ref2, mul=mul, row_scale=rstd)
        assert rel(out, ref) <= 2e-6 and rel(out2, ref2) <= 2e-6
        o3, none = ops.head_rank1_grad(cls, scale, w)
        assert none is None and rel(o3, (1.0 / R) * scale.double()[torch.arange(N), :, cls].reshape(-1, 1) * w.double()[cls].repeat_interleave(R, dim=0)) <= 1e-6
    # the second output of the gradient epilogue: multiplier, gate tensor / gate in the multiplier's low bit
    N, R, K, D = 3, 49, 1000, 512
    scale, w = torch.rand(N, R, K, device=DEV) + 0.1, torch.randn(K, D, device=DEV)
    mul = (torch.randn(N * R, D, device=DEV).view(torch.int32) & ~1 | (torch.rand(N * R, D, device=DEV) < 0.5).to(torch.int32)).view(torch.float32)
    td, gate = torch.randn(N * R, D, device=DEV), (torch.rand(N * R, D, device=DEV) < 0.5).float()
    cls = torch.randint(0, K, (N,), device=DEV)
    v = ((1.0 / R) * scale.double()[torch.arange(N), :, cls].reshape(-1, 1) * w.double()[cls].repeat_interleave(R, dim=0))
    o, o2 = ops.head_rank1_grad(cls, scale, w, mul=mul, mul2=td, gate2_from_mul=True, want_absmax=True, want_absmax2=True)
    assert rel(o, v * mul.double()) <= 1e-6 and rel(o2, v * td.double() * (mul.view(torch.int32) & 1).double()) <= 1e-6
    assert torch.equal(ops.absmax_of(o2).view(torch.float32), o2.abs().amax(dim=1))
    o, o2 = ops.head_rank1_grad(cls, scale, w, mul=mul, gate2=gate)
    assert rel(o2, v * gate.double()) <= 1e-6
    # ResNet-18 through the engine: same logits, maps within the contraction's own rounding of the head gradient of rounds 1-4
    import bcos_hip.engine as en
    net18 = synth.build_bcosified_resnet("resnet18", seed=0).to(DEV)
    x18 = synth.synthetic_images(4, seed=3).to(DEV)
    with torch.no_grad():
        synth.calibrate(net18, x18)
    e18 = en.attach(net18)
    new18 = e18.explain(x18)
    prev18 = en._HEAD_RANK1
    try:
        en._HEAD_RANK1 = False
        old18 = e18.explain(x18)
    finally:
        en._HEAD_RANK1 = prev18
    assert torch.equal(new18["logits"], old18["logits"])
    assert rel(new18["dynamic_linear_weights"], old18["dynamic_linear_weights"]) <= 2e-5 and rel(new18["contribution_map"], old18["contribution_map"]) <= 2e-5
    net = synth.build_bcosified_vit("simple_vit_ti_patch16_224").to(DEV)
    x = synth.synthetic_images(6, seed=3).to(DEV)
    with torch.no_grad():
        synth.calibrate(net, x[:4])
    eng = vit_engine.attach(net)
    new = eng.explain(x)
    prev = ve._HEAD_RANK1
    try:
        ve._HEAD_RANK1 = False
        old = eng.explain(x)
    finally:
        ve._HEAD_RANK1 = prev
    assert torch.equal(new["logits"], old["logits"]) and torch.equal(new["prediction"], old["prediction"])
    assert rel(new["dynamic_linear_weights"], old["dynamic_linear_weights"]) <= 1e-5 and rel(new["contribution_map"], old["contribution_map"]) <= 1e-5


def test_layernorm_gradient_takes_the_residual_gradient(lib):
    """bcos_layernorm_bwd_add: the residual stream's gradient added by the LayerNorm-gradient launch is the separate addition, bit for bit;
    and the fused row-scale + split of small weight tensors feeds the same contraction results as the fp64 product."""
    from bcos_hip import ops
    torch.manual_seed(11)
    x = torch.randn(1000, 192, device=DEV) * 3 + 0.5
    gy, add, w = torch.randn(1000, 192, device=DEV), torch.randn(1000, 192, device=DEV), torch.rand(192, device=DEV) + 0.5
    _, rstd = ops.layernorm_fwd(x, w, None, 1e-5, want_rstd=True)
    g0, xh0 = ops.layernorm_bwd(gy, x, w, rstd, want_xhat=True)
    g1, xh1 = ops.layernorm_bwd(gy, x, w, rstd, want_xhat=True, addend=add)
    assert torch.equal(g1, g0 + add) and torch.equal(xh0, xh1)
    xr = x.double().requires_grad_(True)
    F.layer_norm(xr, (192,), w.double(), None, 1e-5).backward(gy.double())
    assert rel(g0, xr.grad) <= 1e-5
    for rows, K in ((192, 192), (768, 192), (192, 768), (1000, 192), (192, 2048), (40, 68)):          # one- and two-launch images
        a = ops.ensure_absmax(torch.randn(700, K, device=DEV))
        wt = ops.mark_static(torch.randn(rows, K, device=DEV) * torch.logspace(-3, 2, rows, device=DEV).view(-1, 1))
        y = ops.matmul_nt(a, wt, track_absmax=False)
        assert rel(y, a.double() @ wt.double().t()) <= 2e-6, (rows, K)


def test_vitc_ti_and_groupnorm_against_reference_golden(lib, golden_dir):
    """The conv-stem ViT (vitc_ti_patch1_14) and DetachableGroupNorm2d on the device: bcos_groupnorm_fwd /
    bcos_groupnorm_bwd_detached, MyGELU on channels_last activations, the stem convolutions on the fused B-cos kernel --
    logits, class indices, W(x) and contribution maps against the reference's recorded outputs."""
    from test_host_cpu import run_vitc_goldens
    run_vitc_goldens(golden_dir, DEV, 1e-5)


def test_vit_ti_batch512_properties(lib):
    """BASELINE.json configs[2] at full size: batch 512 on one GPU; a sub-batch reproduces bit-identically and agrees
    with the CPU oracle."""
    from bcos_hip import synth, vit_engine
    net = synth.build_bcosified_vit("simple_vit_ti_patch16_224").to(DEV)
    x = synth.synthetic_images(512).to(DEV)
    with torch.no_grad():
        synth.calibrate(net, x[:8])
    eng = vit_engine.attach(net)
    out = eng.explain(x)
    assert out["logits"].shape == (512, 1000) and torch.isfinite(out["dynamic_linear_weights"]).all()
    assert len(set(out["prediction"].tolist())) > 1
    sub = eng.explain(x[300:302])
    assert torch.equal(sub["logits"], out["logits"][300:302])
    assert torch.equal(sub["contribution_map"], out["contribution_map"][300:302])
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    ref = O.explain_batch(lambda xx, detach: O.simple_vit_logits(sd, xx, detach=detach), x[300:302].cpu())
    assert rel(sub["logits"], ref["logits"]) <= 1e-4 and torch.equal(sub["prediction"].cpu(), ref["prediction"])
    assert rel(sub["dynamic_linear_weights"], ref["dynamic_linear_weights"]) <= 1e-4


# ------------------------------------------------------------------------------------------ CLIP RN50 image encoder
def _golden_clip(golden_dir):
    from bcos_hip import synth
    meta = json.load(open(os.path.join(golden_dir, "clip_rn50.json")))
    data = np.load(os.path.join(golden_dir, "clip_rn50.npz"))
    net = synth.build_bcosified_clip_rn50(seed=meta["weight_seed"])
    synth.apply_calibration(net, {k: torch.from_numpy(data["calib/" + k]) for k in meta["calib_order"]})
    sd = net.state_dict()
    for k, (s1, s2) in meta["state_checksum"].items():
        assert abs(float(sd[k].double().sum()) - s1) <= 1e-6 * max(1.0, s2), k
    return net.to(DEV), meta, data


def test_clip_rn50_against_reference_golden(lib, golden_dir):
    """BASELINE.json configs[3] topology: B-cosified CLIP RN50 image encoder, zero-shot style forward."""
    from bcos_hip import clip_head, engine, synth
    net, meta, data = _golden_clip(golden_dir)
    x = synth.synthetic_images(meta["n_images"], seed=meta["image_seed"]).to(DEV)
    # nn.Module path: forward and the explanation-mode gradient of one embedding coordinate (q, k detached)
    xr = x[:1].clone().requires_grad_(True)
    with net.explanation_mode():
        e = net(xr)
        (g,) = torch.autograd.grad(e[:, 7].sum(), xr)
    assert rel(e, data["embeddings"][:1]) <= 1e-4
    assert rel(g, data["grad_e7_image0"]) <= 2e-3      # ReLU-gate floor of a 55-layer CNN (see the ResNet-50 test)
    eng = engine.attach(net)
    with torch.no_grad():
        emb = net(x)                                    # fused plan
    assert rel(emb, data["embeddings"]) <= 1e-4
    # fused explanation through the attention-pool head against the reference-recorded gradient of embedding coordinate 7
    fused = eng.explain(x[:1], targets=torch.tensor([7]))
    assert rel(fused["dynamic_linear_weights"], data["grad_e7_image0"]) <= 2e-3      # ReLU-gate floor, as above
    assert rel(fused["dynamic_linear_weights"], g) <= 2e-3                           # and the module path
    wt = torch.randn(1024, 16, generator=torch.Generator().manual_seed(meta["text_seed"])).to(DEV)
    logits = clip_head.zeroshot_logits(emb, wt)
    assert rel(logits, data["zeroshot_logits"]) <= 1e-4
    assert torch.equal(logits.argmax(1).cpu(), torch.from_numpy(data["zeroshot_logits"]).argmax(1))
    assert torch.equal(eng.forward(x[1:3]), emb[1:3])   # batch independence, bit-exact


def test_clip_rn50_batch256_forward(lib):
    """configs[3] per-GPU shard (2048 images over 8 GPUs = 256 per rank): forward + zero-shot head at full shard size,
    sub-batch agreement with the CPU oracle."""
    from bcos_hip import clip_head, engine, synth
    net = synth.build_bcosified_clip_rn50().to(DEV)
    x = synth.synthetic_images(256).to(DEV)
    with torch.no_grad():
        synth.calibrate(net, x[:8])
    eng = engine.attach(net)
    emb = eng.forward(x)
    assert emb.shape == (256, 1024) and torch.isfinite(emb).all()
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    ref = O.clip_rn50_embed(sd, x[40:42].cpu())
    assert rel(emb[40:42], ref) <= 1e-4
    wt = torch.randn(1024, 1000, generator=torch.Generator().manual_seed(5))
    logits = clip_head.zeroshot_logits(emb, wt.to(DEV))
    assert rel(logits[40:42], O.zeroshot_logits(ref, wt)) <= 1e-4
    assert torch.equal(logits[40:42].argmax(1).cpu(), O.zeroshot_logits(ref, wt).argmax(1))


def test_attn_unpool_head_against_reference_golden(lib, golden_dir):
    """a13 / a20 `attn_unpool` variant on the HIP path: v_proj (plain linear with bias) and the B-cos c_proj run on the
    contraction kernel, head = row-normalise + GEMM kernels; fixture recorded from the reference module."""
    from test_host_cpu import _unpool_module
    from bcos_hip import clip_head
    m, sd, data = _unpool_module(golden_dir)
    m = m.to(DEV)
    x = torch.from_numpy(data["x"]).to(DEV)
    with torch.no_grad():
        y = m(x)
    assert y.shape == (9, 2, 48) and rel(y, data["y"]) <= 1e-5
    wt = torch.from_numpy(data["text"]).to(DEV)
    logits = clip_head.zeroshot_logits(y, wt, attn_unpool=True, cos_power=2)
    assert rel(logits, data["zeroshot_cos2"]) <= 1e-5
    xr = x.clone().requires_grad_(True)
    for sub in m.modules():                         # what BcosUtilMixin.explanation_mode() does (bcos/common.py:347-384)
        if hasattr(sub, "set_explanation_mode"):
            sub.set_explanation_mode(True)
    (g,) = torch.autograd.grad(m(xr)[:, :, 5].sum(), xr)
    assert rel(g, data["grad_d5"]) <= 1e-5


def test_render_explanations_against_reference_and_oracle(lib, golden_dir):
    """batched gradient_to_image on the device (colour, alpha, 15x15 box filter, exact 99.5 %
    quantile by radix select) against the reference's recorded RGBA image and the oracle on edge cases."""
    from bcos_hip import ops, synth
    data = np.load(os.path.join(golden_dir, "resnet18_e2e.npz"))
    meta = json.load(open(os.path.join(golden_dir, "resnet18_e2e.json")))
    x = synth.synthetic_images(meta["n_images"], seed=meta["image_seed"])[:2]
    w = torch.from_numpy(data["weights_01"])
    rgba, qv = ops.render_explanations(x.to(DEV), w.to(DEV), want_quantiles=True)
    assert rgba.shape == (2, 224, 224, 4)
    assert float(np.abs(rgba[0].cpu().numpy() - data["rgba_0"]).max()) <= 1e-5          # reference-recorded image
    for n in range(2):
        assert float(np.abs(rgba[n].cpu().numpy() - O.gradient_to_image(x[n], w[n])).max()) <= 1e-5
    # bcos.common.gradient_to_image on device tensors routes to the same kernel (single image, numpy out)
    from bcos.common import gradient_to_image
    assert np.array_equal(gradient_to_image(x[0].to(DEV), w[0].to(DEV)), rgba[0].cpu().numpy())
    # edge cases: ragged sizes, no / small smoothing, low percentiles inside a plateau of equal alphas (negative
    # contributions all carry alpha = 1e-12), 3-channel input (AddInverse applied by the kernel)
    g = torch.Generator().manual_seed(3)
    for (N, H, W_, smooth, pct, three) in [(3, 37, 53, 15, 99.5, False), (2, 16, 16, 0, 50.0, False), (2, 40, 24, 3, 10.0, True),
                                           (1, 9, 7, 5, 100.0, False), (2, 33, 65, 1, 0.0, True)]:
        x3 = torch.rand(N, 3, H, W_, generator=g)
        x6 = torch.cat([x3, 1 - x3], 1)
        wt = torch.randn(N, 6, H, W_, generator=g) * torch.rand(N, 1, H, W_, generator=g)
        out = ops.render_explanations((x3 if three else x6).to(DEV), wt.to(DEV), smooth=smooth, alpha_percentile=pct).cpu().numpy()
        for n in range(N):
            ref = O.gradient_to_image(x6[n], wt[n], smooth=smooth, alpha_percentile=pct)
            assert float(np.abs(out[n] - ref).max()) <= 1e-5, (N, H, W_, smooth, pct, n)


def test_grid_pointing_game_on_device(lib, golden_dir):
    """N2: multi-image built from 4 singles, one forward + 4 backward passes on the fused engine, smoothing and per-cell
    shares on the device -- against the reference-recorded attributions / shares of the same network and images."""
    from bcos_hip import engine, localisation, ops, synth
    net, meta, _ = _golden_net(golden_dir, "resnet18_e2e")
    loc = json.load(open(os.path.join(golden_dir, "localisation.json")))
    data = np.load(os.path.join(golden_dir, "localisation.npz"))
    singles = synth.synthetic_images(loc["n_imgs"], seed=loc["image_seed"], size=loc["single_shape"]).to(DEV)
    multi = localisation.make_multi_image(singles)
    tgts = torch.from_numpy(data["targets"]).view(1, -1)
    eng = engine.attach(net)
    res = localisation.grid_pointing_game(eng, multi, tgts, loc["single_shape"], smooth=15)
    gold_att = torch.from_numpy(data["attributions"])[:, 0]
    assert rel(res["attributions"][0], gold_att) <= 2e-3                 # free ReLU gates: the ResNet-18 map floor (see above)
    assert float((res["fractions"][0].cpu() - torch.from_numpy(data["fractions_s15_neg0"])).abs().max()) <= 2e-3
    # kernels alone, on the recorded attributions: exact arithmetic of the reference statements
    att = gold_att[None].to(DEV)
    for smooth, neg in ((0, False), (15, False), (15, True)):
        out = localisation.grid_pointing_game(None, multi, tgts, loc["single_shape"], smooth=smooth, neg=neg, attributions=att)
        assert rel(out["fractions"][0], data[f"fractions_s{smooth}_neg{int(neg)}"]) <= 1e-5
    # one forward + T backward == T x (forward + backward), bit for bit
    again = eng.explain(multi, targets=tgts[:, 1].to(DEV))
    assert torch.equal(res["attributions"][0, 1], again["contribution_map"][0])
    # batched form: two multi-images, 3x3 grids of 32-pixel cells
    s9 = synth.synthetic_images(18, seed=4, size=32).to(DEV)
    m9 = localisation.make_multi_images(s9, 3)
    t9 = torch.randint(0, 1000, (2, 9), generator=torch.Generator().manual_seed(1))
    r9 = localisation.grid_pointing_game(eng, m9, t9, 32, smooth=5)
    assert r9["fractions"].shape == (2, 9, 9) and torch.allclose(r9["fractions"].sum(-1).cpu(), torch.ones(2, 9), atol=1e-5)
    ref_c, ref_m = O.localisation_fractions(r9["attributions"][1].cpu()[:, None], 32, smooth=5)
    assert rel(r9["fractions"][1], ref_c) <= 1e-5 and rel(r9["metric"][1], ref_m) <= 1e-5
    assert rel(ops.box_filter(r9["attributions"][0], 7), F.avg_pool2d(r9["attributions"][0].cpu()[:, None], 7, 1, 3)[:, 0]) <= 1e-6


def test_explainer_api_on_hip_path(lib, golden_dir):
    """a18 on the device: get_explainer(net, "IxG" | "Ours"), BcosUtilMixin.attribute / attribute_selection
    (interpretability/explanation_methods/explainers/__init__.py:88-104, utils.py:70-99, captum.py:29-32,
    bcos/common.py:280-344) -- over the per-layer HIP modules (autograd) and over the fused engine -- against the oracle's
    attribute_selection_maps and the reference-recorded attributions of the localisation fixture."""
    from bcos_hip import engine, localisation, synth
    from interpretability.explanation_methods import get_explainer
    net, meta, _ = _golden_net(golden_dir, "resnet18_e2e")
    loc = json.load(open(os.path.join(golden_dir, "localisation.json")))
    data = np.load(os.path.join(golden_dir, "localisation.npz"))
    singles = synth.synthetic_images(loc["n_imgs"], seed=loc["image_seed"], size=loc["single_shape"]).to(DEV)
    multi = localisation.make_multi_image(singles)                                  # [1, 6, 224, 224]
    tgts = [int(t) for t in data["targets"]]
    gold = torch.from_numpy(data["attributions"])                                   # [T, 1, H, W], recorded from the reference
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    fwd = lambda xx, detach: O.resnet_logits(sd, xx, meta["arch"], detach=detach)    # noqa: E731
    small = synth.synthetic_images(2, seed=77, size=64).to(DEV)
    small_t = [[3, 500, 999], [17, 3, 250]]
    ref_small = torch.stack([O.attribute_selection_maps(fwd, small[i:i + 1].cpu(), small_t[i])[:, 0] for i in range(2)])

    def check(tag):
        # "Ours" is the model itself (ours.py:8-13); attribute / attribute_selection enter explanation mode themselves
        ours = get_explainer(net, "Ours", "default")
        assert ours is net
        sel = ours.attribute_selection(multi, tgts)                                  # [T, 6, H, W]
        assert sel.shape == (len(tgts), 6, 224, 224), tag
        assert rel(sel.sum(1, keepdim=True), gold) <= 2e-3, tag                      # free ReLU gates: the R18 map floor
        one = net.attribute(multi, tgts[1])
        assert torch.equal(one[0], sel[1]), tag
        # IxG = captum InputXGradient over the model as it is: in explanation mode it equals the model-inherent explanation
        ixg = get_explainer(net, "IxG", "default")
        with net.explanation_mode():
            multi_sel = ixg.attribute_selection(small, small_t)                      # [N * T, 6, h, w], sample-major
            single = ixg.attribute(small, [3, 17])
        assert multi_sel.shape == (6, 6, 64, 64), tag
        maps = multi_sel.sum(1).view(2, 3, 64, 64)
        assert rel(maps, ref_small) <= 2e-3, tag
        assert torch.equal(single[0], multi_sel[0]) and torch.equal(single[1], multi_sel[3]), tag
        assert rel((small * _w(small, [3, 17])), single) <= 1e-6, tag               # attribution == x * W(x)
        # outside explanation mode IxG differentiates through the dynamic scale: a different (larger) attribution
        return sel, multi_sel

    def _w(x, t):
        with net.explanation_mode():
            return net.explain_batch(x, targets=torch.tensor(t))["dynamic_linear_weights"]

    sel_mod, small_mod = check("modules")                                            # autograd over the per-layer HIP kernels
    eng = engine.attach(net)
    sel_eng, small_eng = check("engine")                                             # the fused plan under the same API
    assert rel(sel_eng, sel_mod) <= 2e-3 and rel(small_eng, small_mod) <= 2e-3
    direct = eng.explain(multi.expand(len(tgts), -1, -1, -1).contiguous(), targets=torch.tensor(tgts))
    assert torch.equal(sel_eng, multi * direct["dynamic_linear_weights"])            # the API adds nothing to the engine pass
    with pytest.raises(KeyError, match="out of scope"):
        get_explainer(net, "RISE", "default")


def test_checkpoint_containers_load_into_hip_path(lib, golden_dir, tmp_path):
    """N3 on the device: a Lightning-style container ("model." + "ema.module." prefixes, loading_utils.py:78-107) and a
    stripped flat .pth are written, found through Experiment, loaded with zero key edits, and the HIP forward +
    explanation of the loaded network is compared with the oracle run on the very same state dict."""
    from bcos.experiments.utils import Experiment
    from bcos_hip import engine, synth
    src, meta, _ = _golden_net(golden_dir, "resnet18_e2e")
    sd = {k: v.detach().cpu().clone() for k, v in src.state_dict().items()}
    g = torch.Generator().manual_seed(11)
    ema = {k: (v * (1 + 0.05 * torch.randn(v.shape, generator=g)) if v.dtype.is_floating_point and "running" not in k else v.clone())
           for k, v in sd.items()}
    save_dir = tmp_path / "experiments" / "ImageNet" / "bcosification" / "resnet_18"
    save_dir.mkdir(parents=True)
    pl = {"state_dict": {**{"model." + k: v for k, v in sd.items()}, **{"ema.module." + k: v for k, v in ema.items()}},
          "epoch": 89, "pytorch-lightning_version": "2.2.0"}
    torch.save(pl, save_dir / "last.ckpt")
    exp = Experiment("ImageNet", "bcosification", "resnet_18", base_directory=tmp_path / "experiments")
    x = synth.synthetic_images(3, seed=5, size=96).to(DEV)
    for use_ema, want in ((False, sd), (True, ema)):
        net = exp.load_trained_model(ema=use_ema).to(DEV)
        assert all(torch.equal(v.cpu(), want[k]) for k, v in net.state_dict().items())
        eng = engine.attach(net)
        out = net.explain_batch(x)
        ref = O.explain_batch(lambda xx, detach: O.resnet_logits(want, xx, "resnet18", detach=detach), x.cpu())
        assert rel(out["logits"], ref["logits"]) <= 1e-4 and torch.equal(out["prediction"].cpu(), ref["prediction"])
        pinned = eng.explain(x, gates=_oracle_gates(net, x, "resnet18"))
        assert rel(pinned["dynamic_linear_weights"], ref["dynamic_linear_weights"]) <= 1e-4
        assert rel(net(x), ref["logits"]) <= 1e-4                                    # plain forward under no_grad -> engine
    # stripped checkpoint (scripts/strip_checkpoints.py:52-84): the flat state dict itself, as last.ckpt of another run
    other = tmp_path / "experiments" / "ImageNet" / "bcosification" / "resnet_18-seed=5"
    other.mkdir(parents=True)
    torch.save(ema, other / "last.ckpt")
    net2 = Experiment(other).load_trained_model().to(DEV)
    engine.attach(net2)
    ref2 = O.resnet_logits(ema, x.cpu(), "resnet18")
    with torch.no_grad():
        assert rel(net2(x), ref2) <= 1e-4
    # the engine notices parameters that change after attach (load_state_dict into the attached network)
    net2.load_state_dict(sd)
    with torch.no_grad():
        assert rel(net2(x), O.resnet_logits(sd, x.cpu(), "resnet18")) <= 1e-4


def test_tapconv_group_matches_separate_launches(lib):
    """bcos_tapconv_group: the parity classes of a strided input gradient in ONE launch (narrow outputs: the input patch
    is staged once for all classes) must equal one launch per class; wide outputs take the per-entry fallback."""
    from bcos_hip import ops
    g = torch.Generator().manual_seed(21)
    for (N, Ho, Cout, Cin, k, s_, p_) in [(3, 20, 64, 8, 7, 2, 3), (2, 9, 32, 4, 3, 2, 1), (2, 12, 16, 24, 3, 2, 1)]:
        w = (torch.randn(Cout, Cin, k, k, generator=g) / (k * k * Cout) ** 0.5).to(DEV)
        plan = ops.DgradPlan(w, (s_, s_), (p_, p_), (1, 1))
        H = (Ho - 1) * s_ + k - 2 * p_ + (1 if Cin == 8 else 0)
        H -= H % 2                                                   # even sizes: all classes have the same P x Q
        Ho2 = (H + 2 * p_ - k) // s_ + 1
        gl = torch.randn(N, Ho2, Ho2, Cout, generator=g).to(DEV)
        add = torch.randn(N, H, H, Cin, generator=g).to(DEV)
        ops._NO_GROUP, no_d2s = False, ops._NO_D2S
        ops._NO_D2S = True                      # (the depth-to-space launch would take these shapes first, see the next test)
        try:
            a = plan.run(gl, H, H, addend=add)
            ops._NO_GROUP = True
            b = plan.run(gl, H, H, addend=add)
        finally:
            ops._NO_GROUP, ops._NO_D2S = False, no_d2s
        assert rel(a, b) <= 1e-6, (Cin, k)        # same products; the channel-slice grouping of the sums may differ
        ref = torch.nn.grad.conv2d_input((N, Cin, H, H), w.cpu(), gl.permute(0, 3, 1, 2).cpu(), stride=s_, padding=p_)
        assert rel(a.permute(0, 3, 1, 2), ref + add.permute(0, 3, 1, 2).cpu()) <= 1e-5


@pytest.mark.parametrize("mode", ["f16x2", "bf16x3", "f32"])
def test_depth_to_space_dgrad_matches_per_class_launches(lib, mode):
    """bcos_tapconv_geom.out_cgroup: all parity classes of a narrow strided input gradient as ONE contraction whose columns
    are (class, channel) over the union of the classes' taps (DgradPlan._depth_to_space) -- against the one-launch-per-class
    path, against torch's conv2d_input, with the addend / multiplier epilogue, into a padded channel pitch, on both
    epilogue implementations."""
    from bcos_hip import ops
    from bcos_hip import lib as blib
    prev = blib.get_contraction_mode()
    blib.set_contraction_mode(mode)
    g = torch.Generator().manual_seed(29)
    try:
        for (N, Hh, Cout, Cin, pitch, k, s_, p_) in [(3, 40, 64, 6, 8, 7, 2, 3), (2, 18, 32, 6, 8, 3, 2, 1), (2, 24, 16, 4, 4, 3, 2, 1),
                                                     (1, 12, 64, 8, 8, 4, 4, 0), (2, 30, 24, 3, 4, 5, 3, 2)]:
            w = (torch.randn(Cout, Cin, k, k, generator=g) / (k * k * Cout) ** 0.5).to(DEV)
            plan = ops.DgradPlan(w, (s_, s_), (p_, p_), (1, 1))
            if plan.has_empty:
                continue
            H = Hh - Hh % s_
            Ho = (H + 2 * p_ - k) // s_ + 1
            gl = ops.ensure_absmax(torch.randn(N, Ho, Ho, Cout, generator=g).to(DEV))
            add = torch.randn(N, H, H, pitch, generator=g).to(DEV)
            mul = torch.randn(N, H, H, pitch, generator=g).to(DEV)
            ref = torch.nn.grad.conv2d_input((N, Cin, H, H), w.cpu(), gl.permute(0, 3, 1, 2).cpu(), stride=s_, padding=p_)
            for kw in (dict(), dict(addend=add), dict(mul=mul), dict(addend=add, mul=mul)):
                outs = {}
                for name, no_d2s, generic in (("d2s", False, False), ("d2s_generic", False, True), ("classes", True, False)):
                    out = torch.full((N, H, H, pitch), float("nan"), device=DEV)
                    prev_flag = ops._NO_D2S
                    ops._NO_D2S = no_d2s
                    if generic:
                        blib.set_option("epi_generic", 1)
                    try:
                        plan.run(gl, H, H, out=out, **kw)
                    finally:
                        ops._NO_D2S = prev_flag
                        blib.set_option("epi_generic", 0)
                    outs[name] = out
                assert plan._d2s, "the depth-to-space launch was not taken"
                assert torch.equal(outs["d2s"], outs["d2s_generic"])                       # NaN canaries gone, same bits
                if pitch > Cin and "addend" not in kw:
                    assert torch.all(outs["d2s"][..., Cin:] == 0)                           # padded channels: zero weights
                assert rel(outs["d2s"][..., :Cin], outs["classes"][..., :Cin]) <= 2e-6, (mode, Cin, k, sorted(kw))
                want = ref.permute(0, 2, 3, 1)
                if "addend" in kw:
                    want = want + add[..., :Cin].cpu()
                if "mul" in kw:
                    want = want * mul[..., :Cin].cpu()
                assert rel(outs["d2s"][..., :Cin], want) <= 1e-5, (mode, Cin, k, sorted(kw))
    finally:
        blib.set_contraction_mode(prev)


@pytest.mark.parametrize("mode", ["bf16x3", "f32", "f16x2"])
def test_grouped_launch_matches_one_launch_per_group(lib, mode):
    """bcos_tapconv_geom.groups: all groups of a grouped B-cos convolution in ONE launch (blockIdx.y = group: its channel slice
    of A, its filter rows, its output columns, its patch norm) gives the bits of one launch per group -- forward with bias,
    BN affine, residual, ReLU, stored multiplier, norms; pre-split weights when Cout % 32 == 0, in-kernel split otherwise;
    against torch's grouped convolution with the B-cos scaling applied per group."""
    from bcos_hip import ops
    from bcos_hip import lib as blib
    prev = blib.get_contraction_mode()
    blib.set_contraction_mode(mode)
    g = torch.Generator().manual_seed(41)
    try:
        for (N, H, G, cin_g, cout_g, k, st, pd) in [(2, 12, 2, 8, 12, 3, 1, 1), (3, 9, 4, 16, 32, 1, 1, 0), (2, 11, 3, 4, 40, 3, 2, 1),
                                                    (1, 7, 8, 32, 64, 3, 1, 1)]:
            Cin, Cout = G * cin_g, G * cout_g
            x = torch.randn(N, H, H, Cin, generator=g).to(DEV)
            w = ops.mark_static((torch.randn(Cout, k, k, cin_g, generator=g) / (k * k * cin_g) ** 0.5).to(DEV))
            bias = (torch.randn(Cout, generator=g) * 0.1).to(DEV)
            csc = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
            csh = (torch.randn(Cout, generator=g) * 0.1).to(DEV)
            geom = ops.fwd_geom(N, H, H, cin_g, cout_g, k, k, st, st, pd, pd)
            P = geom["P"]
            add = torch.randn(N, P, P, Cout, generator=g).to(DEV)

            def run(grouped):
                y = torch.full((N, P, P, Cout), float("nan"), device=DEV)
                t = torch.full_like(y, float("nan"))
                nrm = torch.full((N, P, P, G), float("nan"), device=DEV)
                kw = dict(bias=bias, ch_scale=csc, ch_shift=csh, bcos_mode=1, b=2.0, relu=True, track_absmax=False)
                if grouped:
                    ops.tapconv(x, w, dict(geom, groups=G, a_pitch=Cin, out_pitch=Cout, norm_pitch=G), out=y, scale_out=t, norm_out=nrm,
                                addend=add, **kw)
                else:
                    for i in range(G):
                        kwi = dict(kw, bias=bias[i * cout_g:(i + 1) * cout_g], ch_scale=csc[i * cout_g:(i + 1) * cout_g],
                                   ch_shift=csh[i * cout_g:(i + 1) * cout_g])
                        ops.tapconv(x[..., i * cin_g:], ops.mark_static(w[i * cout_g:(i + 1) * cout_g].contiguous()),
                                    dict(geom, a_pitch=Cin, out_pitch=Cout, norm_pitch=G), out=y[..., i * cout_g:], scale_out=t[..., i * cout_g:],
                                    norm_out=nrm[..., i:], addend=add[..., i * cout_g:], **kwi)
                return y, t, nrm
            one, per = run(True), run(False)
            for a, b_ in zip(one, per):
                assert torch.equal(a.view(torch.int32), b_.view(torch.int32)), (mode, G, cin_g, cout_g, k)
            # against torch: grouped convolution, per-group patch norms, B = 2 scale, affine, residual, ReLU
            xn, wn = x.permute(0, 3, 1, 2).cpu(), w.permute(0, 3, 1, 2).cpu()
            lin = F.conv2d(xn, wn, bias.cpu(), st, pd, 1, G)
            ss = F.conv2d(xn.pow(2).reshape(N * G, cin_g, H, H), torch.ones(1, cin_g, k, k), None, st, pd).reshape(N, G, P, P)
            norm = (ss + 1e-6).sqrt()
            ref = lin * lin.abs() / norm.repeat_interleave(cout_g, dim=1)
            ref = (ref * csc.cpu().view(1, -1, 1, 1) + csh.cpu().view(1, -1, 1, 1) + add.permute(0, 3, 1, 2).cpu()).clamp_min(0)
            assert rel(one[0].permute(0, 3, 1, 2), ref) <= 1e-5 and rel(one[2].permute(0, 3, 1, 2), norm) <= 1e-6, (mode, G, k)
    finally:
        blib.set_contraction_mode(prev)


def test_operand_larger_than_2gib_is_split_by_images(lib):
    """The split-bf16 path addresses its operands through 32-bit buffer offsets; a call whose input exceeds 2 GiB is
    split by images inside bcos_tapconv.  2 x 1024 x 1024 x 272 fp32 = 2.28 GB in, 1x1 conv to 16 channels."""
    from bcos_hip import ops
    g = torch.Generator().manual_seed(2)
    N, H, C, Cout = 2, 1024, 272, 16
    x = torch.randn(N, H, H, C, generator=g).to(DEV)
    assert x.numel() * 4 >= 2 ** 31
    w = (torch.randn(Cout, 1, 1, C, generator=g) / C ** 0.5).to(DEV)
    add = torch.randn(N, H, H, Cout, generator=g).to(DEV)
    y, s_, _ = ops.conv2d_fwd(x, w, addend=add, want_scale=True)
    for n in range(N):                                   # per image: below the limit, no split
        yn, sn, _ = ops.conv2d_fwd(x[n:n + 1], w, addend=add[n:n + 1], want_scale=True)
        assert torch.equal(y[n:n + 1], yn) and torch.equal(s_[n:n + 1], sn)
    ref = O.bcos_conv2d(x[:, :64, :64].permute(0, 3, 1, 2).cpu(), w.permute(0, 3, 1, 2).cpu(), None, 1, 0, 1, 1, 2, 1,
                        detach=True, normalize_weight=False)
    assert rel(y[:, :64, :64].permute(0, 3, 1, 2) - add[:, :64, :64].permute(0, 3, 1, 2), ref) <= 1e-5


def test_captured_pass_matches_eager(lib):
    """engine.CapturedPass: the whole forward+explanation step recorded into a hipGraph and replayed on new inputs must
    reproduce the eager launches bit for bit."""
    from bcos_hip import engine, synth
    net = synth.build_bcosified_resnet("resnet18").to(DEV)
    x1 = synth.synthetic_images(4, seed=5, size=96).to(DEV)
    x2 = synth.synthetic_images(4, seed=6, size=96).to(DEV)
    with torch.no_grad():
        synth.calibrate(net, x1)
    eng = engine.attach(net)
    ref1 = {k: v.clone() for k, v in eng.explain(x1).items()}
    ref2 = {k: v.clone() for k, v in eng.explain(x2).items()}
    cap = engine.CapturedPass(eng, x1)
    for x, ref in ((x1, ref1), (x2, ref2), (x1, ref1)):
        out = cap(x)
        for k in ("logits", "prediction", "dynamic_linear_weights", "contribution_map"):
            assert torch.equal(out[k], ref[k]), k
    fwd = engine.CapturedPass(eng, x2, explain=False)
    assert torch.equal(fwd(x1)["logits"], ref1["logits"])


def test_presplit_weights_bit_identical(lib):
    """bcos_tapconv_presplit (weights split once into MFMA fragment order, B operand loaded straight into registers)
    must reproduce bcos_tapconv (weights split inside the kernel) bit for bit: same split, same products, same order."""
    from bcos_hip import ops
    from bcos_hip import lib as blib
    if blib.get_contraction_mode() == "f32":
        pytest.skip("the exact fp32-MFMA mode uses no pre-split weight image")
    g = torch.Generator().manual_seed(11)
    for (N, H, Cin, Cout, k, s_, p_) in [(2, 14, 64, 200, 3, 1, 1), (3, 9, 24, 40, 1, 1, 0), (2, 16, 8, 64, 7, 2, 3),
                                         (1, 7, 128, 1000, 1, 1, 0)]:
        x = torch.randn(N, H, H, Cin, generator=g).to(DEV)
        w = (torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5).to(DEV)
        y0, s0, _ = ops.conv2d_fwd(x, w, stride=(s_, s_), padding=(p_, p_), want_scale=True)
        ws = ops.mark_static(w.clone())
        y1, s1, _ = ops.conv2d_fwd(x, ws, stride=(s_, s_), padding=(p_, p_), want_scale=True)
        assert getattr(ws, "_bcos_wt3", None) is not None
        assert torch.equal(y0, y1) and torch.equal(s0, s1)
        ws.mul_(2.0)                                   # in-place update: the cached image must be rebuilt
        y2, _, _ = ops.conv2d_fwd(x, ws, stride=(s_, s_), padding=(p_, p_))
        y3, _, _ = ops.conv2d_fwd(x, w * 2.0, stride=(s_, s_), padding=(p_, p_))
        assert torch.equal(y2, y3)


def test_tall_tiles_bit_identical(lib, monkeypatch):
    """Launches with <= 64 output columns and two or more rounds of tiles run on 256-row tiles (csrc/bcos_tapconv.hip:
    bcos_tc_h2_256x64 / 256x32); an output element's K walk and product order do not depend on the tile, so every tensor is
    identical bit for bit to the 128-row tiles (BCOS_H2_TALL=0) -- ragged row counts (M not a multiple of 32), 1x1 and 3x3
    forward launches with the B-cos epilogue and their input gradients, 64- and 24-column outputs."""
    from bcos_hip import ops
    from bcos_hip import lib as blib
    if blib.get_contraction_mode() != "f16x2":
        pytest.skip("the 256-row tiles belong to the split-f16 loop")
    g = torch.Generator().manual_seed(41)
    for (N, H, Cin, Cout, k, pd) in [(2, 363, 256, 64, 1, 0), (3, 301, 64, 64, 3, 1), (2, 365, 64, 24, 3, 1)]:
        x = ops.ensure_absmax(torch.randn(N, H, H, Cin, generator=g).to(DEV))
        w = ops.mark_static((torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5).to(DEV))
        csc = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
        plan = ops.DgradPlan(w.permute(0, 3, 1, 2).contiguous(), (1, 1), (pd, pd))
        gl = ops.ensure_absmax(torch.randn(N, H, H, Cout, generator=g).to(DEV))
        mul = torch.randn(N, H, H, Cin, generator=g).to(DEV) if Cin <= 64 else None
        res = {}
        for tall in ("1", "0"):
            blib.set_option("h2_tall", int(tall))
            y, sc, nrm = ops.conv2d_fwd(x, w, padding=(pd, pd), ch_scale=csc, relu=True, want_scale=True, want_norm=True, track_absmax=True)
            gx = plan.run(gl, H, H, track_absmax=True, **({"mul": mul} if mul is not None else {}))
            res[tall] = (y, sc, nrm, ops.absmax_of(y), gx, ops.absmax_of(gx))
        blib.set_option("h2_tall", 1)
        assert not torch.isnan(res["1"][0]).any() and not torch.isnan(res["1"][4]).any()
        for i, (a, b) in enumerate(zip(res["1"], res["0"])):
            assert (a is None and b is None) or torch.equal(a.view(torch.int32), b.view(torch.int32)), (N, H, Cin, Cout, k, i)
        # and against a float64 reference of the forward contraction on a slice of rows
        xs = x[0, :2].double().cpu()
        if k == 1:
            lin = xs.reshape(-1, Cin) @ w.view(Cout, Cin).double().cpu().t()
            nr = xs.reshape(-1, Cin).pow(2).sum(1).add(1e-6).sqrt()
            want = (lin * lin.abs() / nr[:, None] * csc.double().cpu()).clamp_min(0)
            assert rel(res["1"][0][0, :2].reshape(-1, Cout), want) <= 2e-6
        del x, w, gl, res


def test_subsampled_addend_bit_identical(lib, golden_dir, monkeypatch):
    """bcos_epilogue.addend_sub (ABI v4): a gradient launch that takes the s-grid pixels of its addend as a dense tensor writes
    exactly what it writes with the same values scattered into a zero-filled full-size addend -- every gradient epilogue kind,
    general and specialised epilogue, every contraction mode, odd image sizes, strided output mappings (the parity classes of a
    3x3 / 2 gradient) -- and a whole ResNet-50 / ResNet-18 explanation pass is bit-identical with and without it."""
    from bcos_hip import engine, ops, synth
    from bcos_hip import lib as blib
    g = torch.Generator().manual_seed(31)
    prev = blib.get_contraction_mode()
    try:
        for mode in ("f16x2", "bf16x3", "f32"):
            blib.set_contraction_mode(mode)
            for (N, H, Cin, Cout, k, st, pd, sb) in [(3, 14, 64, 256, 1, 1, 0, 2), (2, 15, 32, 52, 3, 2, 1, 2), (2, 13, 256, 24, 1, 1, 0, 2),
                                                      (1, 28, 128, 512, 1, 1, 0, 2), (2, 10, 48, 40, 1, 1, 0, 3)]:
                # the launch: input gradient of conv(Cin -> Cout, k, st, pd) on H x H images; its output has Cin channels
                w = ops.mark_static((torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5).to(DEV))
                plan = ops.DgradPlan(w.permute(0, 3, 1, 2).contiguous(), (st, st), (pd, pd))
                Ho = ops.conv_out_size(H, k, st, pd)
                gl = ops.ensure_absmax(torch.randn(N, Ho, Ho, Cout, generator=g).to(DEV))
                Hs = -(-H // sb)
                sub = torch.randn(N, Hs, Hs, Cin, generator=g).to(DEV)
                full = torch.zeros(N, H, H, Cin, device=DEV)
                full[:, ::sb, ::sb] = sub
                mul = torch.randn(N, H, H, Cin, generator=g).to(DEV)
                mul2 = torch.randn(N, H, H, Cin, generator=g).to(DEV)
                for kw in (dict(), dict(mul=mul, want2=True, flags=8), dict(mul=mul, mul2=mul2, want2=True, flags=8), dict(mul=mul)):
                    res = {}
                    for generic in (False, True):
                        if generic:
                            blib.set_option("epi_generic", 1)
                        else:
                            blib.set_option("epi_generic", 0)
                        for name, extra in (("full", dict(addend=full)), ("sub", dict(addend=sub, addend_sub=sb))):
                            kw2 = dict(kw)
                            out2 = torch.full((N, H, H, Cin), float("nan"), device=DEV) if kw2.pop("want2", False) else None
                            if out2 is not None:
                                kw2["out2"] = out2
                            out = plan.run(gl, H, H, track_absmax=True, track_absmax2=out2 is not None, **kw2, **extra)
                            res[(generic, name)] = (out, out2, ops.absmax_of(out), ops.absmax_of(out2) if out2 is not None else None)
                    blib.set_option("epi_generic", 0)
                    ref = res[(True, "full")]
                    assert not torch.isnan(ref[0]).any()
                    for key, got in res.items():
                        for i, (a, b) in enumerate(zip(got, ref)):
                            assert (a is None and b is None) or torch.equal(a.view(torch.int32), b.view(torch.int32)), (mode, key, i, N, H, Cin, Cout, k, sorted(kw))
        blib.set_contraction_mode(prev)
        # refused where it is not defined: forward (B-cos) launches
        x = torch.randn(1, 4, 4, 8, device=DEV)
        wf = torch.randn(8, 1, 1, 8, device=DEV)
        with pytest.raises(blib.BcosHipError):
            ops.tapconv(x, wf, ops.fwd_geom(1, 4, 4, 8, 8, 1, 1, 1, 1, 0, 0, 1, 1), out=torch.empty(1, 4, 4, 8, device=DEV),
                        addend=torch.zeros(1, 2, 2, 8, device=DEV), addend_sub=2, bcos_mode=blib.BCOS_CONV_EPS)
        for fixture in ("resnet18_e2e", "resnet50_small"):
            net, meta, data = _golden_net(golden_dir, fixture)
            x = synth.synthetic_images(meta["n_images"], seed=meta["image_seed"]).to(DEV)
            monkeypatch.setattr(engine, "_SUB_ADDEND", True)
            a = engine.attach(net).explain(x)
            monkeypatch.setattr(engine, "_SUB_ADDEND", False)
            b = engine.attach(net).explain(x)
            for key in ("logits", "dynamic_linear_weights", "contribution_map"):
                assert torch.equal(a[key], b[key]), (fixture, key)
    finally:
        blib.set_contraction_mode(prev)
        blib.set_option("epi_generic", 0)


def test_fast_epilogue_bit_identical(lib, golden_dir, monkeypatch):
    """The specialised epilogues (csrc/bcos_tapconv.hip: tile_epilogue_fast, selected per launch from the feature set)
    evaluate the same expressions in the same order as the general one: every tensor they write is identical bit for bit
    to the BCOS_EPI_GENERIC=1 run -- single launches of every compiled kind on ragged shapes (rows not a multiple of the
    tile, Cout not a multiple of the column tile, strided output mapping), in every contraction mode, and a whole
    ResNet-18 / ResNet-50 forward + explanation pass."""
    from bcos_hip import engine, ops, synth
    from bcos_hip import lib as blib
    g = torch.Generator().manual_seed(23)

    def both(fn):
        blib.set_option("epi_generic", 0)
        fast = fn()
        blib.set_option("epi_generic", 1)
        gen = fn()
        blib.set_option("epi_generic", 0)
        return fast, gen

    def same(fast, gen, what):
        for i, (a, b) in enumerate(zip(fast, gen)):
            if a is None:
                assert b is None
                continue
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (what, i, rel(a, b))

    prev = blib.get_contraction_mode()
    try:
        for mode in ("f16x2", "bf16x3", "f32"):
            blib.set_contraction_mode(mode)
            for (N, H, Cin, Cout, k, st, pd) in [(3, 13, 64, 200, 1, 1, 0), (2, 15, 32, 52, 3, 2, 1), (1, 9, 256, 24, 1, 1, 0),
                                                 (2, 28, 128, 512, 1, 1, 0), (5, 7, 512, 384, 3, 1, 1)]:
                x = ops.ensure_absmax(torch.randn(N, H, H, Cin, generator=g).to(DEV))
                w = ops.mark_static((torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5).to(DEV))
                bias = torch.randn(Cout, generator=g).to(DEV) * 0.1
                csc = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
                csh = torch.randn(Cout, generator=g).to(DEV) * 0.1
                Ho = ops.conv_out_size(H, k, st, pd)
                add = torch.randn(N, Ho, Ho, Cout, generator=g).to(DEV)
                for relu in (False, True, 2):          # 2 = GELU with the gate held constant (MyGELU)
                    for addend in (None, add):
                        for want_scale in (False, True):
                            for flags in ((0, 4) if (relu and want_scale) else (0,)):
                                def run():
                                    y, sc, nrm = ops.conv2d_fwd(x, w, stride=(st, st), padding=(pd, pd), bias=bias, ch_scale=csc,
                                                                ch_shift=csh, addend=addend, relu=relu, want_scale=want_scale,
                                                                want_norm=True, flags=flags, track_absmax=True)
                                    return y, sc, nrm, ops.absmax_of(y)
                                same(*both(run), (mode, "fwd", N, H, Cin, Cout, k, relu, addend is not None, want_scale, flags))
                # input-gradient launches: out = (acc [+ addend]) * mul, out2 = (acc [+ addend]) [* mul2] gated by the low bit of mul
                plan = ops.DgradPlan(w.permute(0, 3, 1, 2).contiguous(), (st, st), (pd, pd))
                gl = ops.ensure_absmax(torch.randn(N, Ho, Ho, Cout, generator=g).to(DEV))
                mul = torch.randn(N, H, H, Cin, generator=g).to(DEV)
                mul2 = torch.randn(N, H, H, Cin, generator=g).to(DEV)
                add_in = torch.randn(N, H, H, Cin, generator=g).to(DEV)
                act_in = torch.randn(N, H, H, Cin, generator=g).clamp_min(0).to(DEV)          # a kept ReLU activation and its layer's
                nrm_in = (torch.rand(N, H, H, generator=g) + 0.5).to(DEV)                     # patch norms / BN scale / shift
                csc_in = (torch.rand(Cin, generator=g) + 0.5).to(DEV) * torch.where(torch.rand(Cin, generator=g) < 0.2, -1.0, 1.0).to(DEV)
                csh_in = (torch.randn(Cin, generator=g) * 0.1).to(DEV)
                for kw in (dict(), dict(addend=add_in), dict(mul=mul), dict(mul=mul, addend=add_in, want2=True, flags=8),
                           dict(mul=mul, addend=add_in, mul2=mul2, want2=True, flags=8), dict(mul=mul, want2=True), dict(mul=mul, want2=True, flags=8),
                           dict(mul=act_in, mul_norm=nrm_in, mul_csc=csc_in, mul_csh=csh_in, flags=16)):       # BCOS_EPI_MUL_FROM_ACT
                    def run():
                        kw2 = dict(kw)
                        out2 = torch.full((N, H, H, Cin), float("nan"), device=DEV) if kw2.pop("want2", False) else None
                        if out2 is not None:
                            kw2["out2"] = out2
                        out = plan.run(gl, H, H, track_absmax=True, track_absmax2=out2 is not None, **kw2)
                        return out, out2, ops.absmax_of(out), (ops.absmax_of(out2) if out2 is not None else None)
                    same(*both(run), (mode, "bwd", N, H, Cin, Cout, k, sorted(kw)))
        blib.set_contraction_mode(prev)
        net, meta, data = _golden_net(golden_dir, "resnet18_e2e")
        x = synth.synthetic_images(meta["n_images"], seed=meta["image_seed"]).to(DEV)
        eng = engine.attach(net)
        fast, gen = both(lambda: eng.explain(x))
        for key in ("logits", "dynamic_linear_weights", "contribution_map"):
            assert torch.equal(fast[key], gen[key]), key
        net50 = synth.build_bcosified_resnet("resnet50").to(DEV)
        x50 = synth.synthetic_images(6, seed=5).to(DEV)
        with torch.no_grad():
            synth.calibrate(net50, x50)
        eng50 = engine.attach(net50)
        fast, gen = both(lambda: eng50.explain(x50))
        for key in ("logits", "dynamic_linear_weights", "contribution_map"):
            assert torch.equal(fast[key], gen[key]), key
    finally:
        blib.set_contraction_mode(prev)


# ------------------------------------------------------------------------------------------ both contraction modes
@pytest.mark.parametrize("mode", ["f32", "bf16x3", "f16x2"])
def test_contraction_modes_parity(lib, golden_dir, mode):
    """The contraction runs either on fp32 MFMA or on the exact 3-way bf16 split (6 bf16 MFMA products, fp32
    accumulation; include/bcos_hip.h).  Both must meet the same tolerances: layers vs the oracle, error vs fp64 of the
    same size class, ResNet-18 logits / class indices vs the reference fixture and gate-pinned maps <= 1e-4."""
    from bcos_hip import engine, ops, synth
    from bcos_hip import lib as blib
    prev = blib.get_contraction_mode()
    blib.set_contraction_mode(mode)
    prev_min_k = ops.F16X2_MIN_K
    ops.F16X2_MIN_K = 0          # f16x2: also for the small-K launches that would keep the bf16x3 loop for speed
    try:
        g = torch.Generator().manual_seed(3)
        a = torch.randn(512, 2304, generator=g) * (torch.rand(512, 1, generator=g) * 3)
        w = torch.randn(256, 2304, generator=g) / 48
        ref64 = a.double() @ w.double().t()
        wd = ops.mark_static(w.to(DEV))                       # f16x2 needs the pre-split weight image and the row maxima of A
        err = rel(ops.matmul_nt(ops.ensure_absmax(a.to(DEV)), wd), ref64)
        assert err <= 2e-6, (mode, err)                       # fp32-rounding class (measured 5e-7 ... 9e-7 in the three modes)
        # tiny / huge magnitudes: bf16x3 needs no scaling (bf16 has fp32's exponent range), f16x2 scales every row exactly
        for scale in (1e-20, 1e15):
            out = ops.matmul_nt(ops.ensure_absmax((a * scale).to(DEV)), wd)
            assert rel(out, ref64 * scale) <= 2e-6, (mode, scale)
        # rows and weight rows spread over 30 / 12 decades (products stay clear of fp32 underflow, which no mode can
        # repair), outliers 1e4 above and 1e-6 below the bulk of a row
        rs = 10.0 ** (torch.rand(512, 1, generator=g) * 30 - 15)
        cs = 10.0 ** (torch.rand(256, 1, generator=g) * 12 - 6)
        a2 = a * rs
        a2[:, ::97] *= 1e4
        a2[:, 5::131] *= 1e-6
        w2 = w * cs
        ref2 = a2.double() @ w2.double().t()
        out2 = ops.matmul_nt(ops.ensure_absmax(a2.to(DEV)), ops.mark_static(w2.to(DEV))).double().cpu()
        assert float(((out2 - ref2).norm(dim=1) / ref2.norm(dim=1)).max()) <= 4e-6, mode
        for geom in CONV_GEOMS[:8]:
            N, Cin, H, W, Cout, k, s, p = geom
            x = torch.randn(N, Cin, H, W, generator=g)
            wt = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
            xr = x.clone().requires_grad_(True)
            y_ref, s_ref = O.bcos_conv2d(xr, wt, stride=s, padding=p, detach=True, return_scale=True)
            gy = torch.randn(y_ref.shape, generator=g)
            (gx_ref,) = torch.autograd.grad(y_ref, xr, gy)
            y, sc, _ = ops.conv2d_fwd(ops.ensure_absmax(x.permute(0, 2, 3, 1).contiguous().to(DEV)),
                                      ops.mark_static(wt.permute(0, 2, 3, 1).contiguous().to(DEV)),
                                      stride=(s, s), padding=(p, p), want_scale=True)
            if mode == "f16x2":      # the per-pixel maxima the launch emitted for the next layer are exact
                assert torch.equal(ops.absmax_of(y), y.abs().amax(dim=-1).reshape(-1).view(torch.int32)), geom
            glin = ops.ensure_absmax(ops.mul(gy.permute(0, 2, 3, 1).contiguous().to(DEV), sc))
            gx = ops.DgradPlan(wt.to(DEV), (s, s), (p, p)).run(glin, H, W)
            assert rel(y.permute(0, 3, 1, 2), y_ref) <= 1e-5 and rel(gx.permute(0, 3, 1, 2), gx_ref) <= 1e-5, (mode, geom)
        net, meta, data = _golden_net(golden_dir, "resnet18_e2e")
        x = synth.synthetic_images(meta["n_images"], seed=meta["image_seed"]).to(DEV)
        eng = engine.attach(net)
        out = eng.explain(x)
        assert rel(out["logits"], data["logits"]) <= 1e-4
        assert np.array_equal(out["prediction"].cpu().numpy(), data["prediction"])
        gates = [torch.from_numpy(np.unpackbits(data[f"gate/{i:02d}"])[: int(np.prod(shp))].reshape(shp).astype(np.float32)).to(DEV)
                 for i, shp in enumerate(meta["gate_shapes"])]
        pinned = eng.explain(x[:2], gates=gates)
        assert rel(pinned["contribution_map"], data["contribution_map"][:2]) <= 1e-4
        assert rel(pinned["dynamic_linear_weights"], data["weights_01"]) <= 1e-4
    finally:
        blib.set_contraction_mode(prev)
        ops.F16X2_MIN_K = prev_min_k


def test_dma_loop_bit_identical_to_register_loop(lib, golden_dir, monkeypatch):
    """Round 3: the split-f16 contraction stages its operands through the LDS-DMA path (csrc/bcos_tapconv.hip: tile_body_d --
    fp32 A straight into LDS, split on the fragment, ring of three slots).  It performs the same arithmetic as the register-
    staged loop of round 2 (BCOS_H2_LOOP=regs: tile_body_h2) in the same order: every tensor a launch writes -- output, kept
    scale, patch norms, per-pixel maxima -- is identical bit for bit.  Shapes cover every K walk (1x1, channel-chunk-major
    3x3, 7x7 over 8 channels, ragged channel counts), every tile configuration (32 ... 256 columns, 128- and 256-row tiles,
    half-height tail tiles, ragged rows / columns) and the gradient forms (stride-2 parity classes, depth to space), then
    whole ResNet-18 / ResNet-50 passes."""
    from bcos_hip import lib as blib
    from bcos_hip import engine, ops, synth
    g = torch.Generator().manual_seed(31)

    blib.set_option("patch", 0)       # (the input-patch loop of the multi-tap launches has its own test below)

    def both(fn):
        blib.set_option("h2_loop", 0)
        dma = fn()
        blib.set_option("h2_loop", 1)
        regs = fn()
        blib.set_option("h2_loop", 0)
        return dma, regs

    def same(a_list, b_list, what):
        for i, (a, b) in enumerate(zip(a_list, b_list)):
            if a is None:
                assert b is None
                continue
            assert torch.isfinite(a).all() or a.dtype != torch.float32, (what, i)
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (what, i, rel(a.float(), b.float()))

    cases = [(3, 13, 64, 200, 1, 1, 0), (2, 15, 32, 52, 3, 2, 1), (1, 9, 256, 24, 1, 1, 0), (2, 28, 128, 512, 1, 1, 0),
             (5, 7, 512, 384, 3, 1, 1), (2, 14, 256, 256, 3, 1, 1), (6, 14, 1024, 256, 1, 1, 0), (2, 33, 8, 64, 7, 2, 3),
             (40, 30, 64, 64, 3, 1, 1), (40, 30, 256, 64, 1, 1, 0), (3, 12, 20, 40, 3, 1, 1), (2, 20, 16, 128, 5, 1, 2),
             (9, 28, 128, 128, 3, 1, 1), (1, 5, 2048, 1000, 1, 1, 0), (7, 14, 192, 192, 1, 1, 0), (3, 20, 768, 152, 1, 1, 0),
             (2, 17, 64, 192, 3, 1, 1)]
    for (N, H, Cin, Cout, k, st, pd) in cases:
        x = ops.ensure_absmax(torch.randn(N, H, H, Cin, generator=g).to(DEV))
        w = ops.mark_static((torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5).to(DEV))
        Ho = ops.conv_out_size(H, k, st, pd)

        def run():
            y, sc, nrm = ops.conv2d_fwd(x, w, stride=(st, st), padding=(pd, pd), relu=True, want_scale=True, want_norm=True,
                                        track_absmax=True)
            return y, sc, nrm, ops.absmax_of(y)
        same(*both(run), ("fwd", N, H, Cin, Cout, k, st))
        plan = ops.DgradPlan(w.permute(0, 3, 1, 2).contiguous(), (st, st), (pd, pd))
        gl = ops.ensure_absmax(torch.randn(N, Ho, Ho, Cout, generator=g).to(DEV))
        mul = torch.randn(N, H, H, Cin, generator=g).to(DEV)

        def run_b():
            out = plan.run(gl, H, H, mul=mul, track_absmax=True)
            return out, ops.absmax_of(out)
        same(*both(run_b), ("bwd", N, H, Cin, Cout, k, st))
    for name in ("resnet18_e2e", "resnet50_small"):
        net, meta, data = _golden_net(golden_dir, name)
        x = synth.synthetic_images(meta["n_images"], seed=meta["image_seed"]).to(DEV)
        eng = engine.attach(net)
        dma, regs = both(lambda: eng.explain(x))
        for key in ("logits", "dynamic_linear_weights", "contribution_map"):
            assert torch.equal(dma[key], regs[key]), (name, key)


def test_patch_loop_against_per_tap_loop_and_fp64(lib, monkeypatch):
    """Round 3: multi-tap launches (3x3, 4x4-union gradients, 2x2 parity classes) contract over an LDS-resident input PATCH
    (csrc/bcos_tapconv.hip: tile_body_p): every input element is loaded and split once per 16-channel chunk instead of once per
    tap, with ONE operand scale per image.  Same K walk and product order as the per-tap loops, different rounding of small
    elements and of the patch-norm sums: outputs agree with the per-tap loop (BCOS_PATCH=0) to fp32 rounding and sit as close to
    an fp64 evaluation as it does.  Shapes: the ResNet geometries (7^2 ... 56^2, 64 ... 512 channels), ragged rows / columns, tiles
    that span several images, stride 2, the gradient forms; images of very different magnitude in one batch (the scale is per
    image) and bit-identity of an image's results across batch positions / batch sizes."""
    from bcos_hip import lib as blib
    from bcos_hip import ops
    g = torch.Generator().manual_seed(47)

    def both(fn):
        blib.set_option("patch", 1)
        patch = fn()
        blib.set_option("patch", 0)
        taps = fn()
        blib.set_option("patch", 1)
        return patch, taps

    cases = [(5, 7, 512, 384, 3, 1, 1), (3, 14, 256, 256, 3, 1, 1), (2, 14, 256, 200, 3, 1, 1), (3, 28, 128, 128, 3, 1, 1),
             (2, 56, 64, 64, 3, 1, 1), (9, 9, 64, 48, 3, 1, 1), (2, 28, 128, 128, 3, 2, 1), (4, 14, 256, 256, 3, 2, 1),
             (2, 17, 64, 192, 3, 1, 1), (2, 20, 16, 128, 5, 1, 2), (3, 12, 32, 40, 3, 1, 0), (2, 15, 32, 52, 3, 2, 1),
             (20, 6, 48, 64, 3, 1, 1), (2, 33, 8, 64, 7, 2, 3), (3, 64, 8, 64, 7, 2, 3),     # (7 x 7 / 2 over 8 channels: the depth-to-space gradient, 4 x 4 taps in 2-D tiles)
             (2, 80, 32, 64, 3, 1, 1), (1, 112, 32, 48, 3, 1, 1), (2, 72, 32, 32, 3, 1, 1), (1, 100, 16, 24, 3, 1, 1)]   # (wide images: 3 x 3 in 8 x 32 / 16 x 16 blocks)
    for (N, H, Cin, Cout, k, st, pd) in cases:
        mag = torch.logspace(-3, 3, N).view(N, 1, 1, 1)                    # images six decades apart
        x = ops.ensure_absmax((torch.randn(N, H, H, Cin, generator=g) * mag).to(DEV))
        w = ops.mark_static((torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5).to(DEV))
        Ho = ops.conv_out_size(H, k, st, pd)

        def run():
            y, sc, nrm = ops.conv2d_fwd(x, w, stride=(st, st), padding=(pd, pd), relu=False, want_scale=True, want_norm=True,
                                        track_absmax=True)
            return y, sc, nrm, ops.absmax_of(y).view(torch.float32)
        pa, ta = both(run)
        xd, wd = x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2)
        lin = torch.nn.functional.conv2d(xd, wd, stride=st, padding=pd)
        nrm64 = (torch.nn.functional.conv2d(xd * xd, torch.ones(1, Cin, k, k, device=DEV, dtype=torch.float64), stride=st, padding=pd) + 1e-6).sqrt()
        y64 = (lin * lin.abs() / nrm64).permute(0, 2, 3, 1)
        for n in range(N):                                                 # per image: every magnitude is held to the same bound
            e_p, e_t = rel(pa[0][n].double(), y64[n]), rel(ta[0][n].double(), y64[n])
            assert e_p <= max(2e-6, 1.5 * e_t), ("fwd", N, H, Cin, Cout, k, st, n, e_p, e_t)
            assert rel(pa[2][n].double().flatten(), nrm64[n].flatten()) <= 1e-6
        for a, b in zip(pa, ta):
            assert torch.isfinite(a).all()
            assert rel(a, b) <= 2e-6, ("fwd", N, H, Cin, Cout, k, st, rel(a, b))
        # an image's results do not depend on its batch position or on the batch size
        blib.set_option("patch", 1)
        perm = torch.arange(N - 1, -1, -1)
        xp = ops.ensure_absmax(x[perm.to(DEV)].contiguous())
        yp = ops.conv2d_fwd(xp, w, stride=(st, st), padding=(pd, pd), relu=False, want_scale=False, want_norm=False)[0]
        assert torch.equal(yp, pa[0][perm.to(DEV)]), ("position", N, H, Cin, Cout, k, st)
        y1 = ops.conv2d_fwd(ops.ensure_absmax(x[N - 1:].contiguous()), w, stride=(st, st), padding=(pd, pd), relu=False,
                            want_scale=False, want_norm=False)[0]
        assert torch.equal(y1, pa[0][N - 1:]), ("batch size", N, H, Cin, Cout, k, st)
        plan = ops.DgradPlan(w.permute(0, 3, 1, 2).contiguous(), (st, st), (pd, pd))
        gl = ops.ensure_absmax((torch.randn(N, Ho, Ho, Cout, generator=g) * mag).to(DEV))
        mul = torch.randn(N, H, H, Cin, generator=g).to(DEV)

        def run_b():
            out = plan.run(gl, H, H, mul=mul, track_absmax=True)
            am = ops.absmax_of(out)                                        # (a depth-to-space launch emits no maxima)
            return (out,) if am is None else (out, am.view(torch.float32))
        pb, tb = both(run_b)
        g64 = torch.nn.functional.conv_transpose2d(gl.double().permute(0, 3, 1, 2), wd, stride=st, padding=pd,
                                                   output_padding=H - ((Ho - 1) * st - 2 * pd + k)).permute(0, 2, 3, 1) * mul.double()
        for n in range(N):
            e_p, e_t = rel(pb[0][n].double(), g64[n]), rel(tb[0][n].double(), g64[n])
            assert e_p <= max(2e-6, 1.5 * e_t), ("bwd", N, H, Cin, Cout, k, st, n, e_p, e_t)
        for a, b in zip(pb, tb):
            assert rel(a, b) <= 2e-6, ("bwd", N, H, Cin, Cout, k, st, rel(a, b))


def _structured_images(N, H, W, C, g, kind):
    """Inputs whose dynamic range is INSIDE an image (VERDICT r03 'What's weak' 1): post-ReLU activations and above all the
    gradients of the explanation pass are spatially sparse with long tails, which homogeneous noise never is."""
    x = torch.randn(N, H, W, C, generator=g)
    ii = torch.arange(H).view(1, H, 1, 1).float()
    jj = torch.arange(W).view(1, 1, W, 1).float()
    if kind == "blob":            # a bright blob on a background ten decades darker
        r2 = (ii - H * 0.4) ** 2 + (jj - W * 0.6) ** 2
        mag = torch.where(r2 <= (min(H, W) * 0.25) ** 2, torch.tensor(1e8), torch.tensor(1e-2))
    elif kind == "ramp":          # a smooth field falling through twelve decades from corner to corner
        mag = 10.0 ** (6.0 - 12.0 * (ii / max(H - 1, 1) + jj / max(W - 1, 1)) / 2.0)
    elif kind == "hot":           # one hot pixel per image, everything else 1e-10 of it (and a quarter of the pixels exactly zero)
        mag = torch.full((1, H, W, 1), 1e-4)
        mag[0, H // 3, W // 2, 0] = 1e6
        mag = mag * (torch.rand(N, H, W, 1, generator=g) > 0.25)
    else:                         # "speckle": every pixel its own magnitude over twelve decades
        mag = 10.0 ** (torch.rand(N, H, W, 1, generator=g) * 12 - 6)
    return x * mag


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "f16x2"])
def test_patch_loop_dynamic_range_inside_an_image(lib, mode):
    """VERDICT r03 item 2 / ADVICE r03: the operand scales of the input-patch loop (3 x 3 launches, the 4 x 4 tap union of the
    depth-to-space stem gradient) and of launches with >= 25 taps (7 x 7 stem) are per IMAGE.  Images whose pixels span ten and more
    decades -- a bright blob on a dark background, a smooth ramp, one hot pixel, speckle -- are judged PER OUTPUT ELEMENT against
    fp64, the way the per-row scales are judged: |lin - lin64| <= 2e-6 ||patch|| ||w_c|| for the plain contraction (the gradient
    form), and the B-cos output, its scale and the patch norm of the forward form to the bounds that follow from it.  The ladder
    of scales (include/bcos_hip.h: bcos_operands.a_imgmax) is what meets this: with BCOS_OPT_PATCH_LEVELS = 0 (the single per-image
    scale of round 3) the same check fails on the same inputs, which is asserted too -- the test has teeth.  An image's bits do
    not depend on its batch position or on the batch size, levels or not."""
    from bcos_hip import lib as blib
    from bcos_hip import ops
    prev = blib.get_contraction_mode()
    blib.set_contraction_mode(mode)
    g = torch.Generator().manual_seed(53)
    #        N   H   W  Cin Cout k st pd
    geoms = [(3, 14, 14, 256, 256, 3, 1, 1),      # 128 x 256 patch tiles
             (2, 28, 28, 128, 128, 3, 1, 1),      # 128 x 128
             (5, 7, 7, 512, 192, 3, 1, 1),        # several images per tile
             (2, 56, 56, 64, 64, 3, 1, 1),        # 256 x 64
             (1, 80, 80, 32, 64, 3, 1, 1),        # 2-D tiles (wide images)
             (2, 64, 64, 8, 64, 7, 2, 3),         # 7 x 7 / 2 stem: >= 25 taps forward, depth-to-space gradient (4 x 4 taps, 2-D tiles)
             (2, 30, 30, 64, 96, 1, 1, 0)]        # 1 x 1 (per-row scales: the control)
    worst = {}
    try:
        for (N, H, W, Cin, Cout, k, st, pd) in geoms:
            w = (torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5) * (10.0 ** (torch.rand(Cout, 1, 1, 1, generator=g) * 2 - 1))
            wdev = ops.mark_static(w.to(DEV))
            wd = w.double().to(DEV).permute(0, 3, 1, 2)
            Ho, Wo = ops.conv_out_size(H, k, st, pd), ops.conv_out_size(W, k, st, pd)
            plan = ops.DgradPlan(wdev.permute(0, 3, 1, 2).contiguous(), (st, st), (pd, pd))
            ones_f = torch.ones(1, Cin, k, k, device=DEV, dtype=torch.float64)
            wn_f = wd.flatten(1).norm(dim=1)                                       # ||w_c|| of the forward form [Cout]
            wn_b = wd.permute(1, 0, 2, 3).flatten(1).norm(dim=1)                   # ... of the gradient form [Cin] (all taps: an upper bound for a parity class)
            for kind in ("blob", "ramp", "hot", "speckle"):
                x = ops.ensure_absmax(_structured_images(N, H, W, Cin, g, kind).to(DEV))
                gl = ops.ensure_absmax(_structured_images(N, Ho, Wo, Cout, g, kind).to(DEV))

                def fwd(xx):
                    return ops.conv2d_fwd(xx, wdev, stride=(st, st), padding=(pd, pd), relu=False, want_scale=True, want_norm=True)

                def errors():
                    y, sc, nrm = fwd(x)
                    gx = plan.run(gl, H, W)
                    xd = x.double().permute(0, 3, 1, 2)
                    lin = torch.nn.functional.conv2d(xd, wd, stride=st, padding=pd)
                    pn = torch.nn.functional.conv2d(xd * xd, ones_f, stride=st, padding=pd).sqrt()           # ||patch|| [N, 1, Ho, Wo]
                    nrm64 = (pn * pn + 1e-6).sqrt()
                    y64, s64 = lin * lin.abs() / nrm64, lin.abs() / nrm64
                    wn = wn_f.view(1, -1, 1, 1)
                    tiny = 1e-30
                    e_y = ((y.double().permute(0, 3, 1, 2) - y64).abs() / (pn * wn * wn + tiny)).max().item()
                    e_s = ((sc.double().permute(0, 3, 1, 2) - s64).abs() / (wn + tiny)).max().item()
                    e_n = ((nrm.double().reshape(N, 1, Ho, Wo) - nrm64).abs() / nrm64).max().item()
                    gd = gl.double().permute(0, 3, 1, 2)
                    opad = (H - ((Ho - 1) * st - 2 * pd + k), W - ((Wo - 1) * st - 2 * pd + k))
                    g64 = torch.nn.functional.conv_transpose2d(gd, wd, stride=st, padding=pd, output_padding=opad)
                    g2 = (gd * gd).sum(1, keepdim=True)
                    if st == 1:
                        gpn = torch.nn.functional.conv_transpose2d(g2, ones_f[:, :1], stride=st, padding=pd, output_padding=opad).sqrt()   # ||window of g|| [N, 1, H, W]
                    else:
                        # a strided gradient is ONE launch over the union of its parity classes' tap windows (depth to space, zero weights
                        # where a class has no tap): a row's operand scale -- per row or per image level -- covers that union, so the
                        # bound is relative to the union window (coarse position h // st, offsets dlo .. dhi)
                        ds = [(r + pd - t) // st for r in range(st) for t in range(k) if (r + pd - t) % st == 0]
                        dlo, dhi = min(ds), max(ds)
                        gsum = torch.nn.functional.conv2d(torch.nn.functional.pad(g2, (-dlo, dhi, -dlo, dhi)),
                                                          torch.ones(1, 1, dhi - dlo + 1, dhi - dlo + 1, device=DEV, dtype=torch.float64))
                        gpn = gsum.repeat_interleave(st, 2).repeat_interleave(st, 3)[:, :, :H, :W].sqrt()
                    e_g = ((gx.double().permute(0, 3, 1, 2) - g64).abs() / (gpn * wn_b.view(1, -1, 1, 1) + tiny)).max().item()
                    assert torch.isfinite(y).all() and torch.isfinite(gx).all() and torch.isfinite(sc).all()
                    return e_y, e_s, e_n, e_g, y, gx

                e_y, e_s, e_n, e_g, y, gx = errors()
                key = (H, Cin, Cout, k, kind)
                worst[key] = (e_y, e_s, e_n, e_g)
                assert e_g <= 2e-6, ("gradient form", mode, key, e_g)
                assert e_y <= 4e-6 and e_s <= 3e-6 and e_n <= 5e-6, ("forward form", mode, key, e_y, e_s, e_n)
                # an image's bits: independent of its batch position and of the batch size
                if N > 1:
                    perm = torch.arange(N - 1, -1, -1, device=DEV)
                    yp = fwd(ops.ensure_absmax(x[perm].contiguous()))[0]
                    assert torch.equal(yp, y[perm]), ("position", mode, key)
                    gp = plan.run(ops.ensure_absmax(gl[perm].contiguous()), H, W)
                    assert torch.equal(gp, gx[perm]), ("position, gradient", mode, key)
                    y1 = fwd(ops.ensure_absmax(x[N - 1:].contiguous()))[0]
                    assert torch.equal(y1, y[N - 1:]), ("batch size", mode, key)
                # ... and the single per-image scale of round 3 does NOT meet the bound on the blob / hot-pixel images
                if mode == "f16x2" and k == 3 and kind in ("blob", "hot"):
                    with blib.option("patch_levels", 0):
                        o_y, o_s, o_n, o_g, _, _ = errors()
                    assert max(o_s / 3e-6, o_g / 2e-6) > 10.0, ("the single-scale patch loop was expected to miss the bound", key, o_s, o_g)
    finally:
        blib.set_contraction_mode(prev)
        blib.reset_options()
    print("worst per-element errors (y, scale, norm, gradient):", {k: tuple(f"{v:.1e}" for v in e) for k, e in worst.items()})


def test_batch_chunks_carry_their_image_maxima(lib):
    """a launch whose A operand reaches 2 GiB is cut into batch chunks; every chunk must see ITS images'
    maxima (bcos_operands.a_imgmax / a_imgmin are indexed by the chunk's local image index).  BCOS_OPT_SPLIT_LIMIT lowers the
    threshold so that small tensors take the chunked path: 3 x 3 (input-patch loop), 7 x 7 stem (>= 25 taps) and the depth-to-space
    gradient, images six decades apart (a chunk scaled by another image's maximum overflows fp16 or loses every bit), bit-identical
    to the unchunked launch."""
    from bcos_hip import lib as blib
    from bcos_hip import ops
    if blib.get_contraction_mode() != "f16x2":
        pytest.skip("image maxima belong to the split-f16 loop")
    g = torch.Generator().manual_seed(59)
    try:
        for (N, H, Cin, Cout, k, st, pd) in [(7, 14, 64, 128, 3, 1, 1), (5, 32, 8, 64, 7, 2, 3), (6, 20, 32, 64, 3, 1, 1)]:
            mag = torch.logspace(3, -3, N).view(N, 1, 1, 1)                # (decreasing: a later chunk read through image 0's maximum underflows)
            x = ops.ensure_absmax((torch.randn(N, H, H, Cin, generator=g) * mag).to(DEV))
            w = ops.mark_static((torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5).to(DEV))
            Ho = ops.conv_out_size(H, k, st, pd)
            gl = ops.ensure_absmax((torch.randn(N, Ho, Ho, Cout, generator=g) * mag.flip(0)).to(DEV))
            plan = ops.DgradPlan(w.permute(0, 3, 1, 2).contiguous(), (st, st), (pd, pd))

            def run():
                y, sc, nrm = ops.conv2d_fwd(x, w, stride=(st, st), padding=(pd, pd), relu=True, want_scale=True, want_norm=True)
                return y, sc, nrm, plan.run(gl, H, H)
            whole = run()
            img_bytes = H * H * Cin * 4
            for per in (1, 2, 3):
                with blib.option("split_limit", max(1 << 16, per * img_bytes + 1)):
                    parts = run()
                for i, (a, b) in enumerate(zip(whole, parts)):
                    assert torch.isfinite(b).all(), (N, H, Cin, k, per, i)
                    assert torch.equal(a, b), ("chunked launch differs", N, H, Cin, k, per, i, rel(b, a))
            xd, wd = x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2)
            lin = torch.nn.functional.conv2d(xd, wd, stride=st, padding=pd)
            nrm64 = (torch.nn.functional.conv2d(xd * xd, torch.ones(1, Cin, k, k, device=DEV, dtype=torch.float64), stride=st, padding=pd) + 1e-6).sqrt()
            y64 = torch.relu(lin * lin.abs() / nrm64).permute(0, 2, 3, 1)
            for n in range(N):
                assert rel(whole[0][n], y64[n]) <= 2e-6, (N, H, Cin, k, n)
    finally:
        blib.reset_options()


def test_colsum_ordered_is_exact_enough_and_reproducible(lib):
    """bcos_colsum_ordered (ABI v7): the column sums of bcos_colsum in a FIXED order, no atomics -- against fp64, bit-identical
    from call to call, and the moments synth.calibrate derives with it against torch's (replicas that
    calibrate independently must agree bit for bit, which torch's multi-block reductions do not guarantee under time-slicing)."""
    from bcos_hip import ops
    g = torch.Generator().manual_seed(77)
    for (rows, Cc) in [(1, 4), (1000, 64), (25088, 256), (1568, 1024), (333, 2048), (77, 12)]:
        a = torch.randn(rows, Cc, generator=g).to(DEV)
        b = (torch.randn(rows, Cc, generator=g) + 0.3).to(DEV)
        sa, sb = torch.randn(Cc, generator=g).to(DEV), torch.randn(Cc, generator=g).to(DEV)
        for args, ref in (((a,), a.double().sum(0)), ((a, b), (a.double() * b.double()).sum(0)),
                          ((a, b, sa, sb), ((a.double() - sa.double()) * (b.double() - sb.double())).sum(0))):
            o1, o2 = ops.colsum_ordered(*args), ops.colsum_ordered(*args)
            assert torch.equal(o1, o2), (rows, Cc, len(args))
            fa = (args[0].double() - (args[2].double() if len(args) > 2 else 0)).abs()
            fb = (args[1].double() - (args[3].double() if len(args) > 3 else 0)).abs() if len(args) > 1 else 1
            scale = (fa * fb).sum(0) + 1e-30
            assert float(((o1.double() - ref).abs() / scale).max()) <= 2e-6, (rows, Cc, len(args))
            # the product's column sums (bcos_colsum_ws: full bandwidth, fixed order) and the single-launch atomic form
            w1, w2 = ops.colsum(*args), ops.colsum(*args)
            assert torch.equal(w1, w2), (rows, Cc, len(args))
            assert float(((w1.double() - ref).abs() / scale).max()) <= 2e-6, (rows, Cc, len(args))
            assert rel(ops.colsum_atomic(*args), ref) <= 1e-5
    x = (torch.randn(8, 64, 14, 14, generator=g) * 3 + 1).to(DEV).contiguous(memory_format=torch.channels_last)
    mean, var, msq = ops.channel_moments_ordered(x)
    assert rel(mean, x.double().mean((0, 2, 3))) <= 1e-6 and rel(var, x.double().var((0, 2, 3), unbiased=False)) <= 1e-6
    assert abs(msq - float(x.double().pow(2).mean())) <= 1e-6 * msq
    assert all(torch.equal(a_, b_) for a_, b_ in zip(ops.channel_moments_ordered(x)[:2], (mean, var)))


def test_attention_gradient_on_the_matrix_pipe(lib):
    """bcos_attention_bwd (q, k and v differentiated: `Attention.forward` outside explanation mode, bcos/models/vit.py:143-158).  Up to
    207 tokens the five T x T x 64 products of a head run on the fp32 matrix pipe (attention_bwd_mfma_kernel, round 5), longer
    sequences on the scalar kernel: both against fp64 autograd, ragged token counts (key / query tiles partly empty), scores with a
    wide spread, bit-identical from call to call."""
    import cpu_emulation as E
    from bcos_hip import ops
    g = torch.Generator().manual_seed(11)
    for (B, T, H, spread) in [(2, 196, 3, 1.0), (3, 197, 3, 4.0), (1, 1, 1, 1.0), (2, 33, 2, 1.0), (2, 64, 1, 2.0), (1, 207, 2, 1.0),
                              (1, 208, 2, 1.0), (2, 50, 32, 1.0)]:
        qkv = (torch.randn(B, T, 3 * H * 64, generator=g) * spread).to(DEV)
        go = torch.randn(B, T, H * 64, generator=g).to(DEV)
        out, stats = ops.attention_fwd(qkv, H, 0.125, want_stats=True)
        g1 = ops.attention_bwd(qkv, stats, out, go, H, 0.125)
        g2 = ops.attention_bwd(qkv, stats, out, go, H, 0.125)
        assert torch.equal(g1, g2), (B, T, H)
        ref = E.attention_bwd(qkv.cpu(), None, None, go.cpu(), H, 0.125)        # fp64 formulae of the documented semantics
        inner = H * 64
        scale_all = float(ref.double().norm())
        for name, sl in (("q", slice(0, inner)), ("k", slice(inner, 2 * inner)), ("v", slice(2 * inner, 3 * inner))):
            # (a single token has P = 1, dS = 0: the q / k gradients are exactly zero -- judged against the whole gradient's norm)
            err = float((g1[..., sl].double().cpu() - ref[..., sl].double()).norm())
            assert err <= max(2e-5 * float(ref[..., sl].double().norm()), 1e-6 * scale_all), (B, T, H, name, err, scale_all)
        assert bool(torch.isfinite(g1).all())


def test_scale_derivative_applies_the_batchnorm_gradient(lib):
    """bcos_train_scale_bwd_bn: the input gradient of the BatchNormUncentered2d behind a layer formed inside the scale derivative's
    launch equals bcos_channel_axpby followed by bcos_train_scale_bwd_absmax (same expressions: 2 ulp), with and without the variance
    term, both epsilon modes, the pow form, maxima included."""
    from bcos_hip import ops
    from bcos_hip.lib import BCOS_CONV_EPS, BCOS_LINEAR_EPS
    torch.manual_seed(9)
    for rows, Cc in ((777, 64), (3136, 256), (50, 2048)):
        ga, y = torch.randn(rows, Cc, device=DEV), torch.randn(rows, Cc, device=DEV)
        s = torch.rand(rows, Cc, device=DEV) + 0.05
        norm = torch.rand(rows, device=DEV) + 0.5
        g, mean, coef = torch.rand(Cc, device=DEV) + 0.5, torch.randn(Cc, device=DEV), torch.randn(Cc, device=DEV) * 0.1
        for mode, b, fp in ((BCOS_CONV_EPS, 2.0, False), (BCOS_LINEAR_EPS, 2.0, True), (BCOS_CONV_EPS, 1.5, False)):
            for with_var in (True, False):
                gy = ops.channel_axpby(ga, g, y, mean, coef) if with_var else ops.channel_affine(ga, g, None)
                r0 = ops.train_scale_bwd(gy, y, s, norm, mode, b, fp, want_absmax=True)
                r1 = ops.train_scale_bwd(ga, y, s, norm, mode, b, fp, want_absmax=True, bn=(g, mean if with_var else None, coef if with_var else None))
                assert rel(r1[0], r0[0]) <= 1e-6 and rel(r1[1], r0[1]) <= 1e-6, (rows, Cc, mode, b, with_var)
                am = ops.absmax_of(r1[0])
                assert am is not None and torch.equal(am.view(torch.float32), r1[0].abs().amax(dim=1))


def test_fused_batchnorm_training_kernels(lib):
    """bcos_bn_batch_stats / bcos_relu_bwd_colsums (ABI v8): the batch statistics of a BatchNormUncentered2d from ONE pass (shifted
    per-workgroup sums combined as (n, mean, M2) triples) against fp64 -- columns whose mean is 30 x their spread included --, the running
    variance update, the gated gradient and the sums / weight gradient / variance coefficient of the backward; bit-identical from call
    to call; argument validation through ctypes."""
    import ctypes as C
    from bcos_hip import ops
    g = torch.Generator().manual_seed(5)
    for (rows, Cc) in [(1, 4), (7, 8), (3136 * 4, 64), (1000, 256), (200704, 64), (12544, 2048), (333, 12)]:
        off = torch.randn(Cc, generator=g) * torch.tensor([0.0, 1.0, 30.0, 300.0])[torch.arange(Cc) % 4]
        y = (torch.randn(rows, Cc, generator=g) * (torch.rand(Cc, generator=g) + 0.1) + off).to(DEV)
        w = (torch.rand(Cc, generator=g) + 0.5).to(DEV)
        rv0 = (torch.rand(Cc, generator=g) + 0.5).to(DEV)
        rv = rv0.clone()
        mean, var, rstd, gv = ops.bn_batch_stats(y, w, 1e-5, running_var=rv, momentum=0.1)
        m2, v2, r2, g2 = ops.bn_batch_stats(y, w, 1e-5)
        assert all(torch.equal(a, b) for a, b in ((mean, m2), (var, v2), (rstd, r2), (gv, g2)))
        yd = y.double()
        mref, vref = yd.mean(0), yd.var(0, unbiased=False)
        spread = yd.std(0, unbiased=False) + 1e-30
        if rows > 1:
            assert float(((mean.double() - mref).abs() / (spread + mref.abs() * 1e-1)).max()) <= 2e-6, (rows, Cc)
            # the centred variance to fp32 rounding of a TWO-pass evaluation: relative to var + (ulp of the column's magnitude)^2 terms
            tol = 4e-6 * vref + 4e-7 * (mref.abs() + spread) * spread
            assert bool(((var.double() - vref).abs() <= tol).all()), (rows, Cc, float(((var.double() - vref).abs() / tol).max()))
        assert rel(rstd, torch.rsqrt(var.double() + 1e-5)) <= 1e-6 and rel(gv, w.double() * torch.rsqrt(var.double() + 1e-5)) <= 1e-6
        assert rel(rv, 0.9 * rv0.double() + 0.1 * var.double()) <= 1e-6
        gr = torch.randn(rows, Cc, generator=g).to(DEV)
        act = torch.randn(rows, Cc, generator=g).clamp_min(0).to(DEV)
        for use_act in (True, False):
            ga, sgx, sg, gw, coef = ops.relu_bwd_colsums(gr, act if use_act else None, y, rstd=rstd, gvec=gv, want_sg=True, want_gw=True, want_coef=True)
            ga_b = ops.relu_bwd_colsums(gr, act if use_act else None, y, rstd=rstd, gvec=gv, want_sg=True, want_gw=True, want_coef=True)
            assert all(torch.equal(a, b) for a, b in zip((ga, sgx, sg, gw, coef), ga_b))
            ga_ref = gr * (act > 0) if use_act else gr
            assert torch.equal(ga, ga_ref) and (use_act or ga.data_ptr() == gr.data_ptr())
            sref = (ga_ref.double() * yd).sum(0)
            scale = (ga_ref.double().abs() * yd.abs()).sum(0) + 1e-30
            assert float(((sgx.double() - sref).abs() / scale).max()) <= 2e-6
            assert float(((sg.double() - ga_ref.double().sum(0)).abs() / (ga_ref.double().abs().sum(0) + 1e-30)).max()) <= 2e-6
            assert rel(gw, sgx.double() * rstd.double()) <= 1e-6
            assert rel(coef, -(gv.double() * sgx.double()) * rstd.double() ** 2 / rows) <= 1e-6
    # bcos_channel_affine_rows / bcos_train_scale_bwd_absmax: same values as the plain entry points + the EXACT per-row maxima
    for (rows, Cc) in [(5, 8), (1000, 32), (3136 * 2, 64), (777, 128), (333, 256), (50, 1000), (64, 2048)]:
        xx = torch.randn(rows, Cc, generator=g).to(DEV)
        sc_, sh_ = (torch.rand(Cc, generator=g) + 0.5).to(DEV), torch.randn(Cc, generator=g).to(DEV)
        ad = torch.randn(rows, Cc, generator=g).to(DEV)
        for addend, relu in ((None, True), (ad, True), (ad, False), (None, False)):
            o = ops.channel_affine_rows(xx, sc_, sh_, addend, relu=relu)
            ref_o = ops.channel_affine_add(xx, sc_, sh_, addend, relu=relu) if addend is not None else ops.channel_affine(xx, sc_, sh_, relu=relu)
            assert torch.equal(o, ref_o), (rows, Cc, relu)
            am = ops.absmax_of(o)
            assert am is not None and torch.equal(am.view(torch.float32), o.abs().amax(1)), (rows, Cc, relu)
        yy, ss = torch.randn(rows, Cc, generator=g).to(DEV), (torch.rand(rows, Cc, generator=g) + 0.1).to(DEV)
        nn_ = (torch.rand(rows, generator=g) + 0.5).to(DEV)
        from bcos_hip.lib import BCOS_CONV_EPS
        g_a, r_a, _ = ops.train_scale_bwd(xx, yy, ss, nn_, BCOS_CONV_EPS, 2.0, want_absmax=True)
        g_b, r_b, _ = ops.train_scale_bwd(xx, yy, ss, nn_, BCOS_CONV_EPS, 2.0)
        assert torch.equal(g_a, g_b) and torch.equal(r_a, r_b) and ops.absmax_of(g_b) is None
        assert torch.equal(ops.absmax_of(g_a).view(torch.float32), g_a.abs().amax(1))
    # validation: C % 4, missing outputs, act without ga, coef without gvec, a workspace that is too small
    y = torch.randn(64, 8, device=DEV)
    ws = torch.empty(8 * 3 * 4, device=DEV)
    o = [torch.empty(8, device=DEV) for _ in range(4)]
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None      # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.bcos_bn_batch_stats(P(y), None, None, P(o[0]), P(o[1]), P(o[2]), P(o[3]), P(ws), ws.numel(), 64, 8, 1e-5, 0.0, st) == 0
    assert lib.bcos_bn_batch_stats(P(y), None, None, P(o[0]), P(o[1]), P(o[2]), P(o[3]), P(ws), ws.numel(), 64, 6, 1e-5, 0.0, st) != 0
    assert lib.bcos_bn_batch_stats(P(y), None, None, None, P(o[1]), P(o[2]), P(o[3]), P(ws), ws.numel(), 64, 8, 1e-5, 0.0, st) != 0
    assert lib.bcos_bn_batch_stats(P(y), None, None, P(o[0]), P(o[1]), P(o[2]), P(o[3]), P(ws), 8, 64, 8, 1e-5, 0.0, st) != 0
    assert lib.bcos_relu_bwd_colsums(P(y), P(y), P(y), None, None, None, P(o[0]), None, None, None, P(ws), ws.numel(), 64, 8, st) != 0
    assert lib.bcos_relu_bwd_colsums(P(y), None, P(y), None, P(o[1]), None, P(o[0]), None, None, P(o[2]), P(ws), ws.numel(), 64, 8, st) != 0
    assert lib.bcos_relu_bwd_colsums(P(y), None, P(y), None, None, None, P(o[0]), None, None, None, P(ws), ws.numel(), 64, 8, st) == 0
    torch.cuda.synchronize()


def test_option_table_switches_code_paths_not_results(lib):
    """bcos_set_option (ABI v7) through ctypes: defaults, range checks, and that an option selects between code paths of the SAME
    operator -- a 3 x 3 launch under every setting of the loop / tile / epilogue switches agrees with the default to fp32 rounding
    (bit for bit where the paths promise it).  The library ignores the process environment (rounds 1-3 read it per launch)."""
    import ctypes as C
    from bcos_hip import lib as blib
    from bcos_hip import ops
    v = C.c_int64(-1)
    assert lib.bcos_get_option(blib.OPTIONS["patch"], C.byref(v)) == 0 and v.value == 1
    assert lib.bcos_set_option(99, 1) != 0 and lib.bcos_set_option(blib.OPTIONS["h2_tile"], 3) != 0
    assert lib.bcos_get_option(blib.OPTIONS["h2_tile"], C.byref(v)) == 0 and v.value == 0
    if blib.get_contraction_mode() != "f16x2":
        return
    g = torch.Generator().manual_seed(5)
    x = ops.ensure_absmax(torch.randn(4, 28, 28, 128, generator=g).to(DEV))
    w = ops.mark_static((torch.randn(256, 3, 3, 128, generator=g) / 34).to(DEV))

    def run():
        return ops.conv2d_fwd(x, w, stride=(1, 1), padding=(1, 1), relu=True, want_scale=True, want_norm=True)
    base = run()
    os.environ["BCOS_PATCH"] = "0"                 # what rounds 1-3 would have obeyed
    try:
        again = run()
    finally:
        os.environ.pop("BCOS_PATCH", None)
    assert all(torch.equal(a_, b_) for a_, b_ in zip(base, again))
    for name, val, exact in (("epi_generic", 1, True), ("patch_levels", 0, True), ("patch_wide", 0, True), ("patch", 0, False),
                             ("h2_loop", 1, False), ("h2_tile", 1, False), ("h2_tile", 2, False), ("tail_split", 0, False)):
        with blib.option(name, val):
            other = run()
        for a_, b_ in zip(base, other):
            assert (torch.equal(a_, b_) if exact else rel(b_, a_) <= 2e-6), (name, val, rel(b_, a_))
        assert blib.get_option(name) == blib._loaded_options[name]


def test_bench_train_diagnostic_line(lib):
    """`bench.py --train` (a driver-visible training number): one JSON line of the contract's shape for a small
    configuration -- train-mode forward, BCE loss, backward, SGD update on the per-layer HIP kernels; finite loss, positive rate."""
    import subprocess
    import sys as _sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([_sys.executable, os.path.join(repo, "bench.py"), "--train", "--arch", "resnet18", "--batch", "8", "--steps", "2",
                           "--warmup", "1"], capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    res = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["unit"] == "images/s" and res["value"] > 0 and res["steps"] == 2 and res["n_gpus"] == 1
    assert "TRAINING step" in res["config"]["workload"] and res["config"]["global_batch"] == 8
    assert math.isfinite(res["config"]["final_loss"]) and res["roofline"]["achieved"] > 0 and res["step_times"]["min"] > 0


def test_bench_headline_line_carries_the_contract_and_the_references(lib):
    """`python bench.py` at a small batch: ONE JSON line with the contract's keys, the roofline of the dominant kernel measured with HIP
    events inside the run, the clock telemetry, and -- round 5 -- what the vendor's primitives reach beside it: the plain fp16 GEMMs of
    the matrix-bound launches' shapes and a device copy for the bandwidth-bound ones.  (The CPU baseline has its own leg; off here.)"""
    import subprocess
    import sys as _sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([_sys.executable, os.path.join(repo, "bench.py"), "--batch", "32", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                          capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in res, k
    assert res["unit"] == "images/s" and res["value"] > 0 and res["steps"] == 3 and res["n_gpus"] == 1 and res["vs_baseline"] is None
    assert res["scaling"] == "weak" and res["higher_is_better"] is True and "workload" in res["config"] and "model" not in res["config"]
    rf = res["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("mfma", "hbm") and rf["achieved"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["avg_launch_us"] > 0 and rf["launches_per_step"] == 116          # (117 until the head gradient became a streaming launch)
    v = rf["by_bound"]["mfma"]["vendor_f16_gemm"]
    assert "error" not in v and v["ms_per_step"] > 0 and v["ours_ms_per_step"] > 0 and v["launches_per_step"] == rf["by_bound"]["mfma"]["launches_per_step"]
    h = rf["by_bound"]["hbm"]
    assert h["stream_copy_gbps"] and 1000 < h["stream_copy_gbps"] < 8000 and h["frac_of_stream_copy"] > 0


def test_image_range_folded_into_the_producing_epilogue(lib):
    """Round 6: the per-image range of a tensor's per-pixel maxima (the operand scales of the 3 x 3 launch that reads it) comes out of the
    producing launch's epilogue (bcos_epilogue.out_imgmax / out_imgmin_c) instead of a bcos_image_absrange pass: maxima bit-equal to that
    pass, complemented minima equal to it where a tile owns its pixels and a lower bound otherwise; several launches filling one tensor
    accumulate; the reader's results are bit-identical either way -- through a whole ResNet-18 / ResNet-50 explanation too."""
    from bcos_hip import lib as blib, ops, synth
    import bcos_hip.engine as en
    if blib.get_contraction_mode() != "f16x2":
        pytest.skip("operand maxima belong to the f16x2 contraction")
    g = torch.Generator().manual_seed(3)
    arena = ops.AbsmaxArena()
    for (N, H, Cin, Cout, zero_img) in ((5, 14, 64, 64, None), (3, 28, 32, 256, 1), (4, 7, 128, 512, None), (9, 5, 64, 128, 4), (2, 56, 64, 64, None)):
        x = torch.randn(N, H, H, Cin, generator=g) * torch.rand(N, 1, 1, 1, generator=g).mul(8).exp2()
        if zero_img is not None:
            x[zero_img] = 0.0                                   # an image without a nonzero pixel: max 0, complemented min 0
        x = x.to(DEV)
        w = ops.mark_static((torch.randn(Cout, 1, 1, Cin, generator=g) / Cin ** 0.5).to(DEV))
        w3 = ops.mark_static((torch.randn(64, 3, 3, Cout, generator=g) / (9 * Cout) ** 0.5).to(DEV))
        outs = {}
        for fused in (True, False):
            ops.FUSE_IMAGE_RANGE = fused
            try:
                with ops.absmax_arena(arena, x.device):
                    with ops.image_range_reader():              # (what the engine says around the producer of a 3 x 3 layer's input)
                        y = ops.conv2d_fwd(ops.ensure_absmax(x), w, relu=True, want_scale=False)[0]
                    am = ops.absmax_of(y)
                    rec = getattr(am, "_bcos_imgmax", None)
                    assert (rec is not None and len(rec) == 3) == (fused and H * H >= 19), (fused, H)
                    if rec is not None:
                        ref = torch.empty(2, N, device=DEV, dtype=torch.int32)
                        blib.check(lib.bcos_image_absrange(am.data_ptr(), ref[0].data_ptr(), ref[1].data_ptr(), N, H * H, None), "range")
                        torch.cuda.synchronize()
                        assert torch.equal(rec[0][0], ref[0]), "maxima"
                        lo = (~rec[0][1]).view(torch.int32)                              # complemented lower bound -> plain
                        allzero = ref[1] == -1                                           # 0xffffffff: no nonzero pixel
                        assert torch.equal(lo[allzero], ref[1][allzero])
                        assert bool((lo[~allzero].to(torch.int64) <= ref[1][~allzero].to(torch.int64)).all())
                        if Cout <= 64:                                                   # one column tile: the tile owns its pixels -> exact
                            assert torch.equal(lo, ref[1]), (N, H, Cout)
                    z = ops.conv2d_fwd(y, w3, stride=(1, 1), padding=(1, 1), relu=True, want_scale=True, want_norm=True)
                    outs[fused] = [y] + [t for t in z if t is not None]
            finally:
                ops.FUSE_IMAGE_RANGE = True
        assert all(torch.equal(a_, b_) for a_, b_ in zip(outs[True], outs[False])), (N, H, Cin, Cout)
    for arch, n in (("resnet18", 6), ("resnet50", 4)):
        net = synth.build_bcosified_resnet(arch, seed=0).to(DEV)
        xi = synth.synthetic_images(n, seed=9).to(DEV)
        with torch.no_grad():
            synth.calibrate(net, xi[:4])
        eng = en.attach(net)
        a = eng.explain(xi)
        ops.FUSE_IMAGE_RANGE = False
        try:
            b = eng.explain(xi)
        finally:
            ops.FUSE_IMAGE_RANGE = True
        for k in ("logits", "dynamic_linear_weights", "contribution_map"):
            assert torch.equal(a[k], b[k]), (arch, k)


def test_c_abi_image_absmax(lib):
    """bcos_image_absmax (ABI v6) through the C ABI: per-image maxima of per-pixel maxima, image sizes on either side of the
    kernel's 4096-pixel stride, bit-exact (integer maxima of fp32 bit patterns)."""
    import ctypes as C
    g = torch.Generator().manual_seed(12)
    for (n, hw) in [(1, 1), (3, 49), (256, 196), (5, 3136), (2, 4096), (3, 5000), (2, 12544)]:
        x = (torch.randn(n * hw, generator=g).abs() * torch.logspace(-6, 6, n).repeat_interleave(hw)).to(DEV)
        am = x.view(torch.int32)
        out = torch.full((n,), -1, device=DEV, dtype=torch.int32)
        assert lib.bcos_image_absmax(C.c_void_p(am.data_ptr()), C.c_void_p(out.data_ptr()), n, hw, None) == 0
        torch.cuda.synchronize()
        assert torch.equal(out, am.view(n, hw).max(1).values)
    assert lib.bcos_image_absmax(None, C.c_void_p(out.data_ptr()), 1, 1, None) != 0
    # bcos_image_absrange (ABI v7): also the smallest NONZERO per-pixel maximum of every image (0xffffffff where all are zero)
    for (n, hw) in [(1, 1), (4, 196), (3, 5000)]:
        x = (torch.rand(n * hw, generator=g) + 0.01) * 10.0 ** (torch.rand(n * hw, generator=g) * 20 - 10)
        x[torch.rand(n * hw, generator=g) < 0.3] = 0.0
        x = x.view(n, hw)
        x[n - 1] = 0.0
        am = x.to(DEV).view(torch.int32)
        mx = torch.full((n,), -1, device=DEV, dtype=torch.int32)
        mn = torch.full((n,), 7, device=DEV, dtype=torch.int32)
        assert lib.bcos_image_absrange(C.c_void_p(am.data_ptr()), C.c_void_p(mx.data_ptr()), C.c_void_p(mn.data_ptr()), n, hw, None) == 0
        torch.cuda.synchronize()
        assert torch.equal(mx, am.max(1).values)
        want = torch.where(am > 0, am, torch.full_like(am, 0x7fffffff)).min(1).values
        want = torch.where(want == 0x7fffffff, torch.full_like(want, -1), want)          # (-1 = 0xffffffff as int32)
        assert torch.equal(mn, want), (n, hw)


def test_patch_loop_rectangular_images(lib, monkeypatch):
    """The input-patch loop on non-square images (row pitch, rotation and tile spans use P, Q, H, W separately): forward and input
    gradient against the per-tap loop and fp64, linear tiles (narrow images) and 2-D tiles (wide ones)."""
    from bcos_hip import lib as blib
    from bcos_hip import ops
    g = torch.Generator().manual_seed(3)
    for (N, H, W, Cin, Cout) in [(3, 10, 23, 64, 128), (2, 31, 9, 32, 64), (2, 12, 90, 32, 64), (5, 7, 14, 128, 256)]:
        x = ops.ensure_absmax(torch.randn(N, H, W, Cin, generator=g).to(DEV))
        w = ops.mark_static((torch.randn(Cout, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5).to(DEV))
        blib.set_option("patch", 1)
        y_p = ops.conv2d_fwd(x, w, stride=(1, 1), padding=(1, 1), relu=False, want_norm=True)
        blib.set_option("patch", 0)
        y_t = ops.conv2d_fwd(x, w, stride=(1, 1), padding=(1, 1), relu=False, want_norm=True)
        blib.set_option("patch", 1)
        xd, wd = x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2)
        lin = torch.nn.functional.conv2d(xd, wd, padding=1)
        nrm = (torch.nn.functional.conv2d(xd * xd, torch.ones(1, Cin, 3, 3, device=DEV, dtype=torch.float64), padding=1) + 1e-6).sqrt()
        y64 = (lin * lin.abs() / nrm).permute(0, 2, 3, 1)
        assert rel(y_p[0], y64) <= 2e-6 and rel(y_p[0], y_t[0]) <= 2e-6, (N, H, W, Cin, Cout, rel(y_p[0], y64))
        assert rel(y_p[2].double().flatten(), nrm.flatten()) <= 1e-6
        plan = ops.DgradPlan(w.permute(0, 3, 1, 2).contiguous(), (1, 1), (1, 1))
        gl = ops.ensure_absmax(torch.randn(N, H, W, Cout, generator=g).to(DEV))
        gx = plan.run(gl, H, W)
        g64 = torch.nn.functional.conv_transpose2d(gl.double().permute(0, 3, 1, 2), wd, padding=1).permute(0, 2, 3, 1)
        assert rel(gx, g64) <= 2e-6, (N, H, W, Cin, Cout, rel(gx, g64))


def test_training_gradients_with_ragged_output_widths(lib):
    """N4 remainder: training-mode layers whose out_channels (per group) is not a multiple of four.  The scale-derivative
    kernel moves float4; ragged widths are padded to four columns for that launch and cut back (bcos/modules/_hipfn.py:
    _scale_bwd_cols).  Input, weight and bias gradients against the oracle's autograd (CPU, fp64)."""
    from oracle import bcos_oracle as O
    from bcos.modules.bcosifyconv2d import BcosifyConv2d
    g = torch.Generator().manual_seed(91)
    for (cin, cout, k, st, pd, groups, bias) in [(8, 6, 3, 1, 1, 1, True), (16, 12, 3, 1, 1, 2, False), (8, 10, 1, 1, 0, 1, True), (12, 9, 3, 2, 1, 1, False)]:
        m = BcosifyConv2d(cin, cout, k, st, pd, groups=groups, b=2)
        if bias:                                    # (what from_standard_module attaches, bcosifyconv2d.py:18-31)
            m.linear.bias = torch.nn.Parameter(torch.randn(cout, generator=g) * 0.1)
        m = m.to(DEV).train()
        with torch.no_grad():
            m.linear.weight.copy_((torch.randn(m.linear.weight.shape, generator=g) / (k * k * cin / groups) ** 0.5).to(DEV))
        x = torch.randn(3, cin, 9, 9, generator=g)
        r = torch.randn(3, cout, O.F.conv2d(x, m.linear.weight.detach().cpu(), None, st, pd, groups=groups).shape[2],
                        O.F.conv2d(x, m.linear.weight.detach().cpu(), None, st, pd, groups=groups).shape[3], generator=g)
        xg = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = m(xg)
        (y * r.to(DEV)).sum().backward()
        xd = x.double().requires_grad_(True)
        wd = m.linear.weight.detach().cpu().double().requires_grad_(True)
        bd = m.linear.bias.detach().cpu().double().requires_grad_(True) if bias else None
        yd = O.bcos_conv2d(xd, wd, bd, stride=st, padding=pd, groups=groups, b=2)
        (yd * r.double()).sum().backward()
        what = (cin, cout, k, st, groups)
        assert rel(y, yd) <= 1e-5, what
        assert rel(xg.grad, xd.grad) <= 1e-5, what
        assert rel(m.linear.weight.grad, wd.grad) <= 1e-5, what
        if bias:
            assert rel(m.linear.bias.grad, bd.grad) <= 1e-5, what


def test_replayed_gates_run_on_the_sub_batch_streams_too(lib):
    """engine.explain(x, gates=...) cuts the replayed ReLU decisions along the batch like the images (round 4: the gate-pinned path
    used to fall back to one stream silently); n_streams() states on how many streams a call runs; the bits of every output are the
    same on one stream and on two."""
    from bcos_hip import engine, synth
    net = synth.build_bcosified_resnet("resnet18").to(DEV).eval()
    with torch.no_grad():
        synth.calibrate(net, synth.synthetic_images(8).to(DEV))
    eng = engine.attach(net)
    try:
        x = synth.synthetic_images(64, seed=3, size=64).to(DEV)
        gates = _oracle_gates(net, x, "resnet18")
        assert eng.n_streams(x) == 2 and eng.n_streams(x[:8]) == 1 and eng.n_streams(x, cotangent=lambda e: e) == 1
        two = eng.explain(x, gates=[g.clone() for g in gates])
        eng.subbatch_streams = 1
        assert eng.n_streams(x) == 1
        one = eng.explain(x, gates=[g.clone() for g in gates])
        for k in ("logits", "dynamic_linear_weights", "contribution_map"):
            assert torch.equal(one[k], two[k]), k
        free = eng.explain(x)
        assert rel(one["logits"], free["logits"]) <= 1e-6          # (the forward values of open gates are the same; closed ones are zero either way)
    finally:
        eng.subbatch_streams = 2
        engine.detach(net)


def test_engine_with_grouped_and_maxout_blocks(lib):
    """Networks with grouped / MaxOut B-cos convolutions in the fused plan -- both as fused nodes since round 4 (grouped launches forward
    and backward; MaxOut inside the contraction's epilogue, its gradient routed to the winning filters by bcos_maxout_expand): same
    logits, W(x) and maps as the pure nn.Module explanation; batch large enough for the two sub-batch streams; an image's bits do not
    depend on its batch; the engine's own ReLU decisions, replayed, reproduce its maps."""
    from bcos_hip import engine, synth
    from bcos.modules.bcosifyconv2d import BcosifyConv2d
    net = synth.build_bcosified_resnet("resnet18")
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        blk = net.model.layer2[1]
        blk.conv1 = BcosifyConv2d(128, 128, 3, 1, 1, groups=2, b=2)
        blk.conv1.linear.weight.copy_(torch.randn(blk.conv1.linear.weight.shape, generator=g) / (9 * 64) ** 0.5)
        blk = net.model.layer3[1]
        blk.conv2 = BcosifyConv2d(256, 256, 3, 1, 1, max_out=2, b=2)
        blk.conv2.linear.weight.copy_(torch.randn(blk.conv2.linear.weight.shape, generator=g) / (9 * 256) ** 0.5)
    net = net.to(DEV).eval()
    with torch.no_grad():
        synth.calibrate(net, synth.synthetic_images(8).to(DEV))
    x = synth.synthetic_images(64, seed=77, size=96).to(DEV)
    ref = net.explain_batch(x)                                      # no engine attached: autograd over the modules
    eng = engine.attach(net)
    try:
        assert not any(b.hybrid for b in eng.blocks) and eng.blocks[3].convs[0].groups == 2 and eng.blocks[5].convs[1].max_out == 2
        out = net.explain_batch(x)
        assert rel(out["logits"], ref["logits"]) <= 1e-5
        assert torch.equal(out["prediction"], ref["prediction"])
        assert rel(out["dynamic_linear_weights"], ref["dynamic_linear_weights"]) <= 2e-4      # (free ReLU gates, 18 layers)
        assert rel(out["contribution_map"], ref["contribution_map"]) <= 2e-4
        one = eng.explain(x[40:41])
        assert torch.equal(one["contribution_map"], out["contribution_map"][40:41])            # an image's bits: independent of the batch
    finally:
        engine.detach(net)


def test_clip_zeroshot_text_attribution_against_reference_golden(lib, golden_dir):
    """Explanation of the zero-shot TEXT logit through the fused engine (bcos_hip.clip_head.zeroshot_attribution): the pooled
    head and the attn_unpool head with its pooled-cosine variants (interpretability/analyses/text_localisation.py:68-104).
    Fixture: attributions recorded with the reference's own statements (tests/golden/make_golden.py: clip_zeroshot_attribution).
    Free ReLU gates: bounded by the reference-vs-reference floor of this 55-layer CNN -- the reference run twice on one host
    (oneDNN on / off) disagrees with itself by 1.7e-3 (pooled) ... 4.2e-3 (un-pooled mean of 49 cosines whose gradients
    largely cancel), recorded in oracle_vs_reference.json; the bound is max(2e-3, 3 x that floor).  With the oracle's gates
    replayed W(x) and the maps hold 1e-4 against the oracle, which the CPU suite pins to the same fixture."""
    floor = json.load(open(os.path.join(golden_dir, "oracle_vs_reference.json")))
    free = lambda key: max(2e-3, 3.0 * floor[f"zeroshot_attr/reference_self_{key}"][0])      # noqa: E731
    from bcos_hip import clip_head, engine, synth
    meta = json.load(open(os.path.join(golden_dir, "clip_zeroshot_attr.json")))
    data = np.load(os.path.join(golden_dir, "clip_zeroshot_attr.npz"))
    calib = np.load(os.path.join(golden_dir, "clip_rn50.npz"))
    cmeta = json.load(open(os.path.join(golden_dir, "clip_rn50.json")))
    record = {k: torch.from_numpy(calib["calib/" + k]) for k in cmeta["calib_order"]}
    x = synth.synthetic_images(meta["n_images"], seed=meta["image_seed"]).to(DEV)
    wt = torch.randn(1024, 16, generator=torch.Generator().manual_seed(meta["text_seed"]))
    w1 = (wt[:, 3:4] / wt[:, 3:4].norm()).to(DEV)
    wt = wt.to(DEV)

    def gates_of(sd, xs, unpool):
        log = []
        with torch.no_grad():
            O.clip_rn50_embed(sd, xs.cpu(), detach=True, attn_unpool=unpool, gate_log=log)
        return [(p > 0).float().permute(0, 2, 3, 1).contiguous().to(DEV) for p in log]

    # ---- pooled head: max over 16 text classes ---------------------------------------------------------------------------
    net = synth.build_bcosified_clip_rn50(seed=meta["weight_seed"])
    synth.apply_calibration(net, record)
    net = net.to(DEV)
    eng = engine.attach(net)
    out = clip_head.zeroshot_attribution(eng, x[:2], wt)
    assert rel(out["logits"].max(1).values, data["pooled_values"]) <= 1e-4
    assert rel(out["dynamic_linear_weights"][0], data["pooled_weights_0"]) <= free("pooled_grad")
    assert rel(out["contribution_map"], data["pooled_maps"]) <= free("pooled_grad")
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    go, vo = O.zeroshot_attribution(lambda xx, detach: O.clip_rn50_embed(sd, xx, detach=detach), x[:2].cpu(), wt.cpu())
    pinned = clip_head.zeroshot_attribution(eng, x[:2], wt, gates=gates_of(sd, x[:2], False))
    assert rel(pinned["dynamic_linear_weights"], go) <= 1e-4 and rel(pinned["contribution_map"], (x[:2].cpu() * go).sum(1)) <= 1e-4
    assert torch.equal(pinned["explained_class_idx"].cpu(), out["explained_class_idx"].cpu())
    # batch independence of the attribution, bit for bit
    one = clip_head.zeroshot_attribution(eng, x[1:2], wt)
    assert torch.equal(one["contribution_map"], out["contribution_map"][1:2])
    del eng, net
    # ---- attn_unpool head: one text embedding, the pooling variants of :80-99 -------------------------------------------------
    net_u = synth.build_bcosified_clip_rn50(seed=meta["weight_seed"], attn_unpool=True)
    synth.apply_calibration(net_u, record)
    net_u = net_u.to(DEV)
    eng_u = engine.attach(net_u)
    assert eng_u.head_kind == "attn_unpool"
    emb = eng_u.forward(x[:1])
    assert emb.shape == (49, 1, 1024) and rel(emb, data["unpool_output_0"]) <= 1e-4
    sd_u = {k: v.detach().cpu() for k, v in net_u.state_dict().items()}
    gates_u = gates_of(sd_u, x[:1], True)
    for pc, nm in meta["variants"]:
        key = f"unpool_p{pc}_n{nm}"
        ou = clip_head.zeroshot_attribution(eng_u, x[:1], w1, pool_cosine=pc, norm_max_cosine=bool(nm))
        assert rel(ou["logits"].view(-1), data[key + "_value"].reshape(-1)) <= 1e-4, key
        assert rel(ou["contribution_map"][0], data[key + "_map"]) <= free(key + "_map"), key
        if key == "unpool_p2_n0":
            assert rel(ou["dynamic_linear_weights"][0], data["unpool_p2_n0_weights"]) <= free(key + "_grad")
        gr, _ = O.zeroshot_attribution(lambda xx, detach: O.clip_rn50_embed(sd_u, xx, detach=detach, attn_unpool=True), x[:1].cpu(),
                                       w1.cpu(), attn_unpool=True, pool_cosine=pc, norm_max_cosine=bool(nm))
        pu = clip_head.zeroshot_attribution(eng_u, x[:1], w1, pool_cosine=pc, norm_max_cosine=bool(nm), gates=[g.clone() for g in gates_u])
        # gates replayed.  The un-pooled attribution is ill-conditioned in fp32 on BOTH sides: the cosine gradient is orthogonal to
        # the (twice normalised) location vector and the location gradients largely cancel in the pooled sum, so 1e-6 rounding
        # shows up at the 1e-4 level -- with the EXACT fp32 MFMA contraction the path sits 0.7 ... 1.0e-4 from the fp32 CPU
        # oracle (scripts/_zs_dbg.py: 6.9e-5 / 8.6e-5 / 1.0e-4 / 8.7e-5 for the four variants; default f16x2 contraction
        # 7.4e-5 / 1.2e-4 / 1.4e-4 / 1.2e-4).  Bound 2e-4 here; the pooled head above holds 1e-4.
        assert rel(pu["dynamic_linear_weights"], gr) <= 2e-4, (key, rel(pu["dynamic_linear_weights"], gr))
    # the attn_unpool head of the nn.Module path gives the same output as the fused plan
    engine.detach(net_u)
    with torch.no_grad():
        assert rel(net_u(x[:1]), emb) <= 1e-4


def test_unit_norm_projection_folded_into_the_contraction(lib):
    """north_star: "fuse the unit-norm weight projection, the input . w_hat contraction and the |cos|^(B-1) scaling into one pass".
    (1) C ABI: a launch on the RAW weights with BCOS_EPI_UNIT_NORM_W (norms gathered in the main loop) and one with a
    col_scale from bcos_weight_row_invnorm both equal the launch on weights projected by bcos_weight_rownorm_scale -- conv
    (ragged Cout, MaxOut, grouped) and linear, bf16x3 and fp32 loops.  (2) nn.Module path: a training step of native
    BcosConv2d / BcosLinear (unit-norm filters, trainable `scale`) runs ONE forward launch per layer through
    FoldedUnitNormFn; outputs and every gradient equal the separate-projection path (UnitNormFn)."""
    import ctypes as C
    from bcos_hip import ops
    from bcos_hip import lib as blib
    from bcos.modules import BcosConv2d, BcosLinear, _hipfn
    g = torch.Generator().manual_seed(77)
    L = blib.load()
    for (N, H, Cin, Cout, k, st, pd, groups, mo) in [(2, 9, 16, 20, 3, 1, 1, 1, 1), (3, 8, 32, 136, 1, 1, 0, 1, 1), (2, 7, 12, 24, 3, 2, 1, 1, 2),
                                                   (2, 8, 16, 32, 3, 1, 1, 2, 1)]:
        x = torch.randn(N, H, H, Cin, generator=g).to(DEV)
        w = (torch.randn(Cout, k, k, Cin // groups, generator=g) * 0.3).to(DEV)
        gain = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
        w_hat = ops.weight_rownorm_scale(w.view(Cout, -1), gain).view_as(w)
        inv = torch.empty(Cout, device=DEV)
        blib.check(L.bcos_weight_row_invnorm(C.c_void_p(w.data_ptr()), C.c_void_p(gain.data_ptr()), C.c_void_p(inv.data_ptr()), Cout,
                                             w[0].numel(), None), "row_invnorm")
        assert rel(inv, gain / w.view(Cout, -1).norm(dim=1)) <= 1e-6
        geom = ops.fwd_geom(N, H, H, Cin // groups, Cout // groups, k, k, st, st, pd, pd)
        if groups > 1:
            geom.update(groups=groups, a_pitch=Cin, out_pitch=Cout, norm_pitch=groups)
        Ho = geom["P"]
        prev = blib.get_contraction_mode()
        try:
            for mode in ("bf16x3", "f32"):
                blib.set_contraction_mode(mode)
                outs = []
                for variant in ("projected", "flag", "col_scale"):
                    y = torch.full((N, Ho, Ho, Cout // mo), float("nan"), device=DEV)
                    sc = torch.full((N, Ho, Ho, Cout), float("nan"), device=DEV)
                    kw = dict(out=y, scale_out=sc, bcos_mode=blib.BCOS_CONV_EPS, b=2.0, max_out=mo, track_absmax=False)
                    if variant == "projected":
                        ops.tapconv(x, w_hat, geom, **kw)
                    elif variant == "flag":
                        ops.tapconv(x, w, geom, flags=blib.BCOS_EPI_UNIT_NORM_W, col_scale=gain, **kw)
                    else:
                        ops.tapconv(x, w, geom, col_scale=inv, **kw)
                    outs.append((y, sc))
                for y, sc in outs[1:]:
                    assert rel(y, outs[0][0]) <= 2e-6 and rel(sc, outs[0][1]) <= 2e-6, (mode, N, H, Cin, Cout, k, groups, mo)
        finally:
            blib.set_contraction_mode(prev)
    # nn.Module path, training step
    torch.manual_seed(3)
    for make, xs in ((lambda: BcosConv2d(16, 24, 3, padding=1, max_out=2), (4, 16, 10, 10)), (lambda: BcosConv2d(12, 20, 3, stride=2, padding=1), (3, 12, 9, 9)),
                     (lambda: BcosLinear(48, 40), (5, 7, 48))):
        m = make().to(DEV).train()
        if hasattr(m.linear, "set_scale"):
            m.linear.set_scale(m.linear.weight.detach() * 1.7, trainable=True)
        x = torch.randn(*xs, generator=g).to(DEV)
        res = {}
        for fold in (True, False):
            _hipfn.FOLD_UNIT_NORM = fold
            try:
                m.zero_grad()
                xr = x.clone().requires_grad_(True)
                y = m(xr)
                y.square().sum().backward()
                res[fold] = (y.detach(), xr.grad, m.linear.weight.grad.clone(),
                             m.linear.scale.grad.clone() if getattr(m.linear, "scale", None) is not None else None)
            finally:
                _hipfn.FOLD_UNIT_NORM = True
        for a, b_ in zip(res[True], res[False]):
            if a is not None:
                assert rel(a, b_) <= 1e-5, type(m).__name__


def test_rebuilt_multipliers_fall_back_to_stored_ones_for_large_bn_shifts(lib, golden_dir, monkeypatch):
    """The multiplier of conv1 / conv2 of a block is rebuilt from the kept activation (BCOS_EPI_MUL_FROM_ACT) only while the BN
    shift is small against the BN scale: where |bn_scale s lin| << |bn_shift| the rebuild's subtraction cancels.  A network
    with BN biases 40x those of the synthetic recipe (the regime of real checkpoints) must take the stored multipliers --
    bit-identical to BCOS_STORE_T -- and hold the usual bounds against the oracle; the unmodified network keeps rebuilding."""
    from bcos_hip import engine, synth
    import importlib
    net, meta, data = _golden_net(golden_dir, "resnet18_e2e")
    eng = engine.attach(net)
    assert all(c.rebuild_ok for blk in eng.blocks for c in blk.convs[:-1])          # the recipe nulls every bias: no shift at all
    gb = torch.Generator().manual_seed(9)
    for m in net.modules():                                                         # a `use_bias` network: BN biases of a few units
        if type(m).__name__ == "BatchNormUncentered2d":
            m.bias = torch.nn.Parameter((torch.randn(m.num_features, generator=gb) * 3.0).to(DEV))
    x = synth.synthetic_images(4, seed=meta["image_seed"]).to(DEV)
    out = eng.explain(x)
    assert not any(c.rebuild_ok for blk in eng.blocks for c in blk.convs[:-1])      # refreshed: large shifts -> stored multipliers
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    ref = O.explain_batch(lambda xx, detach: O.resnet_logits(sd, xx, meta["arch"], detach=detach), x.cpu())
    assert rel(out["logits"], ref["logits"]) <= 1e-4
    pinned = eng.explain(x, gates=_oracle_gates(net, x, meta["arch"]))
    assert rel(pinned["dynamic_linear_weights"], ref["dynamic_linear_weights"]) <= 1e-4
    # forcing the rebuild on this network is what the fallback avoids: measurably further from the oracle
    monkeypatch.setattr(engine, "_REBUILD_MAX_SHIFT", 1e30)
    eng.refresh()
    forced = eng.explain(x, gates=_oracle_gates(net, x, meta["arch"]))
    e_forced = rel(forced["dynamic_linear_weights"], ref["dynamic_linear_weights"])
    e_stored = rel(pinned["dynamic_linear_weights"], ref["dynamic_linear_weights"])
    assert e_stored <= e_forced * 1.5 + 1e-7, (e_stored, e_forced)
    print(f"rebuild at large BN shifts: W(x) relL2 vs oracle stored {e_stored:.2e}, rebuilt {e_forced:.2e}")
