"""Host-side logic of pcfa_amd (networks, adapter API, attack schedule) on CPU.

The package has no CPU operators, so these tests inject the oracle through
pcfa_amd.ops.override_for_testing and compare against golden vectors recorded from the real reference
(same seeded weights: tests/golden/make_golden.py loads pcfa_amd's state_dict into the reference's
modules, which also pins parameter-name compatibility)."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from pcfa_amd import attack_PCFA, ops
from pcfa_amd.helper_functions import ownutilities
from tests import closure_util
from tests.util import load_golden, max_abs, rel_l2, t

CASES = {
    "raft": ("RAFT", 128, 160, "change_of_variables", False, "zero", "aee", 1),
    "gma": ("GMA", 128, 160, "change_of_variables", False, "neg_flow", "aee", 2),
    "pwcnet": ("PWCNet", 120, 180, "clipping", True, "zero", "aee", 3),
    "spynet": ("SpyNet", 100, 150, "change_of_variables", False, "zero", "mse", 4),
    # wiring-only fixture: the reference's model code with its CUDA-only extensions bound to the oracle
    # (tests/golden/make_golden.py docstring)
    "flownet2": ("FlowNet2", 128, 192, "change_of_variables", False, "zero", "aee", 5),
}


@pytest.fixture(autouse=True)
def _oracle_backend(oracle_ops):
    torch.set_num_threads(8)
    with ops.override_for_testing(oracle_ops):
        yield


@pytest.mark.parametrize("name", list(CASES))
def test_closure_matches_reference(name):
    """flow, loss and d loss/d nw_input of one closure vs the reference (fp32 CPU, same thread count)."""
    g = load_golden("closure_" + name)
    net, h, w, box, joint, tgt, loss, seed = CASES[name]
    leaves = [t(g["leaf0"])] if joint else [t(g["leaf0"]), t(g["leaf1"])]
    r = closure_util.run_closure(net, h, w, box, joint, tgt, loss, seed, torch.device("cpu"),
                                 images=(t(g["image1"].astype(np.float32)), t(g["image2"].astype(np.float32))),
                                 leaves=leaves)
    scale = float(np.abs(g["flow"]).max())
    assert max_abs(r["flow_init"], t(g["flow_init"])) <= 2e-4 * scale
    assert max_abs(r["flow"], t(g["flow"])) <= 2e-4 * scale
    assert abs(r["loss"] - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    for i, gr in enumerate(r["grads"]):
        assert rel_l2(gr, t(g["grad%d" % i])) < 1e-3


def test_pwcnet_folded_glue_is_the_same_network():
    """Config.pwc_fold_glue (RGB -> BGR in conv1a's weights, `up_flow * s` inside the warp, decoder inputs handed to the
    dense block as parts, one re-gridding copy between the dilated context layers) against the reference's statement
    order (PWCNet.py:227-330) on the CPU port: the same network -- only conv1a's three input channels are summed in
    another order."""
    import dataclasses
    from pcfa_amd import config

    def run(conf):
        closure_util._MODELS.clear()
        r = closure_util.run_closure("PWCNet", 128, 192, "clipping", True, "zero", "aee", 5, torch.device("cpu"),
                                     config=conf)
        closure_util._MODELS.clear()
        return r
    a = run(config.DEFAULT)
    b = run(dataclasses.replace(config.DEFAULT, pwc_fold_glue=False))
    assert config.DEFAULT.pwc_fold_glue
    assert abs(a["loss"] - b["loss"]) <= 1e-6 * abs(b["loss"])
    assert max_abs(a["flow"], b["flow"]) <= 1e-5 * float(b["flow"].abs().max())
    assert rel_l2(a["grads"][0], b["grads"][0]) < 1e-4


def test_pwcnet_regrid_between_subgrid_layouts():
    """nets/pwcnet._regrid: from the d_from*d_from sub-grid layout of a dilated convolution straight to the d_to layout of
    the next one equals batch-to-space followed by space-to-batch, for every pair of layouts the context network chains
    (1 -> 2 -> 4 -> 8 -> 16 -> 1) and for layouts that do not divide each other."""
    from pcfa_amd.nets.pwcnet import _regrid

    def to_grid(x, d):
        B, C, H, W = x.shape
        return x.reshape(B, C, H // d, d, W // d, d).permute(0, 3, 5, 1, 2, 4).reshape(B * d * d, C, H // d, W // d)
    for B in (1, 2):
        x = torch.randn(B, 3, 48, 96)
        for df in (1, 2, 4, 8, 16, 3):
            for dt in (1, 2, 4, 8, 16, 3, 6):
                if 48 % df or 96 % df or 48 % dt or 96 % dt:
                    continue
                assert torch.equal(_regrid(to_grid(x, df), df, dt, B), to_grid(x, dt)), (df, dt)
    xs = to_grid(torch.randn(1, 2, 16, 32), 2).requires_grad_(True)
    _regrid(xs, 2, 8, 1).square().sum().backward()     # one strided copy forward, one backward
    assert torch.equal(xs.grad, 2 * xs.detach())


def test_fp64_port_is_the_arbiter_it_claims_to_be():
    """tools/trajectory_closure_parity.py judges a point where GPU and fp32 port disagree by the port in fp64: the same host
    code with every parameter, buffer, cached weight and stepper tensor in float64 (oracle operators are dtype-generic for
    PWC-Net).  The fp64 closure must actually run in fp64 and agree with the fp32 one to fp32 rounding."""
    import bench
    from tools.trajectory_closure_parity import to_double
    kw = dict(seed=0, boxconstraint="clipping", joint=True)
    a = bench.AttackStepper("PWCNet", 128, 192, torch.device("cpu"), **kw)
    b = bench.AttackStepper("PWCNet", 128, 192, torch.device("cpu"), **kw)
    to_double(b)
    g = torch.Generator().manual_seed(3)
    pert = [0.01 * torch.randn(p.shape, generator=g) for p in a.params]
    grads = []
    for st in (a, b):
        with torch.no_grad():
            for p, d in zip(st.params, pert):
                p.add_(d.to(p.dtype))
        st.optimizer.zero_grad()
        loss = st._closure_body()
        grads.append((loss, torch.cat([p.grad.flatten() for p in st.params])))
    assert grads[1][0].dtype == torch.float64 and grads[1][1].dtype == torch.float64
    assert all(p.dtype == torch.float64 for p in b.model.parameters())
    assert abs(float(grads[0][0]) - float(grads[1][0])) <= 1e-5 * abs(float(grads[1][0]))
    assert rel_l2(grads[0][1].double(), grads[1][1]) < 1e-3


def test_seeded_inputs_are_reproducible():
    """The generated leaves equal the stored ones (so GPU tests can regenerate instead of loading)."""
    g = load_golden("closure_raft")
    i1, i2 = closure_util.test_images(1, 128, 160)
    assert np.array_equal(i1.numpy().astype(np.uint8), g["image1"])
    r = closure_util.run_closure("RAFT", 128, 160, "change_of_variables", False, "zero", "aee", 1,
                                 torch.device("cpu"))
    assert max_abs(r["leaves"][0], t(g["leaf0"])) < 1e-6


def _args(**kw):
    base = dict(net="RAFT", steps=5, joint_perturbation=False, boxconstraint="change_of_variables",
                delta_bound=0.005, target="zero", custom_target_path="", loss="aee", save_frequency=1,
                small_save=False, no_save=True, unregistered_artifacts=True, universal_perturbation=False,
                mu=-1, weights="random:1234")
    base.update(kw)
    return Namespace(**base)


def test_pcfa_attack_trajectory_within_reference_noise():
    """5 steps (55 closures) of pcfa_attack vs the reference's own run; tolerance = 3x the reference's
    self-noise between 8 and 3 CPU threads, floored at 1e-3."""
    g = load_golden("trajectory_raft")
    ref8, ref3 = g["threads8"], g["threads3"]
    model = closure_util.load_model("RAFT", True, torch.device("cpu"))
    res = attack_PCFA.pcfa_attack(model, t(g["image1"].astype(np.float32)), t(g["image2"].astype(np.float32)),
                                  torch.zeros(1, 2, 128, 160), 0, None, 1e-7, torch.device("cpu"), False,
                                  2500. / 0.005, _args())
    res = np.array([np.nan if v is None else float(v) for v in res])
    # unattacked statistics are deterministic
    assert abs(res[1] - ref8[1]) < 1e-4
    for idx in (4, 5, 8, 9, 10, 11):  # aee_adv_tgt, aee_adv_pred, l2_delta12, and the three best-iterate values
        tol = max(1e-3, 3 * abs(ref8[idx] - ref3[idx])) * max(1.0, abs(ref8[idx]))
        assert abs(res[idx] - ref8[idx]) <= tol, (idx, res[idx], ref8[idx], ref3[idx])


def test_universal_attack_within_reference_noise():
    """attack_l2_universal (2 batches x 2 pairs x 2 steps = 44 closures, ONE optimiser across batches) vs the
    reference's own attack_l2_universal run (attack_PCFA.py:297-566); tolerance by the D10 noise rule."""
    g = load_golden("universal_raft")
    args, loader = closure_util.universal_case(g)
    res = attack_PCFA.attack_l2_universal(args, data_loader=loader, has_gt=False)
    assert res["collectives"] == 0 and not res["graphed"]      # single process on CPU: eager, no collective
    closure_util.check_universal_against_golden(res, g, rel_l2)


def test_best_iterate_rule_and_schedule():
    """10 closure evaluations + 1 re-prediction per step and the reference's selection rule."""
    calls = {"n": 0}
    model = closure_util.load_model("SpyNet", True, torch.device("cpu"))
    orig = model.forward

    def counting(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)

    model.forward = counting
    try:
        i1, i2 = closure_util.test_images(5, 64, 64)
        res = attack_PCFA.pcfa_attack(model, i1, i2, torch.zeros(1, 2, 64, 64), 0, None, 1e-7, torch.device("cpu"),
                                      False, 2500. / 0.005, _args(net="SpyNet", steps=3))
    finally:
        model.forward = orig
    assert calls["n"] == 1 + 3 * 11
    assert len(res) == 12
    best_l2, last_l2 = res[11], res[8]
    assert best_l2 <= 0.005 or best_l2 <= last_l2 + 1e-12


def test_adapter_api_shapes_and_padding():
    pad = ownutilities.InputPadder((1, 3, 436, 1024))
    assert pad._pad == [0, 0, 2, 2]
    x = torch.zeros(1, 3, 436, 1024)
    assert pad.pad(x)[0].shape[-2:] == (440, 1024) and pad.unpad(pad.pad(x)[0]).shape == x.shape
    p64 = ownutilities.InputPadder((1, 3, 375, 1242), divisor=64)
    assert p64._pad == [19, 19, 4, 5]
    assert ownutilities.model_takes_unit_input("PWCNet") and not ownutilities.model_takes_unit_input("RAFT")
    padder, [a, b] = ownutilities.preprocess_img("SpyNet", 255 * torch.ones(1, 3, 100, 150), torch.zeros(1, 3, 100, 150))
    assert a.shape[-2:] == (128, 192) and float(a.max()) == 1.0
    with pytest.raises(ValueError):
        from pcfa_amd.helper_functions import targets
        targets.get_target("nope", torch.zeros(1, 2, 4, 4))
    with pytest.raises(NotImplementedError):
        from pcfa_amd.helper_functions import losses
        losses.loss_delta_constraint(torch.zeros(1, 2, 4, 4), torch.zeros(1, 2, 4, 4), torch.zeros(3), torch.zeros(3),
                                     f_type="l1")


def test_joint_cov_is_rejected():
    model = closure_util.load_model("SpyNet", True, torch.device("cpu"))
    i1, i2 = closure_util.test_images(5, 64, 64)
    with pytest.raises(ValueError, match="joint_perturbation"):
        attack_PCFA.pcfa_attack(model, i1, i2, torch.zeros(1, 2, 64, 64), 0, None, 1e-7, torch.device("cpu"), False,
                                5e5, _args(net="SpyNet", joint_perturbation=True))


def _cli(**kw):
    base = dict(net="SpyNet", weights="random:1234", dataset="Synthetic", dataset_stage="evaluation", small_run=False,
                synthetic_size="64x64", synthetic_pairs=2, dstype="final", output_folder="experiment_data",
                small_save=False, save_frequency=1, no_save=True, unregistered_artifacts=True,
                joint_perturbation=False, steps=2, universal_perturbation=False, boxconstraint="clipping",
                batch_size=2, delta_bound=0.005, mu=-1, epochs=2, target="zero", custom_target_path="", loss="aee",
                epsilon=0.00025, perturbation_sourcefolder=None, origin_net=None)
    base.update(kw)
    return Namespace(**base)


def test_fgsm_step_and_driver():
    """attack_FGSM.py:21-56 semantics + the per-pair driver (2 pairs x 2 iterations)."""
    from pcfa_amd import attack_FGSM
    a, b = torch.full((1, 3, 2, 2), 0.5), torch.full((1, 3, 2, 2), 0.9999)
    g1, g2 = torch.tensor([[[[1., -1.], [0., 2.]]]]).expand(1, 3, 2, 2), -torch.ones(1, 3, 2, 2)
    p1, p2 = attack_FGSM.fgsm_attack_step(a, b, 0.01, g1, g2)
    assert torch.allclose(p1[0, 0], torch.tensor([[0.49, 0.51], [0.5, 0.49]])) and float(p2.max()) == 1.0
    j1, j2 = attack_FGSM.fgsm_attack_step(a, b, 0.01, g1, g2, common_perturb=True)
    assert torch.equal(a - j1, (b - j2).clamp(max=0.01)) or torch.allclose(a - j1, 0.01 * (0.5 * (g1 + g2)).sign())
    res = attack_FGSM.attack(_cli(loss="mse"))
    assert res["pairs"] == 2 and 0 < res["l2_delta-avg"] <= 2 * 0.00025 + 1e-9
    assert res["aee_predadv-tgt"] <= res["aee_pred-tgt"] + 1e-3      # two descent steps do not move away


def test_universal_artifacts_round_trip_through_evaluate(tmp_path):
    """attack_l2_universal writes NNNNN_delta1_e{E}.npy; evaluate_PCFA reads them back, also across padding families."""
    from pcfa_amd import evaluate_PCFA
    args = _cli(universal_perturbation=True, no_save=False, output_folder=str(tmp_path), steps=1, epochs=2)
    res = attack_PCFA.attack_l2_universal(args)
    run_dir = None
    for root, dirs, files in os.walk(str(tmp_path)):
        if os.path.basename(root) == "patches" and any(f.endswith("_delta1_e1.npy") for f in files):
            run_dir = os.path.dirname(root)
    assert run_dir is not None
    epochs, d1, d2 = evaluate_PCFA.extract_epoch_patchlist(run_dir)
    assert epochs == 2 and len(d1) == 2 and len(d2) == 2
    assert np.array_equal(np.load(d1[-1]), res["delta1"].numpy())
    ev = evaluate_PCFA.eval_l2_universal(_cli(universal_perturbation=True, perturbation_sourcefolder=run_dir,
                                              origin_net="SpyNet"))
    assert len(ev) == 2 and ev[0]["images"] == 2 and np.isfinite(ev[1]["epoch_aee_pred-predadv"])
    # cross-family re-padding: SpyNet (div 64) -> RAFT (div 8) keeps the un-padded core of delta
    delta = torch.from_numpy(np.load(d1[0]))                       # [3, 64, 64] (64 is already a multiple of 64)
    img = torch.zeros(1, 3, 60, 70)
    padded64 = torch.randn(3, 64, 128)
    re8 = evaluate_PCFA.convert_perturbationsizes(padded64, img, "SpyNet", "RAFT")
    assert re8.shape == (1, 3, 64, 72) and torch.equal(re8[0, :, 2:62, 1:71], padded64[:, 2:62, 29:99])
    assert evaluate_PCFA.convert_perturbationsizes(delta, img, "RAFT", "GMA") is delta


def test_call_spy_sees_calls_of_every_operator_module():
    """tools patched `hip_ops._call`, which the operator modules (bound `from .core import _call`) never
    look at.  `ops.core.set_call_spy` is the hook: every module's launches go through it while it is set."""
    from pcfa_amd import hip_ops
    from pcfa_amd.ops import attack_math, conv, corr, gru, pwc
    seen = []
    hip_ops.set_call_spy(lambda name, args, invoke: seen.append((name, args)))
    try:
        for i, mod in enumerate((attack_math, conv, corr, gru, pwc)):
            mod._call("pcfa_entry_%d" % i, i)
    finally:
        hip_ops.set_call_spy(None)
    assert [n for n, _ in seen] == ["pcfa_entry_%d" % i for i in range(5)] and seen[3][1] == (3,)


def test_lanes_are_thread_local_without_a_gpu_and_nest():
    """ops.core.lane(k): the scratch-buffer lane of attack_PCFA.PairsInFlight.  On the host it is a thread-local that nests
    and is restored on exit; a stream binding (bind_stream) overrides it only for streams that were bound (GPU test:
    test_pairs_in_flight_bit_identical_to_solo)."""
    import threading
    from pcfa_amd import ops
    assert ops.core.current_lane() == 0
    seen = {}

    def worker(k):
        with ops.core.lane(k):
            with ops.core.lane(k + 10):
                seen[(k, "inner")] = ops.core.current_lane()
            seen[(k, "outer")] = ops.core.current_lane()
        seen[(k, "after")] = ops.core.current_lane()
    ts = [threading.Thread(target=worker, args=(k,)) for k in (1, 2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert seen == {(1, "inner"): 11, (1, "outer"): 1, (1, "after"): 0, (2, "inner"): 12, (2, "outer"): 2, (2, "after"): 0}
    assert ops.core.current_lane() == 0


def test_config_rejects_bad_values_and_reads_the_cache_cap():
    import dataclasses
    import pytest
    from pcfa_amd import config
    assert config.DEFAULT.max_cached_shapes >= 1
    with pytest.raises(ValueError):
        dataclasses.replace(config.DEFAULT, max_cached_shapes=0)
    with pytest.raises(ValueError):
        dataclasses.replace(config.DEFAULT, gma_gemm="cublas")


def test_committed_parity_records_are_consistent(tmp_path):
    """The r05 end-of-attack matrix and its fp64-arbiter records (profiles/r05/): every pair outside the schedule rule
    carries an arbiter record whose rule passed at every point; the RAFT config (BASELINE config 2, the bench workload) has all
    eight pairs inside; re-assembling the arbiter records reproduces the committed summary's counts."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    m = json.load(open(os.path.join(repo, "profiles", "r05", "schedule_parity_matrix.json")))
    by = {(c["net"], c["steps"]): c for c in m["configs"]}
    assert by[("RAFT", 20)]["pairs_ok"] == by[("RAFT", 20)]["pairs_total"] == 8
    assert by[("GMA", 20)]["pairs_total"] == 8
    for c in m["configs"]:
        assert sorted(c["pairs_outside"]) == sorted(c["pairs_outside_cleared_by_fp64_arbiter"]), (c["net"], c["steps"])
        assert c["pairs_outside_without_arbiter_record"] == []
        for r in c["pairs"]:
            if r.get("inside_all") is False:
                fa = r["fp64_arbiter"]
                assert fa["rule_ok_everywhere"] and fa["worst_point"]["gpu_vs_fp64"] <= fa["worst_point"]["tolerance"]
        # GPU-vs-port no larger than port-vs-port, read as distributions
        assert c["median_abs_difference_gpu_vs_port16_over_port16_vs_port8"] <= 1.0
    out = tmp_path / "arb.json"
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "parity_arbiter.py"), "assemble", "--dir",
                        os.path.join(repo, "profiles", "r05", "arbiter"), "--out", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = json.load(open(out)), json.load(open(os.path.join(repo, "profiles", "r05", "fp64_arbiter.json")))
    assert a["pairs_total"] == a["pairs_rule_ok"] and a["pairs_total"] >= 15
    assert b["pairs_rule_ok"] == b["pairs_total"]


def test_pairs_in_flight_refuses_library_closures_host_rule():
    """The host rule behind attack_PCFA.PairsInFlight's refusal (the GPU test shows the hang it prevents): closures that keep
    library kernels with per-handle workspaces -- GMA on rocBLAS products, SpyNet, FlowNet2 -- do not go in flight."""
    import dataclasses
    from types import SimpleNamespace
    from pcfa_amd import attack_PCFA
    from pcfa_amd import config as pcfa_config
    rule = attack_PCFA.PairsInFlight._refuse_shared_library_workspaces
    lib = SimpleNamespace(_pcfa_config=dataclasses.replace(pcfa_config.DEFAULT, gma_gemm="lib"))
    hip = SimpleNamespace(_pcfa_config=dataclasses.replace(pcfa_config.DEFAULT, gma_gemm="hip"))
    for net, model, ok in (("RAFT", lib, True), ("PWCNet", lib, True), ("GMA", hip, True), ("GMA", lib, False),
                           ("SpyNet", hip, False), ("FlowNet2", hip, False)):
        attack = SimpleNamespace(args=SimpleNamespace(net=net), model=model)
        if ok:
            rule(attack)
        else:
            with pytest.raises(ValueError, match="in flight"):
                rule(attack)
