#!/usr/bin/env python3
"""PCFA attack-steps/sec on MI355X (BASELINE.json metric; workload = configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one `--steps` iteration of pcfa_attack on one synthetic 436x1024 image pair with RAFT
(seeded random weights -- no checkpoints exist offline): L-BFGS(max_iter=10).step = 10 closure evaluations
(box transform -> RAFT forward -> AEE + L2 penalty -> backward) + 1 re-prediction forward + the step's metrics
(SURVEY.md D3).  The timed object is `pcfa_amd.attack_PCFA.PairAttack.step` -- the body of pcfa_attack's own loop,
not a restatement.  Inputs are resident in HBM before the timed region.  Every rank attacks its own pair
(disjoint perturbations, no data-path collective): weak scaling, value = N*K / max-over-ranks time.

    --universal   BASELINE config 5 instead: ONE perturbation pair shared by all images, data parallel over the
                  batch (`--pairs-per-gpu` pairs on every rank), one RCCL all-reduce of [grad delta1 | grad delta2 |
                  loss] per closure (`pcfa_amd.attack_PCFA.UniversalAttack.step`).  value = pair-steps/s =
                  N * pairs-per-gpu * K / time (weak scaling: per-GPU work fixed).
With N > 1 the default (per-pair) run also appends a short universal leg after the timed region
(`"universal": {...}` in the JSON line) so that a multi-GPU run exercises the all-reduce path too.

Started WITHOUT torchrun (`python bench.py --gpus N`, no RANK/WORLD_SIZE in the environment) the process becomes a
launcher: before anything touches the GPU it starts N fresh rank processes of this script (pcfa_amd.launch.spawn_ranks:
RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, 127.0.0.1 rendezvous), waits for them and exits non-zero if any rank failed;
rank 0 prints the one JSON line with `n_gpus: N` and every rank's own step time (`per_rank_ms_per_step`).

    --rehearse-cpu   launcher / collective rehearsal WITHOUT a GPU (tests/test_bench_launcher_cpu.py): every rank times
                     the CPU port (the cpu_baseline leg's objects: pcfa_amd host code + oracle operators) on a tiny
                     SpyNet pair over gloo and runs the universal leg; the line says `"rehearsal": "cpu-port"` and its
                     metric is `cpu_port_rehearsal_steps_per_sec` -- never a result.

The JSON line also carries
  roofline            the correlation-lookup forward kernel (the kernel BASELINE's north_star names): algorithmic
                      bytes/launch (20.44 MB at 55x128) / mean launch duration, against 8 TB/s HBM;
  kernels             the other hand-written kernels of the path with their own roofline rows (RAFT/GMA: pyramid GEMMs,
                      lookup backward, ...; PWCNet: cost volume forward/backward, warp);
  cpu_baseline        this repo's CPU port (pcfa_amd host code + oracle operators, torch fp32 on host cores) timed on
                      a bounded sample of the same workload, extrapolated to the 10+1 schedule;
  parity_vs_cpu_port  loss / gradient / flow of ONE closure of the same pair at the same (perturbed) variables on the
                      GPU and on the CPU port (the cpu_baseline leg's warm-up closure) -- parity at the BASELINE size.
"""
import argparse
import json
import os
import sys
import time
from argparse import Namespace

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak
PARITY_SEED, PARITY_SIGMA = 7, 0.02


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--net", default="RAFT", choices=["RAFT", "GMA", "PWCNet", "SpyNet", "FlowNet2"])
    ap.add_argument("--size", default="436x1024")
    ap.add_argument("--universal", action="store_true", help="time the universal attack (config 5) instead")
    ap.add_argument("--pairs-per-gpu", type=int, default=1, help="--universal: local batch size")
    ap.add_argument("--no-universal-leg", action="store_true", help="N > 1: skip the short universal leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="no GPU: rehearse launcher + collectives with the CPU port on a tiny workload (not a result)")
    ap.add_argument("--no-shared-forward-leg", action="store_true", help="skip the informational shared-forward leg")
    ap.add_argument("--no-pwcnet-leg", action="store_true", help="skip the PWC-Net (BASELINE config 4) leg")
    ap.add_argument("--no-pairs-in-flight-leg", action="store_true", help="skip the two-pairs-per-GPU leg")
    ap.add_argument("--no-gma-leg", action="store_true", help="skip the GMA (BASELINE config 3 network) leg")
    ap.add_argument("--channels-last", action="store_true", help="experiment: NHWC convolutions")
    ap.add_argument("--no-graph", action="store_true", help="launch the closure eagerly instead of replaying "
                                                            "its hipGraph")
    ap.add_argument("--miopen-find", action="store_true",
                    help="cudnn.benchmark = True (MIOpen exhaustive find; measured equal to the default "
                         "immediate mode on this workload, but costs ~60 s of warm-up)")
    ap.add_argument("--cpu-closures", type=int, default=8, help="closure evaluations in the CPU sample")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="host threads of the CPU baseline (0 = min(cores, 16): the 55x128-feature convolutions "
                         "stop scaling there; 64 threads measured 3x SLOWER than 16 on the 256-core box)")
    return ap.parse_args()


# --------------------------------------------------------------------------------------------------------------
# the product objects under test
# --------------------------------------------------------------------------------------------------------------
def attack_args(net, boxconstraint="change_of_variables", joint=False, universal=False, steps=1, target="zero"):
    """BASELINE config 2 flags (attack_PCFA.py CLI defaults): delta_bound 0.005, zero target, AEE."""
    return Namespace(net=net, steps=steps, joint_perturbation=joint, universal_perturbation=universal,
                     boxconstraint=boxconstraint, delta_bound=0.005, mu=-1., target=target, custom_target_path="",
                     loss="aee", save_frequency=1, small_save=False, no_save=True, unregistered_artifacts=True,
                     weights="random:1234", batch_size=1, epochs=1)


def load_model(net, device, cov, config=None):
    """config: a pcfa_amd.config.Config (None: config.DEFAULT) -- the build switches are fixed at construction."""
    from pcfa_amd import attack_PCFA
    from pcfa_amd.helper_functions import ownutilities
    unit = ownutilities.model_takes_unit_input(net)
    kw = {"eps_box": attack_PCFA.EPS_BOX} if cov else {}
    model = ownutilities.import_and_load(net, make_unit_input=not unit, variable_change=cov,
                                         make_scaled_input_model=True, device=device, weights="random:1234",
                                         config=config, **kw)
    model.eval()
    for p in model.parameters():
        p.requires_grad = False
    return model


def AttackStepper(net, h, w, device, seed, boxconstraint="change_of_variables", use_graph=False, model=None,
                  joint=False, target="zero", config=None):
    """`pcfa_amd.attack_PCFA.PairAttack` on one synthetic pair (what pcfa_attack builds per pair), graph off unless
    asked: call `.enable_graph()` -- exactly what pcfa_attack does on the GPU."""
    from pcfa_amd import attack_PCFA
    from pcfa_amd.helper_functions import datasets
    cov = boxconstraint == "change_of_variables"
    args = attack_args(net, boxconstraint, joint=joint, target=target)
    if model is None:
        model = load_model(net, device, cov, config)
    i1, i2, _ = datasets.synthetic_pair(seed, h, w)
    st = attack_PCFA.PairAttack(model, i1[None], i2[None], None, 0, attack_PCFA.EPS_BOX, device, False,
                                attack_PCFA.default_mu(args), args, use_graph=use_graph)
    st.nw1, st.nw2, st._closure_body = st.nw_input1, st.nw_input2, st.closure_body   # names the tools use
    return st


def UniversalStepper(net, h, w, device, seeds, use_graph=None, model=None):
    """`pcfa_amd.attack_PCFA.UniversalAttack` with this rank's slice of one global batch loaded."""
    from pcfa_amd import attack_PCFA
    from pcfa_amd.helper_functions import datasets, ownutilities
    args = attack_args(net, "clipping", universal=True)
    if model is None:
        model = load_model(net, device, False)
    pairs = [datasets.synthetic_pair(s, h, w) for s in seeds]
    im1 = torch.stack([p[0] for p in pairs])
    im2 = torch.stack([p[1] for p in pairs])
    _, [p1, p2] = ownutilities.preprocess_img(net, im1[:1], im2[:1])
    ua = attack_PCFA.UniversalAttack(model, p1[0], p2[0], device, attack_PCFA.default_mu(args), args,
                                     use_graph=use_graph)
    ua.begin_batch(im1, im2)
    return ua


def parity_closure(st):
    """loss, gradients and flow of ONE closure at the initial variables + PARITY_SIGMA * N(0,1) (seeded on the
    host, so the GPU and the CPU port evaluate the same point); variables restored afterwards."""
    g = torch.Generator().manual_seed(PARITY_SEED)
    saved = [p.detach().clone() for p in st.params]
    with torch.no_grad():
        for p in st.params:
            p.add_((PARITY_SIGMA * torch.randn(p.shape, generator=g)).to(p.device))
    graphed, st.graphed = st.graphed, None        # eager: the graph path is covered by the timed region + its test
    try:
        loss = float(st.closure())
        grads = torch.cat([p.grad.detach().flatten().cpu() for p in st.params])
        with torch.no_grad():
            flow = st.predict().detach().cpu().clone()
    finally:
        st.graphed = graphed
        st.closures -= 1
        with torch.no_grad():
            for p, s in zip(st.params, saved):
                p.copy_(s)
        for p in st.params:
            p.grad = None
    return {"loss": loss, "grads": grads, "flow": flow}


def parity_record(gpu, cpu):
    df = gpu["flow"] - cpu["flow"]
    return {"loss_rel": abs(gpu["loss"] - cpu["loss"]) / abs(cpu["loss"]),
            "grad_rel_l2": float((gpu["grads"] - cpu["grads"]).norm() / cpu["grads"].norm()),
            "flow_aee": float(df.pow(2).sum(1).sqrt().mean()), "flow_max_abs": float(df.abs().max()),
            "loss_gpu": gpu["loss"], "loss_cpu_port": cpu["loss"],
            "point": "one closure at the initial variables + %g*N(0,1) (host seed %d), same pair, same weights"
                     % (PARITY_SIGMA, PARITY_SEED),
            "tolerance": "tests/test_gpu_parity.py::test_closure_at_baseline_size_vs_cpu_port: flow AEE <= 1e-3, "
                         "loss 1e-4, gradient 1e-2 relative L2"}


# --------------------------------------------------------------------------------------------------------------
# per-kernel roofline rows
# --------------------------------------------------------------------------------------------------------------
def event_overhead_us(reps=200):
    """Mean interval of an event BRACKET (record, launch, record) around an empty kernel: what per-launch
    figures would carry if they were taken with bracketing events instead of dispatch-attached ones
    (informational; nothing is subtracted anywhere)."""
    from pcfa_amd import hip_ops
    prof = hip_ops.LaunchProfiler(names=["pcfa_null_launch"])
    hip_ops.set_launch_profiler(prof)
    for _ in range(reps):
        hip_ops._call("pcfa_null_launch")
    hip_ops.set_launch_profiler(None)
    return prof.summary()["pcfa_null_launch"][0]


def null_launch_device_us(reps=50):
    """Device-side duration of a kernel that does nothing (dispatch timestamps of the HIP activity tracer): the floor under
    every launch of a dependent chain."""
    from torch.autograd import DeviceType
    from torch.profiler import ProfilerActivity, profile
    from pcfa_amd import hip_ops
    hip_ops._call("pcfa_null_launch")
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(reps):
            hip_ops._call("pcfa_null_launch")
        torch.cuda.synchronize()
    d = [ev.time_range.elapsed_us() for ev in prof.events() if ev.device_type == DeviceType.CUDA and "null_kernel" in ev.name]
    return sum(d) / len(d) if d else None


TRAFFIC_SOURCE = "profiles/lookup_traffic.json (offline rocprofv3 --pmc passes; committed constant, not measured by this run)"


def lookup_traffic(kernel="corr_lookup_fwd"):
    """HBM bytes per launch of the lookup kernel from rocprofv3 PMC counters (collected offline by
    tools/pmc_traffic.sh with the guide's gfx950 corrections, committed under profiles/); None if absent."""
    path = os.path.join(REPO, "profiles", "lookup_traffic.json")
    try:
        return json.load(open(path))[kernel]["traffic_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def lookup_algorithmic_bytes(hf, wf, levels=4, radius=4):
    """unique texels (Q * levels * (2r+2)^2 * 4 B) + coords + output."""
    q = hf * wf
    n1 = 2 * radius + 1
    return q * levels * (2 * radius + 2) ** 2 * 4 + q * 2 * 4 + q * levels * n1 * n1 * 4


def _row(label, bound, work, us, n, note=None):
    if bound == "hbm":
        ach, peak, unit = work / (us * 1e-6) / 1e9, HBM_PEAK_GBS, "GB/s"
    else:
        ach, peak, unit = work / (us * 1e-6) / 1e12, MFMA_F32_PEAK_TFLOPS, "TFLOP/s"
    r = {"kernel": label, "bound": bound, "per_launch": work, "mean_launch_us": round(us, 2), "launches_timed": n,
         "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4)}
    if note:
        r["note"] = note
    return r


def kernel_table(timings, hf, wf, hp, wp, dim=256, levels=4, radius=4):
    """RAFT/GMA: algorithmic work per launch (SURVEY 8d figures, stated in DESIGN.md) / mean dispatch duration."""
    q = hf * wf
    n1 = 2 * radius + 1
    out_b = q * levels * n1 * n1 * 4
    win_b = q * levels * (2 * radius + 2) ** 2 * 4
    gemm = 2.0 * q * q * dim  # level 0 only: the pooled levels cost no multiplies in the reference (avg_pool2d)
    img = 3 * hp * wp * 4
    work = {  # label -> (bound, algorithmic bytes or flop per launch)
        "corr_lookup_fwd": ("hbm", win_b + q * 8 + out_b),
        "corr_lookup_bwd": ("hbm", out_b + q * 8 + 2 * win_b),
        # lookup + convc1 + ReLU fused: W[256][324] . taps[324][Q] on the fp32 matrix cores (arithmetic intensity
        # 2*256*324*Q flop / (texels + coords + 256*Q*4 B out) = 62 flop/B, above the fp32-matrix ridge of ~20)
        "corr_lookup_convc1_fwd": ("mfma", 2.0 * 256 * levels * n1 * n1 * q),
        "corr_lookup_convc1_bwd": ("mfma", 2.0 * 256 * levels * n1 * n1 * q),
        "corr_pyramid_gemm_fwd": ("mfma", gemm),
        "corr_pyramid_gemm_dfmap1": ("mfma", gemm),
        "corr_pyramid_gemm_df2ext": ("mfma", gemm),
        # the encoders' stride-2 layers (csrc/conv_strided.hip): feature encoder on both images + context encoder on one
        # = 3 images per closure; mean over the launches of a closure (B = 2 and B = 1, two layers for the block entries)
        "conv_s2_stem_fwd": ("mfma", 1.5 * 2.0 * 147 * 64 * (hp // 2) * (wp // 2)),
        "conv_s2_stem_bwd": ("mfma", 1.5 * 2.0 * 147 * 64 * (hp // 2) * (wp // 2)),
        "conv_s2_block_entry_fwd": ("mfma", 0.75 * 2.0 * 10 * (64 * 96 * (hp // 4) * (wp // 4) + 96 * 128 * (hp // 8) * (wp // 8))),
        "conv_s2_block_entry_bwd": ("mfma", 0.75 * 2.0 * 10 * (64 * 96 * (hp // 4) * (wp // 4) + 96 * 128 * (hp // 8) * (wp // 8))),
        "box_transform_fwd": ("hbm", 4 * img),
        "box_transform_bwd": ("hbm", 6 * img),
        "gru_gates_fwd": ("hbm", 8 * q * 128 * 4),
        "gru_update_fwd": ("hbm", 6 * q * 128 * 4),
    }
    s2 = ("mean over the closure's launches (B = 2 feature encoder, B = 1 context encoder%s); 3x3 conv1 + fused 1x1 "
          "downsample = 10 taps; the fp32 matrix pipe sustains ~1.9 GHz, i.e. ~124 of the 157.3 TFLOP/s peak")
    notes = {
        "conv_s2_stem_fwd": "7x7 stem, 147 real of 168 issued k per pixel (window rows paired for the MFMA k = 2)",
        "conv_s2_stem_bwd": "on the 16x16x4 MFMA: 12 of 16 rows carry (channel, parity) classes, 49 of 64 (tap, class) "
                            "products are non-zero -- `achieved` counts the 147 real taps only",
        "conv_s2_block_entry_fwd": s2 % ", 64->96 and 96->128",
        "conv_s2_block_entry_bwd": s2 % ", 64->96 and 96->128",
        "corr_pyramid_gemm_dfmap1": "algorithmic = the dense product the reference's autograd runs; the kernel skips the "
                                    "slab columns no lookup window touched (exact zeros of dpyr), so `achieved` is NOT "
                                    "the matrix pipe's rate and may exceed the peak",
        "corr_pyramid_gemm_df2ext": "as corr_pyramid_gemm_dfmap1 (query rows no window reaches are skipped)",
    }
    return [_row(label, work[label][0], work[label][1], us, n, notes.get(label))
            for label, (us, n) in sorted(timings.items()) if label in work]


PWC_LEVEL_CHANNELS = {2: 32, 3: 64, 4: 96, 5: 128, 6: 196}  # models/PWCNet/PWCNet.py:76-93
PWC_TRAFFIC_FILE = "profiles/r05/pwc_traffic_after_xcd_map.json"


def pwc_traffic():
    """HBM-side bytes per launch of the cost-volume kernels per level, from the committed rocprofv3 --pmc passes
    (tools/pmc_traffic_pwc.sh: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE): {"spatial_corr_fwd": {2: bytes, ...}, ...}.
    One key per launch shape; a kernel's shapes in descending traffic are levels 2, 3, 4, 5, 6.  A constant, not measured by the
    run (the rows say so)."""
    try:
        t = json.load(open(os.path.join(REPO, PWC_TRAFFIC_FILE)))
    except (OSError, ValueError):
        return {}
    out = {}
    for label, frag in (("spatial_corr_fwd", "scorr9_fwd"), ("spatial_corr_bwd", "scorr9_bwd")):
        rows = sorted((v["traffic_bytes"] for k, v in t.items() if k.startswith(frag)), reverse=True)
        if len(rows) == 5:
            out[label] = dict(zip((2, 3, 4, 5, 6), rows))
    return out


def pwc_kernel_table(timings, hp, wp):
    """PWC-Net: the five cost volumes of one closure together (13.9 MB in + 13.3 MB out forward at
    384x1280; backward reads grad_out + both maps and writes both gradients).  Rows are per CLOSURE: the five
    launches (levels 6..2) have different sizes, so bytes and durations are summed over them."""
    fwd = bwd = 0
    for lvl, c in PWC_LEVEL_CHANNELS.items():
        px = (hp >> lvl) * (wp >> lvl)
        fwd += (2 * c + 81) * px * 4
        bwd += (81 + 2 * c + 2 * c) * px * 4
    rows = []
    traffic = pwc_traffic()
    src = PWC_TRAFFIC_FILE + " (offline rocprofv3 --pmc passes; committed constant, not measured by this run)"
    if "spatial_corr_fwd" in timings:
        us, n = timings["spatial_corr_fwd"]
        rows.append(_row("spatial_corr_fwd (5 levels)", "hbm", fwd, us * 5, n // 5, "per closure: 5 launches summed"))
        if "spatial_corr_fwd" in traffic:
            rows[-1].update(traffic=sum(traffic["spatial_corr_fwd"].values()), traffic_source=src)
    if "spatial_corr_bwd" in timings:
        us, n = timings["spatial_corr_bwd"]
        rows.append(_row("spatial_corr_bwd (5 levels, both gradients per launch)", "hbm", bwd, us * 5, n // 5,
                         "per closure: 5 launches summed"))
        if "spatial_corr_bwd" in traffic:
            rows[-1].update(traffic=sum(traffic["spatial_corr_bwd"].values()), traffic_source=src)
    # per level: the forward runs levels 6, 5, 4, 3, 2 in this order in every forward pass, the backward
    # 2, 3, 4, 5, 6 -- the position of a launch in the traced sequence names its level
    seqs = getattr(graph_replay_kernel_times, "sequences", {})
    null_us = getattr(pwc_kernel_table, "null_launch_us", None)
    for label, order, per_px in (("spatial_corr_fwd", (6, 5, 4, 3, 2), lambda c: (2 * c + 81) * 4),
                                 ("spatial_corr_bwd", (2, 3, 4, 5, 6), lambda c: (81 + 4 * c) * 4)):
        seq = seqs.get(label, [])
        if len(seq) % 5 or not seq:
            continue
        for pos, lvl in enumerate(order):
            d = seq[pos::5]
            us = sum(d) / len(d)
            nbytes = per_px(PWC_LEVEL_CHANNELS[lvl]) * (hp >> lvl) * (wp >> lvl)
            note = "level %d: %d x %d pixels, %d channels" % (lvl, hp >> lvl, wp >> lvl, PWC_LEVEL_CHANNELS[lvl])
            if null_us is not None and us < 2.5 * null_us:
                note += "; floor-bound: a launch that does nothing takes %.1f us here" % null_us
            rows.append(_row("%s level %d" % (label, lvl), "hbm", nbytes, us, len(d), note))
            if lvl in traffic.get(label, {}):
                rows[-1].update(traffic=traffic[label][lvl], traffic_source=src)
    if "pwc_warp_fwd" in timings:
        us, n = timings["pwc_warp_fwd"]
        w = sum((2 * PWC_LEVEL_CHANNELS[l] + 2) * (hp >> l) * (wp >> l) * 4 for l in (2, 3, 4, 5))
        rows.append(_row("pwc_warp_fwd (4 levels)", "hbm", w, us * 4, n // 4, "per forward: 4 launches summed"))
    return rows


TRACED = {  # kernel-name fragment -> label
    "corr_lookup_fwd_kernel": "corr_lookup_fwd", "corr_lookup_bwd_kernel": "corr_lookup_bwd",
    "corr_lookup_convc1_fwd_kernel": "corr_lookup_convc1_fwd", "corr_lookup_convc1_bwd_kernel": "corr_lookup_convc1_bwd",
    "gemm_f32_mfma_kernel<true, true,": "corr_pyramid_gemm_fwd",
    "corr_pyramid_pool_gemm_kernel": "corr_pyramid_gemm_fwd",   # levels 1-2 pooled in the epilogue (W % 16 == 0)
    "gemm_f32_mfma_kernel<false, false,": "corr_pyramid_gemm_dfmap1",
    "gemm_f32_mfma_sparse_kernel<false>": "corr_pyramid_gemm_dfmap1",   # only the slab columns the lookup windows touched
    "gemm_f32_mfma_sparse_kernel<true>": "corr_pyramid_gemm_df2ext",
    "corr_pyramid_unpool_gemm_kernel<false>": "corr_pyramid_gemm_dfmap1",   # dpyr un-pooled in the operand loader
    "corr_pyramid_unpool_gemm_kernel<true>": "corr_pyramid_gemm_df2ext",
    "gemm_f32_mfma_kernel<false, true,": "corr_pyramid_gemm_df2ext",
    "box_fwd_kernel": "box_transform_fwd", "box_bwd_kernel": "box_transform_bwd",
    "gru_gates_fwd_kernel": "gru_gates_fwd", "gru_update_fwd_kernel": "gru_update_fwd",
    "scorr9_fwd_kernel": "spatial_corr_fwd", "scorr9_bwd_kernel": "spatial_corr_bwd",
    "pwc_warp_fwd_kernel": "pwc_warp_fwd", "pwc_warp_bwd_kernel": "pwc_warp_bwd",
    "pwc_warp_bwd_det_kernel": "pwc_warp_bwd",
    "conv_s2_fwd_kernel<(anonymous namespace)::S2Cfg<7, 7": "conv_s2_stem_fwd",
    "conv_s2_fwd_kernel<(anonymous namespace)::S2Cfg<3, 3": "conv_s2_block_entry_fwd",
    "conv_s2_bwd_kernel": "conv_s2_block_entry_bwd", "conv_s2_stem_bwd_kernel": "conv_s2_stem_bwd",
    # the kernels that own the step: family rows from hip_ops' work recorder (family_rows below)
    "conv3x3_winograd_kernel": "conv3x3_winograd", "conv3x3_f43_kernel": "conv3x3_f43",
    # the split-K finish launch is booked under the family whose partial sums it adds (template argument FAMILY)
    "f43_finish_kernel<0, 23>": "conv3x3_winograd", "f43_finish_kernel<1, 23>": "conv3x3_winograd",
    "f43_finish_kernel<2, 23>": "conv3x3_winograd", "f43_finish_kernel": "conv3x3_f43",
    "sc5_wino_kernel": "sepconv5_winograd", "sepconv5_kernel": "sepconv5_direct",
    "instnorm_stats_kernel<false>": "instnorm_fwd", "instnorm_apply_kernel<false>": "instnorm_fwd",
    "instnorm_stats_kernel<true>": "instnorm_bwd", "instnorm_apply_kernel<true>": "instnorm_bwd",
    "instnorm_plane_kernel<256, 7, false>": "instnorm_fwd", "instnorm_plane_kernel<1024, 7, false>": "instnorm_fwd",
    "instnorm_plane_kernel<256, 7, true>": "instnorm_bwd", "instnorm_plane_kernel<1024, 7, true>": "instnorm_bwd",
    "conv3x3_fewout_fwd": "conv3x3_fewout_fwd", "conv3x3_fewout_bwd_kernel": "conv3x3_fewout_bwd",
    "leaky_relu_bwd_kernel": "leaky_relu_bwd", "relu_bwd2_kernel": "relu_bwd2", "relu_bwd_kernel": "relu_bwd",
    "add_relu_kernel": "add_relu_fwd", "conv_fewin_packed_fwd_kernel": "conv_fewin_fwd",
    # the optimiser (pcfa_amd/csrc/lbfgs_gram.hip, lbfgs.hip)
    "gram_pass_kernel": "lbfgs_gram_pass", "gram_direction_kernel": "lbfgs_gram_direction",
    "gram_reduce_kernel": "lbfgs_small", "gram_coeff_kernel": "lbfgs_gram_coeff",
    "gram_direction_final_kernel": "lbfgs_small", "gram_reset_kernel": "lbfgs_small",
    "lbfgs_step_kernel": "lbfgs_two_loop_sweep", "lbfgs_pair_kernel": "lbfgs_small",
    "lbfgs_pair_final_kernel": "lbfgs_small",
}
LBFGS_LABELS = ("lbfgs_gram_pass", "lbfgs_gram_direction", "lbfgs_gram_coeff", "lbfgs_small", "lbfgs_two_loop_sweep")


def lbfgs_record(traced, st, m):
    """Device time of the optimiser's own kernels in the traced attack step + what bounds them (m = pairs in the
    history during that step)."""
    if not traced:
        return None
    rows = {k: {"mean_launch_us": round(traced[k][0], 2), "launches": traced[k][1]} for k in LBFGS_LABELS if k in traced}
    total_ms = sum(traced[k][0] * traced[k][1] for k in LBFGS_LABELS if k in traced) * 1e-3
    opt = st.optimizer
    rec = {"lbfgs_ms_per_step": total_ms, "history": m, "direction": getattr(opt, "direction", None), "kernels": rows,
           "note": "device time of the optimiser's kernels in one attack step (10 iterations) at the history above; "
                   "the two host synchronisations per iteration and torch's parameter update are not in it"}
    if m and "lbfgs_gram_pass" in traced:
        n = sum(p.numel() for p in st.params)
        for k in ("lbfgs_gram_pass", "lbfgs_gram_direction"):
            nbytes = (2 * m + 3) * n * 4          # every stored vector once + g, g_prev/d in and out
            rows[k].update(bound="hbm", bytes_per_launch=nbytes,
                           achieved_GBs=round(nbytes / (traced[k][0] * 1e-6) / 1e9, 1),
                           frac=round(nbytes / (traced[k][0] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4))
    return rec


def graph_replay_kernel_times(st):
    """Durations of the pcfa_amd kernels INSIDE the hipGraph replays of one more attack step, from the dispatch
    timestamps the HIP runtime's activity tracer (roctracer, through torch.profiler) records for every kernel of
    the stream -- the source rocprofv3's kernel trace reads.  hipEvents cannot ride inside a captured graph, and
    the same kernels run slower in an eagerly launched step (bench reports both).  label -> (mean us, launches)."""
    from torch.autograd import DeviceType
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        st.step()
        torch.cuda.synchronize()
    acc, seqs = {}, {}
    total = covered = 0.0
    launches = 0
    evs = sorted((ev for ev in prof.events() if ev.device_type == DeviceType.CUDA), key=lambda e: e.time_range.start)
    for ev in evs:
        us = ev.time_range.elapsed_us()
        total += us
        launches += 1
        for frag, label in TRACED.items():
            if frag in ev.name:
                a = acc.setdefault(label, [0.0, 0])
                a[0] += us
                a[1] += 1
                covered += us
                if label in ("spatial_corr_fwd", "spatial_corr_bwd"):
                    seqs.setdefault(label, []).append(us)   # launch order: per-level rows (pwc_kernel_table)
                break
    graph_replay_kernel_times.sequences = seqs
    graph_replay_kernel_times.coverage = {"device_ms_per_step": total * 1e-3, "launches_per_step": launches,
                                          "in_kernel_rows_ms": covered * 1e-3, "frac": covered / total if total else None,
                                          "note": "device time of the traced attack step (10 closure replays + re-prediction "
                                                  "+ optimiser) that belongs to kernels with a roofline row in `kernels` / "
                                                  "`kernel_families` / `lbfgs`"}
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items() if v[1]}


FAMILY_NOTES = {
    "conv3x3_winograd": "Winograd F(2x2,3x3): every 3x3 / stride-1 convolution and data gradient that the policy leaves on it; "
                        "issued = direct-form flop / 2.25",
    "conv3x3_f43": "Winograd F(4x4,3x3) incl. its split-K finish launches; issued = direct-form flop / 4 "
                   "(fp32 rounding 2e-6 relative per layer against 3.5e-7 for F(2x2,3x3))",
    "sepconv5_winograd": "SepConvGRU gate convolutions + fused GRU epilogues as 1-D Winograd F(2,5); issued = direct-form flop x 0.6",
    "sepconv5_direct": "SepConvGRU gate convolutions, direct implicit GEMM (shapes the Winograd kernel does not take)",
    "corr_pyramid_gemm_dfmap1": "sparse product over the K segments the lookup windows touched: issued = executed MFMA flop "
                                "(segments x 128-wide blocks, read back from the kernel's own segment table), direct = the "
                                "dense product the reference's autograd runs",
    "corr_pyramid_gemm_df2ext": "as corr_pyramid_gemm_dfmap1 (K = hull of the query rows that reach the block's tile rows)",
    "instnorm_fwd": "one launch per call on the 55x128 / 110x256 stages (plane in registers), statistics + apply launches on "
                    "220x512; algorithmic bytes = x in + y out",
    "instnorm_bwd": "as instnorm_fwd; algorithmic bytes = x + grad_out in + grad_x out",
    "conv3x3_fewout_fwd": "flow-prediction convolutions (2 output channels): the input once + the output",
    "conv3x3_fewout_bwd": "their data gradient: grad_out + the streamed grad_x",
    "conv_fewin_fwd": "relu(convf1(flow)): 7x7 on two input channels as an im2col-in-LDS MFMA GEMM (3.6 MB out: launch-bound)",
}


def family_rows(traced, work):
    """Roofline rows per kernel FAMILY of one attack step: work from hip_ops' recorder (one eagerly launched step: the
    same launches as a replayed one), device time from the graph-replay trace.  MFMA families report the flop the matrix
    cores actually issue (direct-form flop / the Winograd saving; executed segments for the windowed pyramid products)
    against the fp32 matrix peak, next to the direct-form equivalent."""
    rows = []
    pmc = family_traffic()
    for fam, (direct, issued, calls) in sorted(work.items()):
        if fam not in traced:
            continue
        us, n = traced[fam]
        tot_us = us * n
        if issued == 0.0:   # a stream: `direct` holds its algorithmic bytes
            ach = direct / (tot_us * 1e-6) / 1e9
            rows.append({"kernel": fam, "bound": "hbm", "calls_per_step": calls, "launches_per_step": n,
                         "bytes_per_step": direct, "device_us_per_step": round(tot_us, 1), "achieved": round(ach, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                         "note": FAMILY_NOTES.get(fam)})
            continue
        ach = issued / (tot_us * 1e-6) / 1e12
        rows.append({"kernel": fam, "bound": "mfma", "calls_per_step": calls, "launches_per_step": n,
                     "direct_gflop_per_step": round(direct * 1e-9, 2), "issued_gflop_per_step": round(issued * 1e-9, 2),
                     "device_us_per_step": round(tot_us, 1), "mean_launch_us": round(us, 2),
                     "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4),
                     "direct_equivalent_tflops": round(direct / (tot_us * 1e-6) / 1e12, 2), "note": FAMILY_NOTES.get(fam)})
        if fam in pmc:      # HBM / Infinity-Cache side traffic per launch from the committed PMC passes (not this run)
            rows[-1].update(traffic=pmc[fam], traffic_source=FAMILY_TRAFFIC_FILE + " (offline rocprofv3 --pmc passes over "
                            "one eager attack step; mean per launch over the family's launch shapes; committed constant)")
    return rows


FAMILY_TRAFFIC_FILE = "profiles/r05/raft_step_traffic.json"
FAMILY_TRAFFIC_KERNELS = {"conv3x3_winograd": "conv3x3_winograd_kernel", "sepconv5_winograd": "sc5_wino_kernel"}


def family_traffic():
    """family -> mean PMC traffic bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, tools/pmc_traffic.py) over the launch
    shapes of the family's kernel in tools/pmc_traffic_raft.sh's record; {} without the file."""
    try:
        with open(os.path.join(REPO, FAMILY_TRAFFIC_FILE)) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return {}
    out = {}
    for fam, kern in FAMILY_TRAFFIC_KERNELS.items():
        rows = [v for k, v in rec.items() if k.startswith(kern)]
        n = sum(v["launches"] for v in rows)
        if n:
            out[fam] = sum(v["traffic_bytes"] * v["launches"] for v in rows) / n
    return out


def calibration(dev):
    """What THIS box sustains, measured once per bench run (< 50 ms): a register-only fp32-MFMA loop (the matrix roof:
    no memory, no LDS, pseudo-random operands) and a float4 copy stream (the achievable HBM rate), next to the data-sheet
    peaks every `frac` in this line is taken against."""
    from pcfa_amd import _hip
    lib = _hip.load()
    stream = torch.cuda.current_stream().cuda_stream
    scratch = torch.zeros(64, device=dev)
    n = 1 << 27   # 512 MiB in, 512 MiB out: beyond the 256 MB Infinity Cache
    src = torch.empty(n, device=dev).normal_()
    dst = torch.empty_like(src)

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    flop = [0]

    def mfma():
        flop[0] = lib.pcfa_calib_mfma_f32(scratch.data_ptr(), 2048, 1000, stream)
    t_m = timed(mfma, 3)
    t_c = timed(lambda: _hip.check(lib.pcfa_calib_copy(src.data_ptr(), dst.data_ptr(), n, stream), "pcfa_calib_copy"), 3)
    del src, dst
    tf = flop[0] / t_m / 1e12
    return {"mfma_f32_tflops": round(tf, 1), "mfma_f32_frac_of_peak": round(tf / MFMA_F32_PEAK_TFLOPS, 3),
            "mfma_implied_clock_ghz": round(tf * 1e12 / (256 * 256) / 1e9, 3),
            "copy_GBs": round(2 * 4 * n / t_c / 1e9, 1), "copy_frac_of_peak": round(2 * 4 * n / t_c / 1e9 / HBM_PEAK_GBS, 3),
            "how": "pcfa_calib_mfma_f32: 2048 workgroups x 4 waves x 1000 x 4 independent v_mfma_f32_32x32x2_f32 from registers "
                   "(8 waves per SIMD resident), %.2f ms per launch; pcfa_calib_copy: float4 grid-stride copy of 512 MiB, "
                   "%.2f ms per launch; event brackets around 3 launches after one warm-up" % (t_m * 1e3, t_c * 1e3),
            "peaks_used": {"mfma_f32_tflops": MFMA_F32_PEAK_TFLOPS, "hbm_GBs": HBM_PEAK_GBS}}


SCHEDULE_PARITY_FILE = "profiles/r05/schedule_parity_matrix.json"


def schedule_parity_record():
    """POINTER to the committed end-of-attack parity matrix (tools/parity_matrix.py): file name + pairs inside the
    tolerance per config.  Nothing of it is measured by this run (bench.py times synthetic pair `rank`)."""
    try:
        m = json.load(open(os.path.join(REPO, SCHEDULE_PARITY_FILE)))
    except (OSError, ValueError):
        return None
    rec = {"file": SCHEDULE_PARITY_FILE, "source": "committed file, not measured by this run"}
    for cfg in m.get("configs", []):
        rec["%s_%dsteps" % (cfg["net"].lower(), cfg["steps"])] = [cfg["pairs_ok"], cfg["pairs_total"]]
        if cfg.get("pairs_outside"):   # every outside pair carries an fp64_arbiter record (tools/parity_arbiter.py)
            rec.setdefault("outside_cleared_by_fp64_arbiter", {})["%s_%dsteps" % (cfg["net"].lower(), cfg["steps"])] = [
                len(cfg.get("pairs_outside_cleared_by_fp64_arbiter", [])), len(cfg["pairs_outside"])]
        if cfg["net"] == "RAFT" and cfg["steps"] == 20:
            rec["pairs_ok"], rec["pairs_total"] = cfg["pairs_ok"], cfg["pairs_total"]
    return rec


class ExtraLegGuard:
    """N > 1 only.  The headline figure is complete before the extra (universal) leg starts; that leg is the first place a
    collective sits on the data path, so a rank that fails in it alone would leave the others waiting inside RCCL for ever
    and the job would print nothing.  The guard gives the leg (and the shutdown after it) a deadline: when it passes, rank 0
    prints the line it already has, with the reason in place of the leg's record, and every rank leaves with status 0."""

    def __init__(self, seconds, publish, leave=None):
        import threading
        self._lock = threading.Lock()
        self._published = False
        self._publish = publish
        self._leave = leave if leave is not None else (lambda: os._exit(0))
        self.seconds = seconds
        self._timer = threading.Timer(seconds, self._expired)
        self._timer.daemon = True
        self._timer.start()

    def publish(self, record):
        """Print the line once (first caller wins); True if this call printed it."""
        with self._lock:
            if self._published:
                return False
            self._published = True
        self._publish(record)
        return True

    def _expired(self):
        self.publish({"error": "no result %.0f s after the headline measurement: the leg (or the shutdown after it) did not "
                               "return on every rank; the headline figure above is unaffected" % self.seconds})
        self._leave()

    def cancel(self):
        self._timer.cancel()


# --------------------------------------------------------------------------------------------------------------
# CPU baseline (+ the CPU side of the parity record)
# --------------------------------------------------------------------------------------------------------------
LINE_BUDGET = 6000   # bytes: the driver parses the ONE stdout line; r04's 22 KB line came back `parsed: null`
DETAIL_FILE = "bench_detail.json"


def _sig(x, sig=6):
    """Floats to `sig` significant digits, recursively (the line is for reading; the detail file keeps full precision)."""
    if isinstance(x, float):
        return float("%.*g" % (sig, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _sig(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, sig) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if k in d and d[k] is not None} if isinstance(d, dict) else d


def compact_line(out):
    """The ONE JSON line the driver parses: the contract's keys + `roofline` + `cpu_baseline` + the short form of each
    leg.  Everything else of `out` (kernel rows, families, optimiser, notes, set-up times) lives in DETAIL_FILE.
    Optional keys are dropped in DROP_ORDER until the line fits LINE_BUDGET."""
    roof_keys = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "mean_launch_us",
                 "launches_timed", "bytes_per_launch", "flop_per_launch", "hbm_frac")
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "per_rank_ms_per_step",
                       "higher_is_better", "scaling", "dtype", "data", "rehearsal", "n_ranks"))
    line["vs_baseline"] = out.get("vs_baseline")
    line["config"] = _pick(out.get("config", {}), ("workload", "weights", "closure_evals_per_step", "parallelism",
                                                   "closure_launch"))
    for k in ("closure_evals_per_sec", "final"):
        if k in out:
            line[k] = out[k]
    if "roofline" in out:
        r = dict(out["roofline"])
        if isinstance(r.get("hbm_view"), dict):
            r["hbm_frac"] = r["hbm_view"].get("frac_of_8TBs")
        line["roofline"] = _pick(r, roof_keys)
        line["roofline"].setdefault("traffic", None)
    if "roofline_unfused_lookup" in out:
        line["roofline_unfused_lookup"] = _pick(out["roofline_unfused_lookup"], roof_keys)
    if "cpu_baseline" in out:
        line["cpu_baseline"] = _pick(out["cpu_baseline"], ("value", "unit", "cores", "kind", "cpu", "sample"))
    par = ("loss_rel", "grad_rel_l2", "flow_aee", "flow_max_abs")
    if "parity_vs_cpu_port" in out:
        line["parity_vs_cpu_port"] = _pick(out["parity_vs_cpu_port"], par)
    if "calibration" in out:
        line["calibration"] = _pick(out["calibration"], ("mfma_f32_tflops", "copy_GBs", "error"))
    if "schedule_parity" in out:
        line["schedule_parity"] = _pick(out["schedule_parity"], ("file", "source", "pairs_ok", "pairs_total"))
    for leg in ("pwcnet", "gma"):
        if leg in out:
            src = out[leg]
            rec = _pick(src, ("value", "unit", "ms_per_step", "steps", "warmup", "final", "error"))
            if isinstance(src.get("parity_vs_cpu_port"), dict):
                rec["parity"] = _pick(src["parity_vs_cpu_port"], par)
            rows = [r for r in src.get("kernels", []) if " level " not in r.get("kernel", "")][:2]
            if rows:
                rec["kernels"] = [_pick(r, ("kernel", "bound", "achieved", "unit", "frac", "traffic", "mean_launch_us"))
                                  for r in rows]
            line[leg] = rec
    if "pairs_in_flight" in out:
        line["pairs_in_flight"] = _pick(out["pairs_in_flight"], ("pairs", "value", "unit", "ratio_to_one_pair",
                                                                "bit_identical_to_solo", "error"))
    if "unfused_lookup_batch2" in out and "B2" in out["unfused_lookup_batch2"]:
        line["unfused_lookup_batch2"] = {k: _pick(v, ("mean_launch_us", "frac")) for k, v in out["unfused_lookup_batch2"].items()
                                         if k in ("B1", "B2")}
    if "shared_forward_schedule" in out:
        line["shared_forward_schedule"] = _pick(out["shared_forward_schedule"], ("value", "ms_per_step", "error"))
    if "universal" in out:
        line["universal"] = _pick(out["universal"], ("metric", "value", "unit", "ms_per_step", "steps", "global_batch",
                                                     "pairs_per_gpu", "allreduces_per_closure", "allreduce_bytes",
                                                     "closure_launch", "error"))
    line["detail"] = DETAIL_FILE
    # nested records to 6 significant digits; the contract's own top-level numbers keep full precision (the driver
    # checks value against steps / ms_per_step)
    line = {k: (v if isinstance(v, float) and v == v else _sig(v)) for k, v in line.items()}
    DROP_ORDER = ("unfused_lookup_batch2", "shared_forward_schedule", "calibration", "pairs_in_flight", "roofline_unfused_lookup", "gma",
                  "pwcnet", "schedule_parity", "universal", "parity_vs_cpu_port")
    for k in DROP_ORDER:
        if len(json.dumps(line)) < LINE_BUDGET:
            break
        line.pop(k, None)
    return line


def emit(out, json_out):
    """Write the full record next to bench.py (PCFA_BENCH_DETAIL overrides the path) and print the compact line."""
    path = os.environ.get("PCFA_BENCH_DETAIL", os.path.join(REPO, DETAIL_FILE))
    try:
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
    except OSError as e:
        print("could not write %s: %r" % (path, e), file=sys.stderr)
    text = json.dumps(compact_line(out))
    assert len(text) < LINE_BUDGET and "\n" not in text, "bench line too long: %d bytes" % len(text)
    print(text, file=json_out, flush=True)


# --------------------------------------------------------------------------------------------------------------
# CPU baseline (+ the CPU side of the parity record)
# --------------------------------------------------------------------------------------------------------------
def cpu_baseline(net, h, w, nclosures, threads=0, boxconstraint="change_of_variables", joint=False, target="zero"):
    """Time the CPU port (pcfa_amd host code + oracle operators) on a bounded sample of the workload.  Its warm-up
    closure is evaluated at the parity point, so the same leg yields the CPU side of `parity_vs_cpu_port`."""
    from oracle import ops as oracle_ops
    from pcfa_amd import ops
    cores = threads if threads > 0 else min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    with ops.override_for_testing(oracle_ops):
        st = AttackStepper(net, h, w, torch.device("cpu"), seed=0, boxconstraint=boxconstraint, joint=joint, target=target)
        t0 = time.perf_counter()
        with torch.no_grad():
            st.predict()
        t_fwd = time.perf_counter() - t0  # forward only (includes first-touch warm-up)
        parity = parity_closure(st)       # warm-up closure (excluded from the timing) at the parity point
        times = []
        for _ in range(max(nclosures - 1, 1)):
            t0 = time.perf_counter()
            st.closure()
            times.append(time.perf_counter() - t0)
    t_c = sum(times) / len(times)
    step_s = 10 * t_c + t_fwd
    model_name = "?"
    try:
        model_name = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return {"value": 1.0 / step_s, "unit": "attack_steps/s", "cores": cores, "kind": "port", "cpu": model_name,
            "host_cores_available": os.cpu_count(), "closure_s": t_c, "forward_s": t_fwd,
            "sample": "1 forward + 1 warm-up closure + %d timed closure evals of %s %dx%d on %d host threads, "
                      "extrapolated to the 10 closures + 1 forward of one step" % (len(times), net, h, w, cores)
            }, parity


# --------------------------------------------------------------------------------------------------------------
def _sync():
    if torch.cuda.is_initialized():
        torch.cuda.synchronize()


def timed_steps(st, steps, sharding):
    """EXACTLY `steps` steps between barrier + device synchronisation on both sides; returns this rank's own time up
    to its own synchronisation (`own`: what `per_rank_ms_per_step` reports) and the time including the closing
    barrier."""
    _sync()
    sharding.barrier()
    c0 = st.closures
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        last = st.step()
    _sync()
    own = time.perf_counter() - t0
    sharding.barrier()
    timed_steps.own_s = own
    return time.perf_counter() - t0, st.closures - c0, last


def shared_forward_leg(net, h, w, dev, warmup, steps, sharding, model):
    """Informational, NOT the headline: the same steps with the closure captured as forward + backward graphs
    (pcfa_amd.graphed.SplitGraphedClosure, PCFA_SHARED_FORWARD=1), so that the re-prediction that ends a step doubles as
    the forward of the next step's first closure evaluation -- the reference evaluates that forward twice at the same
    point.  Same kernels on the same data, results unchanged; `value` above keeps the reference's schedule."""
    try:
        st = AttackStepper(net, h, w, dev, seed=0, model=model)
        st.enable_graph(share_forward=True)
        if st.graphed is None:
            return {"error": "capture failed"}
        for _ in range(warmup):
            st.step()
        elapsed, closures, _ = timed_steps(st, steps, sharding)
        return {"value": steps / elapsed, "unit": "attack_steps/s", "ms_per_step": 1e3 * elapsed / steps,
                "closure_evals_per_step": closures / steps,
                "forwards_shared_per_step": st.graphed.forwards_shared / max(1, st.steps_done - 1),
                "note": "re-prediction forward reused by the next step's first closure (backward-only replay); "
                        "off by default, not part of `value`"}
    except Exception as e:  # noqa: BLE001 -- an extra: the headline line must survive it
        return {"error": repr(e)}


def universal_leg(net, h, w, dev, rank, world, pairs_per_gpu, warmup, steps, sharding, cdev, model=None,
                  use_graph=None):
    """K steps of UniversalAttack on one global batch of world*pairs_per_gpu pairs (rank r holds pairs
    r*b .. r*b+b-1); returns the timing record (rank 0 fills the JSON from it)."""
    seeds = [100 + rank * pairs_per_gpu + i for i in range(pairs_per_gpu)]
    t0 = time.perf_counter()
    ua = UniversalStepper(net, h, w, dev, seeds, use_graph=use_graph, model=model)
    setup = time.perf_counter() - t0
    for _ in range(warmup):
        ua.step()
    col0 = ua.reducer.collectives if ua.reducer is not None else 0
    elapsed, closures, last = timed_steps(ua, steps, sharding)
    elapsed = sharding.max_scalar(elapsed, cdev)
    cols = (ua.reducer.collectives if ua.reducer is not None else 0) - col0
    flat = ua.reducer.flat.numel() * 4 if ua.reducer is not None else 0
    return {"metric": "universal_pair_steps_per_sec", "value": world * pairs_per_gpu * steps / elapsed,
            "unit": "pair-steps/s", "optimizer_steps_per_sec": steps / elapsed, "ms_per_step": 1e3 * elapsed / steps,
            "steps": steps, "warmup": warmup, "global_batch": world * pairs_per_gpu, "pairs_per_gpu": pairs_per_gpu,
            "closure_evals_per_step": closures / steps, "allreduces_per_closure": cols / max(closures, 1),
            "allreduce_bytes": flat, "closure_launch": "hipGraph replay" if ua.graphed else "eager",
            "setup_s": setup, "final": last}


def lookup_b2_row(dev, hf, wf, levels=4, radius=4, reps=30):
    """The un-fused lookup kernel on a batch of TWO pairs' queries in one launch (at B = 1 the 20.44 MB of a
    launch are a 7 us latency chain; does twice the work per launch reach the 0.40 target?).  Random feature maps, random
    sub-pixel coordinates inside the map, device timestamps of `reps` launches."""
    from torch.autograd import DeviceType
    from torch.profiler import ProfilerActivity, profile
    from pcfa_amd import hip_ops
    g = torch.Generator().manual_seed(5)
    out = {}
    for B in (1, 2):
        f1 = torch.randn(B, 256, hf, wf, generator=g).to(dev)
        f2 = torch.randn(B, 256, hf, wf, generator=g).to(dev)
        base = torch.stack(torch.meshgrid(torch.arange(wf), torch.arange(hf), indexing="xy"), 0).float()[None].repeat(B, 1, 1, 1)
        coords = (base + 3.0 * torch.randn(B, 2, hf, wf, generator=g)).to(dev)
        with torch.no_grad():
            blk = hip_ops.CorrBlock(f1, f2, num_levels=levels, radius=radius)
            for _ in range(3):
                blk(coords)
            torch.cuda.synchronize()
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                for _ in range(reps):
                    blk(coords)
                torch.cuda.synchronize()
        d = [e.time_range.elapsed_us() for e in prof.events()
             if e.device_type == DeviceType.CUDA and "corr_lookup_fwd_kernel" in e.name]
        us = sum(d) / len(d)
        nbytes = B * lookup_algorithmic_bytes(hf, wf, levels, radius)
        out["B%d" % B] = {"bytes_per_launch": nbytes, "mean_launch_us": us, "launches_timed": len(d),
                          "achieved": nbytes / (us * 1e-6) / 1e9, "frac": nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
        del blk, f1, f2
    out["note"] = ("corr_lookup_fwd_kernel<4> back to back on random maps / coordinates (L2- / MALL-warm: flatters the kernel "
                   "against the in-closure figure of roofline_unfused_lookup), one launch for B pairs' queries; peak 8 TB/s")
    return out


def pairs_in_flight_leg(net, h, w, dev, rank, model, one_pair_value, steps=4, warmup=1, pairs=2):
    """Informational, NOT the headline (one pair per GPU stays `value`): `pairs` independent pairs attacked side by side on
    this GPU -- one host thread + stream + graph set each (pcfa_amd.attack_PCFA.PairsInFlight) -- and then the last of them
    again ALONE, from scratch, for the bit-identity check.  value = pair-steps/s of the concurrent run."""
    from pcfa_amd import attack_PCFA
    seeds = [rank + 500 + k for k in range(pairs)]
    try:
        flight = attack_PCFA.PairsInFlight(lambda k: AttackStepper(net, h, w, dev, seeds[k], use_graph=True, model=model),
                                           pairs, dev)
        if any(st.graphed is None for st in flight.attacks):
            return {"error": "capture failed"}
        flight.run(warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = flight.run(steps)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        a = flight.attacks[-1]
        solo = AttackStepper(net, h, w, dev, seeds[-1], use_graph=True, model=model)   # lane 0, after lane 0's pair is done
        for _ in range(warmup + steps):
            solo_last = solo.step()
        same = bool(torch.equal(a.delta1, solo.delta1) and torch.equal(a.delta2, solo.delta2)
                    and torch.equal(a.flow_pred, solo.flow_pred) and tuple(last[-1]) == tuple(solo_last))
        value = pairs * steps / elapsed
        return {"pairs": pairs, "value": value, "unit": "pair-steps/s", "steps": steps, "warmup": warmup,
                "ms_per_step_per_pair": 1e3 * elapsed / steps, "ratio_to_one_pair": value / one_pair_value,
                "bit_identical_to_solo": same, "final_in_flight": [list(map(float, v)) for v in last],
                "note": "pairs %s side by side, one thread + stream + hipGraph set per pair; the last pair re-run alone "
                        "gives the same bits" % seeds}
    except Exception as e:  # noqa: BLE001 -- an extra: the headline line must survive it
        return {"error": repr(e)}


def pwcnet_leg(a, dev, sharding):
    """BASELINE config 4 beside the headline: PWC-Net on one synthetic KITTI-sized pair (375x1242 -> 384x1280),
    --joint_perturbation --boxconstraint=clipping, delta_bound 0.005, zero target: the config's 50 attack steps (the same
    PairAttack.step), the cost-volume / warp kernels' roofline rows from the graph replays, and one closure against
    the CPU port at the same point (PWCNet.py:45-58,166-206,227-330 through the HIP spatial-correlation sampler)."""
    h, w, steps, warmup = 375, 1242, 50, 2   # config 4 is a 50-step attack
    st = AttackStepper("PWCNet", h, w, dev, seed=0, boxconstraint="clipping", joint=True)
    gpu_parity = None if a.no_cpu_baseline else parity_closure(st)
    st.enable_graph()
    for _ in range(warmup):
        st.step()
    elapsed, closures, last = timed_steps(st, steps, sharding)
    rec = {"metric": "attack_steps_per_sec", "value": steps / elapsed, "unit": "attack_steps/s", "steps": steps,
           "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps, "closure_evals_per_step": closures / steps,
           "closure_launch": "hipGraph replay" if st.graphed is not None else "eager",
           "config": {"workload": "PWCNet, 1 synthetic %dx%d pair (padded %dx%d), joint perturbation, clipping, "
                                  "delta_bound=0.005, zero target, L-BFGS max_iter=10 (BASELINE config 4 shape)"
                                  % (h, w, st.image1.shape[-2], st.image1.shape[-1])},
           "final": {"aee_adv_tgt": last[0], "aee_adv_init": last[1], "l2_delta": last[2]}}
    if os.environ.get("PCFA_BENCH_NO_TRACER", "0") != "1" and st.graphed is not None:
        try:
            traced = graph_replay_kernel_times(st)
            pwc_seq = getattr(graph_replay_kernel_times, "sequences", {})
            pwc_kernel_table.null_launch_us = null_launch_device_us()
            graph_replay_kernel_times.sequences = pwc_seq
            rec["null_launch_device_us"] = pwc_kernel_table.null_launch_us
            rec["kernels"] = pwc_kernel_table(traced, st.image1.shape[-2], st.image1.shape[-1])
        except Exception as e:  # noqa: BLE001
            rec["kernels_error"] = repr(e)
    if gpu_parity is not None:
        cpu, cpu_parity = cpu_baseline("PWCNet", h, w, 3, a.cpu_threads, boxconstraint="clipping", joint=True)
        rec["cpu_baseline"] = cpu
        rec["parity_vs_cpu_port"] = parity_record(gpu_parity, cpu_parity)
    return rec


def gma_leg(a, dev, sharding):
    """BASELINE config 3's network beside the headline: GMA on one synthetic 436x1024 pair (one pair per GPU is config 3's
    sharding), change of variables, delta_bound 0.005, neg_flow target (mu = 7.5e5): a few attack steps of the same
    PairAttack.step and one closure against the CPU port at the same point (models/gma/network.py:72-129, gma.py:34-115)."""
    h, w, steps, warmup = 436, 1024, 5, 1
    st = AttackStepper("GMA", h, w, dev, seed=0, target="neg_flow")
    gpu_parity = None if a.no_cpu_baseline else parity_closure(st)
    st.enable_graph()
    for _ in range(warmup):
        st.step()
    elapsed, closures, last = timed_steps(st, steps, sharding)
    from pcfa_amd import config as pcfa_config
    rec = {"metric": "attack_steps_per_sec", "value": steps / elapsed, "unit": "attack_steps/s", "steps": steps,
           "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps, "closure_evals_per_step": closures / steps,
           "closure_launch": "hipGraph replay" if st.graphed is not None else "eager",
           "attention_products": pcfa_config.cfg(st.model).gma_gemm,
           "config": {"workload": "GMA, 1 synthetic %dx%d pair (padded %dx%d), disjoint delta, change_of_variables, "
                                  "delta_bound=0.005, neg_flow target, L-BFGS max_iter=10 (BASELINE config 3 per GPU)"
                                  % (h, w, st.image1.shape[-2], st.image1.shape[-1])},
           "final": {"aee_adv_tgt": last[0], "aee_adv_init": last[1], "l2_delta": last[2]}}
    if gpu_parity is not None:
        cpu, cpu_parity = cpu_baseline("GMA", h, w, 3, a.cpu_threads, target="neg_flow")
        rec["cpu_baseline"] = cpu
        rec["parity_vs_cpu_port"] = parity_record(gpu_parity, cpu_parity)
    return rec


def rehearse_cpu(a, json_out):
    """--rehearse-cpu: the launcher, the rendezvous, the max-over-ranks timing, the JSON assembly and the universal
    leg's all-reduce with NO GPU -- every rank drives the CPU port (as the cpu_baseline leg does) on a 64x64 SpyNet
    pair over gloo.  Not a measurement of anything."""
    from oracle import ops as oracle_ops
    from pcfa_amd import ops, sharding
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        torch.distributed.init_process_group(backend="gloo")
    rank = sharding.rank()
    dev = torch.device("cpu")
    torch.set_num_threads(2)
    os.environ["PCFA_USE_CPU"] = "1"
    with ops.override_for_testing(oracle_ops):
        st = AttackStepper("SpyNet", 64, 64, dev, seed=rank)
        for _ in range(a.warmup):
            st.step()
        elapsed, closures, last = timed_steps(st, a.steps, sharding)
        per_rank = sharding.all_scalars(timed_steps.own_s, dev)
        elapsed = sharding.max_scalar(elapsed, dev)
        universal = None
        if world > 1 and not a.no_universal_leg:
            universal = universal_leg("SpyNet", 64, 64, dev, rank, world, 1, 0, 1, sharding, dev, use_graph=False)
    if rank == 0:
        out = {"metric": "cpu_port_rehearsal_steps_per_sec", "value": world * a.steps / elapsed, "unit": "steps/s",
               "rehearsal": "cpu-port", "n_gpus": world, "n_ranks": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": 1e3 * elapsed / a.steps, "per_rank_ms_per_step": [1e3 * t / a.steps for t in per_rank],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "REHEARSAL (no GPU): SpyNet 64x64 on the CPU port, one pair per rank, gloo",
                          "closure_evals_per_step": closures / a.steps},
               "final": {"aee_adv_tgt": last[0], "aee_adv_init": last[1], "l2_delta": last[2]}}
        if universal is not None:
            out["universal"] = universal
        emit(out, json_out)
    sharding.shutdown()


def main():
    a = parse()
    from pcfa_amd import launch
    if a.gpus > 1 and not launch.inside_rank():
        # plain `python bench.py --gpus N`: become the launcher.  Nothing in this process has touched the GPU (no
        # torch.cuda call, no kernel library load) and nothing will: it starts N fresh rank processes and waits.
        sys.exit(launch.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], a.gpus))
    from pcfa_amd import sharding
    # stdout carries the ONE JSON line: everything else (the mirrors' chatter, native libraries writing to fd 1) goes
    # to stderr until the line is printed
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    json_out = os.fdopen(saved_fd, "w")
    if a.rehearse_cpu:
        return rehearse_cpu(a, json_out)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    # one process per GPU; PCFA_BENCH_BACKEND=gloo + PCFA_BENCH_SHARE_GPU=1 only exist to exercise the
    # multi-process path on a single-GPU box (RCCL refuses two ranks on one device)
    backend = os.environ.get("PCFA_BENCH_BACKEND", "nccl")
    share = os.environ.get("PCFA_BENCH_SHARE_GPU", "0") == "1"
    dev = torch.device("cuda", 0 if (world == 1 or share) else sharding.local_rank())
    torch.cuda.set_device(dev)
    if world > 1:
        if backend == "nccl":
            sharding.init_from_env("nccl")
        else:
            torch.distributed.init_process_group(backend=backend)
    rank = sharding.rank()
    if world != a.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (a.gpus, world), file=sys.stderr)
    cdev = dev if backend == "nccl" else torch.device("cpu")  # where the scalar collectives live
    torch.backends.cudnn.benchmark = a.miopen_find
    h, w = (int(v) for v in a.size.lower().split("x"))
    use_graph = not a.no_graph

    if a.universal:
        rec = universal_leg(a.net, h, w, dev, rank, world, a.pairs_per_gpu, a.warmup, a.steps, sharding, cdev,
                            use_graph=use_graph)
        if rank == 0:
            out = {"metric": rec["metric"], "value": rec["value"], "unit": rec["unit"], "n_gpus": world,
                   "steps": a.steps, "warmup": a.warmup, "ms_per_step": rec["ms_per_step"],
                   "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                   "data": "synthetic",
                   "config": {"workload": "%s universal perturbation (BASELINE config 5), %d synthetic %dx%d pair(s) "
                                          "per GPU, clipping, delta_bound=0.005, zero target, L-BFGS max_iter=10, "
                                          "one all-reduce of [grad delta1|grad delta2|loss] per closure"
                                          % (a.net, a.pairs_per_gpu, h, w), "weights": "random:1234",
                              "parallelism": "data parallel over the batch, %d rank(s)" % world},
                   "universal": rec}
            emit(out, json_out)
        sharding.shutdown()
        return

    t0 = time.perf_counter()
    st = AttackStepper(a.net, h, w, dev, seed=rank)
    if a.channels_last:
        st.model = st.model.to(memory_format=torch.channels_last)
    setup_eager = time.perf_counter() - t0
    from pcfa_amd import hip_ops
    corr_net = a.net in ("RAFT", "GMA")
    gpu_parity = parity_closure(st) if (rank == 0 and world == 1 and not a.no_cpu_baseline) else None
    # HIP events attached to every dispatch of the named kernel (on the stream it is launched on)
    prof = hip_ops.DispatchTimer() if corr_net else None
    t0 = time.perf_counter()
    if use_graph:
        st.enable_graph()   # what pcfa_attack does per pair: 2 warm-up closures + capture, 1 warm-up + capture forward
        torch.cuda.synchronize()
    graph_active = st.graphed is not None   # False after a failed capture: PairAttack logs it and launches eagerly
    setup_graph = time.perf_counter() - t0
    for _ in range(a.warmup):
        st.step()
    if not use_graph:
        hip_ops.set_dispatch_timer(prof)
    elapsed, closures, last = timed_steps(st, a.steps, sharding)
    hip_ops.set_dispatch_timer(None)
    per_rank = sharding.all_scalars(timed_steps.own_s, cdev)
    elapsed = sharding.max_scalar(elapsed, cdev)
    traced = None
    # PCFA_BENCH_NO_TRACER=1: skip the in-process tracer (set when an external profiler already owns it)
    if use_graph and rank == 0 and os.environ.get("PCFA_BENCH_NO_TRACER", "0") != "1" and \
            a.net in ("RAFT", "GMA", "PWCNet"):
        try:
            traced = graph_replay_kernel_times(st)
        except Exception as e:  # the tracer is an extra: fall back to the eager hipEvent figures
            print("graph-replay kernel trace unavailable: %r" % (e,), file=sys.stderr)
    lbfgs_history = st.optimizer.history_count() if hasattr(st.optimizer, "history_count") else None
    work = None
    if use_graph and corr_net:
        # dispatch-attached events cannot ride inside a captured graph: time the kernels on the same data in one
        # extra, eagerly launched step right after the timed region -- with the lookup -> convc1 fusion switched OFF,
        # so that this step also yields the un-fused lookup kernel's own roofline row
        st.graphed = st.repredict = None
        import dataclasses
        from pcfa_amd import config as pcfa_config
        fused_model = st.model   # same seeded weights, lookup -> convc1 fusion off: a second model (switches are frozen)
        st.model = load_model(a.net, dev, True, dataclasses.replace(pcfa_config.cfg(fused_model), fused_lookup=False))
        # two eagerly launched steps: the first packs the second model's weights and records the work per kernel family
        # (family_rows; host-side bookkeeping in every operator call), the second carries the dispatch-attached events --
        # without the bookkeeping, whose host gaps let the GPU idle between launches and lengthen a 7 us latency chain
        # like the lookup by ~1 us
        work = {}
        hip_ops.set_work_recorder(work)
        st.step()
        hip_ops.set_work_recorder(None)
        torch.cuda.synchronize()
        hip_ops.set_dispatch_timer(prof)
        st.step()
        torch.cuda.synchronize()
        hip_ops.set_dispatch_timer(None)
        st.model = fused_model

    # a second pair of the same shape: PairAttack adopts the first pair's static buffers, graphs and optimiser
    second_pair = None
    if use_graph and world == 1:
        try:
            times = []
            for k in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                st2 = AttackStepper(a.net, h, w, dev, seed=rank + 1000 + k, use_graph=True, model=st.model)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
                reused = bool(st2.graphs_reused)
                del st2
            second_pair = {"setup_s": min(times), "setup_s_each": times, "graphs_reused": reused,
                           "note": "synthetic pair generated on the host + upload + preprocessing + unattacked forward "
                                   "(graph replay) + target + metrics of the NEXT pairs of this shape: no warm-up, no "
                                   "re-capture (three pairs; the first follows the eager measurement step above)"}
        except Exception as e:  # noqa: BLE001 -- informational
            second_pair = {"error": repr(e)}

    out = None
    if rank == 0:
        hp, wp = st.image1.shape[-2:]
        out = {
            "metric": "attack_steps_per_sec", "value": world * a.steps / elapsed, "unit": "attack_steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "per_rank_ms_per_step": [1e3 * t / a.steps for t in per_rank],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s, 1 synthetic %dx%d pair per GPU (padded %dx%d), disjoint delta, "
                                   "change_of_variables, delta_bound=0.005, zero target, L-BFGS max_iter=10"
                                   % (a.net, h, w, hp, wp), "weights": "random:1234",
                       "closure_evals_per_step": closures / a.steps, "parallelism": "pairs sharded 1/GPU",
                       "closure_launch": "hipGraph replay" if graph_active else
                                         ("eager (hipGraph capture FAILED: see the log)" if use_graph else "eager"),
                       "timed_object": "pcfa_amd.attack_PCFA.PairAttack.step (the body of pcfa_attack's loop)"},
            "closure_evals_per_sec": world * closures / elapsed,
            "per_pair_setup_s": {"upload_init_forward_target": setup_eager, "graph_warmup_and_capture": setup_graph,
                                 "second_pair_same_shape": second_pair,
                                 "note": "first pair of a shape: paid once, outside the timed steps; later pairs of "
                                         "the same shape reuse buffers + graphs (second_pair_same_shape)"},
            "final": {"aee_adv_tgt": last[0], "aee_adv_init": last[1], "l2_delta": last[2]},
        }
        lb = lbfgs_record(traced, st, lbfgs_history)
        if lb is not None:
            out["lbfgs"] = lb
        if corr_net:
            timings = prof.summary()
            eager_us, eager_n = timings["corr_lookup_fwd"]
            hf, wf = hp // 8, wp // 8
            nbytes = lookup_algorithmic_bytes(hf, wf)
            how_graph = ("dispatch timestamps of every launch inside the hipGraph replays of one attack step right "
                         "after the timed region (HIP activity tracer via torch.profiler: the timestamps rocprofv3 "
                         "reads; hipEvents cannot be attached inside a captured graph)")
            how_eager = ("hipEvents on the dispatch packet (hipExtLaunchKernel), every launch of one eagerly launched "
                         "step with the lookup -> convc1 fusion switched off")
            out["kernels"] = kernel_table(traced if traced else timings, hf, wf, hp, wp)
            if traced and work:
                out["kernel_families"] = family_rows(traced, work)
                # the windowed pyramid products are priced on the flop they execute (kernel_families); a row against the
                # dense product they skip most of would report a `frac` above 1
                out["kernels"] = [r for r in out["kernels"] if r["kernel"] not in ("corr_pyramid_gemm_dfmap1",
                                                                                  "corr_pyramid_gemm_df2ext")]
                out["kernel_rows_cover"] = getattr(graph_replay_kernel_times, "coverage", None)
            unfused = {"kernel": "corr_lookup_fwd_kernel<4> (un-fused lookup, models/raft/corr.py:29-50)", "bound": "hbm",
                       "bytes_per_launch": nbytes, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "traffic": lookup_traffic("corr_lookup_fwd"), "traffic_source": TRAFFIC_SOURCE,
                       "event_bracket_overhead_us": event_overhead_us()}
            if traced and "corr_lookup_fwd" in traced:     # the closure runs the un-fused kernel (fusion off / n.a.)
                us, n = traced["corr_lookup_fwd"]
                unfused.update(mean_launch_us=us, launches_timed=n, timing=how_graph, eager_step_hip_event_us=eager_us)
            else:
                us, n = eager_us, eager_n
                unfused.update(mean_launch_us=us, launches_timed=n, timing=how_eager)
            unfused["achieved"] = nbytes / (us * 1e-6) / 1e9
            unfused["frac"] = unfused["achieved"] / HBM_PEAK_GBS
            if traced and "corr_lookup_convc1_fwd" in traced:
                # the kernel that runs in the timed region: lookup + convc1 + ReLU in one launch
                us, n = traced["corr_lookup_convc1_fwd"]
                q = hf * wf
                flop = 2.0 * 256 * 324 * q
                fbytes = q * 4 * 100 * 4 + q * 8 + q * 256 * 4 + 256 * 324 * 4
                ach = flop / (us * 1e-6) / 1e12
                out["roofline"] = {
                    "kernel": "corr_lookup_convc1_fwd_kernel (correlation lookup fused with convc1 + bias + ReLU)",
                    "bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / MFMA_F32_PEAK_TFLOPS, "traffic": lookup_traffic("corr_lookup_convc1_fwd"), "traffic_source": TRAFFIC_SOURCE,
                    "flop_per_launch": flop, "mean_launch_us": us, "launches_timed": n, "timing": how_graph,
                    "why_mfma": "2*256*324*Q flop over texels + coords + [256][Q] output + weights = %.0f flop/B, "
                                "above the fp32-matrix ridge (157.3 TFLOP/s / 8 TB/s = 20 flop/B)" % (flop / fbytes),
                    "hbm_view": {"bytes_per_launch": fbytes, "achieved_GBs": fbytes / (us * 1e-6) / 1e9,
                                 "frac_of_8TBs": fbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                 "note": "algorithmic bytes of the fused kernel: 4 levels x 100 texels x 4 B per "
                                         "query + coords + 256-channel output + weights (the 9.1 MB lookup tensor "
                                         "is neither written nor re-read)"},
                    "replaces_per_iteration_us": {"corr_lookup_fwd": eager_us,
                                                  "note": "+ the library GEMM of convc1 (~28 us) + the bias/ReLU "
                                                          "pass (~6 us), profiles/r02_closure_kernel_mix*.txt"}}
                out["roofline_unfused_lookup"] = unfused
            else:
                out["roofline"] = unfused
            out["kernels_eager_step_hip_events"] = kernel_table(timings, hf, wf, hp, wp)
        elif a.net == "PWCNet" and traced:
            rows = pwc_kernel_table(traced, hp, wp)
            out["kernels"] = rows
            if rows:
                r0 = rows[0]
                out["roofline"] = {"kernel": "scorr_fwd (five cost volumes of one closure)", "bound": "hbm",
                                   "achieved": r0["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": r0["frac"], "traffic": None, "bytes_per_launch": r0["per_launch"],
                                   "mean_launch_us": r0["mean_launch_us"], "launches_timed": r0["launches_timed"],
                                   "timing": "dispatch timestamps inside the hipGraph replays (HIP activity tracer)"}
        if world == 1:
            try:
                out["calibration"] = calibration(dev)
            except Exception as e:  # noqa: BLE001 -- informational
                out["calibration"] = {"error": repr(e)}
        sp = schedule_parity_record()
        if sp is not None:
            out["schedule_parity"] = sp
        if world == 1 and corr_net:
            try:
                out["unfused_lookup_batch2"] = lookup_b2_row(dev, hp // 8, wp // 8)
            except Exception as e:  # noqa: BLE001 -- informational
                out["unfused_lookup_batch2"] = {"error": repr(e)}
        if world == 1 and use_graph and a.net == "RAFT" and not a.no_pairs_in_flight_leg:
            out["pairs_in_flight"] = pairs_in_flight_leg(a.net, h, w, dev, rank, st.model, out["value"])
        if world == 1 and use_graph and not a.no_shared_forward_leg:
            out["shared_forward_schedule"] = shared_forward_leg(a.net, h, w, dev, a.warmup, a.steps, sharding, st.model)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"], cpu_parity = cpu_baseline(a.net, h, w, a.cpu_closures, a.cpu_threads)
            if gpu_parity is not None:
                out["parity_vs_cpu_port"] = parity_record(gpu_parity, cpu_parity)
        if world == 1 and a.net == "RAFT" and not a.no_pwcnet_leg:
            try:
                out["pwcnet"] = pwcnet_leg(a, dev, sharding)
            except Exception as e:  # noqa: BLE001 -- the headline line must survive a failure of the extra leg
                out["pwcnet"] = {"error": repr(e)}
        if world == 1 and a.net == "RAFT" and not a.no_gma_leg:
            try:
                out["gma"] = gma_leg(a, dev, sharding)
            except Exception as e:  # noqa: BLE001
                out["gma"] = {"error": repr(e)}
    if world > 1 and not a.no_universal_leg:
        # last, and under a deadline: see ExtraLegGuard
        def publish(record):
            if rank == 0:
                out["universal"] = record
                emit(out, json_out)

        guard = ExtraLegGuard(float(os.environ.get("PCFA_BENCH_EXTRA_LEG_DEADLINE_S", "240")), publish)
        try:
            universal = universal_leg(a.net, h, w, dev, rank, world, 1, 1, 2, sharding, cdev, model=st.model,
                                      use_graph=use_graph)
        except Exception as e:  # noqa: BLE001 -- the headline line must survive a failure of the extra leg
            universal = {"error": repr(e)}
        guard.publish(universal)
        # leave together: a rank that left alone (its leg raised) would make the launcher end the others before the
        # deadline lets rank 0 print.  The store is the rendezvous' own TCP store, not a collective.
        try:
            from datetime import timedelta
            store = torch.distributed.distributed_c10d._get_default_store()
            store.set("pcfa_bench_leg_%d" % rank, "error" if "error" in universal else "ok")
            store.wait(["pcfa_bench_leg_%d" % r for r in range(world)], timedelta(seconds=guard.seconds + 60.0))
            sharding.shutdown()
        except Exception as e:  # noqa: BLE001 -- the line is out; the guard ends this rank at the deadline
            print("rank %d: leaving through the deadline: %r" % (rank, e), file=sys.stderr)
            time.sleep(guard.seconds + 60.0)
        guard.cancel()
        return out
    if rank == 0:
        emit(out, json_out)
    sharding.shutdown()
    return out


if __name__ == "__main__":
    main()
