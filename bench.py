#!/usr/bin/env python3
"""PCFA attack-steps/sec on MI355X (BASELINE.json metric; workload = configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one `--steps` iteration of pcfa_attack on one synthetic 436x1024 image pair with RAFT
(seeded random weights -- no checkpoints exist offline): torch.optim.LBFGS(max_iter=10).step =
10 closure evaluations (box transform -> RAFT forward -> AEE + L2 penalty -> backward) + 1 re-prediction
forward + the step's metrics.  Inputs are resident in HBM before the timed region.
Every rank attacks its own pair (disjoint perturbations, no data-path collective): weak scaling,
value = N*K / max-over-ranks time.

The JSON line also carries
  roofline      the correlation-lookup forward kernel (the kernel BASELINE's north_star names):
                algorithmic bytes/launch (20.44 MB at 55x128) / mean launch duration measured
                with HIP events around every launch inside the timed steps, against 8 TB/s HBM;
  cpu_baseline  this repo's CPU port (pcfa_amd host code + oracle operators, torch fp32 on all host cores)
                timed on a bounded sample of the same workload, extrapolated to the 10+1 schedule.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--net", default="RAFT", choices=["RAFT", "GMA", "PWCNet", "SpyNet", "FlowNet2"])
    ap.add_argument("--size", default="436x1024")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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


class AttackStepper:
    """The body of pcfa_attack's step loop (attack_PCFA.py:155-247) on pre-resident tensors."""

    def __init__(self, net, h, w, device, seed, boxconstraint="change_of_variables"):
        from pcfa_amd import attack_PCFA
        from pcfa_amd.helper_functions import datasets, losses, ownutilities, targets, logging as plog
        self.A, self.losses, self.own, self.plog = attack_PCFA, losses, ownutilities, plog
        self.net, self.device, self.box = net, device, boxconstraint
        self.eps = attack_PCFA.EPS_BOX
        cov = boxconstraint == "change_of_variables"
        unit = ownutilities.model_takes_unit_input(net)
        kw = {"eps_box": self.eps} if cov else {}
        self.model = ownutilities.import_and_load(net, make_unit_input=not unit, variable_change=cov,
                                                  make_scaled_input_model=True, device=device,
                                                  weights="random:1234", **kw)
        self.model.eval()
        for p in self.model.parameters():
            p.requires_grad = False
        i1, i2, _ = datasets.synthetic_pair(seed, h, w)
        i1, i2 = i1[None].to(device), i2[None].to(device)
        if not unit:
            i1, i2 = i1 / 255., i2 / 255.
        self.padder, [self.image1, self.image2] = ownutilities.preprocess_img(net, i1, i2)
        if cov:
            self.nw1 = torch.atanh(2. * (1. - self.eps) * self.image1 - (1 - self.eps))
            self.nw2 = torch.atanh(2. * (1. - self.eps) * self.image2 - (1 - self.eps))
        else:
            self.nw1, self.nw2 = self.image1.clone(), self.image2.clone()
        self.nw1.requires_grad = True
        self.nw2.requires_grad = True
        from pcfa_amd import ops
        self.optimizer = ops.get().LBFGS([self.nw1, self.nw2], max_iter=10)
        self.delta_bound = 0.005
        self.mu = 2500. / self.delta_bound
        with torch.no_grad():
            self.flow_init = self.predict().clone()
        self.target = targets.get_target("zero", self.flow_init, device=device)
        self.closures = 0
        self.graphed = self.repredict = None

    def enable_graph(self):
        """Capture forward+loss+backward (and the re-prediction forward) once; later evaluations replay the
        hipGraphs -- exactly what pcfa_attack does on the GPU."""
        from pcfa_amd.graphed import GraphedClosure, GraphedForward
        self.graphed = GraphedClosure(self._closure_body, [self.nw1, self.nw2])
        self.repredict = GraphedForward(self._repredict_body, self.device)

    def _repredict_body(self):
        d1, d2 = self.A.extract_deltas(self.nw1, self.nw2, self.image1, self.image2, self.box, eps_box=self.eps)
        return d1, d2, self.predict()

    def predict(self):
        out = self.own.compute_flow(self.model, "scaled_input_model", self.nw1, self.nw2, test_mode=True)
        [out] = self.own.postprocess_flow(self.net, self.padder, out)
        return out

    def _closure_body(self):
        flow = self.predict()
        d1, d2 = self.A.extract_deltas(self.nw1, self.nw2, self.image1, self.image2, self.box, eps_box=self.eps)
        loss = self.losses.loss_delta_constraint(flow, self.target, d1, d2, self.device,
                                                 delta_bound=self.delta_bound, mu=self.mu, f_type="aee")
        loss.backward()
        return loss

    def closure(self):
        self.closures += 1
        if self.graphed is not None:
            return self.graphed()
        self.optimizer.zero_grad()
        return self._closure_body()

    def step(self):
        self.optimizer.step(self.closure)
        if self.repredict is not None:
            d1, d2, flow = self.repredict()
        else:
            with torch.no_grad():
                d1, d2, flow = self._repredict_body()
        aee_tgt, aee_init = self.plog.calc_metrics_adv(flow, self.target, self.flow_init)
        l2 = self.plog.calc_delta_metrics(d1, d2)
        return aee_tgt, aee_init, l2[2]


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


def lookup_traffic():
    """HBM bytes per launch of the lookup kernel from rocprofv3 PMC counters (collected offline by
    tools/pmc_traffic.sh with the guide's gfx950 corrections, committed under profiles/); None if absent."""
    path = os.path.join(REPO, "profiles", "lookup_traffic.json")
    try:
        return json.load(open(path))["corr_lookup_fwd"]["traffic_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def lookup_algorithmic_bytes(hf, wf, levels=4, radius=4):
    """unique texels (Q * levels * (2r+2)^2 * 4 B) + coords + output."""
    q = hf * wf
    n1 = 2 * radius + 1
    return q * levels * (2 * radius + 2) ** 2 * 4 + q * 2 * 4 + q * levels * n1 * n1 * 4


MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak


def kernel_table(timings, hf, wf, hp, wp, dim=256, levels=4, radius=4):
    """Per-kernel roofline rows for the RAFT/GMA path: algorithmic work per launch (SURVEY 8d figures, stated
    in DESIGN.md section 4) / mean dispatch duration measured with hipEvents on the dispatch packets."""
    q = hf * wf
    n1 = 2 * radius + 1
    out_b = q * levels * n1 * n1 * 4
    win_b = q * levels * (2 * radius + 2) ** 2 * 4
    gemm = 2.0 * q * q * dim  # level 0 only: the pooled levels cost no multiplies in the reference (avg_pool2d)
    img = 3 * hp * wp * 4
    work = {  # label -> (bound, algorithmic bytes or flop per launch)
        "corr_lookup_fwd": ("hbm", win_b + q * 8 + out_b),
        "corr_lookup_bwd": ("hbm", out_b + q * 8 + 2 * win_b),
        "corr_pyramid_gemm_fwd": ("mfma", gemm),
        "corr_pyramid_gemm_dfmap1": ("mfma", gemm),
        "corr_pyramid_gemm_df2ext": ("mfma", gemm),
        "box_transform_fwd": ("hbm", 4 * img),
        "box_transform_bwd": ("hbm", 6 * img),
        "gru_gates_fwd": ("hbm", 8 * q * 128 * 4),
        "gru_update_fwd": ("hbm", 6 * q * 128 * 4),
    }
    rows = []
    for label, (us, n) in sorted(timings.items()):
        if label not in work:
            continue
        bound, w = work[label]
        if bound == "hbm":
            ach, peak, unit = w / (us * 1e-6) / 1e9, HBM_PEAK_GBS, "GB/s"
        else:
            ach, peak, unit = w / (us * 1e-6) / 1e12, MFMA_F32_PEAK_TFLOPS, "TFLOP/s"
        rows.append({"kernel": label, "bound": bound, "per_launch": w, "mean_launch_us": round(us, 2),
                     "launches_timed": n, "achieved": round(ach, 2), "peak": peak, "unit": unit,
                     "frac": round(ach / peak, 4)})
    return rows


TRACED = {  # kernel-name fragment -> label of kernel_table()
    "corr_lookup_fwd_kernel": "corr_lookup_fwd", "corr_lookup_bwd_kernel": "corr_lookup_bwd",
    "gemm_f32_mfma_kernel<true, true,": "corr_pyramid_gemm_fwd",
    "gemm_f32_mfma_kernel<false, false,": "corr_pyramid_gemm_dfmap1",
    "gemm_f32_mfma_kernel<false, true,": "corr_pyramid_gemm_df2ext",
    "box_fwd_kernel": "box_transform_fwd", "box_bwd_kernel": "box_transform_bwd",
    "gru_gates_fwd_kernel": "gru_gates_fwd", "gru_update_fwd_kernel": "gru_update_fwd",
}


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
    acc = {}
    for ev in prof.events():
        if ev.device_type != DeviceType.CUDA:
            continue
        for frag, label in TRACED.items():
            if frag in ev.name:
                a = acc.setdefault(label, [0.0, 0])
                a[0] += ev.time_range.elapsed_us()
                a[1] += 1
                break
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items() if v[1]}


def cpu_baseline(net, h, w, nclosures, threads=0):
    """Time the CPU port (pcfa_amd host code + oracle operators) on a bounded sample of the workload."""
    from oracle import ops as oracle_ops
    from pcfa_amd import ops
    cores = threads if threads > 0 else min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    with ops.override_for_testing(oracle_ops):
        st = AttackStepper(net, h, w, torch.device("cpu"), seed=0)
        t0 = time.perf_counter()
        with torch.no_grad():
            st.predict()
        t_fwd = time.perf_counter() - t0  # forward only (includes first-touch warm-up)
        times = []
        for _ in range(nclosures):
            t0 = time.perf_counter()
            st.closure()
            times.append(time.perf_counter() - t0)
    t_c = sum(times[1:]) / max(len(times) - 1, 1) if len(times) > 1 else times[0]  # first one warms up
    step_s = 10 * t_c + t_fwd
    model_name = "?"
    try:
        model_name = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return {"value": 1.0 / step_s, "unit": "attack_steps/s", "cores": cores, "kind": "port", "cpu": model_name,
            "host_cores_available": os.cpu_count(), "closure_s": t_c, "forward_s": t_fwd,
            "sample": "%d closure evals + 1 forward of %s %dx%d on %d host threads, extrapolated to the "
                      "10 closures + 1 forward of one step" % (nclosures, net, h, w, cores)}


def main():
    a = parse()
    from pcfa_amd import sharding
    # stdout carries the ONE JSON line: everything else (the mirrors' chatter, native libraries writing to fd 1) goes
    # to stderr until the line is printed
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    json_out = os.fdopen(saved_fd, "w")
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
    cdev = dev if backend == "nccl" else torch.device("cpu")  # where the two scalar collectives live
    torch.backends.cudnn.benchmark = a.miopen_find
    h, w = (int(v) for v in a.size.lower().split("x"))

    st = AttackStepper(a.net, h, w, dev, seed=rank)
    if a.channels_last:
        st.model = st.model.to(memory_format=torch.channels_last)
    from pcfa_amd import hip_ops
    corr_net = a.net in ("RAFT", "GMA")
    # HIP events attached to every dispatch of the named kernel (on the stream it is launched on)
    prof = hip_ops.DispatchTimer() if corr_net else None
    for _ in range(a.warmup):
        st.step()
    use_graph = not a.no_graph
    if use_graph:
        st.enable_graph()
        st.step()  # one untimed step on the graph
    torch.cuda.synchronize()
    sharding.barrier()
    if not use_graph:
        hip_ops.set_dispatch_timer(prof)
    c0 = st.closures
    t0 = time.perf_counter()
    last = None
    for _ in range(a.steps):
        last = st.step()
    torch.cuda.synchronize()
    sharding.barrier()
    elapsed = time.perf_counter() - t0
    hip_ops.set_dispatch_timer(None)
    elapsed = sharding.max_scalar(elapsed, cdev)
    closures = st.closures - c0
    traced = None
    # PCFA_BENCH_NO_TRACER=1: skip the in-process tracer (set when an external profiler already owns it)
    if use_graph and corr_net and rank == 0 and os.environ.get("PCFA_BENCH_NO_TRACER", "0") != "1":
        try:
            traced = graph_replay_kernel_times(st)
        except Exception as e:  # the tracer is an extra: fall back to the eager hipEvent figures
            print("graph-replay kernel trace unavailable: %r" % (e,), file=sys.stderr)
    if use_graph and corr_net:
        # dispatch-attached events cannot ride inside a captured graph: time the same kernel on the same
        # data in one extra, eagerly launched step right after the timed region
        st.graphed = st.repredict = None
        hip_ops.set_dispatch_timer(prof)
        st.step()
        torch.cuda.synchronize()
        hip_ops.set_dispatch_timer(None)

    out = None
    if rank == 0:
        hp, wp = st.image1.shape[-2:]
        out = {
            "metric": "attack_steps_per_sec", "value": world * a.steps / elapsed, "unit": "attack_steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s, 1 synthetic %dx%d pair per GPU (padded %dx%d), disjoint delta, "
                                   "change_of_variables, delta_bound=0.005, zero target, L-BFGS max_iter=10"
                                   % (a.net, h, w, hp, wp), "weights": "random:1234",
                       "closure_evals_per_step": closures / a.steps, "parallelism": "pairs sharded 1/GPU",
                       "closure_launch": "hipGraph replay" if use_graph else "eager"},
            "closure_evals_per_sec": world * closures / elapsed,
            "final": {"aee_adv_tgt": last[0], "aee_adv_init": last[1], "l2_delta": last[2]},
        }
        if corr_net:
            timings = prof.summary()
            eager_us, eager_n = timings["corr_lookup_fwd"]
            nbytes = lookup_algorithmic_bytes(hp // 8, wp // 8)
            if traced and "corr_lookup_fwd" in traced:
                us, n = traced["corr_lookup_fwd"]
                how = ("dispatch timestamps of every launch inside the hipGraph replays of one attack step right "
                       "after the timed region (HIP activity tracer via torch.profiler: the timestamps rocprofv3 "
                       "reads; hipEvents cannot be attached inside a captured graph)")
                out["kernels"] = kernel_table(traced, hp // 8, wp // 8, hp, wp)
            else:
                us, n = eager_us, eager_n
                how = "hipEvents on the dispatch packet (hipExtLaunchKernel), every launch of one eagerly launched step"
                out["kernels"] = kernel_table(timings, hp // 8, wp // 8, hp, wp)
            ach = nbytes / (us * 1e-6) / 1e9
            out["roofline"] = {"kernel": "corr_lookup_fwd_kernel<4>", "bound": "hbm", "achieved": ach,
                               "timing": how, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                               "traffic": lookup_traffic(), "bytes_per_launch": nbytes, "mean_launch_us": us,
                               "launches_timed": n,
                               "eager_step_hip_event_us": eager_us, "eager_step_launches": eager_n,
                               "eager_step_frac": nbytes / (eager_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                               "event_bracket_overhead_us": event_overhead_us()}
            out["kernels_eager_step_hip_events"] = kernel_table(timings, hp // 8, wp // 8, hp, wp)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.net, h, w, a.cpu_closures, a.cpu_threads)
        print(json.dumps(out), file=json_out, flush=True)
    sharding.shutdown()
    return out


if __name__ == "__main__":
    main()
