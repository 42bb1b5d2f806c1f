#!/usr/bin/env python3
"""fp64 ARBITER for the end-of-attack parity matrix (VERDICT r04 "next" item 2): a number, not an argument.

For one synthetic pair of a BASELINE config the CPU port (pcfa_amd host code + oracle operators, fp32, `--threads` host
threads) runs the attack up to the step where the r04 matrix saw the GPU leg leave it, and records its iterates at
    step 0, closure evaluations 0..3   (closure 2 or 3 is where the closure losses of the legs first separate, for every
                                         network and pair: the first curvature pair enters there)
    the first closure evaluation of every step in --at-steps   (where the per-step metrics separate).
At each of these points the SAME variables are evaluated by
    gpu[:variant]   the product path on the GPU (child process per variant; `f23` = PCFA_CONV3X3_ALGO=f23, Winograd
                    F(4x4,3x3) off everywhere; `r04policy` = round 4's F(4x4,3x3) policy, which RAFT / GMA no longer take),
    port<T>         the fp32 port on --threads threads (the recorded run itself),
    port<F>         the fp32 port on --floor-threads threads,
    port_fp64       the port with every tensor, weight and operator in float64  -- the arbiter.
Per point: loss of every leg, relative L2 of every leg's gradient against port_fp64, and the rule
    |g_gpu - g_64| <= max(1e-2 |g_64|, 3 |g_port - g_64|).
A point that passes is recorded as `gpu_as_close_to_fp64_as_the_rule_allows`; one that fails is a kernel bug to bisect.

First curvature pair, per leg, from that leg's OWN arithmetic (what its optimiser sees: torch.optim.LBFGS,
lr = 1, first step t = min(1, 1/|g0|_1), second iteration tests y.s > 1e-10 before it stores the pair and sets
H = y.s / y.y):  |g0|, |g0|_1, t, |g1|, |y|, |s|, y.s (the leg's fp32 dot product and the same in fp64), y.y, H, and the
margin y.s / 1e-10 of the gate.  A pair whose legs sit on different sides of the gate is a threshold bifurcation;
otherwise the amplification |g0| / |y| says how much of the legs' gradient difference reaches the second direction.

    python tools/parity_arbiter.py run --net RAFT --seed 6 --at-steps 0 --out DIR [--threads 16] [--floor-threads 8]
                                       [--gpu-variants default,f23]
    python tools/parity_arbiter.py assemble --dir DIR --out profiles/r05/fp64_arbiter.json
The `run` process never touches the GPU itself (its GPU legs are child processes started before any HIP call).
ORACLE USE: this is a parity checker (tests/test_gpu_parity.py runs it as a child); the oracle is the thing compared against.
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

GRAD_TOL = 1e-2
GATE = 1e-10
VARIANT_ENV = {"default": {}, "f23": {"PCFA_CONV3X3_ALGO": "f23"}, "r04policy": {"PCFA_CONV3X3_ALGO": "r04"}}


def _flat(ts):
    import torch
    return torch.cat([t.detach().flatten() for t in ts])


def _set_point(st, x):
    import torch
    with torch.no_grad():
        for p, v in zip(st.params, x):
            p.copy_(v.to(device=p.device, dtype=p.dtype))


def _eval(st, x):
    """loss, flat gradient of st's closure at variables x (eager, no optimiser involved)."""
    _set_point(st, x)
    st.optimizer.zero_grad()
    loss = float(st.closure_body())
    return loss, _flat([p.grad for p in st.params]).clone()


def first_pair(evaluate, x0, g0, dtype):
    """The first curvature pair as torch.optim.LBFGS (lr = 1) forms it from this leg's own gradient at x0."""
    import torch
    g0 = g0.to(dtype)
    l1 = float(g0.abs().sum())
    t = min(1.0, 1.0 / l1)
    d = g0.neg()
    off, x1 = 0, []
    for v in x0:
        n = v.numel()
        x1.append(v.to(dtype).add(d[off:off + n].view_as(v).to(v.device), alpha=t))
        off += n
    _, g1 = evaluate(x1)
    g1 = g1.to(dtype)
    y = g1 - g0
    s = d * t
    ys_own = float(y.dot(s))                       # the dot product in the leg's own precision (what its gate sees)
    ys64 = float(y.double().dot(s.double()))
    yy = float(y.double().dot(y.double()))
    return {"g0_norm": float(g0.double().norm()), "g0_l1": l1, "t": t, "g1_norm": float(g1.double().norm()),
            "y_norm": yy ** 0.5, "s_norm": float(s.double().norm()), "ys": ys_own, "ys_in_fp64": ys64, "yy": yy,
            "H": ys64 / yy if yy > 0 else None, "gate": GATE, "pair_stored": ys_own > GATE, "gate_margin": ys_own / GATE,
            "amplification_g0_over_y": float(g0.double().norm()) / yy ** 0.5 if yy > 0 else None}, x1, g1


def _stepper(net, device, seed):
    import parity_matrix
    return parity_matrix._stepper(net, device, seed, False)


# ------------------------------------------------------------------------------------------------------------------
def cmd_gpu_eval(a):
    """Child: the product path at the points of --points; gradients + this leg's own first curvature pair."""
    import torch
    pts = torch.load(a.points)
    dev = torch.device("cuda", 0)
    st = _stepper(pts["net"], dev, pts["seed"])
    out = {"losses": [], "grads": []}
    for x in pts["x"]:
        loss, g = _eval(st, x)
        out["losses"].append(loss)
        out["grads"].append(g.cpu())
    x0 = [v.to(dev) for v in pts["x"][0]]
    fp, _, _ = first_pair(lambda x: _eval(st, x), x0, out["grads"][0].to(dev), torch.float32)
    out["first_pair"] = fp
    from pcfa_amd import _hip
    out["conv3x3_algo_256_192_55x128"] = int(_hip.load().pcfa_conv3x3_algo(1, 192, 256, 55, 128))
    torch.save(out, a.out)


def cmd_run(a):
    import torch
    import parity_matrix
    from oracle import ops as oracle_ops
    from pcfa_amd import ops
    from trajectory_closure_parity import to_double
    at_steps = sorted({int(v) for v in a.at_steps.split(",") if v != ""})
    os.makedirs(a.out, exist_ok=True)
    t00 = time.perf_counter()
    torch.set_num_threads(a.threads)
    cpu = torch.device("cpu")
    names, xs, port_loss, port_grad = [], [], [], []

    def log(msg):
        print("[arbiter %s pair %d %.0fs] %s" % (a.net, a.seed, time.perf_counter() - t00, msg), file=sys.stderr, flush=True)

    with ops.override_for_testing(oracle_ops):
        cst = _stepper(a.net, cpu, a.seed)
        inner = cst.closure
        state = {"step": 0, "k": 0}

        def recording():
            want = (state["step"] == 0 and state["k"] < 4) or (state["k"] == 0 and state["step"] in at_steps)
            x = [p.detach().clone() for p in cst.params] if want else None
            loss = inner()
            if want:
                names.append("step%d_closure%d" % (state["step"], state["k"]))
                xs.append(x)
                port_loss.append(float(loss))
                port_grad.append(_flat([p.grad for p in cst.params]).clone())
            state["k"] += 1
            return loss
        cst.closure = recording
        last = max(at_steps + [0])
        for k in range(last + 1):
            state["step"], state["k"] = k, 0
            if k == last and k > 0:
                recording()       # only the step's first closure is wanted: evaluate it at the current iterate and stop
                break
            cst.step()
            log("port%d step %d done" % (a.threads, k))
        cst.closure = inner
        # the same port on the floor thread count
        floor_loss, floor_grad = [], []
        if a.floor_threads:
            torch.set_num_threads(a.floor_threads)
            for x in xs:
                l, g = _eval(cst, x)
                floor_loss.append(l)
                floor_grad.append(g)
            torch.set_num_threads(a.threads)
            log("port%d leg done" % a.floor_threads)
        x_first = [v.clone() for v in xs[0]]
        fp_port, _, _ = first_pair(lambda x: _eval(cst, x), x_first, port_grad[0], torch.float32)
        del cst

    # GPU legs: children, started while this process still has not touched the GPU
    tmp = tempfile.mkdtemp(prefix="arbiter_")
    pts_file = os.path.join(tmp, "points.pt")
    torch.save({"net": a.net, "seed": a.seed, "x": xs}, pts_file)
    gpu = {}
    for var in [v for v in a.gpu_variants.split(",") if v]:
        outf = os.path.join(tmp, "gpu_%s.pt" % var)
        env = dict(os.environ, **VARIANT_ENV[var])
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "gpu-eval", "--points", pts_file, "--out", outf],
                           env=env, capture_output=True, text=True, timeout=3000)
        if r.returncode != 0:
            raise RuntimeError("gpu-eval[%s] failed:\n%s" % (var, r.stderr[-3000:]))
        gpu[var] = torch.load(outf)
        log("gpu[%s] leg done" % var)

    # the arbiter: the port in float64
    with ops.override_for_testing(oracle_ops):
        dst = _stepper(a.net, cpu, a.seed)
        to_double(dst)
        for p in dst.params:
            if p.dtype != torch.float64:
                p.data = p.data.double()
        l64, g64 = [], []
        for x in xs:
            l, g = _eval(dst, x)
            assert g.dtype == torch.float64
            l64.append(l)
            g64.append(g)
            log("fp64 point %d/%d" % (len(g64), len(xs)))
        fp_64, _, _ = first_pair(lambda x: _eval(dst, x), x_first, g64[0], torch.float64)
        del dst

    def rel(g, ref):
        return float((g.double() - ref).norm() / ref.norm())
    points = []
    for i, name in enumerate(names):
        ref = g64[i]
        row = {"point": name, "loss": {"port_fp64": l64[i], "port%d" % a.threads: port_loss[i]},
               "g64_norm": float(ref.norm()),
               "grad_rel_l2_vs_fp64": {"port%d" % a.threads: rel(port_grad[i], ref)},
               "grad_rel_l2_gpu_vs_port%d" % a.threads: {}}
        if a.floor_threads:
            row["loss"]["port%d" % a.floor_threads] = floor_loss[i]
            row["grad_rel_l2_vs_fp64"]["port%d" % a.floor_threads] = rel(floor_grad[i], ref)
            row["grad_rel_l2_port%d_vs_port%d" % (a.threads, a.floor_threads)] = float(
                (port_grad[i] - floor_grad[i]).norm() / floor_grad[i].norm())
        for var, rec in gpu.items():
            key = "gpu" if var == "default" else "gpu:" + var
            row["loss"][key] = rec["losses"][i]
            row["grad_rel_l2_vs_fp64"][key] = rel(rec["grads"][i], ref)
            row["grad_rel_l2_gpu_vs_port%d" % a.threads][key] = float(
                (rec["grads"][i] - port_grad[i]).norm() / port_grad[i].norm())
        e_port = row["grad_rel_l2_vs_fp64"]["port%d" % a.threads]
        if "default" in gpu:
            e_gpu = row["grad_rel_l2_vs_fp64"]["gpu"]
            row["tolerance"] = max(GRAD_TOL, 3 * e_port)
            row["rule_ok"] = e_gpu <= row["tolerance"]
            row["gpu_error_over_port_error"] = e_gpu / e_port if e_port > 0 else None
        points.append(row)
    pair = {"port%d" % a.threads: fp_port, "port_fp64": fp_64}
    for var, rec in gpu.items():
        pair["gpu" if var == "default" else "gpu:" + var] = rec["first_pair"]
    sides = {k: v["pair_stored"] for k, v in pair.items()}
    rec = {"net": a.net, "seed": a.seed, "config": parity_matrix.CONFIGS[a.net], "threads": a.threads,
           "floor_threads": a.floor_threads, "at_steps": at_steps, "host": parity_matrix._cpu_name(),
           "rule": "|g_gpu - g_64| <= max(%g |g_64|, 3 |g_port%d - g_64|), relative L2 over all variables" % (GRAD_TOL, a.threads),
           "points": points, "rule_ok_everywhere": all(p.get("rule_ok", True) for p in points),
           "first_curvature_pair": pair, "legs_on_one_side_of_the_gate": len(set(sides.values())) == 1,
           "gpu_variants": {v: {"conv3x3_algo_256_192_55x128": gpu[v]["conv3x3_algo_256_192_55x128"]} for v in gpu},
           "seconds": time.perf_counter() - t00}
    path = os.path.join(a.out, "%s_pair%d_arbiter.json" % (a.net.lower(), a.seed))
    with open(path, "w") as f:
        f.write(json.dumps(rec, indent=1) + "\n")
    for p in points:
        log("%s: vs fp64 %s  rule_ok %s" % (p["point"], {k: "%.2e" % v for k, v in p["grad_rel_l2_vs_fp64"].items()},
                                            p.get("rule_ok")))
    log("first pair y.s: %s" % {k: "%.3e" % v["ys"] for k, v in pair.items()})
    print(path)
    return 0 if rec["rule_ok_everywhere"] else 1


def _verdict(r):
    """One sentence per pair from the numbers (nothing here is argued: every clause quotes a measured quantity)."""
    fp = r["first_curvature_pair"]
    port = next(k for k in fp if k.startswith("port") and k != "port_fp64")
    e64 = fp["port_fp64"]["ys"]
    dev = {k: abs(v["ys"] - e64) / abs(e64) for k, v in fp.items() if k != "port_fp64"}
    amp = fp["port_fp64"]["amplification_g0_over_y"] or 0.0
    if not r["rule_ok_everywhere"]:
        return "GPU gradient farther from fp64 than the rule allows at %s: kernel bug to bisect" % [
            p["point"] for p in r["points"] if p.get("rule_ok") is False]
    if not r["legs_on_one_side_of_the_gate"]:
        return ("threshold bifurcation: the legs sit on different sides of torch LBFGS's y.s > 1e-10 gate (%s); every "
                "gradient is as close to fp64 as the rule allows" % {k: "%.2e" % v["ys"] for k, v in fp.items()})
    worst = max(r["points"], key=lambda p: p["grad_rel_l2_vs_fp64"].get("gpu", 0.0))
    head = ("no kernel fault and no gate flip: at every point the GPU gradient is as close to fp64 as the rule allows (worst "
            "point %s: gpu %.1e, %s %.1e) and all legs sit on one side of the y.s > 1e-10 gate; "
            % (worst["point"], worst["grad_rel_l2_vs_fp64"]["gpu"], port, worst["grad_rel_l2_vs_fp64"][port]))
    if max(dev.values()) > 0.05:
        return head + ("the first curvature pair is ill-conditioned -- y = g1 - g0 is %.0fx smaller than g0, so y.s (which sets "
                       "H = y.s / y.y, the length of the second move) deviates from its fp64 value by %s on the fp32 legs, the "
                       "port's included: the second iterate of ANY two fp32 evaluations differs by tens of percent"
                       % (amp, {k: "%.0f%%" % (100 * v) for k, v in dev.items()}))
    return head + ("the first curvature pair is well-conditioned (|g0| / |y| = %.0f, y.s within %.1f%% of fp64 on every leg): "
                   "the legs separate later, by the optimiser's amplification of gradient differences that are below the "
                   "fp32 port's own distance from fp64" % (amp, 100 * max(dev.values())))


def cmd_assemble(a):
    recs = [json.load(open(p)) for p in sorted(glob.glob(os.path.join(a.dir, "*_arbiter.json")))]
    out = {"what": __doc__.split("\n\n")[0], "rule": recs[0]["rule"] if recs else None, "pairs": []}
    for r in recs:
        worst = max(r["points"], key=lambda p: p["grad_rel_l2_vs_fp64"].get("gpu", 0.0))
        fp = r["first_curvature_pair"]
        out["pairs"].append({
            "net": r["net"], "pair": r["seed"], "at_steps": r["at_steps"], "rule_ok_everywhere": r["rule_ok_everywhere"],
            "legs_on_one_side_of_the_gate": r["legs_on_one_side_of_the_gate"], "verdict": _verdict(r),
            "worst_point": {k: worst[k] for k in ("point", "grad_rel_l2_vs_fp64", "tolerance", "rule_ok", "loss") if k in worst},
            "points": [{"point": p["point"], "vs_fp64": p["grad_rel_l2_vs_fp64"], "rule_ok": p.get("rule_ok"),
                        "gpu_error_over_port_error": p.get("gpu_error_over_port_error")} for p in r["points"]],
            "first_curvature_pair": {k: {kk: v[kk] for kk in ("ys", "ys_in_fp64", "y_norm", "g0_norm", "g1_norm", "t", "H",
                                                              "pair_stored", "gate_margin", "amplification_g0_over_y")}
                                     for k, v in fp.items()},
            "host": r["host"], "threads": r["threads"], "floor_threads": r["floor_threads"]})
    out["pairs_total"] = len(recs)
    out["pairs_rule_ok"] = sum(r["rule_ok_everywhere"] for r in recs)
    # F(4x4,3x3) against F(2x2,3x3) against the port, all measured against fp64 (RAFT records that carry the variants)
    tab = []
    for r in recs:
        for p in r["points"]:
            v = p["grad_rel_l2_vs_fp64"]
            if "gpu:f23" in v:
                tab.append({"net": r["net"], "pair": r["seed"], "point": p["point"], **v})
    if tab:
        keys = [k for k in tab[0] if k not in ("net", "pair", "point")]
        out["winograd_variants_vs_fp64"] = {
            "rows": tab, "geometric_mean": {k: float(__import__("math").exp(sum(__import__("math").log(t[k]) for t in tab if k in t)
                                                                            / max(1, sum(k in t for t in tab)))) for k in keys},
            "legs": "gpu = the policy of the build that produced the record (r04 policy in the records taken before the r05 "
                    "policy change: their `gpu_variants` field says which algorithm 256->192 at 55x128 took); gpu:f23 = "
                    "Winograd F(4x4,3x3) off everywhere; gpu:r04policy = round 4's policy"}
    with open(a.out, "w") as f:
        f.write(json.dumps(out, indent=1) + "\n")
    for p in out["pairs"]:
        print("%s pair %d: rule_ok %s -- %s" % (p["net"], p["pair"], p["rule_ok_everywhere"], p["verdict"]))
    if tab:
        print("vs fp64, geometric mean over %d points: %s" % (len(tab), {k: "%.2e" % v for k, v in
                                                                         out["winograd_variants_vs_fp64"]["geometric_mean"].items()}))


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("run")
    p.add_argument("--net", default="RAFT")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--at-steps", default="0", help="steps whose first closure evaluation is a point (beside step 0, closures 0..3)")
    p.add_argument("--threads", type=int, default=16)
    p.add_argument("--floor-threads", type=int, default=8)
    p.add_argument("--gpu-variants", default="default")
    p.add_argument("--out", required=True)
    p = sub.add_parser("gpu-eval")
    p.add_argument("--points", required=True)
    p.add_argument("--out", required=True)
    p = sub.add_parser("assemble")
    p.add_argument("--dir", required=True)
    p.add_argument("--out", required=True)
    a = ap.parse_args()
    sys.exit({"run": cmd_run, "gpu-eval": cmd_gpu_eval, "assemble": cmd_assemble}[a.cmd](a) or 0)


if __name__ == "__main__":
    main()
