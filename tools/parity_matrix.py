#!/usr/bin/env python3
"""End-of-attack parity as a MEASURED DISTRIBUTION (VERDICT r03 "next" item 1b), not a single draw.

For BASELINE config 2 (RAFT 436x1024, change of variables, zero target, 20 steps) and config 4 (PWC-Net 375x1242,
joint perturbation, clipping, 20 / 50 steps) and synthetic pairs 0..7:  the GPU product path, the CPU port on 16 host
threads and the CPU port on 8 host threads run the SAME schedule (attack_PCFA.py:155-247, best-iterate rule :226-243).
Per run: the loss of every closure evaluation, the per-step metrics, the best-iterate results.  Per pair the matrix
then holds
    delta           = GPU - port@16 for AEE(adv, target) / AEE(adv, init) at the best iterate and ||delta|| there,
    port_spread     = |port@16 - port@8|  (the reference's own noise floor),
    inside          = |delta| <= max(floor, 3 x port_spread)   (floor 1e-3 AEE, 1e-5 for ||delta||),
    first_step_divergence = the first step whose AEE(adv, target) differs by more than max(0.1, 2 %) between GPU and port@16 -- where
                      the two trajectories leave each other (a shifted overshoot cycle shows as > 1; legs on one branch stay
                      within 0.04 even on overshoot steps).  Also recorded: the first closure evaluation whose loss differs by
                      more than 1e-3 relative (early, at the first overshoot point, for every pair of legs),
    closure_delta   = loss / gradient of the GPU closure evaluated AT THE PORT'S ITERATE at the start of that step
                      against the port's own loss / gradient there (the port run saves its iterates): shows whether the
                      split is a closure error or the optimiser amplifying rounding noise.

    python tools/parity_matrix.py port --net RAFT --seed 0 --threads 16 --steps 20 --out DIR [--snapshots TMPDIR]
    python tools/parity_matrix.py gpu  --net RAFT --seeds 0,1,2 --steps 20 --out DIR [--snapshots TMPDIR]
    python tools/parity_matrix.py assemble --dir DIR --out profiles/r04_schedule_parity_matrix.json
Records are one JSON file per (net, steps, seed, leg) in DIR; legs: gpu, port16, port8 (any --threads T -> portT).
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

CONFIGS = {   # net -> (size, box, joint, target)
    "RAFT": ("436x1024", "change_of_variables", False, "zero"),
    "PWCNet": ("375x1242", "clipping", True, "zero"),
    "GMA": ("436x1024", "change_of_variables", False, "neg_flow"),
}
FLOORS = {"aee_adv_tgt_min": 1e-3, "aee_adv_init_at_min": 1e-3, "l2_delta_min": 1e-5}
DIVERGENCE_REL = 1e-3


def _stepper(net, device, seed, use_graph, model=None):
    import bench
    size, box, joint, target = CONFIGS[net]
    size = os.environ.get("PCFA_MATRIX_SIZE", size)   # tool self-test at a small size only
    h, w = (int(v) for v in size.split("x"))
    return bench.AttackStepper(net, h, w, device, seed=seed, boxconstraint=box, joint=joint, target=target,
                               use_graph=use_graph, model=model)


def _run(st, steps, on_step=None, progress=None):
    """`steps` attack steps; returns per-step closure losses, per-step metrics, best-iterate results."""
    import torch
    losses = []
    inner = st.closure

    def closure():
        out = inner()
        losses[-1].append(float(out))
        return out
    st.closure = closure
    per_step = []
    t0 = time.perf_counter()
    for k in range(steps):
        if on_step is not None:
            on_step(k)
        losses.append([])
        per_step.append([float(v) for v in st.step()])
        if progress:
            print("%s step %d/%d %s (%.0f s)" % (progress, k + 1, steps, per_step[-1], time.perf_counter() - t0),
                  file=sys.stderr, flush=True)
    res = st.result()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return {"closure_losses": losses, "per_step": per_step, "aee_adv_tgt_min": res[9], "aee_adv_init_at_min": res[10],
            "l2_delta_min": res[11], "aee_adv_tgt_final": res[4], "aee_adv_init_final": res[5], "l2_delta_final": res[8],
            "seconds": time.perf_counter() - t0}


def _name(out, net, steps, seed, leg):
    return os.path.join(out, "%s_%dsteps_pair%d_%s.json" % (net.lower(), steps, seed, leg))


def cmd_port(a):
    import torch
    from oracle import ops as oracle_ops
    from pcfa_amd import ops
    torch.set_num_threads(a.threads)
    os.makedirs(a.out, exist_ok=True)
    for seed in a.seeds:
        path = _name(a.out, a.net, a.steps, seed, "port%d" % a.threads)
        if os.path.exists(path) and not a.force:
            continue
        snap = None
        if a.snapshots:
            snap = os.path.join(a.snapshots, "%s_%dsteps_pair%d" % (a.net.lower(), a.steps, seed))
            os.makedirs(snap, exist_ok=True)
        with ops.override_for_testing(oracle_ops):
            st = _stepper(a.net, torch.device("cpu"), seed, False)
            if snap:
                inner = st.closure
                state = {"first": False, "k": -1}

                def recording():
                    out = inner()
                    if state["first"]:   # the first closure of a step evaluates the step's starting point
                        state["first"] = False
                        torch.save({"x": [p.detach().clone() for p in st.params], "loss": float(out),
                                    "grad": [p.grad.detach().clone() for p in st.params]},
                                   os.path.join(snap, "step%02d.pt" % state["k"]))
                    return out
                st.closure = recording

                def on_step(k):
                    state["first"], state["k"] = True, k
            else:
                on_step = None
            rec = _run(st, a.steps, on_step, progress="port[%d thr] %s pair %d" % (a.threads, a.net, seed))
        rec.update(net=a.net, seed=seed, steps=a.steps, leg="port%d" % a.threads, threads=a.threads,
                   host=_cpu_name(), config=CONFIGS[a.net])
        with open(path, "w") as f:
            f.write(json.dumps(rec) + "\n")


def _cpu_name():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "?"


def cmd_gpu(a):
    import torch
    import bench
    dev = torch.device("cuda", 0)
    os.makedirs(a.out, exist_ok=True)
    size, box, joint, target = CONFIGS[a.net]
    model = bench.load_model(a.net, dev, box == "change_of_variables")
    for seed in a.seeds:
        st = _stepper(a.net, dev, seed, True, model=model)
        rec = _run(st, a.steps)
        rec["graphed"] = st.graphed is not None
        del st
        rec.update(net=a.net, seed=seed, steps=a.steps, leg="gpu", config=CONFIGS[a.net])
        snap = a.snapshots and os.path.join(a.snapshots, "%s_%dsteps_pair%d" % (a.net.lower(), a.steps, seed))
        port_path = _name(a.out, a.net, a.steps, seed, "port16")
        if snap and os.path.isdir(snap) and os.path.exists(port_path):
            port = json.load(open(port_path))
            div = first_step_divergence(rec["per_step"], port["per_step"])
            rec["first_step_divergence_vs_port16"] = div
            # closure-level delta at the port's iterate where the trajectories split (or at the last step if they never do)
            k = div["step"] if div else a.steps - 1
            pt = torch.load(os.path.join(snap, "step%02d.pt" % k))
            est = _stepper(a.net, dev, seed, False, model=model)
            with torch.no_grad():
                for p, v in zip(est.params, pt["x"]):
                    p.copy_(v.to(dev))
            est.optimizer.zero_grad()
            loss_g = float(est.closure_body())
            gg = torch.cat([p.grad.detach().flatten().cpu() for p in est.params])
            gc = torch.cat([g.flatten() for g in pt["grad"]])
            rec["closure_delta_at_port_iterate"] = {
                "step": k, "loss_port": pt["loss"], "loss_gpu": loss_g,
                "loss_rel": abs(loss_g - pt["loss"]) / abs(pt["loss"]),
                "grad_rel_l2": float((gg - gc).norm() / gc.norm())}
            del est
        with open(_name(a.out, a.net, a.steps, seed, "gpu"), "w") as f:
            f.write(json.dumps(rec) + "\n")
        print("gpu %s pair %d: %s" % (a.net, seed, {k: rec[k] for k in FLOORS}), file=sys.stderr, flush=True)


STEP_DIVERGENCE_AEE = 0.1   # per-step AEE(adv, target) further apart than this = the two runs are on different branches


def first_step_divergence(sa, sb, thr=STEP_DIVERGENCE_AEE):
    """sa, sb: per-step metric triples.  First step whose AEE(adv, target) differs by more than `thr`.  The fixed-step
    optimiser overshoots into the penalty every second / third step; on those steps ALL legs (the port's two thread counts
    included) sit 0.01-0.04 apart while sharing a branch, and a branch change (the overshoot cycle shifted by one step)
    shows as > 1 -- any threshold between 0.05 and 1 separates the two."""
    for k, (x, y) in enumerate(zip(sa, sb)):
        if abs(x[0] - y[0]) > max(thr, 0.02 * abs(y[0])):   # PWC-Net's AEE is 30-50: legs on one branch sit 0.05-0.3 apart
            return {"step": k, "aee_adv_tgt_a": x[0], "aee_adv_tgt_b": y[0]}
    return None


def first_divergence(la, lb, rel=DIVERGENCE_REL):
    """la, lb: per-step lists of closure losses.  First closure evaluation whose losses differ by more than `rel`."""
    idx = 0
    for k, (sa, sb) in enumerate(zip(la, lb)):
        for j, (x, y) in enumerate(zip(sa, sb)):
            if abs(x - y) > rel * abs(y):
                return {"step": k, "closure_in_step": j, "closure_index": idx + j, "loss_a": x, "loss_b": y,
                        "rel": abs(x - y) / abs(y)}
        if len(sa) != len(sb):
            return {"step": k, "closure_in_step": min(len(sa), len(sb)), "closure_index": idx + min(len(sa), len(sb)),
                    "what": "closure count differs", "a": len(sa), "b": len(sb)}
        idx += len(sa)
    return None


def _arbiter_summary(path):
    """fp64_arbiter record of an outside pair (tools/parity_arbiter.py): what the float64 port says about the GPU gradient
    at the first separated closures and about the first curvature pair."""
    import parity_arbiter
    r = json.load(open(path))
    port = "port%d" % r["threads"]
    worst = max(r["points"], key=lambda p: p["grad_rel_l2_vs_fp64"].get("gpu", 0.0))
    fp = r["first_curvature_pair"]
    return {"file": os.path.relpath(path, ROOT), "rule": r["rule"], "rule_ok_everywhere": r["rule_ok_everywhere"],
            "points": len(r["points"]), "at_steps": r["at_steps"],
            "worst_point": {"point": worst["point"], "gpu_vs_fp64": worst["grad_rel_l2_vs_fp64"].get("gpu"),
                            "port_vs_fp64": worst["grad_rel_l2_vs_fp64"].get(port), "tolerance": worst.get("tolerance")},
            "legs_on_one_side_of_the_gate": r["legs_on_one_side_of_the_gate"],
            "first_curvature_pair": {k: {"ys": v["ys"], "y_norm": v["y_norm"], "g0_norm": v["g0_norm"], "H": v["H"],
                                         "pair_stored": v["pair_stored"]} for k, v in fp.items()},
            "verdict": parity_arbiter._verdict(r)}


def cmd_assemble(a):
    legs = {}
    for path in sorted(glob.glob(os.path.join(a.dir, "*.json"))):
        r = json.load(open(path))
        legs[(r["net"], r["steps"], r["seed"], r["leg"])] = r
    groups = sorted({(n, s) for (n, s, _, _) in legs})
    out = {"what": __doc__.split("\n\n")[1].replace("\n", " "), "rule": "|GPU - port16| <= max(floor, 3 x |port16 - port8|), floors %s" % FLOORS,
           "step_divergence_threshold_aee": STEP_DIVERGENCE_AEE, "closure_loss_threshold_rel": DIVERGENCE_REL, "configs": []}
    for net, steps in groups:
        seeds = sorted({sd for (n, s, sd, _) in legs if (n, s) == (net, steps)})
        rows = []
        for sd in seeds:
            g, pa, pb = (legs.get((net, steps, sd, leg)) for leg in ("gpu", "port16", "port8"))
            row = {"pair": sd, "legs_present": [k for k, v in (("gpu", g), ("port16", pa), ("port8", pb)) if v]}
            for key in FLOORS:
                row[key] = {k: v[key] for k, v in (("gpu", g), ("port16", pa), ("port8", pb)) if v}
            if g and pa:
                inside = {}
                for key, floor in FLOORS.items():
                    spread = abs(pa[key] - pb[key]) if pb else None
                    tol = max(floor, 3 * spread) if spread is not None else floor
                    d = g[key] - pa[key]
                    row[key].update(delta=d, port_spread=spread, tolerance=tol, inside=abs(d) <= tol)
                    inside[key] = abs(d) <= tol
                row["inside_all"] = all(inside.values())
                row["first_step_divergence_gpu_vs_port16"] = first_step_divergence(g["per_step"], pa["per_step"])
                row["first_closure_loss_apart_gpu_vs_port16"] = first_divergence(g["closure_losses"], pa["closure_losses"])
                row["max_step_gap_aee_adv_tgt"] = max(abs(x[0] - y[0]) for x, y in zip(g["per_step"], pa["per_step"]))
                if "closure_delta_at_port_iterate" in g:
                    row["closure_delta_at_port_iterate"] = g["closure_delta_at_port_iterate"]
            if pa and pb:
                row["first_step_divergence_port16_vs_port8"] = first_step_divergence(pa["per_step"], pb["per_step"])
                row["max_step_gap_aee_adv_tgt_port16_vs_port8"] = max(abs(x[0] - y[0]) for x, y in zip(pa["per_step"], pb["per_step"]))
                row["port_hosts"] = {"port16": pa.get("host"), "port8": pb.get("host")}
            if a.arbiter and row.get("inside_all") is False:
                ap = os.path.join(a.arbiter, "%s_pair%d_arbiter.json" % (net.lower(), sd))
                row["fp64_arbiter"] = _arbiter_summary(ap) if os.path.exists(ap) else None
            if g:
                row["per_step_aee_adv_tgt_gpu"] = [round(s[0], 4) for s in g["per_step"]]
            if pa:
                row["per_step_aee_adv_tgt_port16"] = [round(s[0], 4) for s in pa["per_step"]]
            rows.append(row)
        full = [r for r in rows if "inside_all" in r and len(r["legs_present"]) == 3]   # the rule needs all three legs
        cfg = {"net": net, "steps": steps, "config": CONFIGS[net], "pairs": rows, "pairs_total": len(full),
               "pairs_with_two_legs_only": [r["pair"] for r in rows if "inside_all" in r and len(r["legs_present"]) < 3],
               "pairs_ok": sum(r["inside_all"] for r in full),
               "pairs_ok_per_metric": {k: sum(r[k]["inside"] for r in full) for k in FLOORS},
               "pairs_on_the_ports_branch": sum(r["first_step_divergence_gpu_vs_port16"] is None for r in full),
               "fraction_inside": (sum(r["inside_all"] for r in full) / len(full)) if full else None,
               "pairs_outside": [r["pair"] for r in full if not r["inside_all"]],
               "pairs_outside_cleared_by_fp64_arbiter": [r["pair"] for r in full if not r["inside_all"] and
                                                         (r.get("fp64_arbiter") or {}).get("rule_ok_everywhere")],
               "pairs_outside_without_arbiter_record": [r["pair"] for r in full if not r["inside_all"] and
                                                        not r.get("fp64_arbiter")]}
        # the same records read as DISTRIBUTIONS over the pairs (an un-damped L-BFGS attack is chaotic per pair -- the legs
        # take different branches at the same few steps -- so the per-pair rule mostly measures branch luck; what a user of
        # the attack sees is the distribution of its results): per leg mean / std / range of the best-iterate metrics, and
        # the per-pair |difference| of GPU vs port16 beside the port's own port16 vs port8
        def stats(v):
            v = sorted(v)
            m = sum(v) / len(v)
            return {"mean": m, "std": (sum((x - m) ** 2 for x in v) / len(v)) ** .5, "min": v[0], "max": v[-1],
                    "median": v[len(v) // 2] if len(v) % 2 else .5 * (v[len(v) // 2 - 1] + v[len(v) // 2])}
        if full:
            cfg["distribution"] = {key: {leg: stats([r[key][leg] for r in full]) for leg in ("gpu", "port16", "port8")}
                                   for key in FLOORS}
            cfg["abs_difference_over_pairs"] = {key: {"gpu_vs_port16": stats([abs(r[key]["gpu"] - r[key]["port16"]) for r in full]),
                                                      "gpu_vs_port8": stats([abs(r[key]["gpu"] - r[key]["port8"]) for r in full]),
                                                      "port16_vs_port8": stats([abs(r[key]["port16"] - r[key]["port8"]) for r in full])}
                                                for key in FLOORS}
            ad = cfg["abs_difference_over_pairs"]["aee_adv_tgt_min"]
            # one number for "is GPU-vs-port any different from port-vs-port?": the ratio of the median per-pair |differences|
            # (a chaotic attack -- PWC-Net: no two legs share a branch for 20 steps -- makes the per-pair rule a lottery)
            cfg["median_abs_difference_gpu_vs_port16_over_port16_vs_port8"] = (
                ad["gpu_vs_port16"]["median"] / ad["port16_vs_port8"]["median"] if ad["port16_vs_port8"]["median"] > 0 else None)
            cfg["first_split_step"] = {"gpu_vs_port16": [(r["first_step_divergence_gpu_vs_port16"] or {}).get("step") for r in full],
                                       "port16_vs_port8": [(r.get("first_step_divergence_port16_vs_port8") or {}).get("step") for r in full]}
            cl = [r["closure_delta_at_port_iterate"] for r in full if "closure_delta_at_port_iterate" in r]
            if cl:
                cfg["closure_at_the_split_gpu_vs_port16"] = {"max_loss_rel": max(c["loss_rel"] for c in cl),
                                                             "max_grad_rel_l2": max(c["grad_rel_l2"] for c in cl)}
        out["configs"].append(cfg)
    txt = json.dumps(out, indent=1)
    with open(a.out, "w") as f:
        f.write(txt + "\n")
    for cfg in out["configs"]:
        print("%s %d steps: %d/%d pairs inside (per metric %s), %d on the port's branch throughout" % (
            cfg["net"], cfg["steps"], cfg["pairs_ok"], cfg["pairs_total"], cfg["pairs_ok_per_metric"],
            cfg["pairs_on_the_ports_branch"]))
        if "distribution" in cfg:
            d, ad = cfg["distribution"]["aee_adv_tgt_min"], cfg["abs_difference_over_pairs"]["aee_adv_tgt_min"]
            print("    AEE(adv,tgt) at the best iterate, mean +- std over pairs: " + ", ".join(
                "%s %.3f +- %.3f" % (leg, d[leg]["mean"], d[leg]["std"]) for leg in ("gpu", "port16", "port8")))
            print("    per-pair |difference|, median / max: " + ", ".join(
                "%s %.3f / %.3f" % (k, v["median"], v["max"]) for k, v in ad.items()))


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    for name in ("port", "gpu"):
        p = sub.add_parser(name)
        p.add_argument("--net", default="RAFT", choices=sorted(CONFIGS))
        p.add_argument("--seeds", "--seed", dest="seeds", default="0")
        p.add_argument("--steps", type=int, default=20)
        p.add_argument("--out", required=True)
        p.add_argument("--snapshots", default="")
        p.add_argument("--force", action="store_true")
        if name == "port":
            p.add_argument("--threads", type=int, default=16)
    p = sub.add_parser("assemble")
    p.add_argument("--dir", required=True)
    p.add_argument("--out", required=True)
    p.add_argument("--arbiter", default="", help="directory of tools/parity_arbiter.py records: attached to every outside pair")
    a = ap.parse_args()
    if a.cmd != "assemble":
        a.seeds = [int(v) for v in str(a.seeds).split(",")]
    {"port": cmd_port, "gpu": cmd_gpu, "assemble": cmd_assemble}[a.cmd](a)


if __name__ == "__main__":
    main()
