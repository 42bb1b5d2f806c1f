#!/usr/bin/env python3
"""Is the GPU path reproducible from PROCESS to PROCESS?  (VERDICT r03 "next" item 1a.)

    python tools/process_repro.py [--net PWCNet] [--size 375x1242] [--box clipping] [--joint] [--steps 20]
                                  [--procs 2] [--seeds 0] [--out FILE.json]

The parent never touches the GPU: it starts `--procs` FRESH child processes of this script one after the other (no exec
from a GPU process).  Every child, per synthetic pair (`--seeds`):
  1. evaluates ONE eager closure at the initial variables + 0.02 N(0,1) (host seed) with every operator of the table
     (pcfa_amd.ops.get()) and every leaf nn.Module wrapped by a recorder: a bitwise checksum (sum of the int32 views,
     exact and order-free) of each output and of the gradient arriving at it -- the first record that differs between
     two processes names the kernel that is not process-stable;
  2. runs the captured-graph attack for `--steps` steps and records the loss of every closure evaluation, the per-step
     metrics and the best-iterate results as exact hex floats.
The parent compares the children record by record and prints one JSON object:
    {"identical": bool, "first_difference": {...} | null, "per_process": [...]}
Exit code 0 iff all processes agree bit for bit.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(a):
    import torch

    import bench
    from pcfa_amd import hip_ops, ops

    dev = torch.device("cuda", 0)
    h, w = (int(v) for v in a.size.split("x"))
    records = []

    def chk(t):
        t = t.detach().contiguous()
        if t.dtype != torch.float32:
            t = t.float()
        return int(t.view(torch.int32).to(torch.int64).sum().item())

    def note(key, t):
        records.append((key, None if t is None else chk(t)))   # an unused output hands None to its gradient hook

    counter = [0]

    def watch(name, out):
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for j, t in enumerate(outs):
            if torch.is_tensor(t) and t.is_floating_point():
                key = "%04d %s out%d %s" % (counter[0], name, j, tuple(t.shape))
                note(key, t)
                if t.requires_grad:
                    t.register_hook(lambda g, key=key: note(key + " <- grad", g))
        counter[0] += 1

    class Recorder:
        """The operator table with every call recorded (same kernels, same order: a proxy, not another implementation)."""

        def __getattr__(self, name):
            attr = getattr(hip_ops, name)
            if not callable(attr) or isinstance(attr, type):
                return attr

            def wrapped(*args, **kw):
                out = attr(*args, **kw)
                if recording[0]:
                    watch("ops." + name, out)
                return out
            return wrapped

    recording = [False]
    result = {"pairs": []}
    with ops.override_for_testing(Recorder()):
        conf = None
        if a.library_deconv:   # the r03 state: PWC-Net's transposed convolutions and x4 up-sampling in the library
            import dataclasses
            from pcfa_amd import config
            conf = dataclasses.replace(config.DEFAULT, deconv_fewout=False)
        model = bench.load_model(a.net, dev, a.box == "change_of_variables", conf)
        hooks = []
        for name, mod in model.named_modules():
            if not list(mod.children()):
                hooks.append(mod.register_forward_hook(
                    lambda m, i, o, name=name: watch("module." + name, o) if recording[0] else None))
        for seed in a.seeds:
            records.clear()
            counter[0] = 0
            st = bench.AttackStepper(a.net, h, w, dev, seed=seed, boxconstraint=a.box, joint=a.joint, model=model,
                                     use_graph=False)
            g = torch.Generator().manual_seed(7)
            saved = [p.detach().clone() for p in st.params]
            with torch.no_grad():
                for p in st.params:
                    p.add_((0.02 * torch.randn(p.shape, generator=g)).to(dev))
            recording[0] = True
            st.optimizer.zero_grad()
            loss = st.closure_body()
            recording[0] = False
            note("loss", loss.detach().reshape(1))
            for i, p in enumerate(st.params):
                note("grad of variable %d" % i, p.grad)
            with torch.no_grad():
                for p, s in zip(st.params, saved):
                    p.copy_(s)
            for p in st.params:
                p.grad = None
            closure_records = list(records)
            del st

            st = bench.AttackStepper(a.net, h, w, dev, seed=seed, boxconstraint=a.box, joint=a.joint, model=model,
                                     use_graph=True)
            losses = []
            inner = st.closure

            def closure():
                out = inner()
                losses.append(float(out))
                return out
            st.closure = closure
            steps = [st.step() for _ in range(a.steps)]
            res = st.result()
            result["pairs"].append({
                "seed": seed, "closure_records": closure_records,
                "closure_losses_hex": [float(x).hex() for x in losses],
                "steps_hex": [[float(v).hex() for v in s] for s in steps],
                "steps": [[float(v) for v in s] for s in steps],
                "best": {"aee_adv_tgt_min": res[9], "aee_adv_init_at_min": res[10], "l2_delta_min": res[11]},
                "graphed": st.graphed is not None})
            del st
        for hk in hooks:
            hk.remove()
    sys.stdout.write("PROCESS_REPRO " + json.dumps(result) + "\n")


def compare(runs):
    """first difference between process 0 and any other process (closure records first, then the trajectory)."""
    base = runs[0]
    for k, other in enumerate(runs[1:], start=1):
        for pa, pb in zip(base["pairs"], other["pairs"]):
            ra, rb = pa["closure_records"], pb["closure_records"]
            if len(ra) != len(rb):
                return {"process": k, "seed": pa["seed"], "what": "record count", "a": len(ra), "b": len(rb)}
            for (ka, va), (kb, vb) in zip(ra, rb):
                if ka != kb or va != vb:
                    return {"process": k, "seed": pa["seed"], "what": "eager closure record", "key": ka, "key_b": kb,
                            "checksum_a": va, "checksum_b": vb}
            for i, (la, lb) in enumerate(zip(pa["closure_losses_hex"], pb["closure_losses_hex"])):
                if la != lb:
                    return {"process": k, "seed": pa["seed"], "what": "loss of closure evaluation (graph replay)",
                            "closure_index": i, "step": i // 10, "a": float.fromhex(la), "b": float.fromhex(lb)}
            if pa["steps_hex"] != pb["steps_hex"]:
                i = next(j for j, (x, y) in enumerate(zip(pa["steps_hex"], pb["steps_hex"])) if x != y)
                return {"process": k, "seed": pa["seed"], "what": "per-step metrics", "step": i, "a": pa["steps"][i],
                        "b": pb["steps"][i]}
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="PWCNet")
    ap.add_argument("--size", default="375x1242")
    ap.add_argument("--box", default="clipping", choices=["clipping", "change_of_variables"])
    ap.add_argument("--joint", action="store_true")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--procs", type=int, default=2)
    ap.add_argument("--seeds", default="0")
    ap.add_argument("--out", default="")
    ap.add_argument("--library-deconv", action="store_true",
                    help="Config(deconv_fewout=False): PWC-Net's deconv / upfeat layers and the x4 up-sampling in the library "
                         "(what r03 shipped) -- shows what the probe reports for a process-unstable path")
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    a.seeds = [int(v) for v in str(a.seeds).split(",")]
    if a.child:
        child(a)
        return
    runs = []
    for k in range(a.procs):
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--net", a.net, "--size", a.size, "--box", a.box,
               "--steps", str(a.steps), "--seeds", ",".join(str(s) for s in a.seeds)] + (["--joint"] if a.joint else []) + \
              (["--library-deconv"] if a.library_deconv else [])
        p = subprocess.run(cmd, capture_output=True, text=True)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("PROCESS_REPRO ")]
        if p.returncode != 0 or not line:
            sys.stderr.write(p.stdout[-2000:] + p.stderr[-4000:])
            sys.exit(2)
        runs.append(json.loads(line[-1][len("PROCESS_REPRO "):]))
        print("process %d done: %s" % (k, [pr["best"] for pr in runs[-1]["pairs"]]), file=sys.stderr, flush=True)
    diff = compare(runs)
    out = {"what": "%d fresh processes, %s %s (%s%s), %d-step captured-graph attack + one recorded eager closure per pair"
                   % (a.procs, a.net, a.size, a.box, ", joint" if a.joint else "", a.steps),
           "seeds": a.seeds, "identical": diff is None, "first_difference": diff,
           "records_per_closure": len(runs[0]["pairs"][0]["closure_records"]),
           "per_process": [[{"seed": pr["seed"], **pr["best"], "last_step": pr["steps"][-1] if pr["steps"] else None,
                             "graphed": pr["graphed"]} for pr in r["pairs"]] for r in runs]}
    txt = json.dumps(out)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")
    print(txt)
    sys.exit(0 if diff is None else 1)


if __name__ == "__main__":
    main()
