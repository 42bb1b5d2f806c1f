"""Perturbation-Constrained Flow Attack driver -- the hot loop of the reference's attack_PCFA.py.

    extract_deltas / extract_deltas_joint   attack_PCFA.py:20-37   (fused HIP, pcfa_amd.ops)
    pcfa_attack                             attack_PCFA.py:40-294
    attack_l2_universal                     attack_PCFA.py:297-566
    attack_l2                               attack_PCFA.py:570-701

Schedule: the averaged-L2 bound is an exact penalty minimised by
L-BFGS(max_iter=10, no line search: pcfa_amd.lbfgs.LBFGS = torch.optim.LBFGS semantics on HIP kernels); one `--steps` iteration = one LBFGS.step = 10
closure evaluations (forward + loss + backward) followed by one re-prediction forward, and the
reported result is the best iterate with ||delta|| <= bound.  Kept: the optimiser's algorithm and host
decisions, the closure arithmetic and the best-iterate rule.  Dropped because they cannot change a result:
  * the `loss.backward()` before every LBFGS.step (attack_PCFA.py:173) -- the closure's
    zero_grad() discards its gradient before anything reads it;
  * the autograd graph of the re-prediction forward (only metrics read it) -> torch.no_grad();
  * torch.autograd.set_detect_anomaly, the per-forward `.cpu()` round trip, mlflow.

Multi-GPU (one process per GPU, torch.distributed over RCCL): independent pairs are sharded
round-robin over ranks with no collective in the data path (`attack_l2`); the universal attack
is data parallel over the batch with ONE all-reduce of d(loss)/d(delta) per closure
(`attack_l2_universal`), see pcfa_amd/sharding.py.
"""
import os
import weakref
import time

import numpy as np
import torch

from . import config, ops, sharding
from .helper_functions import datasets, logging, losses, ownutilities, parsing_file, targets
from .helper_functions.config_paths import Conf

EPS_BOX = 1e-7  # attack_PCFA.py:608


def extract_deltas(nw_input1, nw_input2, image1, image2, boxconstraint, eps_box=0.):
    """delta_i = box(nw_input_i) - image_i (attack_PCFA.py:20-29)."""
    return ops.get().extract_deltas(nw_input1, nw_input2, image1, image2, boxconstraint, eps_box=eps_box)


def extract_deltas_joint(nw_delta, images_max, images_min):
    """Two-sided clamp of a shared perturbation against both frames (attack_PCFA.py:32-37)."""
    return ops.get().extract_deltas_joint(nw_delta, images_max, images_min)


def default_mu(args):
    """attack_PCFA.py:578-584."""
    optim_mu = args.mu
    if optim_mu == -1.:
        optim_mu = 2500. / args.delta_bound
        if args.target not in ['zero']:
            optim_mu = 1.5 * optim_mu
    return optim_mu


def select_device():
    """attack_PCFA.py:631-634."""
    if Conf.config('useCPU') or not torch.cuda.is_available():
        return torch.device("cpu")
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.device("cuda", sharding.local_rank())
    return torch.device("cuda")


def _should_save(batch, args):
    return ((batch % args.save_frequency == 0 and not args.small_save) or (args.small_save and batch < 32)) \
        and not args.no_save


def _graphs_enabled(device, args):
    return torch.device(device).type == "cuda" and os.environ.get("PCFA_HIP_GRAPH", "1") == "1" and args.steps > 0


class _PairGraphs:
    """What survives from one image pair to the next of the same shape and flags (attack_PCFA.py:668-670 loops over
    a batch-1 loader: every KITTI / Sintel pair has the same size): the static device buffers the captured closure
    reads and writes, the hipGraphs themselves and the optimiser whose flat gradient buffer the closure fills.  A new
    pair copies its images / initial variables / target into these buffers and resets the optimiser -- no warm-up, no
    re-capture (`PairAttack(..., reuse_graphs=True)`, the default on the GPU; results equal a fresh capture:
    tests/test_gpu_parity.py::test_pair_graph_reuse_equals_fresh_capture)."""

    def __init__(self, st):
        self.image1, self.image2 = st.image1, st.image2
        self.images_max, self.images_min = st.images_max, st.images_min
        self.params, self.target, self.optimizer = st.params, st.target, st.optimizer
        self.graphed, self.repredict = st.graphed, st.repredict
        self.pairs = 1
        self.owner = weakref.ref(st)   # the ONE live PairAttack of this key; retired when the next pair adopts the set

    def retire_owner(self):
        st = self.owner()
        if st is not None:
            st._retire()


def _max_cached_shapes(model):
    """Graph sets kept per model (LRU), from the model's Config (default 4; KITTI under /8 padding alone has three padded
    shapes).  Each set pins a closure graph pool + a 2 x 101 x n L-BFGS history (~1.1 GB at 440x1024)."""
    return config.cfg(model).max_cached_shapes


def _log_eviction(what, key):
    import logging as pylog
    pylog.warning("pcfa_amd: %s cache full: dropping the graph set of %r -- the next pair of that shape pays warm-up + "
                  "capture again (raise Config.max_cached_shapes / PCFA_MAX_CACHED_SHAPES if this repeats)", what, key)


def _cache_put(cache, key, kept, cap):
    cache.pop(key, None)
    cache[key] = kept                      # dicts keep insertion order: the last entry is the most recently used
    while len(cache) > cap:
        old_key = next(iter(cache))
        old = cache.pop(old_key)
        _log_eviction("pair-graph", old_key)
        old.retire_owner()                 # its PairAttack must not replay graphs whose buffers are about to go
        old.graphed = old.repredict = old.optimizer = None


def _graph_cache(model):
    cache = getattr(model, "_pcfa_pair_graphs", None)
    if cache is None:
        cache = {}
        try:
            object.__setattr__(model, "_pcfa_pair_graphs", cache)   # plain attribute: not a module / parameter
        except Exception:  # noqa: BLE001
            return {}
    return cache


class PairAttack:
    """Everything pcfa_attack holds for ONE image pair (attack_PCFA.py:40-247): the optimisation variables,
    the optimiser, the target, the best-iterate bookkeeping -- and `step()`, the body of the `--steps` loop.

    `pcfa_attack` is `PairAttack(...)` + `args.steps` x `step()`; bench.py times `step()` of this class, so the
    measured loop IS the product loop.  On the GPU the closure (static shapes, variables updated in place by
    L-BFGS) and the re-prediction forward are captured once per pair into hipGraphs and replayed
    (`use_graph=None`: on for CUDA devices unless PCFA_HIP_GRAPH=0; if capture fails the eager closure -- the same
    kernels in the same order -- is used and a warning is logged)."""

    def __init__(self, model, image1, image2, flow, batch, eps_box, device, has_gt, optim_mu, args, use_graph=None,
                 reuse_graphs=None):
        self.model, self.args, self.device, self.batch = model, args, device, batch
        if reuse_graphs is None:   # pairs of one shape share static buffers + hipGraphs unless the model's config says no
            reuse_graphs = config.cfg(model).reuse_pair_graphs
        self.has_gt, self.optim_mu, self.eps_box = has_gt, optim_mu, eps_box
        curr_step = batch * args.steps

        image1, image2 = image1.to(device), image2.to(device)
        self.flow_gt = flow.to(device) if flow is not None else None
        if not ownutilities.model_takes_unit_input(args.net):
            image1 = image1 / 255.
            image2 = image2 / 255.
        self.padder, [image1, image2] = ownutilities.preprocess_img(args.net, image1, image2)
        image1.requires_grad = False
        image2.requires_grad = False
        cov = args.boxconstraint in ['change_of_variables']
        if args.joint_perturbation and cov:
            raise ValueError("Training a --joint_perturbation with --boxconstraint=change_of_variables is not "
                             "defined. Please use --boxconstraint=clipping.")
        if use_graph is None:
            use_graph = _graphs_enabled(device, args)
        share_forward = os.environ.get("PCFA_SHARED_FORWARD", "0") == "1"
        self.graph_key = (args.net, tuple(image1.shape), args.boxconstraint, bool(args.joint_perturbation), args.loss,
                          float(optim_mu), float(args.delta_bound), float(eps_box), str(device), ops.core.current_lane())
        kept = _graph_cache(model).get(self.graph_key) if (use_graph and reuse_graphs and not share_forward) else None
        self.graphed = self.repredict = None
        self.graphs_reused = kept is not None
        self.retired = False

        if kept is not None:
            # same shape and flags as an earlier pair: its static buffers, graphs and optimiser, refilled.  The earlier
            # PairAttack (if still alive) is retired first: its variables, target and optimiser state are this pair's now.
            kept.retire_owner()
            kept.owner = weakref.ref(self)
            _cache_put(_graph_cache(model), self.graph_key, kept, _max_cached_shapes(model))   # most recently used
            with torch.no_grad():
                kept.image1.copy_(image1)
                kept.image2.copy_(image2)
                kept.images_max.copy_(torch.max(image1, image2))
                kept.images_min.copy_(torch.min(image1, image2))
                if args.joint_perturbation:
                    kept.params[0].zero_()
                elif cov:
                    kept.params[0].copy_(torch.atanh(2. * (1. - eps_box) * image1 - (1 - eps_box)))
                    kept.params[1].copy_(torch.atanh(2. * (1. - eps_box) * image2 - (1 - eps_box)))
                else:
                    kept.params[0].copy_(image1)
                    kept.params[1].copy_(image2)
            image1, image2 = kept.image1, kept.image2
            self.image1, self.image2 = image1, image2
            self.images_max, self.images_min = kept.images_max, kept.images_min
            self.params = kept.params
            for p_ in self.params:
                p_.grad = None
            kept.optimizer.reset()
            self.optimizer = kept.optimizer
            self.graphed, self.repredict = kept.graphed, kept.repredict
            kept.pairs += 1
            delta1 = torch.zeros_like(image1)
            delta2 = torch.zeros_like(image2)
            if args.joint_perturbation:
                self.nw_delta = self.params[0]
                self.nw_input1, self.nw_input2 = image1, image2
                self.fwd_kwargs = {"delta1": self.nw_delta}
            else:
                self.nw_delta = None
                self.nw_input1, self.nw_input2 = self.params
                self.fwd_kwargs = {}
            _, flow0 = self.repredict()           # the captured forward at the initial variables = unattacked flow
            self.flow_pred_init = flow0.detach().clone()
            target = targets.get_target(args.target, self.flow_pred_init,
                                        custom_target_path=args.custom_target_path, device=device).to(device)
            kept.target.copy_(target)
            self.target = kept.target
        else:
            self.image1, self.image2 = image1, image2
            self.images_max = torch.max(image1, image2).detach()
            self.images_min = torch.min(image1, image2).detach()
            delta1 = torch.zeros_like(image1)
            delta2 = torch.zeros_like(image2)
            self.nw_delta = None
            if args.joint_perturbation:
                self.nw_delta = delta1
                self.nw_delta.requires_grad = True
                self.nw_input1, self.nw_input2 = image1, image2
                self.params = [self.nw_delta]
                self.fwd_kwargs = {"delta1": self.nw_delta}
            else:
                if cov:
                    self.nw_input1 = torch.atanh(2. * (1. - eps_box) * (image1 + delta1) - (1 - eps_box))
                    self.nw_input2 = torch.atanh(2. * (1. - eps_box) * (image2 + delta2) - (1 - eps_box))
                else:
                    self.nw_input1 = image1 + delta1
                    self.nw_input2 = image2 + delta2
                self.nw_input1.requires_grad = True
                self.nw_input2.requires_grad = True
                self.params = [self.nw_input1, self.nw_input2]
                self.fwd_kwargs = {}
            self.optimizer = ops.get().LBFGS(self.params, max_iter=10)

            with torch.no_grad():
                self.flow_pred_init = self.predict().detach().clone()
            self.target = targets.get_target(args.target, self.flow_pred_init,
                                             custom_target_path=args.custom_target_path, device=device).to(device)
        self.target.requires_grad = False

        self.aee_tgt = logging.calc_metrics_const(self.target, self.flow_pred_init)
        if has_gt:
            self.aee_gt_tgt, self.aee_gt = logging.calc_metrics_const_gt(self.target, self.flow_pred_init,
                                                                         self.flow_gt)
        else:
            self.aee_gt_tgt, self.aee_gt = None, None
        logging.log_metrics(curr_step, ("aee_pred-tgt", self.aee_tgt), ("aee_gt-tgt", self.aee_gt_tgt),
                            ("aee_pred-gt", self.aee_gt))
        logging.log_metric(key="optim_mu", value=optim_mu, step=curr_step)

        model.zero_grad()
        self.optimizer.zero_grad()

        self.delta_below_threshold = False
        self.delta12_min_val = float('inf')
        self.aee_adv_tgt_min_val = float('inf')
        self.aee_adv_pred_min_val = 0.
        self.delta1_min = self.delta2_min = self.flow_pred_min = None
        self.flow_pred = self.flow_pred_init
        self.delta1, self.delta2 = delta1, delta2
        self.aee_adv_gt = 0.
        self.aee_adv_tgt = self.aee_adv_pred = 0.
        self.l2_delta1 = self.l2_delta2 = self.l2_delta12 = 0.
        self.steps_done = 0
        self.closures = 0

        self.reuse_graphs = reuse_graphs
        if use_graph and kept is None:
            self.enable_graph()

    # ---- pieces of the closure (attack_PCFA.py:175-192) ---------------------------------------------------------
    def predict(self):
        out = ownutilities.compute_flow(self.model, "scaled_input_model", self.nw_input1, self.nw_input2,
                                        test_mode=True, **self.fwd_kwargs)
        [out] = ownutilities.postprocess_flow(self.args.net, self.padder, out)
        return out

    def current_deltas(self):
        if self.args.joint_perturbation:
            return extract_deltas_joint(self.nw_delta, self.images_max, self.images_min)
        return extract_deltas(self.nw_input1, self.nw_input2, self.image1, self.image2, self.args.boxconstraint,
                              eps_box=self.eps_box)

    def closure_body(self):
        flow_closure = self.predict()
        d1, d2 = self.current_deltas()
        loss_closure = losses.loss_delta_constraint(flow_closure, self.target, d1, d2, self.device,
                                                    delta_bound=self.args.delta_bound, mu=self.optim_mu,
                                                    f_type=self.args.loss)
        loss_closure.backward()
        return loss_closure

    def _repredict_body(self):
        return self.current_deltas(), self.predict()

    def _forward_with_loss(self):
        """closure_body without the backward: (loss, ((delta1, delta2), flow)) -- see graphed.SplitGraphedClosure."""
        flow = self.predict()
        d1, d2 = self.current_deltas()
        loss = losses.loss_delta_constraint(flow, self.target, d1, d2, self.device, delta_bound=self.args.delta_bound,
                                            mu=self.optim_mu, f_type=self.args.loss)
        return loss, ((d1, d2), flow)

    def enable_graph(self, share_forward=None):
        """share_forward (default: PCFA_SHARED_FORWARD=1, off otherwise): capture the closure as forward + backward graphs
        and let the re-prediction after a step double as the forward of the next step's first closure evaluation."""
        if share_forward is None:
            share_forward = os.environ.get("PCFA_SHARED_FORWARD", "0") == "1"
        try:
            from .graphed import GraphedClosure, GraphedForward, SplitGraphedClosure
            if share_forward:
                split = SplitGraphedClosure(self._forward_with_loss, self.params)
                self.graphed = split

                def repredict():
                    _, ((d1, d2), flow) = split.forward()
                    return (d1.detach(), d2.detach()), flow.detach()
                self.repredict = repredict
                return
            sink = getattr(self.optimizer, "flat_grad_views", None)
            self.graphed = GraphedClosure(self.closure_body, self.params, grad_sink=sink() if sink else None)
            self.repredict = GraphedForward(self._repredict_body, self.device)
            if self.reuse_graphs and hasattr(self.optimizer, "reset"):
                _cache_put(_graph_cache(self.model), self.graph_key, _PairGraphs(self),
                           _max_cached_shapes(self.model))   # the next pair of this shape adopts them
        except Exception as e:  # noqa: BLE001 -- the eager closure launches the same kernels in the same order
            import logging as pylog
            pylog.warning("hipGraph capture of the closure failed (%r): launching eagerly", e)
            self.graphed = self.repredict = None

    def _retire(self):
        """A later pair of the same shape took over this pair's static buffers, graphs and optimiser (or the set was
        evicted from the per-model cache): results recorded so far stay readable, further evaluation raises."""
        self.retired = True
        self.graphed = self.repredict = None

    def _check_live(self):
        if self.retired:
            raise RuntimeError("this PairAttack was retired: a later pair of the same shape adopted its static buffers, "
                               "graphs and optimiser (one live PairAttack per shape and model; reuse_graphs=False keeps "
                               "pairs independent)")

    def closure(self):
        self._check_live()
        self.closures += 1
        if self.graphed is not None:
            return self.graphed()
        self.optimizer.zero_grad()
        return self.closure_body()

    # ---- one `--steps` iteration (attack_PCFA.py:155-247) -------------------------------------------------------
    def step(self):
        self._check_live()
        args = self.args
        steps = self.steps_done
        curr_step = self.batch * args.steps + steps
        logging.log_metrics(curr_step, ("batch", self.batch), ("steps", steps), ("epoch", 0))

        self.optimizer.step(self.closure)

        if self.repredict is not None:
            (delta1, delta2), flow_pred = self.repredict()
        else:
            with torch.no_grad():
                (delta1, delta2), flow_pred = self._repredict_body()
        self.delta1, self.delta2, self.flow_pred = delta1, delta2, flow_pred

        aee_adv_tgt, aee_adv_pred = logging.calc_metrics_adv(flow_pred, self.target, self.flow_pred_init)
        self.aee_adv_gt = logging.calc_metrics_adv_gt(flow_pred, self.flow_gt) if self.has_gt else None
        logging.log_metrics(curr_step, ("aee_predadv-tgt", aee_adv_tgt), ("aee_pred-predadv", aee_adv_pred),
                            ("aee_predadv-gt", self.aee_adv_gt))
        l2_delta1, l2_delta2, l2_delta12 = logging.calc_delta_metrics(delta1, delta2, curr_step)
        logging.log_metrics(curr_step, ("l2_delta1", l2_delta1), ("l2_delta2", l2_delta2),
                            ("l2_delta-avg", l2_delta12))
        self.aee_adv_tgt, self.aee_adv_pred = aee_adv_tgt, aee_adv_pred
        self.l2_delta1, self.l2_delta2, self.l2_delta12 = l2_delta1, l2_delta2, l2_delta12

        # best-iterate rule, attack_PCFA.py:226-243
        update_minima = False
        if not self.delta_below_threshold:
            if l2_delta12 < self.delta12_min_val or (l2_delta12 == self.delta12_min_val
                                                     and aee_adv_tgt < self.aee_adv_tgt_min_val):
                update_minima = True
                if l2_delta12 <= args.delta_bound:
                    self.delta_below_threshold = True
        elif l2_delta12 <= args.delta_bound and aee_adv_tgt < self.aee_adv_tgt_min_val:
            update_minima = True
        if update_minima:
            self.delta12_min_val = l2_delta12
            self.aee_adv_tgt_min_val = aee_adv_tgt
            self.aee_adv_pred_min_val = aee_adv_pred
            self.delta1_min = delta1.detach().clone()
            self.delta2_min = delta2.detach().clone()
            self.flow_pred_min = flow_pred.detach().clone()
        logging.log_metrics(curr_step, ("aee_pred-tgt_min", self.aee_adv_tgt_min_val),
                            ("l2_delta-avg_min", self.delta12_min_val),
                            ("aee_pred-predadv_min", self.aee_adv_pred_min_val))
        self.steps_done += 1
        return aee_adv_tgt, aee_adv_pred, l2_delta12

    def save(self, distortion_folder):
        batch = self.batch
        for tens, name in ((self.delta1, "delta1_final"), (self.delta2, "delta2_final"),
                           (self.delta1_min, "delta1_best"), (self.delta2_min, "delta2_best"),
                           (self.image1, "image1"), (self.image2, "image2"), (self.target, "target"),
                           (self.flow_pred, "flow_pred_final"), (self.flow_pred_min, "flow_pred_best"),
                           (self.flow_pred_init, "flow_pred_init")):
            logging.save_tensor(tens, name, batch, distortion_folder)
        if self.has_gt:
            logging.save_tensor(self.flow_gt, "flow_gt", batch, distortion_folder)

    def result(self):
        """The reference's 12-tuple (attack_PCFA.py:294)."""
        return (self.aee_gt, self.aee_tgt, self.aee_gt_tgt, self.aee_adv_gt, self.aee_adv_tgt, self.aee_adv_pred,
                self.l2_delta1, self.l2_delta2, self.l2_delta12, self.aee_adv_tgt_min_val,
                self.aee_adv_pred_min_val, self.delta12_min_val)


def pcfa_attack(model, image1, image2, flow, batch, distortion_folder, eps_box, device, has_gt, optim_mu, args,
                statistics_in_every_step=True):
    """Attack one image pair; returns the reference's 12-tuple (attack_PCFA.py:40-294)."""
    st = PairAttack(model, image1, image2, flow, batch, eps_box, device, has_gt, optim_mu, args)
    if args.steps == 0:  # the reference returns its initial zeros for the adversarial statistics
        st.aee_gt = st.aee_gt if has_gt else None
    for _ in range(args.steps):
        st.step()
    if distortion_folder is not None and _should_save(batch, args):
        st.save(distortion_folder)
    return st.result()


def _load_model(args, device, variable_change):
    model_takes_unit_input = ownutilities.model_takes_unit_input(args.net)
    kwargs = {"weights": getattr(args, "weights", "pretrained")}
    if variable_change:
        kwargs["eps_box"] = EPS_BOX
    model = ownutilities.import_and_load(args.net, make_unit_input=not model_takes_unit_input,
                                         variable_change=variable_change, make_scaled_input_model=True,
                                         device=device, **kwargs)
    model.eval()
    for param in model.parameters():
        param.requires_grad = False
    return model


def _output_folder(args, tag):
    if args.no_save:
        return None
    stamp = time.strftime("%Y-%m-%d_%H:%M:%S")
    kind = "%s_PCFA_%s_%s" % (args.net, "cd" if args.joint_perturbation else "dd",
                              "u" if args.universal_perturbation else "-")
    folder = os.path.join(args.output_folder, kind, stamp + tag)
    return logging.create_subfolder(folder, "patches")


class PairsInFlight:
    """Several independent image pairs attacked SIDE BY SIDE on one GPU (attack_PCFA.py:668-670 loops over pairs one after
    the other; the pairs share nothing but the frozen weights).

    (The lane travels with the STREAM, not with the host thread: autograd runs backward nodes on a thread of its own.)
    At one pair per GPU every launch of the 55 x 128 feature maps is 1-2 workgroups per CU and ends in a tail; a second
    pair's launches fill the idle CUs.  Lane k = one host thread + one HIP stream + one set of static buffers, hipGraphs
    and L-BFGS state (`ops.core.lane(k)` keys every shared scratch buffer and the per-model graph cache), so the lanes
    never touch each other's memory and each pair's arithmetic is exactly the solo run's: results are bit-identical to
    attacking the pairs one after the other (tests/test_gpu_parity.py::test_pairs_in_flight_bit_identical_to_solo).

    make_attack(k) -> PairAttack is called in the caller's thread, one lane after the other (weight packs and graph
    captures are built sequentially); `run(steps)` then drives every lane's `step()` from its own thread."""

    def __init__(self, make_attack, n, device):
        self.device = torch.device(device)
        if self.device.index is None:      # "cuda": the caller's current device (a new host thread starts on device 0)
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.streams = ops.core.lane_run_streams(self.device, n)   # (not torch.cuda.Stream(): a pool of 32 shared objects)
        self.attacks = []
        for k in range(n):
            with ops.core.lane(k), torch.cuda.stream(self.streams[k]):
                self.attacks.append(make_attack(k))
            torch.cuda.synchronize(self.device)
            if k == 0 and n > 1:
                self._refuse_shared_library_workspaces(self.attacks[0])

    @staticmethod
    def _refuse_shared_library_workspaces(attack):
        """GMA's attention products run on rocBLAS by default (Config.gma_gemm = "lib").  Every graph of this
        process is captured by ONE host thread, i.e. with one rocBLAS handle, and a handle owns ONE device workspace that
        all its launches share in stream order -- replayed side by side, two lanes' products use it at the same time.
        Observed (r05, tools/dev/flight_scaling.py GMA 436x1024, two lanes): the first step never returns.  The package's own
        products take their split-K scratch from the caller, so the build with gma_gemm = "hip" is safe in flight."""
        net = getattr(getattr(attack, "args", None), "net", None)
        if net in ("SpyNet", "FlowNet2"):
            # their closures keep library convolutions (SpyNet's 7x7 layers, FlowNet2's transposed convolutions): the same
            # question, never validated -- refused rather than left to chance
            raise ValueError("%s keeps library convolutions inside its captured closure: several pairs in flight are "
                             "validated for RAFT, PWCNet and GMA (gma_gemm='hip') only" % net)
        if net == "GMA" and config.cfg(attack.model).gma_gemm != "hip":
            raise ValueError("GMA with Config.gma_gemm='lib' cannot run several pairs in flight (the lanes' captured rocBLAS "
                             "products would share one handle's workspace): build the model with gma_gemm='hip' "
                             "(PCFA_GMA_GEMM=hip) or use --pairs_in_flight 1")

    def run(self, steps):
        """`steps` attack steps on every lane, concurrently; returns each lane's last (aee_adv_tgt, aee_adv_pred, l2)."""
        import threading
        last, errors = [None] * len(self.attacks), []
        start = threading.Barrier(len(self.attacks))

        def drive(k):
            try:
                torch.cuda.set_device(self.device)       # the current device is per host thread
                with ops.core.lane(k), torch.cuda.stream(self.streams[k]):
                    start.wait()
                    for _ in range(steps):
                        last[k] = self.attacks[k].step()
                    self.streams[k].synchronize()
            except BaseException as e:  # noqa: BLE001 -- re-raised in the caller's thread
                errors.append(e)
                start.abort()
        threads = [threading.Thread(target=drive, args=(k,), name="pcfa-lane-%d" % k) for k in range(len(self.attacks))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return last


def _hardware_queues_for(nflight):
    """Kernels of two lanes overlap only if their streams sit on different hardware queues, and the HIP runtime makes four
    per device unless GPU_MAX_HW_QUEUES says otherwise when it initialises.  Up to three pairs in flight that is enough; the
    fourth lane shares a queue (RAFT, four pairs: 9.65 pair-steps/s with 4 queues, 10.73 with 8;
    profiles/r05/pairs_in_flight_scaling.txt).  So ask for 8 -- which only works before the process first touches the GPU."""
    if nflight < 4 or "GPU_MAX_HW_QUEUES" in os.environ:
        return
    if torch.cuda.is_initialized():
        print("--pairs_in_flight %d: the GPU runtime is already initialised with its default of 4 hardware queues; export "
              "GPU_MAX_HW_QUEUES=8 before starting for the fourth lane to overlap" % nflight)
        return
    os.environ["GPU_MAX_HW_QUEUES"] = "8"


def attack_l2(args, data_loader=None, has_gt=None):
    """PCFA on every pair of a dataset, one perturbation (pair) per image pair (attack_PCFA.py:570-701).

    With torch.distributed initialised, pair i is attacked on rank i % world_size and rank 0
    receives every pair's result tuple; there is no collective inside the attack.
    Returns the dict of averaged metrics (also logged under the reference's metric names).
    """
    optim_mu = default_mu(args)
    rank, world = sharding.rank(), sharding.world_size()
    distortion_folder = _output_folder(args, "_r%d" % rank if world > 1 else "")
    if data_loader is None:
        data_loader, has_gt = datasets.prepare_dataloader(args, batch_size=1, shuffle=False)
    _hardware_queues_for(max(1, int(getattr(args, "pairs_in_flight", 1))))
    device = select_device()
    cov = args.boxconstraint in ['change_of_variables']
    model = _load_model(args, device, variable_change=cov)

    local = []
    nflight = max(1, int(getattr(args, "pairs_in_flight", 1)))
    if nflight > 1 and torch.device(device).type != "cuda":
        raise ValueError("--pairs_in_flight needs a GPU (lanes are HIP streams)")
    pending = []   # (batch, image1, image2, flow) of this rank, attacked `nflight` at a time

    def attack_group(group):
        if len(group) == 1 or nflight == 1:
            for batch, image1, image2, flow in group:
                res = pcfa_attack(model, image1, image2, flow, batch, distortion_folder, EPS_BOX, device, has_gt, optim_mu,
                                  args)
                local.append((batch,) + tuple(float('nan') if v is None else float(v) for v in res))
            return
        # the pairs of a dataset are independent (attack_PCFA.py:668-670): lane k = stream + graph set + scratch of its own,
        # every pair's result bit-identical to attacking it alone (PairsInFlight)
        flight = PairsInFlight(lambda k: PairAttack(model, group[k][1], group[k][2], group[k][3], group[k][0], EPS_BOX, device,
                                                    has_gt, optim_mu, args), len(group), device)
        if args.steps > 0:
            flight.run(args.steps)
        for (batch, _, _, _), st in zip(group, flight.attacks):
            if args.steps == 0:
                st.aee_gt = st.aee_gt if has_gt else None
            if distortion_folder is not None and _should_save(batch, args):
                st.save(distortion_folder)
            local.append((batch,) + tuple(float('nan') if v is None else float(v) for v in st.result()))

    for batch, (image1, image2, flow, _) in enumerate(data_loader):
        if batch % world != rank:
            continue
        pending.append((batch, image1, image2, flow))
        if len(pending) == nflight:
            attack_group(pending)
            pending = []
    if pending:
        attack_group(pending)

    rows = sharding.gather_rows(local, width=13, device=device)
    if rank != 0:
        return None
    rows = sorted(rows, key=lambda r: r[0])
    tests = len(rows)
    names = ("aee_avg_pred-gt", "aee_avg_pred-tgt", "aee_avg_gt-tgt", "aee_avg_predadv-gt", "aee_avg_predadv-tgt",
             "aee_avg_pred-predadv", "l2_avg_delta1", "l2_avg_delta2", "l2_avg_delta12", "aee_avg_predadv-tgt_min",
             "aee_avg_pred-predadv_min", "l2_avg_delta12_min")
    sums = {n: float(np.nansum([r[i + 1] for r in rows])) for i, n in enumerate(names)}
    logging.calc_log_averages(tests, *sums.items())
    result = {n: (v / tests if tests else float('nan')) for n, v in sums.items()}
    result["pairs"] = tests
    print("\nFinished attacking with PCFA. The best achieved values are")
    print("\tAEE(f_adv, f_init)=%f" % result["aee_avg_pred-predadv_min"])
    print("\tAEE(f_adv, f_targ)=%f" % result["aee_avg_predadv-tgt_min"])
    print("\tL2(perturbation)  =%f" % result["l2_avg_delta12_min"])
    return result


class _UniversalBatchState:
    """Static device buffers + captured hipGraphs of the universal closure for ONE batch shape."""

    def __init__(self, image1, image2, padder):
        self.image1 = image1.clone()
        self.image2 = image2.clone()
        self.padder = padder
        self.target = None
        self.graphed = None
        self.repredict = None
        self.tried_graph = False


class UniversalAttack:
    """State of attack_l2_universal (attack_PCFA.py:297-566): ONE perturbation (pair) `[3,Hp,Wp]` shared by every
    image, ONE L-BFGS optimiser for the whole run, data parallel over the batch.

    Every rank holds batch_size/world_size pairs of each global batch, a replica of delta and of the L-BFGS
    state.  Per closure: local forward/backward, then ONE all-reduce of a flat buffer holding d(loss)/d(delta) of
    both perturbations and the scalar loss (sharding.FlatReducer).  The penalty depends on delta only, so it is
    identical on every rank and its average is itself; the similarity term is a mean over the batch, so the
    average of the local means equals the reference's single-process mean over the global batch
    (attack_PCFA.py:469-490).

    On the GPU forward + loss + backward (+ the packing of gradients and loss for the all-reduce) is captured ONCE
    per batch shape into a hipGraph: delta is updated in place by L-BFGS and the images / target of a new batch
    are copied into the same static buffers, so every closure evaluation of every batch replays the same graph;
    the collective runs between replays, outside the graph (PCFA_HIP_GRAPH=0 disables capture)."""

    def __init__(self, model, delta_like1, delta_like2, device, optim_mu, args, use_graph=None):
        self.model, self.args, self.device, self.optim_mu = model, args, device, optim_mu
        self.world = sharding.world_size()
        # cosim is a ratio of sums over the global batch: its three sums are all-reduced between the forward and the
        # backward pass of every closure (sharding.BatchSums); that collective cannot sit inside a captured graph
        self.batch_sums = sharding.BatchSums() if (self.world > 1 and args.loss == "cosim") else None
        self.unit_input = ownutilities.model_takes_unit_input(args.net)
        self.nw_delta1 = torch.zeros_like(delta_like1).to(device)
        self.nw_delta2 = torch.zeros_like(delta_like2).to(device)
        self.nw_delta1.requires_grad = True
        if args.joint_perturbation:
            self.params = [self.nw_delta1]
        else:
            self.nw_delta2.requires_grad = True
            self.params = [self.nw_delta1, self.nw_delta2]
        self.optimizer = ops.get().LBFGS(self.params, max_iter=10)
        self.reducer = sharding.FlatReducer(self.params) if self.world > 1 else None
        self.use_graph = _graphs_enabled(device, args) if use_graph is None else use_graph
        if self.batch_sums is not None:
            self.use_graph = False
        self.states = {}   # batch shape -> _UniversalBatchState
        self.st = None
        self.flow_pred_init = None
        self.closures = 0

    def deltas(self):
        if self.args.joint_perturbation:
            return self.nw_delta1, self.nw_delta1
        return self.nw_delta1, self.nw_delta2

    def predict(self, perturbed=True):
        st, kw = self.st, {}
        if perturbed:
            kw = {"delta1": self.nw_delta1} if self.args.joint_perturbation else {"delta1": self.nw_delta1,
                                                                                   "delta2": self.nw_delta2}
        out = ownutilities.compute_flow(self.model, "scaled_input_model", st.image1, st.image2, test_mode=True, **kw)
        [out] = ownutilities.postprocess_flow(self.args.net, st.padder, out)
        return out

    def closure_body(self):
        d1, d2 = self.deltas()
        loss_closure = losses.loss_delta_constraint(self.predict(), self.st.target, d1, d2, self.device,
                                                    delta_bound=self.args.delta_bound, mu=self.optim_mu,
                                                    f_type=self.args.loss, batch_sums=self.batch_sums)
        loss_closure.backward()
        if self.reducer is not None:
            self.reducer.pack(loss_closure)
        return loss_closure

    def closure(self):
        self.closures += 1
        if self.st.graphed is not None:
            loss_closure = self.st.graphed()
        else:
            self.optimizer.zero_grad()
            loss_closure = self.closure_body()
        return self.reducer.reduce() if self.reducer is not None else loss_closure

    def begin_batch(self, image1, image2):
        """Upload a batch (this rank's slice), predict the unattacked flow, set the target (attack_PCFA.py:404-452).
        Returns AEE(target, unattacked flow) averaged over ranks."""
        device = self.device
        image1, image2 = image1.to(device), image2.to(device)
        if not self.unit_input:
            image1 = image1 / 255.
            image2 = image2 / 255.
        padder, [image1, image2] = ownutilities.preprocess_img(self.args.net, image1, image2)
        key = tuple(image1.shape)
        st = self.states.pop(key, None)
        if st is None:
            st = _UniversalBatchState(image1, image2, padder)
        else:
            st.image1.copy_(image1)
            st.image2.copy_(image2)
        self.states[key] = st                              # most recently used last
        while len(self.states) > _max_cached_shapes(self.model):   # every state pins a closure graph pool: keep the newest
            old_key = next(iter(self.states))
            self.states.pop(old_key)
            _log_eviction("universal batch-state", old_key)
        self.st = st
        with torch.no_grad():
            self.flow_pred_init = self.predict(perturbed=False).detach().clone()
        target = targets.get_target(self.args.target, self.flow_pred_init,
                                    custom_target_path=self.args.custom_target_path, device=device).to(device)
        if st.target is None:
            st.target = target.clone()
        else:
            st.target.copy_(target)
        self.model.zero_grad()
        self.optimizer.zero_grad()
        if self.use_graph and not st.tried_graph:
            st.tried_graph = True
            try:
                from .graphed import GraphedClosure, GraphedForward
                st.graphed = GraphedClosure(self.closure_body, self.params)
                st.repredict = GraphedForward(self.predict, device)
            except Exception as e:  # noqa: BLE001 -- the eager closure computes the same thing
                import logging as pylog
                pylog.warning("hipGraph capture of the universal closure failed (%r): launching eagerly", e)
                st.graphed = st.repredict = None
        return sharding.mean_scalar(logging.calc_metrics_const(st.target, self.flow_pred_init), device)

    def step(self):
        """One `--steps` iteration on the current batch (attack_PCFA.py:455-517): L-BFGS step (10 closures, each
        followed by the all-reduce), re-prediction, metrics (averaged over ranks)."""
        st = self.st
        self.optimizer.step(self.closure)
        if st.repredict is not None:
            flow_pred = st.repredict()
        else:
            with torch.no_grad():
                flow_pred = self.predict()
        d1, d2 = self.deltas()
        aee_adv_tgt, aee_adv_pred = logging.calc_metrics_adv(flow_pred, st.target, self.flow_pred_init)
        if self.world > 1:
            aee_adv_tgt, aee_adv_pred = sharding.mean_scalars((aee_adv_tgt, aee_adv_pred), self.device)
        l2_delta1, l2_delta2, l2_delta12 = logging.calc_delta_metrics(d1.detach(), d2.detach())
        return {"aee_predadv-tgt": aee_adv_tgt, "aee_pred-predadv": aee_adv_pred, "l2_delta1": l2_delta1,
                "l2_delta2": l2_delta2, "l2_delta-avg": l2_delta12}

    @property
    def graphed(self):
        return any(s_.graphed is not None for s_ in self.states.values())


def attack_l2_universal(args, data_loader=None, has_gt=None):
    """One perturbation for a whole dataset (attack_PCFA.py:297-566), data parallel over the batch: see
    `UniversalAttack`.  Returns the final perturbations, the per-step metric history and how many collectives ran."""
    optim_mu = default_mu(args)
    rank, world = sharding.rank(), sharding.world_size()
    distortion_folder = _output_folder(args, "") if rank == 0 else None
    device = select_device()
    if data_loader is None:
        if args.batch_size % world != 0:
            raise ValueError("--batch_size %d must be divisible by the %d ranks" % (args.batch_size, world))
        data_loader, has_gt = datasets.prepare_dataloader(args, batch_size=args.batch_size // world, shuffle=True,
                                                          shard=(rank, world))
    model = _load_model(args, device, variable_change=False)  # universal = clipping only (attack_PCFA.py:365)

    image1_init, image2_init, _, _ = next(iter(data_loader))
    _, [image1_init, image2_init] = ownutilities.preprocess_img(args.net, image1_init, image2_init)
    ua = UniversalAttack(model, image1_init[0, :, :, :], image2_init[0, :, :, :], device, optim_mu, args)

    history = []
    batches = []
    batch_ctr = -1
    for epoch in range(args.epochs):
        for batch, (image1, image2, flow, _) in enumerate(data_loader):
            batch_ctr += 1
            curr_step = batch_ctr * args.steps
            aee_tgt = ua.begin_batch(image1, image2)
            logging.log_metrics(curr_step, ("aee_pred-tgt", aee_tgt))
            batches.append({"epoch": epoch, "batch": batch, "aee_pred-tgt": aee_tgt})
            for steps in range(args.steps):
                curr_step = batch_ctr * args.steps + steps
                logging.log_metrics(curr_step, ("steps", steps), ("batch", batch), ("epoch", epoch))
                m = ua.step()
                logging.log_metrics(curr_step, *m.items())
                history.append(dict(m, epoch=epoch, batch=batch, step=steps))
            if distortion_folder is not None and _should_save(batch_ctr, args):
                logging.save_tensor(ua.nw_delta1, "delta1_b" + str(batch_ctr), batch_ctr, distortion_folder)
                logging.save_tensor(ua.deltas()[1], "delta2_b" + str(batch_ctr), batch_ctr, distortion_folder)
        if distortion_folder is not None:
            # `NNNNN_delta1_e{E}.npy`: the pattern evaluate_PCFA.py:42-43 looks for
            logging.save_tensor(ua.nw_delta1, "delta1_e" + str(epoch), batch_ctr, distortion_folder)
            if not args.joint_perturbation:
                logging.save_tensor(ua.nw_delta2, "delta2_e" + str(epoch), batch_ctr, distortion_folder)
    return {"delta1": ua.nw_delta1.detach(), "delta2": ua.deltas()[1].detach(), "history": history,
            "batches": batches, "collectives": ua.reducer.collectives if ua.reducer is not None else 0,
            "batch_sum_collectives": ua.batch_sums.collectives if ua.batch_sums is not None else 0,
            "graphed": ua.graphed}


def main(argv=None):
    parser = parsing_file.create_parser(stage='training', attack_type='pcfa')
    args = parser.parse_args(argv)
    print(args)
    sharding.init_from_env()
    try:
        if args.universal_perturbation:
            return attack_l2_universal(args)
        return attack_l2(args)
    finally:
        sharding.shutdown()


if __name__ == '__main__':
    main()
