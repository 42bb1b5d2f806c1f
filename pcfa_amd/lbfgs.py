"""L-BFGS for the PCFA attack loop with the vector math on hand-written HIP kernels.

Drop-in for ``torch.optim.LBFGS(params, max_iter=10)`` as the reference uses it (attack_PCFA.py:97,114,382,388:
fixed step, no line search): same constructor, same ``step(closure)`` contract, same iteration logic -- every
decision the optimiser takes on the host (curvature test ``y.s > 1e-10``, first-step length, the four stopping
rules, closure-evaluation schedule) is taken from the same quantities in the same order.  What changes is where
the vector work runs:

* the history is two ring buffers ``S, Y [history_size][n]`` instead of Python lists of tensors;
* ``y = g - g_prev``, ``s = t*d``, ``y.s``, ``y.y`` and the ``g_prev`` refresh are ONE pass (``pcfa_lbfgs_pair``);
* ``direction="gram"`` (default, history_size <= 128; pcfa_amd/csrc/lbfgs_gram.hip): the two-loop recursion is linear
  algebra in span{g, s_i, y_i}, so it runs as two triangular substitutions (fp64, one workgroup) on the inner products
  s_i.y_j, y_i.y_j, s_i.g, y_i.g.  An iteration sweeps the history exactly twice -- once to form the inner products of
  the new gradient and the new pair with every stored vector (``pcfa_lbfgs_gram_update``, which also takes the
  curvature decision and commits the pair ON THE DEVICE), once to form d (``pcfa_lbfgs_gram_direction``): 4.3 GB of
  HBM traffic per iteration at the steady-state history of 100 pairs instead of 8.6 GB, two host synchronisations per
  iteration instead of three.  Not the optimiser's rounding order: iterates agree with torch.optim.LBFGS to the
  tolerance stated in tests/test_gpu_parity.py::test_lbfgs_matches_torch_optimizer.
* ``direction="two_loop"`` (r01/r02 path, kept for A/B and for history_size > 128): the recursion as ``2m+1`` fused
  launches without a host round trip (``pcfa_lbfgs_direction``) instead of ``4m+3`` separately launched vector kernels;
  the same sequence of fp32 operations as the optimiser, only the summation order inside a dot product differs.

There is no CPU path: CPU tensors raise (tests drive the host logic of the attack with the oracle's optimiser, which is
torch.optim.LBFGS itself).
"""
import numpy as np
import torch
from torch.optim import Optimizer

from . import _hip


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _call(name, *args):
    _hip.check(getattr(_hip.load(), name)(*args, _stream()), name)


class LBFGS(Optimizer):
    def __init__(self, params, lr=1, max_iter=20, max_eval=None, tolerance_grad=1e-7, tolerance_change=1e-9,
                 history_size=100, line_search_fn=None, direction=None):
        if line_search_fn is not None:
            raise NotImplementedError("pcfa_amd.lbfgs.LBFGS implements the fixed-step variant the PCFA attack uses "
                                      "(line_search_fn=None)")
        if max_eval is None:
            max_eval = max_iter * 5 // 4
        defaults = dict(lr=lr, max_iter=max_iter, max_eval=max_eval, tolerance_grad=tolerance_grad,
                        tolerance_change=tolerance_change, history_size=history_size, line_search_fn=None)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError("LBFGS doesn't support per-parameter options (parameter groups)")
        self._params = self.param_groups[0]["params"]
        for p in self._params:
            if not p.is_cuda or p.dtype != torch.float32:
                raise RuntimeError("pcfa_amd.lbfgs.LBFGS needs float32 GPU parameters (no CPU fallback)")
        self._n = sum(p.numel() for p in self._params)
        self._ld = (self._n + 3) // 4 * 4
        self._bufs = None
        if direction is None:
            direction = "gram" if history_size <= 128 else "two_loop"
        if direction not in ("gram", "two_loop") or (direction == "gram" and history_size > 128):
            raise ValueError("direction must be 'gram' (history_size <= 128) or 'two_loop'")
        self.direction = direction
        self.host_syncs = 0          # host reads of device scalars (each one drains the stream)

    # ---- device buffers --------------------------------------------------------------------------------------
    def _buffers(self):
        if self._bufs is None:
            dev = self._params[0].device
            cap = self.param_groups[0]["history_size"]
            new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
            zeros = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)  # noqa: E731
            lib = _hip.load()
            ws = int(lib.pcfa_lbfgs_workspace_floats())
            if self.direction == "gram":
                ws = max(ws, int(lib.pcfa_lbfgs_gram_workspace_bytes(cap, self._ld)) // 4)
            self._bufs = {
                # the ld - n pad elements stay zero: the Gram sweeps run over whole 16-B groups
                "g": zeros(self._ld), "g_prev": zeros(self._ld), "d": zeros(self._ld),
                # history_size + 1 rows: the candidate pair of an iteration is written before the optimiser knows
                # whether it keeps it, so it needs a row that is not one of the (up to history_size) live pairs
                "S": new(cap + 1, self._ld), "Y": new(cap + 1, self._ld),
                "ro": torch.zeros(cap + 1, dtype=torch.float32, device=dev), "al": new(cap + 1),
                "H": torch.ones(1, dtype=torch.float32, device=dev), "scal": new(4), "ws": new(ws),
            }
            if self.direction == "gram":
                self._bufs["state"] = torch.zeros(int(lib.pcfa_lbfgs_gram_state_bytes(cap)), dtype=torch.uint8,
                                                  device=dev)
                self._bufs["out2"] = new(2)
        return self._bufs

    def reset(self):
        """Forget the history and the iteration state (a new image pair on the same variables and buffers)."""
        self.state.clear()
        self.host_syncs = 0

    def flat_grad_views(self):
        """One view per parameter into the optimiser's flat gradient buffer.  A closure that leaves `p.grad` pointing
        at these views (graphed.GraphedClosure(grad_sink=...)) spares the optimiser its gather copies."""
        g, off, views = self._buffers()["g"], 0, []
        for p in self._params:
            n = p.numel()
            views.append(g[off:off + n].view_as(p))
            off += n
        return views

    def _gather_flat_grad(self, g):
        off = 0
        for p in self._params:
            n = p.numel()
            if p.grad is None:
                g[off:off + n].zero_()
            elif p.grad.data_ptr() == g.data_ptr() + 4 * off and p.grad.is_contiguous() and not p.grad.is_sparse:
                pass                                   # already in place (captured closure with grad_sink)
            else:
                g[off:off + n].copy_((p.grad.to_dense() if p.grad.is_sparse else p.grad).reshape(-1))
            off += n
        return g[:self._n]

    def _add_grad(self, step_size, update):
        off = 0
        for p in self._params:
            n = p.numel()
            p.add_(update[off:off + n].view_as(p), alpha=step_size)
            off += n

    def _read(self, *tensors):
        """One host synchronisation for several device scalars (0-d / 1-element tensors or 1-d packs)."""
        self.host_syncs += 1
        return torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors]).tolist()

    def history_count(self):
        """Pairs currently in the history (torch: len(state['old_dirs'])); a host read in the Gram form."""
        state = self.state[self._params[0]]
        if self.direction == "gram" and self._bufs is not None and state.get("n_iter", 0) > 0:
            return int(self._bufs["state"][:16].view(torch.int32)[1])
        return state.get("count", 0)

    # ---- torch.optim.LBFGS.step, fixed-step branch ---------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure):
        closure = torch.enable_grad()(closure)
        group = self.param_groups[0]
        lr, max_iter, max_eval = float(group["lr"]), group["max_iter"], group["max_eval"]
        tolerance_grad, tolerance_change = group["tolerance_grad"], group["tolerance_change"]
        # torch compares fp32 tensors with these python floats: the scalar is rounded to fp32 first
        tol_grad32 = float(np.float32(tolerance_grad))
        tol_change32 = float(np.float32(tolerance_change))
        cap = group["history_size"]
        gram = self.direction == "gram"
        B = self._buffers()
        n = self._n
        p = lambda t: t.data_ptr()  # noqa: E731

        state = self.state[self._params[0]]
        state.setdefault("func_evals", 0)
        state.setdefault("n_iter", 0)
        state.setdefault("first", 0)   # ring start (row of the oldest pair)        } two_loop only: the Gram form
        state.setdefault("count", 0)   # pairs in the ring                          } keeps both on the device
        state.setdefault("has_prev", False)

        orig_loss = closure()
        if torch.is_tensor(orig_loss):
            # a captured closure hands back a static buffer that later evaluations overwrite: return the value of
            # THIS evaluation, as torch does
            orig_loss = orig_loss.detach().clone()
        current_evals = 1
        state["func_evals"] += 1
        flat_grad = self._gather_flat_grad(B["g"])
        loss, g_absmax = self._read(torch.as_tensor(orig_loss, dtype=torch.float32, device=flat_grad.device),
                                    torch.linalg.vector_norm(flat_grad, float("inf")))
        if g_absmax <= tol_grad32:
            return orig_loss

        d = B["d"][:n]
        t = state.get("t")
        prev_loss = state.get("prev_loss")
        n_iter = 0
        while n_iter < max_iter:
            n_iter += 1
            state["n_iter"] += 1
            first_ever = state["n_iter"] == 1

            # ---- direction -----------------------------------------------------------------------------------
            scalars = None
            if first_ever:
                torch.neg(flat_grad, out=d)
                state["first"], state["count"] = 0, 0
                B["H"].fill_(1.0)
                B["g_prev"][:n].copy_(flat_grad)
                if gram:
                    _call("pcfa_lbfgs_gram_reset", p(B["state"]), cap)
            elif gram:
                # sweep 1: new pair, inner products, curvature decision + commit, coefficients (all on the device);
                # sweep 2: d, g.d and max|d|
                _call("pcfa_lbfgs_gram_update", p(B["g"]), p(B["g_prev"]), p(B["d"]), float(t), p(B["S"]),
                      p(B["Y"]), p(B["state"]), p(B["ws"]), cap, self._ld)
                _call("pcfa_lbfgs_gram_direction", p(B["g"]), p(B["S"]), p(B["Y"]), p(B["state"]), p(B["d"]),
                      p(B["out2"]), p(B["ws"]), cap, self._ld)
                scalars = B["out2"]
            else:
                # candidate pair goes into the row after the newest one; it only counts if y.s > 1e-10
                first, count = state["first"], state["count"]
                rows = cap + 1
                row = (first + count) % rows
                _call("pcfa_lbfgs_pair", p(B["g"]), p(B["g_prev"]), p(B["d"]), float(t), p(B["Y"][row]),
                      p(B["S"][row]), p(B["scal"]), p(B["ws"]), 1, n)
                [ys] = self._read(B["scal"][0])  # the optimiser's own host decision (one synchronisation, as in torch)
                if ys > 1e-10:
                    if count == cap:      # limited memory: forget the oldest pair
                        first = (first + 1) % rows
                    else:
                        count += 1
                    B["ro"][row:row + 1].copy_(B["scal"][2:3])
                    B["H"].copy_(B["scal"][3:4])
                    state["first"], state["count"] = first, count
                if count > 0:
                    _call("pcfa_lbfgs_direction", p(B["g"]), p(B["S"]), p(B["Y"]), p(B["ro"]), p(B["H"]), p(B["al"]),
                          p(B["d"]), p(B["ws"]), first, count, rows, self._ld, n)
                else:
                    torch.mul(flat_grad, B["H"], out=d).neg_()
            prev_loss = loss

            # ---- step length + the directional derivative: ONE host synchronisation ----------------------------
            if scalars is None:
                scalars = torch.stack((flat_grad.dot(d), torch.linalg.vector_norm(d, float("inf"))))
            if first_ever:
                gtd, d_absmax, g_abssum = self._read(scalars, flat_grad.abs().sum())
                t = min(1.0, float(np.float32(1.0) / np.float32(g_abssum))) * lr   # fp32 reciprocal, as torch's
            else:
                gtd, d_absmax = self._read(scalars)
                t = lr
            if gtd > -tolerance_change:
                break

            self._add_grad(t, d)
            ls_func_evals = 0
            if n_iter != max_iter:
                loss_t = closure()
                flat_grad = self._gather_flat_grad(B["g"])
                loss_t = torch.as_tensor(loss_t, dtype=flat_grad.dtype, device=flat_grad.device).detach().reshape(())
                loss, g_absmax = self._read(loss_t, torch.linalg.vector_norm(flat_grad, float("inf")))
                opt_cond = g_absmax <= tol_grad32
                ls_func_evals = 1
            current_evals += ls_func_evals
            state["func_evals"] += ls_func_evals

            # ---- stopping rules ------------------------------------------------------------------------------
            if n_iter == max_iter:
                break
            if current_evals >= max_eval:
                break
            if opt_cond:
                break
            # torch: d.mul(t).abs().max() <= tolerance_change, an fp32 product against the fp32-rounded tolerance
            if float(np.float32(d_absmax) * np.float32(abs(t))) <= tol_change32:
                break
            if abs(loss - prev_loss) < tolerance_change:
                break

        state["t"] = t
        state["prev_loss"] = prev_loss
        return orig_loss
