"""L-BFGS for the PCFA attack loop with the vector math on hand-written HIP kernels.

Drop-in for ``torch.optim.LBFGS(params, max_iter=10)`` as the reference uses it (attack_PCFA.py:97,114,382,388:
fixed step, no line search): same constructor, same ``step(closure)`` contract, same iteration logic -- every
decision the optimiser takes on the host (curvature test ``y.s > 1e-10``, first-step length, the four stopping
rules, closure-evaluation schedule) is taken from the same quantities in the same order.  What changes is where
the vector work runs:

* the history is two ring buffers ``S, Y [history_size][n]`` instead of Python lists of tensors;
* ``y = g - g_prev``, ``s = t*d``, ``y.s``, ``y.y`` and the ``g_prev`` refresh are ONE pass (``pcfa_lbfgs_pair``);
* the two-loop recursion runs as ``2m+1`` fused launches without a host round trip (``pcfa_lbfgs_direction``)
  instead of ``4m+3`` separately launched vector kernels -- at ``m = 100`` pairs that is the difference between
  ~7 ms and ~2 ms per iteration next to a 21 ms closure (tools/step_breakdown.py).

Arithmetic: the same sequence of fp32 operations; only the summation order inside a dot product differs (block
partials summed in index order instead of rocBLAS' order).  Checked against torch.optim.LBFGS on the GPU in
tests/test_gpu_parity.py.  There is no CPU path: CPU tensors raise (tests drive the host logic of the attack with
the oracle's optimiser, which is torch.optim.LBFGS itself).
"""
import torch
from torch.optim import Optimizer

from . import _hip


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _call(name, *args):
    _hip.check(getattr(_hip.load(), name)(*args, _stream()), name)


class LBFGS(Optimizer):
    def __init__(self, params, lr=1, max_iter=20, max_eval=None, tolerance_grad=1e-7, tolerance_change=1e-9,
                 history_size=100, line_search_fn=None):
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

    # ---- device buffers --------------------------------------------------------------------------------------
    def _buffers(self):
        if self._bufs is None:
            dev = self._params[0].device
            cap = self.param_groups[0]["history_size"]
            new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
            ws = int(_hip.load().pcfa_lbfgs_workspace_floats())
            self._bufs = {
                "g": new(self._ld), "g_prev": new(self._ld), "d": new(self._ld),
                # history_size + 1 rows: the candidate pair of an iteration is written before the optimiser knows
                # whether it keeps it, so it needs a row that is not one of the (up to history_size) live pairs
                "S": new(cap + 1, self._ld), "Y": new(cap + 1, self._ld),
                "ro": torch.zeros(cap + 1, dtype=torch.float32, device=dev), "al": new(cap + 1),
                "H": torch.ones(1, dtype=torch.float32, device=dev), "scal": new(4), "ws": new(ws),
            }
        return self._bufs

    def _gather_flat_grad(self, g):
        off = 0
        for p in self._params:
            n = p.numel()
            if p.grad is None:
                g[off:off + n].zero_()
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

    # ---- torch.optim.LBFGS.step, fixed-step branch ---------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure):
        closure = torch.enable_grad()(closure)
        group = self.param_groups[0]
        lr, max_iter, max_eval = float(group["lr"]), group["max_iter"], group["max_eval"]
        tolerance_grad, tolerance_change = group["tolerance_grad"], group["tolerance_change"]
        tol_grad32 = float(torch.tensor(tolerance_grad, dtype=torch.float32))
        cap = group["history_size"]
        B = self._buffers()
        n = self._n
        p = lambda t: t.data_ptr()  # noqa: E731

        state = self.state[self._params[0]]
        state.setdefault("func_evals", 0)
        state.setdefault("n_iter", 0)
        state.setdefault("first", 0)   # ring start (row of the oldest pair)
        state.setdefault("count", 0)   # pairs in the ring
        state.setdefault("has_prev", False)

        orig_loss = closure()
        loss = float(orig_loss)
        current_evals = 1
        state["func_evals"] += 1
        flat_grad = self._gather_flat_grad(B["g"])
        opt_cond = bool(flat_grad.abs().max() <= tolerance_grad)
        if opt_cond:
            return orig_loss

        d = B["d"][:n]
        t = state.get("t")
        prev_loss = state.get("prev_loss")
        n_iter = 0
        while n_iter < max_iter:
            n_iter += 1
            state["n_iter"] += 1

            # ---- direction -----------------------------------------------------------------------------------
            if state["n_iter"] == 1:
                torch.neg(flat_grad, out=d)
                state["first"], state["count"] = 0, 0
                B["H"].fill_(1.0)
                B["g_prev"][:n].copy_(flat_grad)
            else:
                # candidate pair goes into the row after the newest one; it only counts if y.s > 1e-10
                first, count = state["first"], state["count"]
                rows = cap + 1
                row = (first + count) % rows
                _call("pcfa_lbfgs_pair", p(B["g"]), p(B["g_prev"]), p(B["d"]), float(t), p(B["Y"][row]),
                      p(B["S"][row]), p(B["scal"]), p(B["ws"]), 1, n)
                ys = float(B["scal"][0])  # the optimiser's own host decision (one synchronisation, as in torch)
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

            # ---- step length ---------------------------------------------------------------------------------
            if state["n_iter"] == 1:
                t = min(1.0, float(1.0 / flat_grad.abs().sum())) * lr  # fp32 reciprocal, as the tensor op in torch
            else:
                t = lr
            # two host decisions, one synchronisation each (torch reads every scalar on its own): the directional
            # derivative with max|d| (d does not change until the next direction), and below the loss with max|g|
            gtd, d_absmax = torch.stack((flat_grad.dot(d), torch.linalg.vector_norm(d, float("inf")))).tolist()
            if gtd > -tolerance_change:
                break

            self._add_grad(t, d)
            ls_func_evals = 0
            if n_iter != max_iter:
                loss_t = closure()
                flat_grad = self._gather_flat_grad(B["g"])
                loss_t = torch.as_tensor(loss_t, dtype=flat_grad.dtype, device=flat_grad.device).detach().reshape(())
                loss, g_absmax = torch.stack((loss_t, torch.linalg.vector_norm(flat_grad, float("inf")))).tolist()   # max|g|: one pass
                opt_cond = g_absmax <= tol_grad32   # the tensor comparison of torch rounds the tolerance to fp32
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
            if d_absmax * abs(t) <= tolerance_change:
                break
            if abs(loss - prev_loss) < tolerance_change:
                break

        state["t"] = t
        state["prev_loss"] = prev_loss
        return orig_loss
