"""Reading back the DISCRETE DECISIONS a product model took in one forward pass -- which pre-activations passed a
ReLU, which neighbourhood member a max-pool took, whether that member was active -- so that the float64 truth of
oracle/ref_models.py can be evaluated WITH THOSE SAME DECISIONS (`ref_models.imposing`).

Why: the model-level gradient error of an fp32 path against float64 is dominated by the
handful of decisions that sit within fp32 rounding of a tie (a ReLU at 1e-7, two pool members 5e-7 apart); they move
whole gradient rows by O(1) and differ from path to path and seed to seed.  Round 3 absorbed them into the tolerance
through "the distance between two realisations of the fused path" -- a bar a path sets for itself.  Here every path
(fused kernels, layer-by-layer library path) is compared with the truth evaluated on ITS OWN decisions, the decisions
that differ from the float64 run's are counted and each is shown to be a rounding-level tie, and the remaining
(purely arithmetic) error of the fused path is bounded by the plain path's.

Two sources, merged in forward order into one event list:
  * fused stacks (FusedMLPStack / EdgeConvPool): `fused_mlp.TRACE` receives the autograd node; the masks are rebuilt
    exactly from what the node keeps for its backward (raw layer outputs + BN scale/shift: the kernels decide with ONE
    fmaf(y, scale, shift) > 0 whose sign float64 reproduces, tests/mlp_ref.py; 8-bit arg-max rows; the arithmetic
    first layer from its centred offsets);
  * torch-level layers (the FC heads; every layer of the layer-by-layer path): `torch.relu` and `Tensor.amax` are
    wrapped for the duration of the forward pass; the scope is the variable scope the activation is applied in.
"""
import contextlib

import torch

import mlp_ref as MR
from oracle import ref_models as R


def _scope_of(name):
    assert name.endswith("/bn/gamma"), name
    return name[:-len("/bn/gamma")]


class Recorder:
    def __init__(self, net, device="cpu"):
        self.device = device          # where the Decisions tensors are put (the device the truth is evaluated on)
        self.names = {p.data_ptr(): n[len("graph."):] for n, p in net.named_parameters()}
        self.events = []

    @contextlib.contextmanager
    def recording(self):
        from scanobjectnn_amd import fused_mlp
        from scanobjectnn_amd.graph import get_default_graph
        real_relu, real_amax = torch.relu, torch.Tensor.amax
        ev = self.events
        del ev[:]

        def relu(x):
            out = real_relu(x)
            ev.append(("relu", get_default_graph().full_name("")[:-1], out.detach()))
            return out

        def amax(self_, *a, **k):
            out = real_amax(self_, *a, **k)
            ev.append(("amax", self_.detach(), k["dim"] if "dim" in k else a[0], out.detach()))
            return out

        torch.relu, torch.Tensor.amax, fused_mlp.TRACE = relu, amax, ev
        try:
            yield self
        finally:
            torch.relu, torch.Tensor.amax, fused_mlp.TRACE = real_relu, real_amax, None

    def decisions(self):
        """-> ref_models.Decisions on self.device, keyed by the reference's scope names; clears the event list (it pins
        every activation of the pass)"""
        D = R.Decisions()
        last = None
        for e in self.events:
            if isinstance(e, tuple) and e[0] == "relu":
                _, scope, out = e
                D.relu[scope] = (out > 0).reshape(-1, out.shape[-1]).to(self.device)
                last = ("plain", scope)
            elif isinstance(e, tuple) and e[0] == "amax":
                _, inp, dim, out = e
                dims = [d % inp.dim() for d in ((dim,) if isinstance(dim, int) else tuple(dim))]
                dims = [d for d in dims if inp.shape[d] > 1]
                if not dims or last is None:
                    continue                                  # a max over one element decides nothing
                assert len(dims) == 1, (inp.shape, dim)
                d, c = dims[0], inp.shape[-1]
                arg = inp.argmax(dim=d)                       # the first maximal member, as the kernels take it
                if last[0] == "fusedpool":
                    # a max over ALL points of a cloud taken inside the fused stack over chunks of S points (8-bit
                    # arg-max) and finished here over the chunk maxima (dgcnn/tf_util.conv2d_stack_global_max)
                    _, scope, S, arg_in, act_in = last
                    nb, nch = inp.shape[0], inp.shape[d]
                    ch = arg.reshape(nb, c)
                    grp = torch.arange(nb, device=ch.device).view(nb, 1) * nch + ch
                    D.pool[scope] = ((ch * S + torch.gather(arg_in.view(nb * nch, c), 0, grp)).to(self.device),
                                     torch.gather(act_in.view(nb * nch, c), 0, grp).to(self.device))
                else:
                    scope = last[1]
                    D.pool[scope] = (arg.reshape(-1, c).to(self.device), (out.reshape(-1, c) > 0).to(self.device))
                    D.relu.pop(scope, None)                   # pooled layer: the pool decision stands for its ReLU
                last = None
            elif len(e.saved) == 11:                          # EdgeConvPool node: one pooled layer
                _Q, _Ctr, _idx, gamma, _SQ, arg, ysel, _mean, _rstd, scale, shift = e.saved
                c = arg.shape[1]
                active = (ysel.double() * scale[:c].double() + shift[:c].double()) > 0
                D.pool[_scope_of(self.names[gamma.data_ptr()])] = (arg.long().to(self.device), active.to(self.device))
                last = None
            else:                                             # FusedMLPStack node
                masks, argmax = MR.node_pattern(e, virtual_first_layer=True)
                gammas, scales, shifts, ysel = e.saved[13], e.saved[10], e.saved[11], e.saved[15]
                S, pool, L = e.meta[:3]
                for l, g in enumerate(gammas):
                    scope = _scope_of(self.names[g.data_ptr()])
                    if pool and l == L - 1:
                        c = ysel.shape[1]
                        active = (ysel.double() * scales[l][:c].double() + shifts[l][:c].double()) > 0
                        D.pool[scope] = (argmax.long().to(self.device), active.to(self.device))
                        last = ("fusedpool", scope, S, argmax.long(), active)
                    else:
                        assert masks[l] is not None, scope
                        D.relu[scope] = masks[l].to(self.device)
                        last = ("plain", scope)
        del self.events[:]
        return D


def summarise(report, tie):
    """totals of a ref_models report (imposed decisions against the float64 run's own) and the check that every
    decision that differs is a rounding-level tie: the flipped ReLU's float64 pre-activation and the gap between the
    imposed pool member and the float64 maximum are both <= tie"""
    out = {"relu_elements": 0, "relu_flips": 0, "pool_elements": 0, "pool_flips": 0, "active_flips": 0,
           "worst_abs_z": 0.0, "worst_gap": 0.0, "layers_with_flips": []}
    for scope, r in report.items():
        if r["kind"] == "relu":
            out["relu_elements"] += r["elements"]
            out["relu_flips"] += r["flips"]
        else:
            out["pool_elements"] += r["elements"]
            out["pool_flips"] += r["flips"]
            out["active_flips"] += r["active_flips"]
            out["worst_gap"] = max(out["worst_gap"], r["worst_gap"])
        out["worst_abs_z"] = max(out["worst_abs_z"], r["worst_abs_z"])
        if r["flips"] or r.get("active_flips"):
            out["layers_with_flips"].append(scope)
    out["all_ties"] = bool(out["worst_abs_z"] <= tie and out["worst_gap"] <= tie)
    return out
