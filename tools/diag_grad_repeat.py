"""Is the model-level gradient error of the fused path a property of the arithmetic or of WHICH discrete decisions
(ReLU side, arg-max row) a run happened to take?  Same model / seed / inputs as
tests/test_models_parity_gpu.py::test_model_training_gradients, evaluated repeatedly: default (atomic) backward,
deterministic backward, and with the statistics pivot switched off.   python tools/diag_grad_repeat.py [model]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import diag_grad_parity as G  # noqa: E402

if __name__ == "__main__":
    from scanobjectnn_amd import _lib, fused_mlp
    models = sys.argv[1:] or ["dgcnn_bga"]
    for m in models:
        for label, det, piv in (("default", False, True), ("default", False, True), ("default", False, True),
                                ("deterministic", True, True), ("deterministic", True, True),
                                ("no-pivot", False, False), ("no-pivot", False, False)):
            _lib.set_deterministic(det)
            fused_mlp.STAT_PIVOT = piv
            ef, el = G.run(m, 0, False)
            print("%-10s %-14s fused %.3e  plain %.3e" % (m, label, ef, el), flush=True)
        _lib.set_deterministic(False)
        fused_mlp.STAT_PIVOT = True
