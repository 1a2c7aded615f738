"""every tensor's gradient error in creation order, statistics pivot on / off (one model, the parity test's seed)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import diag_grad_parity as G  # noqa: E402
from scanobjectnn_amd import fused_mlp  # noqa: E402
for piv in (True, False):
    fused_mlp.STAT_PIVOT = piv
    print("==== pivot", piv, flush=True)
    print(G.run(sys.argv[1] if len(sys.argv) > 1 else "dgcnn", 0, "all"), flush=True)
