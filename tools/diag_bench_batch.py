import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pytest, torch
import test_models_parity_gpu as T
mp = pytest.MonkeyPatch()
T._no_dropout(mp)
name = sys.argv[1] if len(sys.argv) > 1 else "dgcnn_bga"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
c = T._grad_case(name, 21, mp, batch=batch, num_point=2048, paths=("fused",))
print("em", c["em_fused"], "e", c["e_fused"], "loss", c["loss_fused"], c["loss_ref"])
rel = c["per_variable_fused"]; nrm = c["grad_norm_fused"]
tot = sum(v * v for v in nrm.values()) ** 0.5
def exempt(k):
    return k.endswith("biases") and ((k[:-6] + "bn/gamma") in rel or (k[:-6] + "bn/beta") in rel)
print("exempt residue / total:", sum(nrm[k] ** 2 for k in nrm if exempt(k)) ** 0.5 / tot)
rows = sorted(((rel[k] * nrm[k] / tot, rel[k], nrm[k], k) for k in rel if not exempt(k)), reverse=True)
print("em judged:", sum(r[0] ** 2 for r in rows) ** 0.5)
for contrib, r, n_, k in rows[:15]:
    print("%-40s contrib %.3e  rel %.3e  norm %.3e" % (k, contrib, r, n_))
print(c["flips_fused"])
