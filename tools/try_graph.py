import sys, os, time, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from elg_amd.CVRP.CVRPModel import CVRPModel
from elg_amd import engine as eng, _lib as L
cfg = yaml.safe_load(open(os.path.join(ROOT, "elg_amd/CVRP/config.yml")))
dev = "cuda:0"
model = CVRPModel(**cfg["model_params"]); model.decoder.add_local_policy(dev); model.to(dev).train()
class EncFold(torch.nn.Module):
    def __init__(self, m): super().__init__(); self.m = m
    def forward(self, depot, nxd):
        enc = self.m.encoder(depot, nxd)
        d = self.m.decoder
        dec = {"Wq_last.weight": d.Wq_last.weight, "Wk.weight": d.Wk.weight, "Wv.weight": d.Wv.weight,
               "multi_head_combine.weight": d.multi_head_combine.weight, "multi_head_combine.bias": d.multi_head_combine.bias}
        t = eng.fold_decoder_tables(dec, enc, L.PROBLEM_CVRP)
        loc = d.local_policies[0].folded_tables(41)
        return enc, t["K"], t["V"], t["PK"], t["pb"], t["Q1"], t["wl"], loc
ef = EncFold(model)
depot = torch.rand(64, 1, 2, device=dev); nxd = torch.rand(64, 100, 3, device=dev)
def run(f, n=20):
    for _ in range(3):
        outs = f(depot, nxd); sum(o.sum() for o in outs).backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        outs = f(depot, nxd); sum(o.sum() for o in outs).backward()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager fwd+bwd ms", run(ef))
try:
    g = torch.cuda.make_graphed_callables(ef, (depot, nxd))
    print("graphed fwd+bwd ms", run(g))
    o1 = ef(depot, nxd); o2 = g(depot, nxd)
    print("max diff", max(float((a - b).abs().max()) for a, b in zip(o1, o2)))
except Exception as e:
    print("graph capture failed:", type(e).__name__, str(e)[:300])
