import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instancediff_amd import ops, pipeline
from instancediff_amd.utils.synthetic import make_batch
dev = torch.device("cuda:0")
model, sde = pipeline.build(phase="train", device=dev, T=100, seed=0, dist=False)
model.set_train(); sde.set_seed(1)
batch = make_batch(32, 256, seed=1, mixed=True)
model.feed_data(batch); model.optimize_parameters()
ops.ALGO_TRACE = collections.Counter()
model.feed_data(batch); model.optimize_parameters()
torch.cuda.synchronize()
for k, n in sorted(ops.ALGO_TRACE.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    if k[1] == 1: print("algo %d ks %d Cin %4d Cout %4d %4dx%-4d : %d" % (*k, n))
