"""Which torch ops (not this package's kernels) a training step launches, by input shape: the adds / fills / copies that show
in the rocprof tables as `vectorized_elementwise_kernel`.  `python tools/trace_small_ops.py [workload] [scale]`."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mpnn_amd import synth, parallel
from mpnn_amd.graph import MolGraph

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
dev = torch.device("cuda:0")
mols, hidden, T, dist_name, desc = bench.WORKLOADS[name]
mb = synth.make_molecules(max(16, int(mols * scale)), hidden, seed=317, dist=dist_name, atom_features=False)
g = MolGraph.from_molbatch(mb, dev).prepare(tile_plan=(hidden == 64), wide_plan=(hidden in (128, 256)))
afm = synth.hashed_features(torch.arange(g.num_nodes, device=dev), hidden)
mask = torch.ones(g.num_nodes, 1, device=dev)
model = bench.make_model(name, hidden, T, dev)
hot = [p for n, p in model.named_parameters() if not n.startswith("of.")]
bucket = parallel.GradientBucket(hot)
seed = torch.full((g.num_nodes, hidden), 1e-5, device=dev)
def step():                                            # bench.py's training step
    bucket.zero()
    state, _ = model.message_passing(afm, g, g, mask)
    state.backward(gradient=seed.view_as(state))
for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=40,
                                                          max_shapes_column_width=60))
