"""A prepared message-passing step recorded into a HIP graph and replayed (launch-bound batches: the reference driver's
batches of 16 molecules, test_lipo.py:150, are ~150 launches of a few microseconds each).

What a recorded region may not do, and where this package does it otherwise:
  * read a device value on the host (`.item()`, `bool(tensor)`): MolGraph's lazy properties do (unit-weight check, plan
    sizes) -- `MolGraph.prepare()` evaluates all of them up front;
  * call hipFuncSetAttribute: the kernels that opt in to more than 64 KB of LDS do so in a once-per-process static on their
    first launch -- the warm-up steps below run every kernel family of the step before recording starts;
  * allocate with hipMalloc: every buffer of ops.* is a torch tensor, and torch's allocator serves a recording from a
    private pool;
  * touch the legacy default stream: autograd's AccumulateGrad node remembers the stream it was created on, and a node stays
    alive for as long as any autograd graph of an earlier step does.  `EdgeNetwork.edge_embed` (the reference's cache of the
    bond matrices across the T steps, edge_network.py:39) holds such a graph between steps, so a model that has ever run on
    the default stream carries default-stream AccumulateGrad nodes into the recording, and the implicit synchronisation with
    the default stream inside a capture aborts the process (the round-2/3 core dumps: gpurun_out/gcap.log, b_c1_g.err).
    `drop_cached_autograd_state` clears those caches BEFORE the warm-up, and the warm-up runs on the recording's stream.
"""
import torch


def drop_cached_autograd_state(model):
    """Forget every tensor a module caches across calls that hangs on an autograd graph (EdgeNetwork.edge_embed)."""
    for m in model.modules():
        if getattr(m, "edge_embed", None) is not None:
            m.edge_embed = None


class CapturedStep:
    """`step()` -- a callable without arguments that reads and writes fixed tensors (static inputs, a GradientBucket's flat
    gradient buffer) -- warmed up `warmup` times on a side stream, recorded once, then `replay()`ed.  `result` is what
    `step` returned during the recording: tensors of the recording's memory pool, rewritten by every replay."""

    def __init__(self, step, model=None, warmup=3):
        if not torch.cuda.is_available():
            raise RuntimeError("CapturedStep needs a GPU")
        if model is not None:
            drop_cached_autograd_state(model)
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                step()
                if model is not None:
                    drop_cached_autograd_state(model)
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.result = step()
        if model is not None:
            drop_cached_autograd_state(model)          # the recorded step's graph: nothing eager should hang on it

    def replay(self):
        self.graph.replay()
        return self.result


def capture_training_step(model, afm, graph, mask, seed, bucket, warmup=3):
    """The training step of bench.py / examples/train_lipo.py on one resident batch: zero the flat gradient buffer,
    message passing forward, backward from `seed` (d loss / d state).  Returns a CapturedStep whose `result` is the final
    node state; the gradients land in `bucket.flat`."""
    graph.prepare(tile_plan=(afm.shape[-1] == 64), wide_plan=(afm.shape[-1] in (128, 256)))

    def step():
        bucket.zero()
        state, _ = model.message_passing(afm, graph, graph, mask)
        state.backward(gradient=seed.view_as(state))
        return state.detach()

    return CapturedStep(step, model, warmup)
