"""Fresh batches without stalling the step: batch t + 1 is uploaded and indexed while step t runs.

The reference's loop sees a new batch every step (`for batch in DataLoader(collate_fn=collate_2d_graphs)`,
test_lipo.py:157-165); per NEW batch this package has work the resident-batch step time does not show: the host -> device
copy of the compact batch, the CSR-derived index arrays (type order, transposed graph, destination list) and the tile plan
of the fused message + sum kernel -- ~25 small device launches with three host reads between them (`graph.py`), 7-11 ms
when done in line with 11-214 ms steps.

`BatchStream` does that work for batch t + 1 on a SIDE stream, driven by a worker thread, while the caller's stream runs
step t: the host reads of the plan build then block the worker, not the training loop, and the small indexing kernels fill
what the step's kernels leave free.  The hand-over is one event per batch.  Memory: a batch's tensors belong to the side
stream's pool, so they must not be released while the caller's stream still reads them -- `done_with(batch)` hands the batch
back together with an event recorded behind its step, and the worker drops it only after the side stream waits for that
event (whatever the side stream allocates next is then ordered behind the step).  A pipeline one batch deep, two resident.
"""
import queue
import threading

import torch

from .graph import MolGraph


class StreamedBatch:
    def __init__(self, feats, graph, mask, ready):
        self.feats, self.graph, self.mask, self.ready = feats, graph, mask, ready


class BatchStream:
    """Iterate over `source` (an iterable of host batches) as device batches prepared one ahead.

    make_device_batch(host_batch, device) -> (feats, MolGraph, mask) runs on the worker thread under the side stream; the
    default takes a `synth.MolBatch` (features uploaded from `atom_feat`, or made on the device from `atom_keys` when the
    host batch carries them instead).  `hidden` selects the plan to build (64: tile plan, 128 / 256: wide plan).
    """

    def __init__(self, source, device, hidden, make_device_batch=None):
        self.device = torch.device(device)
        self.hidden = int(hidden)
        self.source = iter(source)
        self.make = make_device_batch or self._default_make
        # HIGH priority: the indexing work is ~60 launches of a few microseconds each; behind the step's chip-filling kernels
        # at equal priority each of them waited for a whole kernel of the step (measured: 25.7 ms per step against 11.8
        # resident); ahead of them they take the workgroup slots that free up all the time
        self.side = torch.cuda.Stream(device=self.device, priority=-1)
        self.q = queue.Queue(maxsize=1)
        self.free = queue.Queue()                      # events: "the caller's stream is done with the batch before last"
        self.err = None
        self.thread = threading.Thread(target=self._work, daemon=True)
        self.thread.start()

    def _default_make(self, hb, device):
        from . import synth
        g = MolGraph.from_molbatch(hb, device)
        g.prepare(tile_plan=(self.hidden == 64), wide_plan=(self.hidden in (128, 256)))
        if getattr(hb, "atom_feat", None) is not None and hb.atom_feat.shape[1] == self.hidden:
            src = torch.from_numpy(hb.atom_feat)
            feats = src.to(device, non_blocking=src.is_pinned())
        else:                                          # structure-only batch: features are a function of the atom index
            feats = synth.hashed_features(torch.arange(g.num_nodes, device=device), self.hidden)
        mask = torch.ones(g.num_nodes, 1, device=device)
        return feats, g, mask

    def _work(self):
        try:
            torch.cuda.set_device(self.device)
            n = 0
            for hb in self.source:
                if n >= 2:                             # the batch before last must be out of use before its memory is reused
                    ev, old = self.free.get()
                    self.side.wait_event(ev)
                    del old                            # (released here, behind the wait: see the module docstring)
                with torch.cuda.stream(self.side):
                    feats, g, mask = self.make(hb, self.device)
                    ready = torch.cuda.Event()
                    ready.record(self.side)
                self.q.put(StreamedBatch(feats, g, mask, ready))
                n += 1
        except BaseException as e:                     # surfaces in the consumer
            self.err = e
        finally:
            self.q.put(None)

    def __iter__(self):
        return self

    def __next__(self):
        b = self.q.get()
        if b is None:
            if self.err is not None:
                raise self.err
            raise StopIteration
        torch.cuda.current_stream(self.device).wait_event(b.ready)
        return b

    def done_with(self, batch):
        """Call after the step that used `batch` has been enqueued on the caller's stream."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.free.put((ev, batch))
