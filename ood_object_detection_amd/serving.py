"""Several batches in flight over one set of weights.

The kernels of one `DetBenchPredict.forward` leave the chip partly idle (about half of the wave cycles wait on memory or
LDS, and the BiFPN / SE / top-k / NMS launches are far smaller than 256 CUs).  A serving loop hides that by keeping a few
requests in flight; `PipelinedPredict` does it with `in_flight` engine instances - own activation buffers and launch plans
on shallow copies of the model, shared parameters - each on its own stream:

    pipe = PipelinedPredict(model, in_flight=3)
    tickets = [pipe.submit(x) for x in batches]            # returns at once; work is queued on the slot's stream
    for t in tickets:
        det, count, ood = pipe.result(t)                   # waits for that batch only

Results are bit-identical to `DetBenchPredict(model, streams=1)(x)`.  `bench.py` measures the same arrangement with one
captured hipGraph per slot (12.7 k img/s at d0 / 640 / batch 64 against 11.5 k with one batch at a time).
"""
import copy

import torch

from .effdet.bench import DetBenchPredict


class PipelinedPredict(object):
    def __init__(self, model, in_flight=3, sub_batches=1):
        if in_flight < 1:
            raise ValueError('in_flight must be >= 1')
        p0 = model.backbone.conv_stem.weight
        if p0.device.type != 'cuda':
            raise RuntimeError('PipelinedPredict needs the model on a GPU (no CPU fallback)')
        self.device = p0.device
        self.slots = [DetBenchPredict(model if i == 0 else copy.copy(model), streams=sub_batches).to(self.device) for i in range(in_flight)]
        self.streams = [torch.cuda.Stream(self.device) for _ in range(in_flight)]
        self._done = [None] * in_flight         # event of the batch that occupies the slot
        self._out = [None] * in_flight
        self._ticket_of = [None] * in_flight
        self._n = 0

    def submit(self, x, img_info=None):
        """Queue one batch; returns a ticket for `result`.  The slot's previous batch must have been collected."""
        k = self._n % len(self.slots)
        if self._ticket_of[k] is not None:
            raise RuntimeError('slot %d still holds the result of ticket %d: call result() first (at most %d batches in flight)'
                               % (k, self._ticket_of[k], len(self.slots)))
        s = self.streams[k]
        s.wait_stream(torch.cuda.current_stream(self.device))          # x was produced on the caller's stream
        with torch.cuda.stream(s), torch.no_grad():
            x.record_stream(s)
            bench = self.slots[k]
            det = bench(x, img_info)
            self._out[k] = (det, bench.last_count, bench.last_ood)
            ev = torch.cuda.Event()
            ev.record(s)
        self._done[k] = ev
        self._ticket_of[k] = self._n
        self._n += 1
        return self._n - 1

    def result(self, ticket):
        """(det [B, max_det, 6], count [B], ood dict) of a submitted batch; blocks until that batch has finished."""
        k = ticket % len(self.slots)
        if self._ticket_of[k] != ticket:
            raise KeyError('ticket %d is not in flight' % ticket)
        self._done[k].synchronize()
        out = self._out[k]
        self._out[k] = self._done[k] = self._ticket_of[k] = None
        return out
