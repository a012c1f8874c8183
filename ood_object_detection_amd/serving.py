"""Several batches in flight over one set of weights.

The kernels of one `DetBenchPredict.forward` leave the chip partly idle (about half of the wave cycles wait on memory or
LDS, and the BiFPN / SE / top-k / NMS launches are far smaller than 256 CUs).  A serving loop hides that by keeping a few
requests in flight; `PipelinedPredict` does it with `in_flight` engine instances - own activation buffers and launch plans
on shallow copies of the model, shared parameters - each on its own stream:

    pipe = PipelinedPredict(model, in_flight=3)
    tickets = [pipe.submit(x) for x in batches]            # returns at once; work is queued on the slot's stream
    for t in tickets:
        det, count, ood = pipe.result(t)                   # waits for that batch only

Results are bit-identical to `DetBenchPredict(model, streams=1)(x)`.  With `graphs=True` every slot replays one captured
hipGraph (fixed batch shape; the ~100 launches of a forward become one submission) - the arrangement `bench.py` measures
(12.7 k img/s at d0 / 640 / batch 64 against 11.5 k with one batch at a time).
"""
import copy

import torch

from .effdet.bench import DetBenchPredict


class PipelinedPredict(object):
    def __init__(self, model, in_flight=3, sub_batches=1, graphs=False):
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
        self.graphs = bool(graphs)
        self._cap = [None] * in_flight          # (graph, static input, static outputs) per slot

    def submit(self, x, img_info=None):
        """Queue one batch; returns a ticket for `result`.  The slot's previous batch must have been collected."""
        k = self._n % len(self.slots)
        if self._ticket_of[k] is not None:
            raise RuntimeError('slot %d still holds the result of ticket %d: call result() first (at most %d batches in flight)'
                               % (k, self._ticket_of[k], len(self.slots)))
        s = self.streams[k]
        cur = torch.cuda.current_stream(self.device)
        bench = self.slots[k]
        if self.graphs:
            if img_info is not None:
                raise NotImplementedError('graphs=True replays a fixed launch list: pass img_info=None (scale / clip the rows afterwards)')
            cap = self._cap[k]
            if cap is None or tuple(cap[1].shape) != tuple(x.shape) or cap[1].dtype != x.dtype:
                xin = x.clone()
                s.wait_stream(cur)
                with torch.cuda.stream(s), torch.no_grad():
                    bench(xin)                                            # plans, buffers and workspaces exist before the capture
                cur.wait_stream(s)
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.no_grad(), torch.cuda.graph(g, stream=s):
                    det = bench(xin)
                    outs = (det, bench.last_count, {k_: v for k_, v in bench.last_ood.items()})
                cap = self._cap[k] = (g, xin, outs)
        s.wait_stream(cur)                                                # x was produced on the caller's stream
        with torch.cuda.stream(s), torch.no_grad():
            x.record_stream(s)
            if self.graphs:
                g, xin, outs = self._cap[k]
                xin.copy_(x, non_blocking=True)
                g.replay()
                # the graph's output tensors are overwritten by the slot's next replay: hand out copies of the small ones
                self._out[k] = (outs[0].clone(), outs[1].clone(),
                                {'energy': outs[2]['energy'].clone(), 'max_logit': outs[2]['max_logit'].clone(),
                                 'anchor_energy': outs[2]['anchor_energy'], 'anchor_max_logit': outs[2]['anchor_max_logit']})
            else:
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
