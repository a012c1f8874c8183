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


class MixedPrecisionEfficientDet(torch.nn.Module):
    """bfloat16 backbone, float32 BiFPN + heads: the point between the bf16 throughput mode and the float32 parity mode.

    The backbone is ~76 % of the MACs and ~78 % of the per-layer bytes and runs on the bf16 kernels; its three
    feature maps (P3 / P4 / P5, 0.56 M values per image at d0 / 640) are widened to float32 while they are copied into the
    float32 engine's input buffers, and BiFPN, class / box heads, the OOD epilogue and everything behind them run in float32
    from float32 master weights.  Built from ONE float32 model (the reference's `create_model(...)` result): the bf16 copy
    keeps only what the backbone needs.  Behaves like an `EfficientDet` for `DetBenchPredict` (`config`, `ood_energy`,
    `ood_max_logit`, `weights_token`, `prepare`); `forward(x)` = reference `mode='full_net'` (effdet/efficientdet.py:895-933).
    """

    def __init__(self, model_f32):
        super().__init__()
        p0 = model_f32.backbone.conv_stem.weight
        if p0.dtype != torch.float32:
            raise ValueError('MixedPrecisionEfficientDet is built from a float32 model')
        self.tail = model_f32                                   # float32: BiFPN + heads (its backbone copy stays unused)
        self.front = copy.deepcopy(model_f32).to(torch.bfloat16)  # bfloat16: backbone
        self.config = model_f32.config
        self.ood_energy = None
        self.ood_max_logit = None

    @property
    def _engine(self):
        return self.tail._engine

    def weights_token(self):
        return (self.front.weights_token(), self.tail.weights_token())

    def prepare(self, batch_size, image_size=None, ood_out=None):
        self.front.prepare(batch_size, image_size)
        return self.tail.prepare(batch_size, image_size, ood_out=ood_out)

    def __copy__(self):
        # shallow copies share the parameters and get their own engines (what DetBenchPredict / PipelinedPredict rely on)
        new = MixedPrecisionEfficientDet.__new__(MixedPrecisionEfficientDet)
        torch.nn.Module.__init__(new)
        new.tail, new.front, new.config = copy.copy(self.tail), copy.copy(self.front), self.config
        new.ood_energy = new.ood_max_logit = None
        return new

    def forward(self, x, mode='full_net'):
        if mode != 'full_net':
            raise ValueError('the mixed-precision wrapper runs mode=\'full_net\' only')
        with torch.no_grad():
            feats = self.front(x if x.dtype == torch.uint8 else x.to(torch.bfloat16), mode='bb')
            # engine_for() keeps an engine prepared for this batch / size / weights (split batches pass ood_out to prepare)
            cls_o, box_o = self.tail(feats, mode='fpn_and_head')
        self.ood_energy, self.ood_max_logit = self.tail.ood_energy, self.tail.ood_max_logit
        return cls_o, box_o
