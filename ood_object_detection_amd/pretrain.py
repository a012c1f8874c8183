"""One iteration of the reference's pretrain loop (pretrain.py:220-276) on the HIP path.

    meta_optimizer.zero_grad()
    qry_imgs = (qry_imgs.float() - imagenet_mean) / imagenet_std          pretrain.py:226   (uint8 input: effdet_normalize_u8)
    feats = model(qry_imgs, mode='bb')                                    :229
    class_out, box_out = model(feats, mode='fpn_and_head')                :232
    qry_loss, ... = loss_fn(class_out, box_out, cls_anchors, bbox_anchors, num_positives)   :233
    qry_loss.backward()                                                   :236
    clip_grad_norm_(model.parameters(), 10.); meta_optimizer.step()       :272-276

Data-parallel training (BASELINE config 5) adds ONE collective: the flat float32 gradient buffer of `FlatAdam`
(15.6 MB for d0) is all-reduced (mean) over RCCL between backward and the optimizer step - one large message, the
shape a point-to-point xGMI fabric wants.  BN stays per replica (no SyncBN in the reference).  Defaults mirror the
script's flags: Adam lr 1e-3, alpha .15, box weight 50, delta .1 (pretrain.py:57-62), backbone BN frozen in eval mode
(`freeze_bb_bn`, :168-176), BiFPN / head BN in batch-statistics mode.
"""
import torch
import torch.nn as nn

from .optim import FlatAdam


def set_bn_eval(module):
    """pretrain.py:169-171"""
    if isinstance(module, nn.modules.batchnorm._BatchNorm):
        module.eval()


class PretrainStep(object):
    def __init__(self, model, lr=1e-3, max_grad_norm=10.0, alpha=0.15, box_loss_weight=50.0, freeze_bn=False,
                 labeler=True, graph=False, graph_warmup=2):
        """graph=True: after `graph_warmup` eager iterations the whole iteration (zero_grad, forward, loss, backward and -
        single GPU - clip + Adam) is captured into one hipGraph and replayed; the ~5 000 launches of a step then cost one
        submission.  Shapes must stay fixed; labels are assigned eagerly and copied into the graph's static buffers."""
        from .effdet.anchors import Anchors, AnchorLabeler
        from .effdet.config import set_config_writeable
        from .effdet.loss import DetectionLoss
        self.model = model
        model.train()
        if freeze_bn:
            model.apply(set_bn_eval)
        else:
            model.backbone.apply(set_bn_eval)
        cfg = model.config
        set_config_writeable(cfg)
        cfg.alpha, cfg.box_loss_weight = alpha, box_loss_weight
        self.loss_fn = DetectionLoss(cfg)
        self.opt = FlatAdam(model.parameters(), lr=lr, max_grad_norm=max_grad_norm)
        self.anchors = Anchors.from_config(cfg).to(model.backbone.conv_stem.weight.device)
        self.labeler = AnchorLabeler(self.anchors, cfg.num_classes, match_threshold=0.5) if labeler else None
        self.num_levels = cfg.num_levels
        self.world = 1
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            self.world = dist.get_world_size()
        self.last_allreduce_ms = 0.0
        self.graph = bool(graph)
        # the stage tables upload at the end of the first backward and at the start of the second forward (host-to-device copies,
        # which cannot be captured): at least two eager iterations before the capture
        self._graph_warmup = max(2, int(graph_warmup))
        self._calls = 0
        self._cap = None                        # (graph, optimizer graph or None, static inputs, static outputs)
        self._static_base = None

    def targets(self, target):
        """target: {'bbox': [per-image [M,4] yxyx], 'cls': [per-image [M]]} (labelled on the GPU, effdet/anchors.py:384-438)
        or the loader's pre-labelled {'label_cls_l', 'label_bbox_l', 'label_num_positives'} (effdet/bench.py:127-133)."""
        if 'label_num_positives' in target:
            return ([target['label_cls_%d' % l] for l in range(self.num_levels)],
                    [target['label_bbox_%d' % l] for l in range(self.num_levels)], target['label_num_positives'])
        if self.labeler is None:
            raise ValueError('pre-labelled targets or labeler=True needed')
        return self.labeler.batch_label_anchors(target['bbox'], target['cls'])

    def detections(self, class_out, box_out):
        """pretrain.py:238-245: _post_process + generate_detections(hard NMS, no clipping) for the whole batch.
        -> det [B, 100, 6] (x1,y1,x2,y2,score,class 1-based, zero padded), count [B]"""
        from .effdet.anchors import batched_detections
        from .effdet.bench import _post_process
        cfg = self.model.config
        with torch.no_grad():
            co, bo = [t.detach() for t in class_out], [t.detach() for t in box_out]
            ct, bt, idx, cl = _post_process(co, bo, cfg.num_levels, cfg.num_classes, cfg.max_detection_points)
            B, k = idx.shape
            det, count, _ = batched_detections(ct.reshape(B, k), bt, self.anchors.boxes, idx, cl, None, None,
                                               max_det_per_image=cfg.max_det_per_image, soft_nms=False)
        return det, count

    def evaluate(self, class_out, box_out, target, evaluator):
        """pretrain.py:246-252: add the batch's detections and ground truth (target['bbox'] yxyx / target['cls'] 1-based lists)
        to a device `ObjectDetectionEvaluator` (cleared by the caller per iteration, as the script does)."""
        det, count = self.detections(class_out, box_out)
        B = det.shape[0]
        M = max(1, max(int(b.shape[0]) for b in target['bbox']))
        gtb = torch.zeros(B, M, 4, dtype=torch.float32, device=det.device)
        gtc = torch.full((B, M), -1, dtype=torch.int64, device=det.device)
        for i, (b, c) in enumerate(zip(target['bbox'], target['cls'])):
            m = int(b.shape[0])
            if m:
                gtb[i, :m] = b.to(det.device, torch.float32)
                gtc[i, :m] = c.to(det.device, torch.int64)
        evaluator.add_batch(det, count, gtb, gtc)
        return det, count

    # ---- captured iteration ---------------------------------------------------------------------------------
    def _capture(self, x, cls_t, box_t, npos):
        model, opt = self.model, self.opt
        from .effdet.loss import _pack_targets
        sx = x.clone()
        # static targets: ONE packed tensor each when they are the labeler's views (one copy per replay instead of one per level)
        cb, bb = _pack_targets(list(cls_t), 0), _pack_targets(list(box_t), 4)
        if self.labeler is not None and cb is cls_t[0]._base and bb is box_t[0]._base:
            self._static_base = (cb.clone(), bb.clone())
            s_cls, s_box = self.labeler._unpack(*self._static_base)
        else:
            self._static_base = None
            s_cls = [t.clone() for t in cls_t]
            s_box = [t.clone() for t in box_t]
        s_np = npos.clone()
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            opt.zero_grad()
            feats = model(sx, mode='bb')
            class_out, box_out = model(feats, mode='fpn_and_head')
            loss, class_loss, box_loss = self.loss_fn(class_out, box_out, s_cls, s_box, s_np)
            loss.backward()
            norm = opt.step_captured() if self.world == 1 else None
            outs = [loss.detach(), class_loss, box_loss, norm]
        g2 = None
        if self.world > 1:
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=g1.pool()):
                outs[3] = opt.step_captured()
        del feats, class_out, box_out, loss
        self._cap = (g1, g2, (sx, s_cls, s_box, s_np), outs)

    def _replay(self, x, cls_t, box_t, npos, time_allreduce):
        g1, g2, (sx, s_cls, s_box, s_np), outs = self._cap
        if tuple(x.shape) != tuple(sx.shape) or x.dtype != sx.dtype:
            raise ValueError('graph mode: input shape / dtype changed (%s %s, captured %s %s)' % (tuple(x.shape), x.dtype, tuple(sx.shape), sx.dtype))
        sx.copy_(x)
        base = self._static_base
        from .effdet.loss import _pack_targets
        # _pack_targets returns the common base only when the level tensors are the labeler's in-order views of it
        views = base is not None and cls_t[0]._base is not None and box_t[0]._base is not None
        cb = _pack_targets(list(cls_t), 0) if views else None
        bb = _pack_targets(list(box_t), 4) if views else None
        if base is not None and cb is not None and bb is not None and cb is cls_t[0]._base and bb is box_t[0]._base and \
                cb.shape == base[0].shape and bb.shape == base[1].shape:
            base[0].copy_(cb)
            base[1].copy_(bb)
        else:
            for d, t in zip(s_cls, cls_t):
                d.copy_(t)
            for d, t in zip(s_box, box_t):
                d.copy_(t)
        s_np.copy_(npos)
        self.opt.advance()
        g1.replay()
        if g2 is not None:
            self._allreduce(time_allreduce)
            g2.replay()
        self.model.invalidate()
        return {'loss': outs[0].clone(), 'class_loss': outs[1].clone(), 'box_loss': outs[2].clone(),
                'grad_norm': None if outs[3] is None else outs[3].clone()}

    def _allreduce(self, time_allreduce):
        import torch.distributed as dist
        opt = self.opt
        if time_allreduce:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        dist.all_reduce(opt.flat_grad, op=dist.ReduceOp.SUM)
        opt.flat_grad.div_(self.world)
        if time_allreduce:
            e1.record()
            e1.synchronize()
            self.last_allreduce_ms = e0.elapsed_time(e1)

    def _engine_flags(self):
        eng = self.model._train_engine
        if eng is not None:
            eng.direct_grad = True          # FlatAdam owns every .grad (views of one flat buffer): add into them directly

    def __call__(self, x, target, time_allreduce=False, evaluator=None):
        model, opt = self.model, self.opt
        self._calls += 1
        if model._train_engine is None or model._train_engine.signature != model.train_signature():
            from .train_engine import TrainEngine
            model._train_engine = TrainEngine(model)
            if self._cap is not None:
                # The captured kernels hold raw addresses of the OLD engine's persistent tensors (folded weights, stage tables,
                # gradient buffers), which return to the allocator with it: replaying the graph would read freed memory.  Drop
                # the graph and its static buffers, run the eager warm-up again on the new engine, then re-capture.
                self._cap = None
                self._static_base = None
                self._calls = 1
        self._engine_flags()
        if self.graph and evaluator is None and self._calls > self._graph_warmup:
            cls_t, box_t, npos = self.targets(target)
            if self._cap is None:
                self._capture(x, cls_t, box_t, npos)
            return self._replay(x, cls_t, box_t, npos, time_allreduce)
        opt.zero_grad()
        cls_t, box_t, npos = self.targets(target)
        feats = model(x, mode='bb')
        class_out, box_out = model(feats, mode='fpn_and_head')
        loss, class_loss, box_loss = self.loss_fn(class_out, box_out, cls_t, box_t, npos)
        loss.backward()
        if evaluator is not None:
            self.evaluate(class_out, box_out, target, evaluator)
        if self.world > 1:
            self._allreduce(time_allreduce)
        norm = opt.step()
        model.invalidate()                      # the inference engine's packed weights are stale now
        return {'loss': loss.detach(), 'class_loss': class_loss, 'box_loss': box_loss, 'grad_norm': norm}
