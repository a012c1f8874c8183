"""Optimizer half of the reference's pretrain step (pretrain.py:272-276): `clip_grad_norm_(params, 10.)` followed by
`torch.optim.Adam(lr=1e-3).step()`, as two HIP launches over ONE flat float32 buffer.

`FlatAdam` re-points the parameters (and their `.grad`s) at slices of flat buffers, so that
* the gradient exchange of data-parallel training is a single RCCL all-reduce of `flat_grad` (15.6 MB for d0 - the
  few-large-messages shape a point-to-point xGMI fabric wants; `sharding.allreduce_gradients` handles the general case),
* the squared-norm reduction and the fused clip + Adam update touch every byte exactly once.
Parameters must be float32 GPU tensors (the pretrain step keeps fp32 master weights)."""
import torch

from . import _lib


class FlatAdam(object):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=10.0):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev = self.params[0].device
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32 or dev.type != 'cuda':
                raise RuntimeError('FlatAdam needs float32 parameters on one GPU (no CPU fallback)')
        self.lib = _lib.load()
        self.lr, self.betas, self.eps, self.max_grad_norm = float(lr), (float(betas[0]), float(betas[1])), float(eps), max_grad_norm
        # every parameter starts on a 64-byte boundary of the flat buffers (16-byte vector loads of the kernels that read
        # the weights in place); the padding stays zero: zero gradient, zero moments, zero update
        n = sum(self._padded(p.numel()) for p in self.params)
        self.flat_param = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat_param[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat_param[off:off + k].view(p.shape)          # parameters now alias the flat buffer
                p.grad = self.flat_grad[off:off + k].view(p.shape)           # autograd accumulates in place
                off += self._padded(k)
        self._ws = torch.empty(int(self.lib.effdet_sqnorm_workspace_floats()), dtype=torch.float32, device=dev)
        self._sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._bc = torch.ones(2, dtype=torch.float32, device=dev)      # device copy of Adam's bias corrections (captured steps)
        self.steps = 0

    @staticmethod
    def _padded(k):
        return (k + 15) // 16 * 16

    def zero_grad(self):
        self.flat_grad.zero_()

    def grad_norm(self):
        """Total gradient L2 norm (what clip_grad_norm_ returns), as a 0-d GPU tensor."""
        st = torch.cuda.current_stream(self.flat_grad.device).cuda_stream
        _lib.check(self.lib.effdet_sqnorm(st, self.flat_grad.data_ptr(), self.flat_grad.numel(), self._ws.data_ptr(),
                                          self._sq.data_ptr(), 0), 'effdet_sqnorm')
        return self._sq.sqrt()[0]

    def step(self):
        """clip_grad_norm_(max_grad_norm) + Adam; returns the pre-clip gradient norm (0-d GPU tensor) or None."""
        st = torch.cuda.current_stream(self.flat_grad.device).cuda_stream
        norm = None
        if self.max_grad_norm is not None:
            norm = self.grad_norm()
        self.steps += 1
        _lib.check(self.lib.effdet_adam_clip_step(
            st, self.flat_param.data_ptr(), self.flat_grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
            self.flat_param.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self.steps,
            float(self.max_grad_norm or 0.0), self._sq.data_ptr() if self.max_grad_norm is not None else None), 'effdet_adam_clip_step')
        return norm

    # ---- hipGraph-friendly variant: nothing the launches depend on lives in host scalars that change per step ----------
    def advance(self):
        """Eagerly, before replaying a captured `step_captured`: count the step and refresh the device bias corrections."""
        import math
        import struct
        self.steps += 1
        # the same arithmetic as effdet_adam_clip_step's host side: the betas arrive there as C floats, powers in double
        b1, b2 = (struct.unpack('f', struct.pack('f', b))[0] for b in self.betas)
        bc = torch.tensor([1.0 - b1 ** self.steps, math.sqrt(1.0 - b2 ** self.steps)], dtype=torch.float32)
        self._bc.copy_(bc)

    def step_captured(self):
        """The launches of `step()` with the bias corrections read from device memory (capturable; call `advance()` first)."""
        st = torch.cuda.current_stream(self.flat_grad.device).cuda_stream
        norm = None
        if self.max_grad_norm is not None:
            norm = self.grad_norm()
        _lib.check(self.lib.effdet_adam_clip_step_dev(
            st, self.flat_param.data_ptr(), self.flat_grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
            self.flat_param.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self._bc.data_ptr(),
            float(self.max_grad_norm or 0.0), self._sq.data_ptr() if self.max_grad_norm is not None else None), 'effdet_adam_clip_step_dev')
        return norm
