"""Fused gradient clipping + Adam over the model's flat parameter buffer.

Replaces `torch.nn.utils.clip_grad_norm_(model.parameters(), 0.25); optimizer.step()`
(reference steps/train_qsub.py:121-122, Adam(lr) defaults from :95) with three launches over one
contiguous buffer: sum of squares -> norm / clip coefficient (kept on the device) -> Adam update.
Optimizer state is not part of the reference's checkpoints (only model.state_dict() is saved,
steps/train_qsub.py:105,150,155); state_dict()/load_state_dict() here are an addition.
"""
import torch

from . import ops


class ClipAdam:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=0.25):
        self.model = model
        self.lr, self.betas, self.eps, self.max_norm = float(lr), betas, float(eps), float(max_norm)
        self.step_count = 0
        self._bound = None
        self.m = self.v = self.scal = None
        self._seen_host = self._seen_ev = None      # skipped_nowait(): last skipped-step count copied to the host
        self._seen = 0

    def _state(self):
        p, g = self.model.flat_parameters()
        if self._bound is None or self._bound.data_ptr() != p.data_ptr():
            self.m = torch.zeros_like(p)
            self.v = torch.zeros_like(p)
            self.scal = torch.zeros(4, device=p.device)     # [norm, clip coef, skip this step, steps skipped so far]
            self._bound = p
        return p, g

    def step(self):
        """Returns the device tensor [total_norm, clip_coef, skipped, skipped_total] (no host sync).  A step whose
        gradients are flagged untrusted by the engine's guard word (a timed-out recurrence launch on any rank)
        changes neither the parameters nor the moments; skipped() / check() report it."""
        p, g = self._state()
        self.step_count += 1
        eng = getattr(self.model, "_engine", None)
        ops.grad_norm(g, self.max_norm, self.scal, guard=eng.guard if eng is not None else None)
        ops.clip_adam(p, g, self.m, self.v, self.scal, self.lr, self.betas[0], self.betas[1], self.eps, self.step_count)
        if eng is not None:
            eng.version += 1            # bf16 operand copies of the weights made before this update are stale now
        # the skipped-step counter also travels to a pinned host word behind every step (4 bytes, asynchronous): a driver
        # can notice a skipped step one step later without ever synchronising (skipped_nowait)
        if self._seen_ev is None or self._seen_ev.query():
            if self._seen_ev is not None:
                self._seen = int(self._seen_host[0])
            if self._seen_host is None:
                self._seen_host = torch.zeros(1, dtype=torch.float32).pin_memory()
            self._seen_host.copy_(self.scal[3:4], non_blocking=True)
            self._seen_ev = torch.cuda.Event()
            self._seen_ev.record()
        return self.scal

    def skipped_nowait(self):
        """The skipped-step count as of the last step whose 4-byte copy has landed (never blocks, may lag a step or two)."""
        if self._seen_ev is not None and self._seen_ev.query():
            self._seen = int(self._seen_host[0])
        return self._seen

    def skipped(self):
        """Number of optimizer steps skipped so far because their gradients were flagged (synchronises)."""
        return int(self.scal[3].item()) if self.scal is not None else 0

    def check(self):
        n = self.skipped()
        if n:
            from ._lib import SepkernError
            raise SepkernError("%d optimizer step(s) were skipped: a persistent BLSTM launch timed out "
                               "(grid not co-resident?); try SEPKERN_LSTM_MODE=2" % n)

    def state_dict(self):
        """'step' is the number of updates actually APPLIED (calls minus skipped calls; the kernel's bias corrections use
        the same number), so a resumed run continues with the corrections an uninterrupted one would use."""
        self._state()
        return {"step": self.step_count - self.skipped(), "m": self.m.clone(), "v": self.v.clone()}

    def load_state_dict(self, sd):
        self._state()
        self.step_count = int(sd["step"])
        self.scal.zero_()
        self._seen, self._seen_ev = 0, None
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
