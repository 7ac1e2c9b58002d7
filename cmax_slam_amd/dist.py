"""Sharding one CMax evaluation across the GPUs of a node.

Events are independent and the IWE is a SUM over events, but the contrast is a non-linear function of the
summed image, so the one exchange step sits between splat and blur/reduce:

    rank r:  accumulate(x)            splat its contiguous range of 100-event batches into partial planes
    all:     all_reduce(sum, fp32)    RCCL over xGMI, in place on the accumulation planes
    rank r:  finish()                 blur + moment reduction on the summed planes (replicated, deterministic)
  with the adjoint gradient (CMX_GRAD_ADJOINT) the planes are I only and finish() splits once more:
    rank r:  finish_begin()           image pass, Itilde, gather over the rank's OWN events -> partial gradient sums
    all:     all_reduce(sum, fp64)    3 (front end) or 3*K_opt doubles
    rank r:  finish_end()             contrast + gradient to the host

Batches are kept whole (the per-batch pose time depends on a batch's first and last event), so the sharded
result has the same batch boundaries as the single-GPU one.  One process per GPU; torch.distributed supplies
the communicator (backend "nccl" == RCCL on ROCm, "gloo" on CPU for the tests).
"""
import numpy as np


def batch_range(n_events, batch_size, rank, world):
    """Contiguous event range [beg, end) of whole batches owned by `rank` (last rank takes the ragged tail)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    nb = (n_events + batch_size - 1) // batch_size
    per = (nb + world - 1) // world
    b0 = min(rank * per, nb)
    b1 = min(b0 + per, nb)
    return min(b0 * batch_size, n_events), min(b1 * batch_size, n_events)


class ShardedEvaluator:
    """Wraps a split-phase evaluator with the exchange steps.

    `ev` provides accumulate(x, want_grad), accum_count(), finish(want_grad) and -- for the adjoint gradient --
    finish_begin(want_grad), grad_count(), finish_end(want_grad).  `accum` is a torch tensor aliasing the
    evaluator's accumulation planes (on the GPU: the tensor whose data_ptr was handed to cmx_set_accum_buffer, so
    RCCL reduces the planes in place, no staging copy); `gsum` likewise aliases the per-rank partial gradient sums
    (float64) that the adjoint mode exchanges after the gather pass."""

    def __init__(self, ev, accum, gsum=None, group=None, force_collectives=False):
        import torch.distributed as dist
        self.ev, self.accum, self.gsum, self.group, self.dist = ev, accum, gsum, group, dist
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.collect = self.world > 1 or (force_collectives and dist.is_initialized())

    def eval(self, x, want_grad=True):
        d = self.dist
        self.ev.accumulate(x, want_grad)
        if self.collect:
            d.all_reduce(self.accum[:self.ev.accum_count()], op=d.ReduceOp.SUM, group=self.group)
        if self.gsum is None:
            return self.ev.finish(want_grad)
        self.ev.finish_begin(want_grad)
        n = self.ev.grad_count()  # 0 unless this evaluation produced per-rank partial gradient sums
        if self.collect and n > 0:
            d.all_reduce(self.gsum[:n], op=d.ReduceOp.SUM, group=self.group)
        return self.ev.finish_end(want_grad)


def attach_torch_accum(ev, device):
    """Allocate the accumulation planes as a torch tensor on `device`, hand them to the evaluator and make it
    run on a torch-owned side stream, so kernels and RCCL collectives are ordered on one stream.
    Returns (accum, gsum, stream); run evaluations inside `with torch.cuda.stream(stream):`."""
    import torch
    n = ev.accum_capacity()
    accum = torch.zeros(n, dtype=torch.float32, device=device)
    gsum = torch.zeros(256, dtype=torch.float64, device=device)
    stream = torch.cuda.Stream(device=device)
    torch.cuda.synchronize(device)
    ev.set_accum_buffer(accum.data_ptr(), n)
    ev.set_grad_buffer(gsum.data_ptr(), gsum.numel())
    ev.set_stream(stream.cuda_stream)
    return accum, gsum, stream
