"""Sharding one CMax evaluation across the GPUs of a node.

Events are independent and the IWE is a SUM over events, but the contrast is a non-linear function of the
summed image, so the one exchange step sits between splat and blur/reduce:

    rank r:  accumulate(x)            splat its contiguous range of 100-event batches into partial planes
    all:     all_reduce(sum, fp32)    RCCL over xGMI, in place on the accumulation planes
    rank r:  finish()                 blur + moment reduction on the summed planes (replicated, deterministic)

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
    """Wraps a split-phase evaluator (accumulate / accum view / finish) with the all-reduce in between.

    `ev` must provide accumulate(x, want_grad), finish(want_grad) and accum_count(); `accum` is a torch tensor
    aliasing the evaluator's accumulation planes (on the GPU: the tensor whose data_ptr was handed to
    cmx_set_accum_buffer, so RCCL reduces the planes in place, no staging copy)."""

    def __init__(self, ev, accum, group=None, grad_is_partial=False):
        import torch.distributed as dist
        self.ev, self.accum, self.group, self.dist = ev, accum, group, dist
        self.grad_is_partial = grad_is_partial  # CMX_GRAD_ADJOINT: finish() returns a per-rank partial gradient
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def eval(self, x, want_grad=True):
        self.ev.accumulate(x, want_grad)
        if self.world > 1:
            n = self.ev.accum_count()
            self.dist.all_reduce(self.accum[:n], op=self.dist.ReduceOp.SUM, group=self.group)
        c, g = self.ev.finish(want_grad)
        if want_grad and self.grad_is_partial and self.world > 1:
            import torch
            t = torch.from_numpy(np.ascontiguousarray(g)).to(self.accum.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            g = t.cpu().numpy()
        return c, g


def attach_torch_accum(ev, device):
    """Allocate the accumulation planes as a torch tensor on `device`, hand them to the evaluator and make it
    run on a torch-owned side stream, so kernels and RCCL collectives are ordered on one stream.
    Returns (accum, stream); run evaluations inside `with torch.cuda.stream(stream):`."""
    import torch
    n = ev.accum_capacity()
    accum = torch.zeros(n, dtype=torch.float32, device=device)
    stream = torch.cuda.Stream(device=device)
    torch.cuda.synchronize(device)
    ev.set_accum_buffer(accum.data_ptr(), n)
    ev.set_stream(stream.cuda_stream)
    return accum, stream
