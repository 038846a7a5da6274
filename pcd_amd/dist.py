"""Multi-GPU host logic for the MSM: one process per GPU, the (scalar, base) pairs are
sharded by contiguous point range, every rank runs the full single-GPU pipeline on its shard, and the only
exchange step is an all-gather of one Jacobian point per rank (120 B ... 1.7 KB) followed by a local sum.
RCCL has no elliptic-curve reduction, so this is an all-gather + EC-add, never an all-reduce."""
import numpy as np


def shard_range(n, rank, world):
    """contiguous range [lo, hi) of rank's pairs; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_points(partial_xyz, device=None):
    """partial_xyz: numpy uint64 Jacobian point of this rank -> (world, limbs) array, identical on all ranks.
    Uses torch.distributed (backend nccl == RCCL on GPUs, gloo on CPU)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    t = torch.from_numpy(np.ascontiguousarray(partial_xyz).view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return np.stack([o.cpu().numpy().view(np.uint64) for o in out])


def sharded_msm(msm_fn, sum_fn, n, device=None):
    """msm_fn(lo, hi) -> Jacobian partial of pairs [lo, hi); sum_fn(points) -> their sum.  Returns the full MSM
    (the same bytes on every rank, since every rank sums the gathered partials in rank order)."""
    import torch.distributed as dist
    lo, hi = shard_range(n, dist.get_rank(), dist.get_world_size())
    part = msm_fn(lo, hi)
    return sum_fn(all_gather_points(part, device))


class DeviceExchange:
    """The exchange step with no host round trip (GPU ranks): the shard's partial lands in a torch tensor that is the send
    buffer of `all_gather_into_tensor` (RCCL), the receive buffer is summed in place on the device."""

    def __init__(self, ctx, curve, group, device):
        import torch
        import torch.distributed as dist
        from . import capi
        self.ctx, self.curve, self.group = ctx, curve, group
        self.world = dist.get_world_size()
        limbs = 3 * capi.point_limbs(curve, group) // 2
        self.send = torch.zeros(limbs, dtype=torch.int64, device=device)
        self.recv = torch.zeros(self.world * limbs, dtype=torch.int64, device=device)
        # the pipelined form (submit / collect): one send and one receive buffer per ticket of the library (four in flight)
        self.limbs = limbs
        self.sends = torch.zeros(4 * limbs, dtype=torch.int64, device=device)
        self.recvs = [torch.zeros(self.world * limbs, dtype=torch.int64, device=device) for _ in range(4)]
        # the collective of the pipelined form goes on a HIGH-PRIORITY stream: its few workgroups must be dispatched ahead of the thousands
        # of accumulate workgroups of the other MSMs in flight, which would otherwise keep it waiting for a free CU
        self.comm_stream = torch.cuda.Stream(device=device, priority=-1)

    def msm(self, bases, scalars_buf):
        import torch
        import torch.distributed as dist
        ts = torch.cuda.current_stream().cuda_stream      # raw hipStream_t the collective is enqueued on
        self.ctx.msm_partial_to_device(bases, scalars_buf, self.send.data_ptr())
        self.ctx.stream_wait(ts, 1)                       # torch's stream waits for the partial (event, no host wait)
        dist.all_gather_into_tensor(self.recv, self.send)
        self.ctx.stream_wait(ts, 0)                       # the library's stream waits for the gathered points
        return self.ctx.points_sum_device(self.curve, self.group, self.recv.data_ptr(), self.world)

    def submit(self, bases, scalars_buf):
        """enqueue this rank's shard MSM on one of the library's side streams (at most four outstanding) -> ticket; the partial
        lands in slot `ticket` of the send buffers"""
        return self.ctx.msm_submit_partial(bases, scalars_buf, self.sends.data_ptr(), 8 * self.limbs)

    def collect(self, ticket):
        """all-gather of the partials of `ticket` (RCCL, on torch's stream, ordered behind the MSM by an event) + the EC sum"""
        import torch
        import torch.distributed as dist
        ts = self.comm_stream.cuda_stream
        self.ctx.msm_ticket_wait(ticket, ts)              # the collective's stream waits for this slot's MSM (no host wait)
        with torch.cuda.stream(self.comm_stream):
            dist.all_gather_into_tensor(self.recvs[ticket], self.sends[ticket * self.limbs:(ticket + 1) * self.limbs])
        self.ctx.stream_wait(ts, 0)                       # the library's stream (and every later submission) waits for the gather
        return self.ctx.points_sum_device(self.curve, self.group, self.recvs[ticket].data_ptr(), self.world)
