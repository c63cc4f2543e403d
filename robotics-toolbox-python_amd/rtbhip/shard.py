"""Multi-GPU batch sharding: every configuration / target / (q,qd,qdd) triple is
independent, so the batch is split into contiguous row blocks, one per rank (one process per GPU),
with the chain tables replicated.  There is NO collective on the data path; the only exchange is
one optional gather of the output shards (RCCL over xGMI when the backend is "nccl", gloo on CPU).
"""
import os

from ._lib import shard_range


class ShardedBatch:
    def __init__(self, N, rank=None, world=None):
        if rank is None or world is None:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                rank, world = dist.get_rank(), dist.get_world_size()
            else:
                rank = int(os.environ.get("RANK", "0"))
                world = int(os.environ.get("WORLD_SIZE", "1"))
        self.N, self.rank, self.world = int(N), int(rank), int(world)
        self.begin, self.count = shard_range(self.N, self.rank, self.world)
        self.max_count = max(shard_range(self.N, r, self.world)[1] for r in range(self.world))

    def local(self, x):
        """This rank's rows of a global (N, ...) array."""
        return x[self.begin:self.begin + self.count]

    def ik_rows(self):
        """Context manager for this rank's IK calls: `with sb.ik_rows(): ets.ik_LM(sb.local(Tep))` draws, for every local row,
        the restart vectors one call over all N targets would draw (rtbhip_ik_target_base(begin)) -- the gathered solutions are
        the single-GPU ones, however the rows were split."""
        from ._lib import ik_target_base
        return ik_target_base(self.begin)

    def gather(self, local_out, to_all=False, dst=0, collective="auto"):
        """ONE collective: all_gather_into_tensor (to_all) or gather-to-dst of equal-size padded
        shards.  Returns the (N, ...) result on the receiving rank(s), None elsewhere.  A world of one rank needs no exchange
        and gets its own array back -- unless collective="always", which runs the collective regardless (a world-size-1 RCCL
        group exercises the communicator and its kernels on a single-GPU box)."""
        import torch
        import torch.distributed as dist
        if collective not in ("auto", "always"):
            raise ValueError("collective must be 'auto' or 'always'")
        if not (dist.is_available() and dist.is_initialized()):
            if collective == "always":
                raise RuntimeError("gather(collective='always') needs an initialised process group")
            return local_out
        if self.world == 1 and collective == "auto":
            return local_out
        tail = tuple(local_out.shape[1:])
        home = local_out.device
        if local_out.is_cuda and dist.get_backend() == "gloo":
            local_out = local_out.cpu()        # gloo moves host memory; RCCL ("nccl") takes the device buffer as it is
        send = local_out
        if self.count != self.max_count:
            send = torch.zeros((self.max_count,) + tail, dtype=local_out.dtype, device=local_out.device)
            send[:self.count] = local_out
        send = send.contiguous()
        if to_all:
            buf = torch.empty((self.world * self.max_count,) + tail, dtype=send.dtype, device=send.device)
            dist.all_gather_into_tensor(buf, send)
            parts = buf.reshape((self.world, self.max_count) + tail)
        else:
            lst = [torch.empty_like(send) for _ in range(self.world)] if self.rank == dst else None
            dist.gather(send, lst, dst=dst)
            if self.rank != dst:
                return None
            parts = torch.stack(lst)
        if self.N == self.world * self.max_count:
            return parts.reshape((self.N,) + tail).to(home)
        rows = [parts[r, :shard_range(self.N, r, self.world)[1]] for r in range(self.world)]
        return torch.cat(rows, dim=0).to(home)
