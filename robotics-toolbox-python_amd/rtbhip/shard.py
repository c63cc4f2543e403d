"""Multi-GPU batch sharding: every configuration / target / (q,qd,qdd) triple is
independent, so the batch is split into contiguous row blocks, one per rank (one process per GPU),
with the chain tables replicated.  There is NO collective on the data path; the only exchange is
one optional gather of the output shards (RCCL over xGMI when the backend is "nccl", gloo on CPU).
"""
import os

from ._lib import shard_range


class Communicator:
    """An RCCL communicator behind the C ABI (rtbhip_shard_comm_*; include/rtbhip.h): what a consumer WITHOUT PyTorch would hold.  One process
    per GPU: rank 0 makes the 128-byte id (`Communicator.new_id()`), ships it to the others, every rank constructs `Communicator(id, world,
    rank)` with its GPU current.  `from_process_group()` does the shipping over an initialised torch.distributed group (any backend) -- the
    group only carries the id; the gather itself is RCCL through librtbhip.so."""

    def __init__(self, id128, world, rank):
        import ctypes as C
        from ._lib import lib, check
        self.world, self.rank = int(world), int(rank)
        self._id = bytes(id128)
        if len(self._id) != 128:
            raise ValueError("the communicator id is 128 bytes (rtbhip_shard_comm_id)")
        h = C.c_void_p()
        check(lib().rtbhip_shard_comm_create(C.c_char_p(self._id), self.world, self.rank, C.byref(h)))
        self._h = h

    @staticmethod
    def new_id():
        import ctypes as C
        from ._lib import lib, check
        buf = C.create_string_buffer(128)
        check(lib().rtbhip_shard_comm_id(buf))
        return buf.raw

    @classmethod
    def from_process_group(cls, group=None):
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.new_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        return cls(box[0], world, rank)

    def info(self):
        """(world, rank, RCCL version) as the communicator itself reports them (ncclCommCount / ncclCommUserRank / ncclGetVersion)."""
        import ctypes as C
        from ._lib import lib, check
        w, r, v = C.c_int32(-1), C.c_int32(-1), C.c_int32(0)
        check(lib().rtbhip_shard_comm_info(self._h, C.byref(w), C.byref(r), C.byref(v)))
        return w.value, r.value, v.value

    def gather(self, local_out, N, root=0, out=None):
        """rtbhip_shard_gather of this rank's rows (a CUDA tensor whose first dimension is its rtbhip_shard_range count of N) on the current
        stream: the (N, ...) tensor on `root` (every rank for root = -1), None elsewhere.  `out` = a receive buffer to reuse."""
        import ctypes as C
        import torch
        from ._lib import lib, check, current_stream_ptr
        if not local_out.is_cuda:
            raise ValueError("Communicator.gather moves device buffers (RCCL)")
        send = local_out.contiguous()
        tail = tuple(send.shape[1:])
        row_bytes = send.element_size()
        for d in tail:
            row_bytes *= int(d)
        receives = root < 0 or root == self.rank
        if receives and out is None:
            out = torch.empty((int(N),) + tail, dtype=send.dtype, device=send.device)
        if receives and (tuple(out.shape) != (int(N),) + tail or not out.is_contiguous() or out.dtype != send.dtype or out.device != send.device):
            raise ValueError("out must be a contiguous (N, ...) tensor like the shards")
        check(lib().rtbhip_shard_gather(self._h, C.c_void_p(send.data_ptr()), int(send.shape[0]), row_bytes, int(N), self.world, self.rank, int(root),
                                        C.c_void_p(out.data_ptr()) if receives else None, current_stream_ptr()))
        return out if receives else None

    def destroy(self):
        from ._lib import lib, check
        if getattr(self, "_h", None) is not None and self._h.value:
            check(lib().rtbhip_shard_comm_destroy(self._h))
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


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

    def gather_rccl(self, comm, local_out, root=0, out=None):
        """The gather through the C ABI (rtbhip_shard_gather on the Communicator `comm`): one ncclGather to `root` (default; root = -1: one
        ncclAllGather) of the device shards, ragged shards as one group of sends / receives straight into place -- no padding, no torch
        collective.  Returns the (N, ...) tensor on the receiving rank(s), None elsewhere."""
        if (comm.world, comm.rank) != (self.world, self.rank):
            raise ValueError("the communicator is rank %d of %d, the batch rank %d of %d" % (comm.rank, comm.world, self.rank, self.world))
        if int(local_out.shape[0]) != self.count:
            raise ValueError("rank %d holds %d rows of this batch, not %d" % (self.rank, self.count, int(local_out.shape[0])))
        return comm.gather(local_out, self.N, root=root, out=out)

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
