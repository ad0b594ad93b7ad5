"""Multi-GPU: independent trials shard across ranks (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The path has no data exchange between trials: every rank recomputes Ks, Kt and their eigendecompositions (3 GF,
deterministic kernels -> bit-identical replicas) and owns a contiguous block of trials.  The only collectives are a
broadcast of the hyper-parameter vector (<= 8+nx doubles) and a sum all-reduce of the partial quadratic term
(1 double, or 1+p with the gradient).  Predictions are per-trial and stay on their rank unless gathered.
"""
import numpy as np


class TrialSharding:
    STAGING_RING = 4            # outstanding asynchronous all-reduces of one message size before the oldest is waited for

    def __init__(self, group=None, gather_predictions=False, device=None, gather_dst=None):
        import torch
        import torch.distributed as td
        if not td.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (launch with torch.distributed.run)")
        self._torch, self._td, self._group = torch, td, group
        self.rank = td.get_rank(group)
        self.world_size = td.get_world_size(group)
        self.gather_predictions = gather_predictions
        self.gather_dst = gather_dst            # rank that receives gathered predictions on the host (None: every rank)
        backend = td.get_backend(group)
        if device is None:
            device = ("cuda:%d" % torch.cuda.current_device()) if backend == "nccl" else "cpu"
        self._device = torch.device(device)

    # contiguous blocks; the first (ntrials % world) ranks get one extra trial
    @staticmethod
    def block(ntrials, rank, world):
        base, extra = divmod(int(ntrials), int(world))
        start = rank * base + min(rank, extra)
        return start, start + base + (1 if rank < extra else 0)

    def local_slice(self, ntrials):
        a, b = self.block(ntrials, self.rank, self.world_size)
        return slice(a, b)

    def _tensor(self, values):
        return self._torch.as_tensor(np.ascontiguousarray(values, dtype=np.float64)).to(self._device)

    def allreduce_sum(self, values):
        t = self._tensor(values)
        self._td.all_reduce(t, op=self._td.ReduceOp.SUM, group=self._group)
        return t.cpu().numpy()

    def allreduce_sum_async(self, values):
        """Start the sum all-reduce and return a callable that waits for it and yields the NumPy result: the caller can
        queue independent GPU work (e.g. predict after loglik) while the collective is in flight.  The message is a few
        doubles, so what it costs is host latency: the staging tensors (pinned host + device) are allocated once per size
        and both copies are non-blocking on a side stream; only the final read waits."""
        values = np.ascontiguousarray(values, dtype=np.float64)
        if self._device.type != "cuda":
            t = self._tensor(values)
            work = self._td.all_reduce(t, op=self._td.ReduceOp.SUM, group=self._group, async_op=True)

            def result_cpu():
                work.wait()
                return t.numpy().copy()
            return result_cpu
        torch = self._torch
        key = values.size
        # a small ring of staging tuples per message size: several collectives of one size may be outstanding (the docstring
        # invites queueing work in between), and a tuple is refilled only after its previous `done` event has completed
        cache = self.__dict__.setdefault("_staging", {})
        ring = cache.setdefault(key, {"slots": [], "next": 0, "stream": torch.cuda.Stream(device=self._device, priority=-1)})
        if len(ring["slots"]) < self.STAGING_RING:
            ring["slots"].append([torch.empty(key, dtype=torch.float64).pin_memory(),
                                  torch.empty(key, dtype=torch.float64, device=self._device), torch.cuda.Event(), None])
            slot = ring["slots"][-1]
        else:
            slot = ring["slots"][ring["next"] % self.STAGING_RING]
            # the oldest collective of this size: its copies have landed before the block is refilled -- and if its result has
            # not been read yet, it is copied out for its owner now (an unread closure must never see a later collective's data)
            slot[2].synchronize()
            if slot[3] is not None and not slot[3]:
                slot[3].append(slot[0].numpy().copy())
        host, dev, done = slot[0], slot[1], slot[2]
        ring["next"] += 1
        stream = ring["stream"]
        host.numpy()[:] = values
        with torch.cuda.stream(stream):
            dev.copy_(host, non_blocking=True)
            self._td.all_reduce(dev, op=self._td.ReduceOp.SUM, group=self._group)      # enqueued on `stream`, returns at once
            host.copy_(dev, non_blocking=True)
            done.record(stream)
        taken = []
        slot[3] = taken                      # the owner of the block's present contents

        def result():
            # the value is copied out at the first read, behind the event -- or by the refill of the block, whichever is first
            if not taken:
                done.synchronize()
                taken.append(host.numpy().copy())
            return taken[0]
        return result

    def broadcast(self, values, src=0):
        t = self._tensor(values)
        self._td.broadcast(t, src=src, group=self._group)
        return t.cpu().numpy()

    def gather_trials(self, local, ntrials_total=None):
        """All-gather arrays whose LAST axis is the local trial block -> full array on every rank."""
        local = np.ascontiguousarray(local, dtype=np.float64)
        counts = self.allreduce_sum(np.eye(self.world_size)[self.rank] * local.shape[-1]).astype(int)
        rmax = int(counts.max())
        pad = np.zeros(local.shape[:-1] + (rmax,))
        pad[..., :local.shape[-1]] = local
        t = self._tensor(np.moveaxis(pad, -1, 0))
        outs = [self._torch.empty_like(t) for _ in range(self.world_size)]
        self._td.all_gather(outs, t, group=self._group)
        parts = [np.moveaxis(o.cpu().numpy(), 0, -1)[..., :c] for o, c in zip(outs, counts)]
        return np.concatenate(parts, axis=-1)

    def on_device(self):
        """True when the collectives run on device tensors (backend nccl = RCCL over xGMI)."""
        return self._device.type == "cuda"

    d2h_bytes = 0               # bytes gather_trials_device has copied to the host on this rank (tests read it)

    def gather_trials_device(self, dev_local, dst=None):
        """All-gather a DEVICE array whose last axis is this rank's trial block -- a prediction left in HBM by
        gpcsd_predict_resident, as the zero-copy view Context.device_array() returns -- over the ranks ON THE DEVICE (one RCCL
        all-gather over xGMI: 230 MB per rank at 384 x 500 x 50 trials), reorder it to (..., all trials) there, and
        copy it out ONCE, into a page-locked array, on rank `dst` only (on every rank when dst is None).  Other ranks get None.
        No host staging: the only device-to-host traffic is the gathered result on the gathering rank(s)."""
        torch, td = self._torch, self._td
        from . import _hip
        x = torch.as_tensor(dev_local, device=self._device)
        counts = self.allreduce_sum(np.eye(self.world_size)[self.rank] * x.shape[-1]).astype(int)
        rmax = int(counts.max())
        xin = x.movedim(-1, 0)                                  # (local trials, ...): the trial blocks concatenate along axis 0
        if x.shape[-1] < rmax:
            xin = torch.cat([xin, xin.new_zeros((rmax - x.shape[-1],) + tuple(xin.shape[1:]))], dim=0)
        xin = xin.contiguous()
        out = torch.empty((self.world_size * rmax,) + tuple(xin.shape[1:]), dtype=xin.dtype, device=self._device)
        td.all_gather_into_tensor(out, xin, group=self._group)
        if dst is not None and self.rank != dst:
            return None
        if int(counts.min()) < rmax:                            # ragged blocks: drop the padding rows
            keep = torch.cat([torch.arange(r * rmax, r * rmax + int(c), device=self._device) for r, c in enumerate(counts)])
            out = out.index_select(0, keep)
        full = out.movedim(0, -1).contiguous()                  # (..., all trials), the reference's layout
        host = _hip.pinned_pool.empty(tuple(full.shape))
        torch.from_numpy(host).copy_(full)                      # the one device-to-host copy (synchronous)
        self.d2h_bytes += host.nbytes
        return host

    def barrier(self):
        self._td.barrier(group=self._group)


def sharded_loglik(local_parts_fn, ntrials_total, sharding):
    """Combine per-rank (sum log D, partial quad) into the global log-likelihood.
    local_parts_fn() -> (sumlog, quad_local)."""
    sumlog, quad = local_parts_fn()
    quad = float(sharding.allreduce_sum(np.array([quad]))[0])
    return -0.5 * ntrials_total * sumlog - 0.5 * quad
