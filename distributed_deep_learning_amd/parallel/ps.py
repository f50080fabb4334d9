"""Parameter-server-style synchronisation (the idiom of the reference's empty MXNet / TensorFlow / Paddle stubs,
``/root/reference/src/{mxnet,tensorflow,paddle}/**``, SURVEY §2.3 "PS-style sync").

There is no server process: every rank IS the server of its shard.  Per bucket of the flat arena

    push   reduce_scatter(grad bucket)            -> each rank holds the averaged gradient of ITS shard
    update fused SGD / LARS / Adam on the shard    (fp32 master + momentum only for 1/world of the parameters;
                                                    LARS norms of tensors cut by shard borders are all-reduced)
    pull   all_gather(master shards)               -> every rank has the full updated parameters again

which on the fully connected xGMI mesh moves S/N bytes to each of the 7 peers at once instead of walking a ring.
The machinery lives in :class:`.ddp.DistributedDataParallel` (``mode="ps"``); this module is the named entry point.
"""
from __future__ import annotations

from .ddp import DistributedDataParallel


class ShardedDataParallel(DistributedDataParallel):
    """``DistributedDataParallel`` fixed to the reduce-scatter -> sharded step -> all-gather schedule (ZeRO-1-like)."""

    def __init__(self, module, **kw):
        kw.pop("mode", None)
        super().__init__(module, mode="ps", **kw)


def wrap(module, sync: str = "allreduce", **kw) -> DistributedDataParallel:
    """``sync`` = ``"allreduce"`` | ``"ps"`` (the ``--sync`` flag)."""
    if sync == "ps":
        return ShardedDataParallel(module, **kw)
    return DistributedDataParallel(module, mode="allreduce", **kw)
