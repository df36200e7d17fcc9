"""The collectives of the sequence-parallel / CFG-parallel paths, behind one replaceable transport.

Production transport: `torch.distributed` as it is -- backend "nccl" is RCCL on ROCm, one process
per GPU, collectives run on RCCL's own stream over xGMI.  The reference calls the same functions
directly (lightx2v/common/ops/attn/utils/all2all.py:29-39,70-81, lightx2v/models/networks/wan/model.py:412-427,493-499).

`HostStagedTransport` exists for functional runs of N ranks that share ONE GPU (tests, rehearsals):
RCCL refuses two ranks on one device, and gloo's own CUDA path stalls there -- a bare loop of
`dist.all_to_all_single` on CUDA tensors between two processes sharing a GPU parks both ranks
inside the call after a handful of iterations (tools/probe_gloo_shared_gpu.py, scenario A; nothing
of this package is involved).  The staged transport hands gloo host tensors only: the payload is
copied to the host when the collective is issued, the exchange itself stays asynchronous (several
may be outstanding, in issue order, like on RCCL), and the result is copied to the GPU on `wait()`.
"""

from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


# RCCL 2.26.6 (ROCm 7.0) moves only the first half of a per-peer payload above 1 GiB and reports success
# (tools/probe_rccl_self_a2a.py: 1024 MiB arrives whole, 1107 MiB arrives as its first 553.5 MiB).
RCCL_MAX_PEER_BYTES = 1 << 30


class EmulatedGroup:
    """Stand-in for a process group of `size` ranks of which this process plays rank `rank`, with no
    peers behind it: `EmulatedPeersTransport` answers its collectives from the rank's own data.  A
    single GPU then runs exactly the kernels one rank of such a group runs (token-wise work on S/N
    rows, attention on H/N heads over all S rows, segmented output projection) -- a way to time one
    rank's compute, never a way to produce results."""

    def __init__(self, size: int, rank: int = 0):
        if not 0 <= rank < size:
            raise ValueError(f"EmulatedGroup: rank {rank} outside a group of {size}")
        self.size, self.rank = size, rank


def group_size(group) -> int:
    return group.size if isinstance(group, EmulatedGroup) else dist.get_world_size(group)


def group_rank(group) -> int:
    return group.rank if isinstance(group, EmulatedGroup) else dist.get_rank(group)


class DistTransport:
    """torch.distributed, unchanged (plus a refusal of payloads RCCL is known to truncate)."""

    name = "torch.distributed"

    def all_to_all_single(self, output: torch.Tensor, input: torch.Tensor, group=None, async_op: bool = False):
        per_peer = input.numel() * input.element_size() // max(1, group_size(group))
        if per_peer > RCCL_MAX_PEER_BYTES and input.is_cuda:
            raise ValueError(f"all_to_all_single: {per_peer / 2**20:.0f} MiB per peer; RCCL moves payloads above 1024 MiB only in part -- split the exchange "
                             "(the Ulysses exchange does: LX2V_ULYSSES_MAX_PEER_MB)")
        return dist.all_to_all_single(output, input, group=group, async_op=async_op)

    def all_gather(self, outputs: List[torch.Tensor], input: torch.Tensor, group=None, async_op: bool = False):
        return dist.all_gather(outputs, input, group=group, async_op=async_op)


class _StagedWork:
    def __init__(self, work, finish):
        self._work, self._finish = work, finish

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._finish()
            self._work = None
        return True


class HostStagedTransport(DistTransport):
    """Collectives on host copies (see the module docstring).  Payloads travel as bytes, so every
    dtype the GPU path uses (bf16, e4m3 bytes, fp32 scales) goes through the same gloo call."""

    name = "host-staged gloo"

    def all_to_all_single(self, output, input, group=None, async_op=False):
        world = dist.get_world_size(group)
        if input.numel() % world != 0 or output.numel() != input.numel() or not output.is_contiguous():
            raise ValueError("host-staged all_to_all_single: equal splits of a contiguous output are supported")
        src = input.contiguous().reshape(world, -1).view(torch.uint8).cpu()  # waits for the producer kernels
        dst = torch.empty_like(src)
        work = dist.all_to_all_single(dst, src, group=group, async_op=True)
        staged = _StagedWork(work, lambda: output.view(torch.uint8).reshape(-1).copy_(dst.reshape(-1)))
        if async_op:
            return staged
        staged.wait()
        return None

    def all_gather(self, outputs, input, group=None, async_op=False):
        src = input.contiguous().reshape(-1).view(torch.uint8).cpu()
        parts = [torch.empty_like(src) for _ in outputs]
        work = dist.all_gather(parts, src, group=group, async_op=True)

        def finish():
            for o, p in zip(outputs, parts):
                if not o.is_contiguous():
                    raise ValueError("host-staged all_gather: outputs must be contiguous")
                o.view(torch.uint8).reshape(-1).copy_(p)

        staged = _StagedWork(work, finish)
        if async_op:
            return staged
        staged.wait()
        return None


class _DoneWork:
    def wait(self):
        return True


class EmulatedPeersTransport(DistTransport):
    """Collectives of an `EmulatedGroup`: every peer is assumed to hold what this rank holds, so an
    all-to-all is a device copy of the send buffer into the receive buffer (slot j keeps this rank's
    slot j: distinct, realistically distributed values in every row) and an all_gather fills every
    output with the input.  The copies run on the current stream; nothing is overlapped and no link
    is involved, which is the point: what remains is one rank's compute."""

    name = "emulated peers (one rank's compute only)"

    def all_to_all_single(self, output, input, group=None, async_op=False):
        if not isinstance(group, EmulatedGroup):
            raise ValueError("EmulatedPeersTransport serves EmulatedGroup groups only")
        if output.numel() != input.numel():
            raise ValueError("emulated all_to_all_single: equal splits only")
        output.view(-1).copy_(input.reshape(-1))
        return _DoneWork() if async_op else None

    def all_gather(self, outputs, input, group=None, async_op=False):
        if not isinstance(group, EmulatedGroup):
            raise ValueError("EmulatedPeersTransport serves EmulatedGroup groups only")
        for o in outputs:
            o.copy_(input)
        return _DoneWork() if async_op else None


_transport: DistTransport = DistTransport()


def transport() -> DistTransport:
    return _transport


def set_transport(t: Optional[DistTransport]) -> DistTransport:
    """Install `t` (None = torch.distributed) for this process; returns the previous transport."""
    global _transport
    old, _transport = _transport, (t if t is not None else DistTransport())
    return old
