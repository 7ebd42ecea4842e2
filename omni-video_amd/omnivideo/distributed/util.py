"""Sequence-parallel process groups and collectives with the reference's names
(reference: omnivideo/distributed/util.py).  One process per GPU; on MI355X the "nccl"
backend of torch.distributed is RCCL over xGMI.  The collectives are written against
`all_to_all_single` / `all_gather_into_tensor` (one contiguous buffer per call instead of
the reference's per-peer tensor lists, util.py:106-109,143-145), which also run on the
gloo backend for CPU tests."""
import torch
import torch.distributed as dist

_SEQUENCE_PARALLEL_GROUP = None
_SEQUENCE_PARALLEL_SIZE = 1


def init_distributed_group():
    """reference util.py:9-13"""
    if not dist.is_initialized():
        dist.init_process_group(backend="nccl")


def _join_partition(blocks):
    """Creates one process group per block of `blocks` (a partition of the world's ranks; every rank
    has to take part in every `new_group` call, in the same order) and returns the group this rank
    belongs to."""
    me, mine = dist.get_rank(), None
    for ranks in blocks:
        group = dist.new_group(list(ranks))
        if me in ranks:
            mine = group
    return mine


def init_sequence_parallel_group(sp_size):
    """Sequence-parallel groups = consecutive blocks of sp_size ranks (reference util.py:16-40); the
    group of this rank is remembered for the getters below."""
    global _SEQUENCE_PARALLEL_GROUP, _SEQUENCE_PARALLEL_SIZE
    if not dist.is_initialized():
        return None
    world = dist.get_world_size()
    if world % sp_size:
        raise AssertionError(f"world size {world} is not a multiple of sp_size {sp_size}")
    _SEQUENCE_PARALLEL_GROUP = _join_partition(range(b * sp_size, (b + 1) * sp_size) for b in range(world // sp_size))
    _SEQUENCE_PARALLEL_SIZE = sp_size
    return _SEQUENCE_PARALLEL_GROUP


def init_distributed_groups(sp_size):
    """(data-parallel group, sequence-parallel group) of this rank: the data-parallel groups are the
    strided complements of the sequence-parallel blocks (reference util.py:43-66)."""
    init_distributed_group()
    world = dist.get_world_size()
    sp_group = init_sequence_parallel_group(sp_size)
    dp_group = _join_partition(range(offset, world, sp_size) for offset in range(sp_size))
    return dp_group, sp_group


def get_sequence_parallel_group():
    return _SEQUENCE_PARALLEL_GROUP


def get_sequence_parallel_size():
    return _SEQUENCE_PARALLEL_SIZE


def get_sequence_parallel_rank():
    if _SEQUENCE_PARALLEL_GROUP is None:
        return 0
    return dist.get_rank(_SEQUENCE_PARALLEL_GROUP)


def get_rank(group=None):
    if group is None:
        group = _SEQUENCE_PARALLEL_GROUP
    if group is None and not dist.is_initialized():
        return 0
    return dist.get_rank(group=group)


def get_world_size(group=None):
    if group is None:
        group = _SEQUENCE_PARALLEL_GROUP
    if group is None and not dist.is_initialized():
        return 1
    return dist.get_world_size(group=group)


def _resolve(group):
    return _SEQUENCE_PARALLEL_GROUP if group is None else group


def _host_only(group):
    """gloo moves host memory only: device tensors are staged through the host (rehearsals of the
    multi-rank path by several processes on one GPU; RCCL takes device tensors directly)."""
    from .host_staged import moves_host_memory_only
    return moves_host_memory_only(_resolve(group))


def all_to_all(x, scatter_dim, gather_dim, group=None, **kwargs):
    """Scatter x along `scatter_dim`, gather along `gather_dim` (reference util.py:95-121).
    Inference only: no autograd function is attached."""
    world = get_world_size(group)
    if world <= 1:
        return x
    if x.size(scatter_dim) % world:
        raise ValueError(f"dim {scatter_dim} of size {x.size(scatter_dim)} is not divisible by {world} ranks")
    send = torch.stack([u for u in x.chunk(world, dim=scatter_dim)]).contiguous()
    recv = torch.empty_like(send)
    if send.is_cuda and _host_only(group):
        from .host_staged import HostStagedComm
        HostStagedComm(_resolve(group)).all_to_all(recv, send)
    else:
        dist.all_to_all_single(recv, send, group=_resolve(group))
    return torch.cat(list(recv.unbind(0)), dim=gather_dim).contiguous()


def all_gather(tensor, group=None):
    """reference util.py:124-130"""
    world = get_world_size(group)
    if world == 1:
        return [tensor]
    dev = torch.device("cpu") if tensor.is_cuda and _host_only(group) else tensor.device
    out = torch.empty(world * tensor.numel(), dtype=tensor.dtype, device=dev)
    dist.all_gather_into_tensor(out, tensor.contiguous().view(-1).to(dev), group=_resolve(group))
    return list(out.to(tensor.device).view((world,) + tuple(tensor.shape)).unbind(0))


def gather_forward(input, dim, group=None):
    """All-gather along `dim` (reference util.py:133-161)."""
    world = get_world_size(group)
    if world <= 1:
        return input
    return torch.cat(all_gather(input, group), dim=dim).contiguous()
