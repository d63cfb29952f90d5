"""The handful of ``torch.distributed`` calls the multi-GPU engines make, behind one seam.

A process group is normally a ``torch.distributed`` group (RCCL on the GPUs, gloo in the CPU tests) and every function
here forwards to ``torch.distributed`` unchanged.  A group object that carries its own collectives -- it has a
``hiprec_collectives`` attribute: an object with ``size / rank / backend / all_reduce / all_to_all_single / all_gather /
broadcast / new_group / ranks`` and, for the C step drivers, ``create_communicator(device)`` -- is served by those
instead.  That is how ``tests/loopback.py`` runs R ranks as R host threads on ONE GPU: the engines' N > 1 code,
including the exchanges the C drivers post themselves, executes on the single-GPU test box.  Plumbing only."""
import torch.distributed as _td

ReduceOp = _td.ReduceOp


def own(group):
    """The group's own collectives, or None for a torch.distributed group."""
    return getattr(group, "hiprec_collectives", None)


def get_world_size(group=None):
    c = own(group)
    return c.size() if c is not None else _td.get_world_size(group)


def get_rank(group=None):
    c = own(group)
    return c.rank() if c is not None else _td.get_rank(group)


def get_backend(group=None):
    c = own(group)
    return c.backend() if c is not None else _td.get_backend(group)


def get_global_rank(group, group_rank):
    return group_rank if own(group) is not None else _td.get_global_rank(group, group_rank)


def get_process_group_ranks(group):
    c = own(group)
    return c.ranks() if c is not None else _td.get_process_group_ranks(group)


def new_group(ranks=None, like=None):
    """A further group over the same ranks as ``like`` (a communicator of its own: no ordering against ``like``'s
    collectives).  Collective over ``like``."""
    c = own(like)
    return c.new_group() if c is not None else _td.new_group(ranks=ranks)


def all_reduce(tensor, op=ReduceOp.SUM, group=None):
    c = own(group)
    return c.all_reduce(tensor, op) if c is not None else _td.all_reduce(tensor, op=op, group=group)


def all_to_all_single(output, input, output_split_sizes=None, input_split_sizes=None, group=None):  # noqa: A002
    c = own(group)
    if c is not None:
        return c.all_to_all_single(output, input, output_split_sizes, input_split_sizes)
    return _td.all_to_all_single(output, input, output_split_sizes=output_split_sizes,
                                 input_split_sizes=input_split_sizes, group=group)


def all_gather(tensor_list, tensor, group=None):
    c = own(group)
    return c.all_gather(tensor_list, tensor) if c is not None else _td.all_gather(tensor_list, tensor, group=group)


def broadcast(tensor, src=0, group=None):
    c = own(group)
    return c.broadcast(tensor, src) if c is not None else _td.broadcast(tensor, src=src, group=group)
