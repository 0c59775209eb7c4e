"""Rendezvous of the test suites' spawned ranks WITHOUT a TCP port: a file:// store.  (Picking a free port by bind / close and
handing it to mp.spawn loses a race now and then - "EADDRINUSE" in the middle of a suite - and a flaky rank test is worse than none.)"""
import os
import tempfile
import uuid


def new_rendezvous():
    """A token for init_gloo: the path of a file that does not exist yet."""
    return os.path.join(tempfile.gettempdir(), f"hk_rendezvous_{os.getpid()}_{uuid.uuid4().hex}")


def init_gloo(rank, world, token):
    import torch.distributed as dist

    if isinstance(token, int):   # (a TCP port: what a launcher outside the suites hands over)
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(token)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("gloo", init_method="file://" + token, rank=rank, world_size=world)
