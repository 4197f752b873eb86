"""Multi-GPU plumbing: one process per GPU, rendezvous and control plane through torch.distributed (backend "nccl" is
RCCL on ROCm, "gloo" on CPU for tests).

Two ways to use N GPUs (bench.py --mode):
* sharded (default): ONE proof split over the ranks (the single-GPU prover with every table a shard,
  csrc/lasso.cpp lasso_prove_sharded).  The data path does not go
  through torch: `attach_sharded` gives the prover's context its own RCCL communicator (lh_ctx_set_comm_rccl; the
  128-byte unique id travels over torch.distributed's broadcast) and every exchange is an ncclAllGather on the
  prover's stream.  With the gloo backend (CPU tests, several ranks on one GPU) the same prover runs over a host
  all-gather callback instead.
* replicas: every rank proves its own independent batch (weak scaling, no data-path collective).
torch.distributed itself only carries the barrier and the max-over-ranks of the timed region that bench.py's
contract asks for.
"""
import os
import sys

SEED_BASE = 0x4C4153534F00


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None):
    """Returns the torch.distributed module (initialised) or None for a single process."""
    rank, local_rank, world = env_rank()
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if backend is None:
        backend = os.environ.get("LH_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def batch_seed(log_n, rank):
    """PRNG seed of the lookup batch that `rank` proves: disjoint streams per rank, reproducible."""
    return SEED_BASE + log_n + 1000003 * rank


def max_over_ranks(dist, seconds):
    if dist is None:
        return seconds
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_objects(dist, obj):
    """every rank's `obj` (anything picklable, small) on every rank, in rank order"""
    if dist is None:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj, group=control_group(dist))
    return out


def barrier(dist):
    if dist is not None:
        dist.barrier()


def job_metrics(elapsed_s, steps, world, lookups_per_proof):
    """bench.py numbers for a weak-scaling job: every rank did `steps` proofs in (max) elapsed_s."""
    ms_per_step = elapsed_s * 1e3 / max(steps, 1)
    return {"ms_per_step": ms_per_step, "value_ms_per_proof": ms_per_step / world,
            "lookups_per_s": lookups_per_proof * world / (ms_per_step / 1e3)}


def host_all_gather(dist, group=None):
    """bytes -> bytes all-gather over CPU tensors (gloo), the communicator of a sharded proof.
    The exchanged data are a few hundred bytes per sum-check round and the residual tables once per
    sum-check; they live on the host anyway (the Fiat-Shamir transcript is there)."""
    import torch

    def all_gather(buf):
        world = dist.get_world_size(group)
        send = torch.frombuffer(bytearray(buf), dtype=torch.uint8)
        recv = torch.empty(world * len(buf), dtype=torch.uint8)
        dist.all_gather_into_tensor(recv, send, group=group)
        return recv.numpy().tobytes()
    return all_gather


def flight_group(dist):
    """A gloo group of all ranks for the host-side exchanges of a SECOND sharded proof in flight on every rank: two proofs
    driven by two host threads must not share one ordered channel.  (Collective: every rank calls it, once.)"""
    return dist.new_group(backend="gloo")


def attach_sharded(ctx, dist, shard_bit, group=None):
    """Give `ctx` the communicator of a sharded proof over all ranks of `dist`; returns "rccl" or "host".
    `group`: the gloo group the host transport uses (default: the control group) - a second proof in flight on the same
    ranks takes its own (flight_group); the RCCL transport creates a communicator of its own per call anyway."""
    import halo2_lasso_amd as hl
    rank, world = dist.get_rank(), dist.get_world_size()
    if dist.get_backend() == "nccl" and os.environ.get("LH_SHARDED_TRANSPORT", "rccl") == "rccl":
        import torch
        ok = 1
        try:
            ids = [hl.rccl_unique_id() if rank == 0 else None]
        except hl.Error:
            ids, ok = [None], 0
        dist.broadcast_object_list(ids, src=0)
        if ids[0] is not None:
            try:
                hl.attach_comm_rccl(ctx, rank, world, ids[0], shard_bit)
            except hl.Error as e:  # e.g. no peer access between the devices of this job
                sys.stderr.write("[lasso-hip] RCCL communicator failed on rank %d (%s); falling back to the host transport\n"
                                 % (rank, e))
                ok = 0
        else:
            ok = 0
        # every rank must end up on the same transport
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            return "rccl"
        hl.detach_comm(ctx)
    hl.attach_comm(ctx, rank, world, host_all_gather(dist, group if group is not None else control_group(dist)), shard_bit)
    return "host"


_control = None


def control_group(dist):
    """A gloo group for host-side exchanges next to an nccl (RCCL) default group (created once: new_group is collective)."""
    global _control
    if dist.get_backend() == "gloo":
        return None
    if _control is None:
        _control = dist.new_group(backend="gloo")
    return _control
