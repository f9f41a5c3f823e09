"""Blob-batch sharding across the GPUs of one node (one process per GPU, torch.distributed).

The hot path shards embarrassingly: every blob is independent, so a batch is split
into contiguous slices, each rank runs its slice on its own GPU with a replicated context (SRS tables,
twiddles), and the only exchange is ONE all-gather of the per-rank result slabs (RCCL over xGMI when the
backend is "nccl"; the CPU tests use "gloo").  No collective sits inside the compute path.
"""
from typing import Callable, List, Sequence, Tuple


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """[lo, hi) of rank's contiguous slice; the first n % world ranks get one extra item."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_flat(local, out=None, dist=None):
    """The one exchange of the sharded path: all-gather equal-size per-rank result tensors (proofs: 128 x 48 B per blob)
    into `out` ([world * local.numel()], allocated if None).  RCCL over xGMI with the "nccl" backend; gloo on CPU."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local if out is None else out.copy_(local)
    if out is None:
        out = torch.empty(dist.get_world_size() * local.numel(), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.reshape(-1))
    return out


def attach_library_comm(ctx, dist, required=False, timeout_s=None) -> bool:
    """Give `ctx` the library's own RCCL communicator (eth_kzg_amd_comm_init): rank 0 draws the id, torch.distributed
    only carries its 128 bytes to the other ranks.  After this, all_gather_proofs runs ncclAllGather inside
    libc_eth_kzg.so, which is also what a C / Go / Java host would call.

    comm_init is itself a collective (ncclCommInitRank), so no rank may enter it unless all do: every rank first runs the
    local probe (RCCL bindable, no communicator attached yet; rank 0 also draws the id) and the ranks agree on the MIN of
    those flags BEFORE anyone calls comm_init.  All ranks or none.  Returns False -- with the reason in
    `attach_library_comm.last_error` -- if some rank cannot take part; raises instead when `required`.

    Watchdog: comm_init runs on a helper thread and is given `timeout_s` seconds ($KZG_COMM_INIT_TIMEOUT_S, default 60).  A
    rank whose ncclCommInitRank has not returned by then reports that as its failure; the ranks then agree (over torch's
    communicator, which is independent of the library's) that the library communicator is unusable and every one of them takes
    the failure path -- nobody waits for ever, the caller falls back to torch.distributed's all-gather and says so.  The stuck
    thread is left behind (`attach_library_comm.stuck` is set: the caller should leave with os._exit when it is done)."""
    import os
    import threading
    attach_library_comm.last_error = ""
    attach_library_comm.stuck = False
    if timeout_s is None:
        timeout_s = float(os.environ.get("KZG_COMM_INIT_TIMEOUT_S", "60"))
    if dist is None or not dist.is_initialized():
        attach_library_comm.last_error = "torch.distributed is not initialised"
        if required:
            raise RuntimeError(attach_library_comm.last_error)
        return False
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    on_cpu = dist.get_backend() != "nccl"
    box, ready, why = [None], 1, ""
    try:
        ctx.comm_probe()
        if rank == 0:
            box[0] = ctx.comm_unique_id()
    except Exception as e:
        ready, why = 0, f"rank {rank}: {e}"
    flag = torch.tensor([ready], dtype=torch.int32, device="cpu" if on_cpu else "cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if not int(flag.item()):
        reasons = [None] * world
        dist.all_gather_object(reasons, why)
        attach_library_comm.last_error = "; ".join(r for r in reasons if r) or "a rank could not bind RCCL"
        if required:
            raise RuntimeError("library communicator unavailable: " + attach_library_comm.last_error)
        return False
    dist.broadcast_object_list(box, src=0)
    # from here on every rank is known to be able to enter the collective
    ok, why = 1, ""
    result = {}

    def enter():
        try:
            ctx.comm_init(box[0], rank, world)
            result["info"] = ctx.comm_info()
        except Exception as e:  # reported below
            result["error"] = e
    th = threading.Thread(target=enter, name="kzg-comm-init", daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        attach_library_comm.stuck = True
        ok, why = 0, f"rank {rank}: ncclCommInitRank did not return within {timeout_s:g} s (watchdog)"
    elif "error" in result:
        ok, why = 0, f"rank {rank}: {result['error']}"
    elif result.get("info") != (rank, world):
        ok, why = 0, f"rank {rank}: communicator reports (rank, world) = {result.get('info')}, expected {(rank, world)}"
    flag = torch.tensor([ok], dtype=torch.int32, device="cpu" if on_cpu else "cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if not int(flag.item()):
        reasons = [None] * world
        dist.all_gather_object(reasons, why)
        attach_library_comm.last_error = "; ".join(r for r in reasons if r)
        if not attach_library_comm.stuck:  # (a communicator another thread is still building is left alone)
            try:
                ctx.comm_destroy()
            except Exception:
                pass
        if required:
            raise RuntimeError("library communicator could not be built: " + attach_library_comm.last_error)
        return False
    return True


attach_library_comm.last_error = ""
attach_library_comm.stuck = False


def all_gather_proofs(ctx, local, out, stream=None):
    """The one exchange of the sharded prover path through the library's communicator: equal-size per-rank proof slabs
    (torch uint8 tensors on the GPU) into out[world * local.numel()], enqueued on `stream` (a torch stream or None)."""
    ctx.all_gather(local.data_ptr(), out.data_ptr(), local.numel(), stream.cuda_stream if stream is not None else None)
    return out


def gather_slabs(local: bytes, item_bytes: int, n_total: int, dist=None) -> bytes:
    """All-gather variable-length per-rank slabs (len(local) = items_on_rank * item_bytes) into the
    full [n_total * item_bytes] byte string on every rank.  Slabs are padded to the largest shard so a
    single fixed-size all_gather does the exchange."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        assert len(local) == n_total * item_bytes
        return local
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    max_items = max(shard_bounds(n_total, world, r)[1] - shard_bounds(n_total, world, r)[0] for r in range(world))
    backend = dist.get_backend()
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    buf = torch.zeros(max_items * item_bytes, dtype=torch.uint8, device=dev)
    if local:
        buf[:len(local)] = torch.frombuffer(bytearray(local), dtype=torch.uint8).to(dev)
    out = torch.empty(world * max_items * item_bytes, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, buf)
    out = out.cpu().numpy().tobytes()
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, world, r)
        parts.append(out[r * max_items * item_bytes: r * max_items * item_bytes + (hi - lo) * item_bytes])
    return b"".join(parts)


def run_sharded(compute: Callable[[Sequence[bytes]], List[bytes]], blobs: Sequence[bytes], item_bytes: int, dist=None) -> List[bytes]:
    """Run `compute` (a per-rank batch function returning one item_bytes-long result per blob) on this rank's
    slice and return the results for ALL blobs, in order, on every rank."""
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    lo, hi = shard_bounds(len(blobs), world, rank)
    local = compute(blobs[lo:hi]) if hi > lo else []
    assert all(len(x) == item_bytes for x in local)
    full = gather_slabs(b"".join(local), item_bytes, len(blobs), dist)
    return [full[i * item_bytes:(i + 1) * item_bytes] for i in range(len(blobs))]


VERIFY_RECORD_BYTES = 104  # 96-byte partial + status byte, padded to 8


def verify_cell_kzg_proof_batch_sharded(ctx, commitments, cell_indices, cells, proofs, dist=None) -> bool:
    """verify_cell_kzg_proof_batch with the cell list split over the ranks.

    Every rank passes the same full batch.  Rank r evaluates cells shard_bounds(n, world, r) on its GPU
    (ctx.verify_cell_kzg_proof_batch_partial: two G1 partial sums, 96 B), the records are all-gathered (< 1 KB per
    rank - the only exchange), and every rank adds them and runs the pairing check (ctx.verify_cell_kzg_proof_batch_combine).
    A rank whose slice is malformed still takes part in the gather, so no rank is left waiting; the error is then
    raised on all ranks, as the unsharded call would raise it."""
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    lo, hi = shard_bounds(len(cells), world, rank)
    status, partial, message = 0, bytes(96), ""
    try:
        partial = ctx.verify_cell_kzg_proof_batch_partial(commitments, cell_indices, cells, proofs, lo, hi)
    except Exception as e:  # reported after the gather
        status, message = 1, str(e)
    record = partial + bytes([status]) + bytes(VERIFY_RECORD_BYTES - 97)
    if world > 1:
        import torch
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        local = torch.frombuffer(bytearray(record), dtype=torch.uint8).to(dev)
        gathered = all_gather_flat(local, None, dist).cpu().numpy().tobytes()
    else:
        gathered = record
    records = [gathered[i * VERIFY_RECORD_BYTES:(i + 1) * VERIFY_RECORD_BYTES] for i in range(world)]
    failed = [r for r, rec in enumerate(records) if rec[96]]
    if failed:
        from . import KzgError
        raise KzgError(message or f"verify_cell_kzg_proof_batch: malformed input in the slice of rank {failed[0]}")
    return ctx.verify_cell_kzg_proof_batch_combine([rec[:96] for rec in records])
