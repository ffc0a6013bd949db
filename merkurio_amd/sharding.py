"""Record sharding across the GPUs of one node and the merge / reduction of per-shard results.

The path shards embarrassingly: records — or mate pairs, which are never split
(src/cmd_extract.rs:463-468,600-606) — are independent and the pattern set is replicated.  One
process per GPU; rank r owns the contiguous unit range shard_bounds(n, world)[r], so that
concatenating per-rank outputs in rank order reproduces the single-device output order.  The
only collective on the path is the sum of the counter vector at the end of a job
(`all_reduce_counters`: RCCL over xGMI with backend "nccl", gloo in the CPU tests).
"""
import numpy as np


def shard_bounds(n_units, world_size):
    """contiguous [lo, hi) ranges, sizes differing by at most one, in rank order"""
    base, rem = divmod(int(n_units), int(world_size))
    bounds, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < rem else 0)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def counters_layout(n_pat):
    """index map of the u64 counter vector of mk_scan_device (include/merkurio_hip.h)"""
    from . import native as mk
    return {"pattern_hit_counts": slice(0, n_pat), "hits": n_pat + mk.MK_SUM_HITS,
            "records_hit": n_pat + mk.MK_SUM_RECORDS_HIT, "records": n_pat + mk.MK_SUM_RECORDS,
            "bases": n_pat + mk.MK_SUM_BASES, "candidates": n_pat + mk.MK_SUM_CANDIDATES}


def all_reduce_counters(t):
    """in-place sum of the int64 counter tensor over all ranks (no-op without a process group)"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def merge_shards(shard_results, bounds):
    """shard_results[r] = (flags: bool/uint8 array over the shard's records, hits: structured
    array with shard-local `rec`) in rank order -> (flags, hits) of the whole batch, hits with
    global record indices, still in emission order (records ascend across shards)."""
    flags = np.concatenate([np.asarray(f, dtype=np.uint8) for f, _ in shard_results]) if shard_results else np.zeros(0, np.uint8)
    parts = []
    for (f, h), (lo, hi) in zip(shard_results, bounds):
        assert len(f) == hi - lo
        h = h.copy()
        h["rec"] += lo
        parts.append(h)
    hits = np.concatenate(parts) if parts else None
    return flags, hits


def gather_to_rank0(flags, hits, bounds):
    """collects every rank's (flags, hits) on rank 0 in rank order (object gather: results are
    sparse and small next to the scan); returns merge_shards(...) on rank 0, None elsewhere"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return merge_shards([(flags, hits)], bounds)
    rank, world = dist.get_rank(), dist.get_world_size()
    out = [None] * world if rank == 0 else None
    dist.gather_object((np.asarray(flags), np.asarray(hits)), out, dst=0)
    return merge_shards(out, bounds) if rank == 0 else None


def agree_on_communicator(lib, handle, rank, world, device, id_bytes=128):
    """One process per GPU: sets up the C ABI's RCCL communicator on every rank (mk_comm_unique_id on rank 0, the id
    carried by torch.distributed, mk_comm_init everywhere) -- or makes every rank give up together.  No rank may
    enter the collective init while a peer has already decided against it, so the ranks agree twice: first that
    each of them could bind librccl at all (a local check), then that every mk_comm_init came back with MK_OK.
    -> (True, "") or (False, reason of the first rank that failed); `device`: where the agreement tensors live
    ("cpu" under gloo, the rank's GPU under RCCL)."""
    import torch
    import torch.distributed as dist

    def all_ok(rc):
        why = "" if rc == 0 else (lib.mk_last_error() or b"").decode(errors="replace")
        ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        whys = [None] * world
        dist.all_gather_object(whys, (rank, why))
        bad = [f"rank {r}: {w}" for r, w in whys if w]
        return int(ok.item()) == 1, "; ".join(bad)

    buf = np.zeros(id_bytes, dtype=np.uint8)
    ok, why = all_ok(lib.mk_comm_unique_id(buf.ctypes.data) if rank == 0 else lib.mk_comm_available())
    if not ok:
        return False, why
    idt = torch.from_numpy(buf).to(device)  # rank 0's id is the job's id
    dist.broadcast(idt, src=0)
    idb = idt.cpu().numpy().copy()
    return all_ok(lib.mk_comm_init(handle, idb.ctypes.data, rank, world))
