"""Data-parallel video QA over the GPUs of one node: one process per GPU, clips sharded by index, every rank holds a
full replica of tower + projector + decoder, and the ONLY collective on the path is the final answer collation
(one all-gather of a fixed-shape int32 token buffer + lengths; RCCL over xGMI with backend "nccl", gloo on CPU).

Replaces the serial `for sample in tqdm(gt_questions)` loops of the reference's eval runners
(video_chatgpt/eval/run_inference_qa_activitynet.py:63-104 and siblings): clips are independent, so the shard loop
needs no data-path communication.  A clip that fails keeps its slot with length 0 (the reference prints and
continues, :103-104), so the gather shape never depends on failures.
"""
from __future__ import annotations

import os
from typing import Callable, List, Sequence

import torch
import torch.distributed as dist


DEFAULT_TIMEOUT_S = 600      # of the rendezvous and of every collective: a rank that died must surface as an error on the others, never as a hang


class CollationError(RuntimeError):
    """The one exchange step of the path (the all-gather of the answers) failed: a peer rank died or did not arrive within the timeout."""


def init_distributed(backend: str | None = None, timeout_s: float | None = None) -> tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; single process when WORLD_SIZE is unset.  `timeout_s` (default
    PGV_DIST_TIMEOUT_S or 600) bounds the rendezvous and every collective: the survivors of a dead rank fail loudly within it."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        import datetime
        if backend is None:
            backend = os.environ.get("PGV_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if timeout_s is None:
            timeout_s = float(os.environ.get("PGV_DIST_TIMEOUT_S", DEFAULT_TIMEOUT_S))
        timeout = datetime.timedelta(seconds=timeout_s)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local), timeout=timeout)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout)
    return rank, world, local


# ---------------------------------------------------------------------------------------------------------------------------
# host-side placement: one rank per GPU, its host threads (frame prefetch, ATen / OMP pool, pinned staging buffers) on the cores of the
# NUMA node that GPU hangs off.  8 ranks x (prefetch thread + 8 OMP threads) on a 2-socket host otherwise wander over both sockets.
# ---------------------------------------------------------------------------------------------------------------------------
def _parse_cpulist(text: str) -> List[int]:
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out += list(range(int(lo), int(hi or lo) + 1))
    return out


def affinity_slice(node_cpus: Sequence[int], ranks_on_node: int, index: int) -> List[int]:
    """The `index`-th of `ranks_on_node` contiguous, near-equal slices of a node's CPU list (every CPU lands in exactly one slice; a rank
    never gets an empty one as long as there are at least as many CPUs as ranks)."""
    n = len(node_cpus)
    if ranks_on_node <= 1 or n < ranks_on_node:
        return list(node_cpus)
    base, rem = divmod(n, ranks_on_node)
    start = index * base + min(index, rem)
    return list(node_cpus[start:start + base + (1 if index < rem else 0)])


def gpu_numa_node(device_index: int):
    """NUMA node of a GPU from sysfs (/sys/bus/pci/devices/<domain:bus:dev.fn>/numa_node); None when unknown (no sysfs entry, -1, no GPU)."""
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip())
        return node if node >= 0 else None
    except Exception:                                            # noqa: BLE001 -- placement is best effort by design
        return None


def pin_rank_to_numa_node(local_rank: int, n_local: int | None = None):
    """Bind this process (all threads started after the call inherit it) to this rank's share of the cores of its GPU's NUMA node.
    Falls back silently -- returns None and changes nothing -- when the topology cannot be read, the mask would be empty, or
    PGV_RANK_AFFINITY=0.  Returns the CPU list otherwise."""
    if os.environ.get("PGV_RANK_AFFINITY", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        n_local = n_local or int(os.environ.get("LOCAL_WORLD_SIZE", "0")) or torch.cuda.device_count()
        nodes = [gpu_numa_node(i) for i in range(n_local)]
        mine = nodes[local_rank] if local_rank < len(nodes) else None
        if mine is None:
            return None
        allowed = os.sched_getaffinity(0)
        cpus = [c for c in _parse_cpulist(open(f"/sys/devices/system/node/node{mine}/cpulist").read()) if c in allowed]
        peers = [i for i, n in enumerate(nodes) if n == mine]
        share = affinity_slice(cpus, len(peers), peers.index(local_rank))
        if not share:
            return None
        os.sched_setaffinity(0, share)
        torch.set_num_threads(max(1, min(torch.get_num_threads(), len(share), 8)))
        return share
    except Exception:                                            # noqa: BLE001
        return None


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Contiguous blocks, remainder spread over the first ranks: rank r owns [start_r, start_r + count_r)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def shard_capacity(n_items: int, world: int) -> int:
    return (n_items + world - 1) // world


def gather_answers(tokens: torch.Tensor, lengths: torch.Tensor, n_items: int, rank: int, world: int, length_offset: int = 0):
    """Collate per-rank answers.  tokens [cap, max_new] int32 and lengths [cap] int32 (cap = shard_capacity; unused slots
    have length -1) -> list of n_items token-id lists in global clip order, identical on every rank.
    length_offset = 1: lengths carry (answer length + 1) and 0 marks a failed clip, which comes back as None -- an empty answer
    (first token EOS) is then distinguishable from a failure.
    world == 1 short-circuits without touching torch.distributed."""
    cap, width = tokens.shape
    assert cap == shard_capacity(n_items, world) and lengths.shape == (cap,)
    if world > 1:
        packed = torch.cat([tokens.reshape(-1), lengths]).contiguous()
        if dist.get_backend() == "gloo":                    # CPU tests and the shared-device bench smoke test: stage through the host
            packed = packed.cpu()
        out = torch.empty(world * packed.numel(), dtype=packed.dtype, device=packed.device)
        try:
            dist.all_gather_into_tensor(out, packed)
            out = out.view(world, -1).cpu()                   # (nccl: the copy synchronises, so an asynchronous failure surfaces here)
        except Exception as e:                                # noqa: BLE001 -- a dead or late peer: never a hang (init_distributed's timeout), always loud
            raise CollationError(f"rank {rank}: the answer all-gather failed ({type(e).__name__}: {e}); a peer rank died or did not arrive within "
                                 f"the timeout.  This rank's own {cap} answer slots were complete.") from e
    else:
        out = torch.cat([tokens.reshape(-1), lengths]).view(1, -1).cpu()
    answers: List[List[int]] = [None] * n_items
    for r in range(world):
        toks = out[r, : cap * width].view(cap, width)
        lens = out[r, cap * width:]
        for slot, idx in enumerate(shard_indices(n_items, r, world)):
            n = int(lens[slot]) - length_offset
            answers[idx] = None if (length_offset and n < 0) else toks[slot, :max(n, 0)].tolist()
    return answers


def run_sharded(n_items: int, infer_batch: Callable[..., tuple], max_new_tokens: int, rank: int, world: int,
                device, per_gpu_batch: int = 8, length_offset: int = 0, prepare: Callable[[Sequence[int]], object] | None = None):
    """Run `infer_batch(indices) -> (tokens [len(indices), <=max_new] int tensor, lengths list)` over this rank's shard in
    groups of `per_gpu_batch`, then collate.  A group that raises keeps its slots with length 0.  `length_offset`: see gather_answers
    (the token count stored for slot j is lengths[j] - length_offset).
    `prepare(indices) -> obj` (optional) is the HOST half of a group (decode / sample the frames, pin them): it runs on a background thread
    ONE GROUP AHEAD of the device half, which then receives its result as `infer_batch(indices, obj)` -- the GPU works on group g while the
    host reads group g + 1.  An exception inside `prepare` is delivered to the group it belongs to (that group fails, the others go on)."""
    mine = shard_indices(n_items, rank, world)
    cap = shard_capacity(n_items, world)
    tokens = torch.zeros(cap, max_new_tokens, dtype=torch.int32, device=device)
    lengths = torch.full((cap,), -1, dtype=torch.int32, device=device)
    groups = [mine[g0:g0 + per_gpu_batch] for g0 in range(0, len(mine), per_gpu_batch)]
    pool = fut = None
    if prepare is not None and groups:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="pgv-prefetch")
        fut = pool.submit(prepare, groups[0])
    try:
        for gi, group in enumerate(groups):
            g0 = gi * per_gpu_batch
            try:
                if pool is not None:
                    prepared_f, fut = fut, (pool.submit(prepare, groups[gi + 1]) if gi + 1 < len(groups) else None)
                    toks, lens = infer_batch(group, prepared_f.result())
                else:
                    toks, lens = infer_batch(group)
                for j in range(len(group)):
                    n = max(int(lens[j]) - length_offset, 0)
                    tokens[g0 + j, :n] = toks[j, :n].to(device=device, dtype=torch.int32)
                    lengths[g0 + j] = int(lens[j])
            except Exception as e:                               # noqa: BLE001 -- same "print and continue" policy as the reference
                print(f"[rank {rank}] Error processing clips {group}: {e}")
                lengths[g0:g0 + len(group)] = 0
    finally:
        if pool is not None:
            pool.shutdown(wait=True, cancel_futures=True)
    return gather_answers(tokens, lengths, n_items, rank, world, length_offset)
