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


# Timeout of the rendezvous and of every collective (PGV_DIST_TIMEOUT_S).  The one collective of the path sits at the END of a shard, so the
# timeout also bounds the SKEW between ranks (unbalanced shards, a rank retrying clips one by one): two hours, not ten minutes -- a rank that
# died still surfaces as an error on the others, never as a hang, and finished work is not lost meanwhile because every rank writes its own
# answers to a rank-local file before collating (run_sharded spill_path).  Backend behaviour on expiry: gloo raises on the waiting ranks
# (-> CollationError below); with nccl (RCCL) the watchdog thread normally ABORTS the process instead of raising, so the `except` branch of
# gather_answers is not reached there -- the rank-local files are what survives in both cases.
DEFAULT_TIMEOUT_S = 7200


class CollationError(RuntimeError):
    """The one exchange step of the path (the all-gather of the answers) failed: a peer rank died or did not arrive within the timeout."""


def init_distributed(backend: str | None = None, timeout_s: float | None = None) -> tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; single process when WORLD_SIZE is unset.  `timeout_s` (default
    PGV_DIST_TIMEOUT_S or DEFAULT_TIMEOUT_S) bounds the rendezvous and every collective: the survivors of a dead rank fail loudly within it."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # PGV_DIST_FORCE_INIT=1: build the process group also for ONE rank (legal for both backends) -- how the RCCL branch below and the
    # device-side all-gather of gather_answers are exercised on a 1-GPU box (tests/test_gpu_runners.py)
    if (world > 1 or os.environ.get("PGV_DIST_FORCE_INIT") == "1") and not dist.is_initialized():
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
            _arm_rccl_debug_capture()
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local), timeout=timeout)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout)
    return rank, world, local


# ---------------------------------------------------------------------------------------------------------------------------
# Proof of transport: what a reader of the bench line / runner log needs to see that the N ranks really ran over RCCL on N distinct GPUs and
# which path the ring took.  Nothing here changes the data path.
# ---------------------------------------------------------------------------------------------------------------------------
_rccl_debug_file = None


def _arm_rccl_debug_capture():
    """Before the process group exists: route this process's RCCL INFO log to a private temp file (NCCL_DEBUG / NCCL_DEBUG_FILE are read when the
    communicator is created), unless the user already directs it somewhere or PGV_RCCL_DEBUG_CAPTURE=0.  The channel set-up lines it holds
    ("Channel 00/0 : 0[0] -> 1[1] via P2P/IPC ...", "... via SHM ...", "... via NET/Socket ...") are what rccl_transport() counts."""
    global _rccl_debug_file
    if os.environ.get("PGV_RCCL_DEBUG_CAPTURE", "1") == "0" or "NCCL_DEBUG_FILE" in os.environ:
        _rccl_debug_file = os.environ.get("NCCL_DEBUG_FILE")         # set by rccl_debug_env() before the interpreter loaded librccl, or by the user
        return
    import tempfile
    fd, path = tempfile.mkstemp(prefix=f"pgv_rccl_{os.getpid()}_", suffix=".log")
    os.close(fd)
    os.environ["NCCL_DEBUG_FILE"] = path
    os.environ["NCCL_DEBUG"] = "INFO"
    _rccl_debug_file = path


def rccl_debug_env(env: dict | None = None) -> dict:
    """The two variables that make RCCL write its INFO log to a per-process file, as a dict to merge into a child's environment -- or, with
    env=None, applied to THIS process (call it before `import torch`: RCCL caches its debug level the first time the library is touched, which
    `import torch` already does -- setting the variables in init_distributed was too late on the MI355X box, gpurun_out/r6c).  Respects what the
    user already set."""
    target = os.environ if env is None else env
    if target.get("PGV_RCCL_DEBUG_CAPTURE", "1") != "0" and "NCCL_DEBUG_FILE" not in target:
        import tempfile
        target["NCCL_DEBUG_FILE"] = os.path.join(tempfile.gettempdir(), "pgv_rccl_%h_%p.log")
        target["NCCL_DEBUG"] = "INFO"          # the channel lines are INFO level (the pool's boxes export NCCL_DEBUG=WARN); PGV_RCCL_DEBUG_CAPTURE=0 leaves both alone
    return target


def rccl_transport(cleanup: bool = True) -> dict:
    """Counts of the channel transports in this process's captured RCCL log: {"p2p", "shm", "net", "xgmi_mentions", "log_lines", "verdict"}.
    verdict: "P2P" (every channel peer-to-peer: xGMI / PCIe P2P inside a node), "SHM" / "NET" (a fallback through host memory / sockets was
    used for at least one channel), "none" (no channel lines: one rank, or the log was not captured)."""
    import re
    out = {"p2p": 0, "shm": 0, "net": 0, "xgmi_mentions": 0, "log_lines": 0, "verdict": "none", "captured": bool(_rccl_debug_file)}
    path = _rccl_debug_file
    if not path:
        return out
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)      # RCCL writes through a buffered FILE*: flush every C stream of the process so that the lines are in the file
    except Exception:                       # noqa: BLE001
        pass
    try:
        for cand in (path, path.replace("%h", os.uname().nodename).replace("%p", str(os.getpid()))):
            if os.path.exists(cand):
                with open(cand, errors="replace") as f:
                    for line in f:
                        out["log_lines"] += 1
                        if re.search(r"\bvia P2P", line):
                            out["p2p"] += 1
                        elif re.search(r"\bvia SHM", line):
                            out["shm"] += 1
                        elif re.search(r"\bvia NET", line):
                            out["net"] += 1
                        if "XGMI" in line.upper():
                            out["xgmi_mentions"] += 1
                if cleanup and os.path.basename(cand).startswith("pgv_rccl_"):
                    try:
                        os.remove(cand)
                    except OSError:
                        pass
                break
    except OSError:
        return out
    out["verdict"] = "NET" if out["net"] else ("SHM" if out["shm"] else ("P2P" if out["p2p"] else "none"))
    return out


def collective_identity(device, rank: int, world: int, allow_shared_device: bool = False) -> dict:
    """Who took part in the process group: backend, RCCL version, and per rank (gathered over the group itself) the host, the GPU's PCI address
    and its NUMA node.  Raises RuntimeError when fewer ranks answer than `world`, or when two ranks of one host sit on the same GPU and
    `allow_shared_device` is not set (a mis-launched job -- every rank on device 0 -- must not produce a scaling number)."""
    backend = dist.get_backend() if dist.is_initialized() else None
    dev_index = (device.index or 0) if isinstance(device, torch.device) else int(device)
    on_gpu = torch.cuda.is_available() and not (isinstance(device, torch.device) and device.type != "cuda")
    me = {"rank": rank, "host": os.uname().nodename, "device_bdf": None, "numa_node": None, "pid": os.getpid()}
    if on_gpu:
        try:
            me["device_bdf"] = _gpu_bdf(dev_index)
            me["numa_node"] = gpu_numa_node(dev_index)
        except Exception:                                        # noqa: BLE001
            pass
    ranks = [me]
    if dist.is_initialized():
        ranks = [None] * dist.get_world_size()
        dist.all_gather_object(ranks, me)
    version = None
    if backend == "nccl":
        try:
            version = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:                                        # noqa: BLE001
            version = None
    seen = sum(1 for r in ranks if r is not None)
    out = {"backend": backend, "rccl_version": version, "world": world, "ranks_answered": seen, "ranks": ranks}
    if seen != world:
        raise RuntimeError(f"collective_identity: {seen} ranks answered, {world} expected: {ranks}")
    where = [(r["host"], r["device_bdf"]) for r in ranks if r["device_bdf"] is not None]
    out["distinct_devices"] = len(set(where))
    if len(set(where)) != len(where) and not allow_shared_device:
        raise RuntimeError(f"collective_identity: two ranks share a GPU ({where}); one rank per GPU is the contract "
                           "(PGV_BENCH_SHARE_DEVICE=1 allows it for control-flow checks)")
    return out


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


def _gpu_bdf(device_index: int):
    pr = torch.cuda.get_device_properties(device_index)
    return f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"


def gpu_numa_node(device_index: int, sysfs_root: str = "/sys"):
    """NUMA node of a GPU from sysfs (/sys/bus/pci/devices/<domain:bus:dev.fn>/numa_node); None when unknown (no sysfs entry, -1, no GPU)."""
    try:
        node = int(open(f"{sysfs_root}/bus/pci/devices/{_gpu_bdf(device_index)}/numa_node").read().strip())
        return node if node >= 0 else None
    except Exception:                                            # noqa: BLE001 -- placement is best effort by design
        return None


def node_gpu_bdfs(numa_node: int, sysfs_root: str = "/sys") -> List[str]:
    """PCI addresses of ALL AMD GPUs / accelerators of the host that hang off `numa_node`, sorted -- independent of which of them this job
    can see (HIP_VISIBLE_DEVICES, a second torchrun job on the same host), so two jobs that share a host derive disjoint core slices from it."""
    out = []
    base = f"{sysfs_root}/bus/pci/devices"
    for bdf in sorted(os.listdir(base)):
        try:
            d = f"{base}/{bdf}"
            if open(f"{d}/vendor").read().strip().lower() != "0x1002":
                continue
            cls = open(f"{d}/class").read().strip().lower()
            if not (cls.startswith("0x03") or cls.startswith("0x12")):      # display controllers / processing accelerators (MI300-class parts)
                continue
            if not bdf.endswith(".0") or int(open(f"{d}/numa_node").read().strip()) != numa_node:
                continue
            out.append(bdf)
        except Exception:                                        # noqa: BLE001
            continue
    return out


_pin_logged = False


def pin_rank_to_numa_node(local_rank: int, n_local: int | None = None, sysfs_root: str = "/sys", bdf: str | None = None, quiet: bool = False):
    """Bind this process (all threads started after the call inherit it -- call it BEFORE init_distributed so the backend's threads do too) to
    this rank's share of the cores of its GPU's NUMA node.  The share is picked by the GPU's position among ALL GPUs of that node in sysfs
    (node_gpu_bdfs), not by the per-job rank order: two jobs on one host get disjoint slices.  Logs the chosen mask and the ATen thread
    count once per process.  Falls back silently -- returns None and changes nothing -- when the topology cannot be read, the mask would be
    empty, or PGV_RANK_AFFINITY=0.  Returns the CPU list otherwise."""
    global _pin_logged
    if os.environ.get("PGV_RANK_AFFINITY", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        bdf = bdf or _gpu_bdf(local_rank)
        mine = int(open(f"{sysfs_root}/bus/pci/devices/{bdf}/numa_node").read().strip())
        if mine < 0:
            return None
        allowed = os.sched_getaffinity(0)
        cpus = [c for c in _parse_cpulist(open(f"{sysfs_root}/devices/system/node/node{mine}/cpulist").read()) if c in allowed]
        peers = node_gpu_bdfs(mine, sysfs_root)
        if bdf not in peers:
            return None
        share = affinity_slice(cpus, len(peers), peers.index(bdf))
        if not share:
            return None
        os.sched_setaffinity(0, share)
        threads = max(1, min(torch.get_num_threads(), len(share), 8))
        torch.set_num_threads(threads)
        if not _pin_logged and not quiet:
            _pin_logged = True
            import sys
            print(f"[pgv] local rank {local_rank}: GPU {bdf} on NUMA node {mine} ({peers.index(bdf) + 1} of {len(peers)} GPUs there) -> CPUs "
                  f"{share[0]}-{share[-1]} ({len(share)}), {threads} ATen threads (PGV_RANK_AFFINITY=0 disables)", file=sys.stderr, flush=True)     # stderr: bench.py's stdout is ONE JSON line
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
    world == 1 without a process group short-circuits without touching torch.distributed."""
    cap, width = tokens.shape
    assert cap == shard_capacity(n_items, world) and lengths.shape == (cap,)
    if world > 1 or (dist.is_available() and dist.is_initialized()):      # a one-rank process group still takes the collective path
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


def spill_file(spill_path: str, rank: int) -> str:
    return f"{spill_path}.rank{rank}.partial.json"


def run_sharded(n_items: int, infer_batch: Callable[..., tuple], max_new_tokens: int, rank: int, world: int,
                device, per_gpu_batch: int = 8, length_offset: int = 0, prepare: Callable[[Sequence[int]], object] | None = None,
                spill_path: str | None = None):
    """Run `infer_batch(indices) -> (tokens [len(indices), <=max_new] int tensor, lengths list)` over this rank's shard in
    groups of `per_gpu_batch`, then collate.  A group that raises keeps its slots with length 0.  `length_offset`: see gather_answers
    (the token count stored for slot j is lengths[j] - length_offset).
    `prepare(indices) -> obj` (optional) is the HOST half of a group (decode / sample the frames, pin them): it runs on a background thread
    ONE GROUP AHEAD of the device half, which then receives its result as `infer_batch(indices, obj)` -- the GPU works on group g while the
    host reads group g + 1.  An exception inside `prepare` is delivered to the group it belongs to (that group fails, the others go on).
    `spill_path` (world > 1): before the collective, this rank's finished answers go to `<spill_path>.rank<r>.partial.json` ({"indices",
    "tokens", "lengths", "length_offset"}), removed again once the all-gather has succeeded -- a peer that died or timed out then costs the
    collation, not the shard's work."""
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
    spilled = None
    if spill_path and world > 1:
        import json
        spilled = spill_file(spill_path, rank)
        lens_h, toks_h = lengths.cpu().tolist(), tokens.cpu()
        try:
            with open(spilled, "w") as f:
                json.dump({"rank": rank, "world": world, "indices": mine, "length_offset": length_offset, "lengths": lens_h[:len(mine)],
                           "tokens": [toks_h[j, :max(lens_h[j] - length_offset, 0)].tolist() for j in range(len(mine))]}, f)
        except OSError as e:                                     # disk full / read-only output dir: the safety net is gone, the collective still has
            import sys                                           # to run -- a rank that skipped it would park its peers for the whole timeout
            print(f"[rank {rank}] could not write the rank-local answer file {spilled}: {e}; continuing into the collation", file=sys.stderr, flush=True)
            spilled = None
    try:
        answers = gather_answers(tokens, lengths, n_items, rank, world, length_offset)
    except CollationError as e:
        if spilled:
            raise CollationError(f"{e}  This rank's answers are kept in {spilled}.") from e
        raise
    if spilled:
        try:
            os.remove(spilled)
        except OSError:
            pass
    return answers
