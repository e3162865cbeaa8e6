"""CPU-only: host logic and the C-ABI surface (no compute calls -- there is no GPU here)."""
import ctypes
import json
import os
import re
import socket
import subprocess
import time
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_every_header_symbol():
    from video_llava_amd import _lib, build
    path = build.build()
    lib = ctypes.CDLL(path)
    header = open(os.path.join(ROOT, "include", "pgv.h")).read()
    declared = set(re.findall(r"\b(pgv_[a-z0-9_]+)\s*\(", header))
    declared -= {"pgv_epi"}
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(lib, name), f"libpgv.so does not export {name}"
        assert name in _lib.PROTOTYPES, f"ctypes binding lacks {name}"
    assert set(_lib.PROTOTYPES) <= declared | {"pgv_version"}
    lib.pgv_version.restype = ctypes.c_int
    assert lib.pgv_version() == _lib.ABI_VERSION


def test_release_library_reads_five_switches_and_the_split_merge_lowers_to_sc1():
    """VERDICT r4 item 8 / ADVICE r4: the release libpgv.so reads exactly the five documented environment switches -- every
    launch-shape A/B switch lives behind -DPGV_LAB in libpgv_lab.so -- and the fence-free cross-workgroup merge of the split decode attention
    still lowers to sc1 stores / loads with no cache maintenance (build.check_isa)."""
    from video_llava_amd import build
    data = open(build.build(), "rb").read()
    names = set(re.findall(rb"\x00(PGV_[A-Z0-9_]{3,})\x00", data))        # NUL-delimited C strings: getenv names (PGV_F16 ... inside messages are not)
    assert names == {b"PGV_DATTN_SPLIT", b"PGV_LLM_NORM_FOLD", b"PGV_NO_GRAPH", b"PGV_VIT_LANES", b"PGV_VIT_LN_FOLD"}, names
    found = build.check_isa()
    split = {k: v for k, v in found.items() if k[0] != "vit_attn"}
    assert len(split) == 6 and all(v["cache_maintenance"] == 0 and v["atomics"] == 1 for v in split.values())
    # ... and the two pieces of inline asm of the ViT attention kernel still assemble as written: the half swap keeps its wait state,
    # every LDS-DMA sits directly behind the s_mov_b32 m0 that belongs to it
    attn = {k: v for k, v in found.items() if k[0] == "vit_attn"}
    assert len(attn) == 4 and all(v["half_swaps_with_wait_state"] >= 2 and v["lds_dma"] == v["lds_dma_with_m0"] >= 2 for v in attn.values()), attn


def test_product_path_fails_loudly_without_gpu():
    from video_llava_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.Context.get(0)
    from video_llava_amd.inference import get_spatio_temporal_features_torch
    with pytest.raises(RuntimeError, match="GPU"):
        get_spatio_temporal_features_torch(torch.zeros(2, 4, 1024, dtype=torch.float16))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "video_llava_amd")
    for dp, _dn, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{fn} imports the oracle"
                assert "/root/reference" not in src, f"{fn} reads the reference tree"


def test_prompts_match_reference(golden_dir):
    from video_llava_amd.video_conversation import conv_templates
    ref = json.load(open(os.path.join(golden_dir, "prompts.json")))
    for mode, want in ref.items():
        c = conv_templates[mode].copy()
        c.append_message(c.roles[0], "what is the person doing?\n<vid_start><vid_patch><vid_patch><vid_end>")
        c.append_message(c.roles[1], None)
        assert c.get_prompt() == want, mode


def test_build_prompt_layout():
    from video_llava_amd.inference import build_prompt
    p, stop = build_prompt("what is happening", "pg-video-llava", 356, True)
    assert stop == "</s>" and p.endswith("ASSISTANT:")
    assert p.count("<vid_patch>") == 356 and p.count("<vid_start>") == 1 and p.index("<vid_start>") < p.index("<vid_end>")
    p2, _ = build_prompt("q", "pg-video-llava", 4, True, transcript="hello")
    assert 'The noisy audio transcript of this video is:\n"hello"' in p2
    p3, stop3 = build_prompt("q", "default", 4, False)
    assert stop3 == "###" and "<vid_start>" not in p3


def test_get_seq_frames_golden(golden_dir):
    from video_llava_amd.eval.model_utils import get_seq_frames
    table = json.load(open(os.path.join(golden_dir, "seq_frames.json")))
    for key, ref in table.items():
        n, k = map(int, key.split(","))
        assert get_seq_frames(n, k) == ref


def test_keywords_stopping_criteria_semantics():
    from video_llava_amd.model.utils import KeywordsStoppingCriteria

    class Tok:
        def __call__(self, s):
            class R: pass
            r = R(); r.input_ids = [1, 2] if s == "</s>" else [7]
            return r

        def batch_decode(self, ids, skip_special_tokens=True):
            return ["".join(chr(96 + int(t)) for t in row) for row in ids]
    ids = torch.tensor([[5, 6]])
    c = KeywordsStoppingCriteria(["cd"], Tok(), ids)
    assert c.keyword_ids == [7]
    assert c(ids, None) is False                      # first call only records the prompt length
    assert c(torch.tensor([[5, 6, 1]]), None) is False
    assert c(torch.tensor([[5, 6, 1, 7]]), None) is True          # single-id keyword matched on the last token
    c2 = KeywordsStoppingCriteria(["</s>"], Tok(), ids)
    assert c2.keyword_ids == []                                      # 2-id keyword is dropped, falls back to the substring search
    c3 = KeywordsStoppingCriteria(["cd"], Tok(), ids); c3(ids, None)
    assert c3(torch.tensor([[5, 6, 3, 4]]), None) is True           # "cd" appears in the decoded tail


def test_shard_indices_cover_and_order():
    from video_llava_amd import parallel
    for n, w in ((64, 8), (10, 4), (3, 8), (0, 2), (7, 1)):
        got = [i for r in range(w) for i in parallel.shard_indices(n, r, w)]
        assert got == list(range(n))
        assert max(len(parallel.shard_indices(n, r, w)) for r in range(w)) == (parallel.shard_capacity(n, w) if n else 0)


def test_gather_answers_single_rank():
    from video_llava_amd import parallel
    toks = torch.arange(12, dtype=torch.int32).view(3, 4)
    lens = torch.tensor([4, 0, 2], dtype=torch.int32)
    out = parallel.gather_answers(toks, lens, 3, 0, 1)
    assert out == [[0, 1, 2, 3], [], [8, 9]]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_dp_collation_world2_gloo(tmp_path):
    """world_size-2 run of the shard + all-gather collation on CPU (gloo), incl. a failing clip and a ragged shard."""
    script = tmp_path / "dp.py"
    script.write_text(f'''
import sys, json, torch
sys.path.insert(0, {ROOT!r})
from video_llava_amd import parallel
rank, world, local = parallel.init_distributed("gloo")
N, NEW = 5, 6
def infer(group):
    if 3 in group: raise RuntimeError("bad clip")
    toks = torch.stack([torch.arange(NEW, dtype=torch.int32) + 100 * i for i in group])
    return toks, [NEW - (i % 3) for i in group]
ans = parallel.run_sharded(N, infer, NEW, rank, world, torch.device("cpu"), per_gpu_batch=1, spill_path={str(tmp_path / "preds")!r})
open({str(tmp_path)!r} + f"/result_{{rank}}.json", "w").write(json.dumps(ans))   # files, not stdout: rank lines interleave
torch.distributed.destroy_process_group()
''')
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    results = {rk: json.load(open(tmp_path / f"result_{rk}.json")) for rk in (0, 1) if (tmp_path / f"result_{rk}.json").exists()}
    assert set(results) == {0, 1} and results[0] == results[1]
    ans = results[0]
    want = [[100 * i + t for t in range(6 - (i % 3))] for i in range(5)]
    want[3] = []                                                        # failed clip keeps an empty slot
    assert ans == want
    assert not list(tmp_path.glob("preds.rank*.partial.json"))          # the rank-local spill files are removed once the all-gather succeeded


def _spawn_ranks(script, world, extra_env=None, timeout=180):
    """Start `world` plain processes (no elastic agent: a dying rank must be noticed by OUR code, not by torchrun killing the group)."""
    port = _free_port()
    procs = []
    for rk in range(world):
        env = dict(os.environ, RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env))
    outs = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            pr.kill()
            out, _ = pr.communicate()
            out += "\n<<HUNG: killed by the test>>"
        outs.append((pr.returncode, out))
    return outs


def test_dp_rank_raising_mid_shard_keeps_its_slots_failed(tmp_path):
    """one rank's infer raises from its second group on (a GPU fault mid-shard): every rank must still finish, with exactly
    that rank's later slots marked failed (None: distinct from an empty answer) -- the skip-bad-clip policy of the reference's runner
    (video_chatgpt/eval/run_inference_qa_activitynet.py:74,103-104) applied per group."""
    script = tmp_path / "raise.py"
    script.write_text(f'''
import sys, json, torch
sys.path.insert(0, {ROOT!r})
from video_llava_amd import parallel
rank, world, local = parallel.init_distributed("gloo", timeout_s=60)
N, NEW = 12, 4
calls = [0]
def infer(group):
    calls[0] += 1
    if rank == 1 and calls[0] >= 2: raise RuntimeError("device fault on rank 1")
    return torch.stack([torch.arange(NEW, dtype=torch.int32) + 10 * i for i in group]), [NEW + 1 for i in group]
ans = parallel.run_sharded(N, infer, NEW, rank, world, torch.device("cpu"), per_gpu_batch=2, length_offset=1)
open({str(tmp_path)!r} + f"/r_{{rank}}.json", "w").write(json.dumps(ans))
torch.distributed.destroy_process_group()
''')
    outs = _spawn_ranks(script, 2)
    assert all(rc == 0 for rc, _ in outs), outs
    a0, a1 = (json.load(open(tmp_path / f"r_{rk}.json")) for rk in (0, 1))
    assert a0 == a1
    want = [[10 * i + t for t in range(4)] for i in range(12)]
    for i in (8, 9, 10, 11):                       # rank 1 owns clips 6..11; its first group (6, 7) succeeded
        want[i] = None
    assert a0 == want


def test_dp_rank_exiting_fails_the_survivor_loudly_not_forever(tmp_path):
    """A rank that EXITS before the collation (killed, out of memory): the survivor must not hang in the all-gather -- it raises
    parallel.CollationError well inside the process-group timeout (here 20 s) and exits non-zero."""
    script = tmp_path / "exit.py"
    script.write_text(f'''
import os, sys, time, torch
sys.path.insert(0, {ROOT!r})
from video_llava_amd import parallel
rank, world, local = parallel.init_distributed("gloo", timeout_s=20)
def infer(group):
    if rank == 1: os._exit(7)
    return torch.zeros(len(group), 3, dtype=torch.int32), [3] * len(group)
t0 = time.time()
try:
    parallel.run_sharded(4, infer, 3, rank, world, torch.device("cpu"), per_gpu_batch=2, spill_path={str(tmp_path / "preds")!r})
except parallel.CollationError as e:
    print("LOUD after %.1f s:" % (time.time() - t0), e); sys.exit(5)
print("collated?!"); sys.exit(0)
''')
    t0 = time.time()
    outs = _spawn_ranks(script, 2, timeout=150)
    assert outs[1][0] == 7
    assert outs[0][0] == 5 and "LOUD" in outs[0][1] and "HUNG" not in outs[0][1], outs[0]
    assert time.time() - t0 < 120
    # the survivor's finished shard is not lost with the collation -- its rank-local file holds it and the error names it
    kept = json.load(open(tmp_path / "preds.rank0.partial.json"))
    assert kept["indices"] == [0, 1] and kept["tokens"] == [[0, 0, 0], [0, 0, 0]] and kept["lengths"] == [3, 3] and "preds.rank0.partial.json" in outs[0][1]


def test_rank_affinity_slices_and_silent_fallback():
    """Per-rank CPU placement (parallel.pin_rank_to_numa_node): the slices of a node's CPU list are disjoint, contiguous and cover it; with
    no GPU / no sysfs entry the call changes nothing and returns None."""
    from video_llava_amd import parallel
    cpus = list(range(64, 128)) + list(range(192, 256))
    parts = [parallel.affinity_slice(cpus, 4, i) for i in range(4)]
    assert sum(parts, []) == cpus and [len(x) for x in parts] == [32, 32, 32, 32]
    parts = [parallel.affinity_slice(list(range(10)), 3, i) for i in range(3)]
    assert sum(parts, []) == list(range(10)) and [len(x) for x in parts] == [4, 3, 3]
    assert parallel.affinity_slice([1, 2], 4, 3) == [1, 2]                      # fewer CPUs than ranks: share them
    assert parallel._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    before = os.sched_getaffinity(0)
    assert parallel.pin_rank_to_numa_node(0, 1) is None                          # no GPU in the build container
    assert os.sched_getaffinity(0) == before


def test_rank_affinity_from_gpu_position_in_sysfs(tmp_path):
    """the core slice comes from the GPU's position among ALL GPUs of its NUMA node in sysfs, not from the per-job rank order -- two
    jobs on one host (each seeing 4 of 8 GPUs) pin disjoint slices that together cover each node.  Fake sysfs: 8 GPUs, 4 per node, plus a
    non-GPU PCI function and another vendor's display controller that must be ignored; the CPU lists are cut to the cores this process may use."""
    from video_llava_amd import parallel
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 8:
        pytest.skip("needs 8 usable cores")
    half = len(allowed) // 2
    node_cpus = [allowed[:half], allowed[half:]]
    root = tmp_path / "sys"
    bdfs = [f"0000:{0x10 + 0x10 * i:02x}:00.0" for i in range(8)]
    def dev(bdf, vendor, cls, node):
        d = root / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True)
        (d / "vendor").write_text(vendor + "\n"); (d / "class").write_text(cls + "\n"); (d / "numa_node").write_text(f"{node}\n")
    for i, b in enumerate(bdfs):
        dev(b, "0x1002", "0x120000", i // 4)
    dev("0000:05:00.0", "0x8086", "0x030000", 0)                                 # another vendor's display controller
    dev("0000:10:00.1", "0x1002", "0x040300", 0)                                 # audio function of a GPU
    for n in (0, 1):
        d = root / "devices" / "system" / "node" / f"node{n}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(",".join(str(c) for c in node_cpus[n]) + "\n")
    assert parallel.node_gpu_bdfs(0, str(root)) == bdfs[:4] and parallel.node_gpu_bdfs(1, str(root)) == bdfs[4:]
    before = os.sched_getaffinity(0)
    try:
        got = {}
        for job, visible in (("A", [0, 2, 4, 6]), ("B", [1, 3, 5, 7])):              # two jobs, interleaved GPUs; local rank k of BOTH jobs used to get slice k
            for local, g in enumerate(visible):
                os.sched_setaffinity(0, before)
                share = parallel.pin_rank_to_numa_node(local, sysfs_root=str(root), bdf=bdfs[g], quiet=True)
                assert share and os.sched_getaffinity(0) == set(share)
                got[g] = share
        for n in (0, 1):
            parts = [got[g] for g in range(4 * n, 4 * n + 4)]
            assert sorted(sum(parts, [])) == node_cpus[n], (n, parts)               # disjoint and covering, across the two jobs
    finally:
        os.sched_setaffinity(0, before)


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_launcher_dry_world2(launcher):
    """`python bench.py --gpus 2` with no torchrun environment must start its own two ranks (the driver's command shape) and print
    ONE rank-0 JSON line with n_gpus = 2 and the collective report; the torchrun form must keep working.  --dry: real sharding,
    barriers, all-gather (gloo) and max-over-ranks timing, no GPU work."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--steps", "3", "--warmup", "1", "--clips-per-gpu", "3", "--new-tokens", "5"]
    if launcher == "self":
        cmd = [sys.executable, *tail]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), *tail]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak"
    assert d["collective"]["backend"] == "gloo" and d["collective"]["world_seen"] == 2 and d["collective"]["gather_ms"] > 0
    assert d["clips_checked"] == 6
    # the identity record of the process group (what proves, on an N-GPU node, that N ranks on N distinct devices took part): gathered over the group
    c = d["collective"]
    assert c["ranks_answered"] == 2 and c["world"] == 2 and [r["rank"] for r in c["ranks"]] == [0, 1] and len({r["pid"] for r in c["ranks"]}) == 2
    assert all(set(r) >= {"host", "device_bdf", "numa_node"} for r in c["ranks"]) and c["rccl_version"] is None and c["distinct_devices"] == 0


@pytest.mark.parametrize("n", [4, 8])
def test_bench_launcher_dry_world_4_8(n):
    """The driver's scaling tier runs N = 1, 2, 4, 8: the same skeleton at 4 and 8 ranks (gloo, --dry) -- every clip of every rank lands in its
    global slot on rank 0 and the per-rank step times are reported for all N ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry", "--steps", "2", "--warmup", "1", "--clips-per-gpu", "2", "--new-tokens", "4"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["config"]["parallelism"] == f"dp{n}" and d["scaling"] == "weak"
    assert d["collective"]["world_seen"] == n and d["clips_checked"] == 2 * n
    assert len(d["collective"].get("rank_ms_per_step", [0] * n)) == n


def test_collective_identity_refuses_shared_devices_and_parses_the_rccl_log(tmp_path, monkeypatch):
    """parallel.collective_identity: two ranks of one host on one GPU are refused unless the caller allows it (bench.py: PGV_BENCH_SHARE_DEVICE=1);
    parallel.rccl_transport: the channel lines of an RCCL INFO log are counted into a P2P / SHM / NET verdict."""
    from video_llava_amd import parallel
    import torch.distributed as dist

    class FakeDist:
        @staticmethod
        def is_initialized():
            return True

        @staticmethod
        def get_backend():
            return "gloo"

        @staticmethod
        def get_world_size():
            return 2

        @staticmethod
        def all_gather_object(out, me):
            out[0] = dict(me, rank=0, device_bdf="0000:05:00.0")
            out[1] = dict(me, rank=1, device_bdf="0000:05:00.0", pid=me["pid"] + 1)
    monkeypatch.setattr(parallel, "dist", FakeDist)
    with pytest.raises(RuntimeError, match="share a GPU"):
        parallel.collective_identity(torch.device("cpu"), 0, 2)
    ok = parallel.collective_identity(torch.device("cpu"), 0, 2, allow_shared_device=True)
    assert ok["ranks_answered"] == 2 and ok["distinct_devices"] == 1
    with pytest.raises(RuntimeError, match="2 ranks answered, 3 expected"):
        parallel.collective_identity(torch.device("cpu"), 0, 3, allow_shared_device=True)
    monkeypatch.setattr(parallel, "dist", dist)
    log = tmp_path / "rccl.log"
    log.write_text("vm:1:1 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC comm 0x1 nRanks 02\n"
                   "vm:1:1 [0] NCCL INFO Channel 01/0 : 1[1] -> 0[0] via P2P/IPC comm 0x1 nRanks 02\n"
                   "vm:1:1 [0] NCCL INFO === System : maxBw 48.0 totalBw 48.0 === XGMI\n")
    monkeypatch.setattr(parallel, "_rccl_debug_file", str(log))
    t = parallel.rccl_transport(cleanup=False)
    assert t["p2p"] == 2 and t["shm"] == 0 and t["net"] == 0 and t["verdict"] == "P2P" and t["xgmi_mentions"] == 1 and t["log_lines"] == 3
    log.write_text(log.read_text() + "vm:1:1 [0] NCCL INFO Channel 00 : 0[0] -> 1[1] via SHM/direct/direct\n")
    assert parallel.rccl_transport(cleanup=False)["verdict"] == "SHM"


def test_chat_turn_bookkeeping_follows_the_reference_rules(capsys):
    """Host logic of the chat front end without a device (reference video_chatgpt/chat.py:90-104 add_text, :201-222 print_state / _post_process_code):
    the first turn of a clip is cut at 1200 characters, gets the `<video>` marker once and carries the clip path; later turns are cut at 1536 and
    carry nothing; every turn queues a (user, text) and an open (assistant, None) message; escapes inside paired code fences are undone."""
    from types import SimpleNamespace as NS
    from video_llava_amd.chat import VideoChatGPTInterface
    from video_llava_amd.video_conversation import default_conversation
    model = NS(get_model=lambda: NS(vision_config=NS(use_vid_start_end=True)))
    iface = VideoChatGPTInterface(None, None, components=(model, None, None, NS(crop_size={"height": 224, "width": 224}), 356))
    assert iface.replace_token.count("<vid_patch>") == 356 and iface.replace_token.startswith("<vid_start>") and iface.replace_token.endswith("<vid_end>")
    iface.add_text("q" * 2000, "clip.mp4")
    (role_u, first), (role_a, open_slot) = iface.state.messages[-2:]
    assert (role_u, role_a) == tuple(default_conversation.roles[:2]) and open_slot is None
    assert first == ("q" * 1200 + "\n<video>", "clip.mp4") and iface.state.skip_next is False
    iface.first_run = False                                      # what answer() does after the first turn
    iface.add_text("<video> again " + "r" * 2000, "clip.mp4")
    assert iface.state.messages[-2][1] == ("<video> again " + "r" * 2000)[:1536] and iface.state.messages[-1][1] is None
    n0 = len(default_conversation.messages)                      # the template's own example exchange
    assert len(iface.state.messages) == n0 + 4
    iface.clear_history()
    iface.add_text("what is <video> about", None)                # the marker is not added twice
    assert iface.state.messages[-2][1] == ("what is <video> about", None) and len(iface.state.messages) == n0 + 2
    iface.print_state()
    out = capsys.readouterr().out
    assert out.startswith("SYSTEM: " + str(default_conversation.system)) and f"{default_conversation.roles[0]}: what is <video> about" in out
    post = VideoChatGPTInterface._post_process_code
    assert post("no fences \\_") == "no fences \\_"
    assert post("a\n```py\nx\\_y\n```\nb\\_") == "a\n```py\nx_y\n```\nb\\_"
    assert post("a\n```py\nx\\_y") == "a\n```py\nx\\_y"        # an unpaired fence: untouched


def test_kv_reuse_prefix_rules():
    """generate(kv_reuse_key=...) (round 6; a later chat turn prefills only what is behind the cached prefix, reference chat.py:108-160 re-runs the
    whole conversation): which prefix of the cache a new prompt may start from -- host logic, checked here on a stub without a device."""
    from video_llava_amd.model.video_chatgpt import VideoChatGPTLlamaForCausalLM as M, VisionConfig

    class Stub:
        pass
    m = Stub()
    m.model = Stub()
    vc = m.model.vision_config = VisionConfig()
    vc.vid_patch_token, vc.vid_start_token, vc.vid_end_token = 900, 901, 902
    kvh = object()
    cached = [1, 7, 8, 901, 900, 900, 902, 9, 10, 11, 40, 41, 42]        # prompt (with the video run) + the answer tokens that were fed back
    m._kv = {(1, 64): kvh}
    m._reuse = ("clip-a", kvh, 64, cached)
    f = lambda key, ids, new=8: M._reusable_prefix(m, key, ids, new)
    assert f("clip-a", cached + [50, 51]) == len(cached)                  # the new prompt extends the cache: everything is kept
    assert f("clip-a", cached[:10] + [60, 61, 62]) == 10                   # the answer came back re-tokenised: common prefix only
    assert f("clip-a", list(cached)) == len(cached) - 1                    # nothing new: one token still has to run (its logits pick the next)
    assert f("clip-b", cached + [50]) == 0                                 # another clip's conversation
    assert f("clip-a", [1, 7, 99] + cached[3:] + [50]) == 0                # diverges before the video run: the run would lie in the appended part
    assert f("clip-a", cached[:5]) == 0                                    # cut inside the video run
    assert f("clip-a", cached + [50], new=64) == 0                         # would outgrow the cache
    m._kv = {(1, 64): object()}
    assert f("clip-a", cached + [50]) == 0                                 # the cache the record points at is gone
    m._reuse = None
    assert f("clip-a", cached + [50]) == 0


@pytest.mark.parametrize("rel", ["bench.py", "video_llava_amd/benchlib.py", "__graft_entry__.py"])
def test_measurement_scripts_have_no_undefined_names(rel):
    """bench.py's CPU-baseline child only runs at the end of a GPU bench (it builds a 27 GB fp32 model), so a name that went missing in a refactor
    -- `make_prompts` after the measurement library was split out of the script, round 6 -- shows up as `cpu_baseline.value = null` on the
    driver's box and nowhere else.  Every name a function of these files loads must be defined in the file, imported, or a builtin."""
    import ast
    import builtins
    tree = ast.parse(open(os.path.join(ROOT, rel)).read())
    defined = set(dir(builtins)) | {"__file__", "__name__"}
    for n in ast.walk(tree):
        if isinstance(n, (ast.FunctionDef, ast.ClassDef)):
            defined.add(n.name)
        elif isinstance(n, ast.Import):
            defined.update((a.asname or a.name).split(".")[0] for a in n.names)
        elif isinstance(n, ast.ImportFrom):
            defined.update(a.asname or a.name for a in n.names)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            defined.add(n.id)
        elif isinstance(n, ast.arg):
            defined.add(n.arg)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            defined.add(n.name)
    used = {n.id for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load)}
    assert not (used - defined), sorted(used - defined)


def test_bench_gpus_mismatch_is_loud():
    """A WORLD_SIZE that contradicts --gpus is an error, not a silently different run."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_release_library_has_no_lab_switches():
    """Switches that produce garbage by design (timing ablations) must not exist in the library the product loads: they are compiled
    only into libpgv_lab.so (-DPGV_LAB, video_llava_amd.build.build(lab=True)), which nothing under video_llava_amd/ binds."""
    from video_llava_amd import build
    lib = build.build()
    blob = open(lib, "rb").read()
    for needle in (b"PGV_GEMM_ABLATE", b"PGV_ATTN_ABLATE", b"ABLATE"):
        assert needle not in blob, f"{needle!r} found in {lib}"
    src = "".join(open(os.path.join(ROOT, "video_llava_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "video_llava_amd")) if f.endswith(".py") and f != "_lib.py" and f != "build.py")
    assert "use_lab_build" not in src and "libpgv_lab" not in src


def test_sanitizer_build_of_the_host_shim_builds_and_loads():
    """libpgv_ubsan.so -- UBSan (non-recoverable) + libstdc++ container assertions on the host code of every translation unit,
    release kernels -- must build for gfx950 and export the whole C ABI under its preloaded runtime (the GPU suite runs under it with
    scripts/sessions/r4_ubsan.sh: 198 passed, 0 reports, profiles/r04_a_ubsan_gpu_tests.txt).  Nothing in the product binds it."""
    from video_llava_amd import _lib, build
    lib = build.build_sanitizer()
    rt = build.sanitizer_runtime()
    code = ("import ctypes, sys; sys.path.insert(0, %r); from video_llava_amd import _lib; _lib.LIB_PATH = %r; l = _lib.load(build_if_missing=False); "
            "print('OK', l.pgv_version(), len(_lib.PROTOTYPES))" % (ROOT, lib))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, LD_PRELOAD=rt))
    assert r.returncode == 0 and r.stdout.startswith(f"OK {_lib.ABI_VERSION} "), (r.stdout, r.stderr[-2000:])
    src = "".join(open(os.path.join(ROOT, "video_llava_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "video_llava_amd")) if f.endswith(".py") and f != "build.py")
    assert "libpgv_ubsan" not in src and "build_sanitizer" not in src


def test_no_kernel_spills_to_scratch():
    """Every gfx950 kernel of libpgv must fit its registers: a spilled accumulator or DMA offset inside a GEMM/attention loop
    costs 2-3x (seen while building the 4-wave GEMM), so scratch use is a build failure, not a perf note."""
    from video_llava_amd import build
    build.build()
    llvm = "/opt/rocm/lib/llvm/bin"
    if not all(os.path.exists(os.path.join(llvm, t)) for t in ("clang-offload-bundler", "llvm-readelf", "llvm-objcopy")):
        pytest.skip("ROCm llvm tools not present")
    import tempfile
    checked = 0
    bad = []
    regs = {}
    with tempfile.TemporaryDirectory() as td:
        for src in build.sources_present():
            obj = os.path.join(build.OBJ, src.replace(".hip", ".o"))
            fat, co = os.path.join(td, "x.fatbin"), os.path.join(td, "x.co")
            subprocess.run([f"{llvm}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
            if not os.path.exists(fat) or os.path.getsize(fat) == 0:
                continue                                    # host-only translation unit
            r = subprocess.run([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
            assert r.returncode == 0, f"{src}: {r.stderr}"
            notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            names = re.findall(r"\.name:\s+(\S+)", notes)
            segs = [int(x) for x in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes)]
            spills = [int(x) for x in re.findall(r"\.vgpr_spill_count:\s+(\d+)", notes)]
            vgprs = [int(x) for x in re.findall(r"\.vgpr_count:\s+(\d+)", notes)]
            assert len(names) == len(segs) == len(spills) == len(vgprs), src
            checked += len(names)
            bad += [(n, sg, v) for n, sg, v in zip(names, segs, spills) if v > 0]
            regs.update(zip(names, vgprs))
            os.remove(fat)
    assert checked >= 20, f"only {checked} kernels found"
    assert not bad, f"kernels spilling VGPRs: {bad[:6]}"
    # Register budgets the launch shapes rely on (gfx950: 512 VGPRs per SIMD lane): a kernel that grows past its budget silently halves its
    # occupancy -- two waves per SIMD for the ViT attention (256), three workgroups of 8 waves per CU for the split decode attention (80:
    # all 640 workgroups of a 13B launch resident), one workgroup per CU with two waves per SIMD for the widest GEMV shapes (256).
    def worst(fragment):
        hit = {n: v for n, v in regs.items() if fragment in n}
        assert hit, fragment
        return max(hit.values())
    assert worst("vit_attn_kernel") <= 256
    assert worst("decode_attn_split_kernel") <= 80
    assert worst("decode_attn_kernelI") <= 128
    assert worst("gemv_mfma_kernel") <= 256 and worst("gemv_k8_kernel") <= 168


def test_gemm_tile_order_is_a_bijection(tmp_path):
    """The persistent GEMM's tile order (csrc/gemm.hip tile_coords_v: the W-resident order for wide outputs with an even tile-row count,
    the band order otherwise) must visit every output tile exactly once for every grid the path launches.  The function is not ported: its
    SOURCE TEXT is cut out of gemm.hip and compiled for the host (it is plain integer arithmetic), then brute-forced over the tile grids of
    the ViT passes (224 px: N = 257 tokens, 336 px: 577; 1 .. 1024 frames per lane; qkv 12 / fc1 16 / out_proj, fc2 4 tile columns), the
    LLaMA prefill projections (7B: 48 / 16 / 86; 13B: 60 / 20 / 108 tile columns) and every small grid."""
    import shutil
    gxx = shutil.which("g++")
    assert gxx, "g++ is part of the image"
    src = open(os.path.join(ROOT, "video_llava_amd", "csrc", "gemm.hip")).read()
    i = src.index("__device__ __forceinline__ void tile_coords_v")
    depth, j = 0, src.index("{", i)
    for j in range(j, len(src)):
        depth += src[j] == "{"
        depth -= src[j] == "}"
        if depth == 0:
            break
    fn = src[i:j + 1]
    assert "wn = ntn >> 2" in fn and "band" in fn            # both orders are in the cut
    prog = """#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define __device__
#define __forceinline__ inline
using std::min;
""" + fn + """
int main(int argc, char** argv) {
    long checked = 0;
    for (int a = 1; a + 1 < argc; a += 2) {
        const int ntm = atoi(argv[a]), ntn = atoi(argv[a + 1]), total = ntm * ntn;
        std::vector<char> seen(total, 0);
        for (int vb = 0; vb < total; ++vb) {
            int tm = -1, tn = -1;
            tile_coords_v(vb, total, ntm, ntn, tm, tn);
            if (tm < 0 || tm >= ntm || tn < 0 || tn >= ntn || seen[tm * ntn + tn]) { printf("BAD ntm=%d ntn=%d vb=%d -> (%d,%d)\\n", ntm, ntn, vb, tm, tn); return 1; }
            seen[tm * ntn + tn] = 1;
        }
        ++checked;
    }
    printf("OK %ld\\n", checked);
    return 0;
}
"""
    cpp, exe = tmp_path / "tiles.cpp", tmp_path / "tiles"
    cpp.write_text(prog)
    subprocess.run([gxx, "-O1", "-o", str(exe), str(cpp)], check=True)
    grids = set()
    for tokens in (257, 577):
        for frames in list(range(1, 130)) + [200, 256, 399, 400, 401, 512, 799, 800, 1000, 1024]:
            ntm = (frames * tokens + 255) // 256
            for ntn in (4, 12, 16):
                grids.add((ntm, ntn))
        for frames in range(1, 130):
            grids.add(((frames * (tokens - 1) + 255) // 256, 4))                 # patch-embed GEMM: M = frames x patches
    for rows in (1, 2, 441, 454, 3528, 7264):                                   # prefill: sum of prompt lengths (1 .. 16 sequences)
        for ntn in (48, 16, 86, 60, 20, 108, 1, 2, 3, 5, 8, 9, 17):
            grids.add(((rows + 255) // 256, ntn))
    for ntm in range(1, 41):
        for ntn in range(1, 41):
            grids.add((ntm, ntn))
    args = [str(x) for g in sorted(grids) for x in g]
    r = subprocess.run([str(exe), *args], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr
    assert (402, 12) in grids and (804, 16) in grids                           # the two grids bench.py times


# --------------------------------------------------------------------------------------------------
# offline feature extraction (scripts/save_spatio_temporal_clip_features.py) and the QA runner: host logic
# --------------------------------------------------------------------------------------------------
def test_extraction_cli_and_numpy_twin(tmp_path, golden_dir):
    from video_llava_amd import feature_extraction as fx
    a = fx.parse_args(["--llava", "1.5", "--video_dir_path", "v", "--clip_feat_path", "o"])
    assert a.infer_batch == 32 and fx.LLAVA_VERSIONS[a.llava][1] == (336, 336)
    with pytest.raises(SystemExit):
        fx.parse_args(["--llava", "2.0", "--video_dir_path", "v", "--clip_feat_path", "o"])
    # numpy twin == the golden produced by the reference's own function
    g = np.load(os.path.join(golden_dir, "pool.npz"))                      # outputs of the reference script's own function
    for name, (T, P) in {"t8_p16": (8, 16), "t100_p16": (100, 16), "t3_p4": (3, 4)}.items():
        rng = np.random.default_rng(int(g[name + "_seed"]))
        f16 = (rng.standard_normal((T, P, 1024), dtype=np.float32) * 1.5).astype(np.float16)
        assert np.array_equal(fx.get_spatio_temporal_features(f16), g[name + "_numpy"])
    x = np.random.default_rng(0).standard_normal((7, 16, 1024)).astype(np.float16)
    y = fx.get_spatio_temporal_features(x)
    assert y.dtype == np.float16 and y.shape == (116, 1024) and not y[7:100].any()
    assert np.array_equal(y[:7], np.mean(x, axis=1)) and np.array_equal(y[100:], np.mean(x, axis=0))


def test_extraction_load_video_npy_sampling_and_resize(tmp_path):
    from video_llava_amd import feature_extraction as fx
    rng = np.random.default_rng(1)
    clip = rng.integers(0, 256, (130, 60, 80, 3), dtype=np.uint8)
    p = tmp_path / "clip.npy"
    np.save(p, clip)
    out = fx.load_video(str(p), num_frm=100, shape=(224, 224))
    assert out.shape == (100, 224, 224, 3) and out.dtype == np.uint8
    idx = fx.get_seq_frames(130, 100)
    ref = torch.nn.functional.interpolate(torch.from_numpy(clip[idx]).permute(0, 3, 1, 2).float(), size=(224, 224))
    assert np.array_equal(out, ref.permute(0, 2, 3, 1).to(torch.uint8).numpy())          # the reference's nearest resize
    short = fx.load_video(str(p), num_frm=200, shape=(60, 80))
    assert short.shape == (130, 60, 80, 3) and np.array_equal(short, clip[fx.get_seq_frames(130, 130)])


def test_extraction_loop_skips_flushes_and_survives_failures(tmp_path):
    """run(): idempotent skip of existing .pkl, periodic flush, a broken clip is reported and skipped; pkl = np.float16 array."""
    import pickle
    from video_llava_amd import feature_extraction as fx
    vd, od = tmp_path / "videos", tmp_path / "feats"
    vd.mkdir(); od.mkdir()
    for name in ("a.npy", "b.npy", "c.npy", "bad.npy"):
        np.save(vd / name, np.zeros((3, 8, 8, 3), np.uint8))
    (vd / "bad.npy").write_bytes(b"not a numpy file")
    with open(od / "a.pkl", "wb") as f:
        pickle.dump(np.ones((2, 2), np.float16), f)
    calls = []
    orig = fx.extract_clip_features
    fx.extract_clip_features = lambda path, tower, size: (calls.append(os.path.basename(path)), np.load(path),
                                                          np.full((356, 1024), len(calls), np.float16))[2]
    try:
        logs = []
        args = fx.parse_args(["--llava", "1.1", "--video_dir_path", str(vd), "--clip_feat_path", str(od)])
        n = fx.run(args, vision_tower=object(), save_every=2, log=logs.append)
    finally:
        fx.extract_clip_features = orig
    assert n == 2 and calls == ["b.npy", "bad.npy", "c.npy"] and len(logs) == 1 and "bad.npy" in logs[0]
    assert pickle.load(open(od / "a.pkl", "rb")).shape == (2, 2)                        # untouched
    for key in ("b", "c"):
        arr = pickle.load(open(od / f"{key}.pkl", "rb"))
        assert isinstance(arr, np.ndarray) and arr.dtype == np.float16 and arr.shape == (356, 1024)
    assert not (od / "bad.pkl").exists()


def test_qa_runner_host_logic(tmp_path):
    from video_llava_amd.eval import run_inference_qa_activitynet as qa
    a = qa.parse_args(["--video_dir", "v", "--gt_file_question", "q", "--gt_file_answers", "a", "--output_dir", "o", "--output_name", "n",
                       "--model-name", "m", "--projection_path", "p"])
    assert a.conv_mode == "pg-video-llava" and not a.use_asr
    vd = tmp_path / "v"; vd.mkdir()
    (vd / "v_x1.mov").write_bytes(b""); (vd / "v_x1.mkv").write_bytes(b""); (vd / "v_y.npy").write_bytes(b"")
    assert qa.find_video(str(vd), "x1").endswith("v_x1.mov")           # reference order: mp4, avi, mov, mkv
    assert qa.find_video(str(vd), "y").endswith("v_y.npy") and qa.find_video(str(vd), "zz") is None
    q = [{"video_name": "x1", "question": "what?", "question_id": "x1_0"}, {"video_name": "zz", "question": "why?", "question_id": "zz_0"}]
    an = [{"answer": "yes"}, {"answer": "no"}]
    (tmp_path / "q.json").write_text(json.dumps(q)); (tmp_path / "a.json").write_text(json.dumps(an))
    samples = qa.load_samples(str(tmp_path / "q.json"), str(tmp_path / "a.json"))
    assert samples[1] == {"video_name": "zz", "question": "why?", "id": "zz_0", "answer": "no"}
    out = qa.build_output(samples, ["a dog", None])
    assert out == [{"id": "x1_0", "question": "what?", "answer": "yes", "pred": "a dog"}]


def test_sibling_runners_host_logic(tmp_path, monkeypatch):
    """Dataset adapters of the other eval runners (file naming, sample / output schema, drop-on-failure policy) with the sharded
    answering core stubbed out: reference video_chatgpt/eval/run_inference_{benchmark_general,benchmark_consistency,qa_msrvtt,qa_msvd,qa_tgif}.py."""
    from PIL import Image
    from video_llava_amd.eval import _sharded
    from video_llava_amd.eval import run_inference_benchmark_consistency as cons
    from video_llava_amd.eval import run_inference_benchmark_general as gen
    from video_llava_amd.eval import run_inference_qa_msrvtt as msrvtt
    from video_llava_amd.eval import run_inference_qa_msvd as msvd
    from video_llava_amd.eval import run_inference_qa_tgif as tgif

    class _IP:
        crop_size = {"height": 224, "width": 224}
    seen = {}

    def fake_setup(args, components=None):
        return 0, 1, (None, None, None, _IP(), 356)

    def fake_answer(args, tasks, components, load_frames, rank, world):
        seen["tasks"] = tasks
        return [None if t["path"] is None else f"ans:{t['question']}" for t in tasks]
    monkeypatch.setattr(_sharded, "setup", fake_setup)
    monkeypatch.setattr(_sharded, "answer_tasks", fake_answer)
    vd = tmp_path / "v"; vd.mkdir()
    out = tmp_path / "o"
    common = ["--video_dir", str(vd), "--output_dir", str(out), "--output_name", "r", "--model-name", "m", "--projection_path", "p"]

    # general + consistency: {video_name}.{mp4,avi,mov,mkv}; a sample with any failed question is dropped
    (vd / "a.avi").write_bytes(b""); (vd / "a.mkv").write_bytes(b"")
    gt = [{"video_name": "a", "Q": "q?", "A": "x", "Q1": "q1?", "Q2": "q2?"}, {"video_name": "missing", "Q": "z?", "A": "y", "Q1": "m1", "Q2": "m2"}]
    (tmp_path / "gt.json").write_text(json.dumps(gt))
    r = gen.run_inference(gen.parse_args(common + ["--gt_file", str(tmp_path / "gt.json")]))
    assert seen["tasks"][0]["path"].endswith("a.avi") and seen["tasks"][1]["path"] is None
    assert r == [dict(gt[0], pred="ans:q?")] and json.load(open(out / "r.json")) == r
    r = cons.run_inference(cons.parse_args(common + ["--gt_file", str(tmp_path / "gt.json")]))
    assert [t["question"] for t in seen["tasks"]] == ["q1?", "q2?", "m1", "m2"]
    assert r == [dict(gt[0], pred1="ans:q1?", pred2="ans:q2?")]

    # MSRVTT: video{video_id}.mp4; MSVD: mapper lines "<name> vid<id>" and .avi
    (vd / "video7.mp4").write_bytes(b""); (vd / "clipA.avi").write_bytes(b"")
    gt2 = [{"video_id": 7, "question": "who?", "answer": "man", "id": 1}, {"video_id": 8, "question": "what?", "answer": "dog", "id": 2}]
    (tmp_path / "gt2.json").write_text(json.dumps(gt2))
    r = msrvtt.run_inference(msrvtt.parse_args(common + ["--gt_file", str(tmp_path / "gt2.json")]))
    assert seen["tasks"][0]["path"].endswith("video7.mp4") and r == [dict(gt2[0], pred="ans:who?")]
    (tmp_path / "map.txt").write_text("clipA vid7\nclipB vid8\n")
    a = msvd.parse_args(common + ["--gt_file", str(tmp_path / "gt2.json"), "--mapper", str(tmp_path / "map.txt")])
    assert msvd.load_mapper(a.mapper) == {7: "clipA", 8: "clipB"}
    r = msvd.eval_model(a)
    assert seen["tasks"][0]["path"].endswith("clipA.avi") and seen["tasks"][1]["path"] is None and r == [dict(gt2[0], pred="ans:who?")]

    # TGIF: tab-separated rows, {gif_name}.gif, 8 segment-centred frames at native size
    assert tgif.gif_frame_indices(40) == [2, 7, 12, 17, 22, 26, 31, 36] and tgif.gif_frame_indices(3) == [0, 0, 0, 1, 1, 1, 2, 2]
    frames = [Image.fromarray(np.full((20, 30, 3), 10 * i, dtype=np.uint8)) for i in range(12)]
    frames[0].save(vd / "g1.gif", save_all=True, append_images=frames[1:], duration=50, loop=0)
    imgs = tgif.load_video_from_gif(str(vd / "g1.gif"))
    assert len(imgs) == 8 and imgs[0].size == (30, 20) and imgs[0].mode == "RGB"
    (tmp_path / "t.tsv").write_text("gif_name\tquestion\tdescription\tanswer\ng1\thow many?\ta cat\t2\ng9\twhere?\ta dog\tpark\n")
    r = tgif.run_inference(tgif.parse_args(common + ["--gt_file", str(tmp_path / "t.tsv")]))
    assert r == [{"gif_name": "g1", "question": "how many?", "description": "a cat", "answer": "2", "pred": "ans:how many?"}]


def test_first_stop_length_equals_per_token_criterion_loop():
    """_sharded's batched path cuts an answer where the reference's per-token loop would have stopped: first_stop_length must agree with
    driving KeywordsStoppingCriteria token by token (HF calls it after every appended token; its first call only records the start)."""
    from video_llava_amd.model.utils import KeywordsStoppingCriteria, first_stop_length

    class Tok:
        def __call__(self, s):
            class R: pass
            r = R(); r.input_ids = [1] + [ord(c) for c in s]
            return r

        def batch_decode(self, ids, skip_special_tokens=True):
            return ["".join(chr(int(t)) for t in row if int(t) > 2) for row in ids]

    tok = Tok()
    rng = np.random.default_rng(0)
    prompt = torch.tensor([[1, 70, 71]])
    for trial in range(300):
        n = int(rng.integers(1, 30))
        toks = rng.choice([ord("a"), ord("b"), ord("#"), ord(" ")], n, p=[0.4, 0.3, 0.2, 0.1]).tolist()
        stop = ["###", "#", "ab#"][trial % 3]
        crit = KeywordsStoppingCriteria([stop], tok, prompt)
        want = None
        for i in range(1, n + 1):
            if crit(torch.tensor([prompt[0].tolist() + toks[:i]]), None):
                want = i
                break
        assert first_stop_length(toks, tok, [stop]) == want, (toks, stop)


def test_gather_answers_distinguishes_empty_from_failed():
    from video_llava_amd import parallel
    toks = torch.tensor([[5, 6, 7], [0, 0, 0], [0, 0, 0], [9, 0, 0]], dtype=torch.int32)
    lens = torch.tensor([4, 1, 0, 2], dtype=torch.int32)              # length + 1; 0 = failed
    out = parallel.gather_answers(toks, lens, 4, 0, 1, length_offset=1)
    assert out == [[5, 6, 7], [], None, [9]]


def test_nearest_resize_index_rule():
    """pgv_ingest_u8 implements F.interpolate(mode='nearest') as src = min(floorf(dst * (float)in / out), in - 1) per axis; pin that rule
    against torch itself on the shapes load_video meets (reference eval/model_utils.py:38-43), including up-sampling and odd ratios."""
    from video_llava_amd.feature_extraction import resize_nearest
    rng = np.random.default_rng(0)
    for (h, w, s) in ((360, 640, 224), (360, 640, 336), (20, 20, 56), (481, 853, 224), (224, 224, 224), (112, 112, 224), (1080, 1920, 336)):
        a = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
        got = resize_nearest(a, (s, s))
        sh, sw = np.float32(h) / np.float32(s), np.float32(w) / np.float32(s)
        iy = np.minimum(np.floor(np.arange(s, dtype=np.float32) * sh).astype(np.int64), h - 1)
        ix = np.minimum(np.floor(np.arange(s, dtype=np.float32) * sw).astype(np.int64), w - 1)
        assert np.array_equal(got, a[:, iy][:, :, ix]), (h, w, s)


def test_frame_directory_front_end(tmp_path):
    """load_video without decord: a directory of image files (one frame per file) is sampled with get_seq_frames exactly like a decoded
    video (reference eval/model_utils.py:27-36) and resized with the reference's nearest rule."""
    from PIL import Image
    from video_llava_amd import feature_extraction as fx
    from video_llava_amd.eval.model_utils import get_seq_frames
    rng = np.random.default_rng(3)
    clip = rng.integers(0, 256, (130, 24, 40, 3), dtype=np.uint8)
    d = tmp_path / "clip_frames"
    d.mkdir()
    for i, f in enumerate(clip):
        Image.fromarray(f).save(d / f"{i:06d}.png")                      # lossless
    (d / "notes.txt").write_text("ignored")
    native = fx.sample_frames(str(d))
    assert native.shape == (100, 24, 40, 3) and np.array_equal(native, clip[get_seq_frames(130, 100)])
    got = fx.load_video(str(d), shape=(14, 14))
    assert got.shape == (100, 14, 14, 3) and np.array_equal(got, fx.resize_nearest(native, (14, 14)))
    nf = fx.load_video(str(d), shape=(14, 14), device_resize=True)
    assert isinstance(nf, fx.NativeFrames) and nf.shape == (14, 14) and np.array_equal(nf.resized(), got)
    with pytest.raises(ValueError, match="no image files"):
        (tmp_path / "empty").mkdir()
        fx.sample_frames(str(tmp_path / "empty"))


def test_animated_gif_front_end(tmp_path):
    """The TGIF runner's files are .gif: without decord they are decoded by Pillow.  Frames are sampled with get_seq_frames and must equal an
    independent frame-by-frame decode of the same file (GIF itself is lossy: compare against the file, not against the source array)."""
    from PIL import Image
    from video_llava_amd import feature_extraction as fx
    from video_llava_amd.eval import _sharded
    from video_llava_amd.eval.model_utils import get_seq_frames
    rng = np.random.default_rng(5)
    src = [Image.fromarray(rng.integers(0, 256, (20, 28, 3), dtype=np.uint8)) for _ in range(120)]
    path = tmp_path / "tumblr_x.gif"
    src[0].save(path, save_all=True, append_images=src[1:], duration=40, loop=0)
    with Image.open(path) as im:
        assert im.n_frames == 120
        ref = []
        for i in range(120):
            im.seek(i)
            ref.append(np.asarray(im.convert("RGB")))
    ref = np.stack(ref)
    got = fx.sample_frames(str(path))
    assert got.shape == (100, 20, 28, 3) and np.array_equal(got, ref[get_seq_frames(120, 100)])
    small = fx.load_video(str(path), shape=(14, 14))
    assert small.shape == (100, 14, 14, 3) and np.array_equal(small, fx.resize_nearest(got, (14, 14)))
    # runner-side discovery: the reference's extension order first, then the decord-free forms (array, directory of frames)
    assert _sharded.first_existing(str(tmp_path), "tumblr_x", [".gif", *_sharded.DECORD_FREE_FORMATS]) == str(path)
    (tmp_path / "clip7").mkdir()
    assert _sharded.first_existing(str(tmp_path), "clip7", [".mp4", *_sharded.DECORD_FREE_FORMATS]) == str(tmp_path / "clip7")
    assert _sharded.first_existing(str(tmp_path), "absent", [".mp4", *_sharded.DECORD_FREE_FORMATS]) is None


def test_load_video_decord_branches_match_reference(golden_dir, tmp_path):
    """The decord branches of both `load_video` mirrors (reference video_chatgpt/eval/model_utils.py:12-52 and
    scripts/save_spatio_temporal_clip_features.py:13-32) executed through a stub `decord.VideoReader` (decord is not installable offline):
    frame sampling -> get_batch -> nearest resize -> PIL / uint8 array, byte-identical to what the REFERENCE's own functions returned
    for the same stub clips (tests/golden/load_video.npz, written by oracle/gen_golden.py::gen_load_video)."""
    from oracle import decord_stub
    from video_llava_amd import feature_extraction as fx
    from video_llava_amd.eval import model_utils as mu
    g = np.load(os.path.join(golden_dir, "load_video.npz"))
    decord_stub.install()
    try:
        for i, (path, shape) in enumerate(zip(g["paths"].tolist(), g["shapes"].tolist())):
            want = g[f"case{i}"]
            shape = tuple(int(x) for x in shape)
            pil = mu.load_video(path, shape=shape)                           # eval/model_utils.py mirror: list of PIL images
            assert len(pil) == want.shape[0] and all(im.size == (shape[1], shape[0]) and im.mode == "RGB" for im in pil)
            assert np.array_equal(np.stack([np.asarray(im) for im in pil]), want), path
            arr = fx.load_video(path, shape=shape)                           # extraction-script mirror: uint8 [k, h, w, 3]
            assert arr.dtype == np.uint8 and np.array_equal(arr, want), path
            nat = fx.load_video(path, shape=shape, device_resize=True)       # runner front end: native frames, resized on the device later
            assert isinstance(nat, fx.NativeFrames) and np.array_equal(nat.resized(), want), path
        # a real file path goes through the same branch (the stub reads .npy): the eval runners' route for v_<name>.mp4
        clip = np.random.default_rng(9).integers(0, 256, (12, 9, 11, 3), dtype=np.uint8)
        p = tmp_path / "v_x.npy"
        np.save(p, clip)
        os.rename(p, tmp_path / "v_x.mp4")
        got = fx.sample_frames(str(tmp_path / "v_x.mp4"))
        assert np.array_equal(got, clip[mu.get_seq_frames(12, 12)])          # note: not the identity -- the reference's midpoint rule repeats frames
    finally:
        decord_stub.uninstall()
    with pytest.raises(RuntimeError, match="decord"):                        # without decord the error names what is missing
        mu.load_video("synth:3x4x4:1")


def test_run_sharded_prefetches_one_group_ahead():
    """parallel.run_sharded with a `prepare` half: the host half of group g + 1 runs on a background thread while the device half of group g
    runs; results and failure isolation are those of the serial loop (a prepare() exception fails only its own group)."""
    import threading
    import time
    from video_llava_amd import parallel
    events, lock = [], threading.Lock()

    def log(*e):
        with lock:
            events.append(e)

    def prepare(group):
        log("prep_start", tuple(group), threading.current_thread().name)
        time.sleep(0.05)
        if 4 in group:
            raise IOError("corrupt clip")
        log("prep_end", tuple(group))
        return {i: i * 10 for i in group}

    def infer(group, prepared):
        log("infer_start", tuple(group), threading.current_thread().name)
        time.sleep(0.1)
        toks = torch.tensor([[prepared[i], prepared[i] + 1, 0] for i in group], dtype=torch.int32)
        log("infer_end", tuple(group))
        return toks, [3 if i % 2 else 2 for i in group]           # length_offset 1: answers of 2 / 1 tokens

    ans = parallel.run_sharded(7, infer, 3, 0, 1, torch.device("cpu"), per_gpu_batch=2, length_offset=1, prepare=prepare)
    assert ans == [[0], [10, 11], [20], [30, 31], None, None, [60]]            # group (4, 5) failed in prepare; the rest is intact
    names = [e[0:2] for e in events]
    # overlap: the preparation of group 1 starts before the inference of group 0 ends, on another thread
    assert names.index(("prep_start", (2, 3))) < names.index(("infer_end", (0, 1)))
    main = threading.current_thread().name
    assert all(e[2] != main for e in events if e[0] == "prep_start") and all(e[2] == main for e in events if e[0] == "infer_start")
    # never more than one group ahead
    assert names.index(("prep_start", (4, 5))) > names.index(("infer_start", (0, 1)))


def test_first_stop_length_survives_a_non_monotone_decoder():
    """`keyword in decode(tokens[:n])` need not be monotone (byte-fallback pieces decode to U+FFFD until complete).  The search
    result is verified and a linear scan takes over: the answer must equal the reference's per-token loop for such a decoder too."""
    from video_llava_amd.model.utils import first_stop_length

    class Tok:
        """Token 9 followed by token 8 decodes to 'ab'; a trailing lone 9 decodes to U+FFFD (an incomplete byte-fallback piece); 5 erases the
        previous character only when it directly follows 8 at the END of the sequence (a decoder that strips a trailing piece)."""
        def __call__(self, s):
            class R: pass
            r = R(); r.input_ids = [1, 2, 3]; return r
        def batch_decode(self, rows, skip_special_tokens=True):
            out = []
            for row in rows:
                t = ""
                for i, x in enumerate(row):
                    if x == 9:
                        t += "a" if i + 1 < len(row) else "�"
                    elif x == 8:
                        t += "b"
                    elif x == 5:
                        t = t[:-1] if i + 1 == len(row) else t + "y"
                    else:
                        t += "x"
                out.append(t)
            return out

    def per_token(toks, kw):                       # the reference's loop: stop at the first n >= 2 whose decoded prefix holds the keyword
        tk = Tok()
        return next((n for n in range(2, len(toks) + 1) if kw in tk.batch_decode([toks[:n]])[0]), None)

    for toks in ([7, 9, 8, 7, 7], [7, 7, 9, 8, 5, 7, 7, 7], [7, 9, 8, 5], [7, 7, 7], [7, 9], [9, 8], [7, 7, 7, 9, 8, 9], [9, 8, 9, 8, 9]):
        assert first_stop_length(toks, Tok(), ["ab"]) == per_token(toks, "ab"), toks


def test_device_resize_upload_is_bounded(tmp_path):
    """the native-resolution upload of the runners (`device_resize=True`) is capped; above the cap the frames are resized on the host."""
    from video_llava_amd import feature_extraction as fx
    small = np.zeros((4, 48, 64, 3), np.uint8); np.save(tmp_path / "s.npy", small)
    assert isinstance(fx.load_video(str(tmp_path / "s.npy"), shape=(14, 14), device_resize=True), fx.NativeFrames)
    old = fx.DEVICE_RESIZE_MAX_BYTES
    fx.DEVICE_RESIZE_MAX_BYTES = small.nbytes - 1
    try:
        out = fx.load_video(str(tmp_path / "s.npy"), shape=(14, 14), device_resize=True)
        assert isinstance(out, np.ndarray) and out.shape == (4, 14, 14, 3)
    finally:
        fx.DEVICE_RESIZE_MAX_BYTES = old


def test_runner_batch_auto_picks_the_widest_group_that_fits():
    """`--batch auto` (the runners' default since round 5): the largest of 8 / 16 / 32 / 64 clips per group whose KV cache + tower workspace
    fit 70 % of the free device memory; explicit values still parse, out-of-range ones are rejected."""
    import argparse
    from video_llava_amd.eval import _sharded
    from video_llava_amd.model.video_chatgpt import VideoChatGPTConfig
    c7 = VideoChatGPTConfig(vocab_size=32003)
    c13 = VideoChatGPTConfig(vocab_size=32003, hidden_size=5120, intermediate_size=13824, num_hidden_layers=40, num_attention_heads=40)
    GiB = 2 ** 30
    need = [_sharded.batch_bytes(b, c7, 224, 1024) for b in (8, 16, 32, 64)]
    assert need == sorted(need) and 4 * GiB < need[0] < 20 * GiB and 40 * GiB < need[3] < 160 * GiB, [n / GiB for n in need]
    assert _sharded.pick_batch(c7, 224, 1024, 270 * GiB) == 64                 # a free MI355X next to 13.5 GB of weights
    assert _sharded.pick_batch(c7, 224, 1024, int(need[2] / 0.7) + GiB) == 32
    assert _sharded.pick_batch(c7, 224, 1024, 8 * GiB) == 8                    # never below 8: a too-small GPU fails in the allocation, loudly
    assert _sharded.batch_bytes(64, c13, 336, 1024) > _sharded.batch_bytes(64, c7, 224, 1024)
    p = _sharded.add_runtime_arguments(argparse.ArgumentParser())
    assert p.parse_args([]).batch == "auto" and p.parse_args(["--batch", "16"]).batch == 16
    with pytest.raises(SystemExit):
        p.parse_args(["--batch", "65"])


def test_generate_stop_string_ahead_of_eos_in_the_same_chunk():
    """`generate(stop_strings=...)` at B = 1 whose chunk ends with the sequence at EOS must still cut at a stop string that appears
    BEFORE that EOS in the same chunk (the reference's per-token loop, model/utils.py:6-26, stops at the string first).  Host logic only: the
    device calls (prefill / decode_greedy) are scripted, so this runs without a GPU."""
    import torch.nn as nn
    from video_llava_amd.model.video_chatgpt import VideoChatGPTConfig, VideoChatGPTLlamaForCausalLM
    script = [11, 12, 13, 14, 15, 16, 17, 18] + [19] * 60             # 14 decodes to the stop string, 16 plays EOS: both inside the first chunk of 32
    STOP_ID, EOS = 14, 16

    class Tok:
        def __call__(self, text):                                     # "###" is not a single-id keyword: only the decoded-tail branch can see it
            class R:
                input_ids = [1, 40, 41, 42]
            return R()

        def batch_decode(self, ids, skip_special_tokens=True):
            return [" ".join("###" if int(t) == STOP_ID else str(int(t)) for t in row) for row in ids]

    def fake_model(eos):
        m = object.__new__(VideoChatGPTLlamaForCausalLM)
        nn.Module.__init__(m)
        m.config = VideoChatGPTConfig(eos_token_id=eos)
        m.device_ = torch.device("cpu")
        cur = {"i": 0, "done": False}

        def prefill(seqs, feats, max_seq, want_logits=False):
            cur.update(i=1, done=script[0] == eos)
            return None, torch.tensor([script[0]], dtype=torch.int32), None

        def decode_greedy(kv, first, n, eos_id=-1):                   # the device loop: EOS is sticky (csrc/sampling.hip argmax_parts_kernel)
            out = []
            for _ in range(n):
                t = eos_id if cur["done"] else script[cur["i"]]
                cur["i"] += 1
                cur["done"] = cur["done"] or (eos_id >= 0 and t == eos_id)
                out.append(t)
            return torch.tensor([out], dtype=torch.int32)
        m.prefill, m.decode_greedy = prefill, decode_greedy
        return m
    ids = [1, 5, 6]
    out = fake_model(EOS).generate([ids], max_new_tokens=40, stop_strings=["###"], tokenizer=Tok(), chunk=32)
    assert out[0, len(ids):].tolist() == script[:4], out                # cut at the stop string (token 14), not at the EOS two tokens later
    out = fake_model(EOS).generate([ids], max_new_tokens=40, chunk=32)  # without a stop string the same run ends at the EOS
    assert out[0, len(ids):].tolist() == script[:6]
    out = fake_model(None).generate([ids], max_new_tokens=40, stop_strings=["###"], tokenizer=Tok(), chunk=32, eos_token_id=None)
    assert out[0, len(ids):].tolist() == script[:4]                     # no EOS at all: still one chunk, cut at the string
    timings = {}
    fake_model(EOS).generate([ids], max_new_tokens=40, stop_strings=["###"], tokenizer=Tok(), chunk=4, timings=timings)
    assert timings["steps"] <= 8                                        # chunk of 4: the stop at token 4 ends the loop after the first chunk


def test_generate_chunk_bookkeeping_per_sequence():
    """The per-chunk token bookkeeping of generate() (vectorised per sequence in round 5): every sequence keeps its tokens up to and including
    its first EOS, finished sequences ignore later chunks, the loop ends with the chunk in which the last sequence finishes, `steps` counts up
    to that token.  Scripted device calls (no GPU): three sequences whose EOS falls in the first chunk, in the second chunk, never / late."""
    import torch.nn as nn
    from video_llava_amd.model.video_chatgpt import VideoChatGPTConfig, VideoChatGPTLlamaForCausalLM
    EOS = 9

    def run(eos_at, max_new, chunk):
        B = len(eos_at)
        scripts = [[100 * (b + 1) + i for i in range(200)] for b in range(B)]
        for b, k in enumerate(eos_at):
            if k is not None:
                scripts[b][k] = EOS
        m = object.__new__(VideoChatGPTLlamaForCausalLM)
        nn.Module.__init__(m)
        m.config = VideoChatGPTConfig(eos_token_id=EOS)
        m.device_ = torch.device("cpu")
        st = {"i": 0, "done": [False] * B, "calls": 0}

        def prefill(seqs, feats, max_seq, want_logits=False):
            st["i"] = 1
            st["done"] = [scripts[b][0] == EOS for b in range(B)]
            return None, torch.tensor([scripts[b][0] for b in range(B)], dtype=torch.int32), None

        def decode_greedy(kv, first, n, eos_id=-1):
            st["calls"] += 1
            out = [[0] * n for _ in range(B)]
            for j in range(n):
                for b in range(B):
                    t = eos_id if st["done"][b] else scripts[b][st["i"]]
                    st["done"][b] = st["done"][b] or t == eos_id
                    out[b][j] = t
                st["i"] += 1
            return torch.tensor(out, dtype=torch.int32)
        m.prefill, m.decode_greedy = prefill, decode_greedy
        tm = {}
        out = m.generate([[1, 2, 3]] * B, max_new_tokens=max_new, chunk=chunk, timings=tm)
        return out[:, 3:].tolist(), tm["steps"], st["calls"], scripts

    rows, steps, calls, sc = run([2, 40, None], 50, 32)
    assert steps == 50 and calls == 2                                             # 1 + 32 + 17 tokens
    assert rows[0][:3] == sc[0][:3] and rows[0][3:] == [EOS] * 47                  # finished at its EOS, padded with EOS
    assert rows[1][:41] == sc[1][:41] and rows[1][41:] == [EOS] * 9
    assert rows[2] == sc[2][:50]
    rows, steps, calls, sc = run([2, 40, 45], 120, 32)
    assert steps == 46 and calls == 2 and len(rows[0]) == 46                      # ends with the chunk in which the last sequence finished; steps up to its EOS
    assert rows[2] == sc[2][:46] and rows[1] == sc[1][:41] + [EOS] * 5
    rows, steps, calls, sc = run([0, 0], 20, 8)
    assert steps == 1 and calls == 0 and rows == [[EOS], [EOS]]                   # both finished by the prefill's token
