"""Build libpgv.so (hand-written HIP for gfx950) in-tree with hipcc.

`hipcc --offload-arch=gfx950` cross-compiles without a GPU, so this runs in the build container
as well as on the GPU box.  Objects are cached under video_llava_amd/_build keyed by source mtime.
"""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libpgv.so")
SOURCES = ["api.hip", "gemm.hip", "elementwise.hip", "vit_attn.hip", "weights.hip", "fp8.hip", "vit.hip", "llm_prefill.hip", "gemv.hip", "decode_attn.hip", "sampling.hip", "llm.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-ffp-contract=fast"]
# Attention kernels run VALU softmax on the MFMA results every chunk: keep their accumulators in the VGPR file (hipcc otherwise puts
# them in AGPRs and moves ~100 registers per chunk back and forth with v_accvgpr_read/write).
EXTRA_FLAGS = {"vit_attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libpgv needs the ROCm toolchain (no CPU fallback exists)")


def _newest_header() -> float:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "pgv.h"))
    return max(os.path.getmtime(h) for h in hs)


LAB_OBJ = os.path.join(HERE, "_build_lab")
LAB_LIB = os.path.join(HERE, "libpgv_lab.so")     # -DPGV_LAB: timing-ablation switches compiled in (results are garbage under them); never loaded by the product


SANITIZER_FLAGS = ["-fsanitize=undefined", "-fno-sanitize=vptr", "-fno-sanitize-recover=undefined", "-shared-libsan", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-g",
                   "-D_GLIBCXX_ASSERTIONS"]


def sanitizer_runtime() -> str:
    """Path of the UBSan runtime that must be LD_PRELOADed into python to load libpgv_ubsan.so."""
    for c in (os.path.join(os.path.dirname(os.path.realpath(_hipcc())), "..", "lib", "llvm", "bin", "clang"), "/opt/rocm/lib/llvm/bin/clang"):
        if os.path.exists(c):
            r = subprocess.run([c, "-print-file-name=libclang_rt.ubsan_standalone-x86_64.so"], capture_output=True, text=True)
            if r.returncode == 0 and os.path.exists(r.stdout.strip()):
                return r.stdout.strip()
    raise RuntimeError("libclang_rt.ubsan_standalone-x86_64.so not found in the ROCm toolchain")


def build_sanitizer(verbose: bool = False) -> str:
    """SURVEY.md 5 (aux, "new"): a sanitizer build of the HOST side of libpgv -- weight packing, row maps, workspace arena, graph capture,
    C-ABI argument handling -- as libpgv_ubsan.so: UndefinedBehaviorSanitizer (non-recoverable: the first report aborts the test) plus libstdc++'s
    container assertions (every std::vector / std::string index of the host shim is bounds-checked) on the host code of every translation
    unit; -fno-gpu-sanitize: the gfx950 kernels are the release kernels.  Run the GPU tests under it with scripts/sessions/r4_ubsan.sh
    (LD_PRELOAD of sanitizer_runtime() + scripts/lab/with_lib.py).  Never loaded by the product.
    AddressSanitizer is not usable here: ROCm's ASan runtime intercepts hsa_amd_memory_pool_allocate and aborts inside the HIP runtime that
    torch bundles ("allocator is trying to allocate 0x400000 bytes", before any libpgv code runs; gpurun_out/r4asan2, r4asan3)."""
    return build_variant("ubsan", [], verbose=verbose, extra_flags=SANITIZER_FLAGS)


def build_variant(name: str, defines: list[str], verbose: bool = False, csrc: str | None = None, extra_flags: list[str] | None = None) -> str:
    """Lab A/B builds: libpgv_<name>.so compiled with extra -D flags (objects under _build_<name>/), optionally from another source
    directory (e.g. `git worktree` of an older commit's csrc).  Bound only by scripts/lab/with_lib.py."""
    extra_flags = extra_flags or []
    CSRC = csrc or globals()["CSRC"]
    odir = os.path.join(HERE, f"_build_{name}")
    os.makedirs(odir, exist_ok=True)
    lib = os.path.join(HERE, f"libpgv_{name}.so")
    objs = []
    def one(src):
        opath = os.path.join(odir, src.replace(".hip", ".o"))
        cmd = [_hipcc(), *FLAGS, *extra_flags, *[f"-D{d}" for d in defines], *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", opath]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        return opath
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(one, [s_ for s_ in SOURCES if os.path.exists(os.path.join(CSRC, s_))]))
    link_extra = [f for f in extra_flags if f.startswith("-fsanitize") or f.startswith("-fno-sanitize") or f in ("-fno-gpu-sanitize", "-shared-libsan")]
    r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *link_extra, "-o", lib, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    return lib


def _compile(src: str, force: bool, verbose: bool, lab: bool = False) -> tuple[str, bool]:
    spath = os.path.join(CSRC, src)
    opath = os.path.join(LAB_OBJ if lab else OBJ, src.replace(".hip", ".o"))
    stamp = max(os.path.getmtime(spath), _newest_header())
    if not force and os.path.exists(opath) and os.path.getmtime(opath) >= stamp:
        return opath, False
    cmd = [_hipcc(), *FLAGS, *(["-DPGV_LAB"] if lab else []), *EXTRA_FLAGS.get(src, []), "-c", spath, "-o", opath]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    return opath, True


def sources_present() -> list[str]:
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def build(force: bool = False, verbose: bool = False, lab: bool = False) -> str:
    """Compile every HIP translation unit and link libpgv.so; returns the library path.  lab=True builds the separate libpgv_lab.so with
    -DPGV_LAB (the ablation switches of scripts/microbench.py); the release library contains none of them."""
    os.makedirs(LAB_OBJ if lab else OBJ, exist_ok=True)
    srcs = sources_present()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force, verbose, lab), srcs))
    objs = [o for o, _ in results]
    LIB = LAB_LIB if lab else globals()["LIB"]
    if force or any(changed for _, changed in results) or not os.path.exists(LIB):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


def check_isa() -> dict:
    """Build-time check of the one place where the code leans on the gfx950 LOWERING rather than on the HIP memory model: the
    cross-workgroup merge of decode_attn_split_kernel moves every shared byte with relaxed agent-scope atomics and no fences, which is correct
    only while those lower to write-through stores / memory-side loads (`sc1`).  Compiles csrc/decode_attn.hip to device assembly and requires,
    in every instantiation of the kernel: the partial-state and ticket stores and the merge loads carry sc1, there is exactly one ticket
    atomic, and no cache write-back / invalidate was inserted.  Raises RuntimeError otherwise (a compiler upgrade that changes the lowering
    fails the build instead of silently reading stale partial states; PGV_DATTN_SPLIT=1 is the run-time fallback).  Returns the counts."""
    import re
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "decode_attn.s")
        r = subprocess.run([_hipcc(), *[f for f in FLAGS if f != "-fPIC"], "--cuda-device-only", "-S", os.path.join(CSRC, "decode_attn.hip"), "-o", out],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"check_isa: hipcc -S failed:\n{r.stderr}")
        text = open(out).read()
    found = {}
    for m in re.finditer(r"^(_ZN\S*decode_attn_split_kernelI(\w+?)Li(\d)E\S*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        name, dt, split, body = m.group(1), m.group(2), int(m.group(3)), m.group(4)
        st = len(re.findall(r"global_store_dword\b.*\bsc1\b", body))
        ld = len(re.findall(r"global_load_dword\b.*\bsc1\b", body))
        at = len(re.findall(r"global_atomic_add\b", body))
        bad = len(re.findall(r"buffer_wbl2|buffer_inv", body))
        found[(dt, split)] = dict(sc1_stores=st, sc1_loads=ld, atomics=at, cache_maintenance=bad)
        if st < 4 or ld < 3 * split or at != 1 or bad:
            raise RuntimeError(f"check_isa: {name}: sc1 stores {st} (need >= 4), sc1 loads {ld} (need >= {3 * split}), ticket atomics {at} (need 1), "
                               f"cache maintenance instructions {bad} (need 0): the fence-free merge of the split decode attention is not safe with this "
                               "compiler's lowering -- set PGV_DATTN_SPLIT=1 and fix csrc/decode_attn.hip")
    if len(found) != 6:
        raise RuntimeError(f"check_isa: expected 6 instantiations of decode_attn_split_kernel (2 dtypes x splits 2/4/8), found {sorted(found)}")
    found.update(_check_isa_vit_attn())
    return found


def _check_isa_vit_attn() -> dict:
    """Second place that leans on hand-placed ISA: csrc/vit_attn.hip's half_swap wraps v_permlane32_swap_b32 in inline asm with its own
    wait state (`s_nop 1`: the hazard recognizer cannot see inside inline asm, and the first version without it returned wrong maxima at N = 5),
    and the K / V staging issues global_load_lds by inline asm with M0 set in the same statement.  Requires, in every instantiation of
    vit_attn_kernel: every v_permlane32_swap_b32 that comes from the inline asm is directly preceded by an s_nop of at least 1, and every
    global_load_lds_dwordx4 is directly preceded by the s_mov_b32 m0 that belongs to it (nothing scheduled in between)."""
    import re
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "vit_attn.s")
        r = subprocess.run([_hipcc(), *[f for f in FLAGS if f != "-fPIC"], *EXTRA_FLAGS.get("vit_attn.hip", []), "--cuda-device-only", "-S",
                            os.path.join(CSRC, "vit_attn.hip"), "-o", out], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"check_isa: hipcc -S failed:\n{r.stderr}")
        text = open(out).read()
    res = {}
    for m in re.finditer(r"^(_ZN\S*vit_attn_kernelI(\w+?)Li0ELi(\d)E\S*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        name, dt, nw, body = m.group(1), m.group(2), int(m.group(3)), m.group(4)
        ins = [ln.strip() for ln in body.split("\n") if ln.strip() and not ln.strip().startswith((";", ".", "//"))]
        swaps = [i for i, ln in enumerate(ins) if ln.startswith("v_permlane32_swap_b32")]
        guarded = [i for i in swaps if i > 0 and re.match(r"s_nop\s+([1-9]\d*)", ins[i - 1])]
        dmas = [i for i, ln in enumerate(ins) if ln.startswith("global_load_lds_dwordx4")]
        m0ok = [i for i in dmas if i > 0 and ins[i - 1].startswith("s_mov_b32 m0,")]
        res[("vit_attn", dt, nw)] = dict(half_swaps=len(swaps), half_swaps_with_wait_state=len(guarded), lds_dma=len(dmas), lds_dma_with_m0=len(m0ok))
        # the asm half swaps (maxima + row sums; the output transposition uses the builtin, whose hazards the compiler handles) must keep their nop
        if len(guarded) < 2 or len(dmas) < 2 or len(m0ok) != len(dmas):
            raise RuntimeError(f"check_isa: {name}: {len(guarded)} of {len(swaps)} v_permlane32_swap_b32 directly behind an s_nop >= 1 (need >= 2), "
                               f"{len(m0ok)} of {len(dmas)} global_load_lds_dwordx4 directly behind their s_mov_b32 m0 (need all, >= 2): the inline asm of "
                               "csrc/vit_attn.hip no longer assembles the way it was written")
    if len(res) != 4:
        raise RuntimeError(f"check_isa: expected 4 instantiations of vit_attn_kernel (2 dtypes x 4 / 8 waves), found {sorted(res)}")
    return res


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, lab="--lab" in sys.argv))
    if "--check-isa" in sys.argv:
        print(check_isa())
