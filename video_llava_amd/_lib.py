"""ctypes binding of libpgv.so (include/pgv.h).

There is NO fallback: if the HIP library cannot be built/loaded, or there is no gfx950 device when a
context is requested, this module raises.  Nothing here (or anywhere under video_llava_amd/) imports the
CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpgv.so")

ABI_VERSION = 310          # PGV_VERSION of include/pgv.h this table was written against
PGV_OK, PGV_EINVAL, PGV_EHIP, PGV_ENOMEM, PGV_ESTATE, PGV_ENAME = 0, 1, 2, 3, 4, 5
PGV_F16, PGV_BF16, PGV_F32 = 0, 1, 2
EPI_NONE, EPI_BIAS, EPI_BIAS_QGELU, EPI_BIAS_GELU, EPI_RESID, EPI_BIAS_RESID, EPI_SWIGLU, EPI_F32 = range(8)
FAMILIES = ("gemm", "vit_attn", "llm_prefill_attn", "decode_gemv", "decode_attn", "other", "decode_small")


class VitConfig(C.Structure):
    _fields_ = [("hidden", C.c_int), ("inter", C.c_int), ("layers", C.c_int), ("heads", C.c_int),
                ("image", C.c_int), ("patch", C.c_int), ("eps", C.c_float)]


class LlmConfig(C.Structure):
    _fields_ = [("vocab", C.c_int), ("hidden", C.c_int), ("inter", C.c_int), ("layers", C.c_int), ("heads", C.c_int),
                ("eps", C.c_float), ("rope_theta", C.c_float)]


_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); every symbol declared in include/pgv.h
PROTOTYPES = {
    "pgv_version": (_i, []),
    "pgv_last_error": (C.c_char_p, []),
    "pgv_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "pgv_ctx_destroy": (None, [_vp]),
    "pgv_ctx_workspace_bytes": (C.c_size_t, [_vp]),
    "pgv_prof_enable": (_i, [_vp, _i]),
    "pgv_prof_reset": (_i, [_vp]),
    "pgv_prof_get": (_i, [_vp, _i, C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pgv_prof_calibrate": (_i, [_vp, _vp, _i, C.POINTER(C.c_double)]),
    "pgv_vit_create": (_i, [_vp, C.POINTER(VitConfig), _i, C.POINTER(_vp)]),
    "pgv_vit_destroy": (None, [_vp]),
    "pgv_vit_load_tensor": (_i, [_vp, C.c_char_p, _vp, _i, _i, _i64, _vp]),
    "pgv_vit_missing": (_i, [_vp]),
    "pgv_preprocess_u8": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "pgv_vit_forward": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "pgv_st_pool": (_i, [_vp, _vp, _i, _i, _i, _i, _i64, _i, _vp, _i, _vp]),
    "pgv_projector": (_i, [_vp, _i, _i, C.POINTER(_vp), C.POINTER(_vp), _i, _i, _vp, _i, _vp, _vp]),
    "pgv_llm_create": (_i, [_vp, C.POINTER(LlmConfig), _i, C.POINTER(_vp)]),
    "pgv_llm_destroy": (None, [_vp]),
    "pgv_llm_load_tensor": (_i, [_vp, C.c_char_p, _vp, _i, _i, _i64, _vp]),
    "pgv_llm_missing": (_i, [_vp]),
    "pgv_llm_load_rows": (_i, [_vp, C.c_char_p, _vp, _i, _i, _i, _i, _i64, _vp]),
    "pgv_llm_resize_vocab": (_i, [_vp, _i, _vp]),
    "pgv_llm_vocab": (_i, [_vp]),
    "pgv_kv_create": (_i, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    "pgv_kv_destroy": (None, [_vp]),
    "pgv_kv_len": (_i, [_vp, _i]),
    "pgv_llm_prefill": (_i, [_vp, _vp, _vp, _vp, C.POINTER(C.c_int32), _i, _vp, _i, C.POINTER(C.c_int32), _vp, _vp, _vp, _i, _vp]),
    "pgv_llm_prefill_append": (_i, [_vp, _vp, _vp, _vp, C.POINTER(C.c_int32), _i, _vp, _i, C.POINTER(C.c_int32), _vp, _vp, _vp, _i, _vp]),
    "pgv_kv_truncate": (_i, [_vp, _i, _i, _vp]),
    "pgv_llm_decode": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pgv_llm_decode_greedy": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "pgv_llm_sample": (_i, [_vp, _vp, _vp, _f, _i, _vp, _vp, _vp]),
    "pgv_llm_decode_sample": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp, _vp, _vp]),
    "pgv_sample_logits": (_i, [_vp, _vp, _i, _i, _f, _i, _vp, _vp, _vp]),
    "pgv_ingest_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "pgv_gemm": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pgv_vit_attention": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pgv_gemv": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "pgv_pack_blocked": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp]),
    "pgv_unpack_blocked": (_i, [_vp, _i, _vp, _vp, _i, _i, _vp]),
    "pgv_quantize_fp8_blocked": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "pgv_gemv_fp8": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "pgv_llm_quantize_fp8": (_i, [_vp, _vp, _vp]),
    "pgv_llm_is_fp8": (_i, [_vp]),
    "pgv_llm_get_weight": (_i, [_vp, _vp, C.c_char_p, _vp, _vp]),
    "pgv_layernorm": (_i, [_vp, _i, _vp, _vp, _vp, _f, _vp, _i, _i, _vp]),
    "pgv_rmsnorm": (_i, [_vp, _i, _vp, _vp, _f, _vp, _i, _i, _vp]),
}

_lib = None
_lock = threading.Lock()


def use_lab_build() -> None:
    """Lab scripts only (scripts/microbench.py ablate): bind the -DPGV_LAB library, which carries the timing-ablation switches.  Must be called
    before the first load(); nothing in the product calls it and no environment variable selects it."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("libpgv is already loaded")
    from . import build as _build
    LIB_PATH = _build.build(lab=True)


def load(build_if_missing: bool = True) -> C.CDLL:
    """Load libpgv.so (building it with hipcc first if it is not there).  Raises on any failure."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            if not build_if_missing:
                raise RuntimeError(f"{LIB_PATH} is missing; run `python -m video_llava_amd.build`")
            from . import build as _build
            _build.build()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)          # AttributeError = header/library mismatch: fail loudly
            fn.restype = res
            fn.argtypes = args
        if lib.pgv_version() != ABI_VERSION:
            raise RuntimeError(f"libpgv version {lib.pgv_version()} does not match the binding ({ABI_VERSION}); rebuild with "
                               "`python -m video_llava_amd.build --force`")
        _lib = lib
        return lib


def check(rc: int, what: str = "") -> None:
    if rc == PGV_OK:
        return
    msg = load().pgv_last_error().decode("utf-8", "replace")
    text = f"{what}: {msg}" if what else msg
    if rc == PGV_EINVAL or rc == PGV_ENAME:
        raise ValueError(text)
    if rc == PGV_ENOMEM:
        raise MemoryError(text)
    raise RuntimeError(text)


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float16:
        return PGV_F16
    if dt == torch.bfloat16:
        return PGV_BF16
    if dt == torch.float32:
        return PGV_F32
    raise ValueError(f"unsupported dtype {dt}")


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class Context:
    """One pgv_ctx per GPU (per rank).  Use Context.get(device)."""
    _cache: dict[int, "Context"] = {}

    def __init__(self, device: int):
        if not torch.cuda.is_available():
            raise RuntimeError("video_llava_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False "
                               "and there is no CPU fallback")
        self.lib = load()
        self.device = int(device)
        h = _vp()
        check(self.lib.pgv_ctx_create(self.device, C.byref(h)), "pgv_ctx_create")
        self.handle = h

    @classmethod
    def get(cls, device=None) -> "Context":
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        if isinstance(device, torch.device):
            device = device.index if device.index is not None else torch.cuda.current_device()
        device = int(device)
        if device not in cls._cache:
            cls._cache[device] = Context(device)
        return cls._cache[device]

    # ---- profiling -----------------------------------------------------------------------
    def prof_enable(self, on: bool = True):
        check(self.lib.pgv_prof_enable(self.handle, 1 if on else 0))

    def prof_reset(self):
        check(self.lib.pgv_prof_reset(self.handle))

    def prof_get(self) -> dict[str, dict]:
        out = {}
        for i, name in enumerate(FAMILIES):
            n, ms, fl, by = _i64(), C.c_double(), C.c_double(), C.c_double()
            check(self.lib.pgv_prof_get(self.handle, i, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)))
            out[name] = {"launches": n.value, "ms": ms.value, "flops": fl.value, "bytes": by.value}
        return out

    def prof_calibrate(self, n: int = 512) -> float:
        """Mean elapsed ms of an empty event pair on the current stream (the fixed cost inside every per-launch measurement)."""
        ms = C.c_double()
        check(self.lib.pgv_prof_calibrate(self.handle, stream_ptr(), n, C.byref(ms)), "pgv_prof_calibrate")
        return ms.value

    # ---- thin op wrappers used by tests and by the Python mirrors ------------------------------
    def gemm(self, a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None = None, epi: int = EPI_NONE,
             out: torch.Tensor | None = None) -> torch.Tensor:
        """out = a @ w.T with the fused epilogue `epi` (see include/pgv.h enum pgv_epi)."""
        assert a.is_cuda and w.is_cuda and a.dtype == w.dtype and a.dim() == 2 and w.dim() == 2
        assert a.stride(1) == 1 and w.stride(1) == 1
        M, K = a.shape
        N = w.shape[0]
        if bias is not None:
            assert bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == N
        if out is None:
            if epi == EPI_F32:
                out = torch.empty(M, N, dtype=torch.float32, device=a.device)
            elif epi == EPI_SWIGLU:
                out = torch.empty(M, N // 2, dtype=a.dtype, device=a.device)
            elif epi in (EPI_RESID, EPI_BIAS_RESID):
                raise ValueError("residual epilogues accumulate into `out`; pass it")
            else:
                out = torch.empty(M, N, dtype=a.dtype, device=a.device)
        check(self.lib.pgv_gemm(self.handle, dtype_code(a.dtype), epi, a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0),
                                bias.data_ptr() if bias is not None else None, out.data_ptr(), out.stride(0),
                                M, N, K, stream_ptr(a.device)), "pgv_gemm")
        return out

    def layernorm(self, x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, dtype: torch.dtype) -> torch.Tensor:
        assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
        y = torch.empty(x.shape, dtype=dtype, device=x.device)
        check(self.lib.pgv_layernorm(self.handle, dtype_code(dtype), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps,
                                     y.data_ptr(), x.shape[0], x.shape[1], stream_ptr(x.device)), "pgv_layernorm")
        return y

    def rmsnorm(self, x: torch.Tensor, gamma: torch.Tensor, eps: float, dtype: torch.dtype) -> torch.Tensor:
        assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
        y = torch.empty(x.shape, dtype=dtype, device=x.device)
        check(self.lib.pgv_rmsnorm(self.handle, dtype_code(dtype), x.data_ptr(), gamma.data_ptr(), eps,
                                   y.data_ptr(), x.shape[0], x.shape[1], stream_ptr(x.device)), "pgv_rmsnorm")
        return y

    def st_pool(self, feats: torch.Tensor, n_temporal: int = 100, out_dtype: torch.dtype = torch.float16) -> torch.Tensor:
        """[T, P, C] (any frame stride, unit channel stride) -> [n_temporal + P, C]."""
        assert feats.is_cuda and feats.dim() == 3 and feats.stride(2) == 1 and feats.stride(1) == feats.shape[2]
        T, P, Cc = feats.shape
        out = torch.empty(n_temporal + P, Cc, dtype=out_dtype, device=feats.device)
        check(self.lib.pgv_st_pool(self.handle, feats.data_ptr(), dtype_code(feats.dtype), T, P, Cc, feats.stride(0), n_temporal,
                                   out.data_ptr(), dtype_code(out_dtype), stream_ptr(feats.device)), "pgv_st_pool")
        return out

    def ingest_u8(self, frames: torch.Tensor, size: int, dtype: torch.dtype) -> torch.Tensor:
        """uint8 [T, H, W, 3] on the device at native resolution -> nearest resize (load_video's rule) + CLIP normalisation -> [T, 3, size, size]."""
        assert frames.is_cuda and frames.dtype == torch.uint8 and frames.is_contiguous() and frames.dim() == 4 and frames.shape[-1] == 3
        T, H, W = frames.shape[0], frames.shape[1], frames.shape[2]
        out = torch.empty(T, 3, size, size, dtype=dtype, device=frames.device)
        check(self.lib.pgv_ingest_u8(self.handle, frames.data_ptr(), T, H, W, size, dtype_code(dtype), out.data_ptr(),
                                     stream_ptr(frames.device)), "pgv_ingest_u8")
        return out

    def sample_logits(self, logits: torch.Tensor, u: torch.Tensor, temperature: float, top_k: int = 50) -> torch.Tensor:
        """Inverse-CDF pick from softmax(logits / temperature) restricted to the top_k logits (<= 0: all) with uniforms u [B] -> int32 [B]."""
        assert logits.is_cuda and logits.dtype == torch.float32 and logits.is_contiguous() and logits.dim() == 2
        assert u.is_cuda and u.dtype == torch.float32 and u.is_contiguous() and u.numel() == logits.shape[0]
        out = torch.empty(logits.shape[0], dtype=torch.int32, device=logits.device)
        check(self.lib.pgv_sample_logits(self.handle, logits.data_ptr(), logits.shape[1], logits.shape[0], float(temperature), int(top_k),
                                         u.data_ptr(), out.data_ptr(), stream_ptr(logits.device)), "pgv_sample_logits")
        return out

    def preprocess_u8(self, frames: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        """uint8 [T, S, S, 3] on the device -> normalised [T, 3, S, S]."""
        assert frames.is_cuda and frames.dtype == torch.uint8 and frames.is_contiguous() and frames.shape[-1] == 3
        T, S = frames.shape[0], frames.shape[1]
        assert frames.shape[2] == S
        out = torch.empty(T, 3, S, S, dtype=dtype, device=frames.device)
        check(self.lib.pgv_preprocess_u8(self.handle, frames.data_ptr(), T, S, dtype_code(dtype), out.data_ptr(),
                                         stream_ptr(frames.device)), "pgv_preprocess_u8")
        return out
