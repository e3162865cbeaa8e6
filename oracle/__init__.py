"""CPU oracle for the PG-Video-LLaVA hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, fp32/fp64) restatement of the arithmetic
the reference delegates to HuggingFace `transformers` (pinned upstream at git
`cae78c46`, a 4.28-dev commit -- requirements.txt:21 of the reference; the copy
installed in this image is 5.15.0) plus the reference's own pooling / projector /
splice code.  Every function cites the reference file:line (paths relative to
/root/reference) or the HF source line (prefix `HF:` = transformers/models/...)
it follows.

Who may import it: only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg -- and there only as the checker / the timed CPU baseline.
Nothing under `video_llava_amd/` imports it; the product path fails loudly when
the HIP library is missing instead of falling back to this code.

Pinning status: the reference ships no tests or golden vectors (SURVEY.md 4,
8c).  The oracle is therefore pinned against outputs of the reference itself,
generated in the build container by `oracle/gen_golden.py` (which imports
/root/reference and HF transformers on CPU) and committed under
`tests/golden/`.  `tests/test_oracle_golden.py` re-checks the oracle against
those fixtures on every run.
"""
