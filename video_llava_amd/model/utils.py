"""Stop-string criterion with the reference's semantics (video_chatgpt/model/utils.py:6-26): a keyword whose
tokenisation is exactly one id is matched on the last generated id; otherwise the generated tail is decoded and
searched for the keyword."""
from __future__ import annotations


class KeywordsStoppingCriteria:
    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = keywords
        ids = [tokenizer(k).input_ids for k in keywords]
        self.keyword_ids = [i[0] for i in ids if isinstance(i, list) and len(i) == 1]
        self.tokenizer = tokenizer
        self.start_len = None
        self.input_ids = input_ids

    def __call__(self, output_ids, scores=None, **kwargs) -> bool:
        if self.start_len is None:
            # first call only records where generation starts (reference :16-17)
            self.start_len = self.input_ids.shape[1]
            return False
        last = int(output_ids[0, -1])
        if any(last == k for k in self.keyword_ids):
            return True
        text = self.tokenizer.batch_decode(output_ids[:, self.start_len:], skip_special_tokens=True)[0]
        return any(k in text for k in self.keywords)


LINEAR_SCAN_MAX = 192      # answers up to this many tokens are scanned prefix by prefix like the reference (O(n^2) decodes, negligible at this size)


def first_stop_length(new_tokens, tokenizer, keywords, start: int = 2):
    """Number of generated tokens a per-token loop with `KeywordsStoppingCriteria(keywords, ...)` keeps, or None when it never fires.
    The criterion's first call (one generated token) only records the start; from the second token on it fires when the last id is a
    single-id keyword or the decoded tail contains a keyword -- so this is the smallest n >= 2 with either property.  `keyword in
    decode(tokens[:n])` is monotone in n for the ASCII stop strings of the conversation templates, hence a binary search instead of the
    reference's decode-per-token; monotonicity can fail for byte-fallback pieces (a prefix decodes to U+FFFD until the piece completes) or
    decoders that merge / strip across tokens, so the result is verified (`has(n - 1)` must be False) and a linear scan takes over if not.
    `start` (>= 2): the caller has already established that no prefix shorter than `start` fires (the per-chunk check of the batched
    greedy loop: lengths up to the previous chunk's end were scanned then), so short answers are scanned from there."""
    ids = [tokenizer(k).input_ids for k in keywords]
    keyword_ids = {i[0] for i in ids if isinstance(i, list) and len(i) == 1}
    toks = [int(t) for t in new_tokens]
    start = max(2, int(start))
    n_id = next((i + 1 for i in range(start - 1, len(toks)) if toks[i] in keyword_ids), None)

    def has(n):
        text = tokenizer.batch_decode([toks[:n]], skip_special_tokens=True)[0]
        return any(k in text for k in keywords)

    n_txt = None
    if len(toks) <= LINEAR_SCAN_MAX:
        # short answers (the common case): exactly the reference's loop, prefix by prefix -- no assumption about the decoder at all
        n_txt = next((n for n in range(start, len(toks) + 1) if has(n)), None)
    elif has(len(toks)):
        lo, hi = 2, len(toks)                       # invariant: has(hi)
        while lo < hi:
            mid = (lo + hi) // 2
            if has(mid):
                hi = mid
            else:
                lo = mid + 1
        n_txt = hi
        if n_txt > 2 and has(n_txt - 1):            # not monotone for this tokenizer / keyword: do what the reference does, token by token
            n_txt = next(n for n in range(2, len(toks) + 1) if has(n))
    elif "\ufffd" in tokenizer.batch_decode([toks], skip_special_tokens=True)[0]:
        # has(len) is False, yet a lossy decoder (replacement characters) may have shown the keyword at a shorter prefix: only a scan can tell
        n_txt = next((n for n in range(2, len(toks)) if has(n)), None)
    if n_id is None:
        return n_txt
    return n_id if n_txt is None else min(n_id, n_txt)
