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
