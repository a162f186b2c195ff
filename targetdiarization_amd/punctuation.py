"""CT-Transformer punctuation restorer: `ASRProcessor.punctuation_restore` (ASRProcessor.py:880-897) calls funasr's
`CTTransformer.inference(input=text)`; funasr is third-party and absent — parity unpinned.  The neural forward (Embedding -> SANM
encoder -> Linear) runs on the device (tdx_punc_*, csrc/punc.hip); the mini-sentence windows, the sentence cache and the assembly of
the punctuated text are host logic restated from the published funasr code [upstream-recall]:

    punc = CTTransformer(state_dict, device="cuda:0", vocab={"token": id, ...})      # funasr's tokens.json is absent: a dict or a callable
    text = punc("今天天气怎么样 hello world")                                         # -> (the reference keeps res[0]['text'])
    TargetDiarization(..., punctuation=punc) / ASRProcessor(..., punctuation=punc)
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from .weights import pack_blob

PUNC_LIST = ("<unk>", "_", "，", "。", "？", "、")


def split_words(text: str):
    """funasr split_words: whitespace-separated segments; inside a segment ASCII characters accumulate into words, every other
    character (a CJK ideograph, a full-width sign) is a word of its own"""
    words = []
    for seg in text.split():
        cur = ""
        for ch in seg:
            if len(ch.encode()) == 1:
                cur += ch
            else:
                if cur:
                    words.append(cur); cur = ""
                words.append(ch)
        if cur:
            words.append(cur)
    return words


def split_to_mini_sentence(words: Sequence, word_limit: int = 20):
    assert word_limit > 1
    if len(words) <= word_limit:
        return [list(words)]
    n = len(words) // word_limit
    out = [list(words[i * word_limit:(i + 1) * word_limit]) for i in range(n)]
    if len(words) % word_limit:
        out.append(list(words[n * word_limit:]))
    return out


class CTTransformer:
    def __init__(self, state_dict, device="cuda:0", vocab: Union[dict, Callable, None] = None, punc_list: Sequence[str] = PUNC_LIST,
                 sentence_end_id: int = 3, unk_id: int = 0, num_blocks: Optional[int] = None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.TdxError("CTTransformer needs a HIP device")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self._l = _lib.lib()
        if num_blocks is None:
            num_blocks = 2 + max([int(k.split("encoders.")[1].split(".")[0]) for k in state_dict if ".encoders." in k] + [-1])
        self.vocab_size = int(state_dict["embed.weight"].shape[0])
        self.punc_list = list(punc_list)
        if int(state_dict["decoder.weight"].shape[0]) != len(self.punc_list):
            raise _lib.TdxError("CTTransformer: decoder rows != len(punc_list)")
        self.sentence_end_id, self.unk_id = sentence_end_id, unk_id
        self.vocab = vocab
        blob = pack_blob(state_dict)
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        h = C.c_void_p()
        with torch.cuda.device(idx):
            _lib.check(self._l.tdx_punc_create(num_blocks, self.vocab_size, len(self.punc_list), buf, len(blob), idx, C.byref(h)))
        self._h = h
        self._guard = _lib.HandleGuard(self.device)

    # ---- tokenizer (CharTokenizer over tokens.json in funasr; here a dict / callable, unknown words -> unk_id)
    def encode(self, words):
        if callable(self.vocab):
            return [int(self.vocab(w)) for w in words]
        if isinstance(self.vocab, dict):
            return [int(self.vocab.get(w, self.vocab.get(w.lower(), self.unk_id))) for w in words]
        # no vocabulary: a deterministic stand-in id per word (recipe weights carry no meaning either)
        import zlib
        return [1 + zlib.crc32(w.encode()) % (self.vocab_size - 1) for w in words]

    # ---- the neural forward: ids [T] or [B, T] -> logits
    def punc_forward(self, ids) -> np.ndarray:
        a = np.ascontiguousarray(np.asarray(ids, dtype=np.int32))
        single = a.ndim == 1
        if single:
            a = a[None]
        B, T = a.shape
        if T == 0:
            return np.zeros((0, len(self.punc_list)), np.float32) if single else np.zeros((B, 0, len(self.punc_list)), np.float32)
        nb = int(self._l.tdx_punc_workspace_bytes(self._h, B, T))
        if nb == 0:
            raise _lib.TdxError(f"CTTransformer: bad shape B={B} T={T} (T <= 1024)")
        ids_d = torch.from_numpy(a).to(self.device)
        out = torch.empty(B, T, len(self.punc_list), device=self.device)
        with torch.cuda.device(self.device), self._guard.call():
            ws = self._guard.workspace(nb)
            st = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self._l.tdx_punc_forward(self._h, ids_d.data_ptr(), B, T, out.data_ptr(), ws.data_ptr(), ws.numel(), st))
        r = out.cpu().numpy()
        return r[0] if single else r

    # ---- CTTransformer.inference
    def inference(self, text: str, split_size: int = 20, cache_pop_trigger_limit: int = 200, forward: Optional[Callable] = None):
        """-> (punctuated text, punctuation id per word).  `forward(ids[T]) -> logits[T, npunc]` overrides the device forward (tests)."""
        fwd = forward or self.punc_forward
        pl = self.punc_list
        words = split_words(text)
        if not words:
            return "", []
        ids = self.encode(words)
        minis, minis_id = split_to_mini_sentence(words, split_size), split_to_mini_sentence(ids, split_size)
        cache, cache_id = [], []
        new_text, new_punc = "", []
        for mi in range(len(minis)):
            sent = cache + minis[mi]
            sent_id = cache_id + minis_id[mi]
            punc = [int(x) for x in np.argmax(np.asarray(fwd(np.asarray(sent_id, dtype=np.int32))), axis=-1)]
            if mi < len(minis) - 1:                     # cut after the last sentence end, carry the rest into the next window
                end, last_comma = -1, -1
                for i in range(len(punc) - 2, 1, -1):
                    if pl[punc[i]] in ("。", "？"):
                        end = i
                        break
                    if last_comma < 0 and pl[punc[i]] == "，":
                        last_comma = i
                if end < 0 and len(sent) > cache_pop_trigger_limit and last_comma >= 0:
                    end = last_comma                    # the sentence is too long: cut at a comma, which becomes a period
                    punc[end] = self.sentence_end_id
                cache, cache_id = sent[end + 1:], sent_id[end + 1:]
                sent, punc = sent[:end + 1], punc[:end + 1]
            new_punc += punc
            pieces = []
            for i, w in enumerate(sent):
                ascii_w = len(w[0].encode()) == 1
                if ascii_w and (i == 0 or pl[punc[i - 1]] in ("。", "？")):
                    w = w.capitalize()
                if i > 0 and ascii_w and len(sent[i - 1][0].encode()) == 1:
                    w = " " + w
                elif i == 0 and ascii_w and new_text and len(new_text[-1].encode()) == 1:
                    w = " " + w
                pieces.append(w)
                if pl[punc[i]] != "_":
                    p = pl[punc[i]]
                    if ascii_w:
                        p = {"，": ",", "。": ".", "？": "?"}.get(p, p)
                    pieces.append(p)
            new_text += "".join(pieces)
        if new_text:                                    # the text ends with a sentence end
            last = new_text[-1]
            if last in ("，", "、"):
                new_text = new_text[:-1] + "。"; new_punc[-1] = self.sentence_end_id
            elif last == ",":
                new_text = new_text[:-1] + "."; new_punc[-1] = self.sentence_end_id
            elif last not in ("。", "？", ".", "?"):
                new_text += "。" if len(last.encode()) != 1 else "."
                new_punc[-1] = self.sentence_end_id
        return new_text, new_punc

    def __call__(self, text):
        """the `punctuation(text) -> text` plug-in form (lists of texts like ASRProcessor.punctuation_restore :893-896)"""
        if isinstance(text, (list, tuple)):
            return [self.inference(t)[0] for t in text]
        return self.inference(text)[0] if text else text

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._l.tdx_punc_destroy(self._h); self._h = None
        except Exception:
            pass
