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
        self._marks = ["" if p == "_" else p for p in self.punc_list]
        blob = pack_blob(state_dict)
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        h = C.c_void_p()
        with torch.cuda.device(idx):
            _lib.check(self._l.tdx_punc_create(num_blocks, self.vocab_size, len(self.punc_list), buf, len(blob), idx, C.byref(h)))
        self._h = h
        self._guard = _lib.HandleGuard(self.device)

    @staticmethod
    def _is_ascii(w: str) -> bool:
        return len(w[0].encode()) == 1

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
    def punc_forward(self, ids, lens=None) -> np.ndarray:
        """ids int [T] or [B, T]; lens (optional, [B]): row b holds lens[b] tokens, the rest is padding (masked on the device like
        funasr's padding mask: the valid positions equal the unbatched result) -> logits [T, npunc] or [B, T, npunc]"""
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
        lens_d = None
        if lens is not None:
            ln = np.ascontiguousarray(np.asarray(lens, dtype=np.int32).reshape(-1))
            if ln.shape[0] != B or (ln < 0).any() or (ln > T).any():
                raise _lib.TdxError("CTTransformer: lens must be [B] with 0 <= lens[b] <= T")
            lens_d = torch.from_numpy(ln).to(self.device)
        out = torch.empty(B, T, len(self.punc_list), device=self.device)
        with torch.cuda.device(self.device), self._guard.call():
            ws = self._guard.workspace(nb)
            st = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self._l.tdx_punc_forward(self._h, ids_d.data_ptr(), lens_d.data_ptr() if lens_d is not None else None, B, T, out.data_ptr(),
                                                ws.data_ptr(), ws.numel(), st))
        r = out.cpu().numpy()
        return r[0] if single else r

    # ---- CTTransformer.inference as a per-text state machine: windows() yields the token ids of the next window (sentence cache + the
    # next mini-sentence), send(logits) consumes the model's answer; the same code serves one text at a time and many texts in lock step
    class _Text:
        def __init__(self, model, text, split_size, cache_pop_trigger_limit):
            self.m = model
            self.words = split_words(text)
            self.ids = model.encode(self.words) if self.words else []
            self.minis = split_to_mini_sentence(self.words, split_size) if self.words else []
            self.minis_id = split_to_mini_sentence(self.ids, split_size) if self.words else []
            self.limit = cache_pop_trigger_limit
            self.i = 0
            self.cache, self.cache_id = [], []
            self.text, self.punc = "", []

        @property
        def done(self):
            return self.i >= len(self.minis)

        def window(self):
            self.sent = self.cache + self.minis[self.i]
            self.sent_id = self.cache_id + self.minis_id[self.i]
            return self.sent_id

        def consume(self, logits):
            m, pl = self.m, self.m.punc_list
            sent, sent_id = self.sent, self.sent_id
            punc = [int(x) for x in np.argmax(np.asarray(logits)[:len(sent_id)], axis=-1)]
            if self.i < len(self.minis) - 1:              # cut after the last sentence end, carry the rest into the next window
                end, last_comma = -1, -1
                for i in range(len(punc) - 2, 1, -1):
                    if pl[punc[i]] in ("。", "？"):
                        end = i
                        break
                    if last_comma < 0 and pl[punc[i]] == "，":
                        last_comma = i
                if end < 0 and len(sent) > self.limit and last_comma >= 0:
                    end = last_comma                      # the sentence is too long: cut at a comma, which becomes a period
                    punc[end] = m.sentence_end_id
                self.cache, self.cache_id = sent[end + 1:], sent_id[end + 1:]
                sent, punc = sent[:end + 1], punc[:end + 1]
            self.punc += punc
            if not any(self.m._is_ascii(w) for w in sent):        # pure CJK window (the common case): word + mark, nothing else to decide
                marks = self.m._marks
                self.text += "".join([w + marks[p] for w, p in zip(sent, punc)])
                self.i += 1
                return
            pieces = []
            for i, w in enumerate(sent):
                ascii_w = len(w[0].encode()) == 1
                if ascii_w and (i == 0 or pl[punc[i - 1]] in ("。", "？")):
                    w = w.capitalize()
                if i > 0 and ascii_w and len(sent[i - 1][0].encode()) == 1:
                    w = " " + w
                elif i == 0 and ascii_w and self.text and len(self.text[-1].encode()) == 1:
                    w = " " + w
                pieces.append(w)
                if pl[punc[i]] != "_":
                    p = pl[punc[i]]
                    if ascii_w:
                        p = {"，": ",", "。": ".", "？": "?"}.get(p, p)
                    pieces.append(p)
            self.text += "".join(pieces)
            self.i += 1

        def result(self):
            new_text, new_punc, m = self.text, self.punc, self.m
            if new_text:                                  # the text ends with a sentence end
                last = new_text[-1]
                if last in ("，", "、"):
                    new_text = new_text[:-1] + "。"; new_punc[-1] = m.sentence_end_id
                elif last == ",":
                    new_text = new_text[:-1] + "."; new_punc[-1] = m.sentence_end_id
                elif last not in ("。", "？", ".", "?"):
                    new_text += "。" if len(last.encode()) != 1 else "."
                    new_punc[-1] = m.sentence_end_id
            return new_text, new_punc

    def inference(self, text: str, split_size: int = 20, cache_pop_trigger_limit: int = 200, forward: Optional[Callable] = None):
        """-> (punctuated text, punctuation id per word).  `forward(ids[T]) -> logits[T, npunc]` overrides the device forward (tests)."""
        fwd = forward or self.punc_forward
        t = self._Text(self, text, split_size, cache_pop_trigger_limit)
        while not t.done:
            t.consume(fwd(np.asarray(t.window(), dtype=np.int32)))
        return t.result()

    def inference_batch(self, texts, split_size: int = 20, cache_pop_trigger_limit: int = 200, forward: Optional[Callable] = None):
        """many texts in lock step: window i of every text that still has one goes through ONE padded, length-masked forward
        (`forward(ids[B, T], lens[B]) -> logits[B, T, npunc]` overrides the device forward) -> [(text, punctuation ids), ...] equal to
        inference() text by text — funasr walks one text at a time; on the device that is ~30 launches and a host sync per window"""
        fwd = forward or self.punc_forward
        st = [self._Text(self, t, split_size, cache_pop_trigger_limit) for t in texts]
        while True:
            act = [t for t in st if not t.done]
            if not act:
                break
            wins = [t.window() for t in act]
            lens = np.asarray([len(w) for w in wins], dtype=np.int32)
            ids = np.zeros((len(act), int(lens.max())), dtype=np.int32)
            for r, w in enumerate(wins):
                ids[r, :len(w)] = w
            lg = np.asarray(fwd(ids, lens))
            for r, t in enumerate(act):
                t.consume(lg[r])
        return [t.result() for t in st]

    def __call__(self, text):
        """the `punctuation(text) -> text` plug-in form (lists of texts like ASRProcessor.punctuation_restore :893-896)"""
        if isinstance(text, (list, tuple)):
            return [r[0] for r in self.inference_batch(list(text))]
        return self.inference(text)[0] if text else text

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._l.tdx_punc_destroy(self._h); self._h = None
        except Exception:
            pass
