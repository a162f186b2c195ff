"""CT-Transformer punctuation (N3): CPU tests of the host logic (word splitting, mini-sentence windows, the sentence cache, text
assembly) with a scripted forward, and -m gpu tests of the device forward vs oracle/punc_oracle.py (parity unpinned: funasr is absent)."""
import numpy as np
import pytest
import torch

from targetdiarization_amd.punctuation import PUNC_LIST, split_to_mini_sentence, split_words


def test_split_words_and_mini_sentences():
    assert split_words("今天 hello世界 ok2go") == ["今", "天", "hello", "世", "界", "ok2go"]
    assert split_words("") == [] and split_words("  a  b ") == ["a", "b"]
    w = list(range(45))
    assert [len(x) for x in split_to_mini_sentence(w, 20)] == [20, 20, 5] and split_to_mini_sentence(w[:20], 20) == [w[:20]]


class _HostOnly:
    """inference() without a device: CTTransformer's host logic over a scripted forward"""
    def __init__(self):
        from targetdiarization_amd.punctuation import CTTransformer
        self.__class__ = type("H", (CTTransformer,), {"__init__": lambda s: None, "__del__": lambda s: None})
        self.punc_list, self.sentence_end_id, self.unk_id, self.vocab, self.vocab_size = list(PUNC_LIST), 3, 0, None, 1000
        self._marks = ["" if p == "_" else p for p in self.punc_list]


def test_inference_cache_and_assembly():
    h = _HostOnly()
    calls = []

    def fwd(ids):                 # every 7th word of a window ends a sentence, every 3rd gets a comma
        calls.append(len(ids))
        lg = np.zeros((len(ids), 6), np.float32); lg[:, 1] = 1.0
        for i in range(len(ids)):
            if i % 7 == 6:
                lg[i, 3] = 2.0
            elif i % 3 == 2:
                lg[i, 2] = 2.0
        return lg
    text = "字" * 50
    out, punc = h.inference(text, forward=fwd)
    # window 1: 20 words, last sentence end before the final word at i = 13 -> 14 words emitted, 6 carried; window 2: 6 + 20 = 26 words ...
    assert calls[0] == 20 and calls[1] == 26
    assert out.count("字") == 50 and out.endswith("。") and len(punc) == 50
    assert "，" in out and out.index("。") == 7 + 2          # seven words + two commas before the first period
    # ASCII words: spaces, ASCII punctuation, capital letters after a sentence end
    def fwd2(ids):
        lg = np.zeros((len(ids), 6), np.float32); lg[:, 1] = 1.0
        lg[1, 3] = 2.0
        return lg
    out2, _ = h.inference("hello world how are you", forward=fwd2)
    assert out2 == "Hello world. How are you."
    assert h.inference("", forward=fwd2) == ("", [])


def test_inference_batch_in_lock_step_equals_text_by_text():
    h = _HostOnly()

    def logits_of(ids):           # a deterministic function of the token AND its position in the window (like a real model: context matters)
        ids = np.asarray(ids)
        lg = np.zeros((len(ids), 6), np.float32); lg[:, 1] = 1.0
        for i, t in enumerate(ids):
            k = (int(t) * 7 + i * 3) % 11
            if k == 0: lg[i, 3] = 2.0
            elif k == 1: lg[i, 2] = 2.0
            elif k == 2: lg[i, 4] = 2.0
        return lg

    def fwd_b(ids, lens):         # padded batch: garbage in the padding must not matter
        out = np.full((ids.shape[0], ids.shape[1], 6), 9.0, np.float32)
        for r in range(ids.shape[0]):
            out[r, :lens[r]] = logits_of(ids[r, :lens[r]])
        return out
    texts = ["字" * 50, "", "hello world how are you doing today my friend", "一二三四五六七八九十" * 9 + " ok fine", "短"]
    one = [h.inference(t, forward=logits_of) for t in texts]
    calls = []
    many = h.inference_batch(texts, forward=lambda ids, lens: (calls.append(ids.shape), fwd_b(ids, lens))[1])
    assert many == one
    assert len(calls) == 5 and calls[0][0] == 4 and calls[-1][0] == 1          # 92 words = 5 windows; four non-empty texts start together


@pytest.mark.gpu
def test_padded_batch_with_lens_equals_single_rows():
    """rows of different lengths in one launch sequence (lens_dev: attention mask + zero FSMN memory beyond the row's length) give the
    logits of the rows run alone; inference_batch() gives the texts of inference()"""
    from targetdiarization_amd.punctuation import CTTransformer
    from targetdiarization_amd.weights import recipe_punc_state_dict
    m = CTTransformer(recipe_punc_state_dict(0, num_blocks=4, vocab=4096), "cuda:0")
    rng = np.random.default_rng(1)
    lens = np.array([37, 1, 20, 0, 226, 5], dtype=np.int32)
    ids = rng.integers(0, 4096, (len(lens), int(lens.max())))
    out = m.punc_forward(ids, lens)
    for r, n in enumerate(lens):
        if n == 0:
            continue
        ref = m.punc_forward(ids[r, :n])
        assert np.abs(out[r, :n] - ref).max() <= 2e-5 * np.abs(ref).max(), (r, n)
    texts = ["今天天气怎么样我们去公园散步好不好 then we can have lunch together 之后再回家休息一下明天还要上班" * 3, "", "你好", "hello world " * 30]
    assert m.inference_batch(texts) == [m.inference(t) for t in texts]
    assert m(texts) == [m.inference(t)[0] for t in texts]
    with pytest.raises(Exception):
        m.punc_forward(ids, lens + 300)


@pytest.mark.gpu
def test_device_forward_vs_oracle():
    from oracle import punc_oracle as po
    from targetdiarization_amd.punctuation import CTTransformer
    from targetdiarization_amd.weights import recipe_punc_state_dict
    sd = recipe_punc_state_dict(0, num_blocks=4, vocab=4096)
    m = CTTransformer(sd, "cuda:0")
    sd64 = {k: v.double() for k, v in sd.items()}
    rng = np.random.default_rng(0)
    for (B, T) in [(1, 1), (1, 20), (2, 37), (1, 226), (3, 8)]:
        ids = rng.integers(0, 4096, (B, T))
        ref = po.punc_forward(torch.from_numpy(ids), sd64).numpy()
        out = m.punc_forward(ids)
        assert out.shape == (B, T, 6)
        assert np.linalg.norm(out - ref) / np.linalg.norm(ref) < 1e-4, (B, T)
        assert (np.argmax(out, -1) == np.argmax(ref, -1)).mean() > 0.99
    # end to end: the device forward and the oracle forward give the same punctuated text (where the top-2 margin is not a tie)
    text = "今天天气怎么样我们去公园散步好不好 then we can have lunch together 之后再回家休息一下明天还要上班" * 3
    got, _ = m.inference(text)
    want, _ = m.inference(text, forward=lambda ids: po.punc_forward(torch.from_numpy(np.asarray(ids, dtype=np.int64))[None], sd64)[0].numpy())
    assert got == want and got.count("今") == 3


@pytest.mark.gpu
def test_punctuation_plugs_into_asr_processor():
    from targetdiarization_amd.asr_processor import ASRProcessor
    from targetdiarization_amd.punctuation import CTTransformer
    from targetdiarization_amd.weights import recipe_punc_state_dict
    m = CTTransformer(recipe_punc_state_dict(0, vocab=512), "cuda:0")
    asrp = ASRProcessor(punctuation=m)
    t = asrp.punctuation_restore("大家好 我们 开始 吧")
    assert isinstance(t, str) and t.replace("，", "").replace("。", "").replace("？", "").replace("、", "").replace(" ", "") == "大家好我们开始吧"
    assert m(["你好", ""]) [1] == ""
    # the constructor path of the reference (`is_punc` + a model directory there; a state dict here): same texts, lists handled like :893-896
    asrp2 = ASRProcessor(punc_state_dict=recipe_punc_state_dict(0, vocab=512))
    assert asrp2.is_punc and asrp2.punctuation_restore("大家好 我们 开始 吧") == t
    assert asrp2.punctuation_restore(["大家好 我们 开始 吧", ""]) == [t, ""]
    assert ASRProcessor().punctuation_restore("abc") == "abc" and not ASRProcessor().is_punc
