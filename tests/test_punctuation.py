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
