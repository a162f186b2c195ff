"""-m gpu tests of the threading contract (include/tdx.h "threading", _lib.HandleGuard, _lib.GraphRunner):
  * C level: ONE tdx_mf2 handle driven by two host threads at once, each with its own workspace and stream, equals the
    sequential run bit for bit (the fork/join side stream + events are per caller stream, csrc/mf2.hip side_for);
  * Python level: ONE model object called from two threads on different streams (shared grow-only workspace, serialised by
    the guard) equals the sequential run bit for bit — separator, speaker embedder, Paraformer encoder + decoder;
  * graph replay: shapes that never repeat are not captured; a repeated shape is captured at its second sighting and the
    replay equals the eager forward bit for bit; eviction under many shapes keeps results right;
  * serving shell: a REST request during an open WebSocket stream on the SAME device model equals the serial results."""
import ctypes as C
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch.device("cuda:0")


def _waves(seed, B, T):
    from targetdiarization_amd.weights import recipe_wave
    return torch.from_numpy(recipe_wave(f"conc{seed}", B, T))


def test_c_handle_two_threads_own_workspace_and_stream(sd2, dev):
    from targetdiarization_amd import _lib
    from targetdiarization_amd.separator import MossFormer2Separator
    sep = MossFormer2Separator(sd2, device=dev, graph_rows=0)
    l, h = sep._l, sep._h
    shapes = [(2, 4803), (1, 16000)]                        # different inputs AND different shapes per thread
    xs = [_waves(i, *s).to(dev) for i, s in enumerate(shapes)]

    def forward(i, stream, n_iter, outs):
        B, T = shapes[i]
        ws = torch.empty(sep.workspace_bytes(B, T), dtype=torch.uint8, device=dev)
        with torch.cuda.stream(stream):
            for _ in range(n_iter):
                out = torch.empty(B, 2, T, device=dev)
                _lib.check(l.tdx_mf2_forward(h, xs[i].data_ptr(), B, T, out.data_ptr(), ws.data_ptr(), ws.numel(), stream.cuda_stream))
                outs.append(out)
        stream.synchronize()
    torch.cuda.synchronize()
    ref = [[], []]
    for i in range(2):
        forward(i, torch.cuda.Stream(dev), 1, ref[i])
    got = [[], []]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    ts = [threading.Thread(target=forward, args=(i, streams[i], 6, got[i])) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert all(not t.is_alive() for t in ts)
    for i in range(2):
        assert len(got[i]) == 6
        for o in got[i]:
            assert torch.equal(o, ref[i][0]), f"thread {i}: concurrent forward differs from the sequential one"


def test_model_objects_two_threads_equal_sequential(sd2, dev):
    from targetdiarization_amd.paraformer import ParaformerDecoder, ParaformerEncoder
    from targetdiarization_amd.separator import MossFormer2Separator
    from targetdiarization_amd.speaker import ERes2NetV2
    from targetdiarization_amd.weights import recipe_eres2netv2_state_dict, recipe_paraformer_decoder_state_dict, recipe_paraformer_state_dict
    sep = MossFormer2Separator(sd2, device=dev)
    spk = ERes2NetV2(recipe_eres2netv2_state_dict(0), device=dev)
    asr_sd = dict(recipe_paraformer_state_dict(0, 2)); asr_sd.update(recipe_paraformer_decoder_state_dict(0, 2))
    enc = ParaformerEncoder({k: v for k, v in asr_sd.items() if k.startswith("encoder.")}, dev)
    dec = ParaformerDecoder(asr_sd, dev)
    jobs = [[_waves(10 + i, 1, T).to(dev) for T in (4803, 9000, 16000)] for i in range(2)]       # per thread: growing workspaces

    def run_all(x):
        y = sep(x)
        e = spk(x)
        z = enc(x)
        d = dec.decode(z)
        return y, e, z, d
    ref = [[run_all(x) for x in jobs[i]] for i in range(2)]
    torch.cuda.synchronize()
    got, errs = [None, None], []

    def worker(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream(dev)):
                r = []
                for _ in range(2):
                    r = [run_all(x) for x in jobs[i]]
                torch.cuda.current_stream(dev).synchronize()
                got[i] = r
        except Exception as e:                           # noqa: BLE001
            errs.append(repr(e))
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    assert not errs, errs
    for i in range(2):
        for (y, e, z, d), (y0, e0, z0, d0) in zip(got[i], ref[i]):
            assert torch.equal(y, y0) and torch.equal(e, e0) and torch.equal(z, z0)
            assert [r["token_ids"] for r in d] == [r["token_ids"] for r in d0]


def test_graph_runner_second_sighting_lru_and_distinct_shapes(sd2, dev):
    import time
    from targetdiarization_amd.separator import MossFormer2Separator
    sep = MossFormer2Separator(sd2, device=dev)
    eager = MossFormer2Separator(sd2, device=dev, graph_rows=0)
    # (1) a stream of distinct lengths (the streaming session's pattern): nothing is captured, results equal the eager object
    Ts = [4000 + 371 * i for i in range(24)]
    xs = [_waves(20 + i, 1, T).to(dev) for i, T in enumerate(Ts)]
    for T in (3999, 4100):
        w = _waves(19, 1, T).to(dev); eager(w); sep(w)     # warm both objects (shapes that do not come back)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs_e = [eager(x) for x in xs]
    torch.cuda.synchronize(); t_eager = time.perf_counter() - t0
    t0 = time.perf_counter()
    outs_g = [sep(x) for x in xs]
    torch.cuda.synchronize(); t_graph = time.perf_counter() - t0
    assert sep._graphs.captures == 0
    assert all(torch.equal(a, b) for a, b in zip(outs_e, outs_g))
    assert t_graph < 1.5 * t_eager + 0.05, (t_graph, t_eager)     # no capture cost for shapes that do not come back
    # (2) a repeated shape: eager at the first sighting, captured at the second, replayed afterwards — all bit-identical
    x = _waves(99, 1, 12345).to(dev)
    want = eager(x)
    y1 = sep(x); assert sep._graphs.captures == 0
    y2 = sep(x); assert sep._graphs.captures == 1
    y3 = sep(x); assert sep._graphs.captures == 1
    assert torch.equal(y1, want) and torch.equal(y2, want) and torch.equal(y3, want)
    # (3) more repeated shapes than entries: least recently used entries are evicted, results stay right
    reps = [_waves(200 + i, 1, 5000 + 800 * i).to(dev) for i in range(sep._graphs.max_entries + 3)]
    wants = [eager(r) for r in reps]
    for _ in range(3):
        for r, w in zip(reps, wants):
            assert torch.equal(sep(r), w)
    assert len(sep._graphs._g) <= sep._graphs.max_entries
    # replays of one entry from two different streams (static buffers shared): ordered by the guard's event chain
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    a, b = _waves(300, 1, 12345).to(dev), _waves(301, 1, 12345).to(dev)
    wa, wb = eager(a), eager(b)
    torch.cuda.synchronize()
    res = []
    for k in range(6):
        with torch.cuda.stream(s1 if k % 2 == 0 else s2):
            res.append(sep(a if k % 2 == 0 else b))
    torch.cuda.synchronize()
    for k, r in enumerate(res):
        assert torch.equal(r, wa if k % 2 == 0 else wb)


def _load(gold, name):
    import os
    import wave
    with wave.open(os.path.join(gold, name), "rb") as w:
        return np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0


def test_rest_during_an_open_websocket_stream_on_the_device_model(gold, sd2):
    """the reference's server shares ONE model between the REST route and WebSocket worker threads (main.py:42,366-367)"""
    pytest.importorskip("fastapi")
    import base64
    import os
    from fastapi.testclient import TestClient
    from targetdiarization_amd.server import create_app
    from targetdiarization_amd.target_diarization_stream import TargetDiarizationStream
    from targetdiarization_amd.weights import recipe_eres2netv2_state_dict, recipe_paraformer_decoder_state_dict, recipe_paraformer_state_dict
    asr_sd = dict(recipe_paraformer_state_dict(0, 2)); asr_sd.update(recipe_paraformer_decoder_state_dict(0, 2))
    model = TargetDiarizationStream(max_buffer_duration=3.0, cuda_device=0, sep_state_dict=sd2, spk_state_dict=recipe_eres2netv2_state_dict(0),
                                    asr_state_dict=asr_sd, od_pipeline=lambda a: [])
    mix, tgt = _load(gold, "chat_mix.wav"), _load(gold, "female_a.wav")
    chunks = [(mix[i:i + 16000] * 32767).astype(np.int16) for i in range(0, mix.shape[0] - 8000, 16000)]
    # serial references: the stream through a private session, the REST decisions through a direct infer()
    ref_stream = [(r[0]["speaker"], r[0]["text"], r[0]["timerange"]) for _, r, _ in
                  model.session().infer_stream(iter([c.astype(np.float32) / 32767.0 for c in chunks]), target_file=None)]
    spk, res, _ = model.infer(mix, tgt, output_target_audio=False)
    app = create_app(model)
    got, errs = {}, []

    def ws_client(name):
        try:
            with TestClient(app).websocket_connect("/diarization/stream") as ws:
                ws.send_json({"type": "config", "data": {}})
                assert ws.receive_json()["type"] == "config_ack"
                for c in chunks:
                    ws.send_json({"type": "audio_chunk", "data": base64.b64encode(c.tobytes()).decode()})
                ws.send_json({"type": "end"})
                segs = []
                while True:
                    m = ws.receive_json()
                    if m["type"] == "segment_result":
                        s_ = m["data"]["segment"]; segs.append((s_["speaker"], s_["text"], s_["timerange"]))
                    elif m["type"] == "status":
                        break
                    else:
                        raise AssertionError(m)
            got[name] = segs
        except Exception as e:                           # noqa: BLE001
            errs.append((name, repr(e)))

    def rest_client():
        try:
            with open(os.path.join(gold, "chat_mix.wav"), "rb") as f, open(os.path.join(gold, "female_a.wav"), "rb") as g:
                j = TestClient(app).post("/diarization/infer?output_target_audio=false",
                                         files={"audio_file": ("m.wav", f.read(), "audio/wav"), "target_file": ("t.wav", g.read(), "audio/wav")}).json()
            assert j["success"], j["error"]
            got["rest"] = j["data"]
        except Exception as e:                           # noqa: BLE001
            errs.append(("rest", repr(e)))
    ts = [threading.Thread(target=ws_client, args=("ws1",)), threading.Thread(target=ws_client, args=("ws2",)), threading.Thread(target=rest_client)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    assert not errs, errs
    for k in ("ws1", "ws2"):
        assert [(s_[0], s_[1]) for s_ in got[k]] == [(s_[0], s_[1]) for s_ in ref_stream], k
        assert np.allclose([s_[2] for s_ in got[k]], [s_[2] for s_ in ref_stream], atol=2e-3)
    d = got["rest"]
    assert d["target_speaker_id"] == spk
    assert [(r["speaker"], r["type"], r["text"]) for r in d["results"]] == [(r["speaker"], r["type"], r["text"]) for r in res]
    assert model.target_embedding is None                 # sessions kept their state to themselves
