"""CPU tests of the serving shell (N4): the reference's REST / WebSocket wire format (main.py:46-288) around a FAKE model —
no kernels involved: multipart upload parsing, response schema, error convention, streaming message sequence."""
import base64
import io
import wave

import numpy as np
import pytest

fastapi = pytest.importorskip("fastapi")
from fastapi.testclient import TestClient  # noqa: E402

from targetdiarization_amd.server import build_response_data, create_app, format_speaker_info  # noqa: E402


class FakeModel:
    def __init__(self):
        self.calls = []

    def infer(self, wav_file, target_file=None, sampling_rate=16000, is_single=False, output_target_audio=True):
        self.calls.append((wav_file.shape, None if target_file is None else target_file.shape, is_single))
        dur = round(wav_file.shape[0] / 16000.0, 3)
        res = [{"speaker": "0", "timerange": [0.0, dur / 2], "text": "a", "type": "single", "score": 0.9},
               {"speaker": "1", "timerange": [dur / 2, dur], "text": "b", "type": "overlap", "score": 0.1},
               {"speaker": "-1", "timerange": [dur, dur], "text": "", "type": "single", "score": -1.0}]
        return "0", res, (np.ones(160, np.float32) * 0.5 if output_target_audio else None)


def wav_bytes(n, sr=16000):
    b = io.BytesIO()
    with wave.open(b, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr)
        w.writeframes((np.arange(n) % 200).astype(np.int16).tobytes())
    return b.getvalue()


def test_rest_wire_format():
    m = FakeModel()
    c = TestClient(create_app(m))
    assert c.get("/health").json()["model_loaded"] is True
    assert c.get("/").json()["endpoints"]["inference"] == "/diarization/infer"
    r = c.post("/diarization/infer?is_single=false", files={"audio_file": ("mix.wav", wav_bytes(32000), "audio/wav"),
                                                            "target_file": ("t.wav", wav_bytes(8000), "audio/wav")})
    j = r.json()
    assert r.status_code == 200 and j["success"] is True and j["error"] is None and j["processing_time"] >= 0
    d = j["data"]
    assert d["target_speaker_id"] == "0" and d["total_speakers"] == 2
    assert [x["speaker_type"] for x in d["results"]] == ["target", "other", "uncertain"]
    assert set(d["results"][0]) == {"speaker", "speaker_type", "timerange", "text", "type", "score"}
    assert d["statistics"] == {"total_duration": 2.0, "target_speaker_duration": 1.0, "other_speakers_duration": 1.0}
    pcm = np.frombuffer(base64.b64decode(d["target_audio_base64"]), dtype=np.int16)
    assert pcm.shape == (160,) and pcm[0] == int(0.5 * 32767)
    assert m.calls == [((32000,), (8000,), False)]
    # no target file, no target audio
    j2 = c.post("/diarization/infer?output_target_audio=false", files={"audio_file": ("mix.wav", wav_bytes(16000), "audio/wav")}).json()
    assert j2["success"] and "target_audio_base64" not in j2["data"] and m.calls[-1][1] is None


def test_rest_error_convention():
    c = TestClient(create_app(None))
    assert c.get("/health").json()["model_loaded"] is False
    assert c.post("/diarization/infer", files={"audio_file": ("a.wav", wav_bytes(100), "audio/wav")}).status_code == 500
    c2 = TestClient(create_app(FakeModel()))
    j = c2.post("/diarization/infer", files={"audio_file": ("a.wav", b"not a wav", "audio/wav")}).json()
    assert j["success"] is False and j["error"].startswith("Inference failed:")       # reported in the body (main.py:222-233)


def test_websocket_stream_sequence():
    m = FakeModel()
    c = TestClient(create_app(m, max_buffer_duration=1.0))
    chunk = base64.b64encode((np.arange(8000) % 100).astype(np.int16).tobytes()).decode()
    with c.websocket_connect("/diarization/stream") as ws:
        ws.send_json({"type": "config", "data": {"has_target_file": True}})
        ws.send_json({"type": "target_audio", "data": chunk})
        ack = ws.receive_json()
        assert ack["type"] == "config_ack" and ack["data"]["target_file_loaded"] is True
        for _ in range(3):
            ws.send_json({"type": "audio_chunk", "data": chunk})       # 0.5 s each: flush after 2 chunks, the third at the end
        ws.send_json({"type": "end"})
        msgs = []
        while True:
            msg = ws.receive_json()
            msgs.append(msg)
            if msg["type"] == "status":
                break
    segs = [x for x in msgs if x["type"] == "segment_result"]
    assert len(segs) == 6 and msgs[-1] == {"type": "status", "message": "completed"}
    assert segs[0]["data"]["segment"]["speaker_type"] == "target" and segs[3]["data"]["segment"]["timerange"][0] == 1.0   # offset of the 2nd flush
    assert [c_[0] for c_ in m.calls] == [(16000,), (8000,)] and m.calls[0][1] == (8000,)


def test_helpers():
    assert format_speaker_info("2", "2") == "target" and format_speaker_info("-1", "2") == "uncertain" and format_speaker_info("1", "2") == "other"
    d = build_response_data("", [], None, True)
    assert d["statistics"]["total_duration"] == 0.0 and d["results"] == [] and "target_audio_base64" not in d


def test_websocket_drives_a_streaming_session():
    """a model with infer_stream (TargetDiarizationStream) is fed chunk by chunk from a worker thread; results are forwarded as they come"""
    class StreamModel(FakeModel):
        def infer_stream(self, gen, target_file=None, sampling_rate=16000, is_single=False, output_target_audio=False):
            t = 0.0
            for i, c in enumerate(gen):
                d = c.shape[0] / 16000.0
                yield "1", [{"speaker": "1" if i % 2 == 0 else "0", "timerange": [t, t + d], "text": f"seg{i}", "type": "single"}], None
                t += d
    c = TestClient(create_app(StreamModel()))
    chunk = base64.b64encode((np.arange(8000) % 100).astype(np.int16).tobytes()).decode()
    with c.websocket_connect("/diarization/stream") as ws:
        ws.send_json({"type": "config", "data": {}})
        assert ws.receive_json()["type"] == "config_ack"
        for _ in range(3):
            ws.send_json({"type": "audio_chunk", "data": chunk})
        ws.send_json({"type": "end"})
        msgs = []
        while True:
            m = ws.receive_json()
            msgs.append(m)
            if m["type"] == "status":
                break
    segs = [m["data"]["segment"] for m in msgs if m["type"] == "segment_result"]
    assert [s_["text"] for s_ in segs] == ["seg0", "seg1", "seg2"] and [s_["speaker_type"] for s_ in segs] == ["target", "other", "target"]
    assert segs[2]["timerange"] == [1.0, 1.5]


def test_two_stream_sessions_and_a_post_share_one_model():
    """ONE model object behind two open WebSocket streams and a REST request (the reference's arrangement, main.py:42,366-367):
    every connection has its own session state (buffer, clock, target embedding) and every result equals the serial one."""
    import threading
    import targetdiarization_amd.target_diarization as td_mod
    from tests.test_stream_session import SR, FakeHotPath, energy_vad, tone

    class CountingHotPath(FakeHotPath):
        depth, max_depth, lock = 0, 0, threading.Lock()

        def encode_streams(self, audios):                 # called inside the model's gpu_lock: never by two threads at once
            cls = CountingHotPath
            with cls.lock:
                cls.depth += 1; cls.max_depth = max(cls.max_depth, cls.depth)
            import time; time.sleep(0.02)
            with cls.lock:
                cls.depth -= 1
            return list(audios)

    saved = td_mod.HotPath
    td_mod.HotPath = CountingHotPath
    try:
        from targetdiarization_amd.target_diarization_stream import TargetDiarizationStream

        def build():
            return TargetDiarizationStream(vad=energy_vad, stream_vad=energy_vad, od_pipeline=lambda a: [], max_buffer_duration=30.0,
                                           decoder=lambda enc: ("x" * max(1, int(np.abs(enc).sum() > 0) * (enc.shape[0] // 1600)), []))
        sil = np.zeros(SR // 2, np.float32)
        streams = {"a": [np.concatenate([tone(200, 1.0), sil]), np.concatenate([tone(500, 1.0), sil])],      # target = the 200 Hz voice
                   "b": [np.concatenate([tone(500, 0.8), sil]), np.concatenate([tone(500, 0.6), sil]), np.concatenate([tone(200, 0.7), sil])]}

        def serial(chunks):
            out = list(build().infer_stream(iter(chunks)))
            return [(r[1][0]["speaker"], r[1][0]["text"], [round(t, 3) for t in r[1][0]["timerange"]]) for r in out]
        want = {k: serial(v) for k, v in streams.items()}
        assert [w[0] for w in want["a"]] == ["1", "0"] and [w[0] for w in want["b"]] == ["1", "1", "0"]

        model = build()
        model.infer = lambda **kw: ("0", [{"speaker": "0", "timerange": [0.0, 1.0], "text": "rest", "type": "single", "score": 0.5}], None)
        app = create_app(model)
        got, errors = {}, []

        def pcm(x):
            return base64.b64encode((x * 32767).astype(np.int16).tobytes()).decode()

        def client(name):
            try:
                c = TestClient(app)
                with c.websocket_connect("/diarization/stream") as ws:
                    ws.send_json({"type": "config", "data": {}})
                    assert ws.receive_json()["type"] == "config_ack"
                    for ch in streams[name]:
                        ws.send_json({"type": "audio_chunk", "data": pcm(ch)})
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
            except Exception as e:                       # noqa: BLE001
                errors.append((name, repr(e)))

        def rest():
            try:
                j = TestClient(app).post("/diarization/infer", files={"audio_file": ("a.wav", wav_bytes(16000), "audio/wav")}).json()
                got["rest"] = j["success"] and j["data"]["results"][0]["text"]
            except Exception as e:                       # noqa: BLE001
                errors.append(("rest", repr(e)))
        ts = [threading.Thread(target=client, args=("a",)), threading.Thread(target=client, args=("b",)), threading.Thread(target=rest)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(60)
        assert not errors, errors
        # int16 round trip of the chunks: same segmentation, text lengths and (to 1 ms) clocks as the serial float run
        for k in ("a", "b"):
            assert [(s_[0], len(s_[1])) for s_ in got[k]] == [(w[0], len(w[1])) for w in want[k]], (k, got[k], want[k])
            assert all(abs(g[2][0] - w[2][0]) < 2e-3 and abs(g[2][1] - w[2][1]) < 2e-3 for g, w in zip(got[k], want[k]))
        assert got["rest"] == "rest"
        assert CountingHotPath.max_depth == 1             # buffers of different sessions were processed one at a time
        assert model.target_embedding is None and not model.vad_buffer      # the shared model object carries no session state
    finally:
        td_mod.HotPath = saved


def test_env_configuration_maps_to_constructor_kwargs(tmp_path, monkeypatch):
    """N4: main.py:105-129 — the reference's .env names -> TargetDiarizationStream kwargs, verbatim"""
    import targetdiarization_amd.target_diarization as td_mod
    from targetdiarization_amd.server import env_to_kwargs, load_dotenv, model_from_env
    from tests.test_stream_session import FakeHotPath
    envfile = tmp_path / ".env"
    envfile.write_text("### TargetDiarization ###\nHF_TOKEN=\n# comment\nCUDA_DEVICE=0\nVERBOSE_LOG=1\nPYANNOTE_CLUSTERING_THRESHOLD=0.0\nASR_ENGINE=paraformer\n"
                       "ASR_MODEL_DIR=iic/asr\nDIARIZATION_PIPELINE_DIR=iic/sd\nOD_MODEL_DIR=pyannote/od\nMDX_WEIGHTS_FILE=\"mdx/w.onnx\"\n"
                       "EMBEDDING_MODEL_DIR=iic/emb\nVAD_MODEL_DIR=iic/vad\nSEPARATER_WEIGHTS_FOLDER=checkpoints/mf2\nRESTORER_WEIGHTS_FOLDER=\n"
                       "TARGET_SIMILARITY_THRESHOLD=0.2\nIS_VAD_BUFFER=0\nMAX_BUFFER_DURATION=12.5\nUSE_ASR_PROMPT=0\nSIMILARITY_THRESHOLD=0.4\n"
                       "LOUDNESS_DIFF_THRESHOLD=12.0\nexport VAD_MIN_SILENCE=0.25\n", encoding="utf-8")
    env = {"SIMILARITY_THRESHOLD": "0.55"}                       # already set: the file does not override it (load_dotenv default)
    assert load_dotenv(str(envfile), env) is True and load_dotenv(str(tmp_path / "missing.env"), env) is False
    kw = env_to_kwargs(env)
    assert kw == {"verbose_log": True, "cuda_device": 0, "target_similarity_threshold": 0.2, "pyannote_clustering_threshold": 0.0,
                  "diarization_pipeline_dir": "iic/sd", "od_model_dir": "pyannote/od", "mdx_weights_file": "mdx/w.onnx",
                  "embedding_model_dir": "iic/emb", "asr_model_dir": "iic/asr", "vad_model_dir": "iic/vad",
                  "separater_weights_folder": "checkpoints/mf2", "restorer_weights_folder": "", "is_vad_buffer": False,
                  "max_buffer_duration": 12.5, "vad_min_silence": 0.25, "use_asr_prompt": False, "similarity_threshold": 0.55,
                  "loudness_diff_threshold": 12.0}
    # nothing set: only the three flags the reference always passes
    assert env_to_kwargs({}) == {"verbose_log": False, "is_vad_buffer": True, "use_asr_prompt": True}
    monkeypatch.setattr(td_mod, "HotPath", FakeHotPath)
    m = model_from_env(str(envfile), dict(env), od_pipeline=lambda a: [])
    assert m.max_buffer_duration == 12.5 and m.is_vad_buffer is False and m.similarity_threshold == 0.55 and m.target_similarity_threshold == 0.2
    assert m.cuda_device == 0 and m.verbose_log is True and m.loudness_diff_threshold == 12.0 and m.vad_min_silence == 0.25
