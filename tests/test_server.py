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
