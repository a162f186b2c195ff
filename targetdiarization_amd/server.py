"""Serving shell: the wire format of the reference's FastAPI app (main.py:46-288) around the MI355X
`TargetDiarization` — `GET /`, `GET /health`, `POST /diarization/infer` (multipart upload: audio_file, optional target_file; query:
sampling_rate, is_single, output_target_audio) and `WS /diarization/stream` (config -> optional target_audio -> audio_chunk* -> end).

Differences, all on the host side: uploads are decoded with the stdlib `wave` module (16-bit PCM; the reference hands the temp file
to audioread) and resampled on the device; the multipart body is parsed with the stdlib `email` parser (python-multipart is not a
dependency); the streaming endpoint drives `model.infer_stream` (target_diarization_stream.TargetDiarizationStream: the reference's
VAD-buffer router with plug-in detectors) in a worker thread like main.py:330-400; a model without `infer_stream` gets fixed
`max_buffer_duration` buffers (default 10 s) through `infer()`.

Concurrency: ONE model serves every request, like main.py:42.  The model objects serialise their GPU calls (`_lib.HandleGuard`)
and `infer()` / each streamed buffer hold the model's `gpu_lock`; here every blocking model call runs in the thread pool (the
event loop stays free: /health answers during a long inference) and every WebSocket connection gets its OWN session state
(`model.session()`), so concurrent clients do not share buffers, clocks or target embeddings.

    app = create_app(model)          # model: targetdiarization_amd.target_diarization.TargetDiarization (or None: 500 on infer)
    uvicorn.run(app, host="0.0.0.0", port=8000)
"""

import base64
import io
import time
import wave
from email.parser import BytesParser
from email.policy import HTTP
from typing import Any, Dict, List, Optional

import numpy as np


def format_speaker_info(speaker_id: str, target_speaker_id: str) -> str:
    """main.py:62-68"""
    if speaker_id == target_speaker_id:
        return "target"
    if speaker_id == "-1":
        return "uncertain"
    return "other"


def audio_to_base64(audio_data: Optional[np.ndarray]) -> str:
    """main.py:72-79: float32 -> int16 (x 32767) -> base64"""
    if audio_data is None:
        return ""
    if audio_data.dtype == np.float32:
        audio_data = (audio_data * 32767).astype(np.int16)
    return base64.b64encode(audio_data.tobytes()).decode("utf-8")


def decode_wav(data: bytes):
    """PCM16 .wav bytes -> (float32 [n] or [n, ch] in [-1, 1], sampling rate)"""
    with wave.open(io.BytesIO(data), "rb") as w:
        if w.getsampwidth() != 2:
            raise ValueError("only 16-bit PCM .wav uploads are decoded here")
        x = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0
        ch, sr = w.getnchannels(), w.getframerate()
    return (x.reshape(-1, ch) if ch > 1 else x), sr


def parse_multipart(content_type: str, body: bytes) -> Dict[str, Dict[str, Any]]:
    """multipart/form-data -> {field name: {"filename": str | None, "data": bytes}}"""
    msg = BytesParser(policy=HTTP).parsebytes(b"Content-Type: " + content_type.encode() + b"\r\nMIME-Version: 1.0\r\n\r\n" + body)
    out = {}
    if not msg.is_multipart():
        return out
    for part in msg.iter_parts():
        name = part.get_param("name", header="content-disposition")
        if name:
            out[name] = {"filename": part.get_filename(), "data": part.get_payload(decode=True) or b""}
    return out


def build_response_data(target_spk: str, final_result: List[dict], target_audio, output_target_audio: bool) -> Dict[str, Any]:
    """main.py:196-213: the `data` object of a successful /diarization/infer response"""
    results = [{"speaker": r["speaker"], "speaker_type": format_speaker_info(r["speaker"], target_spk), "timerange": [float(t) for t in r["timerange"]],
                "text": r["text"], "type": r["type"], "score": float(r.get("score", -1.0))} for r in final_result]
    data = {
        "target_speaker_id": target_spk,
        "total_speakers": len(set(r["speaker"] for r in final_result if r["speaker"] != "-1")),
        "results": results,
        "statistics": {
            "total_duration": round(max(r["timerange"][1] for r in final_result) if final_result else 0.0, 3),
            "target_speaker_duration": round(sum(r["timerange"][1] - r["timerange"][0] for r in final_result if r["speaker"] == target_spk), 3),
            "other_speakers_duration": round(sum(r["timerange"][1] - r["timerange"][0] for r in final_result
                                                 if r["speaker"] != target_spk and r["speaker"] != "-1"), 3),
        },
    }
    if output_target_audio and target_audio is not None:
        data["target_audio_base64"] = audio_to_base64(np.asarray(target_audio))
    return data


def load_dotenv(path: str = ".env", environ=None) -> bool:
    """What main.py:105 gets from python-dotenv (absent here), for the reference's `.env` files (.env.example): KEY=VALUE lines,
    `#` comment lines, optional `export ` prefix and matching quotes; variables already present in the environment win
    (load_dotenv's default override=False).  Returns False when the file does not exist."""
    import os
    environ = os.environ if environ is None else environ
    if not os.path.isfile(path):
        return False
    with open(path, encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#") or "=" not in line:
                continue
            if line.startswith("export "):
                line = line[7:].lstrip()
            key, value = line.split("=", 1)
            key, value = key.strip(), value.strip()
            if len(value) >= 2 and value[0] == value[-1] and value[0] in "\"'":
                value = value[1:-1]
            elif " #" in value:
                value = value.split(" #", 1)[0].rstrip()
            if key and key not in environ:
                environ[key] = value
    return True


def env_to_kwargs(environ=None) -> Dict[str, Any]:
    """main.py:106-129 verbatim in meaning: environment names -> TargetDiarizationStream constructor kwargs; unset variables are left to
    the constructor defaults, EXCEPT the three flags the reference always passes (VERBOSE_LOG == "1"; IS_VAD_BUFFER and
    USE_ASR_PROMPT true unless "0").  (ASR_ENGINE is in .env.example but main.py does not read it — neither does this.)"""
    import os
    env = os.environ if environ is None else environ

    def num(name, cast):
        return cast(env[name]) if env.get(name) is not None else None
    args = {
        "verbose_log": env.get("VERBOSE_LOG") == "1",
        "cuda_device": num("CUDA_DEVICE", int),
        "target_similarity_threshold": num("TARGET_SIMILARITY_THRESHOLD", float),
        "pyannote_clustering_threshold": num("PYANNOTE_CLUSTERING_THRESHOLD", float),
        "diarization_pipeline_dir": env.get("DIARIZATION_PIPELINE_DIR"),
        "od_model_dir": env.get("OD_MODEL_DIR"),
        "mdx_weights_file": env.get("MDX_WEIGHTS_FILE"),
        "embedding_model_dir": env.get("EMBEDDING_MODEL_DIR"),
        "asr_model_dir": env.get("ASR_MODEL_DIR"),
        "vad_model_dir": env.get("VAD_MODEL_DIR"),
        "separater_weights_folder": env.get("SEPARATER_WEIGHTS_FOLDER"),
        "restorer_weights_folder": env.get("RESTORER_WEIGHTS_FOLDER"),
        "is_vad_buffer": env.get("IS_VAD_BUFFER") != "0",
        "max_buffer_duration": num("MAX_BUFFER_DURATION", float),
        "vad_min_silence": num("VAD_MIN_SILENCE", float),
        "use_asr_prompt": env.get("USE_ASR_PROMPT") != "0",
        "similarity_threshold": num("SIMILARITY_THRESHOLD", float),
        "loudness_diff_threshold": num("LOUDNESS_DIFF_THRESHOLD", float),
    }
    return {k: v for k, v in args.items() if v is not None}


def model_from_env(dotenv_path: str = ".env", environ=None, **plugins):
    """main.py:101-137 (the startup hook): `.env` -> constructor kwargs -> TargetDiarizationStream.  `plugins`: what the environment
    cannot carry here — the state dicts (no checkpoint ships with the reference) and the third-party detectors
    (sep_state_dict, spk_state_dict, asr_state_dict, sd_pipeline, od_pipeline, vad, stream_vad, ...)."""
    from .target_diarization_stream import TargetDiarizationStream
    load_dotenv(dotenv_path, environ)
    kwargs = env_to_kwargs(environ)
    print("Init args:", kwargs)
    kwargs.update(plugins)
    return TargetDiarizationStream(**kwargs)


def create_app(model=None, max_buffer_duration: float = 10.0):
    from fastapi import FastAPI, HTTPException, Request, WebSocket, WebSocketDisconnect
    from fastapi.middleware.cors import CORSMiddleware
    from starlette.concurrency import run_in_threadpool

    app = FastAPI(title="Target Diarization API")
    app.add_middleware(CORSMiddleware, allow_origins=["*"], allow_credentials=True, allow_methods=["*"], allow_headers=["*"])
    app.state.model = model

    def to16k(x, sr):
        if x.ndim > 1:
            x = x.mean(axis=1).astype(np.float32)
        if sr != 16000:
            x = app.state.model.hp.ap.audio_resample(x, sr, 16000, output_audio_only=True)
        return x

    @app.get("/")
    async def root():
        return {"message": "Target Diarization API", "version": "1.0.0",
                "endpoints": {"inference": "/diarization/infer", "streaming": "/diarization/stream", "health": "/health"}}

    @app.get("/health")
    async def health_check():
        return {"status": "healthy", "model_loaded": app.state.model is not None, "timestamp": int(time.time())}

    @app.post("/diarization/infer")
    async def diarization_infer(request: Request, sampling_rate: int = 16000, is_single: bool = False, output_target_audio: bool = True):
        start = time.time()
        if app.state.model is None:
            raise HTTPException(status_code=500, detail="Model not loaded")
        try:
            form = parse_multipart(request.headers.get("content-type", ""), await request.body())
            if "audio_file" not in form:
                raise HTTPException(status_code=422, detail="audio_file is required")
            audio, sr = decode_wav(form["audio_file"]["data"])
            target, tsr = None, 16000
            if "target_file" in form and form["target_file"]["data"]:
                target, tsr = decode_wav(form["target_file"]["data"])

            def work():         # (blocking, off the event loop; the model serialises concurrent requests itself)
                a = to16k(audio, sr)
                t = to16k(target, tsr) if target is not None else None
                return app.state.model.infer(wav_file=a, target_file=t, sampling_rate=16000, is_single=is_single,
                                             output_target_audio=output_target_audio)
            target_spk, final_result, target_audio = await run_in_threadpool(work)
            return {"success": True, "data": build_response_data(target_spk, final_result, target_audio, output_target_audio), "error": None,
                    "processing_time": round(time.time() - start, 3)}
        except HTTPException:
            raise
        except Exception as e:                       # main.py:222-233: errors are reported in the body, not as HTTP errors
            return {"success": False, "data": None, "error": f"Inference failed: {e}", "processing_time": round(time.time() - start, 3)}

    @app.websocket("/diarization/stream")
    async def diarization_stream(websocket: WebSocket):
        await websocket.accept()
        try:
            if app.state.model is None:
                await websocket.send_json({"type": "error", "message": "Model not loaded"})
                return
            config = (await websocket.receive_json()).get("data", {})
            target = None
            if config.get("has_target_file", False):
                msg = await websocket.receive_json()
                if msg.get("type") == "target_audio":
                    target = np.frombuffer(base64.b64decode(msg.get("data")), dtype=np.int16).astype(np.float32) / 32767.0
            await websocket.send_json({"type": "config_ack", "data": {"config": config, "target_file_loaded": target is not None}})
            if hasattr(app.state.model, "infer_stream"):
                # the reference's arrangement (main.py:330-400): the session's generator runs in a worker thread, fed through a queue
                import asyncio
                import queue
                import threading
                inq, outq, loop = queue.Queue(), asyncio.Queue(), asyncio.get_running_loop()
                # per-connection session state (buffer, clock, target embedding); a model without `session()` is driven directly
                sess = app.state.model.session() if hasattr(app.state.model, "session") else app.state.model

                def chunks():
                    while True:
                        c = inq.get()
                        if c is None:
                            return
                        yield c

                def worker():
                    try:
                        for spk, res, _ in sess.infer_stream(chunks(), target_file=target, sampling_rate=16000,
                                                             is_single=bool(config.get("is_single", False)), output_target_audio=False):
                            loop.call_soon_threadsafe(outq.put_nowait, ("seg", spk, res))
                    except Exception as e:
                        loop.call_soon_threadsafe(outq.put_nowait, ("err", str(e), None))
                    loop.call_soon_threadsafe(outq.put_nowait, ("done", None, None))

                threading.Thread(target=worker, daemon=True).start()

                async def feed():
                    try:
                        while True:
                            msg = await websocket.receive_json()
                            if msg.get("type") == "audio_chunk":
                                inq.put(np.frombuffer(base64.b64decode(msg.get("data")), dtype=np.int16).astype(np.float32) / 32767.0)
                            elif msg.get("type") == "end":
                                break
                    finally:
                        inq.put(None)

                feeder = asyncio.ensure_future(feed())
                while True:
                    kind, a, res = await outq.get()
                    if kind == "seg":
                        for seg in res:
                            await websocket.send_json({"type": "segment_result", "data": {"target_speaker_id": a, "segment": {
                                "speaker": seg["speaker"], "speaker_type": format_speaker_info(seg["speaker"], a),
                                "timerange": [round(float(seg["timerange"][0]), 3), round(float(seg["timerange"][1]), 3)], "text": seg["text"], "type": seg["type"]}}})
                    elif kind == "err":
                        await websocket.send_json({"type": "error", "message": f"Processing error: {a}"})
                    else:
                        break
                await feeder
                await websocket.send_json({"type": "status", "message": "completed"})
                return
            buf, offset = [], 0.0
            limit = int(float(config.get("max_buffer_duration", max_buffer_duration)) * 16000)

            async def flush():
                nonlocal buf, offset
                if not buf:
                    return
                audio = np.concatenate(buf)
                buf = []
                spk, res, _ = await run_in_threadpool(lambda: app.state.model.infer(
                    wav_file=audio, target_file=target, sampling_rate=16000, is_single=bool(config.get("is_single", False)), output_target_audio=False))
                for seg in res:
                    await websocket.send_json({"type": "segment_result", "data": {"target_speaker_id": spk, "segment": {
                        "speaker": seg["speaker"], "speaker_type": format_speaker_info(seg["speaker"], spk),
                        "timerange": [round(seg["timerange"][0] + offset, 3), round(seg["timerange"][1] + offset, 3)], "text": seg["text"], "type": seg["type"]}}})
                offset += audio.shape[0] / 16000.0
            while True:
                msg = await websocket.receive_json()
                if msg.get("type") == "audio_chunk":
                    buf.append(np.frombuffer(base64.b64decode(msg.get("data")), dtype=np.int16).astype(np.float32) / 32767.0)
                    if sum(b.shape[0] for b in buf) >= limit:
                        await flush()
                elif msg.get("type") == "end":
                    break
            await flush()
            await websocket.send_json({"type": "status", "message": "completed"})
        except WebSocketDisconnect:
            pass
        except Exception as e:
            try:
                await websocket.send_json({"type": "error", "message": f"Processing error: {e}"})
            except Exception:
                pass

    return app
