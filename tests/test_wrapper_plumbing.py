"""CPU: the drop-in boundary -- Separate / BaseWrapper / ProjectFiles / WAV I/O -- with the engine
stubbed out (no kernels): kwargs filtering, cache hit/miss, filenames, separation_info.json, pruning
(reference wrappers/separate.py:233-412, util/data_classes.py:10-67, stem_separator.py:625-677)."""
import json
import os

import numpy as np
import pytest


@pytest.fixture()
def sandbox(tmp_path, monkeypatch):
    from audiolab_amd.handlers import config
    monkeypatch.setattr(config, "output_path", str(tmp_path / "outputs"))
    from audiolab_amd.wrappers.separate import Separate
    Separate._instance = None
    return tmp_path


def make_wav(path, n=4000, sr=44100, seed=0, subtype="FLOAT"):
    from audiolab_amd import wavio
    x = (np.random.default_rng(seed).standard_normal((2, n)) * 0.1).astype(np.float32)
    wavio.write_wav(str(path), x, sr, subtype=subtype)
    return x


def test_wav_roundtrip(tmp_path):
    from audiolab_amd import wavio
    x = make_wav(tmp_path / "a.wav")
    y, sr = wavio.read_wav(str(tmp_path / "a.wav"))
    assert sr == 44100 and y.shape == x.shape and np.array_equal(x, y)          # float32 stems are exact
    make_wav(tmp_path / "b.wav", subtype="PCM_16")
    z, _ = wavio.read_wav(str(tmp_path / "b.wav"))
    assert np.max(np.abs(z - x)) <= 0.5 / 32768 + 1e-9                          # PCM16 quantisation (stem_separator.py:74)
    with pytest.raises(ValueError):
        (tmp_path / "c.wav").write_bytes(b"not a wav")
        wavio.read_wav(str(tmp_path / "c.wav"))
    import struct
    short_fmt = b"RIFF" + struct.pack("<I", 24) + b"WAVE" + b"fmt " + struct.pack("<I", 8) + b"\x01\x00\x02\x00\x44\xac\x00\x00"
    zero_ch = (b"RIFF" + struct.pack("<I", 36) + b"WAVE" + b"fmt " + struct.pack("<I", 16) +
               struct.pack("<HHIIHH", 1, 0, 44100, 0, 0, 16) + b"data" + struct.pack("<I", 0))
    for blob in (short_fmt, zero_ch):                                             # malformed headers: ValueError, not struct.error
        (tmp_path / "d.wav").write_bytes(blob)
        with pytest.raises(ValueError):
            wavio.read_wav(str(tmp_path / "d.wav"))


def test_project_files_contract(sandbox):
    """util/data_classes.py:10-67: directory layout, hash tag, dict bookkeeping, all_outputs filter."""
    import xxhash
    from audiolab_amd.util.data_classes import ProjectFiles
    src = sandbox / "my song.wav"
    make_wav(src)
    p = ProjectFiles(str(src))
    tag = xxhash.xxh64(src.read_bytes()).hexdigest()[:8]
    assert p.file_hash == tag
    assert p.project_dir == os.path.join(str(sandbox / "outputs"), "process", f"my song_{tag}")
    assert p.src_file == os.path.join(p.project_dir, "source", "my song.wav") and os.path.isfile(p.src_file)
    assert p.last_outputs == [] and p.video_sources == {} and p.output_dict == {}
    assert p.file_dict["source"][0] == p.src_file
    a, b = os.path.join(p.project_dir, "stems", "a.wav"), os.path.join(p.project_dir, "stems", "gone.wav")
    os.makedirs(os.path.dirname(a))
    open(a, "wb").close()
    p.add_output("stems", [a, b])
    p.add_output("merge", a)
    assert p.last_outputs == [a] and p.output_dict == {"stems": [a, b], "merge": [a]} and p.file_dict["stems"] == [a, b]
    assert p.all_outputs() == [a]                                                 # existing, non-terminal, no repeats
    again = ProjectFiles(str(src))                                                # files of earlier runs are indexed by folder
    assert again.project_dir == p.project_dir and a in again.file_dict["stems"] and again.output_dict == {}


def test_allowed_kwargs_match_reference_table():
    """SURVEY Appendix B."""
    from audiolab_amd.wrappers.separate import Separate
    ak = Separate.allowed_kwargs
    assert list(ak)[:15] == ["delete_extra_stems", "separate_bg_vocals", "bg_vocal_layers", "vocals_only", "store_reverb_ir",
                             "separate_drums", "separate_woodwinds", "alt_bass_model", "reverb_removal", "echo_removal",
                             "crowd_removal", "noise_removal", "noise_removal_model", "delay_removal_model", "crowd_removal_model"]
    # this build's engine knobs follow the reference's keys and are never rendered
    assert list(ak)[15:] == ["precision", "chunker", "overlap", "num_gpus"] and all(ak[k].render is False for k in list(ak)[15:])
    assert ak["chunker"].field.default == "ola" and ak["precision"].choices == ["fp16", "bf16", "fp32"] and ak["num_gpus"].field.le == 8
    assert ak["vocals_only"].field.default is True and ak["separate_bg_vocals"].field.default is False
    assert ak["bg_vocal_layers"].field.ge == 1 and ak["bg_vocal_layers"].field.le == 10 and ak["bg_vocal_layers"].render is False
    assert ak["reverb_removal"].choices == ["Nothing", "Main Vocals", "All Vocals", "All"]
    assert ak["crowd_removal_model"].field.default == "UVR-MDX-NET_Crowd_HQ_1.onnx"
    s = Separate()
    assert s is Separate() and s.title == "Separate" and s.priority == 1 and s.default is True and s.required is False


def test_process_audio_plumbing_with_stub_engine(sandbox, monkeypatch):
    from audiolab_amd import wavio
    from audiolab_amd.util.data_classes import ProjectFiles
    from audiolab_amd.wrappers import separate as sep_mod
    calls = []

    def fake_separate_music(input_dict, callback=None, **kw):
        calls.append((input_dict, kw))
        outs = []
        for folder, files in input_dict.items():
            for f in files:
                base = os.path.splitext(os.path.basename(f))[0]
                for label in ("(Vocals)", "(Instrumental)"):
                    p = os.path.join(folder, f"{base}__{label}.wav")
                    wavio.write_wav(p, np.zeros((2, 10), np.float32) + 0.25, 44100)
                    outs.append(p)
                open(os.path.join(folder, "tmp_scratch.wav"), "wb").write(b"x")          # an extra file to prune
        if callback:
            callback(0.5, "half", 2)
        return outs
    monkeypatch.setattr(sep_mod, "separate_music", fake_separate_music)
    src1, src2 = sandbox / "songA.wav", sandbox / "TTS_hello.wav"
    make_wav(src1)
    make_wav(src2, seed=1)
    projects = [ProjectFiles(str(src1)), ProjectFiles(str(src2))]
    assert os.path.basename(projects[0].project_dir).startswith("songA_") and len(projects[0].file_hash) == 8
    seen = []
    out = sep_mod.Separate().process_audio(projects, callback=lambda *a: seen.append(a), vocals_only=True,
                                            not_an_option=123, delete_extra_stems=True)
    assert len(calls) == 1 and "not_an_option" not in calls[0][1] and calls[0][1]["vocals_only"] is True
    assert seen == [(0.5, "half", 2)]
    by = {os.path.basename(p.src_file): p for p in out}
    stems = by["songA.wav"].last_outputs
    assert sorted(os.path.basename(s) for s in stems) == ["songA__(Instrumental).wav", "songA__(Vocals).wav"]
    stem_dir = os.path.join(by["songA.wav"].project_dir, "stems")
    assert "tmp_scratch.wav" not in os.listdir(stem_dir)                                   # pruned (:376-386)
    info = json.load(open(os.path.join(stem_dir, "separation_info.json")))
    assert set(info) == {"config", "stems"} and len(info["config"]) == 16 and info["config"]["separate_bg_vocals"] is True
    assert all(set(s) == {"path", "hash"} and len(s["hash"]) == 64 for s in info["stems"])
    assert by["TTS_hello.wav"].last_outputs[0].endswith("TTS_hello(Vocals).wav")           # special input (:247-272)
    # second call: cache hit -> engine not called again (:293-313)
    out2 = sep_mod.Separate().process_audio([ProjectFiles(str(src1))], vocals_only=True)
    assert len(calls) == 1 and sorted(out2[0].last_outputs) == sorted(stems)
    # changed option -> cache miss; tampered stem -> cache miss
    sep_mod.Separate().process_audio([ProjectFiles(str(src1))], vocals_only=False)
    assert len(calls) == 2
    open(stems[0], "ab").write(b"\0")
    sep_mod.Separate().process_audio([ProjectFiles(str(src1))], vocals_only=False)
    assert len(calls) == 3


def test_progress_adapter_accepts_two_arg_callables():
    """the chain API's progress object takes (progress, desc) only."""
    from audiolab_amd.separator.stem_separator import _call_progress
    got = []
    _call_progress(lambda p, d: got.append((p, d)), 0.25, "x", 8)
    _call_progress(lambda p, d, t: got.append((p, d, t)), 0.5, "y", 8)
    assert got == [(0.25, "x"), (0.5, "y", 8)]


def test_patch_separator_is_a_noop_without_audio_separator():
    from audiolab_amd.handlers import patch_separate
    assert patch_separate.patch_separator() is False        # audio_separator is not installed here

    class Obj:
        pass
    o = Obj()
    patch_separate.bind_model_run(o, lambda spek: spek)
    assert o.model_run(3) == 3
