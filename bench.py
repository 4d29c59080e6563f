#!/usr/bin/env python3
"""bench.py -- stems x realtime-factor of the MDX separation hot path on MI355X.

Workload (BASELINE.json configs[1]): MDX-Net UVR 4-stem -- four TFC-TDF U-Nets (L=11, g=48,
dim_f 3072, dim_t 256, n_fft 6144, hop 1024) -- over 5 min of 44.1 kHz stereo per GPU, bf16,
synthetic audio and random-init weights (no dataset / checkpoint is reachable offline).
One step = one full pass of the hot path over the track: for each of the 4 models
STFT -> network -> iSTFT -> stitch (reference mdxnet.py:143-197), inputs resident in HBM.
With N GPUs the track is N x 5 min: model windows are sharded over the ranks and the stem
segments are all-gathered once per model (RCCL), so per-GPU work is fixed ("weak").

python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

SR = 44100
TRACK_SECONDS = 300
N_STEMS = 4
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def conv_flops_per_chunk(cfg, levels) -> float:
    """FLOPs of the 3x3 conv launches of the given U-Net levels, per chunk-forward."""
    total = 0.0
    for i, (c, t, f) in enumerate(cfg.levels()):
        if i in levels:
            total += (1 if i == cfg.n else 2) * cfg.l * 2.0 * 9 * c * c * t * f
    return total


def conv_kernel_levels(cfg, bf16: bool, batch: int):
    """Which U-Net levels run which 3x3 kernel -- mirrors run_conv_dma() in tdfnet.hip:
    level 0 (c = 48) -> persistent register-weight kernel; c = 96 / 144 with T % 8 == 0 and >= 96 8x64 tiles -> the
    8-wave big-tile kernel (NY = 2 / 3); the other levels with F % 64 == 0 -> conv3x3_bf16_kernel<64>.
    Returns {class: [levels]} with classes "regw", "big", "big3", "plain"."""
    lv = cfg.levels()
    out = {"regw": [], "big": [], "big3": [], "plain": []}
    for i, (c, t, f) in enumerate(lv):
        if f % 64:
            continue                                          # TW < 64 tiles: ALSEP_PROF_CONV3X3_SMALL, not reported
        if bf16 and c == 48 and t % 4 == 0 and os.environ.get("ALSEP_CONV_REGW", "1") != "0":
            out["regw"].append(i)
        elif (bf16 and c == 96 and t % 8 == 0 and batch * (t // 8) * (f // 64) >= 96
              and os.environ.get("ALSEP_CONV_BIG", "1") != "0"):
            out["big"].append(i)
        elif (bf16 and c == 144 and t % 8 == 0 and batch * (t // 8) * (f // 64) >= 96
              and os.environ.get("ALSEP_CONV_BIG", "1") != "0" and os.environ.get("ALSEP_CONV_BIG3", "1") != "0"):
            out["big3"].append(i)
        else:
            out["plain"].append(i)
    return out


PMC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")
_PMC = None


def pmc_traffic(kernel_prefix: str, windows_per_launch=None):
    """HBM bytes per launch of a kernel from the committed PMC summary (profiles/pmc_traffic.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this very command; FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for 16-byte-per-lane streams on gfx950).  The file carries the hash of the kernel sources it was
    collected from and the windows per network launch of that run: with any other sources (or no file) the figure is None --
    never a stale number -- and with another launch size it is scaled (a network kernel's traffic is linear in its windows)."""
    global _PMC
    if _PMC is None:
        _PMC = {}
        try:
            from audiolab_amd.buildinfo import source_hash
            d = json.load(open(PMC_FILE))
            if d.get("_build", {}).get("source_hash") == source_hash():
                _PMC = d
        except Exception:
            _PMC = {}
    v = _PMC.get(kernel_prefix, {}).get("hbm_bytes_per_launch")
    if v is None or windows_per_launch is None:
        return v
    wpl = _PMC.get("_build", {}).get("windows_per_launch")
    return v * windows_per_launch / wpl if wpl else None


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, sd, mix_np, n_windows: int):
    """The oracle (CPU restatement of the reference path: numpy STFT/iSTFT/chunker + torch-CPU fp32 network, frames
    batched per segment as mdxnet.py:164-167) timed on the host cores, SURVEY 8(d): thread count chosen by a quick
    sweep on one TFC conv of the network, one warm-up window, then the median of 3 runs over ``n_windows`` model
    windows of one stem.  A baseline, not a target."""
    import statistics
    from oracle import mdx_oracle, tdfnet_oracle
    g = mdx_oracle.MDXGeometry(cfg.dim_f, cfg.dim_t, cfg.n_fft, cfg.hop)
    cores = os.cpu_count() or 1
    # thread sweep on the dominant op of the oracle (a level-0 3x3 conv over one window)
    x = torch.randn(1, cfg.g, cfg.dim_f, cfg.dim_t)
    w = torch.randn(cfg.g, cfg.g, 3, 3)
    best, best_t = cores, float("inf")
    for th in sorted({t for t in (8, 16, 32, 64, 128, cores) if t <= cores}):
        torch.set_num_threads(th)
        with torch.no_grad():
            torch.nn.functional.conv2d(x, w, padding=1)
            t0 = time.perf_counter()
            for _ in range(3):
                torch.nn.functional.conv2d(x, w, padding=1)
            dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = th, dt
    torch.set_num_threads(best)

    def model_run(spek):
        with torch.no_grad():
            return tdfnet_oracle.forward(sd, torch.from_numpy(np.ascontiguousarray(spek, dtype=np.float32)),
                                         cfg.num_blocks, cfg.l, cfg.bn).numpy()

    last = {}

    def run(windows: int) -> float:
        n = windows * g.gen_size - 1                     # exactly `windows` model windows (pad = 1)
        t0 = time.perf_counter()
        out = mdx_oracle.demix(mix_np[:, :n], g, model_run, chunks=0, margin=SR, dtype=np.float32)
        dt = time.perf_counter() - t0
        assert out.shape[-1] == n
        last["out"] = out
        return dt
    warm = run(1)
    reps = 3 if warm * n_windows * 3 <= 45.0 else 1      # keep the default bench run within minutes
    times = [run(n_windows) for _ in range(reps)]
    dt = statistics.median(times)
    seconds = (n_windows * g.gen_size - 1) / SR
    cpu_baseline.stems = last["out"]                     # the fp32 oracle's stems of the sample: the `accuracy` object's reference
    return {"value": round(seconds / dt, 4), "unit": "stems*x_realtime", "cores": best, "kind": "port",
            "host": f"{_cpu_model()} ({cores} logical cores; {best} torch threads picked by a sweep over one 3x3 conv)",
            "sample": f"1 of {N_STEMS} models, first {n_windows} model windows ({seconds:.2f} s of audio), fp32, "
                      f"oracle/mdx_oracle.demix + oracle/tdfnet_oracle.forward; 1 warm-up window ({warm:.1f} s), "
                      f"median of {reps} run(s): {dt:.1f} s wall"}


# ---- supplementary workloads: BASELINE configs[2] / [3] / [4] and single models of the reference's roster --------------------------------
# (category, kernel name(s) of that class, bound, peak, unit): the classes a family's step can be dominated by
def _kernel_classes(lib_mod):
    return [
        ("nn_gemm_h_kernel", lib_mod.PROF_NN_GEMM_H, PEAK_BF16_TFLOPS), ("nn_attn_h_kernel", lib_mod.PROF_NN_ATTN_H, PEAK_BF16_TFLOPS),
        ("nn_conv_hh_kernel (f16 MFMA convolution)", lib_mod.PROF_NN_CONV_H, PEAK_BF16_TFLOPS),
        ("nn_gemm_tn_kernel / nn_bgemm_kernel (f32 MFMA)", lib_mod.PROF_NN_GEMM, PEAK_F32_TFLOPS),
        ("nn_conv2d_tiled_kernel / nn_conv2d_kernel (f32 MFMA)", lib_mod.PROF_NN_CONV, PEAK_F32_TFLOPS),
        ("conv3x3_bf16_m0_kernel / regw (level 0)", lib_mod.PROF_CONV3X3_REGW, PEAK_BF16_TFLOPS),
        ("conv3x3_bf16_mq_kernel (level 1)", lib_mod.PROF_CONV3X3_BIG, PEAK_BF16_TFLOPS),
        ("conv3x3_bf16_big_kernel<3> (level 2)", lib_mod.PROF_CONV3X3_BIG3, PEAK_BF16_TFLOPS),
        ("conv3x3_bf16_kernel<64> (levels 3+)", lib_mod.PROF_CONV3X3, PEAK_BF16_TFLOPS),
    ]


def family_roofline(ctx, lib_mod, run_once, fence):
    """Roofline object of the kernel class with the largest share of one step: one untimed pass per class with that class's launches
    bracketed by HIP events on the launch stream (alsep_profile_*); flops / minimal bytes as the launch sites count them.  The passes
    run the model's units one after the other (one lane): kernels of concurrent lanes would share the chip and stretch each other."""
    best = None
    for name, cat, peak_tf in _kernel_classes(lib_mod):
        ctx.profile_begin(cat)
        run_once()
        fence()
        flops, byts = ctx.profile_work()
        ms, launches = ctx.profile_end()
        if launches and (best is None or ms > best[1]):
            best = (name, ms, launches, flops, byts, peak_tf)
    if best is None:
        return None
    name, ms, launches, flops, byts, peak_tf = best
    tfl, gbs = flops / (ms * 1e-3) / 1e12, byts / (ms * 1e-3) / 1e9
    ridge = peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)
    e = {"kernel": name}
    if flops / max(byts, 1.0) >= ridge:
        e.update({"bound": "mfma", "achieved": round(tfl, 2), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(tfl / peak_tf, 4), "gbs": round(gbs, 1)})
    else:
        e.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                  "tflops": round(tfl, 2)})
    e.update({"traffic": None, "launches": launches, "avg_us": round(ms * 1e3 / launches, 2), "flops_per_launch": flops / launches,
              "bytes_per_launch": byts / launches, "class_ms_per_step": round(ms, 2), "lanes": 1})
    return e


def _timed_cpu(fn, stems: float, seconds: float, what: str):
    """cpu_baseline object: ``fn`` (an oracle run over ``seconds`` of audio yielding ``stems`` stems) timed once after setting the torch
    thread count to the host's cores (capped at 64: the oracle's convolutions stop scaling there, see the mdx4 line's sweep)"""
    cores = os.cpu_count() or 1
    th = min(cores, 64)
    torch.set_num_threads(th)
    t0 = time.perf_counter()
    fn()
    dt = time.perf_counter() - t0
    return {"value": round(stems * seconds / dt, 4), "unit": "stems*x_realtime", "cores": th, "kind": "port",
            "host": f"{_cpu_model()} ({cores} logical cores; {th} torch threads)", "sample": f"{what}; one run: {dt:.1f} s wall"}


def other_workload(args) -> None:
    """Bench lines for BASELINE configs[2] / [3] / [4] and for single models of the reference's roster (``--workload model --model NAME``:
    e.g. the two Roformers the unchanged wrapper runs by default, stem_separator.py:379-381, 998).  Same JSON shape as the mdx4 line,
    with the roofline of the step's dominant kernel class and a bounded CPU baseline (the oracle of that family on the host)."""
    import hashlib
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU; there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    from audiolab_amd import _lib
    from audiolab_amd.engine import MODEL_ROSTER, Separator
    from audiolab_amd.synth import synth_mix
    device = torch.device("cuda", local_rank)
    ctx = _lib.Context(device)
    wl = args.workload
    weak = args.scaling == "weak"
    if args.batch <= 0:
        args.batch = 32                                        # windows / chunks per network launch of the MDX-Net models here
    seed_of = lambda name: int.from_bytes(hashlib.sha256(name.encode()).digest()[:4], "little")
    cpu_fn = None                                              # () -> cpu_baseline object, rank 0 at N = 1 only

    def one_lane_engine(**kw):                                 # the engine of the roofline passes: one unit at a time, plain launches
        # The runners read these when a model is LOADED (which may be after this returns), so they stay set: the timed region is over,
        # and the engine it used has its runners already.  Launches replayed from a HIP graph would not pass through the profiling brackets.
        os.environ["ALSEP_RUNNER_LANES"] = os.environ["ALSEP_DEMUCS_LANES"] = "1"
        os.environ["ALSEP_RUNNER_GRAPH"] = "0"
        return Separator(ctx=ctx, allow_synthetic=True, **kw)

    if wl == "demucs6":                                       # configs[2]: htdemucs 6-stem, 10 min, overlap 0.25, segments sharded
        seconds = args.seconds if args.seconds != TRACK_SECONDS else 600
        n = seconds * SR * (world if weak else 1)
        mix = torch.from_numpy(synth_mix(n)).to(device)
        eng = Separator(ctx=ctx, use_autocast=False, allow_synthetic=True, sharded=world > 1)
        eng.load_model("htdemucs_6s.yaml")
        stems, audio_s, sr = 6, n / SR, SR
        desc = f"htdemucs_6s (HTDemucs 6 sources, fp32), {seconds} s 44.1 kHz stereo {'per GPU' if weak else 'in total'}, shifts 2, overlap 0.25"
        dtype_name, sharding = "f32", f"(shift, segment) units/{world} + all_reduce of the weighted sums"

        def step():
            return eng.separate_array(mix)

        def make_prof():
            e1 = one_lane_engine(use_autocast=False)
            e1.load_model("htdemucs_6s.yaml")
            short = mix[:, : min(n, 60 * SR)]
            return lambda: e1.separate_array(short)

        def cpu_fn():
            from oracle import htdemucs_oracle as ho
            ocfg = ho.HTDemucsConfig()
            sd = ho.synthetic_state_dict(ocfg, seed_of("htdemucs_6s.yaml"))
            k = ocfg.segment_samples
            x = torch.from_numpy(synth_mix(k))
            return _timed_cpu(lambda: ho.separate(ocfg, sd, x, shifts=2, overlap=0.25, seed=0), 6, k / SR,
                              f"oracle/htdemucs_oracle.separate on one {k / SR:.1f} s segment (two shifted passes), torch-CPU fp32")
    elif wl == "tracks":                                      # configs[3]: a batch of tracks, MDX ensemble + Demucs per track, replicas
        from audiolab_amd.separator.stem_separator import EnsembleDemucsMDXMusicSeparationModel
        seconds = args.seconds if args.seconds != TRACK_SECONDS else 180
        n = seconds * SR
        tracks = [torch.from_numpy(synth_mix(n, seed=1000 + rank * 100 + k)).to(device) for k in range(args.tracks)]
        # configs[3] is "MDX+Demucs": the roster is cut to the MDX-Net files and htdemucs, so that the first two ensemble members the
        # orchestrator finds are the reference's MDX-Net vocal models (with the full roster they are its two Roformers)
        mdx_demucs = {k: v for k, v in MODEL_ROSTER.items() if k.endswith(".onnx") or v[0] == "demucs"}
        eng = Separator(ctx=ctx, dtype=torch.bfloat16, allow_synthetic=True, max_batch=args.batch, roster=mdx_demucs)
        model = EnsembleDemucsMDXMusicSeparationModel({"ensemble_strength": 2, "vocals_only": False}, separator=eng)
        stems, audio_s, sr = 7, args.tracks * world * n / SR, SR
        desc = (f"{args.tracks} tracks x {seconds} s per GPU: 2 MDX-Net vocal models (n_fft 7680, bf16) blended + de-bleed, then "
                f"htdemucs_6s on the mix (fp32): vocals, instrumental + 5 Demucs stems per track")
        dtype_name, sharding = "bf16+f32", f"track replicas x{world}, no data-path collective"

        def run_tracks(m, tl):
            files = [{"base_name": f"t{k}", "mix": t, "sr": SR, "output_folder": "/mem"} for k, t in enumerate(tl)]
            res = m._ensemble_separate_all(files)
            m._multistem_separation_all(res)
            return res

        def step():
            return run_tracks(model, tracks)

        def make_prof():
            e1 = one_lane_engine(dtype=torch.bfloat16, max_batch=args.batch, roster=mdx_demucs)
            m1 = EnsembleDemucsMDXMusicSeparationModel({"ensemble_strength": 2, "vocals_only": False}, separator=e1)
            return lambda: run_tracks(m1, tracks[:1])

        def cpu_fn():
            from oracle import htdemucs_oracle as ho, mdx_oracle, tdfnet_oracle
            from audiolab_amd.synth import synthetic_state_dict
            cfg = mdx_demucs["UVR-MDX-NET-Voc_FT.onnx"][2]
            g = mdx_oracle.MDXGeometry(cfg.dim_f, cfg.dim_t, cfg.n_fft, cfg.hop)
            k = g.gen_size - 1                                 # one model window
            x = synth_mix(k)
            sds = [synthetic_state_dict(cfg, seed=seed_of(m)) for m in ("UVR-MDX-NET-Voc_FT.onnx", "Kim_Vocal_2.onnx")]
            ocfg = ho.HTDemucsConfig()
            dsd = ho.synthetic_state_dict(ocfg, seed_of("htdemucs_6s.yaml"))

            def run():
                for sd in sds:
                    mdx_oracle.demix(x, g, lambda sp, sd=sd: tdfnet_oracle.forward(sd, torch.from_numpy(np.ascontiguousarray(sp, dtype=np.float32)),
                                                                                   cfg.num_blocks, cfg.l, cfg.bn).numpy(), chunks=0, margin=SR, dtype=np.float32)
                ho.separate(ocfg, dsd, torch.from_numpy(x), shifts=2, overlap=0.25, seed=0)
            return _timed_cpu(run, 7, k / SR, f"one {k / SR:.2f} s track through both MDX-Net vocal models (oracle/mdx_oracle + tdfnet_oracle, one "
                                              f"window each) and htdemucs_6s (oracle/htdemucs_oracle), torch-CPU fp32; blend / de-bleed not timed")
    elif wl == "longform":                                    # configs[4]: 60 min 48 kHz 8 channels, overlap 0.75
        seconds = args.seconds if args.seconds != TRACK_SECONDS else 3600
        sr = 48000
        n = seconds * sr * (world if weak else 1)
        mix8 = torch.from_numpy(np.concatenate([synth_mix(n, sr=sr, seed=50 + c) for c in range(4)])).to(device)
        lf_roster = {"longform_vocals.onnx": ("Vocals", "Instrumental", _bench_cfg())}
        eng = Separator(ctx=ctx, dtype=torch.float16, allow_synthetic=True, max_batch=args.batch, chunker="ola", overlap=0.75,
                        sharded=world > 1, roster=lf_roster)
        eng.load_model("longform_vocals.onnx")
        stems, audio_s = 2, n / sr
        desc = (f"8 channels (4 stereo pairs) x {seconds} s at 48 kHz (native rate), one MDX-Net model (bench geometry, fp16 storage + f16 MFMA), "
                f"Hann overlap-add at overlap 0.75, normalisation 0.9 + spectral inversion for the second stem, chunks sharded")
        dtype_name, sharding = "f16", f"chunks/{world} + all_reduce of the seam sums"

        def step():
            return eng.separate_array(mix8)

        def make_prof():
            short = mix8[:2, : min(n, 120 * sr)].contiguous()
            return lambda: eng.separate_array(short)

        def cpu_fn():
            from oracle import mdx_oracle, tdfnet_oracle
            from audiolab_amd.synth import synthetic_state_dict
            cfg = _bench_cfg()
            sd = synthetic_state_dict(cfg, seed=seed_of("longform_vocals.onnx"))
            g = mdx_oracle.MDXGeometry(cfg.dim_f, cfg.dim_t, cfg.n_fft, cfg.hop)
            k = 100000
            x = synth_mix(k, sr=sr, seed=50)
            run = lambda sp: tdfnet_oracle.forward(sd, torch.from_numpy(np.ascontiguousarray(sp, dtype=np.float32)), cfg.num_blocks, cfg.l, cfg.bn).numpy()
            return _timed_cpu(lambda: mdx_oracle.separate_ola(x, g, run, overlap=0.75, compensate=1.0), 2, k / sr,
                              f"oracle/mdx_oracle.separate_ola on one stereo pair, {k / sr:.2f} s ({len(mdx_oracle.ola_plan(k, g, 0.75)['starts'])} "
                              f"chunks at overlap 0.75), torch-CPU fp32")
    else:                                                     # one model of the roster (--model), e.g. the default ensemble's Roformers
        name = args.model
        if name not in MODEL_ROSTER:
            raise SystemExit(f"--model {name!r} is not in the roster: {sorted(MODEL_ROSTER)}")
        seconds = args.seconds if args.seconds != TRACK_SECONDS else 120
        sr = SR
        n = seconds * SR
        mix = torch.from_numpy(synth_mix(n)).to(device)
        half = args.dtype != "f32"
        eng = Separator(ctx=ctx, use_autocast=half, allow_synthetic=True)
        eng.load_model(name)
        fam = MODEL_ROSTER[name][0] if isinstance(MODEL_ROSTER[name][0], str) and MODEL_ROSTER[name][0] in ("roformer", "mdx23c", "vr", "demucs") else "mdx"
        probe = eng.separate_array(mix[:, : min(n, 10 * SR)])
        stems, audio_s = len(probe), n / SR
        mode = "IEEE-half MFMA operands (the reference's use_autocast=True)" if (half and fam in ("roformer", "mdx", "mdx23c")) else "fp32"
        desc = f"{name} ({fam}), {seconds} s 44.1 kHz stereo, {mode}, {stems} stems out, world {world}: replicas"
        dtype_name, sharding = ("f16" if (half and fam in ("roformer", "mdx", "mdx23c")) else "f32"), f"replicas x{world}"

        def step():
            return eng.separate_array(mix)

        def make_prof():
            e1 = one_lane_engine(use_autocast=half)
            e1.load_model(name)
            short = mix[:, : min(n, 30 * SR)]
            return lambda: e1.separate_array(short)

        if fam == "roformer":
            def cpu_fn():
                import dataclasses
                from oracle import roformer_oracle as ro
                ocfg = ro.RoformerConfig(**dataclasses.asdict(MODEL_ROSTER[name][1]))
                sd = ro.synthetic_state_dict(ocfg, seed_of(name))
                x = torch.from_numpy(synth_mix(ocfg.chunk_size))
                return _timed_cpu(lambda: ro.forward(ocfg, sd, x[None]), stems, ocfg.chunk_size / SR * (ocfg.chunk_size // ocfg.num_overlap) / ocfg.chunk_size,
                                  f"oracle/roformer_oracle.forward on one {ocfg.chunk_size / SR:.0f} s chunk (the runner advances "
                                  f"{ocfg.chunk_size // ocfg.num_overlap / SR:.0f} s per chunk: the rate counts that), torch-CPU fp32")
        elif fam == "mdx23c":
            def cpu_fn():
                import dataclasses
                from oracle import mdx23c_oracle as m3
                ocfg = m3.MDX23CConfig(**dataclasses.asdict(MODEL_ROSTER[name][1]))
                sd = m3.synthetic_state_dict(ocfg, seed_of(name))
                x = torch.from_numpy(synth_mix(ocfg.chunk_size))
                return _timed_cpu(lambda: m3.forward(ocfg, sd, x[None]), stems, (ocfg.chunk_size // ocfg.num_overlap) / SR,
                                  f"oracle/mdx23c_oracle.forward on one {ocfg.chunk_size / SR:.1f} s chunk (the runner advances "
                                  f"{ocfg.chunk_size // ocfg.num_overlap / SR:.2f} s per chunk: the rate counts that), torch-CPU fp32")

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    roofline = cpu = None
    if rank == 0 and world == 1:
        run_once = make_prof()
        run_once()                                             # builds the one-lane engine's plans / workspaces
        fence()
        roofline = family_roofline(ctx, _lib, run_once, fence)
        if cpu_fn is not None and not args.no_cpu_baseline:
            cpu = cpu_fn()
    if rank == 0:
        mult = 4 if wl == "longform" else 1                    # stereo pairs count as separate 2-channel programmes
        print(json.dumps({
            "metric": "stems*realtime-factor (44.1kHz stereo)", "value": round(stems * mult * audio_s * args.steps / dt, 2),
            "unit": "stems*x_realtime", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": dtype_name, "data": "synthetic",
            "config": {"workload": desc, "stems": stems, "audio_seconds": audio_s, "sample_rate": sr, "sharding": sharding},
            "realtime_factor": round(mult * audio_s * args.steps / dt, 2),
            "roofline": roofline, "cpu_baseline": cpu}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _bench_cfg():
    from audiolab_amd.tdfnet import TDFNetConfig
    return TDFNetConfig()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--batch", type=int, default=0,
                    help="model windows per network launch (0 = all of a rank's windows in one launch, at most 64: 18 GiB of workspace per "
                         "model at 52 windows; profiles/r02_batch_sweep.txt: 8 -> 220, 16 -> 216, 26 -> 212, 52 -> 207 ms per step)")
    ap.add_argument("--seconds", type=int, default=TRACK_SECONDS, help="audio seconds per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-precision", action="store_true",
                    help="skip the `precision` / `accuracy` objects (full steps in the other storage types after the timed region; N = 1 only)")
    ap.add_argument("--cpu-windows", type=int, default=2)
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="strong (default): the --seconds track in total, its model windows sharded over the N GPUs -- what 'scaling to 8 GPUs' "
                         "means for one track (52 windows: 6-7 per rank at N = 8); weak: N x --seconds of audio on N GPUs (per-GPU work fixed). "
                         "At N > 1 the line carries the other mode too (`other_scaling`), measured after the timed region.")
    ap.add_argument("--model", default="vocals_mel_band_roformer.ckpt", help="roster name of the model workload")
    ap.add_argument("--workload", default="mdx4", choices=["mdx4", "demucs6", "tracks", "longform", "model"],
                    help="mdx4 (default): BASELINE configs[1], the graded line.  Supplementary lines for the other configs: demucs6 = "
                         "configs[2] (htdemucs 6-stem, segments sharded over the ranks), tracks = configs[3] (batch of tracks per GPU, MDX "
                         "ensemble + htdemucs, replicas), longform = configs[4] (48 kHz 8-channel, Hann overlap-add at 0.75, chunks sharded)")
    ap.add_argument("--tracks", type=int, default=8, help="tracks per GPU of the tracks workload")
    args = ap.parse_args()
    if args.workload != "mdx4":
        return other_workload(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU; there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from audiolab_amd import _lib
    from audiolab_amd.mdx import Predictor
    from audiolab_amd.synth import synth_mix, synthetic_state_dict
    from audiolab_amd.tdfnet import TDFNet, TDFNetConfig

    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    device = torch.device("cuda", local_rank)
    ctx = _lib.Context(device)
    cfg = TDFNetConfig()
    n_samples = args.seconds * SR * (world if args.scaling == "weak" else 1)
    mix_np = synth_mix(n_samples)
    mix = torch.from_numpy(mix_np).to(device)
    sds = [synthetic_state_dict(cfg, seed=s) for s in range(N_STEMS)]
    if args.batch <= 0:                                       # this rank's window count (as computed after the timed region), capped
        from audiolab_amd.dist import window_range as _wr
        _gen = cfg.hop * (cfg.dim_t - 1) - cfg.n_fft
        _lo, _hi = _wr(n_samples // _gen + 1, world, rank)
        args.batch = max(1, min(_hi - _lo, 64))
    nets = [TDFNet(cfg, sd, ctx=ctx, dtype=dtype, max_batch=args.batch) for sd in sds]
    pargs = types.SimpleNamespace(margin=SR, chunks=0, denoise=False, dim_f=cfg.dim_f, dim_t=8, n_fft=cfg.n_fft)
    preds = [Predictor(pargs, net, ctx=ctx, max_batch=0, sharded=world > 1) for net in nets]

    def step():
        if world > 1:                                       # keep each model's all-gather in flight under the next model
            pending = [p.demix(mix, defer=True) for p in preds]
            return [f() for f in pending]
        return [p.demix(mix) for p in preds]

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        stems = step()
    fence()
    half = dtype in (torch.bfloat16, torch.float16)          # the f16 build runs the same kernels (tdfnet_f16.hip)
    klevels = conv_kernel_levels(cfg, half, args.batch)
    KCAT = {"big": _lib.PROF_CONV3X3_BIG, "big3": _lib.PROF_CONV3X3_BIG3, "regw": _lib.PROF_CONV3X3_REGW, "plain": _lib.PROF_CONV3X3}
    # the dominant kernel = the 3x3 class with the largest share of a step: one untimed pass per class decides which one the HIP
    # events bracket during the timed steps
    share = {}
    for cls in KCAT:
        if klevels[cls]:
            ctx.profile_begin(KCAT[cls])
            step()
            fence()
            share[cls] = ctx.profile_end()[0]
    # the level-1 class has three kernels behind one profiling category: report the one that ran (launch counts by name)
    big_name = next((n for n in ("conv3x3_bf16_mq_kernel", "conv3x3_bf16_mny_kernel<2>") if ctx.launch_count(n) > 0),
                    "conv3x3_bf16_big_kernel<2>")
    big3_name = "conv3x3_bf16_mny_kernel<3>" if ctx.launch_count("conv3x3_bf16_mny_kernel<3>") > 0 else "conv3x3_bf16_big_kernel<3>"
    l0_name = "conv3x3_bf16_m0_kernel" if ctx.launch_count("conv3x3_bf16_m0_kernel") > 0 else "conv3x3_bf16_regw_kernel<1>"
    KCLASS = {"big": (big_name, KCAT["big"]), "big3": (big3_name, KCAT["big3"]), "regw": (l0_name, KCAT["regw"]),
              "plain": ("conv3x3_bf16_kernel<64>" if half else "conv3x3_kernel<f32,16,48,64>", KCAT["plain"])}
    primary = max(share, key=share.get) if share else "plain"
    ctx.profile_begin(KCLASS[primary][1])
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stems = step()
    fence()
    dt = time.perf_counter() - t0
    conv_ms, conv_launches = ctx.profile_end()
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    assert stems[0].shape == (1, 2, n_samples) and bool(torch.isfinite(stems[0]).all())
    # the other scaling mode, after the timed region (N > 1 only; at N = 1 the two coincide): same models, the other track length
    other_scaling = None
    if world > 1:
        o_mode = "weak" if args.scaling == "strong" else "strong"
        o_samples = args.seconds * SR * (world if o_mode == "weak" else 1)
        o_mix = torch.from_numpy(synth_mix(o_samples)).to(device)
        from audiolab_amd.dist import window_range as _wr2
        _gen2 = cfg.hop * (cfg.dim_t - 1) - cfg.n_fft
        _lo2, _hi2 = _wr2(o_samples // _gen2 + 1, world, rank)
        for net in nets:
            net.max_batch = max(1, min(_hi2 - _lo2, 64))

        def o_step():
            pending = [p.demix(o_mix, defer=True) for p in preds]
            return [f() for f in pending]
        o_step()
        fence()
        o_steps = max(1, min(args.steps, 3))
        t0 = time.perf_counter()
        for _ in range(o_steps):
            o_out = o_step()
        fence()
        o_dt = time.perf_counter() - t0
        t = torch.tensor([o_dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        o_dt = float(t[0])
        other_scaling = {"scaling": o_mode, "value": round(N_STEMS * (o_samples / SR) * o_steps / o_dt, 2), "ms_per_step": round(o_dt / o_steps * 1e3, 2),
                         "steps": o_steps, "audio_seconds": o_samples / SR, "windows_per_rank": int(_hi2 - _lo2)}
        del o_out, o_mix
        for net in nets:
            net.max_batch = args.batch
        torch.cuda.empty_cache()

    # windows this rank pushed through the network per step (per model)
    gen = cfg.hop * (cfg.dim_t - 1) - cfg.n_fft
    n_win = n_samples // gen + 1
    from audiolab_amd.dist import window_range
    w_lo, w_hi = window_range(n_win, world, rank)
    es = 2 if half else 4
    peak = PEAK_BF16_TFLOPS if half else PEAK_F32_TFLOPS

    def conv_entry(cls, ms, launches, passes):
        """Roofline object of one 3x3-conv kernel class from its HIP-event time over `passes` steps."""
        levels = klevels[cls]
        fl = conv_flops_per_chunk(cfg, levels) * (w_hi - w_lo) * N_STEMS * passes
        byts = sum(2.0 * c * t * f * es * (1 if i == cfg.n else 2) * cfg.l for i, (c, t, f) in enumerate(cfg.levels())
                   if i in levels) * (w_hi - w_lo) * N_STEMS * passes        # each conv reads X and writes Y once
        tfl = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        gbs = byts / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        e = {"kernel": KCLASS[cls][0]}
        if cls == "regw":                                   # c = 48: 216 flop/B < the 312 flop/B ridge -> HBM side
            e.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                      "frac": round(gbs / PEAK_HBM_GBS, 4), "tflops": round(tfl, 1)})
        else:
            e.update({"bound": "mfma", "achieved": round(tfl, 2), "peak": peak, "unit": "TFLOP/s",
                      "frac": round(tfl / peak, 4), "gbs": round(gbs, 1)})
        e.update({"traffic": pmc_traffic(KCLASS[cls][0], args.batch) if dtype == torch.bfloat16 else None,
                  "launches": launches, "avg_us": round(ms * 1e3 / max(launches, 1), 2),
                  "flops_per_launch": fl / max(launches, 1), "bytes_per_launch": byts / max(launches, 1),
                  "levels": levels})
        return e

    roofline = conv_entry(primary, conv_ms, conv_launches, args.steps)
    # the other 3x3 kernel classes, one extra untimed pass each
    other = {}
    for cls in ("regw", "big", "big3", "plain"):
        if cls == primary or not klevels[cls]:
            continue
        ctx.profile_begin(KCLASS[cls][1])
        step()
        fence()
        ms, launches = ctx.profile_end()
        e = conv_entry(cls, ms, launches, 1)
        other[e.pop("kernel")] = e

    # per-stage HBM rooflines (outside the timed region): STFT and iSTFT over this rank's windows
    stages = {}

    def fft_stage_lines(plan, tag):
        """STFT / iSTFT of `plan` over this rank's windows of the track: HIP-event time of 10 launches each"""
        p_gen = plan.chunk_size - 2 * plan.trim
        nb = min(n_samples // p_gen + 1, 52)
        pad_len = plan.trim * 2 + nb * p_gen + plan.chunk_size
        buf = torch.zeros((2, pad_len), device=device)
        buf[:, plan.trim:plan.trim + min(n_samples, pad_len - 2 * plan.trim)] = mix[:, :min(n_samples, pad_len - 2 * plan.trim)]
        spec = plan.stft_strided(buf, pad_len, p_gen, nb, dtype, _lib.LAYOUT_NHWC)
        outb = torch.empty((2, nb * p_gen), device=device)
        torch.cuda.synchronize()
        for name, cat in (("stft", _lib.PROF_STFT), ("istft", _lib.PROF_ISTFT)):
            reps = 10
            ctx.profile_begin(cat)
            for _ in range(reps):
                if name == "stft":
                    plan.stft_strided(buf, pad_len, p_gen, nb, dtype, _lib.LAYOUT_NHWC, out=spec)
                else:
                    plan.istft_strided(spec, _lib.LAYOUT_NHWC, outb, nb * p_gen, p_gen, plan.trim, plan.chunk_size - plan.trim, nb * p_gen)
            ms, launches = ctx.profile_end()
            spec_bytes = 4 * plan.dim_f * plan.dim_t * es
            if name == "stft":
                alg = 2 * plan.chunk_size * 4 + spec_bytes                     # PCM read + spec write
            else:
                alg = spec_bytes + 3 * 2 * plan.chunk_size * 4                 # spec read + acc/div read-modify-write
            gbs = alg * nb * reps / (ms * 1e-3) / 1e9
            # PMC HBM bytes per launch of the stage's kernel, scaled to this launch's chunk count (the committed pass ran
            # the same 52-chunk launches; traffic is linear in the chunk count)
            r2 = plan.n_fft // 256                                             # the three-pass kernels are keyed by their radix (scripts/pmc_summary.py)
            tr = pmc_traffic(f"stft_r16_kernel<{r2}>" if name == "stft" else f"istft_r16_kernel<{r2}>") if (not tag and plan.n_fft in (4096, 6144)) else None
            stages[name + tag] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                  "frac": round(gbs / PEAK_HBM_GBS, 4), "n_fft": plan.n_fft, "dim_f": plan.dim_f,
                                  "bytes_per_chunk": alg, "chunks_per_launch": nb,
                                  "us_per_launch": round(ms * 1e3 / max(launches, 1), 2), "us_per_chunk": round(ms * 1e3 / max(launches, 1) / nb, 3),
                                  "traffic": None if tr is None else tr * nb / float(_PMC.get("_build", {}).get("windows_per_launch") or 52),
                                  "algorithmic_bytes_per_launch": alg * nb}

    fft_stage_lines(preds[0].model_.plan, "")
    # the front end as the step actually runs it: STFT + the network's first 1x1 convolution in ONE kernel (no spectrogram in HBM).
    # Algorithmic bytes: the PCM read + the level-0 activation the network needs anyway (dim_t x dim_f x 48 channels).
    if half:
        plan0 = preds[0].model_.plan
        p_gen = plan0.chunk_size - 2 * plan0.trim
        nb = min(n_samples // p_gen + 1, args.batch)
        pad_len = plan0.trim * 2 + nb * p_gen + plan0.chunk_size
        buf = torch.zeros((2, pad_len), device=device)
        buf[:, plan0.trim:plan0.trim + min(n_samples, pad_len - 2 * plan0.trim)] = mix[:, :min(n_samples, pad_len - 2 * plan0.trim)]
        if nets[0].forward_pcm(plan0, buf, pad_len, p_gen, nb) is not None:
            torch.cuda.synchronize()
            reps = 5
            ctx.profile_begin(_lib.PROF_STFT)
            for _ in range(reps):
                nets[0].forward_pcm(plan0, buf, pad_len, p_gen, nb)
            ms, launches = ctx.profile_end()
            alg = 2 * plan0.chunk_size * 4 + plan0.dim_f * plan0.dim_t * cfg.g * es
            gbs = alg * nb * reps / (ms * 1e-3) / 1e9
            stages["stft_first_conv"] = {"kernel": "stft_r16_kernel<FUSE> (STFT + first 1x1 conv + BN + ReLU)", "bound": "hbm", "achieved": round(gbs, 1),
                                         "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "n_fft": plan0.n_fft,
                                         "bytes_per_chunk": alg, "chunks_per_launch": nb, "us_per_launch": round(ms * 1e3 / max(launches, 1), 2),
                                         "us_per_chunk": round(ms * 1e3 / max(launches, 1) / nb, 3),
                                         "traffic": pmc_traffic(f"stft_r16_kernel<{plan0.n_fft // 256}><fused>", nb),
                                         "algorithmic_bytes_per_launch": alg * nb,
                                         "note": "replaces stages.stft + first_conv_kernel in the timed step; bit-identical to them"}
        del buf
    if cfg.n_fft != 7680:                                    # the geometry of the reference's own vocal models (Voc_FT, Kim_Vocal_*: n_fft 7680, dim_f 3072)
        from audiolab_amd.mdx import StftPlan
        plan7 = StftPlan(ctx, 7680, 1024, 3072, 256)
        fft_stage_lines(plan7, "_7680")
        if half and cfg.dim_f == plan7.dim_f and cfg.dim_t == plan7.dim_t:   # the fused front end at that geometry (same first layer)
            p_gen = plan7.chunk_size - 2 * plan7.trim
            nb = min(n_samples // p_gen + 1, args.batch)
            pad_len = plan7.trim * 2 + nb * p_gen + plan7.chunk_size
            buf = torch.zeros((2, pad_len), device=device)
            buf[:, plan7.trim:plan7.trim + min(n_samples, pad_len - 2 * plan7.trim)] = mix[:, :min(n_samples, pad_len - 2 * plan7.trim)]
            import dataclasses as _dc
            net7 = TDFNet(_dc.replace(cfg, n_fft=7680), sds[0], ctx=ctx, dtype=dtype, max_batch=args.batch)   # same weights: only the front end differs
            if net7.forward_pcm(plan7, buf, pad_len, p_gen, nb) is not None:
                torch.cuda.synchronize()
                reps = 5
                ctx.profile_begin(_lib.PROF_STFT)
                for _ in range(reps):
                    net7.forward_pcm(plan7, buf, pad_len, p_gen, nb)
                ms, launches = ctx.profile_end()
                alg = 2 * plan7.chunk_size * 4 + plan7.dim_f * plan7.dim_t * cfg.g * es
                gbs = alg * nb * reps / (ms * 1e-3) / 1e9
                stages["stft_first_conv_7680"] = {"kernel": "stft_r16_kernel<30, FUSE>", "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS,
                                                  "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "n_fft": 7680, "bytes_per_chunk": alg,
                                                  "chunks_per_launch": nb, "us_per_launch": round(ms * 1e3 / max(launches, 1), 2),
                                                  "us_per_chunk": round(ms * 1e3 / max(launches, 1) / nb, 3),
                                                  "traffic": pmc_traffic("stft_r16_kernel<30><fused>", nb),
                                                  "algorithmic_bytes_per_launch": alg * nb}
            del buf, net7

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, sds[0], mix_np, args.cpu_windows)

    # `precision`: the same step (4 models x the 5-min track) in every storage type, after the timed region; `accuracy`: model 0 over the
    # CPU baseline's sample (its first windows) in every storage type against the fp32 oracle's stems of that sample -- so that the
    # throughput of the mode that meets the north_star's 1e-4 (fp32) and the error of the modes that are faster travel in ONE line.
    precision, accuracy = None, None
    if rank == 0 and world == 1 and not args.no_precision:
        n_acc = args.cpu_windows * gen - 1
        ref = getattr(cpu_baseline, "stems", None) if cpu is not None else None
        precision = {args.dtype: {"ms_per_step": round(dt / args.steps * 1e3, 2), "value": round(N_STEMS * (n_samples / SR) * args.steps / dt, 2),
                                  "steps": args.steps, "windows_per_launch": args.batch}}
        accuracy = None if ref is None else {
            "reference": f"oracle/mdx_oracle.demix + oracle/tdfnet_oracle.forward (torch-CPU fp32), model 0 of {N_STEMS}, first "
                         f"{args.cpu_windows} model windows ({n_acc / SR:.2f} s)", "peak": round(float(np.max(np.abs(ref))), 4)}

        def acc_entry(got):
            d = (got.astype(np.float64) - ref.astype(np.float64))
            r = float(np.sqrt((d ** 2).sum() / (ref.astype(np.float64) ** 2).sum()))
            return {"rel_l2": float(f"{r:.4g}"), "sdr_db": round(-20.0 * float(np.log10(max(r, 1e-30))), 2),
                    "max_abs": float(f"{float(np.max(np.abs(d))):.4g}")}
        if accuracy is not None:
            accuracy[args.dtype] = acc_entry(preds[0].demix(mix[:, :n_acc]).cpu().numpy())
        del stems, preds, nets
        torch.cuda.empty_cache()
        # "f32": float32 storage, contractions as split-half products on the f16 matrix pipe (TDFNet's float32 default: the 1e-4 mode as
        # shipped); "f32_exact": the same storage on v_mfma_f32_16x16x4_f32 (fmaf chains)
        for name in ("f16", "bf16", "f32", "f32_exact"):
            if name == args.dtype:
                continue
            dtp = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "f32_exact": torch.float32}[name]
            pb = args.batch if not name.startswith("f32") else min(args.batch, 13)   # fp32 activations: twice the workspace per window
            p_nets = [TDFNet(cfg, sd, ctx=ctx, dtype=dtp, max_batch=pb, contraction="exact" if name == "f32_exact" else None) for sd in sds]
            p_preds = [Predictor(pargs, net, ctx=ctx, max_batch=0) for net in p_nets]
            if accuracy is not None:
                accuracy[name] = acc_entry(p_preds[0].demix(mix[:, :n_acc]).cpu().numpy())      # also the warm-up of this type's kernels
            else:
                p_preds[0].demix(mix[:, :2 * gen - 1])
            p_steps = 1 if name == "f32_exact" else max(1, min(args.steps, 3))
            [p.demix(mix) for p in p_preds]                  # untimed: workspaces of every model allocated, every kernel of this type loaded
            fence()
            t0 = time.perf_counter()
            for _ in range(p_steps):
                out_p = [p.demix(mix) for p in p_preds]
            fence()
            dtp_s = time.perf_counter() - t0
            assert bool(torch.isfinite(out_p[0]).all())
            precision[name] = {"ms_per_step": round(dtp_s / p_steps * 1e3, 2), "value": round(N_STEMS * (n_samples / SR) * p_steps / dtp_s, 2),
                               "steps": p_steps, "windows_per_launch": pb}
            if name.startswith("f32"):
                precision[name]["contraction"] = p_nets[0].contraction
                precision[name]["realtime_factor_4stem"] = round((n_samples / SR) * p_steps / dtp_s, 1)
            del out_p, p_preds, p_nets
            torch.cuda.empty_cache()

    if rank == 0:
        audio_seconds = n_samples / SR
        value = N_STEMS * audio_seconds * args.steps / dt
        line = {
            "metric": "stems*realtime-factor (44.1kHz stereo)",
            "value": round(value, 2),
            "unit": "stems*x_realtime",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"MDX-Net UVR 4-stem (4x TFC-TDF U-Net L=11 g=48 dim_f=3072 dim_t=256 n_fft=6144), "
                                   f"{args.seconds} s 44.1 kHz stereo {'per GPU' if args.scaling == 'weak' and world > 1 else 'in total'}, margin chunker, "
                                   f"windows/launch={args.batch}",
                       "stems": N_STEMS, "audio_seconds": audio_seconds, "sharding": f"windows/{world} + all_gather"},
            "realtime_factor_4stem": round(audio_seconds * args.steps / dt, 2),
            "roofline": roofline,
            "kernels": other,
            "stages": stages,
            "precision": precision,
            "accuracy": accuracy,
            "cpu_baseline": cpu,
            "other_scaling": other_scaling,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
