#!/usr/bin/env python3
"""bench.py — train samples/s of the SDUMC two-stream self-distillation step at MOSEI feature shapes.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = main_frame_val_text_missing.py:119-150 on one batch of synthetic pre-extracted features
already resident in HBM: forward of both streams, the 6-term loss, backward, Adam.  Nothing is skipped.
Workload = BASELINE.json configs[1] ("CMU-MOSEI features, batch=64 ... 1xMI355X fp32") per GPU;
N > 1 shards the batch (weak scaling: 64 samples per GPU, global batch 64*N) with ONE RCCL all-reduce
of the flat gradient bucket per step plus the two exactness exchanges (sdumc_amd/trainer.py).

Output: ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel,
HIP events on the launch stream) and `cpu_baseline` (the CPU oracle = port of the reference's torch-eager
path, timed on this box's host cores; rank 0, N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU = 64
T_MOSEI = (375, 32, 225, 32)            # (T_audio, T_text, T_video, T_feat4)   SURVEY §8 C2
DIMS = (1024, 4096, 1024, 4096)         # WavLM-L / Vicuna-7B / MANet / Vicuna-7B
TRAIN_FLOPS_PER_SAMPLE = 1980.7e6       # algorithmic, audio/video projection counted once
PEAK_F32_MFMA_TFLOPS = 157.3            # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0          # same guide: dense bf16 matrix peak (the bf16-operand kernels are priced against it)
# fp32 GEMM kernels whose products run on the bf16 matrix pipe (variant names "..._bf16x3"): every fp32 product costs six
# v_mfma_f32_32x32x16_bf16 terms, so the matrix-pipe ceiling for fp32-equivalent FLOPs is a sixth of the dense bf16 peak
PEAK_F32_ON_BF16_PIPE_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0
F32_ARITHMETIC = ("f32 in, out and accumulation; GEMM products on the bf16 matrix pipe from operands split exactly into three bf16 "
                  "parts each (six of the nine part products, each exact in f32; error per product < 2^-23, as close to f64 as the "
                  "v_mfma_f32_32x32x2_f32 path: tools/gg_split_check.py, tests/test_gpu_split.py); side.c2_f32_mfma is the same step "
                  "with every product on the f32 MFMAs")
PEAK_HBM_GBS = 8000.0                  # same guide: HBM3E peak (6.3 TB/s achievable with a streaming copy)
# The default is the configuration the metric is quoted on (configs[1]); the others are the parity-test shapes of
# SURVEY §8, selectable for side measurements (`--workload c1|c5`), never what the driver's default run reports.
WORKLOADS = {   # name: (batch per GPU, T, dims, algorithmic train FLOPs per sample, description)
    "c2": (64, (375, 32, 225, 32), (1024, 4096, 1024, 4096), 1980.7e6,
           "BASELINE configs[1]: CMU-MOSEI-shaped features, batch=64 per GPU, fp32, both streams "
           "(text + text-missing/feat4) with self-distillation"),
    "c1": (16, (200, 16, 120, 16), (1024, 4096, 1024, 4096), 1072.7e6,
           "BASELINE configs[0] shapes (CMU-MOSI-shaped features, batch=16) on the GPU path, both streams with self-distillation"),
    "c5": (32, (512, 512, 512, 512), (1024, 1024, 1024, 1024), 4696.9e6,
           "BASELINE configs[4] per-GPU slice: synthetic long sequences T=512, d=1024, batch=32 per GPU, both streams "
           "with self-distillation"),
    "c5g": (256, (512, 512, 512, 512), (1024, 1024, 1024, 1024), 4696.9e6,
            "BASELINE configs[4] at its GLOBAL batch on every GPU: synthetic long sequences T=512, d=1024, batch=256 per GPU, "
            "both streams with self-distillation"),
}
WORKLOAD_TEXT = WORKLOADS["c2"][4]
N_RESIDENT = 4     # distinct batches the timed loop of the default line rotates through
RESIDENT_TEXT = ("; the timed loop rotates through {k} DISTINCT batches resident in HBM in the layout DeviceFeatureStore(planes=True) holds "
                 "and gathers (f32 rows for the weight gradients + three-bf16-plane rows, split once when the batch was installed, for the "
                 "frame projections; bf16 storage: bf16 rows) -- per step only input pointers change, nothing per-batch is outside the clock")
EPOCH_TEXT = ("ragged epoch at BASELINE configs[1] widths: batches of 64 utterances drawn from a 2048-utterance DeviceFeatureStore with per-sample "
              "lengths U{ceil(T/4)..T} (f32 rows + bf16 planes split once per dataset; bf16 storage: bf16 rows), every batch right-zero-padded to "
              "its own maximum like read_data.py:223-248 and assembled ON THE DEVICE by ONE gather launch that the previous step issues beside "
              "its own middle and backward (engine.FusedTrainer.run_epoch: index vectors uploaded once per epoch, two input sets, next shape's "
              "keep-bits laid out by the running step), all inside one capacity-sized arena")


def source_sha():
    """sha1 over the kernel sources the shipped libsdumc_hip.so is built from: ties a committed PMC summary to the code
    it was collected on (tools/pmc_traffic_summary.py records the same value)."""
    import glob
    import hashlib
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(ROOT, "sdumc_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "sdumc_amd", "csrc", "*.h"))
                   + glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def spawn_ranks(n, timeout_s=None):
    """`python3 bench.py --gpus N` without a launcher: start the N rank processes ourselves.  This parent makes NO GPU call
    (torch is imported, nothing under torch.cuda is touched) and never exec()s: the ranks are ordinary children, rank 0's
    stdout is relayed.  Watchdog: every child is polled; on the first non-zero exit (or when `timeout_s` runs out) the others
    are terminated -- a rank that died in its set-up would otherwise leave its peers blocked in the rendezvous or in a
    collective until torch's own timeout -- and that exit code is returned."""
    import socket
    import subprocess
    import threading
    import time
    if timeout_s is None:
        timeout_s = float(os.environ.get("SDUMC_BENCH_TIMEOUT", "1500"))
    with socket.socket() as sk:      # (the port is free now; a race with another process for it shows up as a failed rank)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = []
    drain = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)   # rank 0's pipe must not fill up
    drain.start()
    t0 = time.time()
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.time() - t0 > timeout_s:
            failed = (-1, 124)
            break
        time.sleep(0.05)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t1 = time.time()
        while any(p.poll() is None for p in procs) and time.time() - t1 < 5.0:
            time.sleep(0.05)
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p in procs:
        p.wait()
    drain.join(timeout=5.0)
    sys.stdout.write(b"".join(out0).decode(errors="replace"))
    sys.stdout.flush()
    if failed is not None:
        what = "timed out after %.0f s" % timeout_s if failed[0] < 0 else "rank %d exited with code %d" % failed
        sys.stderr.write(f"bench.py: {what}; the other ranks were terminated\n")
        return failed[1] or 1
    return 0


def synthetic_shard(B, rank, seed=1234, k=0):
    """SURVEY §8(d) synthetic inputs (features ~N(0,1), labels ~U(-3,3)); one generator per rank (and per resident batch k)."""
    g = torch.Generator().manual_seed(seed + rank + 7919 * k)
    feats = [torch.randn(B, T_MOSEI[i], DIMS[i], generator=g) for i in range(4)]
    vals = torch.rand(B, generator=g) * 6 - 3
    return feats[0], feats[1], feats[2], feats[3], vals


def init_flat_params(engine, device, seed=0):
    """Random-init weights of the reference architecture (nn.Linear defaults, xavier context vectors)."""
    import math
    lay = engine.ParamLayout.get(*DIMS[:3])
    flat = torch.zeros(lay.total)
    g = torch.Generator().manual_seed(seed)
    views = lay.views(flat)
    for name in lay.order:
        v = views[name]
        if name.endswith("attention_context_vector"):
            v.copy_(torch.randn(v.shape, generator=g) * math.sqrt(2.0 / (v.shape[0] + v.shape[1])))
        elif name == "prelu.weight":
            v.fill_(0.25)
        elif name == "layer_normali.weight":
            v.fill_(1.0)
        elif name == "layer_normali.bias":
            v.zero_()
        else:
            fan_in = v.shape[1] if v.dim() == 2 else views[name[:-5] + ".weight"].shape[1]
            v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) / math.sqrt(fan_in))
    return flat.to(device), lay


def roofline_leg(_lib, launch, steps, traffic_ok=True):
    """Eager (un-captured) replays of the same step with a HIP event pair around every GEMM launch,
    on the launch stream.  Returns the dominant GEMM variant's achieved TFLOP/s."""
    lib = _lib.lib
    torch.cuda.synchronize()
    try:
        lib.sdumc_set_concurrency(0)      # one lane: a kernel's event-bracketed duration is then its own
        lib.sdumc_profile_enable(1)
        for _ in range(steps):
            launch()
        torch.cuda.synchronize()
        arr = (_lib.ProfEntry * 32)()
        n = lib.sdumc_profile_report(arr, 32)
    finally:                              # (process-wide switches: never leave them flipped behind an exception)
        lib.sdumc_profile_enable(0)
        lib.sdumc_set_concurrency(1)
    rows = []
    for i in range(max(n, 0)):
        e = arr[i]
        if e.launches:
            rows.append({"kernel": e.name.decode(), "launches_per_step": e.launches / steps,
                         "avg_us": 1e3 * e.total_ms / e.launches,
                         "gflop_per_launch": e.total_flops / e.launches / 1e9,
                         "tflops": e.total_flops / (e.total_ms * 1e-3) / 1e12,
                         "ms_per_step": e.total_ms / steps})
    rows.sort(key=lambda r: -r["ms_per_step"])
    top = rows[0]
    # the committed PMC summary was collected on the default workload in fp32: it describes no other configuration
    traffic, traffic_src = recorded_traffic(top["kernel"]) if traffic_ok else (None, None)
    bf16_kernel = top["kernel"].startswith("gemm_bf16") or top["kernel"].endswith("_bf16")   # bf16-operand / bf16-storage kernels
    split_kernel = top["kernel"].endswith("_bf16x3")                                          # f32 operands, six bf16 MFMA terms per product
    peak = PEAK_BF16_MFMA_TFLOPS if bf16_kernel else round(PEAK_F32_ON_BF16_PIPE_TFLOPS, 1) if split_kernel else PEAK_F32_MFMA_TFLOPS
    # the profiler's clock for the same kernel, when profiles/rocprof_kernel_avg.json (tools/kstats_summary.py over a rocprofv3
    # --kernel-trace --stats run of this command with --serial-lanes) was collected on these kernel sources: the two clocks differ by
    # box and by the profiler's own overhead, so the line carries both and the fraction each gives
    rp = recorded_rocprof(top["kernel"])
    rocprof = {}
    if rp is not None:
        rp_tf = top["gflop_per_launch"] / (rp["avg_us"] * 1e-6) / 1e3
        rocprof = {"rocprof_avg_us": rp["avg_us"], "rocprof_calls": rp["calls"], "rocprof_tflops": round(rp_tf, 2),
                   "frac_rocprof": round(rp_tf / peak, 4), "rocprof_source": rp["source"]}
    return {"bound": "mfma", "achieved": round(top["tflops"], 2), "peak": peak, "unit": "TFLOP/s", **rocprof,
            "peak_note": ("dense bf16 MFMA peak / 6: this kernel computes every f32 product as six bf16 MFMA terms (f32-equivalent FLOPs counted)"
                          if split_kernel else "dense bf16 MFMA peak" if bf16_kernel else "dense f32 MFMA peak"),
            "frac_of_f32_mfma_peak": round(top["tflops"] / PEAK_F32_MFMA_TFLOPS, 4),
            "frac": round(top["tflops"] / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
            "kernel": top["kernel"], "avg_launch_us": round(top["avg_us"], 2),
            "gflop_per_launch": round(top["gflop_per_launch"], 3),
            "launches_per_step": top["launches_per_step"],
            "gemm_ms_per_step": round(sum(r["ms_per_step"] for r in rows), 4),
            "all_gemm_variants": [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()} for r in rows]}


def resident_step(engine, flat, batches, bf16=False, seed=2024, planes=True, bits_next=True):
    """One TrainStep over an arena with len(batches) input sets, each holding one installed batch (f32 / bf16 rows, and P3 planes in
    fp32 storage): `run()` points the step at the next set and launches it -- the loop a training epoch over resident features runs."""
    K = len(batches)
    arena = engine.StepArena(flat, B_PER_GPU, T_MOSEI, DIMS, bf16=bf16, sets=K, planes=planes, bits_next=bits_next)
    ts = engine.TrainStep(flat, B_PER_GPU, T_MOSEI, DIMS, seed=seed, bf16=bf16, arena=arena)
    for k in range(K):
        ts.use_set(k)
        ts.set_batch(*batches[k])
    count = [0]

    def run():
        ts.use_set(count[0] % K)
        count[0] += 1
        ts.launch()
        return ts.losses
    return ts, run


def timed(run, steps, warmup):
    for _ in range(warmup):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def hbm_roofline_bf16(ms_per_step, mfma):
    """bf16 storage: at bf16 MFMA rates (16x fp32) the step is bound by HBM traffic and by the latency of its launch chain, not by
    the matrix cores.  achieved = HBM-side bytes per step (profiles/pmc_traffic_bf16.json: separate rocprofv3 FETCH_SIZE /
    WRITE_SIZE passes over this command, fetch x2 gfx950 correction) / the measured step time; the MFMA figure of the dominant
    GEMM stays as a side key.  Bytes are reported only when the summary was collected on these kernel sources."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic_bf16.json")
    traffic, src = None, None
    try:
        with open(path) as f:
            doc = json.load(f)
        if doc.get("source_sha") == source_sha():
            traffic = int(doc["per_step"]["fetch_bytes"] + doc["per_step"]["write_bytes"])
            src = "profiles/pmc_traffic_bf16.json (whole step, rocprofv3 PMC passes on these sources)"
        else:
            src = f"profiles/pmc_traffic_bf16.json is stale (collected on sources {doc.get('source_sha')}, this build is {source_sha()})"
    except (OSError, ValueError, KeyError):
        pass
    ach = traffic / (ms_per_step * 1e-3) / 1e9 if traffic else None
    return {"bound": "hbm", "achieved": round(ach, 1) if ach else None, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(ach / PEAK_HBM_GBS, 4) if ach else None, "traffic": traffic, "traffic_source": src,
            "kernel": "whole step (HBM-side bytes of every launch)", "mfma": mfma}


def c3_bf16_side_leg(engine, flat0, batches, args):
    """BASELINE configs[2] (MOSEI shapes, B = 64, text-missing stream + self-distillation, bf16) under the same clock as the
    headline: the bf16-storage step over the same rotating resident batches, same --steps / --warmup, timed the same way, right after
    the headline's timed region.  A side block of the one JSON line; the headline keys stay those of configs[1]."""
    step, run = resident_step(engine, flat0.clone(), batches, bf16=True)
    dt = timed(run, args.steps, args.warmup)
    losses = step.losses.cpu()
    if not torch.isfinite(losses).all():
        raise SystemExit(f"non-finite loss in the bf16-storage side leg: {losses.tolist()}")
    ms = 1e3 * dt / args.steps
    roof = hbm_roofline_bf16(ms, None)
    roof.pop("mfma", None)
    return {"workload": "BASELINE configs[2]: the headline's rotating batches in bf16 storage of features / frames / keys / frame-level gradients, "
                        "f32 accumulation, softmax, utterance-level layers, losses and Adam",
            "value": round(B_PER_GPU * args.steps / dt, 2), "unit": "samples/s", "ms_per_step": round(ms, 4),
            "steps": args.steps, "warmup": args.warmup, "final_loss": round(float(losses[0]), 5), "roofline": roof}


def set_batch_loop_side_leg(engine, flat0, batches, args):
    """The loop of INTEGRATION.md section 2 -- `for data in loader: step.set_batch(...); step.run()` -- with set_batch INSIDE the clock:
    every step copies a fresh batch (device to device here: the loader's tensors are already on the GPU) into the step's buffers; planes
    off (the default of TrainStep: a split for one use costs more than it saves)."""
    step = engine.TrainStep(flat0.clone(), B_PER_GPU, T_MOSEI, DIMS, seed=2024)
    K = len(batches)
    count = [0]

    def run():
        step.set_batch(*batches[count[0] % K])
        count[0] += 1
        step.run()
    dt = timed(run, args.steps, args.warmup)
    losses = step.losses.cpu()
    if not torch.isfinite(losses).all():
        raise SystemExit(f"non-finite loss in the set_batch-loop side leg: {losses.tolist()}")
    # the same loop with the zero-copy hand-over: the step reads the loader's device tensors where they are
    step2 = engine.TrainStep(flat0.clone(), B_PER_GPU, T_MOSEI, DIMS, seed=2024)
    count[0] = 0

    def run2():
        step2.use_batch(*batches[count[0] % K])
        count[0] += 1
        step2.run()
    dt2 = timed(run2, args.steps, args.warmup)
    losses2 = step2.losses.cpu()
    return {"workload": "the headline's batches through TrainStep.set_batch + run per step (a 224 MB device copy per step inside the clock, "
                        "no planes: frame projections split in-kernel)",
            "value": round(B_PER_GPU * args.steps / dt, 2), "unit": "samples/s", "ms_per_step": round(1e3 * dt / args.steps, 4),
            "steps": args.steps, "warmup": args.warmup, "final_loss": round(float(losses[0]), 5),
            "use_batch": {"workload": "the same loop through TrainStep.use_batch (the step reads the loader's device tensors in place: no copy)",
                          "value": round(B_PER_GPU * args.steps / dt2, 2), "ms_per_step": round(1e3 * dt2 / args.steps, 4),
                          "loss_equal_to_set_batch": bool(torch.equal(losses, losses2))}}


def c2_f32_mfma_side_leg(engine, _lib, flat0, batches, args, headline_losses):
    """The headline step with every GEMM product on v_mfma_f32_32x32x2_f32 (sdumc_set_split_(0)): what the default's products on
    the bf16 matrix pipe buy, under the same clock.  It starts from a clone of the INITIAL parameters and runs the same number of
    steps over the same rotating batches as the headline, so `loss_abs_diff_vs_headline` is a driver-run cross-check of the split
    arithmetic against the f32 MFMAs."""
    lib = _lib.lib
    prev = split_mask_default()
    try:
        lib.sdumc_set_split_(0)
        step, run = resident_step(engine, flat0.clone(), batches, planes=False)
        dt = timed(run, args.steps, args.warmup)
        losses = step.losses.cpu()
    finally:
        lib.sdumc_set_split_(prev)
    if not torch.isfinite(losses).all():
        raise SystemExit(f"non-finite loss in the f32-MFMA side leg: {losses.tolist()}")
    return {"workload": "the headline step with sdumc_set_split_(0): every GEMM product on v_mfma_f32_32x32x2_f32",
            "value": round(B_PER_GPU * args.steps / dt, 2), "unit": "samples/s", "ms_per_step": round(1e3 * dt / args.steps, 4),
            "steps": args.steps, "warmup": args.warmup, "final_loss": round(float(losses[0]), 5),
            "loss_abs_diff_vs_headline": round(abs(float(losses[0]) - float(headline_losses[0])), 7),
            "loss_terms_max_abs_diff_vs_headline": round(float((losses[:7] - headline_losses[:7]).abs().max()), 7)}


def split_mask_default():
    """The process default of the split switch (environment SDUMC_SPLIT, else every kernel family)."""
    try:
        return int(os.environ.get("SDUMC_SPLIT", "15"))
    except ValueError:
        return 15


def prewarm_leg(engine, flat0, batch, bf16, min_s):
    """Wall-clock pre-warm: a THROWAWAY step object runs the same step until `min_s` seconds have passed, so that the timed
    steps are not read while the chip is still ramping its clock (~0.1 s from idle: profiles/README.md, round 3).  The headline
    step object starts from the initial parameters and step counter afterwards."""
    if min_s <= 0:
        return 0.0, 0
    st = engine.TrainStep(flat0.clone(), B_PER_GPU, T_MOSEI, DIMS, seed=2024, bf16=bf16, planes=True)
    st.set_batch(*batch)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < min_s:
        for _ in range(8):
            st.run()
        n += 8
        torch.cuda.synchronize()
    del st
    torch.cuda.synchronize()
    return time.perf_counter() - t0, n


def recorded_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed PMC summary (profiles/pmc_traffic.json, written by
    tools/pmc_traffic_summary.py from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over this same
    command; counters cannot be read from inside the process).  None when the kernel is not in the summary."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            doc = json.load(f)
        k = doc["kernels"].get(kernel)
    except (OSError, ValueError, KeyError):
        return None, None
    if not k:
        return None, None
    if doc.get("source_sha") != source_sha():     # collected on other kernel sources: stale, not reported
        return None, f"profiles/pmc_traffic.json is stale (collected on sources {doc.get('source_sha')}, this build is {source_sha()})"
    return k["traffic_bytes"], "profiles/pmc_traffic.json (rocprofv3 PMC passes on these sources, fetch x2 gfx950 correction, average per launch)"


def recorded_rocprof(kernel):
    """Average launch duration of `kernel` from the committed rocprofv3 summary (profiles/rocprof_kernel_avg.json); None when absent
    or collected on other kernel sources."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "rocprof_kernel_avg.json")
    try:
        with open(path) as f:
            doc = json.load(f)
        if doc.get("source_sha") != source_sha():
            return None
        k = doc["kernels"].get(kernel)
        if not k:
            return None
        return {"avg_us": k["avg_us"], "calls": k["calls"],
                "source": f"profiles/rocprof_kernel_avg.json <- {doc.get('from')} (rocprofv3 --kernel-trace --stats, serial lanes, these sources)"}
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline_leg(steps=3):
    """The CPU oracle (torch-eager port of the reference op sequence, native dropout RNG exactly like the
    reference) on this box's host cores: the same B=64 MOSEI-shaped two-stream train step."""
    from oracle import sdumc_oracle as O
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    P = O.init_params(DIMS, seed=0)
    batch = O.synthetic_batch(B_PER_GPU, T_MOSEI, DIMS, seed=1234)
    state = {}
    # torch eager does not scale to all hardware threads on this op mix (measured on the 2x64-core GPU
    # host: 8-16 threads ~88 samples/s, 64 threads 49, 256 threads 0.3): probe a few thread counts,
    # bounded to ~30 s in total, and report the best one together with the thread count it used.
    budget_end = time.perf_counter() + 30.0
    best, best_threads, step_idx, tried = float("inf"), 0, 0, []
    for n in sorted({t for t in (8, 16, 32, 64) if t <= cores} or {cores}):
        if time.perf_counter() > budget_end:
            break
        torch.set_num_threads(n)
        O.train_step(P, state, *batch, mode="native", step=step_idx)        # warm-up at this thread count
        step_idx += 1
        mine = float("inf")
        for _ in range(steps):
            t0 = time.perf_counter()
            O.train_step(P, state, *batch, mode="native", step=step_idx)
            step_idx += 1
            mine = min(mine, time.perf_counter() - t0)
            if time.perf_counter() > budget_end:
                break
        tried.append((n, round(B_PER_GPU / mine, 1)))
        if mine < best:
            best, best_threads = mine, n
    return {"value": round(B_PER_GPU / best, 2), "unit": "samples/s", "cores": best_threads, "kind": "port",
            "sample": f"best of <= {steps} train steps per thread count of the same B={B_PER_GPU} MOSEI-shaped batch, "
                      f"torch {torch.__version__} eager fp32; (threads, samples/s) tried: {tried}; host has {cores} hw threads",
            "sec_per_step": round(best, 4)}


def epoch_leg(args, engine, flat, lay, dev, bf16=None, nb=None, warmup=None):
    """One epoch of REAL loader behaviour -- every batch has its own padded shape and is assembled from the resident store inside the
    clock -- against the static-shape step on one resident batch at the same mean padded frame counts."""
    from sdumc_amd.data import DeviceFeatureStore
    from sdumc_amd.engine import bf16_mode
    bf16 = args.bf16 if bf16 is None else bf16
    nb, B = max(args.steps if nb is None else nb, 1), B_PER_GPU
    warmup = args.warmup if warmup is None else warmup
    hf = bf16_mode(bf16, DIMS) == 2
    store = DeviceFeatureStore.synthetic(2048, T_MOSEI, DIMS, seed=1234, device=dev, bf16=hf, planes=not hf)
    g = torch.Generator().manual_seed(7)
    batches = [torch.randperm(len(store), generator=g)[:B] for _ in range(nb + warmup)]
    tr = engine.FusedTrainer(flat, DIMS, capacity=(B, T_MOSEI), seed=2024, bf16=bf16)
    plan_w, plan_t = store.plan_epoch(batches[:warmup]) if warmup else None, store.plan_epoch(batches[warmup:])
    shapes = [sh[1] for sh in (plan_w.shapes if plan_w else [])] + [sh[1] for sh in plan_t.shapes]
    if plan_w:
        # warm-up by the WALL clock as well (the chip ramps its clock for ~0.1 s after an idle spell, and building the store and the
        # plans is one: ten warm-up batches are 13 ms; a cold start showed as a fixed ~50 ms on the timed epoch, 1.8 instead of 1.29 ms
        # per step over 100 batches)
        t_w = time.perf_counter()
        tr.run_epoch(store, plan_w)
        torch.cuda.synchronize()
        while time.perf_counter() - t_w < args.prewarm_s:
            tr.run_epoch(store, plan_w)
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.run_epoch(store, plan_t)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    losses = tr.state.losses.cpu()
    if not torch.isfinite(losses).all():
        raise SystemExit(f"non-finite loss: {losses.tolist()}")
    # the static-shape step at the epoch's mean padded shape (rounded up), on one resident batch (planes split once)
    tshapes = shapes[warmup:]
    mean_T = tuple(int(-(-sum(s[i] for s in tshapes) // len(tshapes))) for i in range(4))
    flat2 = flat.clone()
    st = engine.TrainStep(flat2, B, mean_T, DIMS, seed=2024, bf16=bf16, planes=True)
    gg = torch.Generator(device=dev).manual_seed(3)
    st.set_batch(*[torch.randn(B, mean_T[i], DIMS[i], device=dev, generator=gg) for i in range(4)],
                 torch.rand(B, device=dev, generator=gg) * 6 - 3)
    ds = timed(st.run, nb, warmup)
    ev, sv = B * nb / dt, B * nb / ds
    return {"metric": "train samples/sec over a ragged epoch (side measurement)", "value": round(ev, 2), "unit": "samples/s",
            "n_gpus": 1, "steps": nb, "warmup": warmup, "ms_per_step": round(1e3 * dt / nb, 4),
            "host_enqueue_ms_per_step": round(1e3 * t_enq / nb, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16 storage, f32 accumulation" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": EPOCH_TEXT, "batch_per_gpu": B, "distinct_batch_shapes": len(set(shapes)),
                       "mean_padded_T": list(mean_T), "capacity_T": list(T_MOSEI), "feature_dims": list(DIMS),
                       "store_utterances": len(store), "store_gb": round(store.nbytes / 1e9, 2), "cached_steps": len(tr._steps)},
            "static_shape_at_mean_T": {"value": round(sv, 2), "ms_per_step": round(1e3 * ds / nb, 4)},
            "epoch_over_static": round(ev / sv, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as one hipGraph instead of launching eagerly (measured slower: graph replay "
                         "serialises the engine's three lanes, eager launches overlap them)")
    ap.add_argument("--no-graph", action="store_true", help=argparse.SUPPRESS)   # old spelling of the default
    ap.add_argument("--serial-lanes", action="store_true",
                    help="keep every kernel on one stream (for rocprofv3 --kernel-trace: per-kernel durations)")
    ap.add_argument("--bf16", action="store_true",
                    help="BASELINE configs[2] arithmetic: bf16 operands (fp32 accumulate) in the frame-level projections, "
                         "forward and backward; NOT the default workload (configs[1] is fp32)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + ["epoch"], default="c2",
                    help="c2 = BASELINE configs[1], the configuration the metric is quoted on (default); c1 / c5 / c5g = side measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--prewarm-s", type=float, default=0.3,
                    help="seconds of untimed steps on a throwaway step object before --warmup (clock ramp; 0 = off)")
    ap.add_argument("--no-bits-next", action="store_true", help="A/B: generate every step's keep-bits at its head (no sdumc_net_io.bits_next)")
    ap.add_argument("--no-side", action="store_true", help="skip the side legs of the default line (profiling runs)")
    ap.add_argument("--resident", type=int, default=N_RESIDENT, help="distinct resident batches the timed loop rotates through (single GPU)")
    ap.add_argument("--epoch-batches", type=int, default=200, help="batches of the ragged-epoch side legs")
    ap.add_argument("--stream-priority", type=int, default=None, help="experiment: run the timed loop on a torch stream of this priority (-1 = high)")
    args = ap.parse_args()
    global B_PER_GPU, T_MOSEI, DIMS, TRAIN_FLOPS_PER_SAMPLE, WORKLOAD_TEXT
    epoch = args.workload == "epoch"
    B_PER_GPU, T_MOSEI, DIMS, TRAIN_FLOPS_PER_SAMPLE, WORKLOAD_TEXT = WORKLOADS["c2" if epoch else args.workload]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))      # no launcher: this process becomes the parent of N rank processes
    if world != args.gpus:
        args.gpus = world
    if world > 1 and os.environ.get("SDUMC_TEST_FAIL_RANK") == str(rank):   # test hook (tests/test_gpu_dp.py): this rank dies in set-up
        raise SystemExit(3)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: sdumc_amd has no CPU fallback")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:      # debugging aid only (two ranks on one GPU with SDUMC_DIST_BACKEND=gloo)
        local_rank %= ndev
    if world > ndev:            # several processes share a GPU: the clustered utterance-level kernels assume they own it
        os.environ.setdefault("SDUMC_CHAIN_CLUSTER", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    from sdumc_amd import _lib, engine
    # SDUMC_FORCE_DP=1 at world size 1: a one-rank RCCL communicator and the data-parallel step with every collective
    # issued -- exercises the N > 1 code path (communicator stream ordering, async early-slice all-reduce) on one GPU.
    force_dp = world == 1 and os.environ.get("SDUMC_FORCE_DP", "0") == "1"
    if force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SDUMC_DIST_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if args.serial_lanes:
        _lib.lib.sdumc_set_concurrency(0)
    flat, lay = init_flat_params(engine, dev)
    if epoch:
        if world != 1:
            raise SystemExit("--workload epoch is a single-GPU side measurement")
        print(json.dumps(epoch_leg(args, engine, flat, lay, dev)), flush=True)
        return
    single = world == 1 and not force_dp
    rotate = single and not args.graph
    nres = max(1, args.resident) if rotate else 1
    batches = [[t.to(dev) for t in synthetic_shard(B_PER_GPU, rank, k=k)] for k in range(nres)]
    batch = batches[0]
    flat0 = flat.clone()          # the initial parameters: what the side legs start from (the step updates `flat` in place)
    prewarm_s, prewarm_steps = prewarm_leg(engine, flat0, batch, args.bf16, args.prewarm_s)

    if rotate:
        # K distinct batches installed up front in the store's layout; the loop below only switches input pointers between steps
        step, run = resident_step(engine, flat, batches, bf16=args.bf16, bits_next=not args.no_bits_next)
    elif single:
        step = engine.TrainStep(flat, B_PER_GPU, T_MOSEI, DIMS, seed=2024, bf16=args.bf16, bits_next=not args.no_bits_next, planes=True)
        step.set_batch(*batch)
        step.capture()
        run = step.run
    else:
        from sdumc_amd.trainer import DataParallelStep
        step = DataParallelStep(flat, B_PER_GPU, T_MOSEI, DIMS, seed=2024, exact=True, bf16=args.bf16,
                                force_collectives=force_dp, planes=True)      # (one resident shard per rank)
        step.set_batch(*batch)
        run = step.step

    import contextlib
    prio_ctx = contextlib.nullcontext()
    if args.stream_priority is not None:      # (experiment: the caller's stream in the priority class of the engine's lanes)
        prio_stream = torch.cuda.Stream(priority=args.stream_priority)
        prio_stream.wait_stream(torch.cuda.current_stream())
        prio_ctx = torch.cuda.stream(prio_stream)
    prio_ctx.__enter__()
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    if world > 1 or force_dp:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    t_enq = time.perf_counter() - t0          # host time to ENQUEUE the K steps (the GPU is still running): host-bound if ~= dt
    torch.cuda.synchronize()
    prio_ctx.__exit__(None, None, None)
    if world > 1 or force_dp:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dp_extra = None
    if world > 1 or force_dp:
        mine = torch.tensor([dt], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [float(t.item()) for t in every]
        dt = max(per_rank)                                   # MAX over ranks, as the contract says
        # the gradient all-reduce on its own (the flat 15.4 MB bucket, same call the step makes), outside the timed region
        bucket = step.be.grads
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(bucket)
        torch.cuda.synchronize()
        ar_ms = 1e3 * (time.perf_counter() - t1) / 20
        bucket.zero_()
        dp_extra = {"per_rank_ms_per_step": [round(1e3 * t / args.steps, 4) for t in per_rank],
                    "allreduce_ms": round(ar_ms, 4), "allreduce_bytes": bucket.numel() * 4,
                    "backend": dist.get_backend(), "overlap": bool(step.overlap)}

    losses = (step.losses if (world == 1 and not force_dp) else step.be.losses).cpu()
    if not torch.isfinite(losses).all():
        raise SystemExit(f"non-finite loss: {losses.tolist()}")

    value = world * B_PER_GPU * args.steps / dt
    out = {
        "metric": "train samples/sec at MOSEI feature shapes (two-stream forward + 6-term loss + backward + Adam)",
        "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "prewarm_s": round(prewarm_s, 3), "prewarm_steps": prewarm_steps,
        "ms_per_step": round(1e3 * dt / args.steps, 4), "host_enqueue_ms_per_step": round(1e3 * t_enq / args.steps, 4),
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16 storage of features / frames / keys / frame-level gradients with f32 accumulation; f32 softmax, utterance-level layers, losses and Adam (f32 master weights)" if args.bf16 else "f32", "data": "synthetic",
        **({} if args.bf16 else {"arithmetic": F32_ARITHMETIC}),
        "config": {"workload": ("bf16 storage (--bf16, the dtype of BASELINE configs[2]/[4]) on: " if args.bf16 else "") + WORKLOAD_TEXT
                               + (RESIDENT_TEXT.format(k=nres) if rotate else "; one resident batch per rank (f32 rows + bf16 planes split when it was installed), replayed"),
                   "resident_batches": nres,
                   "batch_per_gpu": B_PER_GPU, "global_batch": world * B_PER_GPU,
                   "T_audio_text_video_feat4": list(T_MOSEI), "feature_dims": list(DIMS),
                   "parallelism": f"dp{world}" if world > 1 else ("dp1 (one-rank RCCL communicator, all collectives issued)" if force_dp else "single"),
                   "launch": "hipGraph replay" if (args.graph and world == 1) else "eager, 4 lanes (caller stream + 2 high-priority modality side streams + 1 side stream for the grouped weight-gradient launches, keep-bits and the forward Cross_Attention key GEMMs)",
                   "params": sum(int(torch.Size(shape).numel()) for _, shape, _ in lay.entries.values()),
                   "param_buffer_floats": lay.total, "final_loss": round(float(losses[0]), 5)},
        "whole_step_tflops": round(value * TRAIN_FLOPS_PER_SAMPLE / 1e12, 2),
        "whole_step_frac_of_f32_mfma_peak": round(value * TRAIN_FLOPS_PER_SAMPLE / 1e12 / (world * PEAK_F32_MFMA_TFLOPS), 4),
    }
    if dp_extra is not None:
        out["data_parallel"] = dp_extra
    if rotate and args.workload == "c2" and not args.bf16 and not args.serial_lanes and not args.no_side:
        out["side"] = {"c3_bf16": c3_bf16_side_leg(engine, flat0, batches, args),
                       "c2_f32_mfma": c2_f32_mfma_side_leg(engine, _lib, flat0, batches, args, losses),
                       "set_batch_loop": set_batch_loop_side_leg(engine, flat0, batches, args)}
        # the ragged epoch (every batch assembled from the store inside the clock) in both storage modes, 200 batches each
        for name, hf in (("epoch", False), ("epoch_bf16", True)):
            e = epoch_leg(args, engine, flat0.clone(), lay, dev, bf16=hf, nb=args.epoch_batches, warmup=10)
            out["side"][name] = {k: e[k] for k in ("value", "unit", "ms_per_step", "host_enqueue_ms_per_step", "steps", "warmup", "dtype",
                                                   "static_shape_at_mean_T", "epoch_over_static")}
            out["side"][name]["workload"] = e["config"]["workload"]
            out["side"][name]["config"] = {k: e["config"][k] for k in ("distinct_batch_shapes", "mean_padded_T", "store_utterances", "store_gb")}
    if not args.no_roofline:     # every rank runs it (the DP step has collectives); rank 0 reports
        roof = roofline_leg(_lib, run if single else step.step, max(3, min(10, args.steps)),
                            traffic_ok=(args.workload == "c2" and not args.bf16))
        if rank == 0:
            out["roofline"] = hbm_roofline_bf16(out["ms_per_step"], roof) if (args.bf16 and args.workload == "c2" and world == 1) else roof
    if rank == 0 and world == 1:
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_leg()
            out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
    if world > 1 or force_dp:
        dist.destroy_process_group()
    # RCCL writes a version banner through C stdio, which reaches a redirected stdout only when its buffer is flushed
    # (normally at exit, i.e. AFTER anything Python printed): flush it first so that the JSON line is the last line.
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
