#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X CSI-NN2 backend.

    python bench.py --gpus N --steps K --warmup W          (no launcher: N > 1 spawns N ranks itself, rank r on GPU r)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (ranks from the launcher)

Workload (BASELINE.json configs[1]): MobileNetV1 int8 NHWC, batch 1 per
GPU -- the 28 convolution layers of example/c906_mobilenetv1_f16.c (conv1, 13 x (depthwise 3x3 +
pointwise 1x1) with fused ReLU, 1x1 classifier), synthetic seeded tensors in the exact-arithmetic
quantisation regime, activations resident in HBM.  A "step" is one pass of all 28 layers over one
image, dispatched through the C operator API (csinn_conv2d_relu on CSINN_MI355X) and replayed as
one hipGraph so that the timed region contains no host work.  Batch-1 inference does not shard:
with N GPUs every rank runs its own replica on its own image ("replicas only", weak scaling); the
only communication is the one-time RCCL broadcast of rank 0's packed weights.

Timing: W untimed warm-up steps, then WINDOWS (default 7) windows of exactly K steps each, every window
bracketed by barrier + synchronisation on both sides and reduced with MAX over ranks; `ms_per_step` /
`value` are the MEDIAN window (one 20-step window of a 75 us step is 1.5 ms: a single sample of that
length is at the mercy of one scheduler hiccup), all windows are listed in `windows_ms`.

The JSON line also carries
  roofline     -- the dominant kernel function of the step (largest share of GPU time), its
                  algorithmic bytes per launch over its average launch duration measured with HIP
                  events on the launch stream, against the 8 TB/s HBM peak (MobileNetV1 at batch 1
                  is HBM/latency bound: 78.7 op/B vs a ridge of ~630 op/B);
  cpu_baseline -- the reference C backend (kind "reference": the genuine library built by
                  oracle/Makefile.ref, or kind "port": this repo's oracle) timed on the host cores
                  on a bounded sample of the same network;
  configs      -- the other configurations of BASELINE.json, each timed as ONE captured pass of its layers (median of
                  windows of replays) with its own roofline.  N = 1: configs[2] ResNet-50 3x3 set int8 batch 128 in
                  NCHW (as BASELINE names it) and in NHWC against the int8 MFMA peak (`kernel_only` = the dominant
                  kernel's launches alone, `relayout_us` = what layers that still go through a re-layout pass cost),
                  configs[3] MobileNetV1 binary16 NCHW batch 1 (c906_mobilenetv1_f16 shapes), and MobileNetV1 int8 at
                  batch 128 (throughput view of configs[1]).  N > 1: configs[4] -- the ResNet-50 3x3 set with a total
                  batch of 128 N SHARDED over the ranks (128 images per GPU), rank 0's packed weights broadcast once
                  with RCCL behind the C-ABI.  --no-configs skips them; --extra adds serving views.

--workload resnet50_3x3 --total-batch 1024 runs configs[4] as the headline instead.
Process group: torch.distributed is bootstrap only (gloo: the 128-byte ncclUniqueId, agreement flags, barriers and
the MAX over ranks of the window times, all on CPU tensors); the weights move through RCCL behind the C-ABI
(shl_mi355x_comm_* / shl_mi355x_bcast_const_blocks).  Device memory comes from the C-ABI allocator, not from torch.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
I8_MFMA_PEAK_TOPS = 5000.0  # dense int8 MFMA, 2x the bf16 2.5 PF (MI355X_MICROARCH.md)
F16_MFMA_PEAK_TFLOPS = 2500.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="mobilenetv1", choices=["mobilenetv1", "resnet50_3x3"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: 1 mobilenetv1, 128 resnet50_3x3)")
    ap.add_argument("--dtype", default="int8", choices=["int8", "f16"])
    ap.add_argument("--layout", default="", choices=["", "NHWC", "NCHW"])
    ap.add_argument("--total-batch", type=int, default=0,
                    help="total batch sharded over the ranks (BASELINE configs[4]: resnet50_3x3 --total-batch 1024)")
    ap.add_argument("--windows", type=int, default=7, help="timed windows of --steps steps each (median reported)")
    ap.add_argument("--no-configs", action="store_true", help="headline only: skip the configs[2] / configs[3] entries")
    ap.add_argument("--extra", action="store_true", help="serving views: concurrent streams, csinn_session_run end to end")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fuse", action="store_true", help="every layer its own launch (no pointwise+depthwise fusion)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--detail", action="store_true", help="print the per-layer table to stderr")
    ap.add_argument("--steps-only", action="store_true",
                    help="run warmup + timed steps and stop (for rocprofv3 --pmc passes: only step launches are seen)")
    ap.add_argument("--traffic-file", default="",
                    help="JSON from tools/pmc_traffic.py (default profiles/traffic_<workload>_<dtype>_<layout>_b<batch>.json)")
    args = ap.parse_args()
    # with a profiler attached every pass is launched kernel by kernel instead of as a graph replay
    # (workloads.PROFILER_ATTACHED: rocprofv3 --kernel-trace of ROCm 7.2 crashes in hipGraphLaunch)
    args.profiler = any(k.startswith(("ROCPROF_", "ROCP_TOOL")) for k in os.environ)
    return args


# ------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline_worker(args):
    """Runs in its own process (the genuine library and this repo's front-end export the same
    csinn_* symbols).  Times whole MobileNetV1 images layer by layer through csinn_conv2d on
    CSINN_REF (layer mode, NHWC, int8) until ~cpu-seconds have elapsed."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    pkg = cases.pkg
    wl = importlib.import_module("csi-nn2_amd.workloads")
    layers = wl.MOBILENETV1
    kind = "reference" if cases.have_reference() else "port"
    cores = 1
    if kind == "reference":
        fe = cases.load_reference_frontend()
    else:
        cores = cases.oracle_lib().oracle_num_threads()
    budget = args.cpu_seconds
    t_used, ops_done, layers_done = 0.0, 0, 0
    images = 0
    done = False
    while not done:
        for i, L in enumerate(layers):
            case = cases.make_case(50 + i, n=1, h=L["h"], w=L["w"], c=L["cin"], co=L["cout"], k=(L["k"], L["k"]),
                                   stride=(L["stride"],) * 2, pad=(L["pad"],) * 4, depthwise=L["depthwise"],
                                   act=1 if L["act"] else 0)
            t0 = time.perf_counter()
            if kind == "reference":
                cases.csinn_run(fe, pkg.API_REF, case)
            else:
                cases.oracle_run(case, "ref")
            t_used += time.perf_counter() - t0
            ops_done += wl.layer_ops(L)
            layers_done += 1
            if t_used > budget and layers_done >= len(layers):
                done = True
                break
        images += 1
        if t_used > budget:
            done = True
    extra = []
    if kind == "reference":
        # configs[2]'s own reference path: NCHW -> shl_ref_conv2d_nchw_f32 -> conv_im2col_sgemm_avx (conv_avx.h, its
        # pragmas say omp num_threads(8)); ONE image per call (the x86 path computes image 0 of a batch only)
        t2, ops2, calls2 = 0.0, 0, 0
        seen = set()
        for L in wl.RESNET50_3X3:
            key = (L["cin"], L["h"], L["stride"])
            if key in seen:
                continue
            seen.add(key)
            case = cases.make_case(70 + calls2, layout=cases.NCHW, n=1, h=L["h"], w=L["w"], c=L["cin"], co=L["cout"],
                                   k=(3, 3), stride=(L["stride"],) * 2, pad=(1,) * 4, act=1)
            cases.csinn_run(fe, pkg.API_REF, case)  # warm (thread team, page faults)
            reps = 0
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.7:
                cases.csinn_run(fe, pkg.API_REF, case)
                reps += 1
            t2 += (time.perf_counter() - t0) / reps
            ops2 += wl.layer_ops(L)
            calls2 += 1
        extra.append({"baseline_config": "configs[2]", "value": ops2 / t2 / 1e9, "unit": "GOPS", "cores": 8, "kind": "reference",
                      "sample": "the 7 distinct ResNet-50 3x3 shapes, int8 NCHW, one image per csinn_conv2d call on CSINN_REF "
                                "(im2col + 8x8 AVX2 sgemm, omp num_threads(8) in conv_avx.h; tensor set-up of the call included), %.1f ms per image of the seven" % (t2 * 1e3)})
        # configs[3]: binary16 NCHW, the first layers of MobileNetV1 (c906 example shapes)
        t3, ops3, calls3 = 0.0, 0, 0
        for i, L in enumerate(wl.MOBILENETV1[:7]):
            case = cases.make_case(90 + i, dtype="f16", layout=cases.NCHW, n=1, h=L["h"], w=L["w"], c=L["cin"], co=L["cout"],
                                   k=(L["k"], L["k"]), stride=(L["stride"],) * 2, pad=(L["pad"],) * 4, depthwise=L["depthwise"],
                                   act=1 if L["act"] else 0)
            reps = 0
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.5:
                cases.csinn_run(fe, pkg.API_REF, case)
                reps += 1
            t3 += (time.perf_counter() - t0) / reps
            ops3 += wl.layer_ops(L)
            calls3 += 1
        extra.append({"baseline_config": "configs[3]", "value": ops3 / t3 / 1e9, "unit": "GFLOPS", "cores": 8, "kind": "reference",
                      "sample": "the first %d layers of MobileNetV1 in binary16 NCHW via csinn_conv2d / csinn_depthwise_conv2d on CSINN_REF "
                                "(dense layers on the AVX sgemm with 8 threads, depthwise serial), %.1f ms for the %d layers" % (calls3, t3 * 1e3, calls3)})
    print(json.dumps({"value": ops_done / t_used / 1e9, "unit": "GOPS", "cores": cores, "kind": kind, "configs": extra,
                      "imgs_per_sec": (ops_done / sum(wl.layer_ops(l) for l in layers)) / t_used,
                      "sample_short": "%.2f MobileNetV1 int8 images via csinn_conv2d on CSINN_REF, %.0f s" % (
                          ops_done / sum(wl.layer_ops(l) for l in layers), t_used),
                      "sample": "%d MobileNetV1 int8 NHWC layer calls (%.2f images) via csinn_conv2d on %s, %.1f s"
                                % (layers_done, ops_done / sum(wl.layer_ops(l) for l in layers),
                                   "CSINN_REF of the genuine library" if kind == "reference" else "the oracle port",
                                   t_used)}))


def run_cpu_baseline(args):
    try:
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker",
                              "--cpu-seconds", str(args.cpu_seconds)], capture_output=True, text=True, timeout=600)
        line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:  # the baseline is reported, never required
        return {"value": None, "unit": "GOPS", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}


# ------------------------------------------------------------------------------------ GPU side
class CHbm:
    """Device memory from the C-ABI (shl_mi355x_malloc / _upload): raw pointers go into csinn_tensor.data."""

    def __init__(self, hip):
        self.hip, self.live = hip, []

    def alloc(self, nbytes):
        p = self.hip.shl_mi355x_malloc(max(int(nbytes), 16))
        if not p:
            raise RuntimeError("shl_mi355x_malloc(%d): %s" % (nbytes, self.hip.shl_mi355x_last_error().decode()))
        self.live.append(p)
        return p

    def upload(self, ptr, host):
        a = np.ascontiguousarray(host)
        if self.hip.shl_mi355x_upload(ptr, a.ctypes.data, a.nbytes, None) != 0 or self.hip.shl_mi355x_stream_sync(None) != 0:
            raise RuntimeError("upload: " + self.hip.shl_mi355x_last_error().decode())

    def free_all(self):
        self.hip.shl_mi355x_stream_sync(None)
        for p in self.live:
            self.hip.shl_mi355x_free(p)
        self.live = []


def timed_windows(replay, sync, steps, warmup, windows, dist=None, torch=None):
    """W warm-up steps, then `windows` windows of exactly `steps` steps, each bracketed by barrier + synchronisation on
    both sides and reduced with MAX over ranks (CPU tensors: the process group is bootstrap only)."""
    for _ in range(warmup):
        replay()
    sync()
    out = []
    for _ in range(max(1, windows)):
        sync()
        if dist:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            replay()
        sync()
        w = time.perf_counter() - t0
        if dist:
            dist.barrier()
            t = torch.tensor([w], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            w = float(t.item())
        out.append(w)
    return out


PROFILER = any(k.startswith(("ROCPROF_", "ROCP_TOOL")) for k in os.environ)


def time_groups(chain, hip, opt, stream, reps=20):
    """Average duration of every launch of one pass (a layer, or a fused pointwise + depthwise pair),
    measured with HIP events recorded on the launch stream around `reps` back-to-back launches of it
    (captured in a graph: no host gaps)."""
    out = []
    ev0, ev1 = hip.shl_mi355x_event_create(), hip.shl_mi355x_event_create()
    ms = C.c_float()
    for u in range(len(chain.units)):
        opt.shl_mi355x_set_stream(stream)
        if PROFILER:  # no graphs under rocprofv3 (workloads.PROFILER_ATTACHED)
            best = []
            for _ in range(2):
                hip.shl_mi355x_event_record(ev0, stream)
                for _ in range(reps):
                    chain.run_unit(u)
                hip.shl_mi355x_event_record(ev1, stream)
                hip.shl_mi355x_event_elapsed_ms(ev0, ev1, C.byref(ms))
                best.append(ms.value / reps)
            out.append(min(best) * 1e-3)
            continue
        hip.shl_mi355x_graph_begin(stream)
        for _ in range(reps):
            chain.run_unit(u)
        g = hip.shl_mi355x_graph_end(stream)
        hip.shl_mi355x_graph_launch(g, stream)  # warm
        hip.shl_mi355x_stream_sync(stream)
        best = []
        for _ in range(3):
            hip.shl_mi355x_event_record(ev0, stream)
            hip.shl_mi355x_graph_launch(g, stream)
            hip.shl_mi355x_event_record(ev1, stream)
            hip.shl_mi355x_event_elapsed_ms(ev0, ev1, C.byref(ms))
            best.append(ms.value / reps)
        hip.shl_mi355x_graph_destroy(g)
        out.append(float(np.median(best)) * 1e-3)  # seconds per launch
    hip.shl_mi355x_event_destroy(ev0)
    hip.shl_mi355x_event_destroy(ev1)
    return out


def summarise_kernels(chain, wl, per_layer_s, bound_hint):
    groups = {}
    for u, t in enumerate(per_layer_s):
        g = groups.setdefault(chain.unit_kernel_name(u), dict(time=0.0, bytes=0, ops=0, launches=0))
        g["time"] += t
        g["bytes"] += chain.unit_bytes(u)
        g["ops"] += chain.unit_ops(u)
        g["launches"] += 1
    name, g = max(groups.items(), key=lambda kv: kv[1]["time"])
    if bound_hint == "hbm":
        achieved, peak, unit = g["bytes"] / g["time"] / 1e9, HBM_PEAK_GBS, "GB/s"
    else:
        achieved = g["ops"] / g["time"] / 1e12
        peak = I8_MFMA_PEAK_TOPS if chain.dtype == "int8" else F16_MFMA_PEAK_TFLOPS
        unit = "TOP/s" if chain.dtype == "int8" else "TFLOP/s"
    roof = {"bound": bound_hint, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak,
            "traffic": None, "kernel": name, "launches_per_step": g["launches"],
            "avg_launch_us": g["time"] / g["launches"] * 1e6,
            "algorithmic_bytes_per_launch": g["bytes"] / g["launches"],
            "ops_per_launch": g["ops"] / g["launches"],
            "share_of_step_time": g["time"] / sum(per_layer_s)}
    return roof, groups


def attach_traffic(roof, path):
    """roofline.traffic = HBM bytes per launch of the dominant kernel from the committed PMC passes
    (tools/pmc_traffic.py over two `rocprofv3 --pmc` runs of `bench.py --steps-only`); null when the
    file is absent.  Counters cannot be collected from inside the timed process."""
    try:
        with open(path) as f:
            rec = json.load(f).get(roof["kernel"])
    except (OSError, ValueError):
        rec = None
    if rec:
        roof["traffic"] = rec["hbm_bytes_per_launch"]
        roof["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, %s" % os.path.relpath(
            path, os.path.dirname(os.path.abspath(__file__)))


def measure_config(tag, name, layers_x, batch_x, dtype_x, layout_x, bound_x, chained_x, env, steps=20, warmup=3, windows=5,
                   dist=None, seed=4321, bcast_from=None, total_batch=None):
    """One configuration of BASELINE.json as ONE captured pass of its layers: median of `windows` windows of `steps`
    replays, plus the per-launch table (HIP events around 20 back-to-back launches of every layer) behind the roofline
    of its dominant kernel.  With a process group: every rank runs its shard, rank 0's weights are broadcast first."""
    torch, fe, hip, opt, wl, par, stream, detail = (env[k] for k in ("torch", "fe", "hip", "opt", "wl", "par", "stream", "detail"))
    hbm = CHbm(hip)
    # chained layers run as csinn_session_setup would launch them: the library says which neighbours share a launch
    # (pointwise+depthwise in latency form at small batches, depthwise+pointwise in bandwidth form at throughput batches)
    fuse = chained_x and ((dtype_x == "int8" and layout_x == "NHWC") or (dtype_x == "f16" and layout_x == "NCHW"))
    rc = wl.LayerChain(fe, hip, opt, layers_x, batch_x, hbm.alloc, hbm.upload, dtype=dtype_x, layout=layout_x, seed=seed,
                       chained=chained_x, fuse=fuse)
    how = None
    if dist is not None:
        how = par.broadcast_weights(rc, torch, dist, hip, opt, env["rank"], env["world"], src=0, prefer_c=bcast_from)
        par.assert_replicas_agree(rc, torch, dist, hip)
    rc.capture(stream)
    wins = timed_windows(rc.replay, lambda: hip.shl_mi355x_stream_sync(stream), steps, warmup, windows, dist, torch)
    t_pass = float(np.median(wins)) / steps
    world = env["world"] if dist is not None else 1
    images = total_batch if total_batch else batch_x * world
    ops_total = rc.total_ops() // batch_x * images
    unit = "GOPS" if dtype_x == "int8" else "GFLOPS"
    peak = I8_MFMA_PEAK_TOPS if dtype_x == "int8" else F16_MFMA_PEAK_TFLOPS
    entry = {"baseline_config": tag, "workload": name, "metric": "conv_" + unit.lower(), "value": ops_total / t_pass / 1e9,
             "unit": unit, "images_per_sec": images / t_pass, "ms_per_pass": t_pass * 1e3, "n_gpus": world,
             "windows_ms": [w / steps * 1e3 for w in wins],
             "timing": "one captured pass of the %d launches, median of %d windows of %d replays%s" % (
                 len(rc.units), len(wins), steps, ", barrier + MAX over ranks" if dist is not None else ""),
             "dtype": "i8" if dtype_x == "int8" else "f16", "layers": len(layers_x), "launches": len(rc.units),
             "mfma_frac_whole_set": ops_total / world / t_pass / 1e12 / peak}
    if how:
        entry["parallelism"] = "batch shard x%d, %d images per GPU (weights broadcast once: %s)" % (world, batch_x, how)
    if not chained_x and dist is None and dtype_x == "int8" and not PROFILER:
        # what the launch boundaries of the pass cost (informational; `ms_per_pass` above is the figure of merit): the set's
        # layers do not depend on each other, so the even and the odd ones can run as two captured graphs on two streams --
        # a kernel's ramp and drain then overlap the neighbour's.  A network whose layers DO depend on each other has no such
        # second graph (tools/dev/two_stream.py --interleave; profiles/r06_notes.md)
        try:
            hb2 = CHbm(hip)
            pairs = []
            for k, sub in enumerate((layers_x[0::2], layers_x[1::2])):
                ck = wl.LayerChain(fe, hip, opt, sub, batch_x, hb2.alloc, hb2.upload, dtype=dtype_x, layout=layout_x, seed=seed + 50 * (k + 1),
                                   chained=False, fuse=False)
                sk = hip.shl_mi355x_stream_create()
                ck.capture(sk)
                pairs.append((ck, sk))

            def replay2():
                for ck, _ in pairs:
                    ck.replay()

            def sync2():
                for _, sk in pairs:
                    hip.shl_mi355x_stream_sync(sk)
            w2 = timed_windows(replay2, sync2, steps, warmup, windows)
            entry["two_streams"] = {"ms_per_pass": float(np.median(w2)) / steps * 1e3,
                                    "note": "even / odd layers of the set as two graphs on two streams (the layers are independent): "
                                            "launch ramp / drain / boundary overlapped; not the figure of merit"}
            for ck, sk in pairs:
                ck.release()
                hip.shl_mi355x_stream_destroy(sk)
            hb2.free_all()
        except Exception as e:  # informational only
            entry["two_streams"] = {"error": repr(e)}
    if chained_x and dist is None and dtype_x == "int8" and batch_x >= 64 and batch_x % 2 == 0 and not PROFILER:
        # the same question for a DEPENDENT network at a throughput batch: the batch as two half-batch chains on two streams (each
        # chain's launches fill the other's boundaries; tools/dev/two_stream.py).  Informational: csinn_session_run captures ONE chain
        try:
            hb2 = CHbm(hip)
            pairs = []
            for k in range(2):
                ck = wl.LayerChain(fe, hip, opt, layers_x, batch_x // 2, hb2.alloc, hb2.upload, dtype=dtype_x, layout=layout_x, seed=seed + k,
                                   chained=True, fuse=fuse)
                sk = hip.shl_mi355x_stream_create()
                ck.capture(sk)
                pairs.append((ck, sk))

            def replay2():
                for ck, _ in pairs:
                    ck.replay()

            def sync2():
                for _, sk in pairs:
                    hip.shl_mi355x_stream_sync(sk)
            w2 = timed_windows(replay2, sync2, steps, warmup, windows)
            entry["two_streams"] = {"ms_per_pass": float(np.median(w2)) / steps * 1e3,
                                    "note": "the batch as two half-batch chains on two streams (per %d images); not the figure of merit" % batch_x}
            for ck, sk in pairs:
                ck.release()
                hip.shl_mi355x_stream_destroy(sk)
            hb2.free_all()
        except Exception as e:
            entry["two_streams"] = {"error": repr(e)}
    if env["rank"] == 0:
        rt = time_groups(rc, hip, opt, stream, reps=20)
        rroof, rgroups = summarise_kernels(rc, wl, rt, bound_x)
        wname = "resnet50_3x3" if layers_x is wl.RESNET50_3X3 else "mobilenetv1"
        attach_traffic(rroof, os.path.join(ROOT, "profiles", "traffic_%s_%s_%s_b%d.json" % (wname, dtype_x, layout_x, batch_x)))
        entry["roofline"] = rroof
        # the dominant kernel alone, and what the layers that still go through a re-layout pass cost (NCHW: the
        # stride-2 layers; their launch = [C][HW] -> [HW][C] copy + NHWC kernel with an NCHW-writing epilogue)
        entry["kernel_only"] = {"kernel": rroof["kernel"], "launches": rroof["launches_per_step"], "us_per_launch": rroof["avg_launch_us"],
                                "achieved": rroof["achieved"], "unit": rroof["unit"]}
        if layout_x == "NCHW" and not chained_x:
            # (the int8 row-patch kernel is NCHW-native under its one name; the binary16 one reports "patch_nchw_f16" when it is)
            native = (lambda k: "patch" in k or "nchw" in k) if dtype_x == "int8" else (lambda k: "nchw" in k)
            rel = [(rc.unit_name(u), t) for u, t in enumerate(rt) if not native(rc.unit_kernel_name(u))]
            entry["relayout_layers"] = {"count": len(rel), "us_total_with_relayout": sum(t for _, t in rel) * 1e6,
                                        "note": "launches of layers the NCHW-native kernels do not take (re-layout pass + NHWC kernel)"}
        entry["sum_launch_us"] = sum(rt) * 1e6
        entry["kernels"] = {k: {"launches": v["launches"], "us_total": v["time"] * 1e6, "TOPs": v["ops"] / v["time"] / 1e12,
                                "GBps": v["bytes"] / v["time"] / 1e9} for k, v in rgroups.items()}
        if detail:
            sys.stderr.write("---- %s\n" % name)
            for u, t in enumerate(rt):
                sys.stderr.write("%-52s %-32s %8.2f us %8.1f GB/s %8.2f TOP/s\n" % (
                    rc.unit_name(u), rc.unit_kernel_name(u), t * 1e6, rc.unit_bytes(u) / t / 1e9, rc.unit_ops(u) / t / 1e12))
    rc.release()
    hbm.free_all()
    return entry


def _r(x, nd=5):
    """numbers of the stdout line: five significant digits"""
    if isinstance(x, float):
        return float("%.*g" % (nd, x))
    return x


def compact_line(result):
    """The ONE stdout line, <= ~1.9 KB so that the driver's stored tail (2 000 characters) holds all of it: the contract's
    headline fields, `roofline`, `cpu_baseline`, and per configuration {baseline_config, ms_per_pass, value, dtype,
    roofline: {kernel, frac, bound, traffic}}.  Everything else -- per-kernel tables, windows, timing notes -- is the
    FULL record: stderr (one line, prefix "BENCH_FULL ") and bench_full.json next to this file (SHL_BENCH_FULL overrides)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    out = {k: _r(result.get(k)) for k in keep}
    cfg = result.get("config", {})
    out["config"] = {k: cfg.get(k) for k in ("workload_short", "parallelism_short", "rccl_nranks", "distinct_devices", "transport",
                                              "ops_per_image", "algorithmic_bytes_per_image", "algorithmic_bytes_per_image_survey")
                     if cfg.get(k) is not None}
    if out.get("n_gpus") == 1:
        out["config"].pop("distinct_devices", None)
    out["config"]["workload"] = out["config"].pop("workload_short", str(cfg.get("workload"))[:96])
    out["config"]["parallelism"] = out["config"].pop("parallelism_short", cfg.get("parallelism"))
    if result.get("rccl_error"):
        out["rccl_error"] = str(result["rccl_error"])[:200]
    roof = result.get("roofline")
    if roof:
        out["roofline"] = {k: _r(roof.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel")}
    cb = result.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": _r(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                               "sample": str(cb.get("sample_short") or cb.get("sample"))[:64]}
        for e in cb.get("configs", []) or []:
            out["cpu_baseline"][e["baseline_config"]] = "%.4g %s, %d cores" % (e["value"], e["unit"], e["cores"])
    if result.get("mfma_rate"):  # measured TOP/s of the two int8 matrix instructions
        out["mfma_rate"] = {k.replace("_TOPs", ""): _r(v, 4) for k, v in result["mfma_rate"].items() if k.endswith("_TOPs")}
    sd = result.get("session_device_io")
    if sd:
        out["session_ms_per_image"] = _r(sd["ms_per_image"])
    cl = []
    for e in result.get("configs", []) or []:
        if e["baseline_config"] == "configs[2] (binary16 NHWC view)":
            continue  # (a view nobody asked for: in the full record only)
        if "error" in e:
            cl.append({"baseline_config": e["baseline_config"], "error": str(e["error"])[:80]})
            continue
        r = e.get("roofline", {})
        # (value: GOPS for dtype i8, GFLOPS for f16; kernel names without the family prefix)
        c = {"baseline_config": _short_tag(e["baseline_config"]), "ms_per_pass": _r(e["ms_per_pass"]), "value": _r(e["value"], 4),
             "dtype": e["dtype"],
             "roofline": {"kernel": str(r.get("kernel")).replace("conv_igemm_", "").replace("_mfma32x32x32", "").replace("_mfma32x32x16", ""), "frac": _r(r.get("frac"), 4), "bound": r.get("bound"),
                          "traffic": _r(r.get("traffic"), 4)}}
        if e.get("n_gpus", 1) > 1:
            c["n_gpus"] = e["n_gpus"]
        if "ms_per_pass" in e.get("two_streams", {}):  # (informational: the set's independent layers on two streams)
            c["two_streams_ms"] = _r(e["two_streams"]["ms_per_pass"], 4)
        cl.append(c)
    if cl:
        out["configs"] = cl
    return out


def _short_tag(tag):
    return tag.replace(" (NHWC view)", " NHWC").replace(" (binary16 NHWC view)", " f16 NHWC").replace(" (binary16 NCHW view)", " f16 NCHW").replace(
        " (throughput view)", " b128")


def emit(result):
    """the ONE JSON line, after everything native code may have left in C stdio buffers (librccl prints a version banner
    to stdout when it is first used) and as the last thing written.  The full record goes to stderr and to a file first."""
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    full = json.dumps(result)
    sys.stderr.write("BENCH_FULL " + full + "\n")
    sys.stderr.flush()
    try:
        with open(os.environ.get("SHL_BENCH_FULL", os.path.join(ROOT, "bench_full.json")), "w") as f:
            f.write(full + "\n")
    except OSError:
        pass
    sys.stdout.flush()
    print(json.dumps(compact_line(result), separators=(",", ":")), flush=True)


def launch_plan(gpus, environ):
    """What `bench.py --gpus N` does about ranks.  "run": this process IS a rank (N = 1, or an external launcher --
    torch.distributed.run -- already set RANK / WORLD_SIZE); ("spawn", N): no launcher in sight, so this process becomes
    one: N children, one rank per GPU.  A WORLD_SIZE that contradicts --gpus is an error, not something to guess about."""
    ws = environ.get("WORLD_SIZE")
    if ws is not None and "RANK" in environ:
        if int(ws) != gpus:
            raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (gpus, ws))
        return ("run", int(ws))
    if gpus <= 1:
        return ("run", 1)
    return ("spawn", gpus)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N copies of this command line with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (what torch.distributed.run would export), rank r on GPU r.  Rank 0
    inherits stdout (its JSON line stays the last line written); the other ranks' stdout goes to stderr.  Any rank
    failing fails the run: its siblings are terminated by PID and the exit code is the first non-zero one."""
    import socket
    # (the port is free when it is picked, not reserved: should another process take it before rank 0 binds it, the children's
    # rendezvous fails, every rank exits non-zero and so does this process -- loudly; SO_REUSEADDR keeps TIME_WAIT out of the way)
    with socket.socket() as s:
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    t_kill = 0.0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                sys.stderr.write("bench.py: rank %d exited with %d; stopping the other ranks\n" % (r, code))
                for q in pending:
                    procs[q].terminate()
                t_kill = time.time() + 20.0  # a rank stuck inside a HIP / RCCL call ignores SIGTERM: SIGKILL after a grace period
        if rc != 0 and pending and time.time() > t_kill:
            for q in pending:
                procs[q].kill()
        time.sleep(0.05)
    sys.exit(rc)


def main():
    args = parse_args()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args)
    mode, nranks = launch_plan(args.gpus, os.environ)
    if mode == "spawn":
        return spawn_ranks(nranks)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook: run every rank on ONE device (a 1-GPU box can then exercise the N > 1 code path: weight broadcast
    # through the fallback transport, replica agreement, barriers, max-over-ranks); never set by the driver
    single_dev = os.environ.get("SHL_BENCH_SINGLE_DEVICE") == "1"
    if single_dev:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = torch = None
    if world > 1:
        # bootstrap only (gloo, CPU tensors): the ncclUniqueId, agreement flags, barriers, MAX of the window times.
        # The weights travel through RCCL behind the C-ABI (csrc/comm_rccl.hip), not through this group.
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")

    pkg = importlib.import_module("csi-nn2_amd")
    wl = importlib.import_module("csi-nn2_amd.workloads")
    par = importlib.import_module("csi-nn2_amd.sharding")
    fe = pkg.load_frontend("standalone")
    hip, opt = pkg.load_backend(fe)
    if hip.shl_mi355x_device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: " + hip.shl_mi355x_last_error().decode())
    # (a launcher that narrows every rank to its own GPU -- HIP_VISIBLE_DEVICES per rank -- leaves one visible device: index 0;
    # the distinct-devices check below compares PCI bus ids, so two ranks on one GPU are still caught)
    local_rank %= hip.shl_mi355x_device_count()
    pkg.check(hip.shl_mi355x_set_device(local_rank), hip, "set_device")
    if torch is not None and torch.cuda.is_available():
        torch.cuda.set_device(local_rank)  # the fallback transport of the weight broadcast stages through torch buffers
    arch = C.create_string_buffer(64)
    cus = C.c_int32()
    hip.shl_mi355x_device_info(arch, 64, C.byref(cus), None)

    hbm = CHbm(hip)
    if args.workload == "mobilenetv1":
        layers, batch, chained, bound = wl.MOBILENETV1, args.batch or 1, True, "hbm"
        layout = args.layout or "NHWC"
    else:
        layers, batch, chained, bound = wl.RESNET50_3X3, args.batch or 128, False, "mfma"
        layout = args.layout or "NHWC"
    sharded = args.total_batch > 0
    if sharded:  # BASELINE configs[4]: one batch cut into contiguous per-rank slices (no data-path collective)
        lo, hi = par.shard_batch(args.total_batch, world, rank)
        batch = hi - lo
        if batch == 0:
            raise SystemExit("rank %d owns no image of a total batch of %d" % (rank, args.total_batch))
    # the graph-level rewrite csinn_session_setup applies on this backend (session.c plan_fusion):
    # pointwise + the depthwise layer that consumes it = one launch
    fuse = chained and not args.no_fuse and ((args.dtype == "int8" and layout == "NHWC") or (args.dtype == "f16" and layout == "NCHW"))
    # rank 0 owns the real weights; other ranks build their plans from a different seed and must
    # receive rank 0's packed blocks over RCCL before they can agree with it
    chain = wl.LayerChain(fe, hip, opt, layers, batch, hbm.alloc, hbm.upload, dtype=args.dtype, layout=layout,
                          seed=1234 if rank == 0 else 999 + rank, chained=chained, fuse=fuse)
    bcast = None
    if world > 1:
        def on_stall(msg):
            # the one place a multi-GPU run can wait for ever (a peer that never joins the communicator): say so and stop
            if rank == 0:
                emit({"metric": "mobilenetv1_int8_images_per_sec" if args.workload == "mobilenetv1" else "resnet50_3x3_int8_conv_gops",
                      "value": None, "unit": "img/s" if args.workload == "mobilenetv1" else "GOPS", "n_gpus": world, "steps": args.steps,
                      "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                      "dtype": "i8" if args.dtype == "int8" else "f16", "data": "synthetic", "rccl_error": msg,
                      "config": {"workload": args.workload, "parallelism": "x%d, never got past the weight broadcast" % world}})
            else:
                sys.stderr.write("bench.py rank %d: %s\n" % (rank, msg))
            os._exit(5)
        par.STALL_HANDLER = on_stall
        bcast = par.broadcast_weights(chain, torch, dist, hip, opt, rank, world, src=0, prefer_c=not single_dev)
        par.assert_replicas_agree(chain, torch, dist, hip)
    elif sharded:
        bcast = par.broadcast_weights_one_rank(chain, hip, opt)  # the same C entry points with a communicator of one

    if sharded and os.environ.get("SHL_BENCH_SHARD_CHECK") and rank == world - 1:
        # test hook (tests/test_sharding.py): the LAST rank runs layer 0 of its shard on a seeded input with the weights it
        # RECEIVED and leaves input and output behind, so that the test can hold them against the oracle computed from rank
        # 0's weights -- agreement between ranks alone would not notice a broadcast that corrupts every rank alike
        e0 = chain.entries[0]
        rng = np.random.default_rng(4242 + rank)
        n_in, n_out = int(np.prod(e0["in_dims"])), int(np.prod(e0["out_dims"]))
        x = rng.integers(-64, 64, n_in, dtype=np.int8)
        hip.shl_mi355x_upload(e0["d_in"], x.ctypes.data, x.nbytes, None)
        hip.shl_mi355x_stream_sync(None)
        opt.shl_mi355x_set_stream(None)
        assert e0["run"](*e0["args"]) == 1
        hip.shl_mi355x_stream_sync(None)
        y = np.empty(n_out, dtype=np.int8)
        hip.shl_mi355x_download(y.ctypes.data, e0["d_out"], y.nbytes, None)
        hip.shl_mi355x_stream_sync(None)
        np.savez(os.environ["SHL_BENCH_SHARD_CHECK"], x=x.reshape(e0["in_dims"]), y=y.reshape(e0["out_dims"]), rank=rank,
                 lo=lo, hi=hi, layout=layout)
    rccl_error = None
    if world > 1 and not single_dev:
        # N ranks must mean N devices and an RCCL communicator of N: anything else is not the run that was asked for.
        # The run still finishes on the fallback transport (the weights are the same bytes either way) and prints its line
        # WITH `rccl_error` and the transport named -- a flagged number beats an empty record -- and exits non-zero.
        got_dev, got_nr = par.LAST_BROADCAST.get("distinct_devices"), par.LAST_BROADCAST.get("rccl_nranks")
        if got_dev != world or got_nr != world:
            rccl_error = "%s [--gpus %d: %s distinct devices, RCCL communicator of %s ranks]" % (
                par.LAST_BROADCAST.get("rccl_error") or bcast, world, got_dev, got_nr)
            sys.stderr.write("bench.py: " + rccl_error + "\n")

    stream = hip.shl_mi355x_stream_create()
    chain.capture(stream)
    windows = timed_windows(chain.replay, lambda: hip.shl_mi355x_stream_sync(stream), args.steps, args.warmup, args.windows, dist, torch)
    elapsed = float(np.median(windows))

    total_images = (args.total_batch if sharded else world * batch) * args.steps
    ops_per_image = chain.total_ops() // batch
    ops_total = ops_per_image * total_images
    result = {
        "metric": "mobilenetv1_int8_images_per_sec" if args.workload == "mobilenetv1" else "resnet50_3x3_int8_conv_gops",
        "value": total_images / elapsed if args.workload == "mobilenetv1" else ops_total / elapsed / 1e9,
        "unit": "img/s" if args.workload == "mobilenetv1" else "GOPS",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if sharded else "weak",
        "vs_baseline": None, "dtype": "i8" if args.dtype == "int8" else "f16", "data": "synthetic",
        "conv_gops": ops_total / elapsed / 1e9,
        "images_per_sec": total_images / elapsed,
        "windows_ms": [w * 1e3 for w in windows],
        "profiler_attached": bool(args.profiler),
        "timing": "median of %d windows of %d steps, each bracketed by barrier + synchronise, MAX over ranks" % (
            len(windows), args.steps),
        "config": {"workload": "%s %s %s, %d conv layers in %d launches%s, batch %d per GPU%s, hipGraph replay via csinn_* C API"
                               % (args.workload, args.dtype, layout, len(layers), len(chain.units),
                                  " (pointwise+depthwise pairs fused as csinn_session_setup does)" if len(chain.units) < len(layers) else "",
                                  batch, " (total batch %d sharded)" % args.total_batch if sharded else ""),
                   "workload_short": "%s %s %s batch %d/GPU%s, %d conv layers in %d launches, hipGraph replay via csinn_*" % (
                       args.workload, args.dtype, layout, batch, " (total %d sharded)" % args.total_batch if sharded else "",
                       len(layers), len(chain.units)),
                   "parallelism_short": ("batch shard x%d" % world if sharded else "replicas x%d" % world),
                   "transport": par.LAST_BROADCAST.get("transport"),
                   # SURVEY.md 8(d)'s figure: every layer reads its input and writes its output (14.445 MB per image for
                   # MobileNetV1 int8); `algorithmic_bytes_per_image` beside it leaves out the intermediates of fused pairs
                   "algorithmic_bytes_per_image_survey": sum(wl.layer_bytes(L, 1, 1 if args.dtype == "int8" else 2) for L in layers),
                   "per_gpu_batch": batch,
                   "parallelism": ("batch shard x%d" % world if sharded else "replicas x%d" % world) +
                                  (" (weights broadcast once: %s)" % bcast if bcast else ""),
                   "process_group": "none" if world == 1 else "gloo (bootstrap only: RCCL id, flags, barriers, MAX of window times)",
                   "device_memory": "C-ABI allocator (shl_mi355x_malloc)",
                   "ops_per_image": ops_per_image, "algorithmic_bytes_per_image": chain.total_bytes() // batch,
                   "device": arch.value.decode(), "compute_units": cus.value,
                   "kernel_choice": ("selection rules only (SHL_MI355X_TUNE=0)" if os.environ.get("SHL_MI355X_TUNE", "")[:1] == "0"
                                     else "forced family " + os.environ["SHL_MI355X_IGEMM"] if os.environ.get("SHL_MI355X_IGEMM")
                                     else "measured per plan at creation (conv_plan.hip:tune_plan: rules vs every implicit-GEMM family, cached per shape)"),
                   # what proves N distinct devices took part: every rank's PCI bus id (hipDeviceGetPCIBusId through the
                   # C-ABI, gathered over the bootstrap group) and the rank count RCCL itself reports for the communicator
                   "device_bus_ids": par.LAST_BROADCAST.get("bus_ids") or [par.device_bus_id(hip)],
                   "distinct_devices": par.LAST_BROADCAST.get("distinct_devices", 1),
                   "rccl_nranks": par.LAST_BROADCAST.get("rccl_nranks")},
    }

    if rccl_error:
        result["rccl_error"] = rccl_error

    def finish():
        chain.release()
        if dist:
            dist.barrier()
            dist.destroy_process_group()
        if rccl_error:
            sys.exit(3)

    if args.steps_only:
        if rank == 0:
            emit(result)
        return finish()
    env = dict(torch=torch, fe=fe, hip=hip, opt=opt, wl=wl, par=par, stream=stream, detail=args.detail, rank=rank, world=world)
    if rank == 0:
        per_layer = time_groups(chain, hip, opt, stream)
        roof, groups = summarise_kernels(chain, wl, per_layer, bound)
        roof["mfma_frac_info"] = (chain.total_ops() / sum(per_layer) / 1e12) / (
            I8_MFMA_PEAK_TOPS if args.dtype == "int8" else F16_MFMA_PEAK_TFLOPS)
        attach_traffic(roof, args.traffic_file or os.path.join(
            ROOT, "profiles", "traffic_%s_%s_%s_b%d.json" % (args.workload, args.dtype, layout, batch)))
        result["roofline"] = roof
        result["kernels"] = {k: {"launches": v["launches"], "us_total": v["time"] * 1e6,
                                 "GBps": v["bytes"] / v["time"] / 1e9, "TOPs": v["ops"] / v["time"] / 1e12}
                             for k, v in groups.items()}
        result["sum_layer_us"] = sum(per_layer) * 1e6
        if args.detail:
            for u, t in enumerate(per_layer):
                sys.stderr.write("%-52s %-32s %8.2f us %8.1f GB/s %8.2f TOP/s\n" % (
                    chain.unit_name(u), chain.unit_kernel_name(u), t * 1e6,
                    chain.unit_bytes(u) / t / 1e9, chain.unit_ops(u) / t / 1e12))
    if rank == 0 and not args.no_configs:
        # the two int8 matrix instructions, measured on this device: v_mfma_i32_32x32x32_i8 (what the kernels issue) and
        # v_mfma_i32_32x32x16_i8 (the form BASELINE.json's north_star names: half the K per instruction at the same pass
        # count, i.e. half the rate -- kept as the named baseline); `frac` stays against the 5 POP/s specification peak
        rates = {}
        for key, form in (("i32_32x32x32_i8_TOPs", 0), ("i32_32x32x16_i8_TOPs", 1), ("f32_32x32x16_f16_TFLOPs", 2)):
            tops, ns = C.c_double(), C.c_double()
            if hip.shl_mi355x_debug_mfma_rate(form, 2, C.byref(tops), C.byref(ns)) == 0:
                rates[key] = tops.value
        result["mfma_rate"] = rates
    if args.workload == "mobilenetv1" and not args.no_configs and not sharded:
        result["configs"] = []
        if world == 1:
            # the other single-GPU configurations of BASELINE.json
            for tag, name, layers_x, batch_x, dtype_x, layout_x, bound_x, chained_x, steps_x in (
                    ("configs[2]", "resnet50 3x3 set int8 NCHW batch 128", wl.RESNET50_3X3, 128, "int8", "NCHW", "mfma", False, 20),
                    ("configs[2] (NHWC view)", "resnet50 3x3 set int8 NHWC batch 128", wl.RESNET50_3X3, 128, "int8", "NHWC", "mfma", False, 20),
                    ("configs[2] (binary16 NHWC view)", "resnet50 3x3 set binary16 NHWC batch 128", wl.RESNET50_3X3, 128, "f16", "NHWC", "mfma", False, 10),
                    ("configs[2] (binary16 NCHW view)", "resnet50 3x3 set binary16 NCHW batch 128", wl.RESNET50_3X3, 128, "f16", "NCHW", "mfma", False, 10),
                    ("configs[3]", "mobilenetv1 fp16 NCHW batch 1 (c906_mobilenetv1_f16 shapes)", wl.MOBILENETV1, 1, "f16", "NCHW", "hbm", True, 50),
                    ("configs[1] (throughput view)", "mobilenetv1 int8 NHWC batch 128, launches as csinn_session_setup fuses them (depthwise+pointwise pairs of the 32 / 64 / 128 / 256-channel blocks in one launch each)", wl.MOBILENETV1, 128,
                     "int8", "NHWC", "hbm", True, 20)):
                try:
                    result["configs"].append(measure_config(tag, name, layers_x, batch_x, dtype_x, layout_x, bound_x, chained_x, env, steps=steps_x,
                                                            windows=5))
                except Exception as e:  # an entry that fails must not take the headline with it
                    result["configs"].append({"baseline_config": tag, "workload": name, "error": repr(e)})
        else:
            # BASELINE configs[4] at this N: the ResNet-50 3x3 set (NCHW, as BASELINE names configs[2]) with a total batch
            # of 128 N sharded over the ranks, rank 0's packed weights broadcast once (RCCL behind the C-ABI)
            try:
                e4 = measure_config("configs[4]", "resnet50 3x3 set int8 NCHW, total batch %d sharded over %d GPUs" % (128 * world, world),
                                    wl.RESNET50_3X3, 128, "int8", "NCHW", "mfma", False, env, steps=20, dist=dist,
                                    seed=4321 if rank == 0 else 777 + rank, bcast_from=not single_dev, total_batch=128 * world)
                e4["scaling"] = "weak (128 images per GPU; batch 1024 at N = 8)"
                result["configs"].append(e4)
            except Exception as e:
                result["configs"].append({"baseline_config": "configs[4]", "error": repr(e)})
    if rank != 0:
        return finish()
    if args.extra and args.workload == "mobilenetv1":
        # serving view: independent batch-1 requests in flight on several streams of ONE GPU (the
        # headline `value` is strictly one request at a time; a single chain leaves most CUs idle)
        nstreams = 4
        chains, streams = [], []
        for k in range(nstreams):
            hk = CHbm(hip)
            ck = wl.LayerChain(fe, hip, opt, layers, batch, hk.alloc, hk.upload, dtype=args.dtype, layout=layout,
                               seed=1234, chained=chained, fuse=fuse)
            sk = hip.shl_mi355x_stream_create()
            ck.capture(sk)
            chains.append((ck, hk))
            streams.append(sk)
        for ck, _ in chains:
            ck.replay()
        for sk in streams:
            hip.shl_mi355x_stream_sync(sk)
        reps = 100
        t0 = time.perf_counter()
        for _ in range(reps):
            for ck, _ in chains:
                ck.replay()
        for sk in streams:
            hip.shl_mi355x_stream_sync(sk)
        dt_c = time.perf_counter() - t0
        result["concurrent_streams"] = {"streams": nstreams, "images_per_sec": nstreams * reps * batch / dt_c,
                                        "note": "independent batch-1 chains on separate HIP streams of one GPU"}
        for ck, hk in chains:
            ck.release()
            hk.free_all()
        del chains
        # whole model through csinn_session_run with HOST input / output tensors: H2D + one
        # hipGraph replay (28 convs + avgpool + softmax) + D2H + sync per image
        ms = wl.ModelSession(fe, pkg.API_MI355X, args.dtype, layout)
        mode = opt.shl_mi355x_session_is_device_resident(ms.sess)
        x = ms.synthetic_input(0)
        for _ in range(10):
            ms.run(x)
        t0 = time.perf_counter()
        reps = 200
        for _ in range(reps):
            ms.run(x)
        dt_s = (time.perf_counter() - t0) / reps
        result["end_to_end_session"] = {
            "workload": "mobilenetv1 %s %s whole model via csinn_session_run, host tensors (PCIe-inclusive)" % (args.dtype, layout),
            "images_per_sec": 1.0 / dt_s, "ms_per_image": dt_s * 1e3, "layers": ms.n_layers,
            "device_mode": {0: "host-staged", 1: "device eager", 2: "device hipGraph"}[mode]}
        ms.close()
    if args.workload == "mobilenetv1" and world == 1 and batch == 1 and not args.no_configs:
        # the headline through the API ("print both"): the whole model -- the same convolution launches
        # + global_avgpool2d + softmax -- as ONE csinn_session_run per image, tensors in HBM
        reps = 200
        # the same session with its input and output tensors in HBM (csinn_update_input / _output with
        # device buffers): csinn_session_run only enqueues the captured graph; one sync at the end
        hio = CHbm(hip)
        d_in = hio.alloc(224 * 224 * 3 * (1 if args.dtype == "int8" else 2))
        d_out = hio.alloc(1000 * (1 if args.dtype == "int8" else 2))
        msd = wl.ModelSession(fe, pkg.API_MI355X, args.dtype, layout, dev_in=d_in, dev_out=d_out)
        hio.upload(d_in, msd.synthetic_input(0))
        sst = opt.shl_mi355x_session_stream(msd.sess)
        for _ in range(10):
            msd.run_async()
        hip.shl_mi355x_stream_sync(sst)
        t0 = time.perf_counter()
        for _ in range(reps):
            msd.run_async()
        hip.shl_mi355x_stream_sync(sst)
        dt_d = (time.perf_counter() - t0) / reps
        result["session_device_io"] = {
            "workload": "mobilenetv1 %s %s whole model via csinn_session_run, input/output tensors in HBM" % (args.dtype, layout),
            "images_per_sec": 1.0 / dt_d, "ms_per_image": dt_d * 1e3, "layers": msd.n_layers,
            "fused_pairs": opt.shl_mi355x_session_fused_pairs(msd.sess),
            "vs_headline": "the headline's launches + global_avgpool2d + softmax (+ a relu the session folds); headline ms_per_step %.4f" % result["ms_per_step"]}
        msd.close()
        hio.free_all()
    if world > 1:  # the CPU baseline is a property of the node: measured by the N=1 run only
        result["cpu_baseline"] = {"value": None, "unit": "GOPS", "cores": 0, "kind": "reference",
                                  "sample": "measured by the N=1 run only"}
    elif not args.no_cpu_baseline:
        result["cpu_baseline"] = run_cpu_baseline(args)
    else:
        result["cpu_baseline"] = {"value": None, "unit": "GOPS", "cores": 0, "kind": "port", "sample": "skipped"}
    emit(result)
    finish()


if __name__ == "__main__":
    main()
