#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X CSI-NN2 backend.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

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
  configs      -- (N = 1 runs) the other single-GPU configurations of BASELINE.json, each with its own
                  roofline: configs[2] ResNet-50 3x3 set int8 batch 128 in NCHW (as BASELINE names it)
                  and in NHWC, against the int8 MFMA peak; configs[3] MobileNetV1 binary16 NCHW batch 1
                  (c906_mobilenetv1_f16 shapes).  --no-configs skips them; --extra adds serving views.

--workload resnet50_3x3 --total-batch 1024 is BASELINE configs[4]: the batch is SHARDED over the ranks
(sharding.shard_batch: 128 images per GPU at N = 8), weights broadcast once from rank 0.
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
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--detail", action="store_true", help="print the per-layer table to stderr")
    ap.add_argument("--steps-only", action="store_true",
                    help="run warmup + timed steps and stop (for rocprofv3 --pmc passes: only step launches are seen)")
    ap.add_argument("--traffic-file", default="",
                    help="JSON from tools/pmc_traffic.py (default profiles/traffic_<workload>_<dtype>_<layout>_b<batch>.json)")
    return ap.parse_args()


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
    print(json.dumps({"value": ops_done / t_used / 1e9, "unit": "GOPS", "cores": cores, "kind": kind,
                      "imgs_per_sec": (ops_done / sum(wl.layer_ops(l) for l in layers)) / t_used,
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
class TorchHBM:
    """Device memory from torch (plumbing only): raw pointers go into csinn_tensor.data."""

    def __init__(self, torch, device):
        self.torch, self.device, self.live = torch, device, []

    def alloc(self, nbytes):
        t = self.torch.empty(max(int(nbytes), 16), dtype=self.torch.uint8, device=self.device)
        self.live.append(t)
        return t.data_ptr()

    def upload(self, ptr, host):
        t = next(x for x in self.live if x.data_ptr() == ptr)
        flat = self.torch.from_numpy(np.ascontiguousarray(host).view(np.uint8).reshape(-1))
        t[:flat.numel()].copy_(flat)
        self.torch.cuda.synchronize()


def time_groups(chain, hip, opt, stream, reps=20):
    """Average duration of every launch of one pass (a layer, or a fused pointwise + depthwise pair),
    measured with HIP events recorded on the launch stream around `reps` back-to-back launches of it
    (captured in a graph: no host gaps)."""
    out = []
    ev0, ev1 = hip.shl_mi355x_event_create(), hip.shl_mi355x_event_create()
    ms = C.c_float()
    for u in range(len(chain.units)):
        opt.shl_mi355x_set_stream(stream)
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


def main():
    args = parse_args()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook: run every rank on ONE device over gloo (a 1-GPU box can then exercise the N > 1 code
    # path: weight broadcast, replica agreement, barriers, max-over-ranks); never set by the driver
    single_dev = os.environ.get("SHL_BENCH_SINGLE_DEVICE") == "1"
    if single_dev:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if single_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    pkg = importlib.import_module("csi-nn2_amd")
    wl = importlib.import_module("csi-nn2_amd.workloads")
    par = importlib.import_module("csi-nn2_amd.sharding")
    fe = pkg.load_frontend("standalone")
    hip, opt = pkg.load_backend(fe)
    pkg.check(hip.shl_mi355x_set_device(local_rank), hip, "set_device")
    arch = C.create_string_buffer(64)
    cus = C.c_int32()
    hip.shl_mi355x_device_info(arch, 64, C.byref(cus), None)

    hbm = TorchHBM(torch, torch.device("cuda", local_rank))
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
    fuse = chained and args.dtype == "int8" and layout == "NHWC" and not args.no_fuse
    # rank 0 owns the real weights; other ranks build their plans from a different seed and must
    # receive rank 0's packed blocks over RCCL before they can agree with it
    chain = wl.LayerChain(fe, hip, opt, layers, batch, hbm.alloc, hbm.upload, dtype=args.dtype, layout=layout,
                          seed=1234 if rank == 0 else 999 + rank, chained=chained, fuse=fuse)
    bcast = None
    if world > 1:
        bcast = par.broadcast_weights(chain, torch, dist, hip, opt, rank, world, src=0, prefer_c=not single_dev)
        par.assert_replicas_agree(chain, torch, dist, hip)

    stream = hip.shl_mi355x_stream_create()
    chain.capture(stream)
    for _ in range(args.warmup):
        chain.replay()
    hip.shl_mi355x_stream_sync(stream)
    windows = []
    for _ in range(max(1, args.windows)):
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            chain.replay()
        hip.shl_mi355x_stream_sync(stream)
        torch.cuda.synchronize()
        w = time.perf_counter() - t0
        if dist:
            dist.barrier()
            t = torch.tensor([w], dtype=torch.float64, device="cpu" if single_dev else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            w = float(t.item())
        windows.append(w)
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
        "timing": "median of %d windows of %d steps, each bracketed by barrier + synchronise, MAX over ranks" % (
            len(windows), args.steps),
        "config": {"workload": "%s %s %s, %d conv layers in %d launches%s, batch %d per GPU%s, hipGraph replay via csinn_* C API"
                               % (args.workload, args.dtype, layout, len(layers), len(chain.units),
                                  " (pointwise+depthwise pairs fused as csinn_session_setup does)" if len(chain.units) < len(layers) else "",
                                  batch, " (total batch %d sharded)" % args.total_batch if sharded else ""),
                   "per_gpu_batch": batch,
                   "parallelism": ("batch shard x%d" % world if sharded else "replicas x%d" % world) +
                                  (" (weights broadcast once: %s)" % bcast if bcast else ""),
                   "ops_per_image": ops_per_image, "algorithmic_bytes_per_image": chain.total_bytes() // batch,
                   "device": arch.value.decode(), "compute_units": cus.value},
    }

    if args.steps_only:
        if rank == 0:
            print(json.dumps(result))
        chain.release()
        if dist:
            dist.barrier()
            dist.destroy_process_group()
        return
    if rank == 0:
        per_layer = time_groups(chain, hip, opt, stream)
        roof, groups = summarise_kernels(chain, wl, per_layer, bound)
        roof["mfma_frac_info"] = (chain.total_ops() / sum(per_layer) / 1e12) / (
            I8_MFMA_PEAK_TOPS if args.dtype == "int8" else F16_MFMA_PEAK_TFLOPS)
        attach_traffic(roof, args.traffic_file or os.path.join(
            os.path.dirname(os.path.abspath(__file__)), "profiles",
            "traffic_%s_%s_%s_b%d.json" % (args.workload, args.dtype, layout, batch)))
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
        if world == 1 and args.workload == "mobilenetv1" and not args.no_configs:
            # the other single-GPU configurations of BASELINE.json: per-layer graph timing (independent layers,
            # every layer its own launch), each with the roofline of its dominant kernel
            others = [
                ("configs[2]", "resnet50 3x3 set int8 NCHW batch 128", wl.RESNET50_3X3, 128, "int8", "NCHW", "mfma", 3),
                ("configs[2] (NHWC view)", "resnet50 3x3 set int8 NHWC batch 128", wl.RESNET50_3X3, 128, "int8", "NHWC", "mfma", 3),
                ("configs[3]", "mobilenetv1 fp16 NCHW batch 1 (c906_mobilenetv1_f16 shapes)", wl.MOBILENETV1, 1, "f16", "NCHW", "hbm", 20),
            ]
            if args.extra:
                others.append(("configs[1] (throughput view)", "mobilenetv1 int8 NHWC batch 128, every layer its own launch",
                               wl.MOBILENETV1, 128, "int8", "NHWC", "hbm", 3))
            result["configs"] = []
            prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
            for tag, name, layers_x, batch_x, dtype_x, layout_x, bound_x, reps_x in others:
                hbm2 = TorchHBM(torch, torch.device("cuda", local_rank))
                rc = wl.LayerChain(fe, hip, opt, layers_x, batch_x, hbm2.alloc, hbm2.upload, dtype=dtype_x,
                                   layout=layout_x, seed=4321, chained=False)
                rt = time_groups(rc, hip, opt, stream, reps=reps_x)
                rroof, rgroups = summarise_kernels(rc, wl, rt, bound_x)
                wname = "resnet50_3x3" if layers_x is wl.RESNET50_3X3 else "mobilenetv1"
                attach_traffic(rroof, os.path.join(prof, "traffic_%s_%s_%s_b%d.json" % (wname, dtype_x, layout_x, batch_x)))
                unit = "GOPS" if dtype_x == "int8" else "GFLOPS"
                peak = I8_MFMA_PEAK_TOPS if dtype_x == "int8" else F16_MFMA_PEAK_TFLOPS
                entry = {"baseline_config": tag, "workload": name, "metric": "conv_" + unit.lower(),
                         "value": rc.total_ops() / sum(rt) / 1e9, "unit": unit,
                         "images_per_sec": batch_x / sum(rt), "ms_per_pass": sum(rt) * 1e3,
                         "dtype": "i8" if dtype_x == "int8" else "f16", "layers": len(layers_x),
                         "mfma_frac_whole_set": rc.total_ops() / sum(rt) / 1e12 / peak,
                         "roofline": rroof,
                         "kernels": {k: {"launches": v["launches"], "us_total": v["time"] * 1e6,
                                         "TOPs": v["ops"] / v["time"] / 1e12, "GBps": v["bytes"] / v["time"] / 1e9}
                                     for k, v in rgroups.items()}}
                result["configs"].append(entry)
                if args.detail:
                    sys.stderr.write("---- %s\n" % name)
                    for u, t in enumerate(rt):
                        sys.stderr.write("%-52s %-32s %8.2f us %8.1f GB/s %8.2f TOP/s\n" % (
                            rc.unit_name(u), rc.unit_kernel_name(u), t * 1e6, rc.unit_bytes(u) / t / 1e9,
                            rc.unit_ops(u) / t / 1e12))
                rc.release()
                del hbm2
        if args.extra and args.workload == "mobilenetv1":
            # serving view: independent batch-1 requests in flight on several streams of ONE GPU (the
            # headline `value` is strictly one request at a time; a single chain leaves most CUs idle)
            nstreams = 4
            chains, streams = [], []
            for k in range(nstreams):
                hk = TorchHBM(torch, torch.device("cuda", local_rank))
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
            for ck, _ in chains:
                ck.release()
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
            # the same session with its input and output tensors in HBM (csinn_update_input / _output with
            # device buffers): csinn_session_run only enqueues the captured graph; one sync at the end
            hio = TorchHBM(torch, torch.device("cuda", local_rank))
            d_in = hio.alloc(224 * 224 * 3 * (1 if args.dtype == "int8" else 2))
            d_out = hio.alloc(1000 * (1 if args.dtype == "int8" else 2))
            hio.upload(d_in, x)
            msd = wl.ModelSession(fe, pkg.API_MI355X, args.dtype, layout, dev_in=d_in, dev_out=d_out)
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
                "fused_pairs": opt.shl_mi355x_session_fused_pairs(msd.sess)}
            msd.close()
            del hio
        if world > 1:  # the CPU baseline is a property of the node: measured by the N=1 run only
            result["cpu_baseline"] = {"value": None, "unit": "GOPS", "cores": 0, "kind": "reference",
                                      "sample": "measured by the N=1 run only"}
        elif not args.no_cpu_baseline:
            result["cpu_baseline"] = run_cpu_baseline(args)
        else:
            result["cpu_baseline"] = {"value": None, "unit": "GOPS", "cores": 0, "kind": "port", "sample": "skipped"}
        print(json.dumps(result))
    chain.release()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
