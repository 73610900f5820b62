"""Headline benchmark: train images/sec of one full G+D step at 1024x1024 (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1, the driver's form)
`python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks itself (re-exec under
torch.distributed.run on 127.0.0.1); rank 0 prints the line.

A step = SKITGModel.optimize_parameters on one synthetic batch (4 images / GPU, 64 tactile
patches each) that is already resident in HBM: generator forward, patch gather, D1 update,
D2 update (incl. the reference's full-resolution visualisation pass), G update, three fused
Adam steps.  LPIPS / CLIP terms are off (their weights cannot exist offline; stated in `config`).
Rank 0 prints ONE JSON line with the extra `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "visual-tactile-synthesis_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

# Host threads: torch's CPU ops fan out over an OpenMP pool (128 workers on the MI355X hosts) whose workers SPIN after each
# parallel region; that starved the HIP runtime's completion thread and stalled graph-replayed steps by 70..170 ms
# (tools/probes/stall_bisect2.py).  Passive waiting must be chosen before libgomp starts, i.e. before `import torch`.
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
# RCCL's own account of the communicator: rank 0 parses "nranks N" out of its init lines into comm.nranks_seen.  RCCL caches its debug
# settings when the library initialises, so they are chosen here, before torch loads it.
RCCL_LOG = None
if (int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("VTS_DDP_FORCE", "0") == "1") and "NCCL_DEBUG_FILE" not in os.environ:
    RCCL_LOG = "/tmp/vts_rccl_%d.log" % os.getpid()
    os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT", NCCL_DEBUG_FILE=RCCL_LOG)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
MFMA_F32_PEAK_TF = 157.3   # dense fp32 MFMA peak (v_mfma_f32_16x16x4_f32)


def build_model(size, batch, model_name, quiet=True, netG="unet256_custom", lpips=False, p2p_vgg=False):
    import contextlib
    import io

    from models import create_model
    from options.train_options import TrainOptions

    flags = ("--model %s --gpu_ids 0 %s--use_vision_aided_loss False "
             "--checkpoints_dir /tmp/vts_bench --name bench --crop_size %d --batch_size %d --netG %s"
             % (model_name, "" if lpips else "--lambda_G1_lpips 0 --lambda_G2_lpips 0 ", size, batch, netG))
    if model_name == "pix2pixHD":   # reference defaults (ngf 64, 4 downsamplings, 9 blocks), VGG term off (no weights offline)
        flags = ("--model pix2pixHD --gpu_ids 0 --no_vgg_loss %s --checkpoints_dir /tmp/vts_bench --name bench --batch_size %d "
                 "--dataset_mode patchskit" % ("False" if p2p_vgg else "True", batch))
    ctx = contextlib.redirect_stdout(io.StringIO()) if quiet else contextlib.nullcontext()
    with ctx:
        opt = TrainOptions(cmd_line=flags).parse()
        opt.gpu_ids = [torch.cuda.current_device()]
        model = create_model(opt)
        model.setup(opt)
        model.parallelize()
        model.train()
    return model, opt


def make_batch(size, batch, rank, style_dim, quantize8=False):
    from torch.utils.data import default_collate

    from data.synthetic_dataset import make_sample

    return default_collate([make_sample(size, 64, 64, 1234 + 100003 * rank + i, style_dim=style_dim, quantize8=quantize8) for i in range(batch)])


def make_patch_batch(batch, rank, patch=32, height=None, width=None):
    from torch.utils.data import default_collate

    from data.synthetic_dataset import make_patch_sample

    return default_collate([make_patch_sample(1234 + 100003 * rank + i, patch=patch, height=height, width=width) for i in range(batch)])


def csrc_sha16():
    """hash of the kernel sources: PMC summaries under profiles/ carry the hash they were measured on, and are only quoted while it matches"""
    import glob
    import hashlib

    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "visual-tactile-synthesis_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "visual-tactile-synthesis_amd", "csrc", "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def newest_profile(pattern):
    """newest profiles/<pattern> JSON measured on the CURRENT kernel sources (tools/pmc_summary.py / tools/pmc_mfma.py stamp csrc_sha16), or None"""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("csrc_sha16") == csrc_sha16():
            return os.path.basename(f), d
    return None, None


def kernel_roofline(model, batch_dict, detail_path=None):
    """One extra (untimed-for-throughput) step with HIP events around every launch, on the launch stream."""
    from vts import ops

    from vts import engine
    graph_flag, model.opt.use_hip_graph = model.opt.use_hip_graph, False   # per-launch events need eager launches
    par_flag, engine.PARALLEL_SCALES = engine.PARALLEL_SCALES, False       # ... issued one after the other on one stream
    ops.TIMER = []
    model.optimize_parameters(epoch=1)
    torch.cuda.synchronize()
    rec, ops.TIMER = ops.TIMER, None
    model.opt.use_hip_graph = graph_flag
    engine.PARALLEL_SCALES = par_flag
    agg, det = {}, {}
    for label, nbytes, flops, e0, e1, detail in rec:
        dt = e0.elapsed_time(e1) * 1e-3
        a = agg.setdefault(label, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += dt
        a[2] += nbytes
        a[3] += flops
        if detail_path and detail:
            b = det.setdefault(label + " | " + detail, [0, 0.0, 0.0, 0.0])
            b[0] += 1
            b[1] += dt
            b[2] += nbytes
            b[3] += flops
    if detail_path:
        rows = sorted(det.items(), key=lambda kv: -kv[1][1])
        with open(detail_path, "w") as f:
            f.write("# launches  total_ms  avg_us  achieved_TFLOP/s  achieved_GB/s  kernel | shape\n")
            for k, (c, t, nb, fl) in rows:
                f.write("%4d %8.3f %8.1f %8.2f %8.1f  %s\n" % (c, t * 1e3, t / c * 1e6, fl / t / 1e12, nb / t / 1e9, k))
    total = sum(a[1] for a in agg.values())
    # Roofline time of the WHOLE step: every launch's algorithmic max(bytes / HBM peak, flops / fp32 MFMA peak), summed (launches without a
    # work model -- reductions of partials, finalisers -- count as zero: they are overhead, not work).  main() divides it by ms_per_step.
    step_t_roof = sum(max(nbytes / (HBM_PEAK_GBS * 1e9), flops / (MFMA_F32_PEAK_TF * 1e12)) for _, nbytes, flops, _, _, _ in rec)
    # The dominant KERNEL = the kernel template (all of its instances: tile shapes are a dispatch detail) with the largest total time.
    # Labels naming several kernels of one C call ("a_kernel+b_kernel") are not one kernel.  Its launches are HBM-bound on some
    # shapes and MFMA-bound on others (per-launch t_roof = max(bytes / BW, flops / P)); `bound` is the class holding more of the
    # family's time, `achieved` = that class's algorithmic work / that class's measured time, and `frac_all_launches` =
    # sum(t_roof) / sum(t) over every launch of the family, whatever its bound.
    def fam_of(lbl):
        return lbl.split("<")[0].split("+")[0]

    fams = {}
    for label, nbytes, flops, e0, e1, detail in rec:
        if "_kernel" not in label or "_kernel+" in label:
            continue
        dt = e0.elapsed_time(e1) * 1e-3
        t_hbm, t_mfma = nbytes / (HBM_PEAK_GBS * 1e9), flops / (MFMA_F32_PEAK_TF * 1e12)
        cls = "hbm" if t_hbm >= t_mfma else "mfma"
        f = fams.setdefault(fam_of(label), {"t": 0.0, "n": 0, "t_roof": 0.0, "bytes": 0.0, "flops": 0.0,
                                            "hbm": [0, 0.0, 0.0, 0.0], "mfma": [0, 0.0, 0.0, 0.0], "inst": {}})
        f["t"] += dt
        f["n"] += 1
        f["t_roof"] += max(t_hbm, t_mfma)
        f["bytes"] += nbytes
        f["flops"] += flops
        c = f[cls]
        c[0] += 1
        c[1] += dt
        c[2] += nbytes
        c[3] += flops
        i = f["inst"].setdefault(label, [0, 0.0, 0.0, 0.0, 0.0])
        i[0] += 1
        i[1] += dt
        i[2] += nbytes
        i[3] += flops
        i[4] += max(t_hbm, t_mfma)
    fam, f = max(fams.items(), key=lambda kv: kv[1]["t"])
    bound = "hbm" if f["hbm"][1] >= f["mfma"][1] else "mfma"
    cnt, t, nbytes, flops = f[bound]
    if bound == "hbm":
        roof = {"bound": "hbm", "achieved": nbytes / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    else:
        roof = {"bound": "mfma", "achieved": flops / t / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s"}
    roof["frac"] = roof["achieved"] / roof["peak"]
    roof["frac_all_launches"] = f["t_roof"] / f["t"]
    # HBM bytes per launch from the PMC passes of tools/pmc_traffic.sh (separate FETCH_SIZE / WRITE_SIZE runs of this same command,
    # FETCH_SIZE doubled per MI355X_MICROARCH.md), mean over ALL launches of the family (PMC rows are per instance, not per shape);
    # compare with algorithmic_bytes_per_launch_family.  PMC counters cannot be read from inside this process, so the figure comes
    # from profiles/ -- but ONLY from a summary measured on exactly these kernel sources (csrc_sha16); otherwise null.
    roof["traffic"] = None
    src, prof = newest_profile("*_traffic_pmc.json")
    traffic_prof = prof
    if prof is not None:
        num = den = 0.0
        for lbl, i in f["inst"].items():
            ent = prof["kernels"].get(lbl.split("+")[0])
            if ent:
                num += i[0] * ent["hbm_bytes_per_launch"]
                den += i[0]
        if den == f["n"]:
            roof["traffic"] = num / den
            roof["traffic_source"] = src
    # MFMA pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 256 CUs * 4 SIMDs), tools/pmc_mfma.sh), same staleness rule
    roof["mfma_util"] = roof["mfma_util_step"] = None
    src, prof = newest_profile("*_mfma_util.json")
    if prof is not None:
        num = den = 0.0
        for lbl, i in f["inst"].items():
            ent = prof["kernels"].get(lbl.split("+")[0])
            if ent:
                num += i[1] * ent["mfma_util"]
                den += i[1]
        roof["mfma_util"] = num / den if den > 0 else None   # time-weighted over the family's instances
        roof["mfma_util_step"] = prof.get("step_mfma_util")
        roof["mfma_util_source"] = src
    roof["kernel"] = fam
    roof["launches_per_step"] = cnt
    roof["avg_launch_us"] = t / cnt * 1e6
    roof["algorithmic_bytes_per_launch"] = nbytes / cnt
    roof["algorithmic_flops_per_launch"] = flops / cnt
    roof["launches_per_step_family"] = f["n"]
    roof["algorithmic_bytes_per_launch_family"] = f["bytes"] / f["n"]
    roof["share_of_timed_kernels"] = f["t"] / total
    inst = sorted(f["inst"].items(), key=lambda kv: -kv[1][1])[:6]
    def pmc_ratio(k, v):   # measured HBM bytes per launch of this instance (train-step launches only) / its algorithmic bytes per launch
        ent = traffic_prof["kernels"].get(k.split("+")[0]) if traffic_prof else None
        return round(ent["hbm_bytes_per_launch"] / (v[2] / v[0]), 3) if ent and v[2] > 0 else None

    roof["instances"] = [{"kernel": k, "launches": v[0], "avg_us": round(v[1] / v[0] * 1e6, 2), "bytes_per_launch": round(v[2] / v[0]),
                          "flops_per_launch": round(v[3] / v[0]), "frac": round(v[4] / v[1], 4), "traffic_ratio": pmc_ratio(k, v)} for k, v in inst]
    if roof["traffic"] is not None:
        roof["traffic_ratio"] = roof["traffic"] / roof["algorithmic_bytes_per_launch_family"]
    breakdown = sorted(((k, v[1] * 1e3, v[0]) for k, v in agg.items()), key=lambda x: -x[1])
    roof["breakdown_ms"] = {k: round(ms, 3) for k, ms, _ in breakdown[:8]}
    roof["step_t_roof_ms"] = step_t_roof * 1e3
    roof["launches_per_step_all"] = len(rec)
    roof["serialized_kernel_ms"] = total * 1e3
    # The figures above describe the EAGER single-stream schedule (one launch after the other: clean per-launch times).  The step that
    # `ms_per_step` times is the replayed LANE schedule: the discriminator scales, the D1 pass on the real images beside the generator
    # forward and the weight gradients run on side streams, which un-batches some passes -- more launches of the same kernels.  Its
    # launch list is taken from one more eager step with the lanes on (events on each launch's own stream: counts are exact, durations
    # include the slow-down of kernels that overlap), its node count from the captured graphs themselves.
    ops.TIMER = []
    graph_flag, model.opt.use_hip_graph = model.opt.use_hip_graph, False
    model.optimize_parameters(epoch=1)
    torch.cuda.synchronize()
    rec2, ops.TIMER = ops.TIMER, None
    model.opt.use_hip_graph = graph_flag
    fam2 = [(e0.elapsed_time(e1) * 1e-3) for label, _, _, e0, e1, _ in rec2 if "_kernel" in label and "_kernel+" not in label and fam_of(label) == fam]
    roof["lane_schedule"] = {"launches_all": len(rec2), "launches_family": len(fam2), "family_ms_overlapped": round(sum(fam2) * 1e3, 3)}
    nodes = getattr(model, "graph_nodes", None)
    roof["launches_per_step_replayed"] = sum(k for _, k in nodes) if nodes else None          # kernel nodes of the captured step's graphs
    roof["graph_nodes_per_step"] = sum(n for n, _ in nodes) if nodes else None                # ... all nodes (memset / memcpy / event nodes included)
    return roof


def cpu_baseline(size, style_dim, netG="unet256_custom", warmup=3, steps=10, budget_s=60.0):
    """The CPU oracle (PyTorch-CPU restatement pinned to the reference) on this host's cores, N=1: BASELINE.md's protocol, 3 warm-up +
    10 timed steps, median (bounded: stops early if the budget runs out and says how many steps it timed)."""
    from torch.utils.data import default_collate

    from data.synthetic_dataset import make_sample
    from oracle import detrand, nets, step

    # PyTorch-CPU convolutions with 3..160 channels stop scaling (and collapse from oversubscription) well
    # before a 256-thread host is full: 16 threads is near the best rate for this workload.
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    g_shapes = (nets.resnet_param_shapes(n_blocks=int(netG[len("resnet_")])) if netG.startswith("resnet_")
                else nets.g_param_shapes(style_nc=style_dim))
    sd = (detrand.test_weights(g_shapes, 1), detrand.test_weights(nets.d_param_shapes(4), 2),
          detrand.test_weights(nets.d_param_shapes(7), 3))
    batch = default_collate([make_sample(size, 64, 64, 99, style_dim=style_dim)])
    import random

    random.seed(0)
    cnt = int(nets.dilated_mask_positions(batch["M"].float()).shape[0])
    adam = {k: step.new_adam_state() for k in ("G", "D", "D2")}
    style = batch.get("style_code")
    times, t_start = [], time.time()
    for it in range(warmup + steps):
        draws = {"aug": torch.rand(4, 1), "more_idx": torch.tensor([random.sample(range(cnt), 32)])}
        t0 = time.time()
        step.train_step(sd[0], sd[1], sd[2], adam, batch, draws, opt=step.hp(netG=netG), style_code=style, record=False)
        times.append(time.time() - t0)
        if it >= warmup and time.time() - t_start > budget_s:
            break
    timed = sorted(times[warmup:]) or sorted(times[-1:])
    t = timed[len(timed) // 2]
    return {"value": 1.0 / t, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "median of %d oracle train steps after %d warm-up, N=1, %dx%d, same flags; %d of %d host cores: PyTorch-CPU's convolutions "
                      "with 3..160 channels stop scaling there and their weight gradient drifts 5e-3 from the single-thread result beyond"
                      % (len(timed), min(warmup, len(times) - len(timed)), size, size, torch.get_num_threads(), os.cpu_count() or 1)}


def collective_ab(model, world, dev, barrier, steps=10):
    """Warm-up A/B of the two bucket collectives at world > 1: torch.distributed's all_reduce (RCCL's choice of algorithm) against the
    library's explicit reduce-scatter + all-gather on its own communicator (VTS_DDP_DIRECT).  The direct path is first
    CHECKED against all_reduce on a buffer whose length is not a multiple of the world size (it had never run on more than one device);
    any failure keeps the torch path.  The faster one is left switched on for the timed region; both times go into the `comm` block."""
    from vts import ddp

    def timed():
        for _ in range(2):
            model.optimize_parameters(epoch=1)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            model.optimize_parameters(epoch=1)
        barrier()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) / steps * 1e3

    res = {"torch_all_reduce_ms_per_step": timed(), "direct_ms_per_step": None, "direct_checked": False, "chosen": "torch"}

    def all_agree(flag):      # every rank takes the same branch below: a local failure must not leave the others inside a collective
        t = torch.tensor([1.0 if flag else 0.0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() == 1.0)

    ok = True
    try:
        ddp.direct_comm()                                     # the library's communicator (its id travels over the default group)
    except Exception as e:      # noqa: BLE001
        ok = False
        res["direct_error"] = repr(e)[:300]
    if all_agree(ok):
        n = 1000003                                           # not a multiple of any world size: exercises the tail all-reduce
        x = (torch.arange(n, device=dev, dtype=torch.float32) % 97.0) * float(dist.get_rank() + 1)
        ref = x.clone()
        dist.all_reduce(ref)
        torch.cuda.synchronize()
        try:
            ddp.DIRECT = True
            b = ddp.GradBucket(x)
            b.start()
            b.wait()
            torch.cuda.synchronize()
            ok = bool(torch.equal(x, ref))
        except Exception as e:      # noqa: BLE001
            ok = False
            res["direct_error"] = repr(e)[:300]
        ddp.DIRECT = False
        res["direct_checked"] = all_agree(ok)
        if res["direct_checked"]:
            ddp.DIRECT = True
            res["direct_ms_per_step"] = timed()
            ddp.DIRECT = False
    use_direct = bool(res["direct_checked"] and res["direct_ms_per_step"] is not None and res["direct_ms_per_step"] < res["torch_all_reduce_ms_per_step"])
    flag = torch.tensor([1.0 if use_direct else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)            # one decision for all ranks
    ddp.DIRECT = bool(flag.item() == 1.0)
    res["chosen"] = "direct" if ddp.DIRECT else "torch"
    barrier()
    return res


def rccl_nranks(path):
    """largest `nranks N` in RCCL's own init log of this process (NCCL_DEBUG=INFO, NCCL_DEBUG_FILE), or None"""
    import re

    if not path or not os.path.exists(path):
        return None
    found = [int(m) for m in re.findall(r"nranks (\d+)", open(path, errors="replace").read())]
    return max(found) if found else None


def infer_measure(args, steps=None, warmup=None):
    """Generator-only forward (test() of the model, BASELINE config 4: 16 images/GPU): seconds per step, images per step, the options"""
    batch_n = 16 if args.batch == 4 else args.batch
    steps, warmup = steps or args.steps, warmup or args.warmup
    model, opt = build_model(args.size, batch_n, args.model)
    opt.use_hip_graph = not args.no_graph
    opt.skip_D2_visualisation_pass = bool(args.no_viz)
    style_dim = opt.style_code_dim if getattr(opt, "use_style_code", False) else 0
    model.eval()
    model.set_input(make_batch(args.size, batch_n, 0, style_dim), phase="test")
    for _ in range(warmup):
        model.test()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.test()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, batch_n, opt


def infer_bench(args):
    """ms per image, inputs resident in HBM."""
    dt, batch_n, opt = infer_measure(args)
    emit(json.dumps({
        "metric": "inference_ms_per_image", "value": dt / batch_n * 1e3, "unit": "ms/image", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": False,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s generator forward (test()), %dx%d, %d images/GPU" % (args.model, args.size, args.size, batch_n),
                   "hip_graph": bool(opt.use_hip_graph)},
    }))


_JSON_FD = None


def emit(line):
    """the ONE JSON line goes to the real stdout; everything else that writes to fd 1 (RCCL prints a version banner there when
    the communicator is created) has been sent to stderr by main()"""
    sys.stdout.flush()
    os.write(_JSON_FD if _JSON_FD is not None else 1, (line + "\n").encode())


def spawn_ranks(n):
    """`python bench.py --gpus N` outside a launcher: re-exec this command line under torch.distributed.run, one rank per GPU of this
    node (rendezvous on 127.0.0.1, a free port).  Rank 0's JSON line reaches the real stdout through the inherited descriptor."""
    import socket
    import subprocess

    if torch.cuda.device_count() < n:
        raise SystemExit("bench.py: --gpus %d but this node exposes %d GPU(s)" % (n, torch.cuda.device_count()))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    os.dup2(_JSON_FD, 1)      # the children write the line themselves
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the timed region is >= 1.5 s of steady state (20 steps = 0.14 s was too short for the driver's clock / gpu_busy sampling)
    ap.add_argument("--steps", type=int, default=250)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU")
    ap.add_argument("--model", type=str, default="skitG")
    ap.add_argument("--netG", type=str, default="unet256_custom",
                    help="generator: unet256_custom (headline config) | resnet_{4,6,9}blocks (alternate; needs --model sinskitG)")
    ap.add_argument("--p2p_size", type=int, default=32, help="pix2pixHD only: side of the (square) training images / patches")
    ap.add_argument("--p2p_h", type=int, default=0, help="pix2pixHD only: image height (with --p2p_w: BASELINE config 3 is --p2p_h 1024 --p2p_w 2048 --batch 1)")
    ap.add_argument("--p2p_w", type=int, default=0, help="pix2pixHD only: image width")
    ap.add_argument("--p2p_vgg", action="store_true", help="pix2pixHD only: keep the reference's default VGG19 feature loss on (seeded stand-in weights)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--lpips", action="store_true",
                    help="SECONDARY workload: the same step with the reference's default LPIPS-VGG16 terms on (lambda_G1_lpips 1, lambda_G2_lpips 10; "
                         "stand-in VGG weights -- the arithmetic is the same): ~7.7 TFLOP of 3x3 convolutions per step on the GEMM-class kernels")
    ap.add_argument("--no_graph", action="store_true", help="launch every kernel eagerly instead of replaying HIP graphs")
    ap.add_argument("--no_viz", action="store_true",
                    help="leave out the reference step's full-resolution D2 visualisation pass (2.85 GFLOP/image, no gradient): SURVEY 8d "
                         "asks for the rate with and without it; the default (and the headline) includes it")
    ap.add_argument("--infer", action="store_true",
                    help="measure the inference forward instead (BASELINE config 4: generator only, 16 images/GPU): ms per image")
    ap.add_argument("--detail", type=str, default=None, help="write a per-(kernel, shape) timing table to this path")
    ap.add_argument("--train_only", action="store_true",
                    help="profiling form: warm-up + timed train steps on the resident batch and NOTHING else in the process (no fresh-input loop, "
                         "no per-launch event step, no host-enqueue probes, no CPU baseline, no inference leg), so that every kernel row of a "
                         "rocprofv3 / PMC pass over this command is launches-per-step x (warmup + steps) train-step launches")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)

    from vts import ddp

    rccl_log = RCCL_LOG
    rank, world = ddp.init_from_env("cuda")
    if world != args.gpus and not (world == 1 and ddp.FORCE):
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(1234 + rank)      # per-rank DiffAugment draws (the default generator is seeded identically on every rank)
    if args.infer:
        return infer_bench(args)
    model, opt = build_model(args.size, args.batch, args.model, netG=args.netG, lpips=args.lpips, p2p_vgg=args.p2p_vgg)
    if args.lpips and args.steps == 250:
        args.steps, args.warmup = 30, 4      # ~70 ms per step
    opt.use_hip_graph = not args.no_graph
    opt.skip_D2_visualisation_pass = bool(args.no_viz)
    style_dim = opt.style_code_dim if getattr(opt, "use_style_code", False) else 0
    # pix2pixHD: the reference trains it on 32x32 patches (default); --p2p_size S feeds S x S images instead (BASELINE config 3)
    p2p_h, p2p_w = args.p2p_h or args.p2p_size, args.p2p_w or args.p2p_size
    batch = (make_patch_batch(args.batch, rank, args.p2p_size, p2p_h, p2p_w) if args.model == "pix2pixHD"
             else make_batch(args.size, args.batch, rank, style_dim))
    model.set_input(batch, phase="train")       # H2D once: inputs are resident in HBM before the timed region

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        model.optimize_parameters(epoch=1)
    barrier()
    ab = None
    if ((world > 1 and os.environ.get("VTS_DDP_AB", "1") != "0") or (ddp.active() and os.environ.get("VTS_DDP_AB") == "force")) and "VTS_DDP_DIRECT" not in os.environ:
        ab = collective_ab(model, world, dev, barrier)      # torch.distributed all_reduce vs the library's reduce-scatter + all-gather
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # one event per step on the launch stream: spread, no sync
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        model.optimize_parameters(epoch=1)
        marks[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    spread = {"min": per_step[0], "median": per_step[len(per_step) // 2], "max": per_step[-1], "p90": per_step[int(0.9 * (len(per_step) - 1))]}
    per_rank = None
    if world > 1 or ddp.active():
        # every rank's own clock and device identity, gathered over the collective backend itself: a SCALE record then shows by itself
        # that N ranks on N different devices took part
        props = torch.cuda.get_device_properties(dev)
        bus = float(getattr(props, "pci_bus_id", -1)) + 256.0 * float(getattr(props, "pci_domain_id", 0))
        mine = torch.tensor([float(rank), float(torch.cuda.current_device()), bus, dt / args.steps * 1e3], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(max(world, 1))]
        dist.all_gather(allr, mine)
        per_rank = [[float(v) for v in r.tolist()] for r in allr]
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    losses = model.get_current_losses()
    finite = all(v == v and abs(v) < 1e30 for v in losses.values())
    # the same K steps with a FRESH host batch per step: set_input (H2D of S / I / M / patches, masking, candidate map) inside the timed
    # region, as a train.py loop pays it (`value` keeps the contract: inputs resident in HBM)
    fresh_ms = None
    if world == 1 and args.model != "pix2pixHD" and not args.lpips and not args.train_only:
        def pinned(b):   # what the package's DataLoader hands over (data/__init__.py: pin_memory=True)
            return {k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in b.items()}

        # (as a train.py loop over a material gets them: 8-bit PNG pixels; the dataset front-ends hand the bytes over next to the float
        #  tensors -- S_u8 / I_u8 / M_u8 -- and set_input uploads those, a quarter of the PCIe traffic, expanded on the device bit for bit)
        batches = [pinned(make_batch(args.size, args.batch, rank + k, style_dim, quantize8=True)) for k in (0, 1)]
        for i in range(max(2, args.warmup)):
            model.set_input(batches[i % 2], phase="train")
            model.optimize_parameters(epoch=1)
        barrier()
        fresh_steps = min(args.steps, 100)
        tf = time.perf_counter()
        for i in range(fresh_steps):
            model.set_input(batches[i % 2], phase="train")
            model.optimize_parameters(epoch=1)
        barrier()
        fresh_ms = (time.perf_counter() - tf) / fresh_steps * 1e3
        model.set_input(batch, phase="train")
    comm = None
    if world > 1 or ddp.active():
        # (VTS_DDP_FORCE=1 at one rank: the same block -- what the segmented schedule and the collectives' launch cost on one device)
        # exposed communication: the same K steps without the gradient all-reduces (the replicas drift apart: timing only, last)
        ddp.COMM_OFF = True
        for _ in range(2):
            model.optimize_parameters(epoch=1)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            model.optimize_parameters(epoch=1)
        barrier()
        dt_off = time.perf_counter() - t1
        t = torch.tensor([dt_off], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_off = float(t.item())
        ddp.COMM_OFF = False
        comm = {"ms_per_step_without_allreduce": dt_off / args.steps * 1e3,
                "exposed_allreduce_ms_per_step": (dt - dt_off) / args.steps * 1e3,
                "buckets": {k: int(b.buf.numel()) * 4 for k, b in model.ddp.buckets.items()} if getattr(model, "ddp", None) else None,
                "collective": "reduce-scatter + all-gather (library communicator)" if ddp.DIRECT else "torch.distributed all_reduce (RCCL)",
                "ranks": world, "graph_segments": len(model._graphs) if getattr(model, "_graphs", None) else None,
                "ms_per_step_per_rank": {"min": min(r[3] for r in per_rank), "max": max(r[3] for r in per_rank)} if per_rank else None,
                "devices_seen": sorted({(int(r[1]), int(r[2])) for r in per_rank}) if per_rank else None,      # (local device index, PCI bus) per rank
                "ranks_gathered": len(per_rank) if per_rank else None, "nranks_seen": rccl_nranks(rccl_log), "collective_ab": ab}

    if rank == 0:
        roof = kernel_roofline(model, batch, args.detail) if world == 1 and not args.train_only else None
        if roof is not None:
            # host enqueue time of one step (no sync): tells whether the step is launch-bound
            for key, flag in (("host_enqueue_ms_eager", False), ("host_enqueue_ms", model.opt.use_hip_graph)):
                keep, model.opt.use_hip_graph = model.opt.use_hip_graph, flag
                torch.cuda.synchronize()
                th = time.perf_counter()
                model.optimize_parameters(epoch=1)
                roof[key] = (time.perf_counter() - th) * 1e3
                torch.cuda.synchronize()
                model.opt.use_hip_graph = keep
        cpu = None
        if world == 1 and not args.no_cpu_baseline and args.model != "pix2pixHD" and not args.lpips and not args.train_only:
            cpu = cpu_baseline(args.size, style_dim, netG=args.netG)
        ms = dt / args.steps * 1e3
        infer_ms = None
        if world == 1 and args.model != "pix2pixHD" and not args.lpips and args.netG == "unet256_custom" and args.batch == 4 and not args.train_only:
            del model                       # (the 16-image forward builds its own model: BASELINE config 4)
            torch.cuda.empty_cache()
            idt, ib, _ = infer_measure(args, steps=50, warmup=5)
            infer_ms = idt / ib * 1e3
        out = {
            "metric": "train_images_per_sec", "value": world * args.batch * args.steps / dt, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "ms_per_step_spread": {k: round(v, 4) for k, v in spread.items()},
            "ms_per_step_fresh_input": fresh_ms, "images_per_sec_fresh_input": (args.batch * 1e3 / fresh_ms) if fresh_ms else None,
            "inference_ms_per_image": infer_ms,                                         # generator forward, 16 images (BASELINE config 4)
            "step_roofline_frac": (roof["step_t_roof_ms"] / ms) if roof else None,      # sum of per-launch roofline times / ms_per_step
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": ("pix2pixHD G+D+D2 train step (GlobalGenerator ngf 64, ndf 64), %d %dx%d images/GPU, %s"
                             % (args.batch, p2p_w, p2p_h, "VGG19 feature loss ON (the reference's default; seeded stand-in VGG weights)" if args.p2p_vgg
                                else "VGG term off (no weights offline)")) if args.model == "pix2pixHD" else
                            "%s%s G+D1+D2 train step, %dx%d sketch->(RGB,tactile), %d images/GPU, 64 tactile patches/image, "
                            "%s" % (args.model, "" if args.netG == "unet256_custom" else " (netG %s)" % args.netG,
                                    args.size, args.size, args.batch,
                                    "LPIPS-VGG16 terms ON (reference default lambdas; seeded stand-in VGG weights), CLIP term off" if args.lpips
                                    else "LPIPS/CLIP terms off (no weights offline)"),
                "global_batch": world * args.batch, "parallelism": "dp%d" % world, "losses_finite": finite,
                "hip_graph": bool(opt.use_hip_graph), "d2_visualisation_pass": not args.no_viz,
                "train_only": bool(args.train_only),
            },
            "roofline": roof, "cpu_baseline": cpu,
        }
        if comm is not None:
            out["comm"] = comm
        emit(json.dumps(out))
    elif world > 1:
        pass
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
