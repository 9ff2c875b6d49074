"""bench.py -- the hot path's headline measurement on MI355X.

    python bench.py --gpus N --steps K --warmup W           (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W        (N > 1, one rank per GPU over RCCL)

A STEP is one training iteration of the hot path on one batch of synthetic cameras, through the
reference's own call sequence (train_internal.py:134-329): start_strategy_final -> GT staging ->
distributed_preprocess3dgs_and_all2all_final (activations + K1 per camera + the sparse exchange)
-> render_final (K3-K8) -> batched_loss_computation (band-local L1 + SSIM) -> backward (K10, mirror
exchange, K11, activation backward) -> finish_strategy_final -> grad /= bsz -> Adam step -> zero_grad
(K11 and the Adam step are ONE kernel unless --no-fuse-backward: fused_optim.py).
Nothing is skipped or cached inside the timed region.  Inputs (parameters, cameras, uint8 ground-truth
images) are resident in HBM before the timed region.

Workloads (`--workload`, default `auto`):
  c1   BASELINE.json configs[1]: 1,000,000 Gaussians, 1920x1080, SH 3, bsz 1, ONE GPU.      <- auto at N = 1
  c2   configs[2] shape: 6,000,000 Gaussians sharded over the N ranks, 1920x1080, bsz 1: ONE image split into N
       row bands (Grendel's pixel partition), dynamic load balancing on.                      <- auto at N > 1
  c4   configs[4] shape: 40,000,000 Gaussians sharded over N ranks, 3840x2160, bsz 1, image split N ways
       (reported beside c2 in `extra_workloads` at N > 1 unless --no-extra).
  weak bsz = N cameras per step on the c1 scene sharded N ways (every rank renders a whole image).
At N > 1 the total work of c2 / c4 does not depend on N ("scaling": "strong"); every line also carries
`same_workload_1gpu` -- the SAME scene and camera on one GPU, measured in the same run on rank 0's device -- and
`speedup_vs_1gpu`, which is the pixel-partition scaling north_star asks about.

One JSON line is printed by rank 0; besides the contract fields it carries
  kernels      : every HIP kernel group of the step with HIP-event time (recorded on the launch stream) and the
                 algorithmic bytes OF THE SAME LAUNCHES (each launch's own N / P / D / Px).  The events are recorded in
                 an instrumented REPLAY of the timed steps (same cameras, same order, same sizes) right after the contract
                 region: sixteen event packets per iteration cost ~5 % of the step, so `value` is measured without
                 them and the replay's own step time is `kernels_region_ms_per_step`;
  roofline     : the dominant kernel against the 8 TB/s HBM peak; `traffic` (PMC HBM bytes per launch) and `valu`
                 (SQ counters) come from profiles/r02_pmc.json, which tools/pmc_collect.py derives from rocprofv3
                 --pmc passes over THIS command -- used only while its source hash matches the kernels being run;
  timing       : median / p10 / p90 ms per step over `--repeats` timed regions (value = the first region);
  cpu_baseline : oracle/gsraster_ref.c (the plain-C "port") timed on this box's host cores on a
                 bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import hashlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "grendel-gs_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

# hipGraph replays (--graph on): the HIP runtime's "graph packet capture" leaves a graph's pre-built packets stale once a
# few hundred ordinary launches have run between two replays -- the next replay faults (measured on ROCm 7.0.2,
# tools/probes/graph_bench_probe2.py).  The flag is read when libamdhip64 is loaded: before torch.
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling

WORKLOADS = {
    # name: (gaussians_total, width, height, bsz or None = world, description)
    "c1": (1_000_000, 1920, 1080, 1, "BASELINE configs[1]: Mip360-bicycle-sized scene on one GPU"),
    "c1_4k": (1_000_000, 3840, 2160, 1, "the configs[1] scene at 3840x2160 (north_star: 1080p and 4K)"),
    "c2": (6_000_000, 1920, 1080, 1, "BASELINE configs[2] shape: one 1080p image split into row bands over the ranks"),
    "c4": (40_000_000, 3840, 2160, 1, "BASELINE configs[4] shape: one 4K image split into row bands over the ranks"),
    "weak": (1_000_000, 1920, 1080, None, "bsz = N whole images per step on the configs[1] scene sharded N ways"),
}


def algorithmic_bytes(kernel, m):
    """SURVEY.md 8(d) algorithmic HBM bytes of ONE launch from that launch's own sizes (restated in DESIGN.md 6)"""
    coeffs = m.get("M", 16)
    in_g = 44 + 12 * coeffs  # xyz 12 + scale 12 + rot 16 + opacity 4 + SH 12/coefficient = 236 at 16
    if kernel == "preprocess_forward":  # batched launch: parameters read once, 44 B of outputs per camera
        return m["N"] * (in_g + 44 * m.get("B", 1))
    if kernel == "preprocess_backward":
        return m["N"] * (in_g + (44 + 36) * m.get("B", 1) + in_g)
    if kernel == "preprocess_backward_adam":  # K11's reads + the optimizer's moments read, parameters / moments written:
        return m["N"] * (in_g + (44 + 36) * m.get("B", 1) + 5 * in_g)  # the 2 x 236 B gradient round trip is gone
    if kernel == "binning":  # the reference's algorithm: 64-bit (tile | depth) keys, LSD passes of 8 bits
        key_bits = 32 + max(1, math.ceil(math.log2(max(m["tiles"], 2))))
        return m["P"] * 16 + m["D"] * 12 + m["D"] * 24 * math.ceil(key_bits / 8) + m["D"] * 8
    if kernel == "binning_own":  # what THIS build's binning moves by construction
        # prepare step: 64 B / Gaussian as in rounds 1-5 (K3, four depth passes, the offsets scan) + 4 (the segment offsets).
        # Row-major pipeline (round 6, from 5 M pairs on; R = row segments, measured per launch): 28 B / segment (sorted
        # word + index written 8, read by the scan 4, its offset written 4, all three read by the pair pass 12) + 4 B /
        # pair (its index, written once).  Two-pass pipelines (below that, or R unknown): 40 B / pair.
        if m.get("R") and m["D"] >= (5 << 20):
            return 68 * m["P"] + 28 * m["R"] + 4 * m["D"]
        return 64 * m["P"] + 40 * m["D"]
    if kernel == "composite_forward":
        return 40 * m["D"] + 20 * m["Px"]
    if kernel == "composite_backward":
        return 76 * m["D"] + 20 * m["Px"]
    if kernel == "l1_ssim_forward":   # image 12 + uint8 GT 3 + three derivative maps 36 per pixel
        return 51 * m["Px"]
    if kernel == "l1_ssim_backward":  # maps 36 + image 12 + GT 3 + gradient 12
        return 63 * m["Px"]
    if kernel == "adam":              # p, m, v read + written, g read: 28 B per element
        return 28 * m["numel"]
    return 0


def source_hash():
    """sha256 over the kernel sources: a profiles/rNN_pmc.json is only quoted for the build it was measured on"""
    h = hashlib.sha256()
    d = os.path.join(PKG, "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(W, H, n_total, seconds=12.0):
    """the C restatement (oracle/, "port") on a 1/16-area crop x 1/16 of the Gaussians, all host cores"""
    from oracle import cref as C
    import synthetic_scene as S

    w, h, n = W // 4, H // 4, n_total // 16
    g = S.make_gaussians(n, w, h, seed=0)
    cam = S.orbit_cameras(8, w, h)[0]
    kw = dict(viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, campos=cam.camera_center,
              W=w, H=h, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2), sh_degree=3)
    keys = ["means3D", "scales", "rotations", "shs", "opacities"]
    mask = torch.ones((h + 15) // 16, (w + 15) // 16, dtype=torch.bool)
    bg = torch.zeros(3)
    wgt = torch.rand(3, h, w, generator=torch.Generator().manual_seed(1))

    def one():
        m2, rgb, co, radii, depths, cov3D, clamped = C.preprocess_forward(*[g[k] for k in keys], **kw)
        pl, ranges, _ = C.bin_and_sort(m2, radii, depths, mask, w, h)
        img, fT, nc = C.render_forward(m2, co, rgb, mask, bg, w, h, pl, ranges)
        d2, dco, drgb = C.render_backward(m2, co, rgb, mask, bg, w, h, pl, ranges, fT, nc, wgt)
        C.preprocess_backward(g["means3D"], g["scales"], g["rotations"], g["shs"], radii, cov3D, clamped, d2, dco,
                              drgb, **kw)

    one()
    t0 = time.time()
    it = 0
    while time.time() - t0 < seconds or it < 3:
        one()
        it += 1
    dt = (time.time() - t0) / it
    return {"value": 1.0 / dt, "unit": "images/s", "cores": C.num_threads(), "kind": "port",
            "sample": f"{n} Gaussians, {w}x{h} (1/16 of the Gaussians on a 1/16-area image), rasterizer "
                      f"fwd+bwd only (no loss/optimizer), {it} iterations, oracle/gsraster_ref.c with OpenMP",
            "full_size_equiv": round(1.0 / (16.0 * dt), 4),
            "full_size_equiv_note": "images/s the same port would reach on the FULL workload if its cost scaled with "
                                    "Gaussians and pairs (x16): the figure to hold against `value`, not the sample's"}


def load_pmc(src):
    """-> (blob | None, note): the newest profiles/r*_pmc.json (tools/pmc_collect.py) measured on THIS kernel source"""
    import glob

    note = "no profiles/r*_pmc.json"
    for pp in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        try:
            blob = json.load(open(pp))
        except Exception as e:  # noqa: BLE001
            note = f"{os.path.basename(pp)} unreadable: {e}"
            continue
        if blob.get("source_hash") == src:
            return blob, f"{os.path.basename(pp)}: {blob.get('command', '')}"
        note = (f"{os.path.basename(pp)} was measured on source hash {blob.get('source_hash')}, this build is {src}: "
                f"not quoted")
    return None, note


class _SingleRankView:
    """run the W = 1 path inside a W > 1 process (the same-workload single-GPU leg): the process-global state the
    mirror reads is switched to a one-rank view and restored afterwards"""

    def __init__(self, utils):
        self.u = utils

    def __enter__(self):
        u = self.u
        self.saved = (u.GLOBAL_RANK, u.WORLD_SIZE, u.DEFAULT_GROUP, u.IN_NODE_GROUP)
        u.GLOBAL_RANK, u.WORLD_SIZE = 0, 1
        u.DEFAULT_GROUP = u.IN_NODE_GROUP = u.SingleGPUGroup()

    def __exit__(self, *exc):
        u = self.u
        u.GLOBAL_RANK, u.WORLD_SIZE, u.DEFAULT_GROUP, u.IN_NODE_GROUP = self.saved


def self_launch_command(n, argv):
    """argv of `python -m torch.distributed.run` that runs this file with one rank per GPU on this node (rendezvous on
    127.0.0.1: the container's hostname may not resolve)"""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


_LAUNCHER_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK",
                 "ROLE_WORLD_SIZE", "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "NCCL_ASYNC_ERROR_HANDLING",
                 "TORCH_NCCL_ASYNC_ERROR_HANDLING")


def run_guarded_leg(n, argv, timeout_s):
    """One more measurement leg as a CHILD launch of this file (python -m torch.distributed.run, one rank per GPU) with
    a wall-clock limit: -> the child's JSON line as a dict, or {"error": ...}.  A leg that hangs (an RCCL collective
    captured in a hipGraph has never run with two ranks) or dies costs one field of the line, not the line: the child
    runs in its own session and exactly that process group is killed on time-out."""
    import signal
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in _LAUNCHER_ENV and not k.startswith("TORCHELASTIC")}
    cmd = self_launch_command(n, argv)
    t0 = time.time()
    try:
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, start_new_session=True, text=True)
    except OSError as e:
        return {"error": f"could not launch the leg: {e}"}
    try:
        so, se = p.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)  # the session this function started, nothing else
        except ProcessLookupError:
            pass
        so, se = p.communicate()
        return {"error": f"timed out after {timeout_s} s", "stderr_tail": (se or "")[-600:]}
    for line in reversed((so or "").strip().splitlines()):
        if line.startswith("{"):
            try:
                d = json.loads(line)
                d["leg_wall_s"] = round(time.time() - t0, 1)
                return d
            except json.JSONDecodeError:
                break
    lines = [ln.strip() for ln in (se or "").splitlines()]
    # the rank's own complaint (an RCCL / HIP / Python error) says more than the launcher's summary of it
    why = [ln for ln in lines if any(k in ln for k in ("Duplicate GPU", "ncclInvalidUsage", "NCCL error", "HIP error",
                                                       "RuntimeError", "Memory access fault", "ValueError", "SystemExit"))]
    if not why:
        why = [ln for ln in lines if "Error" in ln or "error:" in ln or "fault" in ln]
    return {"error": f"exit code {p.returncode}, no JSON line" + (f"; {why[0][:300]}" if why else ""),
            "stderr_tail": (se or "")[-600:]}


def brief_leg(d):
    """the fields of a leg's line that the parent line carries"""
    if d is None or "error" in d:
        return d
    keep = ("value", "unit", "ms_per_step", "steps", "warmup", "timing", "graph", "speedup_vs_1gpu", "leg_wall_s")
    out = {k: d[k] for k in keep if k in d}
    out["balance_timing"] = d.get("config", {}).get("balance_timing")
    return out


def percentile(xs, q):
    xs = sorted(xs)
    k = (len(xs) - 1) * q
    lo, hi = int(math.floor(k)), int(math.ceil(k))
    return xs[lo] + (xs[hi] - xs[lo]) * (k - lo)


def densify_leg(a, model, opt, train_step, timed, state, steps, dgr):
    """configs[4] says "densification on" (densification.py:5-86 of the reference: statistics every
    iteration, densify_and_prune every 100 iterations between backward and optimizer step).  Three timed regions of the
    same `steps` steps: plain; + the per-iteration statistics (max_radii2D, add_densification_stats); + densify_and_prune
    every --densify-every-th step (a quantile threshold clones / splits ~2 % of the rows, opacity < 0.005 prunes).  What
    a densification EVENT costs the loop is (t3 - t2) / events: the row surgery on 59 floats + 2 x 59 moments per
    Gaussian, the new sizes' first pass through the caching allocator, the regrowth of the sort scratch when the pair
    count outgrows it, the re-keyed FusedAdam state -- amortised over the reference's interval of 100 iterations."""
    import densification_ops as D

    every = int(a.densify_every)
    dev = model._xyz.device
    model.optimizer = opt
    model.percent_dense = 0.01

    def fresh_stats():
        n = model._xyz.shape[0]
        model.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        model.denom = torch.zeros((n, 1), device=dev)
        model.max_radii2D = torch.zeros((n,), device=dev)
        model.sum_visible_count_in_one_batch = torch.zeros((n,), device=dev)
        model.send_to_gpui_cnt = None

    fresh_stats()
    ev = {"events": 0, "rows": [int(model._xyz.shape[0])], "mode": "stats", "each_ms": []}
    scratch0 = sum(int(b.numel()) for b in dgr._SORT_SCRATCH.values())

    def event_due():
        return ev["mode"] == "densify" and state["it"] % every == 0

    def hook(pkg):
        """pkg: inside the iteration, between backward and step (the reference's spot) -- the statistics (one captured
        launch when the iteration is a hipGraph) and, in the eager loop, the event.  None: called by train_step AFTER a
        graphed iteration (the event's host reads cannot be captured; the iteration's step is then applied before the
        surgery instead of being dropped for the re-created parameters -- one step's difference per event)"""
        with torch.no_grad():  # densification.py:13-25
            if pkg is not None:
                for k in range(len(pkg["batched_locally_preprocessed_radii"])):
                    D.update_densification_stats(model, pkg["batched_locally_preprocessed_mean2D"][k],
                                                 pkg["batched_locally_preprocessed_radii"][k])
            graph = state.get("graph")
            in_graph = graph is not None and graph.enabled
            if event_due() and ((pkg is None) == in_graph):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                parts = {}
                if in_graph:
                    # the graphs go FIRST: they hold the tensors the surgery is about to replace (and their private
                    # pools), and with those alive the new rows come from fresh hipMallocs -- measured 246 ms per event
                    # instead of 4-8 ms
                    graph.reset()
                    torch.cuda.synchronize()
                    parts["graph_reset"] = time.perf_counter() - t0
                gr = (model.xyz_gradient_accum / model.denom.clamp(min=1)).squeeze(1)
                # the threshold that selects ~2 % of the rows (the reference's is a constant of the real scene, 0.0002)
                thr = torch.kthvalue(gr[:4_000_000], max(int(0.98 * min(gr.numel(), 4_000_000)), 1)).values.item()
                parts["threshold"] = time.perf_counter() - t0 - parts.get("graph_reset", 0.0)
                t1 = time.perf_counter()
                D.densify_and_prune(model, max(thr, 1e-30), 0.005, 4.0, None)
                torch.cuda.synchronize()
                parts["densify_and_prune"] = time.perf_counter() - t1
                ev["each_ms"].append(round(1e3 * (time.perf_counter() - t0), 3))
                ev.setdefault("parts_ms", []).append({k: round(1e3 * v, 3) for k, v in parts.items()})
                ev["events"] += 1
                ev["rows"].append(int(model._xyz.shape[0]))

    t_plain = timed(train_step, steps)
    state["densify"] = hook
    if state.get("graph") is not None:
        state["graph"].reset()  # (the body has changed: graphs captured without the statistics launch must not replay)
    timed(train_step, min(steps, 5))  # (the statistics' temporaries are new to the caching allocator)
    t_stats = timed(train_step, steps)
    ev["mode"] = "densify"
    it0 = state["it"]
    state["it"] = 0  # events at steps every, 2 * every, ...
    n_dens = max(steps, 3 * every)
    t_dens = timed(train_step, n_dens)
    state["it"] = it0
    state["densify"] = None
    if state.get("graph") is not None:
        state["graph"].reset()
    scratch1 = sum(int(b.numel()) for b in dgr._SORT_SCRATCH.values())
    # what an event costs the LOOP: the region's excess over the same steps without events, per event; the first event pays
    # one-off costs (code objects of the torch kernels densification uses are loaded at first use), hence `steady`
    per_event = (t_dens / n_dens - t_stats / steps) * n_dens / max(ev["events"], 1)
    steady = sorted(ev["each_ms"][1:])[len(ev["each_ms"][1:]) // 2] * 1e-3 if len(ev["each_ms"]) > 1 else per_event
    return {"every": every, "steps_plain": steps, "steps_with_events": n_dens, "events": ev["events"],
            "gaussians": ev["rows"],
            "ms_per_step_plain": round(1e3 * t_plain / steps, 4),
            "ms_per_step_with_statistics": round(1e3 * t_stats / steps, 4),
            "ms_per_step_with_events": round(1e3 * t_dens / n_dens, 4),
            "ms_per_event_in_loop_mean": round(1e3 * per_event, 3),
            "ms_per_event_each": ev["each_ms"], "ms_per_event_parts": ev.get("parts_ms"),
            "ms_per_event_steady": round(1e3 * steady, 3),
            "ms_per_step_amortised_at_100": round(1e3 * (t_stats / steps + steady / 100.0), 4),
            "sort_scratch_bytes": [scratch0, scratch1],
            "note": "statistics = max_radii2D + add_densification_stats every iteration; event = densify_and_prune "
                    "(clone + split at the 98th percentile of the accumulated statistic, prune opacity < 0.005) between "
                    "backward and optimizer step, timed between two device synchronisations (each) and as the region's "
                    "excess (in_loop_mean); steady = median of the events after the first (which loads the code objects "
                    "of torch kernels at first use); amortised_at_100 = statistics + one steady event per 100 iterations, the "
                    "reference's densification_interval; the eager loop has no graph to re-capture (a GraphedIteration "
                    "re-captures once per event: its sizes are part of the capture)"}


def run_workload(a, name, world, rank, dev, steps, warmup, repeats, render_steps, single_view=False,
                 collect_kernels=True, two_kernel_leg=None):
    """-> dict of measurements of one workload on the current process group view (world ranks)"""
    import diff_gaussian_rasterization as dgr
    import synthetic_scene as S
    import utils.general_utils as utils
    from fused_optim import FusedAdam
    from gaussian_renderer import distributed_preprocess3dgs_and_all2all_final, render_final
    from gaussian_renderer import settle as settle_views
    from gaussian_renderer.loss_distribution import batched_loss_computation, load_camera_from_cpu_to_all_gpu
    dgr.release_workspaces()  # a previous workload's sort scratch (GBs at the 40 M / 4K shape) is not this one's
    torch.cuda.empty_cache()
    from gaussian_renderer.workload_division import (DivisionStrategyHistoryFinal, finish_strategy_final,
                                                     start_strategy_final)

    n_total, W, H, bsz, desc = WORKLOADS[name]
    n_total = a.gaussians or n_total
    W, H = a.width or W, a.height or H
    real_world = int(os.environ.get("WORLD_SIZE", 1))
    # (`weak`: one image per GPU of the run -- also in the same-workload leg on ONE GPU, which then steps over N images)
    bsz = a.bsz or (bsz if bsz is not None else max(1, real_world if single_view else world))
    utils.set_args(utils.default_args(bsz=bsz))
    utils.set_img_size(H, W)
    utils.set_cur_iter(1)

    if name.startswith("c1") and not a.device_scene:  # the headline scene: SURVEY.md 8(d) generator on the host, seed 0
        model = S.SyntheticGaussianModel(n_total, W, H, seed=0, rank=rank, world_size=world, device=dev,
                                         opacity_logit_mean=a.opacity_logit_mean, opacity_logit_std=a.opacity_logit_std)
        scene = "host generator, seed 0"
    else:  # large / sharded scenes are drawn on the device; shard r of R is a pure function of (seed, r, R)
        shards = range(real_world) if single_view else None
        model = S.SyntheticGaussianModel(n_total, W, H, seed=0, rank=rank, world_size=real_world if single_view else world,
                                         device=dev, on_device=True, shards=shards,
                                         opacity_logit_mean=a.opacity_logit_mean, opacity_logit_std=a.opacity_logit_std)
        scene = "device generator, seed 0, shard = f(seed, rank, world)"
    n_views = max(a.views, bsz)
    cameras = S.orbit_cameras(n_views, W, H, device=dev)
    for k, cam in enumerate(cameras):
        cam.original_image_backup = S.make_gt_image(W, H, seed=1 + k, device=dev)  # preloaded to HBM
    history = DivisionStrategyHistoryFinal(S.SyntheticDataset(cameras), world, rank)
    bg = torch.zeros(3, dtype=torch.float32, device=dev)
    pipe = type("Pipe", (), {"debug": False})()
    # scene/gaussian_model.py:292 settings; grad /= bsz (train_internal.py:319-324) folded into the update.  With
    # fuse_backward the projection backward K11 runs inside the optimizer kernel (fused_optim.py): same arithmetic, the
    # six parameter gradients never reach HBM; --no-fuse-backward measures the two-kernel form
    opt = FusedAdam(model.param_groups(), lr=0.0, eps=1e-15, fuse_backward=not a.no_fuse_backward,
                    grad_scale=1.0 / bsz)
    state = {"it": 0, "sizes": None}

    def batch():
        s = (state["it"] * bsz) % n_views
        state["it"] += 1
        return [cameras[(s + j) % n_views] for j in range(bsz)]

    host_phases = {} if os.environ.get("GSR_HOST_PHASES") == "1" else None  # diagnostics: host seconds per call site

    def _ph(name, t0):
        t1 = time.perf_counter()
        if host_phases is not None:
            e = host_phases.setdefault(name, [0.0, 0])
            e[0] += t1 - t0
            e[1] += 1
        return t1

    from diff_gaussian_rasterization import capturing as _dgr_capturing

    def iteration(cams, strategies, tasks, between=None):
        """GT staging .. optimizer step of one batch (train_internal.py:134-208, 316-329); `between` runs where the
        reference calls finish_strategy_final, between backward and step"""
        t = time.perf_counter()
        load_camera_from_cpu_to_all_gpu(cams, strategies, tasks)
        t = _ph("load_camera", t)
        pkg = distributed_preprocess3dgs_and_all2all_final(cams, model, pipe, bg, batched_strategies=strategies,
                                                           mode="train")
        t = _ph("preprocess_and_exchange", t)
        images, masks = render_final(pkg, strategies)
        t = _ph("render_final", t)
        stats = [ca["stats_collector"] for ca in pkg["batched_cuda_args"]]
        if _dgr_capturing() is None:  # (a capture's run of this body records no events: its dicts hold placeholders)
            state["stats"] = stats
        loss, _ = batched_loss_computation(images, cams, masks, strategies, stats)
        t = _ph("loss", t)
        loss.backward()
        t = _ph("backward", t)
        if between is not None:
            between(stats)
        t = _ph("finish_strategy", t)
        if state.get("densify") is not None:
            state["densify"](pkg)
        opt.step()
        t = _ph("optimizer_step", t)
        opt.zero_grad(set_to_none=True)
        for cam in cams:
            cam.original_image = None
        state["sizes"] = pkg["gpui_to_gpuj_imgk_size"]
        state["loss"] = loss
        _ph("tail", t)
        return loss

    # --graph: the iteration replayed as ONE hipGraph (graphed_step.py).  With live heuristics the replays carry device
    # timestamps in place of the eager ops' HIP events (GraphedIteration(timings=True)) and the row bands are device data,
    # so the balancer keeps moving the cut points under one graph; `--balance-every R` (rounds 4-5: every R-th iteration
    # an eager, timed one, the partition frozen in between) is still there
    from gaussian_renderer.workload_division import timings_have_consumer
    graphed = None
    if getattr(a, "graph", "off") == "on" and opt.fuse_backward:
        from graphed_step import GraphedIteration
        # timings: with live heuristics the replays carry device timestamps, so that the load balancer keeps running
        # on replayed iterations (round 6; before, a graph needed frozen heuristics or --balance-every probes)
        graphed = GraphedIteration(opt, iteration, timings=not utils.get_args().no_heuristics_update and
                                   os.environ.get("GSR_GRAPH_TIMINGS", "1") != "0")
    state["graph"] = graphed
    every = max(int(getattr(a, "balance_every", 0) or 0), 0)
    args_ns = utils.get_args()
    live_default = not args_ns.no_heuristics_update

    def train_step():
        cams = batch()
        utils.set_cur_iter(utils.get_cur_iter() + bsz)
        if graphed is not None and every and live_default:
            args_ns.no_heuristics_update = (state["it"] % every) != 1  # probe iterations keep the reference's mode
        strategies, tasks = start_strategy_final(cams, history)
        state["bands"] = [(s.gpu_ids, s.division_pos) for s in strategies]
        if graphed is not None and (graphed.timings or not timings_have_consumer()):
            state["stats"] = None
            graphed(cams, strategies, tasks)
            idle = {"forward_render_time": 0.0, "backward_render_time": 0.0, "forward_loss_time": 0.0}
            # a replay: its device timestamps (or zeros when nothing consumes timings); an iteration the wrapper ran
            # eagerly (warm-up, a repeated one): the events its ops recorded
            stats = graphed.last_stats or state.get("stats") or [dict(idle) for _ in cams]
            finish_strategy_final(cams, history, strategies, stats)
            if state.get("densify") is not None:
                state["densify"](None)  # (a densification event due now: after the replay, see densify_leg)
            return
        if graphed is not None:
            graphed.validate()
        iteration(cams, strategies, tasks, between=lambda stats: finish_strategy_final(cams, history, strategies, stats))

    def render_step():
        with torch.no_grad():
            cams = batch()
            strategies, tasks = start_strategy_final(cams, history)
            pkg = distributed_preprocess3dgs_and_all2all_final(cams, model, pipe, bg, batched_strategies=strategies,
                                                               mode="test")
            images, _ = render_final(pkg, strategies)
        return images

    def fence():
        if graphed is not None:
            graphed.validate()  # the replay in flight counted (or has been repeated eagerly) before the clock stops
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    host_enqueue = {}  # id(fn) -> seconds the host needed to ENQUEUE the n calls (before it waited for the device)

    def timed(fn, n):
        fence()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        host_enqueue[fn.__name__] = time.perf_counter() - t0
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt

    # Set-up for the LARGE / multi-rank workloads, before the warmup the contract counts: one untimed pass over the DISTINCT
    # cameras of the synthetic dataset.  A view's pair count sizes its buffers, and the first sight of a larger view costs
    # the caching allocator a hipMalloc (multi-GB at the 40 M / 4K shape) and the speculative tile sort a fall-back --
    # first-epoch effects that would dominate the few steps those legs run.  Not used for the headline workload c1
    # (measured there: no effect).  Reported as `setup.priming_steps`.
    # The step of a multi-rank run is host-bound: keep the cyclic garbage collector from walking the (large, static)
    # object graph of the imported libraries every few hundred allocations -- everything alive now moves to the
    # permanent generation (measured on one rank of a fake 8-rank world at bsz 8: 12.7 -> see profiles/r04_gc_freeze.txt)
    if os.environ.get("GSR_GC_FREEZE", "1") != "0":
        import gc

        gc.collect()
        gc.freeze()
    priming = -(-n_views // bsz) if (name != "c1" and not a.no_priming) else 0  # (c1_4k etc.: primed)
    for _ in range(priming):
        train_step()
    state["it"] = 0
    for _ in range(warmup):
        train_step()
    dgr.kernel_timer.reset()
    if host_phases is not None:
        host_phases.clear()  # (diagnostics: the timed region and what follows only, not the cold steps)
    it0 = state["it"]
    dt = timed(train_step, steps)  # THE timed region of the contract: exactly `steps` steps between two fences
    # Per-kernel HIP events: an instrumented REPLAY of the same steps (same cameras in the same order, hence the same
    # launches with the same sizes) right after the contract region.  Sixteen event packets per iteration cost ~5 % of
    # the step (measured: 1.40 against 1.32 ms), so they are kept out of the region `value` is computed from; the
    # replay's own step time is reported as `kernels_region_ms_per_step`.
    dt_instr = None
    launches = {}
    walked = (0, 0)
    if collect_kernels:
        state["it"] = it0
        if graphed is not None:  # per-kernel events need the eager launches
            graphed.validate()
            graphed.enabled = False
        dgr.composite_walked(reset=True)
        dgr.kernel_timer.enabled = True
        dt_instr = timed(train_step, steps)
        dgr.kernel_timer.enabled = False
        walked = dgr.composite_walked(reset=True)  # list entries K8 / K10 went through in exactly these launches
        if graphed is not None:
            graphed.enabled = graphed.stats["disabled"] is None
        torch.cuda.synchronize()
        launches = dgr.kernel_timer.launches()
        dgr.kernel_timer.reset()
    extra = [timed(train_step, steps) for _ in range(max(repeats - 1, 0))]
    per_step = [1e3 * x / steps for x in [dt] + extra]
    fused_steps = opt.fused_steps
    dt_unfused = None
    if two_kernel_leg is None:
        two_kernel_leg = repeats > 1
    if graphed is not None:
        graphed.validate()
        graphed.enabled = False  # the legs below (two-kernel step, forward-only render) are eager
    if opt.fuse_backward and two_kernel_leg:  # the same steps with K11 and Adam as two kernels (gradients through HBM)
        opt.set_fuse_backward(False)
        timed(train_step, min(steps, 5))  # the gradient tensors are new to the caching allocator
        dt_unfused = min(timed(train_step, steps), timed(train_step, steps))
        opt.set_fuse_backward(True)

    densification = None
    if getattr(a, "densify_every", 0) and world == 1:
        if graphed is not None:  # --graph on: the statistics are part of the replays, an event drops the graphs
            graphed.enabled = graphed.stats["disabled"] is None
            before = dict(graphed.stats)
        densification = densify_leg(a, model, opt, train_step, timed, state, steps, dgr)
        n_total = int(model._xyz.shape[0])
        if graphed is not None:
            graphed.validate()
            densification["graph"] = {k: (graphed.stats[k] - before[k] if isinstance(before[k], int) else graphed.stats[k])
                                      for k in before}
            graphed.enabled = False

    if host_phases:
        print("# host phases (us per call, from the timed region on): " + ", ".join(
            f"{k} {1e6 * v[0] / max(v[1], 1):.0f}" for k, v in host_phases.items()), file=sys.stderr, flush=True)
    lrs = {g.get("name", str(i)): g["lr"] for i, g in enumerate(opt.param_groups)}
    out = {"name": name, "desc": desc, "learning_rates": lrs, "gaussians_total": n_total, "gaussians_this_rank": int(model._xyz.shape[0]),
           "image": [W, H], "bsz": bsz, "world": world, "scene": scene, "dt": dt, "steps": steps,
           "ms_per_step": 1e3 * dt / steps, "images_per_s": bsz * steps / dt, "priming_steps": priming,
           "kernels_region_ms_per_step": (1e3 * dt_instr / steps) if dt_instr else None,
           "optimizer": {"fuse_backward": bool(opt.fuse_backward), "fused_steps": fused_steps,
                         "materialized_steps": opt.materialized_steps,
                         "ms_per_step_two_kernels": round(1e3 * dt_unfused / steps, 4) if dt_unfused else None,
                         "images_per_s_two_kernels": round(bsz * steps / dt_unfused, 3) if dt_unfused else None,
                         "note": "fuse_backward: K11 runs inside the optimizer kernel (same arithmetic bit for bit, "
                                 "tests/test_gpu_loss_and_step.py); *_two_kernels = the same steps with K11 and Adam "
                                 "as separate launches, best of two regions timed after the contract's regions"},
           "graph": (dict(graphed.stats) if graphed is not None else None),
           "densification": densification,
           "timing": {"repeats": len(per_step), "ms_per_step_median": round(percentile(per_step, 0.5), 4),
                      "ms_per_step_p10": round(percentile(per_step, 0.1), 4),
                      "ms_per_step_p90": round(percentile(per_step, 0.9), 4),
                      "ms_per_step_all": [round(x, 4) for x in per_step]}}
    if render_steps > 0:
        for _ in range(3):
            render_step()
        dt_r = timed(render_step, render_steps)
        out["rendered_views_per_sec_one_stream"] = bsz * render_steps / dt_r
        out["render_host_ms_per_view"] = 1e3 * host_enqueue.get("render_step", 0.0) / (bsz * render_steps)
        # `rendered_views_per_sec` IS the sequential loop on one stream -- the reference's render driver (render.py:87-95);
        # the two loops below are extensions of this build and are reported under their own keys only (advisor r05)
        out["rendered_views_per_sec"] = out["rendered_views_per_sec_one_stream"]
        if world == 1 and dev.type == "cuda" and os.environ.get("GSR_RENDER_STREAMS", "2") != "1":
            # Forward-only views are independent: consecutive views go to TWO streams alternately, so that a view's
            # binning (latency chains: VALU 0.4 busy, 0.2 of HBM) runs beside the previous view's composite (VALU-bound,
            # 0.07 of HBM).  Same kernels, same images.
            side = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
            turn = [0]

            def render_step_two_streams():
                st = side[turn[0] & 1]
                turn[0] += 1
                with torch.cuda.stream(st):
                    return render_step()

            # ... and PIPELINED (round 6): view i's pair count is settled after view i + 1 has been enqueued
            # (render_final(..., late=[...]) + gaussian_renderer.settle): the host never waits for a kernel it has just
            # launched.  A view's image is final once its count is settled -- one view later, like a render server that
            # sends frame i while frame i + 1 is being drawn.
            pending = [[], []]

            def render_step_pipelined():
                k = turn[0] & 1
                st = side[k]
                turn[0] += 1
                with torch.cuda.stream(st), torch.no_grad():
                    cams = batch()
                    strategies, tasks = start_strategy_final(cams, history)
                    pkg = distributed_preprocess3dgs_and_all2all_final(cams, model, pipe, bg, batched_strategies=strategies,
                                                                       mode="test")
                    images, _ = render_final(pkg, strategies, late=pending[k])
                settle_views(pending[k ^ 1])  # the previous view (other stream): its kernels have long started
                return images

            for st in side:
                st.wait_stream(torch.cuda.current_stream(dev))
            for _ in range(4):
                render_step_two_streams()
            dt_2 = timed(render_step_two_streams, render_steps)
            out["rendered_views_per_sec_two_streams"] = bsz * render_steps / dt_2
            out["render_host_ms_per_view_two_streams"] = 1e3 * host_enqueue.get("render_step_two_streams", 0.0) / (
                bsz * render_steps)
            for _ in range(4):
                render_step_pipelined()
            settle_views(pending[0]), settle_views(pending[1])
            dt_3 = timed(render_step_pipelined, render_steps)
            settle_views(pending[0]), settle_views(pending[1])
            out["rendered_views_per_sec_pipelined"] = bsz * render_steps / dt_3
            out["render_host_ms_per_view_pipelined"] = 1e3 * host_enqueue.get("render_step_pipelined", 0.0) / (
                bsz * render_steps)
            torch.cuda.current_stream(dev).wait_stream(side[0])
            torch.cuda.current_stream(dev).wait_stream(side[1])

    # ---- per-kernel: time and algorithmic bytes of the SAME launches
    kern = {}
    px_cache = {}
    for kname, recs in launches.items():
        tot_ms, tot_b, D_sum, px_sum = 0.0, 0, 0, 0
        for ms, meta in recs:
            m = dict(meta)
            if "mask" in m:
                key = (m["mask"].data_ptr(), m["W"], m["H"])
                if key not in px_cache:
                    px_cache[key] = dgr.local_pixels(m["mask"], m["W"], m["H"])
                m["Px"] = px_cache[key]
            tot_ms += ms
            tot_b += algorithmic_bytes(kname, m)
            D_sum += m.get("D", 0) or 0
            px_sum += m.get("Px", 0) or 0
        n = len(recs)
        k = kern[kname] = {"launches": n, "avg_ms": round(tot_ms / n, 5)}
        gbps = (tot_b / (tot_ms * 1e-3) / 1e9) if tot_ms > 0 and tot_b else None
        if D_sum:
            k["mean_pairs_D"] = D_sum // n
        if kname in ("composite_forward", "composite_backward"):
            # The D-based formula (40 / 76 B per pair + 20 B per pixel) credits the WHOLE tile lists;
            # early termination leaves most of every list unread, so it is no roofline (it exceeded 1.0 "of peak" on
            # the 6 M-Gaussian shape).  What the kernel can have asked memory for follows from the entries it WALKED
            # (counted by the kernels themselves, gsr_composite_walked): 40 B (K8) / 76 B (K10: + the 36-byte gradient
            # record) per walked entry of a tile + 20 B per pixel.  The kernels are bound by VALU issue (`bound`).
            w_entries = walked[0 if kname == "composite_forward" else 1] / n
            per = 40 if kname == "composite_forward" else 76
            wb = per * w_entries + 20 * px_sum / n
            k.update({"bound": "valu", "hbm_formula_MB": round(tot_b / n / 1e6, 3),
                      "hbm_formula_note": "whole-list formula of SURVEY.md 8(d), credits bytes the kernel never touches: "
                                          "no roofline",
                      "walked_entries": int(w_entries),
                      "walked_frac_of_D": round(w_entries / (D_sum / n), 4) if D_sum else None,
                      "hbm_walked_MB": round(wb / 1e6, 3),
                      "hbm_walked_frac": round(wb / (tot_ms / n * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tot_ms > 0 else None})
        elif kname == "binning":
            # bytes THIS build's split-key binning moves by construction; the reference's 64-bit-key
            # algorithm would move `reference_algorithm_MB` -- a speed-up statement, not a roofline
            own = sum(algorithmic_bytes("binning_own", dict(meta)) for _, meta in recs)
            R_sum = sum((meta.get("R") or 0) for _, meta in recs)
            k.update({"bound": "hbm / launch latency", "algo_MB": round(own / n / 1e6, 3),
                      "mean_row_segments_R": R_sum // n,
                      "algo_note": "row-major pipeline (>= 5 M pairs): 68 B x P + 28 B x R + 4 B x D; two-pass pipelines: "
                                   "64 B x P + 40 B x D -- what this build's binning moves by construction",
                      "algo_two_pass_MB": round(sum(64 * meta["P"] + 40 * (meta.get("D") or 0) for _, meta in recs) / n / 1e6, 3),
                      "GBps": round(own / (tot_ms * 1e-3) / 1e9, 1) if tot_ms > 0 else None,
                      "frac_hbm_peak": round(own / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tot_ms > 0 else None,
                      "reference_algorithm_MB": round(tot_b / n / 1e6, 3)})
        else:
            k.update({"bound": "hbm", "algo_MB": round(tot_b / n / 1e6, 3), "GBps": round(gbps, 1) if gbps else None})
            if gbps:
                k["frac_hbm_peak"] = round(gbps / HBM_PEAK_GBS, 4)
    out["kernels"] = kern
    if world > 1 and state["sizes"] is not None:
        sizes = state["sizes"]  # sizes[i][j][k]: rows rank i sends to rank j for camera k (last step)
        rows = [[sum(sizes[i][j]) for j in range(world)] for i in range(world)]
        import gaussian_renderer as gr_

        pl = gr_._planner(utils.DEFAULT_GROUP, world, len(sizes[0][0]))
        caps = pl.caps_list  # caps[i][j][k]: slab rows reserved for what rank i sends rank j of camera k (None: exact layout)
        slab_rows = ([sum(sum(caps[i][j]) for j in range(world) if j != i) for i in range(world)] if caps else None)
        useful = [sum(r) - r[i] for i, r in enumerate(rows)]
        out["exchange"] = {
            "slab_rows_sent_per_rank": slab_rows,
            "bytes_padded_fwd_per_rank": ([44 * (slab_rows[i] - useful[i]) for i in range(world)] if slab_rows else None),
            "bytes_padded_bwd_per_rank": ([36 * (slab_rows[i] - useful[i]) for i in range(world)] if slab_rows else None),
            "planner": {"capacity": "1.25 x the largest count of the last 64 iterations + 256 rows, rounded up to 256, per "
                                    "(source, destination, camera)", "slabs_per_rank": (world - 1) * len(sizes[0][0]),
                        "layouts": dict(gr_.exchange_stats)},
            "rows_sent_per_rank": [sum(r) - r[i] for i, r in enumerate(rows)],
            "bytes_fwd_per_rank": [44 * (sum(r) - r[i]) for i, r in enumerate(rows)],  # 11 floats per row
            "bytes_bwd_per_rank": [36 * (sum(r) - r[i]) for i, r in enumerate(rows)],  # 9 gradient floats back
            "rows_kept_local_per_rank": [rows[i][i] for i in range(world)],
            "bands_last_step": [[list(g), list(d)] for g, d in state.get("bands", [])],
            "note": "last step; all-to-all-v over xGMI, rank i -> rank j peer copies; local rows do not leave the GPU; "
                    "bytes_fwd / bytes_bwd are the USEFUL rows, bytes_padded_* the zero rows that fill the capacity slabs "
                    "on top of them (the wire carries both)"}
    opt.set_fuse_backward(False)
    del model, opt, cameras, history
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps; value = the first one")
    ap.add_argument("--workload", default="auto", choices=["auto"] + list(WORKLOADS))
    ap.add_argument("--gaussians", type=int, default=0, help="override the workload's Gaussian count")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--bsz", type=int, default=0, help="cameras per step (default: the workload's; independent of --gpus)")
    ap.add_argument("--views", type=int, default=8, help="distinct synthetic cameras to cycle through")
    ap.add_argument("--opacity-logit-mean", type=float, default=0.0, help="SURVEY 8(d) generator: N(0, 2^2)")
    ap.add_argument("--opacity-logit-std", type=float, default=2.0)
    ap.add_argument("--device-scene", action="store_true", help="draw the c1 scene on the device as well")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-priming", action="store_true", help="skip the untimed set-up pass over the distinct cameras")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra workloads (N = 1: 4K, low opacity, 6 M "
                                                                "Gaussians; N > 1: c4)")
    ap.add_argument("--balance-timing", default="exact", choices=["exact", "pipelined"],
                    help="N > 1 with live heuristics: how finish_strategy_final gets its timings.  `exact` (default, the "
                         "headline) is the reference's schedule: wait for the iteration's own events every step "
                         "(workload_division.py:944-998); `pipelined` uses the previous iteration's and is reported as a leg")
    ap.add_argument("--no-legs", action="store_true",
                    help="N > 1: skip the extra legs of the line (the same workload with pipelined balance timings, and "
                         "the iteration replayed as one hipGraph -- the latter as a guarded child launch)")
    ap.add_argument("--leg-child", action="store_true", help=argparse.SUPPRESS)  # this process IS a leg: no legs of its own
    ap.add_argument("--leg-timeout", type=int, default=180, help="wall-clock limit of a guarded leg (seconds)")
    ap.add_argument("--with-pipelined-leg", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-fuse-backward", action="store_true",
                    help="run K11 and Adam as two kernels (the parameter gradients go through HBM) instead of the fused "
                         "K11 + Adam launch")
    ap.add_argument("--graph", default="off", choices=["off", "on"],
                    help="replay the training iteration as ONE hipGraph (graphed_step.py); with live load-balancer "
                         "heuristics the replays carry device timestamps that feed finish_strategy_final (round 6: one "
                         "graph serves every band); results are the eager loop's (capacity overflows are repeated eagerly)")
    ap.add_argument("--balance-every", type=int, default=0,
                    help="with --graph on and live heuristics: every R-th iteration is an eager, timed one that feeds "
                         "the load balancer; the partition is frozen in between (0, default: every iteration feeds it -- "
                         "replays through their device timestamps)")
    ap.add_argument("--no-1gpu-leg", action="store_true", help="N > 1: skip the same-workload single-GPU leg")
    ap.add_argument("--render-steps", type=int, default=20, help="forward-only views/sec leg (untimed by driver)")
    ap.add_argument("--densify-every", type=int, default=0,
                    help="also time the steps with the densification statistics every iteration and densify_and_prune "
                         "every N-th (densification.py:5-86 of the reference): reported as `densification`")
    ap.add_argument("--pmc-calib", action="store_true", help="also run a 256 MiB streaming multiply (PMC calibration)")
    a = ap.parse_args()

    import utils.general_utils as utils

    world = int(os.environ.get("WORLD_SIZE", 1))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become the launcher the reference's README uses (torchrun --standalone
        # --nproc-per-node N, README.md:199-202) -- one process per GPU over RCCL; rank 0 prints the JSON line
        # -- as a SUPERVISOR: the eager leg and the graph leg are guarded child launches, so that whatever happens inside
        # them (RCCL refusing two ranks on one device, a captured collective that never returns) this process still
        # prints ONE JSON line, with the error of each leg in its place
        argv = [x for x in sys.argv[1:]]
        main_leg = run_guarded_leg(a.gpus, argv + ["--leg-child"] + ([] if a.no_legs else ["--with-pipelined-leg"]),
                                   a.leg_timeout * 2)
        out = main_leg if "error" not in main_leg else {
            "metric": "training iters/sec (fwd+bwd)", "value": None, "unit": "images/s", "n_gpus": a.gpus,
            "steps": a.steps, "warmup": a.warmup, "higher_is_better": True, "data": "synthetic", "dtype": "f32",
            "error": main_leg["error"], "stderr_tail": main_leg.get("stderr_tail")}
        if not a.no_legs and a.graph == "off":
            g = run_guarded_leg(a.gpus, argv + ["--leg-child", "--graph", "on", "--no-extra", "--no-1gpu-leg",
                                                "--repeats", "1", "--render-steps", "0"], a.leg_timeout)
            out["graph_leg"] = brief_leg(g)
        print(json.dumps(out), flush=True)
        sys.exit(0 if "error" not in main_leg else 1)
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if local_rank >= torch.cuda.device_count():  # several ranks sharing one device (tools/, tests): not a bench mode
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    utils.init_distributed(backend="nccl" if world > 1 else None)
    rank = utils.GLOBAL_RANK

    if world > 1:
        from gaussian_renderer.workload_division import set_balance_timing

        set_balance_timing(a.balance_timing)
    name = a.workload if a.workload != "auto" else ("c1" if world == 1 else "c2")
    main_res = run_workload(a, name, world, rank, dev, a.steps, a.warmup, a.repeats, a.render_steps)
    legs = {}
    legs_wanted = world > 1 and not a.no_legs and a.graph == "off" and (not a.leg_child or a.with_pipelined_leg)
    if legs_wanted and a.balance_timing == "exact":
        # the same workload with the previous iteration's timings feeding the load balancer (no wait for the iteration's
        # own events): not the reference's schedule, hence a leg and not the headline
        from gaussian_renderer.workload_division import set_balance_timing

        set_balance_timing("pipelined")
        try:
            r = run_workload(a, name, world, rank, dev, min(a.steps, 20), min(a.warmup, 3), 1, 0, collect_kernels=False)
            legs["pipelined"] = {"value": round(r["images_per_s"], 3), "unit": "images/s",
                                 "ms_per_step": round(r["ms_per_step"], 4), "steps": r["steps"], "timing": r["timing"],
                                 "balance_timing": "pipelined"}
        except Exception as e:  # noqa: BLE001  (every rank fails alike)
            legs["pipelined"] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
        set_balance_timing(a.balance_timing)

    def same_workload_1gpu(wname):
        """the same scene + camera on ONE GPU (rank 0's device); the other ranks wait at the barrier"""
        res = None
        if rank == 0:
            try:
                with _SingleRankView(utils):
                    res = run_workload(a, wname, 1, 0, dev, min(a.steps, 10), min(a.warmup, 3), 1, 0, single_view=True,
                                       collect_kernels=False)
            except Exception as e:  # noqa: BLE001  (the other ranks wait at the barrier below: never leave them there)
                res = {"error": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()
        dist.barrier()
        return res

    extras = []
    one_gpu = None
    extras_1gpu = []
    if world == 1 and a.workload == "auto" and not a.no_extra and not (a.gaussians or a.width or a.height or a.bsz):
        import copy

        # the same JSON line tells the rest of the story: north_star's 4K, the hard (low-opacity:
        # tile lists walked ~3x deeper) variant of the headline scene, and configs[2]'s 6 M Gaussians on one GPU
        variants = {"opacity_logit_mean": "low opacity: logit ~ N(-2, 1)",
                    "bsz": "bsz 4 on one GPU: Grendel's batched multi-view mode (configs[3]'s mode; train_internal.py:95-101)",
                    "densify_every": "densification in the loop (configs[4] says 'densification on'): see `densification`"}
        for wname, over in (("c1_4k", {}), ("c1", {"opacity_logit_mean": -2.0, "opacity_logit_std": 1.0}), ("c2", {}),
                            ("c1", {"bsz": 4}), ("c1", {"densify_every": 10})):
            b = copy.copy(a)
            for k_, v_ in over.items():
                setattr(b, k_, v_)
            try:
                r = run_workload(b, wname, 1, 0, dev, 10, 3, 1, 0 if over.get("densify_every") else 5,
                                 two_kernel_leg=not over.get("densify_every"))
                r["variant"] = next((v for k_, v in variants.items() if k_ in over), None)
            except Exception as e:  # noqa: BLE001
                r = {"name": wname, "error": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()
            extras_1gpu.append(r)
    if world > 1:
        if not a.no_1gpu_leg:
            one_gpu = same_workload_1gpu(name)
        if a.workload == "auto" and not a.no_extra:
            # configs[4]'s shape (strong scaling of one 4K image over a 40 M-Gaussian scene) and the weak-scaling leg:
            # bsz = N whole images per step on the configs[1] scene sharded N ways (Grendel's batched multi-view mode,
            # configs[3]) -- its `value` is directly comparable with the N = 1 line (same scene, one image per GPU)
            for wname, st in (("c4", max(a.steps // 3, 5)), ("weak", a.steps)):
                try:
                    ex = run_workload(a, wname, world, rank, dev, st, min(a.warmup, 3), 1, 0, collect_kernels=False)
                except Exception as e:  # noqa: BLE001  (every rank fails alike: sizes are a function of (N, workload))
                    ex = {"name": wname, "error": f"{type(e).__name__}: {e}"}
                    torch.cuda.empty_cache()
                ex1 = None if (a.no_1gpu_leg or "error" in ex) else same_workload_1gpu(wname)
                extras.append((ex, ex1))

    if a.pmc_calib:
        x = torch.rand(64 * 1024 * 1024, device=dev)  # 256 MiB
        y = torch.empty_like(x)
        for _ in range(3):
            torch.mul(x, 2.0, out=y)  # vectorised elementwise kernel: 256 MiB read (16 B/lane) + 256 MiB written
        torch.cuda.synchronize()

    if world > 1:
        dist.barrier()  # every rank is done measuring (rank 0 may launch a guarded leg on the same GPUs afterwards)
    if rank != 0:
        return

    def brief(res, res1):
        if "error" in res:
            return {"workload": res["name"], "error": res["error"]}
        d = {"workload": f"{res['name']}: {res['desc']}", "gaussians_total": res["gaussians_total"],
             "image": res["image"], "bsz": res["bsz"], "value": round(res["images_per_s"], 3), "unit": "images/s",
             "ms_per_step": round(res["ms_per_step"], 4), "steps": res["steps"], "timing": res["timing"]}
        if "exchange" in res:
            d["exchange"] = res["exchange"]
        if res1 is not None and "error" in res1:
            d["same_workload_1gpu"] = res1
        elif res1 is not None:
            d["same_workload_1gpu"] = {"value": round(res1["images_per_s"], 3), "unit": "images/s",
                                       "ms_per_step": round(res1["ms_per_step"], 4), "steps": res1["steps"],
                                       "note": "same scene and camera on ONE GPU, measured in this run on rank 0"}
            d["speedup_vs_1gpu"] = round(res["images_per_s"] / res1["images_per_s"], 3)
        return d

    kern = main_res["kernels"]
    dom = max(kern, key=lambda k: kern[k]["avg_ms"] * kern[k]["launches"]) if kern else None
    roofline = None
    src = source_hash()
    pmc, pmc_note = load_pmc(src)
    for kname, kv in kern.items():  # measured HBM bytes (PMC, separate rocprofv3 passes of this command) per kernel
        pk_ = (pmc or {}).get("kernels", {}).get(kname, {})
        if pk_.get("hbm_bytes_per_launch") and pk_.get("avg_ms"):
            kv["measured_MB"] = round(pk_["hbm_bytes_per_launch"] / 1e6, 3)
            kv["frac_measured"] = round(pk_["hbm_bytes_per_launch"] / (pk_["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if pk_.get("valu_busy_frac") is not None:
            kv["valu_busy"] = pk_.get("valu_busy_frac")
    if dom:
        dk = kern[dom]
        pk = (pmc or {}).get("kernels", {}).get(dom, {})
        valu = None
        if pk.get("SQ_INSTS_VALU"):
            # measured issue-pipe view: SQ_ACTIVE_INST_VALU (quad-cycles, summed over the SIMDs) against the SIMD cycles
            # of the launch (SQ_BUSY_CYCLES is summed over the 32 shader engines): the fraction of time the VALU pipes
            # were busy; a wave64 VALU instruction occupies its SIMD ~4.1-4.2 cycles on this chip (same counters)
            valu = {"insts_per_launch": pk["SQ_INSTS_VALU"], "active_quad_cycles_per_launch": pk.get("SQ_ACTIVE_INST_VALU"),
                    "busy_cycles_per_launch": pk.get("SQ_BUSY_CYCLES"), "profiled_avg_ms": pk.get("avg_ms"),
                    "frac": pk.get("valu_busy_frac"), "mfma_frac": pk.get("mfma_busy_frac"),
                    "cycles_per_inst": pk.get("cycles_per_valu_inst"),
                    "derivation": "frac = 4 * SQ_ACTIVE_INST_VALU / (SQ_BUSY_CYCLES / 32 * 1024 SIMDs); mfma_frac = "
                                  "SQ_VALU_MFMA_BUSY_CYCLES / the same cycles (fp32 MFMA and VALU do not co-issue)"}
        traffic = pk.get("hbm_bytes_per_launch")
        traffic_frac = (round(traffic / (pk["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                        if traffic and pk.get("avg_ms") else None)
        if dk.get("bound") == "valu":
            # the dominant kernel is bound by instruction issue, not by HBM: the roofline is the SIMDs' issue capacity
            # (VALU-busy + MFMA-busy fraction of the SIMD cycles, PMC of THIS build or null); the HBM views sit beside it
            issue = (round((pk.get("valu_busy_frac") or 0.0) + (pk.get("mfma_busy_frac") or 0.0), 4)
                     if pk.get("valu_busy_frac") is not None else None)
            roofline = {"kernel": dom, "bound": "valu", "achieved": issue, "peak": 1.0,
                        "unit": "fraction of SIMD issue cycles busy (VALU + fp32 MFMA)", "frac": issue,
                        "hbm_walked_frac": dk.get("hbm_walked_frac"), "hbm_walked_bytes": int(dk["hbm_walked_MB"] * 1e6),
                        "hbm_formula_bytes": int(dk["hbm_formula_MB"] * 1e6),
                        "hbm_formula_frac": round(dk["hbm_formula_MB"] * 1e6 / (dk["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "hbm_formula_note": dk["hbm_formula_note"], "hbm_peak_GBps": HBM_PEAK_GBS,
                        "traffic": traffic, "traffic_frac": traffic_frac, "avg_ms": dk["avg_ms"], "valu": valu,
                        "pmc_source": pmc_note, "source_hash": src,
                        "note": "K10 walks " + str(dk.get("walked_frac_of_D")) + " of the pair entries (early "
                                "termination); hbm_walked_frac = (76 B x walked entries + 20 B x pixels) / HIP-event "
                                "time / 8 TB/s, traffic = PMC FETCH_SIZE + WRITE_SIZE of the same command"}
        else:
            ach = dk.get("GBps")
            roofline = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4) if ach else None, "traffic": traffic,
                        "traffic_frac": traffic_frac, "avg_ms": dk["avg_ms"],
                        "algorithmic_bytes": int(dk["algo_MB"] * 1e6), "valu": valu, "pmc_source": pmc_note,
                        "source_hash": src,
                        "note": "achieved = algorithmic bytes of the launches / their HIP-event time, taken in the "
                                "instrumented replay of the timed steps (same launches, kernels_region_ms_per_step)"}
    n_total, W, H = main_res["gaussians_total"], main_res["image"][0], main_res["image"][1]
    out = {
        "metric": "training iters/sec (fwd+bwd)",
        "value": round(main_res["images_per_s"], 3),
        "unit": "images/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(main_res["ms_per_step"], 3),
        "higher_is_better": True,
        "scaling": "weak" if (name == "weak" or world == 1) else "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{name}: {main_res['desc']}; {n_total} Gaussians total, SH degree 3, {W}x{H}, "
                               f"bsz {main_res['bsz']}, full train iteration (activations, preprocess, exchange, render, "
                               f"L1+SSIM loss, backward, Adam)",
                   "gaussians_total": n_total, "gaussians_per_gpu": main_res["gaussians_this_rank"], "image": [W, H],
                   "bsz": main_res["bsz"],
                   "parallelism": f"pixel-partition x{world} (row bands), Gaussian-sharded x{world}",
                   "scene": main_res["scene"], "opacity_logit": [a.opacity_logit_mean, a.opacity_logit_std], "seed": 0,
                   "learning_rates": main_res["learning_rates"],
                   "learning_rates_note": "per parameter group, the reference's (arguments/__init__.py:109-118); the "
                                          "constructor default lr=0.0 (scene/gaussian_model.py:292) is overridden by "
                                          "every group, so the parameters DO move in the timed steps",
                   "optimizer": ("FusedAdam, K11 fused into the step (gsr_preprocess_backward_adam_raw_batched): same "
                                 "arithmetic as K11 + Adam, parameter gradients never written to HBM"
                                 if main_res["optimizer"]["fuse_backward"] else "FusedAdam after K11 (two kernels)")},
        "optimizer": main_res["optimizer"],
        "graph": main_res.get("graph"),
        **({"densification": main_res["densification"]} if main_res.get("densification") else {}),
        "timing": main_res["timing"],
        "setup": {"priming_steps": main_res["priming_steps"],
                  "note": "untimed pass over the distinct synthetic cameras before the W warmup steps (large / multi-rank "
                          "workloads only: allocator and sort-capacity first-epoch effects); 0 = not used"},
        "kernels_region_ms_per_step": (round(main_res["kernels_region_ms_per_step"], 4)
                                       if main_res.get("kernels_region_ms_per_step") else None),
        "rendered_views_per_sec": round(main_res.get("rendered_views_per_sec", 0.0), 3),
        "rendered_views": {"one_stream": round(main_res.get("rendered_views_per_sec_one_stream", 0.0), 3),
                           "two_streams": (round(main_res["rendered_views_per_sec_two_streams"], 3)
                                           if "rendered_views_per_sec_two_streams" in main_res else None),
                           "two_streams_pipelined": (round(main_res["rendered_views_per_sec_pipelined"], 3)
                                                     if "rendered_views_per_sec_pipelined" in main_res else None),
                           "host_ms_per_view": (round(main_res["render_host_ms_per_view"], 4)
                                                if "render_host_ms_per_view" in main_res else None),
                           "host_ms_per_view_two_streams": (round(main_res["render_host_ms_per_view_two_streams"], 4)
                                                            if "render_host_ms_per_view_two_streams" in main_res else None),
                           "host_ms_per_view_pipelined": (round(main_res["render_host_ms_per_view_pipelined"], 4)
                                                          if "render_host_ms_per_view_pipelined" in main_res else None),
                           "note": "rendered_views_per_sec = one_stream = the reference's sequential render loop "
                                   "(render.py:87-95), every image final before the next view starts.  Extensions of this "
                                   "build, own keys only: two_streams = consecutive views alternate between two HIP "
                                   "streams (binning of view k+1 beside the composite of view k); two_streams_pipelined = "
                                   "the same with view k's pair count settled after view k+1 has been enqueued "
                                   "(render_final(late=...) + gaussian_renderer.settle: an image is final one view later). "
                                   "host_ms_per_view* = wall time of the host's loop per view before the final fence "
                                   "(includes the time it waits for a pair count)"},
        "kernels": kern,
        "roofline": roofline,
        "reference_published": {"a100_bicycle_1gpu_images_per_s": 16.6, "note": "README.md:342 of the reference; "
                                "other hardware, real data at 1237x822 -- not comparable, hence vs_baseline null"},
    }
    if "exchange" in main_res:
        out["exchange"] = main_res["exchange"]
    if one_gpu is not None:
        b = brief(main_res, one_gpu)
        out["same_workload_1gpu"], out["speedup_vs_1gpu"] = b["same_workload_1gpu"], b.get("speedup_vs_1gpu")
    if extras:
        out["extra_workloads"] = [brief(ex, ex1) for ex, ex1 in extras]
    if extras_1gpu:
        ews = []
        for r in extras_1gpu:
            if "error" in r:
                ews.append({"workload": r["name"], "error": r["error"]})
                continue
            top = sorted(r["kernels"].items(), key=lambda kv: -kv[1]["avg_ms"] * kv[1]["launches"])[:3]
            ews.append({"workload": f"{r['name']}: {r['desc']}" + (f" [{r['variant']}]" if r.get("variant") else ""),
                        "gaussians_total": r["gaussians_total"], "image": r["image"], "bsz": r["bsz"],
                        "value": round(r["images_per_s"], 3), "unit": "images/s",
                        "ms_per_step": round(r["ms_per_step"], 4), "steps": r["steps"],
                        "rendered_views_per_sec": round(r.get("rendered_views_per_sec", 0.0), 3),
                        "ms_per_step_two_kernels": r["optimizer"]["ms_per_step_two_kernels"],
                        "dominant_kernels": {k_: {"avg_ms": v_["avg_ms"], "bound": v_.get("bound"),
                                                  **({"frac_hbm_peak": v_["frac_hbm_peak"]}
                                                     if "frac_hbm_peak" in v_ else {}),
                                                  **({"hbm_walked_frac": v_["hbm_walked_frac"]}
                                                     if "hbm_walked_frac" in v_ else {})} for k_, v_ in top},
                        **({"densification": r["densification"]} if r.get("densification") else {})})
        out["extra_workloads"] = ews
    if world > 1:
        out["config"]["balance_timing"] = a.balance_timing
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:  # noqa: BLE001
            rccl = f"unknown ({type(e).__name__})"
        out["distributed"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": rccl,
                              "devices_visible": torch.cuda.device_count(), "launcher": "torch.distributed.run"}
        import gaussian_renderer as gr

        out["exchange_layouts"] = dict(gr.exchange_stats)
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(W, H, n_total)
    if legs:
        out["legs"] = legs
    if world > 1 and not a.leg_child and not a.no_legs and a.graph == "off":
        # launched by the driver (torch.distributed.run ... bench.py --gpus N): the graph leg is a guarded CHILD launch on
        # the same GPUs once this job's ranks have left their process group -- an RCCL collective captured in a hipGraph
        # has never run with two ranks, and whatever it does must not cost this line
        try:
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
        torch.cuda.empty_cache()
        argv = [x for x in sys.argv[1:] if x not in ("--with-pipelined-leg",)]
        g = run_guarded_leg(world, argv + ["--leg-child", "--graph", "on", "--no-extra", "--no-1gpu-leg", "--repeats", "1",
                                           "--render-steps", "0"], a.leg_timeout)
        out["graph_leg"] = brief_leg(g)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
    sys.stdout.flush()
    if "graphed_step" in sys.modules and int(os.environ.get("WORLD_SIZE", 1)) > 1:
        # measured on RCCL 2.26 / ROCm 7.0 (tools/probes/rccl_capture_probe.py): destroy_process_group() after an
        # all-to-all has been captured in a hipGraph does not return.  The JSON line is out: leave without the teardown.
        os._exit(0)
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
