"""bench.py -- images/sec of one MUNIT training step (dis_update + gen_update) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 256] [--batch 8]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one synthetic batch: MUNIT_Trainer.update_learning_rate,
dis_update, gen_update (each consumes `batch` images of each domain, runs forward, backward, the
data-parallel gradient all-reduce when N > 1, and the Adam step).  Workload at N = 1 is BASELINE.json
configs[1]: config_256.yaml geometry (AdaINGen_double + MsImageDis, gen_state 1, guided 1,
recon_mask 1, aux losses 0), 256x256, batch 8, fp32.  N > 1 is weak scaling: every rank runs the same
per-GPU batch on its own shard of the global batch (data seed 7 + rank, identical model seed).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel, chosen by measured time: every convolution pass (forward, backward-data,
                  backward-weight) is bracketed by HIP events in extra instrumented steps of this same run; `frac` =
                  executed FLOPs / duration / fp32 MFMA peak (<= 1), `algorithmic_frac` next to it, a top-3 table, and the
                  step-level fractions on executed and on algorithmic FLOPs
  cpu_baseline -- the CPU oracle (oracle/munit_oracle.py, a torch-CPU restatement of the
                  reference step: kind "port") timed on this host's cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: dense bf16 MFMA (16x the fp32 matrix rate)
GFLOP_PER_PAIR_256 = 2789.6   # algorithmic conv+linear FLOPs of dis_update+gen_update


def bench_hp(size, batch, gen_state=1):
    """configs/config_256.yaml with the overrides of SURVEY.md section 8(d)."""
    return dict(
        batch_size=batch, weight_decay=1e-4, beta1=0.5, beta2=0.999, init="kaiming", lr=1e-4,
        lr_policy="step", step_size=100000, gamma=0.5, gan_w=3, recon_x_w=12, recon_s_w=1, recon_c_w=2,
        recon_x_cyc_w=12, vgg_w=0,
        adaptation=dict(full_adaptation=0, output_classifier_lambda=0, output_adv_lambda=0, output_classif_freq=1,
                        adv_lambda=0, dfeat_lambda=0, classif_frequency=15, sem_seg_lambda=0),
        semantic_w=0, recon_mask=1, domain_adv_w=0, recon_synth_w=0, gen_state=gen_state, guided=1,
        gen=dict(dim=64, mlp_dim=256, style_dim=16, activ="relu", n_downsample=2, n_res=4, pad_type="reflect"),
        dis=dict(dim=64, norm="none", activ="lrelu", n_layer=4, gan_type="lsgan", num_scales=3, pad_type="reflect"),
        ratio_disc_gen=5, input_dim_a=3, input_dim_b=3, display_size=8, optimizer="adam",
        crop_image_height=size, crop_image_width=size, new_size=size, num_workers=0)


def make_batch(batch, size, rank=0):
    """Synthetic two-domain batch: x = 2U - 1, mask = (U > 0.5); seed 7 + rank."""
    g = torch.Generator().manual_seed(7 + rank)
    x_a = 2 * torch.rand(batch, 3, size, size, generator=g) - 1
    x_b = 2 * torch.rand(batch, 3, size, size, generator=g) - 1
    m_a = (torch.rand(batch, 1, size, size, generator=g) > 0.5).float()
    m_b = (torch.rand(batch, 1, size, size, generator=g) > 0.5).float()
    return x_a, x_b, m_a, m_b


PMC_SUMMARY = "profiles/r04_pmc_hbm_mfma.txt"


def lib_fingerprint():
    """sha256 (first 16 hex digits) over the kernel sources (munit_amd/csrc/*.hip, *.h): ties a committed PMC summary to
    the code it was collected on (the .so itself is not in git; the public header holds declarations only)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "munit_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "munit_amd", "csrc", "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


class PmcSummary:
    """HBM bytes per launch from the committed PMC summary (tools/pmc_step.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in
    separate passes, FETCH doubled per MI355X_MICROARCH.md; bench.py cannot collect PMC itself).  The summary records the
    fingerprint of the kernel sources it was collected on; when the sources have changed since, the figures are stale and
    nothing is reported (`why` says so)."""

    def __init__(self):
        self.rows, self.why, self.source = [], None, PMC_SUMMARY
        try:
            lines = open(os.path.join(ROOT, PMC_SUMMARY)).read().splitlines()
        except OSError:
            self.why = "no PMC summary at %s" % PMC_SUMMARY
            return
        fp = [l.split()[-1] for l in lines if l.startswith("# kernel-source fingerprint:")]
        if not fp or fp[0] != lib_fingerprint():
            self.why = "%s was collected on kernel sources %s, this run is %s: stale, not reported" % (
                PMC_SUMMARY, fp[0] if fp else "unknown", lib_fingerprint())
            return
        import re
        pat = re.compile(r"^(.*?)\s+blocks=\s*(\d+)\s+launches=\s*(\d+)\s+fetch\s+([\d.]+) MB\s+write\s+([\d.]+) MB")
        for line in lines:
            m = pat.match(line)
            if m:
                self.rows.append((m.group(1).strip(), int(m.group(3)), float(m.group(4)) * 1e6, float(m.group(5)) * 1e6, int(m.group(2))))
        if not self.rows:
            self.why = "no kernel rows in %s" % PMC_SUMMARY

    def lookup(self, kname, launches_per_step):
        """(read bytes, written bytes) per launch of the kernel that leads the group `kname`, or None.  The PMC run holds 2
        steps; a summary row is keyed by (kernel, total grid size), so it is used only when its launch count says it holds
        exactly this group's launches (a row that mixes layer shapes is not attributed)."""
        lead = kname.split(" + ")[0].split(" x4")[0].split(" (")[0].strip()
        exact = [r for r in self.rows if r[0].startswith(lead[:40]) and r[1] == 2 * launches_per_step]
        if len(exact) == 1:
            return int(exact[0][2]), int(exact[0][3])
        return None


def cpu_quota():
    """CPUs the cgroup grants this process (None: unlimited).  The GPU boxes show 256 logical CPUs and grant 16."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else max(1, -(-int(quota) // int(period)))
    except (OSError, ValueError):
        return None


def usable_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup's CPU quota (the GPU boxes show 256 logical
    CPUs and grant 16) -- the same rule as tests/conftest.py.  Everything in this file that sizes a thread pool goes through
    here; os.cpu_count() counts what the box SHOWS."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    q = cpu_quota()
    return max(1, n if q is None else min(n, q))


def rank_threads(n_ranks):
    """Host threads per rank when n_ranks share this machine's CPU grant (OMP_NUM_THREADS of the self-launched ranks)."""
    return max(1, usable_cpus() // max(1, n_ranks))


def host_cores():
    """Threads of the CPU baseline: the physical cores of one socket, capped by what the process may
    use (affinity and cgroup quota) -- more threads than that only get throttled."""
    return max(1, min(_socket_cores(), usable_cpus()))


def _socket_cores():
    try:
        seen = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        sockets = len({p for p, _ in seen}) or 1
        if seen:
            return max(1, len(seen) // sockets)
    except OSError:
        pass
    return max(1, usable_cpus())


def cpu_baseline(size, seconds_budget=45.0):
    """The oracle's dis_update + gen_update on the host cores, batch 1: BASELINE.json configs[0] (128x128) and the bench
    resolution; 1 warm-up + >= 3 timed steps each.  `value` is the bench-resolution sample."""
    from oracle import munit_oracle as O
    from tests.parity import oracle_states
    cores = host_cores()
    prev = torch.get_num_threads()
    torch.set_num_threads(cores)

    def sample(sz, budget):
        hp = O.default_hp(sz, 1, 1)
        gen, dis_a, dis_b = oracle_states(hp, torch.float32)
        orc = O.OracleTrainer(hp, gen, dis_a, dis_b)
        x_a, x_b, m_a, m_b = O.synthetic_batch(1, sz, seed=7)

        def one():
            orc.update_learning_rate()
            orc.dis_update(x_a, x_b)
            orc.gen_update(x_a, x_b, m_a, m_b)

        one()  # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            one()
            n += 1
            el = time.perf_counter() - t0
            if n >= 3 and (el + el / n > budget or n >= 8):
                break
        return n / (time.perf_counter() - t0), n

    try:
        v128, n128 = sample(128, seconds_budget / 4)
        v, n = (v128, n128) if size == 128 else sample(size, seconds_budget)
    finally:
        torch.set_num_threads(prev)
    return {"value": round(v, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "oracle/munit_oracle.py (torch %s CPU fp32, %d threads = physical cores of one socket capped by the cgroup CPU quota), %dx%d batch "
                      "1, 1 warm-up + %d timed dis_update+gen_update steps" % (torch.__version__, cores, size, size, n),
            "config0_128x128_b1": {"value": round(v128, 4), "unit": "images/s", "timed_steps": n128}}


def time_configuration(dev, size, batch, precision, peak_tflops, warmup=3, steps=5):
    """One of BASELINE.json's other single-GPU configurations, timed the way the metric is (update_learning_rate + dis_update +
    gen_update on a resident synthetic batch, wall clock between device synchronisations) on a fresh trainer."""
    from munit_amd.trainer import MUNIT_Trainer
    hp = bench_hp(size, batch)
    hp["precision"] = precision
    torch.manual_seed(1234)
    tr = MUNIT_Trainer(hp)
    tr.to(dev)
    x_a, x_b, m_a, m_b = (t.to(dev) for t in make_batch(batch, size))

    def step():
        tr.update_learning_rate()
        tr.dis_update(x_a, x_b, hp)
        tr.gen_update(x_a, x_b, hp, m_a, m_b)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    loss = float(tr.loss_gen_total)
    assert loss == loss, "loss is NaN"
    step_flop = GFLOP_PER_PAIR_256 * 1e9 * (size / 256.0) ** 2 * batch
    return {"workload": "%dx%d, batch %d, %s" % (size, size, batch, precision), "ms_per_step": round(1e3 * dt, 3),
            "images_per_s": round(batch / dt, 3), "steps": steps, "warmup": warmup, "loss_gen_total": round(loss, 5),
            "roofline": {"bound": "mfma", "peak": peak_tflops, "unit": "TFLOP/s",
                         "achieved": round(step_flop / dt / 1e12, 2), "frac": round(step_flop / dt / 1e12 / peak_tflops, 4),
                         "pricing": "algorithmic conv+linear FLOPs of the step (SURVEY.md 8d) / step time",
                         "step_algorithmic_tflop": round(step_flop / 1e12, 3)},
            "note": "supplementary; not the reported metric"}


def self_launch(n):
    """`python bench.py --gpus N` from a bare shell: run the N ranks as children of this (GPU-free) process through
    torch.distributed.run, relay rank 0's JSON line on stdout (everything else goes to stderr) and return non-zero
    if any rank failed or no line was produced."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(rank_threads(n)))      # the CPUs the cgroup grants, shared by the ranks
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    line_out = None
    for line in proc.stdout:
        if line.startswith("{") and '"metric"' in line:
            line_out = line.rstrip("\n")
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc != 0:
        sys.stderr.write("bench.py: a rank failed (torch.distributed.run exit code %d)\n" % rc)
        return rc
    if line_out is None:
        sys.stderr.write("bench.py: the ranks finished without a result line\n")
        return 1
    print(line_out, flush=True)
    return 0


def dry_run(args, world, rank):
    """--dry-run: rendezvous + one all-reduce on the CPU (gloo), no GPU work.  Checks the launch path (self-launch,
    env contract, rank-0 JSON relay) on machines without a GPU; `value` is null."""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        assert float(t) == world * (world + 1) / 2, float(t)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "dry-run (no GPU work)", "value": None, "n_gpus": world, "steps": 0, "warmup": 0,
                          "config": {"workload": "launcher check", "parallelism": "dp%d" % world}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (pairs per update)")
    ap.add_argument("--gen-state", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dry-run", action="store_true", help="launcher check only: gloo rendezvous on the CPU, no GPU work")
    ap.add_argument("--no-modes", action="store_true", help="skip the supplementary timing of the opt-in reuse_dis_forward mode")
    ap.add_argument("--precision", choices=["f32", "bf16", "bf16s", "f32x3"], default="f32",
                    help="f32 = the reference's arithmetic (the headline metric, BASELINE.json configs[1]); "
                         "bf16s = configs[2] (use with --batch 32): bf16 storage of the trunk activations + bf16 MFMA, "
                         "fp32 accumulate / statistics / weights; bf16 = bf16 MFMA operands only, fp32 storage")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: start the N ranks ourselves.  Nothing in this process has touched the GPU yet
        # (importing torch does not), and the ranks are CHILD processes -- never an exec of a GPU-initialised process.
        raise SystemExit(self_launch(args.gpus))
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d does not match WORLD_SIZE=%d of the launcher" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, world, rank)
    import torch.distributed as dist
    use_dist = "WORLD_SIZE" in os.environ       # under a launcher the RCCL group is built even for one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if use_dist:
        torch.set_num_threads(rank_threads(world))      # host threads: this rank's share of the CPUs the cgroup grants

    from munit_amd import _lib, ops
    _lib.load()  # fail loudly before anything else if the HIP library is missing
    from munit_amd.trainer import MUNIT_Trainer

    hp = bench_hp(args.size, args.batch, args.gen_state)
    hp["precision"] = args.precision
    torch.manual_seed(1234)
    trainer = MUNIT_Trainer(hp)
    trainer.to(dev)
    x_a, x_b, m_a, m_b = (t.to(dev) for t in make_batch(args.batch, args.size, rank))

    def step():
        trainer.update_learning_rate()
        trainer.dis_update(x_a, x_b, hp)
        trainer.gen_update(x_a, x_b, hp, m_a, m_b)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.steps):
        step()
        marks[k + 1].record()        # on the caller's stream, which every update re-joins before it returns
    fence()
    elapsed = time.perf_counter() - t0
    per_step_ms = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    loss_total = float(trainer.loss_gen_total)
    assert loss_total == loss_total, "loss is NaN"

    ms_per_step = 1e3 * elapsed / args.steps
    value = args.batch * world * args.steps / elapsed
    mode_txt = {"f32": "fp32", "bf16": "bf16 MFMA operands / fp32 accumulate and storage",
                "bf16s": "bf16 storage of the content-encoder / decoder activations + bf16 MFMA / fp32 accumulate, statistics, "
                         "weights, style encoder and discriminators",
                "f32x3": "fp32 via exact 3-way bf16 split (6 product terms), fp32 accumulate"}[args.precision]
    out = {
        "metric": "images/sec (gen_update+dis_update) @%dx%d bs=%d" % (args.size, args.size, args.batch),
        "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": "config_256.yaml AdaINGen_double+MsImageDis dis_update+gen_update, %dx%d, "
                               "per-GPU batch %d, %s" % (args.size, args.size, args.batch, mode_txt),
                   "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                   "gen_state": args.gen_state, "loss_gen_total": round(loss_total, 5)},
        "timing": {"method": "value = batch * n_gpus * steps / wall clock (max over ranks) between device-synchronised "
                             "barriers, i.e. the MEAN step; ms_per_step_median = median of per-step HIP-event intervals "
                             "on rank 0's stream in the same region",
                   "ms_per_step_median": round(per_step_ms[len(per_step_ms) // 2], 3),
                   "ms_per_step_min": round(per_step_ms[0], 3), "ms_per_step_max": round(per_step_ms[-1], 3)},
    }

    if not args.no_roofline and args.precision == "f32":
        # Every rank runs these extra steps (they contain the gradient all-reduces, so the collectives must stay
        # matched across ranks); rank 0 reports.
        step_flop = GFLOP_PER_PAIR_256 * 1e9 * (args.size / 256.0) ** 2 * args.batch
        # single-kernel timing: the instrumented steps run on one stream (in the timed region other kernels share
        # the chip with every launch, which stretches each launch's own duration while shortening the step)
        from munit_amd import trainer as trainer_mod
        saved = (ops.SIDE_STREAM_WGRAD, trainer_mod.BRANCH_STREAMS)
        ops.SIDE_STREAM_WGRAD = trainer_mod.BRANCH_STREAMS = False
        ops.FLOPS = {"alg": 0.0, "exec": 0.0}
        step()                                   # also counts the FLOPs of one step, algorithmic and executed
        torch.cuda.synchronize()
        flops, ops.FLOPS = ops.FLOPS, None
        n_prof = 2
        ops.PROFILE = []
        for _ in range(n_prof):
            step()
        torch.cuda.synchronize()
        recs, ops.PROFILE = ops.PROFILE, None
        ops.SIDE_STREAM_WGRAD, trainer_mod.BRANCH_STREAMS = saved
        # every convolution pass (forward / backward-data / backward-weight) of the two steps was bracketed by HIP events:
        # group by (pass, kernel, layer geometry) and rank by measured time -- the dominant kernel is whatever comes out on top
        groups = {}
        for (which, pl, e0, e1) in recs:
            g = groups.setdefault((which, pl.kname[which], pl.layer), {"n": 0, "ms": 0.0, "alg": 0.0, "exe": 0.0, "bytes": 0.0})
            g["n"] += 1
            g["ms"] += e0.elapsed_time(e1)
            g["alg"] += pl.flop
            g["exe"] += pl.flop_exec[which]
            g["bytes"] += pl.bytes_alg[which]
        ranked = sorted(groups.items(), key=lambda kv: -kv[1]["ms"])
        if ranked and rank == 0:
            pmc = PmcSummary() if (args.size, args.batch) == (256, 8) else None

            def row(key, g):
                which, kname, layer = key
                sec_ = g["ms"] * 1e-3
                r = {"kernel": kname, "pass": ("forward", "backward-data", "backward-weight")[which], "layer": layer,
                     "launches_per_step": g["n"] // n_prof, "avg_launch_us": round(1e3 * g["ms"] / g["n"], 2),
                     "ms_per_step": round(g["ms"] / n_prof, 3),
                     "executed_frac": round(g["exe"] / sec_ / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                     "algorithmic_frac": round(g["alg"] / sec_ / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                     "traffic_algorithmic": int(g["bytes"] / g["n"])}
                t = pmc.lookup(kname, g["n"] // n_prof) if pmc is not None else None
                r["traffic"] = None if t is None else t[0] + t[1]
                r["traffic_over_algorithmic"] = None if t is None else round((t[0] + t[1]) / r["traffic_algorithmic"], 2)
                if t is not None:
                    r["traffic_read_write"] = [t[0], t[1]]
                return r

            top = [row(k, g) for k, g in ranked[:3]]
            key, g = ranked[0]
            d = top[0]
            sec = ms_per_step * 1e-3
            exe_tf = g["exe"] / (g["ms"] * 1e-3) / 1e12
            alg_tf = g["alg"] / (g["ms"] * 1e-3) / 1e12
            traffic_src = (pmc.source if pmc is not None and d["traffic"] is not None else
                           (pmc.why if pmc is not None else "the committed PMC summary is of the 256x256 batch-8 workload"))
            out["roofline"] = {
                # `achieved` / `frac` price the dominant kernel on the FLOPs it ISSUES on the fp32 matrix pipe, so frac <= 1 is
                # a pipe utilisation; the ALGORITHMIC pricing (2*B*Ho*Wo*Cout*KH*KW*Cin per launch) stands next to
                # it: Winograd kernels issue 16 of every 36 algorithmic multiply-accumulates, so algorithmic_frac can exceed 1
                "bound": "mfma", "achieved": round(exe_tf, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(exe_tf / PEAK_F32_MFMA_TFLOPS, 4),
                "algorithmic_achieved": round(alg_tf, 2), "algorithmic_frac": round(alg_tf / PEAK_F32_MFMA_TFLOPS, 4),
                "traffic": d["traffic"], "traffic_algorithmic": d["traffic_algorithmic"],
                "traffic_over_algorithmic": d["traffic_over_algorithmic"], "traffic_source": traffic_src,
                "kernel": d["kernel"], "pass": d["pass"], "layer": d["layer"],
                "selection": "largest total measured time among all (pass, kernel, layer) groups of the instrumented steps",
                "launches_per_step": d["launches_per_step"], "avg_launch_us": d["avg_launch_us"], "ms_per_step": d["ms_per_step"],
                "algorithmic_gflop_per_launch_avg": round(g["alg"] / g["n"] / 1e9, 3),
                "executed_gflop_per_launch_avg": round(g["exe"] / g["n"] / 1e9, 3),
                "method": "HIP events on the launch stream around every convolution pass (forward, backward-data, backward-weight) "
                          "in %d extra instrumented steps after the timed region, run on a single stream (the timed region "
                          "overlaps kernels on 3 streams); traffic = rocprofv3 --pmc FETCH_SIZE (doubled, gfx950) + WRITE_SIZE "
                          "per launch from the committed summary when its kernel-source fingerprint matches this build; "
                          "traffic_algorithmic = input + weight + output tensors once" % n_prof,
                "top": top,
                # step level: executed FLOPs of the whole step / step time / peak, and the algorithmic figure next to it
                "step_frac": round(flops["exec"] / sec / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                "step_achieved": round(flops["exec"] / sec / 1e12, 2),
                "step_executed_tflop": round(flops["exec"] / 1e12, 3),
                "step_algorithmic_frac": round(step_flop / sec / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                "step_algorithmic_achieved": round(step_flop / sec / 1e12, 2),
                "step_algorithmic_tflop": round(step_flop / 1e12, 3),
                # the same step counted launch by launch (cross-check of SURVEY.md 8d's analytic figure)
                "step_counted_algorithmic_tflop": round(flops["alg"] / 1e12, 3),
                "conv_ms_per_step_single_stream": round(sum(g_["ms"] for g_ in groups.values()) / n_prof, 2),
            }
    if rank == 0 and args.precision in ("bf16", "bf16s") and not args.no_roofline:
        # build extension (no reference counterpart): step-level fraction against the dense bf16 MFMA peak and, for
        # comparison with the headline, against the fp32 MFMA peak ("report vs both")
        step_flop = GFLOP_PER_PAIR_256 * 1e9 * (args.size / 256.0) ** 2 * args.batch
        sec = ms_per_step * 1e-3
        out["roofline"] = {"bound": "mfma", "achieved": round(step_flop / sec / 1e12, 2), "peak": PEAK_BF16_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": round(step_flop / sec / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4), "traffic": None,
                           "kernel": "whole step (algorithmic conv+linear FLOPs / step time)",
                           "step_frac_vs_f32_mfma_peak": round(step_flop / sec / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                           "step_algorithmic_tflop": round(step_flop / 1e12, 3)}
    if rank == 0 and world == 1 and args.precision == "f32" and not args.no_modes:
        del trainer
        torch.cuda.empty_cache()
        out["other_modes"] = {}
        # supplementary, NOT the metric: fp32 with `reuse_dis_forward: 1` -- gen_update continues from the generator forward
        # that dis_update ran on the same batch instead of recomputing it as the reference does (same losses bit for
        # bit, gradients to fp32 summation order, 11 % fewer multiply-accumulates)
        hp3 = dict(hp)
        hp3["reuse_dis_forward"] = 1
        torch.manual_seed(1234)
        tr3 = MUNIT_Trainer(hp3)
        tr3.to(dev)

        def step3():
            tr3.update_learning_rate()
            tr3.dis_update(x_a, x_b, hp3)
            tr3.gen_update(x_a, x_b, hp3, m_a, m_b)
        for _ in range(2):
            step3()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            step3()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        out["other_modes"]["reuse_dis_forward"] = {"ms_per_step": round(1e3 * dt, 3), "images_per_s": round(args.batch / dt, 3),
                                                   "note": "opt-in; not the reported metric"}
        del tr3
        if (args.size, args.batch, args.gen_state) == (256, 8, 1):
            # The two other single-GPU configurations of BASELINE.json, timed in the same process after the metric's own timed
            # region (3 warm-up + 5 timed steps each, wall clock between device synchronisations; never `value`): configs[3] =
            # config_HD.yaml 512x512 batch 4 fp32, configs[2] = 256x256 batch 32 with bf16 storage + bf16 MFMA.  `step_frac` =
            # SURVEY.md 8d's algorithmic FLOPs of the step / step time / the dense MFMA peak of the mode's arithmetic.
            torch.cuda.empty_cache()
            for name, size_m, batch_m, prec_m, peak in (("config_hd_512_b4", 512, 4, "f32", PEAK_F32_MFMA_TFLOPS),
                                                        ("bf16s_256_b32", 256, 32, "bf16s", PEAK_BF16_MFMA_TFLOPS)):
                out["other_modes"][name] = time_configuration(dev, size_m, batch_m, prec_m, peak)
                torch.cuda.empty_cache()
            ops.set_compute("f32")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.size)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
