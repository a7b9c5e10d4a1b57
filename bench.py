#!/usr/bin/env python3
"""Headline benchmark: Deep-TICA training frames/sec on a synthetic 10M x 512 feature matrix
(BASELINE.json metric; SURVEY.md section 8d / BASELINE.md config C4: MLP 512-256-128-4, lag 10,
global batch 8192 pairs), frame-sharded over N GPUs.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one optimiser step over one global batch of time-lagged pairs.  The matrix is
generated on the device, standardised by the HIP statistics / normalise kernels and stays
resident in HBM; the timed region holds exactly K training steps (plus the validation pass at
every epoch boundary they cross, as the reference's fit loop does) between barriers.  Rank 0
prints one JSON line.  `value` is the contract configuration (global batch 8192); the same fit at
a large global batch (524 208 pairs: MFMA-bound instead of latency-bound) is reported beside it
under `large_batch`, with its own roofline."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, FP32-input MFMA (dense)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide, BF16 MFMA (dense); the split arithmetic issues 6 bf16 MFMA flops per fp32 flop


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2000)
    p.add_argument("--warmup", type=int, default=100)
    p.add_argument("--frames", type=int, default=10_000_000)
    p.add_argument("--features", type=int, default=512)
    p.add_argument("--hidden", type=str, default="256,128")
    p.add_argument("--dim", type=int, default=4)
    p.add_argument("--lag", type=int, default=10)
    p.add_argument("--batch", type=int, default=8192, help="global batch (pairs per optimiser step) of the headline: SURVEY 8d / BASELINE.md C4")
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=15.0)
    p.add_argument("--profile-every", type=int, default=0,
                   help="HIP events (stamped with the launch's own begin / end) on the layer-0 products of every n-th timed step, forward "
                        "product and weight gradient on different steps; 0 (default) = every 4th step when --steps <= 100, every 8th otherwise; "
                        "1 = both products of every step (costs ~15 us per step).  The first timed step -- the one right behind the "
                        "barrier: idle clocks, cold L2 -- is never sampled")
    p.add_argument("--large-batch", type=int, default=524208,
                   help="second measurement ('large_batch'): 8 x (65536 - lag) pairs -- every rank's step covers whole 128-row "
                        "tiles at 1/2/4/8 GPUs when the rows of x_t and x_lag are shared; 0 disables")
    p.add_argument("--large-steps", type=int, default=60)
    p.add_argument("--gemm-mode", choices=["split", "native"], default="split",
                   help="arithmetic of the MLP products: 'split' = FP32-accurate split products on the BF16 matrix pipe "
                        "(library default), 'native' = FP32-input MFMA; the other mode is timed too (field 'other_gemm_mode')")
    p.add_argument("--other-mode-steps", type=int, default=200)
    p.add_argument("--backend", default="nccl", help="torch.distributed backend of a multi-rank run: 'nccl' (= RCCL over xGMI; default) or "
                                                    "'gloo' (rehearsals of the N > 1 control flow with several ranks on one GPU)")
    p.add_argument("--native-rccl", action="store_true",
                   help="data-parallel steps through the library's own RCCL communicator (dcv_comm_*: forward, all-reduces, backward and "
                        "update inside one C call) instead of torch.distributed's; torch.distributed still provides the launcher's "
                        "rendezvous, the barriers and the unique-id broadcast")
    p.add_argument("--shuffled-steps", type=int, default=200,
                   help="third measurement ('shuffled'): the reference's DEFAULT loader -- random_split + a fresh permutation per epoch "
                        "(yaml_schemas/train_colvars.py:55-56) -- at the headline batch: gathered two-half evaluation, no row sharing; 0 disables")
    p.add_argument("--c2-steps", type=int, default=400,
                   help="one GPU only: append the bounded BASELINE.json configs[1] run (autoencoder on 1M x 128, batch 4096) as block 'c2'; 0 disables")
    p.add_argument("--ref-small-steps", type=int, default=300,
                   help="one GPU only: append block 'ref_small' -- Deep-TICA on the reference's own network sizes (54-16-8-2 of its test "
                        "configuration, [15, 15] of default_config.yml) at batch 128 and 4096; 0 disables")
    p.add_argument("--fit-epochs", type=int, default=10,
                   help="one GPU only: append block 'calculator_fit' -- wall clock of CVCalculator.train() (the WHOLE host path of a fit: split, "
                        "initialisation, per-epoch permutation, steps, validation, log read-back, metrics, checkpoints) for Deep-TICA and the "
                        "autoencoder on 200 000 x 54 features with the reference's default architecture and loader; 0 disables")
    p.add_argument("--config", choices=["c4", "c2", "ref_small"], default="c4",
                   help="c4 (default): the headline, Deep-TICA on 10M x 512; c2: BASELINE.json configs[1], autoencoder 128-64-32-2-32-64-128 on "
                        "1M x 128 at batch 4096 (its own metric line with roofline and cpu_baseline; one GPU)")
    return p.parse_args()


def init_linears(dims, seed):
    """torch.nn.Linear default initialisation in construction order (what create_model() does)."""
    torch.manual_seed(seed)
    lins = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)]
    return [(l.weight.detach().numpy().copy(), l.bias.detach().numpy().copy()) for l in lins]


def settle_host():
    """Behind a torch-CPU baseline: let its worker threads stop spinning before the next short GPU measurement is issued."""
    time.sleep(0.5)


def cpu_baseline(Xn_host, dims, acts, lag, batch, lr, seconds, linears):
    """The CPU restatement of the reference path (oracle: torch-CPU autograd + Adam, same
    architecture, batch and dtype) timed on this box's host cores on a bounded sample."""
    from oracle import nn as onn

    model = onn.DeepTICAModel(dims, acts, None, None, None, 1e-6)
    lins = [m for m in model.nn if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        for l, (w, b) in zip(lins, linears):
            l.weight.copy_(torch.from_numpy(w))
            l.bias.copy_(torch.from_numpy(b))
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    xt = torch.from_numpy(Xn_host)
    nb = max(1, (xt.shape[0] - lag) // batch)

    def step(i):
        r0 = (i % nb) * batch
        opt.zero_grad()
        loss, _ = model.step(xt[r0:r0 + batch], xt[r0 + lag:r0 + lag + batch])
        loss.backward()
        opt.step()

    step(0)  # warm-up
    t0 = time.perf_counter()
    done = 0
    while True:
        step(done + 1)
        done += 1
        if time.perf_counter() - t0 >= seconds or done >= 2000:
            break
    dt = time.perf_counter() - t0
    return done * batch / dt, done, dt


class Fit:
    """The Deep-TICA fit loop of one rank at a given global batch: K optimiser steps over consecutive batches of the
    resident matrix (lengths [0.8, 0.2], sequential split, no shuffling), the validation pass at every epoch boundary."""

    PRIME = 30   # untimed priming steps in front of the W warm-up steps of every run()

    def __init__(self, hip, dist, Xn, dims, acts, lag, global_batch, world, lr, linears, n_local, step_comm=None, shuffled=False, seed=0):
        self.hip, self.dist, self.Xn, self.dims, self.lag = hip, dist, Xn, dims, lag
        self.step_comm = step_comm if step_comm is not None else dist   # what the data-parallel steps exchange through
        self.gb = global_batch
        self.lb = global_batch // world
        self.shuffled = shuffled
        P_local = n_local - lag
        self.n_train = int(P_local * 0.8) // self.lb * self.lb
        self.steps_per_epoch = self.n_train // self.lb
        self.val_steps = (P_local - self.n_train) // self.lb
        assert self.steps_per_epoch >= 1, "shard too small for the batch"
        self.eng = hip.Mlp("deep_tica", dims, acts, max_batch=self.lb, lag=lag, tica_reg=1e-6, lr=lr)
        self.eng.set_linears(linears)
        self.sv, self.gv = self.eng.stats_view(), self.eng.grads_view()
        if shuffled:
            # the reference's default loader (yaml_schemas/train_colvars.py:55-56: shuffle True, random_split True; DictModule /
            # DictLoader of mlcolvar): ONE random permutation of the pairs splits them into training / validation, the
            # training pairs are re-permuted at every epoch start, a batch is a consecutive slice of that permutation.  A pair
            # index p stands for (row p, row p + lag): the engine gathers both halves through the int64 index (RowMap).
            self.gen = torch.Generator(device=Xn.device)
            self.gen.manual_seed(1234 + seed)
            perm = torch.randperm(P_local, generator=self.gen, device=Xn.device)
            self.train_idx = perm[: self.n_train].contiguous()
            self.val_idx = perm[self.n_train: self.n_train + self.val_steps * self.lb].contiguous()
            self.epoch_idx = None

    def _epoch_start(self):
        self.epoch_idx = self.train_idx[torch.randperm(self.n_train, generator=self.gen, device=self.Xn.device)].contiguous()

    def train_step(self, i):
        eng, j = self.eng, i % self.steps_per_epoch
        if self.shuffled:
            if j == 0 or self.epoch_idx is None:
                self._epoch_start()
            idx = self.epoch_idx[j * self.lb:(j + 1) * self.lb]
            if self.dist is None:
                eng.train_step(self.Xn, idx=idx)
            else:
                eng.data_parallel_step(self.Xn, self.step_comm, self.gb, idx=idx, train=True)
            return
        r0 = j * self.lb
        if self.dist is None:
            eng.train_step(self.Xn, row0=r0, batch=self.lb)
        else:   # statistics all-reduce, then the gradient all-reduce of the upper layers under the layer-0 weight gradient
            eng.data_parallel_step(self.Xn, self.step_comm, self.gb, row0=r0, batch=self.lb, train=True)

    def train_run(self, i, count):
        """Steps i .. i + count - 1 (within one epoch): one dcv_mlp_train_steps call on one GPU -- the epoch loop behind the C-ABI,
        as the calculator's fit runs it when no scheduler steps in between -- else step by step."""
        j = i % self.steps_per_epoch
        assert j + count <= self.steps_per_epoch
        if self.dist is not None or count == 1:
            for k in range(count):
                self.train_step(i + k)
            return
        if self.shuffled:
            if j == 0 or self.epoch_idx is None:
                self._epoch_start()
            self.eng.train_steps(self.Xn, self.lb, count, idx=self.epoch_idx[j * self.lb:])
        else:
            self.eng.train_steps(self.Xn, self.lb, count, row0=j * self.lb)

    @staticmethod
    def runs(steps, picked, steps_per_epoch):
        """The timed steps as (first step, count, sampled) runs: a sampled step stands alone, the unsampled ones between two
        sampled steps (and up to an epoch's end: the validation pass follows there) are one run."""
        i = 0
        while i < steps:
            if i in picked:
                yield i, 1, True
                i += 1
                continue
            j = i + 1
            while j < steps and j not in picked and j % steps_per_epoch != 0:
                j += 1
            yield i, j - i, False
            i = j

    def validation_pass(self):
        eng = self.eng
        if self.dist is None and self.val_steps > 1:   # as the calculator's _validate does: the whole pass in one call
            if self.shuffled:
                eng.eval_steps(self.Xn, self.lb, self.val_steps, idx=self.val_idx)
            else:
                eng.eval_steps(self.Xn, self.lb, self.val_steps, row0=self.n_train)
            return
        for j in range(self.val_steps):
            if self.shuffled:
                idx = self.val_idx[j * self.lb:(j + 1) * self.lb]
                if self.dist is None:
                    eng.eval_step(self.Xn, idx=idx)
                else:
                    eng.data_parallel_step(self.Xn, self.step_comm, self.gb, idx=idx, train=False)
                continue
            r0 = self.n_train + j * self.lb
            if self.dist is None:
                eng.eval_step(self.Xn, row0=r0, batch=self.lb)
            else:
                eng.data_parallel_step(self.Xn, self.step_comm, self.gb, row0=r0, batch=self.lb, train=False)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    @staticmethod
    def sample_plan(steps, profile_every):
        """{step: kinds left UNSAMPLED on that step} for the steps that carry events.  A launch stamped through
        hipExtLaunchKernel costs the step ~7 us of command-processor work (measured in round 4: both layer-0 products of every
        one of 20 steps stamped = 0.119 ms / step against 0.0986 unstamped), so the two products are sampled on DIFFERENT
        steps, each every `every`-th step: every = 4 in a run of <= 100 steps (the driver's 20 steps: five samples of each
        product; round 3 had three, one of them the step right behind the barrier -- 29.8 us live against 22.8 us in every
        rocprofv3 trace), 8 otherwise.  Step 0 (idle clocks, cold L2 behind the synchronisation) is never sampled."""
        every = profile_every if profile_every > 0 else (4 if steps <= 100 else 8)
        if steps == 1:
            return {0: ()}
        if every == 1:
            return {i: () for i in range(1, steps)}
        plan = {}
        for i in range(1, steps):
            if i % every == 1 % every:
                plan[i] = ("wgrad", "dgrad")      # the forward product only
            elif i % every == (1 + every // 2) % every:
                plan[i] = ("fwd",)                 # the weight gradient only
        return plan

    def run(self, steps, warmup, profile_every, with_validation=True):
        """Times exactly `steps` optimiser steps between barriers; returns (seconds [max over ranks], per-kernel
        profile {(layer, kind): (total ms, launches)}, log records)."""
        eng = self.eng
        n_val = (steps // self.steps_per_epoch + 2) * (self.val_steps + 1)
        eng.reset_log((steps + warmup + self.PRIME) * 2 + n_val + 16)
        for i in range(self.PRIME):   # untimed, before the W warm-up steps: code-object loads of every kernel variant, clock ramp
            self.train_step(i)
        for i in range(warmup):
            self.train_step(i)
        self.barrier()
        picked = self.sample_plan(steps, profile_every)
        self.samples = len(picked)
        if picked:
            eng.profile_begin(len(picked) + 1, 1)
        t0 = time.perf_counter()
        self.val_timed = 0
        for i, count, sampled in self.runs(steps, picked, self.steps_per_epoch):
            if picked:
                eng.profile_pause(not sampled, picked.get(i, ()))
            self.train_run(i, count)
            if with_validation and (i + count) % self.steps_per_epoch == 0:
                self.validation_pass()
                self.val_timed += self.val_steps
        self.barrier()
        elapsed = time.perf_counter() - t0
        prof = eng.profile_end() if picked else {}
        log = eng.read_log()
        if self.dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.Xn.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, prof, log

    def time_collectives(self, reps=40):
        """Measured us per all-reduce of the two things a data-parallel step exchanges (HIP events around `reps` back-to-back
        collectives on the step's stream, after a barrier; max over ranks): the 2d + 2d^2 float64 batch statistics and the
        float32 gradient buffer."""
        if self.dist is None:
            return None
        out = {}
        comm = self.step_comm
        for name, buf in (("statistics", self.sv), ("gradients", self.gv)):
            def once():
                if comm is self.dist:
                    self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM)
                else:
                    comm.all_reduce(buf, "sum")
            for _ in range(5):
                once()
            self.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                once()
            e1.record()
            torch.cuda.synchronize()
            t = torch.tensor([e0.elapsed_time(e1) * 1e3 / reps], dtype=torch.float64, device=self.Xn.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            out[name] = {"us_per_allreduce": float(t.item()), "bytes": int(buf.numel() * buf.element_size()), "dtype": str(buf.dtype).replace("torch.", "")}
        out["note"] = (f"{reps} back-to-back all-reduces between HIP events on the launch stream, max over ranks; "
                       "a step issues one of each (the gradient one in two pieces from 32768 rows per rank up)")
        return out

    def roofline(self, prof, gemm_mode, traffic_table):
        """Roofline object of the slower layer-0 product (the dominant kernel of the step): algorithmic flop of one
        launch / its mean duration between HIP events on the launch stream, against the ceiling of the arithmetic."""
        if not prof:
            return None
        # contiguous batches: the network runs once on the batch + lag rows both halves share; gathered (shuffled) batches: 2 x batch rows
        R = 2 * self.lb if self.shuffled else self.lb + self.lag
        per = {f"layer{l}.{k}": (2.0 * R * self.dims[l] * self.dims[l + 1], ms / cnt) for (l, k), (ms, cnt) in prof.items()}
        name, (fl, ms) = max(per.items(), key=lambda kv: kv[1][1])
        achieved = fl / (ms * 1e-3) / 1e12
        if gemm_mode == "split":
            peak = PEAK_BF16_MFMA_TFLOPS / 6.0
            kname = f"{name} (gemm_kernel, 6 x v_mfma_f32_32x32x16_bf16 per FP32-accurate 32x32x16 block)"
            note = ("algorithmic fp32 flop / time against the BF16 dense MFMA peak (2500 TFLOP/s) / 6 products per block; "
                    f"the same rate is {achieved / PEAK_F32_MFMA_TFLOPS:.2f} x the FP32-input MFMA peak of {PEAK_F32_MFMA_TFLOPS} TFLOP/s")
        else:
            peak = PEAK_F32_MFMA_TFLOPS
            kname = f"{name} (gemm_kernel, FP32 MFMA 32x32x2)"
            note = "algorithmic fp32 flop / time against the FP32-input MFMA dense peak"
        traffic = None
        for entry in traffic_table:
            if entry.get("rows_per_launch") == R and entry.get("gemm_mode", "split") == gemm_mode:
                traffic = entry.get("kernels", {}).get(name, {}).get("hbm_bytes_per_launch")
        return {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                "kernel": kname, "rows_per_launch": R, "flop_per_launch": fl, "avg_ms": ms,
                "timing_source": "HIP events on the launch stream stamped with that launch's own begin / end (hipExtLaunchKernel through "
                                 "dcv_mlp_profile_*: the interval rocprofv3 --kernel-trace reports, profiles/), live inside bench.py: "
                                 "each layer-0 product on every 4th timed step of a run of <= 100 steps (8th otherwise), the two products on "
                                 "different steps, never the first step behind the barrier",
                "samples": {k: int(c) for k, c in sorted(((f"layer{l}.{kk}", cnt) for (l, kk), (_, cnt) in prof.items()))},
                "all_kernels_ms": {k: v[1] for k, v in sorted(per.items())}, "note": note}

    def close(self):
        self.eng.close()


def load_traffic_tables():
    out = []
    pdir = os.path.join(ROOT, "profiles")
    for fn in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if fn.startswith("traffic_") and fn.endswith(".json"):
            try:
                tj = json.load(open(os.path.join(pdir, fn)))
            except Exception:
                continue
            out.extend(tj if isinstance(tj, list) else [tj])
    return out


def cpu_baseline_ae(Xn_host, dims, acts_enc, acts_dec, latent, batch, lr, seconds, linears, feat_range):
    """The torch-CPU oracle of the autoencoder step (same MLP, f32, same batches) on this box's host cores, bounded."""
    from oracle import nn as onn

    F = dims[0]
    model = onn.AEModel(dims[:latent + 1], acts_enc, None, dims[latent:], acts_dec, None, np.zeros(F, np.float32), feat_range)
    lins = [m for m in list(model.encoder) + list(model.decoder) if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        for l, (w, b) in zip(lins, linears):
            l.weight.copy_(torch.from_numpy(w))
            l.bias.copy_(torch.from_numpy(b))
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    xt = torch.from_numpy(Xn_host)
    nb = max(1, xt.shape[0] // batch)

    def step(i):
        r0 = (i % nb) * batch
        opt.zero_grad()
        loss, _ = model.step(xt[r0:r0 + batch])
        loss.backward()
        opt.step()

    step(0)
    t0 = time.perf_counter()
    done = 0
    while True:
        step(done + 1)
        done += 1
        if time.perf_counter() - t0 >= seconds or done >= 5000:
            break
    dt = time.perf_counter() - t0
    return done * batch / dt, done, dt


def run_c2(a, steps, warmup, cpu_seconds):
    """BASELINE.json configs[1]: autoencoder CV (2 hidden layers, dim 2) on 1M frames x 128 synthetic features, one MI355X.
    A step = one optimiser step over 4096 frames; the fit loop is the reference's (lengths [0.8, 0.2], sequential split,
    validation pass at every epoch end inside the timed region).  Returns the metric object (its own value / roofline /
    cpu_baseline)."""
    from deep_cartograph_amd import hip
    from deep_cartograph_amd.synth import synth_features

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n, F, d, bs = 1_000_000, 128, 2, 4096
    dims = [F, 64, 32, d, 32, 64, F]
    acts = ["leaky_relu", "leaky_relu", None, "leaky_relu", "leaky_relu", None]
    latent = 3
    X = synth_features(n, F, k_slow=4, shard=0, device=dev)
    st = hip.finalize_stats(hip.col_stats_raw(X), n)
    std = st["std"].copy()
    std[np.abs(std) < 1e-8] = 1.0
    hip.normalize(X, torch.from_numpy(st["mean"]).to(dev), torch.from_numpy(std).to(dev), out=X)
    Xn = X
    linears = init_linears(dims, 43)
    eng = hip.Mlp("ae", dims, acts, max_batch=bs, latent_layer=latent, lr=a.lr)
    eng.set_linears(linears)
    eng.set_feature_range(std.astype(np.float32))
    n_train = int(n * 0.8) // bs * bs
    spe, val_steps = n_train // bs, (n - n_train) // bs

    def train_step(i):
        eng.train_step(Xn, row0=(i % spe) * bs, batch=bs)

    def validation():   # as the calculator's _validate does: the whole pass in one call (one launch for this network)
        if val_steps > 1:
            eng.eval_steps(Xn, bs, val_steps, row0=n_train)
        else:
            for j in range(val_steps):
                eng.eval_step(Xn, row0=n_train + j * bs, batch=bs)

    eng.reset_log(2 * (steps + warmup + 64) + (steps // spe + 2) * (val_steps + 1))
    for i in range(30 + warmup):
        train_step(i)
    torch.cuda.synchronize()
    # one fused launch per step (both classes are that launch): the plan's forward samples alone, every 8th step of a long run
    picked = {i: () for i, skip in Fit.sample_plan(steps, a.profile_every).items() if "fwd" not in skip}
    if picked:
        eng.profile_begin(len(picked) + 1, 1)
    t0 = time.perf_counter()
    n_val = 0
    for i, count, sampled in Fit.runs(steps, picked, spe):   # the unsampled steps between two sampled ones: one dcv_mlp_train_steps call
        if picked:
            eng.profile_pause(not sampled)
        if count == 1:
            train_step(i)
        else:
            eng.train_steps(Xn, bs, count, row0=(i % spe) * bs)
        if (i + count) % spe == 0:
            validation()
            n_val += val_steps
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_end() if picked else {}
    log = eng.read_log()
    sw = sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
    roof = None
    if (0, "fwd") in prof:
        ms, cnt = prof[(0, "fwd")]
        fl = 6.0 * bs * sw
        ach = fl / (ms / cnt * 1e-3) / 1e12
        roof = {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
                "traffic": None, "kernel": "snet_ae_kernel (whole step fused: forward, loss, backward of a row tile per workgroup, "
                                             "v_mfma_f32_16x16x4_f32, weights resident in LDS)",
                "flop_per_launch": fl, "avg_ms": ms / cnt, "samples": cnt,
                "timing_source": "HIP events stamped with the fused launch's own begin / end (hipExtLaunchKernel through dcv_mlp_profile_*), live inside bench.py",
                "note": "algorithmic 6 * batch * sum(in*out) flop of one launch / its mean duration; the kernel is latency-bound at 4096 rows: "
                        "the fraction is reported for the record, not as a claim of MFMA saturation"}
    out = {"metric": "Autoencoder training frames/sec on 1Mx128 feature matrix (BASELINE.json configs[1]) at 1 GPU", "value": steps * bs / elapsed,
           "unit": "frames/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"Autoencoder fit (BASELINE.md C2), {n}x{F} f32 synthetic AR(1) features, MLP {'-'.join(map(str, dims))}, batch {bs}, "
                                  f"Adam lr {a.lr}, lengths [0.8,0.2], sequential split; {n_val} validation steps inside the timed region",
                      "frames": n, "features": F, "global_batch": bs, "steps_per_epoch": spe, "val_steps_per_epoch": val_steps,
                      "validation_steps_timed": n_val, "params": sw},
           "loss_first": float(log[0, 0]) if len(log) else None, "loss_last": float(log[-1, 0]) if len(log) else None, "roofline": roof}
    if cpu_seconds > 0:
        rows = min(n, 40 * bs)
        v, done, dt = cpu_baseline_ae(Xn[:rows].cpu().numpy(), dims, acts[:latent], acts[latent:], latent, bs, a.lr, cpu_seconds, linears,
                                      std.astype(np.float32))
        out["cpu_baseline"] = {"value": v, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{done} optimiser steps of the torch-CPU oracle (same autoencoder, f32, batches of {bs}) on the first {rows} frames, {dt:.1f} s"}
        settle_host()
    eng.close()
    del X, Xn
    return out


def run_calculator_fit(epochs):
    """The calculators themselves, end to end: `train()` of the Deep-TICA and autoencoder calculators (the reference's
    NonLinear.train, cv_calculator.py:1456-1553) on 200 000 x 54 synthetic features with the reference's default architecture
    ([16, 8] encoder), lag 5, its default loader (random split + shuffle), Adam 1e-3, `epochs` epochs with a validation pass each,
    at the batch sizes 256 and 4096.  Wall clock around the call: everything the host does per epoch is inside."""
    import contextlib
    import tempfile

    from deep_cartograph_amd.cv_calculator import cv_calculators_map
    from deep_cartograph_amd.synth import synth_features

    n, F = 200_000, 54
    X = synth_features(n, F, k_slow=2, shard=0, device="cuda").cpu().numpy()
    names = [f"f{i}" for i in range(F)]
    arch = {"encoder": {"layers": [16, 8], "activation": ["leaky_relu", "leaky_relu"], "batchnorm": [False, False], "dropout": [0, 0],
                        "last_layer_activation": None, "last_layer_batchnorm": False, "last_layer_dropout": None},
            "decoder": {"layers": [4, 8], "activation": ["leaky_relu", "leaky_relu"], "batchnorm": [False, False], "dropout": [0, 0],
                        "last_layer_activation": None, "last_layer_batchnorm": False, "last_layer_dropout": None}}
    runs = []
    for kind in ("deep_tica", "ae"):
        for bs in (256, 4096):
            cfg = {"dimension": 2, "lag_time": 5, "features_normalization": "mean_std", "tica_regularization": 1e-6, "architecture": arch,
                   "training": {"general": {"num_tries": 1, "seed": 42, "lengths": [0.8, 0.2], "batch_size": bs, "max_epochs": epochs, "shuffle": True,
                                            "random_split": True, "check_val_every_n_epoch": 1, "save_check_every_n_epoch": 1},
                                "early_stopping": {"patience": 100000, "min_delta": 0.0}, "optimizer": {"name": "Adam", "kwargs": {"lr": 1e-3, "weight_decay": 0}},
                                "lr_scheduler": None, "lr_scheduler_config": None, "save_loss": False, "plot_loss": False, "model_to_save": "last"}}
            with tempfile.TemporaryDirectory() as out, contextlib.redirect_stdout(sys.stderr):   # (stdout carries the one JSON line)
                calc = cv_calculators_map[kind](cfg, out)
                calc.set_training_matrix(X.copy(), names)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ok = calc.train()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            steps = epochs * (int((n - (5 if kind == "deep_tica" else 0)) * 0.8) // bs)
            runs.append({"cv": kind, "batch": bs, "epochs": epochs, "ok": bool(ok), "seconds": dt, "ms_per_epoch": dt / epochs * 1e3,
                         "us_per_training_step_all_inclusive": dt / max(steps, 1) * 1e6, "value": steps * bs / dt, "unit": "frames/s"})
    return {"workload": f"CVCalculator.train() wall clock, {n}x{F} f32 synthetic AR(1) features, encoder [16, 8], dimension 2, lag 5, random split + shuffled "
                        f"loader, Adam 1e-3, {epochs} epochs with validation, one try", "runs": runs}


def run_ref_small(a, steps, cpu_seconds):
    """Deep-TICA on the reference's OWN network sizes (cv_calculator.py:2569-2590): 54-16-8-2 (its test configuration,
    tests/data/input/train_colvars + test_train_colvars.py) and F-15-15-2 (tools/train_colvars/default_config.yml:45-55 `layers:
    [15, 15]`), at batch 128 (the test's clamp of 256) and 4096, with the reference's default loader (random split, fresh
    permutation per epoch: gathered two-half batches).  One line per (network, batch): frames/s, us/step, the step's roofline
    against the FP32-input MFMA peak (algorithmic 12 * sum(in*out) * batch flop per step), the torch-CPU oracle beside it."""
    from deep_cartograph_amd import hip
    from deep_cartograph_amd.synth import synth_features

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n, F, lag = 400_000, 54, 10
    X = synth_features(n, F, k_slow=2, shard=0, device=dev)
    st = hip.finalize_stats(hip.col_stats_raw(X), n)
    std = st["std"].copy()
    std[np.abs(std) < 1e-8] = 1.0
    hip.normalize(X, torch.from_numpy(st["mean"]).to(dev), torch.from_numpy(std).to(dev), out=X)
    rows_out, todo_cpu = [], []
    for hidden in ([16, 8], [15, 15]):
        dims = [F] + hidden + [2]
        acts = ["leaky_relu"] * (len(dims) - 2) + [None]
        linears = init_linears(dims, 43)
        sw = sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
        for bs in (128, 4096):
            fit = Fit(hip, None, X, dims, acts, lag, bs, 1, a.lr, linears, n, shuffled=True)
            elapsed, prof, log = fit.run(steps, 30, a.profile_every, with_validation=False)
            fused = fit.eng.last_path() == 2
            fl = 12.0 * sw * bs
            us = elapsed / steps * 1e6
            rec = {"network": "-".join(map(str, dims)), "batch": bs, "value": steps * bs / elapsed, "unit": "frames/s", "us_per_step": us,
                   "steps": steps, "fused_small_network_path": fused, "loss_last": float(log[-1, 0]) if len(log) else None,
                   "roofline": {"bound": "mfma", "achieved": fl / (us * 1e-6) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                "frac": fl / (us * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                                "note": "whole step (every launch + gaps): algorithmic 12 * sum(in*out) * batch flop / step time; latency-bound by "
                                        "construction at these sizes -- reported for the record"}}
            fit.close()
            rows_out.append(rec)
            todo_cpu.append((rec, dims, acts, bs, linears))
    # the CPU baselines AFTER every GPU measurement of the block: the worker threads of a 128-thread torch-CPU run keep spinning
    # for a while behind it, and a 10 ms GPU measurement started right behind one read 10 x too long now and then
    # (profiles/r04_bench_line_driver_form.json of 07:54: 296 us per step between two runs at 27 and 34)
    if cpu_seconds > 0:
        for rec, dims, acts, bs, linears in todo_cpu:
            sample_rows = min(n, 40 * bs + lag)
            v, done, dt = cpu_baseline(X[:sample_rows].cpu().numpy(), dims, acts, lag, bs, a.lr, cpu_seconds, linears)
            rec["cpu_baseline"] = {"value": v, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": f"{done} optimiser steps of the torch-CPU oracle (same MLP, f32, contiguous batches of {bs} pairs), {dt:.1f} s"}
        settle_host()
    del X
    return {"workload": f"Deep-TICA fit on the reference's own network sizes, {n}x{F} f32 synthetic AR(1) features, lag {lag}, Adam lr {a.lr}, "
                        "random split + per-epoch permutation (the reference's default loader), training steps only",
            "runs": rows_out}


def self_launch(a):
    """`python bench.py --gpus N` with no launcher environment: start N fresh child processes of this script, one rank per
    GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set here), BEFORE anything in this process touches the GPU -- no os.exec*,
    no re-launch of an initialised process.  Rank 0's standard output (the JSON line) is relayed; the exit code is non-zero
    when any rank's is."""
    import socket
    import subprocess

    n = a.gpus
    ndev = torch.cuda.device_count()   # counting devices does not initialise the GPU on this image
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU path")
    if ndev < n and a.backend == "nccl":
        raise SystemExit(f"bench.py --gpus {n}: only {ndev} GPU(s) visible and RCCL wants one device per rank "
                         "(rehearse the multi-rank control flow on one GPU with --backend gloo)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r % ndev), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        out0, _ = procs[0].communicate()
        for pr in procs:
            pr.wait()
            rc = rc or pr.returncode
    finally:
        for pr in procs:   # a rank that died leaves the others in a collective: end exactly the processes started here
            if pr.poll() is None:
                pr.kill()
    sys.stdout.write(out0.decode() if out0 else "")
    sys.stdout.flush()
    if rc != 0:
        raise SystemExit(rc if rc > 0 else 1)


def main_c2(a):
    print(json.dumps(run_c2(a, a.steps, a.warmup, 0.0 if a.no_cpu_baseline else a.cpu_seconds)))


def main():
    a = parse()
    if a.config != "c4":
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU path")
        if a.config == "c2":
            return main_c2(a)
        return print(json.dumps(run_ref_small(a, a.steps, 0.0 if a.no_cpu_baseline else min(a.cpu_seconds, 5.0))))
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return self_launch(a)   # before any GPU call in this process
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} inside a launcher environment of WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("DCV_FORCE_DIST") == "1":  # the env switch exercises the collective path on one GPU
        import torch.distributed as dist  # noqa: F811

        if world == 1:   # forced single-rank group: supply the rendezvous the launcher would
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29531")):
                os.environ.setdefault(k, v)

        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.backend)

    from deep_cartograph_amd import hip
    from deep_cartograph_amd.synth import synth_features

    dev = torch.device("cuda", local_rank)
    F, lag, d = a.features, a.lag, a.dim
    dims = [F] + [int(x) for x in a.hidden.split(",") if x] + [d]
    acts = ["leaky_relu"] * (len(dims) - 2) + [None]
    assert a.batch % world == 0
    n_local = a.frames // world                # frames of this rank's shard (an independent trajectory)

    # ---- data: generate on the device, standardise with the HIP kernels, keep resident
    X = synth_features(n_local, F, k_slow=4, shard=rank, device=dev)
    raw = hip.col_stats_raw(X)
    nglob = torch.tensor([float(n_local)], dtype=torch.float64, device=dev)
    if dist is not None:
        mm = raw[2:].clone()
        dist.all_reduce(raw[:2], op=dist.ReduceOp.SUM)
        dist.all_reduce(mm[0], op=dist.ReduceOp.MIN)
        dist.all_reduce(mm[1], op=dist.ReduceOp.MAX)
        raw[2:] = mm
        dist.all_reduce(nglob, op=dist.ReduceOp.SUM)
    st = hip.finalize_stats(raw, int(nglob.item()))
    mean_t = torch.from_numpy(st["mean"]).to(dev)
    std = st["std"].copy()
    std[np.abs(std) < 1e-8] = 1.0
    range_t = torch.from_numpy(std).to(dev)
    hip.normalize(X, mean_t, range_t, out=X)   # in place: Xn
    Xn = X

    hip.set_gemm_mode(a.gemm_mode)
    linears = init_linears(dims, 43)
    traffic_tables = load_traffic_tables()

    # ---- headline: the contract batch
    step_comm = hip.RcclComm(dist, device=dev) if (dist is not None and a.native_rccl) else None
    fit = Fit(hip, dist, Xn, dims, acts, lag, a.batch, world, a.lr, linears, n_local, step_comm)
    elapsed, prof, log = fit.run(a.steps, a.warmup, a.profile_every)
    val_timed = fit.val_timed
    head_roof = fit.roofline(prof, a.gemm_mode, traffic_tables) if rank == 0 else None

    # the same training steps in the other arithmetic mode (a launch-time switch of the library)
    other = "native" if a.gemm_mode == "split" else "split"
    other_res = None
    if a.other_mode_steps > 0:
        hip.set_gemm_mode(other)
        elo, prof_o, _ = fit.run(a.other_mode_steps, 20, a.profile_every, with_validation=False)
        hip.set_gemm_mode(a.gemm_mode)
        if rank == 0:
            other_res = {"gemm_mode": other, "value": a.other_mode_steps * a.batch / elo, "unit": "frames/s",
                         "ms_per_step": elo / a.other_mode_steps * 1e3, "steps": a.other_mode_steps,
                         "note": "training steps only (no validation pass), same engine, arithmetic switched at launch time"}
            r = fit.roofline(prof_o, other, traffic_tables)
            if r:
                other_res["roofline"] = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_ms")}
    steps_per_epoch, val_steps = fit.steps_per_epoch, fit.val_steps
    collectives = fit.time_collectives()
    comm_world = None
    if dist is not None:
        comm_world = int(hip._lib.load().dcv_comm_world(step_comm.h)) if step_comm is not None else int(dist.get_world_size())
    fit.close()

    # ---- the reference's default loader at the same batch: random split, a fresh permutation per epoch, gathered two-half batches
    shuffled = None
    if a.shuffled_steps > 0:
        fs = Fit(hip, dist, Xn, dims, acts, lag, a.batch, world, a.lr, linears, n_local, step_comm, shuffled=True, seed=rank)
        es, prof_s, log_s = fs.run(a.shuffled_steps, 20, a.profile_every, with_validation=False)
        if rank == 0:
            sw_ = sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
            shuffled = {"global_batch": a.batch, "value": a.shuffled_steps * a.batch / es, "unit": "frames/s", "ms_per_step": es / a.shuffled_steps * 1e3,
                        "steps": a.shuffled_steps, "rows_per_launch": 2 * fs.lb, "flop_per_step": 12.0 * sw_ * fs.lb,
                        "loss_last_train": float(log_s[-1, 0]) if len(log_s) else None,
                        "roofline": fs.roofline(prof_s, a.gemm_mode, traffic_tables),
                        "note": "reference default (yaml_schemas/train_colvars.py:55-56 shuffle / random_split True): x_t and x_lag rows are gathered "
                                "through an int64 index, evaluated as two halves (2 x batch rows per product: twice the flops of the row-shared "
                                "sequential headline), permutation drawn on the device at each epoch start inside the timed region; training steps only"}
        fs.close()

    # ---- the same fit at a large global batch (MFMA-bound instead of launch / latency-bound)
    large = None
    if a.large_batch > 0 and a.large_batch % world == 0 and a.large_batch != a.batch and (n_local - lag) * 0.8 >= a.large_batch // world:
        fl_ = Fit(hip, dist, Xn, dims, acts, lag, a.large_batch, world, a.lr, linears, n_local, step_comm)
        el, prof_l, _ = fl_.run(a.large_steps, 5, a.profile_every)
        if rank == 0:
            large = {"global_batch": a.large_batch, "value": a.large_steps * a.large_batch / el, "unit": "frames/s",
                     "ms_per_step": el / a.large_steps * 1e3, "steps": a.large_steps, "steps_per_epoch": fl_.steps_per_epoch,
                     "roofline": fl_.roofline(prof_l, a.gemm_mode, traffic_tables),
                     "note": "same matrix, model, split and arithmetic; validation pass at each epoch boundary inside the timed region"}
        fl_.close()

    if rank == 0:
        sw = sum(dims[i] * dims[i + 1] + dims[i + 1] for i in range(len(dims) - 1))
        losses = log[:, 0]
        out = {
            "metric": "Deep-TICA training frames/sec on 10Mx512 feature matrix at 1/2/4/8 GPUs",
            "value": a.steps * a.batch / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "arithmetic": ("FP32-input MFMA (exact f32 products, f32 accumulate)" if a.gemm_mode == "native" else
                           "f32 operands split into 3 bf16 pieces, 6 bf16 MFMA products per block, f32 accumulate: f32-accurate "
                           "(error vs float64 at or below the FP32-input MFMA path, tests/test_kernels_gpu.py); covariances / linear CVs "
                           "always use the FP32-input MFMA"),
            "data": "synthetic",
            "config": {
                "workload": f"Deep-TICA fit (BASELINE.md C4), {a.frames}x{F} f32 synthetic AR(1) features (SURVEY 8d), MLP {'-'.join(map(str, dims))}, "
                            f"lag {lag}, global batch {a.batch} pairs, Adam lr {a.lr}, lengths [0.8,0.2], sequential split, "
                            f"validation pass at each epoch end the {a.steps} timed steps cross ({val_timed} validation steps inside the timed "
                            f"region of this run); contiguous batches evaluate the batch + lag rows shared by x_t and x_lag once",
                "frames": a.frames, "features": F, "global_batch": a.batch, "parallelism": f"frame-shard dp{world}",
                "collectives": (None if dist is None else ("libdcv RCCL communicator (dcv_comm_*)" if step_comm is not None else
                                                           f"torch.distributed {a.backend} through the all-reduce callback of dcv_mlp_dp_step")),
                "communicator_world_size": comm_world, "collective_timing": collectives,
                "steps_per_epoch": steps_per_epoch, "val_steps_per_epoch": val_steps, "validation_steps_timed": val_timed, "params": sw,
                "gemm_mode": a.gemm_mode,
            },
            "loss_first": float(losses[0]) if len(losses) else None,
            "loss_last_train": float(losses[-1]) if len(losses) else None,
            "roofline": head_roof,
        }
        if other_res is not None:
            out["other_gemm_mode"] = other_res
        if large is not None:
            out["large_batch"] = large
        if shuffled is not None:
            out["shuffled"] = shuffled
        if not a.no_cpu_baseline and world == 1:
            sample_rows = min(n_local, 40 * a.batch + lag)
            Xh = Xn[:sample_rows].cpu().numpy()
            v, done, dt = cpu_baseline(Xh, dims, acts, lag, min(a.batch, sample_rows - lag), a.lr, a.cpu_seconds, linears)
            out["cpu_baseline"] = {
                "value": v, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{done} optimiser steps of the torch-CPU oracle (same MLP, f32, batches of {a.batch} pairs) on the first "
                          f"{sample_rows} frames, {dt:.1f} s",
            }
            settle_host()
        if world == 1 and dist is None:
            # bounded runs of the other single-GPU configurations, carried by the same JSON line (the driver records one line)
            del X, Xn
            torch.cuda.empty_cache()
            cpu_s = 0.0 if a.no_cpu_baseline else min(a.cpu_seconds, 8.0)
            def block(name, fn):   # a block that fails is reported as such: the headline above is measured and must be printed
                try:
                    out[name] = fn()
                except Exception as e:   # noqa: BLE001
                    import traceback

                    traceback.print_exc(file=sys.stderr)
                    out[name] = {"error": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()

            def c2_block():
                c2 = run_c2(a, a.c2_steps, 50, cpu_s)
                blk = {k: c2[k] for k in ("metric", "value", "unit", "steps", "ms_per_step", "config", "loss_first", "loss_last", "roofline") if k in c2}
                if "cpu_baseline" in c2:
                    blk["cpu_baseline"] = c2["cpu_baseline"]
                return blk

            if a.c2_steps > 0:
                block("c2", c2_block)
            if a.ref_small_steps > 0:
                block("ref_small", lambda: run_ref_small(a, a.ref_small_steps, 0.0 if a.no_cpu_baseline else 3.0))
            if a.fit_epochs > 0:
                block("calculator_fit", lambda: run_calculator_fit(a.fit_epochs))
        print(json.dumps(out))
    if step_comm is not None:
        step_comm.close()   # the library's communicator goes before the process group that bootstrapped it
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
