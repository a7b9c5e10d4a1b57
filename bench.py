#!/usr/bin/env python3
"""Headline benchmark: Deep-TICA training frames/sec on a synthetic 10M x 512 feature matrix
(BASELINE.json metric), frame-sharded over N GPUs.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one optimiser step over one global batch of time-lagged pairs.  The matrix is
generated on the device, standardised by the HIP statistics / normalise kernels and stays
resident in HBM; the timed region holds exactly K training steps (plus the validation pass at
every epoch boundary they cross, as the reference's fit loop does) between barriers.  Rank 0
prints one JSON line."""
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
    p.add_argument("--steps", type=int, default=120)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--frames", type=int, default=10_000_000)
    p.add_argument("--features", type=int, default=512)
    p.add_argument("--hidden", type=str, default="256,128")
    p.add_argument("--dim", type=int, default=4)
    p.add_argument("--lag", type=int, default=10)
    p.add_argument("--batch", type=int, default=524208,
                   help="global batch (pairs per optimiser step); default 8 x (65536 - lag): every rank's step covers whole "
                        "128-row tiles at 1/2/4/8 GPUs when the rows of x_t and x_lag are shared")
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=15.0)
    p.add_argument("--profile-level", type=int, default=1)
    p.add_argument("--alt-batch", type=int, default=8192,
                   help="also report (field 'secondary') the throughput at this global batch -- the size SURVEY.md 8d floated; 0 disables")
    p.add_argument("--alt-steps", type=int, default=300)
    p.add_argument("--gemm-mode", choices=["split", "native"], default="split",
                   help="arithmetic of the MLP products: 'split' = FP32-accurate split products on the BF16 matrix pipe "
                        "(library default), 'native' = FP32-input MFMA; the other mode is timed too (field 'other_gemm_mode')")
    p.add_argument("--other-mode-steps", type=int, default=40)
    return p.parse_args()


def init_linears(dims, seed):
    """torch.nn.Linear default initialisation in construction order (what create_model() does)."""
    torch.manual_seed(seed)
    lins = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)]
    return [(l.weight.detach().numpy().copy(), l.bias.detach().numpy().copy()) for l in lins]


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
        if time.perf_counter() - t0 >= seconds or done >= 200:
            break
    dt = time.perf_counter() - t0
    return done * batch / dt, done, dt


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("DCV_FORCE_DIST") == "1":  # the env switch exercises the collective path on one GPU
        import torch.distributed as dist  # noqa: F811

        if world == 1:   # forced single-rank group: supply the rendezvous the launcher would
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29531")):
                os.environ.setdefault(k, v)

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from deep_cartograph_amd import hip
    from deep_cartograph_amd.synth import synth_features

    dev = torch.device("cuda", local_rank)
    F, lag, d = a.features, a.lag, a.dim
    dims = [F] + [int(x) for x in a.hidden.split(",") if x] + [d]
    acts = ["leaky_relu"] * (len(dims) - 2) + [None]
    assert a.batch % world == 0
    lb = a.batch // world                      # pairs per rank per step
    n_local = a.frames // world                # frames of this rank's shard (an independent trajectory)
    P_local = n_local - lag
    n_train = int(P_local * 0.8) // lb * lb    # lengths [0.8, 0.2], random_split False, shuffle False
    steps_per_epoch = n_train // lb
    val_steps = (P_local - n_train) // lb
    assert steps_per_epoch >= 1, "shard too small for the batch"

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
    eng = hip.Mlp("deep_tica", dims, acts, max_batch=lb, lag=lag, tica_reg=1e-6, lr=a.lr)
    eng.set_linears(linears)
    stats_v = eng.stats_view()
    grads_v = eng.grads_view()
    n_records = (a.steps + a.warmup) * 2 + (a.steps // steps_per_epoch + 2) * (val_steps + 1) + 16
    eng.reset_log(n_records)

    def train_step(i):
        r0 = (i % steps_per_epoch) * lb
        if dist is None:
            eng.train_step(Xn, row0=r0, batch=lb)
        else:
            eng.forward(Xn, row0=r0, batch=lb)
            dist.all_reduce(stats_v, op=dist.ReduceOp.SUM)
            eng.backward(Xn, row0=r0, batch=lb, global_batch=a.batch, train=True)
            dist.all_reduce(grads_v, op=dist.ReduceOp.SUM)
            eng.apply()

    def validation_pass():
        for j in range(val_steps):
            r0 = n_train + j * lb
            if dist is None:
                eng.eval_step(Xn, row0=r0, batch=lb)
            else:
                eng.forward(Xn, row0=r0, batch=lb, train=False)
                dist.all_reduce(stats_v, op=dist.ReduceOp.SUM)
                eng.backward(Xn, row0=r0, batch=lb, global_batch=a.batch, train=False)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        train_step(i)
    barrier()
    eng.profile_begin(a.steps, a.profile_level)
    t0 = time.perf_counter()
    for i in range(a.steps):
        train_step(i)
        if (i + 1) % steps_per_epoch == 0:
            validation_pass()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_end()
    log = eng.read_log()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- the same training steps in the other arithmetic mode (the mode is a launch-time switch of the library)
    other = "native" if a.gemm_mode == "split" else "split"
    other_res = None
    if a.other_mode_steps > 0:
        hip.set_gemm_mode(other)
        for i in range(5):
            train_step(i)
        barrier()
        eng.profile_begin(a.other_mode_steps, 1)
        t0o = time.perf_counter()
        for i in range(a.other_mode_steps):
            train_step(i)
        barrier()
        elo = time.perf_counter() - t0o
        prof_o = eng.profile_end()
        hip.set_gemm_mode(a.gemm_mode)
        if dist is not None:
            t = torch.tensor([elo], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elo = float(t.item())
        other_res = {"gemm_mode": other, "value": a.other_mode_steps * a.batch / elo, "unit": "frames/s",
                     "ms_per_step": elo / a.other_mode_steps * 1e3, "steps": a.other_mode_steps,
                     "note": "training steps only (no validation pass), same engine state, arithmetic switched at launch time"}
        if prof_o:   # the same roofline line for the other arithmetic: slower layer-0 product against that arithmetic's ceiling
            (ol_, ok_), (oms, ocnt) = max(prof_o.items(), key=lambda kv: kv[1][0] / kv[1][1])
            ofl = 2.0 * (lb + lag) * dims[ol_] * dims[ol_ + 1]
            oach = ofl / (oms / ocnt * 1e-3) / 1e12
            opeak = PEAK_F32_MFMA_TFLOPS if other == "native" else PEAK_BF16_MFMA_TFLOPS / 6.0
            other_res["roofline"] = {"bound": "mfma", "kernel": f"layer{ol_}.{ok_}", "achieved": oach, "peak": opeak, "unit": "TFLOP/s",
                                     "frac": oach / opeak, "avg_ms": oms / ocnt}

    if rank == 0:
        R = lb + lag   # contiguous batches: the network runs once on the batch + lag rows both halves share
        flops = {}
        for (layer, kind), (ms, cnt) in prof.items():
            fl = 2.0 * R * dims[layer] * dims[layer + 1]
            flops[(layer, kind)] = (fl, ms / cnt)
        (dl, dk), (dfl, dms) = max(flops.items(), key=lambda kv: kv[1][1])
        achieved = dfl / (dms * 1e-3) / 1e12
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (same launch shape)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_r01.json")
        if os.path.exists(tpath) and F == 512 and dims[1] == 256:
            tj = json.load(open(tpath))
            if tj.get("rows_per_launch") == R:
                traffic = tj["kernels"].get(f"layer{dl}.{dk}", {}).get("hbm_bytes_per_launch")
        if a.gemm_mode == "split":
            # every fp32 flop is 6 bf16 MFMA flops: the ceiling of this arithmetic is the BF16 pipe's dense peak / 6
            peak = PEAK_BF16_MFMA_TFLOPS / 6.0
            kname = f"layer{dl}.{dk} (gemm_kernel, 6 x v_mfma_f32_32x32x16_bf16 per FP32-accurate 32x32x16 block)"
            peak_note = ("algorithmic fp32 flop / time against the BF16 dense MFMA peak (2500 TFLOP/s) / 6 products per block; "
                         f"the same rate is {achieved / PEAK_F32_MFMA_TFLOPS:.2f} x the FP32-input MFMA peak of {PEAK_F32_MFMA_TFLOPS} TFLOP/s "
                         "(which bounds --gemm-mode native, timed in 'other_gemm_mode')")
        else:
            peak = PEAK_F32_MFMA_TFLOPS
            kname = f"layer{dl}.{dk} (gemm_kernel, FP32 MFMA 32x32x2)"
            peak_note = "algorithmic fp32 flop / time against the FP32-input MFMA dense peak"
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
                "workload": f"Deep-TICA fit, {a.frames}x{F} f32 synthetic AR(1) features (SURVEY 8d, C4), MLP {'-'.join(map(str, dims))}, "
                            f"lag {lag}, global batch {a.batch} pairs, Adam lr {a.lr}, lengths [0.8,0.2], sequential split, "
                            f"validation pass at each epoch end inside the timed region; contiguous batches evaluate the "
                            f"batch + lag rows shared by x_t and x_lag once",
                "frames": a.frames, "features": F, "global_batch": a.batch, "parallelism": f"frame-shard dp{world}",
                "steps_per_epoch": steps_per_epoch, "val_steps_per_epoch": val_steps, "params": sw,
            },
            "loss_first": float(losses[0]) if len(losses) else None,
            "loss_last_train": float(losses[-1]) if len(losses) else None,
            "roofline": {
                "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak, "traffic": traffic,
                "kernel": kname, "flop_per_launch": dfl, "avg_ms": dms,
                "all_kernels_ms": {f"layer{l}.{k}": v[1] for (l, k), v in sorted(flops.items())},
                "note": peak_note,
            },
        }
        out["config"]["gemm_mode"] = a.gemm_mode
        if other_res is not None:
            out["other_gemm_mode"] = other_res
    eng.close()
    # ---- secondary: the same fit at the small global batch of SURVEY 8d (launch-latency bound; every rank runs it)
    alt = None
    if a.alt_batch > 0 and a.alt_batch % world == 0 and a.alt_batch != a.batch:
        alb = a.alt_batch // world
        a_train = int(P_local * 0.8) // alb * alb
        a_spe = a_train // alb
        if a_spe >= 1:
            eng2 = hip.Mlp("deep_tica", dims, acts, max_batch=alb, lag=lag, tica_reg=1e-6, lr=a.lr)
            eng2.set_linears(linears)
            sv2, gv2 = eng2.stats_view(), eng2.grads_view()
            eng2.reset_log(a.alt_steps + 64)

            def alt_step(i):
                r0 = (i % a_spe) * alb
                if dist is None:
                    eng2.train_step(Xn, row0=r0, batch=alb)
                else:
                    eng2.forward(Xn, row0=r0, batch=alb)
                    dist.all_reduce(sv2, op=dist.ReduceOp.SUM)
                    eng2.backward(Xn, row0=r0, batch=alb, global_batch=a.alt_batch, train=True)
                    dist.all_reduce(gv2, op=dist.ReduceOp.SUM)
                    eng2.apply()

            for i in range(50):
                alt_step(i)
            barrier()
            t0 = time.perf_counter()
            for i in range(a.alt_steps):
                alt_step(i)
            barrier()
            el = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([el], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            alt = {"global_batch": a.alt_batch, "value": a.alt_steps * a.alt_batch / el, "unit": "frames/s",
                   "ms_per_step": el / a.alt_steps * 1e3, "steps": a.alt_steps,
                   "note": "training steps only, same matrix / model / split; launch-latency bound at this size"}
            eng2.close()
    if rank == 0:
        if alt is not None:
            out["secondary"] = alt
        if not a.no_cpu_baseline and world == 1:
            cpu_batch = min(a.batch, 65536)   # bounded sample: frames/s of the CPU GEMMs does not depend on the batch size
            sample_rows = min(n_local, 4 * cpu_batch + lag)
            Xh = Xn[:sample_rows].cpu().numpy()
            v, done, dt = cpu_baseline(Xh, dims, acts, lag, min(cpu_batch, sample_rows - lag), a.lr, a.cpu_seconds, linears)
            out["cpu_baseline"] = {
                "value": v, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{done} optimiser steps of the torch-CPU oracle (same MLP, f32, batches of {cpu_batch} pairs) on the first "
                          f"{sample_rows} frames, {dt:.1f} s",
            }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
