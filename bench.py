#!/usr/bin/env python3
"""bench.py -- blobs/s of compute_cells_and_kzg_proofs on MI355X (BASELINE.json's metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (blob bytes -> 128 cells + 128 proofs per blob) over one batch of
synthetic blobs that is already resident in HBM, through the device-pointer C ABI
(eth_kzg_amd_compute_cells_and_kzg_proofs_device).  One process per GPU; the batch is sharded by
contiguous blob index with no data-path collective; with N > 1 each step ends with one RCCL
all-gather of the proof vectors (48 B x 128 per blob), which is the only exchange north_star names.
Weak scaling: the per-GPU batch is fixed (default 2048 blobs: the batch that fills all 1024 SIMDs of the chip in
the wave-per-butterfly G1-FFT stage; BASELINE.json's config-4 and config-2 sizes are reported alongside).

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BYTES_PER_BLOB, CELLS, BYTES_PER_CELL = 131072, 128, 2048
ALG_BYTES_PER_BLOB = BYTES_PER_BLOB + CELLS * BYTES_PER_CELL + CELLS * 48  # 399,360 B
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_INT_PEAK_GOPS = 34000.0     # measured on MI355X with tools/ubench.hip: integer VALU ops (v_mad_u64_u32,
                                 # v_addc_co_u32, ...) all issue at ~31-35 T lane-ops/s chip-wide
FP_MUL_PEAK_G = 78.1             # measured ceiling of the 14x29-bit Montgomery multiplication (392 v_mad_u64_u32 + 68 shifts /
                                 # masks as verbatim chains, fp29_mac.hpp) at 4 waves/SIMD: tools/ubench_fp29.hip,
                                 # profiles/r1e_ubench_fp29.log (66.0 G/s at the 2 waves/SIMD the point kernels can hold)


def fp_mul_eq_per_blob(window_bits):
    """Fp multiplication equivalents this build spends per blob, counted in multiply-add passes of 392 MACs
    (M = 1, squaring S = 301/392, fused pair a*b + c*d with one reduction F = 588/392):
    stage D: 128 MSMs x 64 bases x W windows XYZZ mixed additions (6M + 2S + F);
    stages E+F: 642 twiddle multiplications x (1 + 128 doublings (2M + 3S + F) + ~43 + 7 additions (10M + 4S + F) + 8 beta-muls)
    + 14 x 64 butterfly additions."""
    w = (255 + window_bits) // window_bits
    S, F = 301 / 392, 588 / 392
    madd, dbl, add = 6 + 2 * S + F, 2 + 3 * S + F, 10 + 4 * S + F
    return 128 * 64 * w * madd + 642 * (129 * dbl + 50 * add + 8) + 14 * 64 * 1.5 * add


def synth_blobs(n, seed):
    """n valid blobs: every 32-byte element uniform in [0, 2^254) < r, deterministic in (seed)."""
    import numpy as np
    rng = np.random.RandomState(seed)
    a = rng.randint(0, 256, size=(n, 4096, 32), dtype=np.uint8)
    a[:, :, 0] &= 0x3F
    return a.reshape(n, BYTES_PER_BLOB)


def cpu_baseline(blobs, budget_s=15.0, gpu_first=None):
    """Time the CPU oracle (C restatement of the reference algorithm: FK20, width-8 window tables, batched
    affine additions) on this box's host cores.  Two CPU configurations are timed on a bounded sample:
      * blob-parallel: one single-threaded prover per worker thread, distinct blobs, all host cores busy
        (the strongest CPU arrangement for a throughput metric) -> cpu_baseline.value
      * rayon-like: one blob at a time with OpenMP threads over the axes maybe_rayon parallelises
        (fk20/batch_toeplitz.rs:50,68,95,104,114; polynomial/src/fft.rs:72,119) -> cpu_baseline.rayon_like
    """
    import concurrent.futures as cf
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # a cgroup CPU quota caps the usable cores below the visible count (the GPU boxes: 256 visible, quota 16)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per))))
    except Exception:
        pass
    import oracle_lib
    try:
        # a native build of the same sources for this host (the prebuilt liboracle.so is generic x86-64)
        native = os.path.join("/tmp", "liboracle_native_%d.so" % os.getuid())
        src = [os.path.join(ROOT, "oracle", f) for f in ("field.c", "g1.c", "pairing.c", "sha256.c", "kzg.c")]
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-std=gnu11", "-shared", "-o", native] + src,
                              stderr=subprocess.DEVNULL)
        oracle_lib._SO = native
    except Exception:
        pass
    from oracle_lib import Oracle
    # --- rayon-like (intra-blob parallel), best of a few thread counts
    rayon = {}
    for th in sorted({min(cores, t) for t in (8, 16, 32)}):
        o = Oracle(use_precomp=True, threads=th)
        o.compute_cells_and_kzg_proofs(blobs[0])
        n, t0 = 0, time.time()
        while time.time() - t0 < budget_s / 6 and n < 200:
            o.compute_cells_and_kzg_proofs(blobs[n % len(blobs)])
            n += 1
        rayon[th] = n / (time.time() - t0)
        o.close()
    best_th = max(rayon, key=rayon.get)
    # --- blob-parallel over all cores (ctypes releases the GIL; the context is read-only while computing)
    o = Oracle(use_precomp=True, threads=1)
    t1 = time.time()
    ref_cells, ref_proofs = o.compute_cells_and_kzg_proofs(blobs[0])
    single_thread = 1.0 / (time.time() - t1)
    agrees = None
    if gpu_first is not None:  # the oracle doubles as the checker here: the GPU's bytes for blob 0 of the timed batch
        agrees = (b"".join(ref_cells) == gpu_first[0]) and (b"".join(ref_proofs) == gpu_first[1])
        if not agrees:
            raise SystemExit("bench.py: GPU cells/proofs of blob 0 differ from the CPU oracle")
    workers = cores
    per_worker = 2
    def work(w):
        for k in range(per_worker):
            o.compute_cells_and_kzg_proofs(blobs[(w * per_worker + k) % len(blobs)])
        return per_worker
    t0 = time.time()
    with cf.ThreadPoolExecutor(max_workers=workers) as ex:
        done = sum(ex.map(work, range(workers)))
    dt = time.time() - t0
    o.close()
    par = done / dt
    best_value, best_cores = (par, workers) if par >= rayon[best_th] else (rayon[best_th], best_th)
    return {"value": best_value, "unit": "blobs/s", "cores": best_cores, "kind": "port",
            "sample": f"best of two CPU arrangements of the C oracle (width-8 tables, portable __int128 field arithmetic -- not "
                      f"blst assembly): (a) {done} x compute_cells_and_kzg_proofs over {len(blobs)} distinct synthetic blobs in "
                      f"{dt:.1f} s with one single-threaded prover per host thread ({workers} threads) = {par:.1f} blobs/s; "
                      f"(b) one blob at a time with OpenMP over the maybe_rayon axes = {rayon[best_th]:.1f} blobs/s at "
                      f"{best_th} threads; single thread = {single_thread:.2f} blobs/s",
            "gpu_matches_oracle_on_blob0": agrees,
            "blob_parallel": {"value": par, "threads": workers},
            "single_thread": single_thread,
            "rayon_like": {"value": rayon[best_th], "threads": best_th, "all": {str(k): round(v, 2) for k, v in rayon.items()},
                           "note": "one blob at a time, OpenMP over the maybe_rayon axes"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blobs-per-gpu", type=int, default=int(os.environ.get("KZG_BENCH_BLOBS", "2048")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency-probe", action="store_true",
                    help="skip the 1-blob latency launches (profiling runs: keeps rocprofv3's per-kernel averages to full-batch launches)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    kzg = importlib.import_module("rust-eth-kzg_amd")
    sharding = importlib.import_module("rust-eth-kzg_amd.sharding")
    t_ctx = time.perf_counter()
    ctx = kzg.DASContext(use_precomp=True, device=local_rank)  # no CPU fallback: raises/aborts without the HIP path
    t_ctx = time.perf_counter() - t_ctx

    B = args.blobs_per_gpu
    blobs_h = synth_blobs(B, seed=0x4B5A47 + rank)
    d_blobs = torch.from_numpy(blobs_h).to(dev)
    d_cells = torch.empty(B * CELLS * BYTES_PER_CELL, dtype=torch.uint8, device=dev)
    d_proofs = torch.empty(B * CELLS * 48, dtype=torch.uint8, device=dev)
    d_all_proofs = torch.empty(world * B * CELLS * 48, dtype=torch.uint8, device=dev) if world > 1 else None
    stream = torch.cuda.Stream(device=dev)  # a real (non-null) HIP stream: kernels are enqueued on it without host syncs

    def step():
        with torch.cuda.stream(stream):
            ctx.compute_cells_and_kzg_proofs_device(B, d_blobs.data_ptr(), d_cells.data_ptr(), d_proofs.data_ptr(),
                                                    want_status=False, stream=stream.cuda_stream)
            if world > 1:
                sharding.all_gather_flat(d_proofs, d_all_proofs, dist)  # the only exchange: proof vectors over RCCL/xGMI

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # correctness gate on the first run (status + the data-in-first-half invariant, fk20/prover.rs:251-275)
    st = ctx.compute_cells_and_kzg_proofs_device(B, d_blobs.data_ptr(), d_cells.data_ptr(), d_proofs.data_ptr(),
                                                 want_status=True, stream=None)
    assert st == [0] * B, "synthetic blobs rejected"
    torch.cuda.synchronize(dev)
    first_half = d_cells.view(B, 2, 64 * BYTES_PER_CELL)[:, 0, :]
    assert torch.equal(first_half, d_blobs), "cells[0..63] != blob"

    for _ in range(args.warmup):
        step()
    fence()
    ctx.set_profiling(True)
    ctx.get_stage_times()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    stages = ctx.get_stage_times()
    ctx.set_profiling(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # single-blob latency (BASELINE.json config 2), outside the timed region
    lat = []
    for _ in range(0 if args.no_latency_probe else 3):
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        with torch.cuda.stream(stream):
            ctx.compute_cells_and_kzg_proofs_device(1, d_blobs.data_ptr(), d_cells.data_ptr(), d_proofs.data_ptr(),
                                                    want_status=False, stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        lat.append(time.perf_counter() - t1)

    if rank == 0:
        total_blobs = B * world * args.steps
        value = total_blobs / dt
        # dominant kernel = the one with the largest accumulated HIP-event time
        dom = max(stages, key=lambda s: stages[s][0])
        dom_ms, dom_launches = stages[dom]
        per_launch_s = dom_ms * 1e-3 / max(1, dom_launches)
        # algorithmic bytes per launch of each kernel family (DESIGN.md "kernels"):
        alg_bytes = {
            "blob_to_coeffs": B * (BYTES_PER_BLOB + 4096 * 32),
            "coeffs_to_cells": B * (4096 * 32 * 2 + 8192 * 32),
            "fk20_scalars": B * (4096 * 32 + 128 * 64 * 32),
            # scalars in + one 96-B table entry per (scalar, window) + 128 Jacobian sums out
            "msm_fixed": B * (128 * 64 * 32 + 128 * 64 * ((255 + ctx.window_bits()) // ctx.window_bits()) * 112 + 128 * 168),
            # one radix-2 layer: 64 butterflies x (2 points in, 2 points out) x 168 B per blob
            "g1_ifft": B * 64 * 4 * 168,
            "g1_fft": B * 64 * 4 * 168,
            "compress": B * 128 * (168 + 48),
            # compiled linear map: ~3.9 k point operations per blob (350 constant multiplications, 3.2 k additions, 0.4 k
            # doubling runs), each reading one or two 168-B points and writing one, over its ~74 launches
            "g1_linmap": B * 3900 * 3 * 168 // 74,
        }[dom]
        achieved = alg_bytes / per_launch_s / 1e9
        # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE, KB units),
        # valid only for the configuration they were collected on
        traffic = None
        try:
            import glob
            pm = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_b2048_w14.json")))[-1]))["kernels"]  # newest round
            key = {"msm_fixed": "void kzg::k_msm_fixed<14>", "g1_ifft": "kzg::k_g1_twiddle_mul", "g1_fft": "kzg::k_g1_twiddle_mul"}.get(dom)
            if key in pm and B == 2048 and ctx.window_bits() == 14:
                # gfx950 correction (MI355X_MICROARCH.md, calibrated for this kernel's 16-B-per-lane gathers in
                # profiles/r1f_calib_fetch.log): FETCH_SIZE tallies every 128-B line request at 64 B -> double it; WRITE_SIZE is exact
                traffic = (2.0 * pm[key]["FETCH_SIZE_per_launch_max"] + pm[key]["WRITE_SIZE_per_launch_max"]) * 1024.0
        except Exception:
            pass
        stage_ms_per_step = {s: round(stages[s][0] / args.steps, 3) for s in stages}
        # integer-VALU view (the bound that actually binds)
        mac_rate = value * 1.0e9 / 1e9  # reference-algorithm count: ~1.0e9 32x32 MACs per blob
        mul_eq = fp_mul_eq_per_blob(ctx.window_bits())
        mul_rate = value * mul_eq / 1e9
        out = {
            "metric": "blobs/sec compute_cells_and_kzg_proofs (4096-pt blob)",
            "value": value, "unit": "blobs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "dtype_note": "381-bit Fp as 14x29-bit / 12x32-bit limbs, 255-bit Fr as 8x32-bit limbs, Montgomery integers; exact arithmetic", "data": "synthetic",
            "config": {"workload": f"compute_cells_and_kzg_proofs, batch-saturated: {B} synthetic blobs per GPU per step "
                                   f"(config 2's single blob is reported as single_blob_latency_ms)",
                       "blobs_per_gpu": B, "use_precomp": True, "window_bits": ctx.window_bits(),
                       "table_GB": round(ctx.table_bytes() / 1e9, 2),
                       "exchange": "RCCL all-gather of proofs per step" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_ms": per_launch_s * 1e3, "launches_per_step": dom_launches // max(1, args.steps),
                         "note": "the path is integer-VALU bound, not HBM bound (SURVEY.md 8d); see roofline_valu"},
            "roofline_valu": {"bound": "valu-int", "achieved": mul_rate, "peak": FP_MUL_PEAK_G, "unit": "G Fp-mul/s",
                              "frac": mul_rate / FP_MUL_PEAK_G, "fp_mul_eq_per_blob": round(mul_eq),
                              "reference_algorithm_mac_rate_G": mac_rate, "mac_issue_peak_G": VALU_INT_PEAK_GOPS,
                              "note": "blobs/s x Fp-multiplication equivalents this build executes per blob / measured "
                                      "multiplication ceiling; reference_algorithm_mac_rate = blobs/s x 1.0e9 MACs (SURVEY 8d)"},
            "whole_path_hbm_frac": value * ALG_BYTES_PER_BLOB / 1e9 / HBM_PEAK_GBS,
            "stage_ms_per_step": stage_ms_per_step,
            "single_blob_latency_ms": (min(lat) * 1e3) if lat else None,
            "context_creation_s": round(t_ctx, 2),
        }
        if not args.no_cpu_baseline and world == 1:
            gpu_first = (bytes(d_cells[:CELLS * BYTES_PER_CELL].cpu().numpy()), bytes(d_proofs[:CELLS * 48].cpu().numpy()))
            out["cpu_baseline"] = cpu_baseline([bytes(blobs_h[i].tobytes()) for i in range(min(B, 16))], gpu_first=gpu_first)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
