"""A/B of library builds on ONE GPU box, alternated (boxes differ by several per cent, and the point kernels run power-limited: the
clock a build gets depends on what ran just before).  Each run is a fresh process: the 2048-blob device-resident step under the
per-stage HIP events of the library, stage times printed as one JSON line; package power and shader clock are sampled beside it.

    python tools/exp_variants.py <rounds> <table setting> <name=lib.so[@VAR=value ...]> [<name=lib.so> ...] [--check]

table setting: "max", a budget in GB, or "w15" (ETH_KZG_AMD_GLV_WINDOW=15 with whatever memory that takes).
--check: blob 0's proofs are compared between the first build and every other one (a timing-only build says `differs`).
Writes gpurun_out/exp_variants_<tag>.log.  The child mode (one build, one process) is `--child`."""
import hashlib
import importlib
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    import torch
    kzg = importlib.import_module("rust-eth-kzg_amd")
    torch.zeros(1, device="cuda")
    n = int(os.environ.get("EXP_BLOBS", "2048"))
    steps = int(os.environ.get("EXP_STEPS", "8"))
    ctx = kzg.DASContext(use_precomp=True)
    g = torch.Generator(device="cuda").manual_seed(11)
    blobs = torch.randint(0, 256, (n, 131072), dtype=torch.uint8, device="cuda", generator=g)
    blobs.view(n, 4096, 32)[:, :, 0] &= 0x3F
    cells = torch.empty(n * 128 * 2048, dtype=torch.uint8, device="cuda")
    proofs = torch.empty(n * 128 * 48, dtype=torch.uint8, device="cuda")
    run = lambda: ctx.compute_cells_and_kzg_proofs_device(n, blobs.data_ptr(), cells.data_ptr(), proofs.data_ptr(), want_status=False)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) * 1e3 / steps
    ctx.set_profiling(True)
    ctx.get_stage_times()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    st = ctx.get_stage_times()
    ctx.set_profiling(False)
    digest = hashlib.sha256(proofs[:128 * 48].cpu().numpy().tobytes()).hexdigest()[:16]
    print("RESULT " + json.dumps({"window_bits": ctx.window_bits(), "table_GB": round(ctx.table_bytes() / 1e9, 1), "step_ms": round(step_ms, 3),
                                  "stages_ms": {k: round(v[0] / steps, 3) for k, v in st.items() if v[0] > 0}, "proofs0": digest}))
    ctx.close()


def sample_smi(stop, rows):
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            sclk = [l for l in out.splitlines() if "sclk" in l]
            pw = [l for l in out.splitlines() if "Package Power" in l or "Socket Power" in l]
            rows.append((time.time(), int(sclk[0].split("(")[1].split("Mhz")[0]) if sclk else None, float(pw[0].split(":")[-1]) if pw else None))
        except Exception:  # noqa: BLE001
            pass
        time.sleep(0.25)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rounds, setting, builds = int(args[0]), args[1], [a.split("=", 1) for a in args[2:]]
    check = "--check" in sys.argv
    env0 = dict(os.environ)
    if setting == "w15":
        env0["ETH_KZG_AMD_GLV_WINDOW"] = "15"
        env0["ETH_KZG_AMD_TABLE_GB"] = "max"
    else:
        env0["ETH_KZG_AMD_TABLE_GB"] = setting
    tag = setting + "_" + "_".join(n for n, _ in builds)
    log = open(os.path.join(ROOT, "gpurun_out", f"exp_variants_{tag}.log"), "w")
    results = {n: [] for n, _ in builds}
    for r in range(rounds):
        for name, lib in builds:
            lib, *sets = lib.split("@")  # name=lib.so@VAR=value@VAR2=value: environment of this build's runs only
            env = dict(env0, ETH_KZG_AMD_LIB=os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib)
            for kv in sets:
                k, v = kv.split("=", 1)
                env[k] = v
            stop, rows = threading.Event(), []
            th = threading.Thread(target=sample_smi, args=(stop, rows))
            th.start()
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], capture_output=True, text=True, env=env, timeout=900)
            stop.set()
            th.join()
            line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                log.write(f"{name} round {r}: FAILED\n{out.stdout[-2000:]}\n{out.stderr[-2000:]}\n")
                print(name, "FAILED", out.stderr[-500:])
                continue
            res = json.loads(line[0][7:])
            busy = [x for x in rows if x[2] and x[2] > 900]  # samples taken while the point kernels ran
            res["smi_busy"] = {"samples": len(busy), "sclk_median": sorted(x[1] for x in busy)[len(busy) // 2] if busy else None,
                               "W_median": sorted(x[2] for x in busy)[len(busy) // 2] if busy else None}
            results[name].append(res)
            msg = f"{name:10s} round {r}: " + json.dumps(res)
            print(msg)
            log.write(msg + "\n")
            log.flush()
    first = builds[0][0]
    for name, rs in results.items():
        if not rs:
            continue
        med = lambda xs: sorted(xs)[len(xs) // 2]
        summary = {"build": name, "step_ms_median": med([x["step_ms"] for x in rs]),
                   "stages_ms_median": {k: med([x["stages_ms"].get(k, 0) for x in rs]) for k in rs[0]["stages_ms"]}}
        if check and results[first]:
            summary["proofs_of_blob0"] = "same as " + first if rs[0]["proofs0"] == results[first][0]["proofs0"] else "DIFFERS from " + first
        msg = "SUMMARY " + json.dumps(summary)
        print(msg)
        log.write(msg + "\n")
    log.close()


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        main()
