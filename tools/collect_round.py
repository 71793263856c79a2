"""Copies the summaries of a tools/measure_round.sh run (gpurun_out/final) into profiles/ under a tag and rebuilds
profiles/pmc_traffic.json.  Usage: python tools/collect_round.py r2          (here, after the gpurun call)
                                   python tools/collect_round.py r2 --pmc-only   (on the GPU box, by measure_round.sh: only the stamped
                                   PMC file, so that the bench runs that follow on the same box carry roofline.traffic)"""
import csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

tag = sys.argv[1]
pmc_only = "--pmc-only" in sys.argv[2:]
F, P = os.path.join(ROOT, "gpurun_out", "final"), os.path.join(ROOT, "profiles")
for src, dst in () if pmc_only else (("bench.json", "bench.json"), ("bench_bf16.json", "bench_gemm_bf16.json"), ("pmc_sq.txt", "pmc_sq.txt"), ("step_timeline.txt", "step_timeline.txt"), ("traffic_by_family.txt", "traffic_by_family.txt"),
                 ("bench_native_rccl_eager.json", "bench_native_rccl_eager.json"),
                 ("bench_native_rccl_graph.json", "bench_native_rccl_graph.json"), ("bench_graph.json", "bench_graph.json"), ("bench_eager.json", "bench_eager.json"),
                 ("bench_deep16_split.json", "bench_deep16_split.json"), ("bench_under_rocprof.json", "bench_under_rocprof.json"),
                 ("conc_kernel_stats.csv", "bench_kernel_stats.csv"), ("serial_bench.json", "serial_bench.json"),
                 ("serial_kernel_stats.csv", "serial_bench_kernel_stats.csv"), ("forward_bench.json", "forward_bench.json"),
                 ("fwd_kernel_stats.csv", "forward_kernel_stats.csv"), ("host_phases.txt", "host_phases.txt"), ("next_rows_bench.txt", "next_rows_bench.txt"),
                 ("conv_ablation.txt", "conv_ablation.txt"), ("dma_issue_microbench.txt", "dma_issue_microbench.txt"),
                 ("h3u_sweep.txt", "h3u_threshold_sweep.txt"), ("h3u_ablation.txt", "h3u_ablation.txt"), ("h3u_stage_timeline.txt", "h3u_stage_timeline.txt"),
                 ("round4_vs_round5_same_box.txt", "round4_vs_round5_same_box.txt"), ("prev_vs_cur_same_box.txt", "previous_round_vs_this_same_box.txt"),
                 ("upt_ab.txt", "upt_ab_same_box.txt"), ("wgrad_xcd_ab.txt", "wgrad_xcd_ab_same_box.txt"),
                 ("evop_ab.txt", "evop_eval_ab.txt"), ("git_state.txt", "git_state.txt"), ("gpu_tests.txt", "gpu_tests.txt")):
    if not os.path.exists(os.path.join(F, src)):
        continue
    stamp = os.path.join(F, "git_state.txt")        # (gpurun merges into gpurun_out/ without deleting: leftovers of an earlier round are older than this run's stamp)
    if os.path.exists(stamp) and os.path.getmtime(os.path.join(F, src)) < os.path.getmtime(stamp) - 5:
        continue
    if src.endswith(".json"):           # the bench line only (RCCL prints its banner into the same stream on some paths)
        lines = [ln for ln in open(os.path.join(F, src)).read().splitlines() if ln.startswith("{")]
        if lines:
            open(os.path.join(P, f"{tag}_{dst}"), "w").write(lines[-1] + "\n")
            continue
    shutil.copy(os.path.join(F, src), os.path.join(P, f"{tag}_{dst}"))
# the PMC runs execute 3 steps (1 warm-up + 2): launches / 3 = launches per step.  FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE
# under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section): doubled.
def lib_name(k):
    """rocprofv3's kernel name -> the name the library's HIP-event profiler (and bench.py's roofline) uses for the same kernel:
    the split kernels drop their boolean template arguments (bf16 is spelled out)."""
    if "<" not in k:
        return k
    base, args = k.split("<", 1)
    a = [t.strip() for t in args.rstrip(">").split(",")]
    if base == "conv_h3d_kernel":
        mode = a[5] if len(a) > 5 else "0"           # (BSUM template argument: 1 / 2 = BatchNorm-backward sums in the epilogue, 3 = UPT)
        return "%s<%s, %s, %s%s>" % (base, a[0], a[1], a[2], ", bf16" if len(a) > 3 and a[3] == "true" else ", evop" if len(a) > 4 and a[4] == "true" else
                                     ", upt" if mode == "3" else ", bsum" if mode in ("1", "2") else "")
    if base == "conv_h3u_kernel":
        return "%s<%s>" % (base, a[0])
    if base == "wgrad_h3d_kernel":
        return "%s<%s, %s%s>" % (base, a[0], a[1], ", bf16" if a[3] == "true" else "")
    if base == "wgrad_h3_kernel":
        return "%s<%s, %s%s>" % (base, a[0], a[1], ", bf16" if len(a) > 4 and a[4] == "true" else "")
    return k


def section(name):
    raw = json.load(open(os.path.join(F, f"pmc_raw_{name}.json")))
    acc = {}
    for k, (n, fs) in raw["fetch"].items():
        wn, ws = raw["write"].get(k, [0, 0.0])
        e = acc.setdefault(lib_name(k), [0, 0.0, 0, 0.0])
        e[0] += n; e[1] += fs; e[2] += wn; e[3] += ws
    kern, whole = {}, 0.0
    for k, (n, fs, wn, ws) in acc.items():
        per = (2.0 * fs / n + (ws / wn if wn else 0.0)) * 1024.0
        kern[k] = {"launches_per_step": n / 3.0, "FETCH_SIZE_KiB": round(fs / n, 1), "WRITE_SIZE_KiB": round(ws / wn if wn else 0.0, 1),
                   "hbm_bytes_per_launch": int(per)}
        whole += per * n / 3.0
    return {"whole_step_bytes": whole, "kernels": kern}


head = section("train")
out = {"_note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of bench.py --steps 2 --warmup 1; HBM bytes per "
                "launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 correction); whole_step_bytes = sum over ALL kernels of one step, "
                "torch's own (optimizer state init, fills) included.  Top level: the headline workload (12 levels, batch 64, split GEMMs); "
                "sections: the same passes of bench.py --gemm fp32, of --gemm bf16 --layers 16 --frame 65536 --batch 32 (configs[4]) and of --mode forward (configs[1])",
       "tag": tag, "source_hash": bench.source_hash(), "whole_step_bytes": head["whole_step_bytes"], "kernels": head["kernels"], "sections": {}}
for name in ("gemm_fp32", "deep16_bf16", "eval_forward"):
    if os.path.exists(os.path.join(F, f"pmc_raw_{name}.json")):
        out["sections"][name] = section(name)
json.dump(out, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(P, "pmc_traffic.json"), os.path.join(P, f"{tag}_pmc_traffic.json"))
print("whole step HBM bytes %.3f GB" % (head["whole_step_bytes"] / 1e9), {k: round(v["whole_step_bytes"] / 1e9, 3) for k, v in out["sections"].items()})
if pmc_only:
    sys.exit(0)
for name in ("bench.json", "bench_gemm_bf16.json", "bench_deep16_bf16.json", "bench_deep16_split.json", "serial_bench.json", "forward_bench.json"):
    pth = os.path.join(P, f"{tag}_{name}")
    if not os.path.exists(pth):
        continue
    j = json.loads(open(pth).read().strip().splitlines()[-1]); r = j.get("roofline") or {}
    print(name, round(j["value"]), round(j["ms_per_step"], 3), r.get("kernel"), r.get("achieved") and round(r["achieved"], 1), r.get("frac") and round(r["frac"], 3), r.get("avg_launch_ms"))
rows = list(csv.DictReader(open(os.path.join(P, f"{tag}_serial_bench_kernel_stats.csv"))))
for r in rows[:5]:
    print("serial", r["Name"][:60], r["Calls"], "avg %.4f ms" % (float(r["AverageNs"]) / 1e6))
