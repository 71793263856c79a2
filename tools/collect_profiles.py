"""Copies the summaries of a `tools/final_measurements.sh` run (gpurun_out/final) into profiles/ under a prefix and rebuilds the PMC
traffic table.  Usage: python tools/collect_profiles.py r1_h3"""
import collections, csv, json, shutil, subprocess, sys, os
pre = sys.argv[1]; F = "gpurun_out/final"
for src, dst in (("bench.json", "bench.json"), ("bench_fp32.json", "bench_gemm_fp32.json"), ("bench_forward.json", "bench_forward.json"),
                 ("bench_under_rocprof.json", "bench_under_rocprof.json"), ("conc/conc_kernel_stats.csv", "bench_kernel_stats.csv"),
                 ("serial_bench.json", "serial_bench.json"), ("serial/serial_kernel_stats.csv", "serial_bench_kernel_stats.csv")):
    shutil.copy(f"{F}/{src}", f"profiles/{pre}_{dst}")
with open(f"profiles/{pre}_per_layer.txt", "w") as f:
    subprocess.run([sys.executable, "tools/per_layer.py", f"{F}/serial", "serial"], stdout=f, env=dict(os.environ, PYTHONPATH="."), check=True)
def agg(path, ctr):
    a = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != ctr: continue
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        a[k][0] += 1; a[k][1] += float(r["Counter_Value"])
    return a
f = agg(f"{F}/pmc_fetch/f_counter_collection.csv", "FETCH_SIZE"); w = agg(f"{F}/pmc_write/w_counter_collection.csv", "WRITE_SIZE")
old = json.load(open("profiles/r1_pmc_traffic.json"))
for k in f:
    if "h3" not in k and "mfma" not in k and "conv_first" not in k: continue
    n = f[k][0]; fs = f[k][1] / n; ws = w[k][1] / max(1, w[k][0])
    old["kernels"][k] = {"launches": n, "FETCH_SIZE_KiB": round(fs, 1), "WRITE_SIZE_KiB": round(ws, 1), "hbm_bytes_per_launch": int((2 * fs + ws) * 1024)}
json.dump(old, open("profiles/r1_pmc_traffic.json", "w"), indent=1)
for name in ("bench.json", "bench_gemm_fp32.json", "bench_forward.json", "serial_bench.json"):
    j = json.loads(open(f"profiles/{pre}_{name}").read().strip().splitlines()[-1]); r = j.get("roofline") or {}
    print(name, round(j["value"]), round(j["ms_per_step"], 3), r.get("kernel"), r.get("achieved") and round(r["achieved"], 1), r.get("frac") and round(r["frac"], 3), r.get("avg_launch_ms"))
rows = list(csv.DictReader(open(f"profiles/{pre}_serial_bench_kernel_stats.csv")))
for r in rows[:4]: print("serial", r["Name"][:50], r["Calls"], "avg %.4f ms" % (float(r["AverageNs"]) / 1e6))
