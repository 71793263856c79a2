"""CPU checks of the measurement plumbing (no GPU, no timing): the roofline object bench.py builds from per-kernel rows, the PMC
sections it attaches, the staleness guard of profiles/pmc_traffic.json, and the per-family traffic table over the committed profiles."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _rows():
    return [
        {"kernel": "conv_h3d_kernel<5, 3, 1>", "launches": 16, "ms": 1.6, "flops": 16 * 20.0e9, "bytes": 16 * 175.0e6},
        {"kernel": "wgrad_h3d_kernel<5, 3>", "launches": 12, "ms": 0.8, "flops": 12 * 15.0e9, "bytes": 12 * 100.0e6},
        {"kernel": "conv_mfma_kernel<15, 3, 1>", "launches": 8, "ms": 0.1, "flops": 8 * 1.0e9, "bytes": 8 * 1.0e6},
        {"kernel": "pass_a_kernel<UP>", "launches": 20, "ms": 0.7, "flops": 0.0, "bytes": 2.8e9},
        {"kernel": "pass_a_kernel<ENC>", "launches": 22, "ms": 0.6, "flops": 0.0, "bytes": 2.6e9},
        {"kernel": "prep_h3_kernel<3>", "launches": 20, "ms": 0.4, "flops": 0.0, "bytes": 1.9e9},
    ]


def test_roofline_object_of_the_contract():
    pmc = {"kernels": {"conv_h3d_kernel<5, 3, 1>": {"hbm_bytes_per_launch": 191000000}}, "whole_step_bytes": 18.0e9, "why": "unit test"}
    r = bench.roofline_of(_rows(), 2, pmc, 6.37e9)
    assert r["kernel"] == "conv_h3d_kernel<5, 3, 1>" and r["bound"] == "mfma" and r["unit"] == "TFLOP/s"
    assert r["avg_launch_ms"] == pytest.approx(0.1) and r["achieved"] == pytest.approx(200.0)          # 20 GFLOP in 0.1 ms
    assert r["peak"] == pytest.approx(2500.0 / 3.0) and r["frac"] == pytest.approx(0.24)               # three f16 passes per product
    assert r["traffic"] == 191000000 and r["traffic_whole_step"]["ratio"] == pytest.approx(18.0 / 6.37)
    assert r["launches_per_step"] == 8 and r["mfma_kernels_ms_per_step"] == pytest.approx(1.25)
    fam = {m["kernel"]: m for m in r["memory_bound_kernels"]}
    assert set(fam) == {"pass_a_kernel", "prep_h3_kernel"} and fam["pass_a_kernel"]["launches_per_step"] == 21
    assert fam["pass_a_kernel"]["algorithmic_GBps"] == pytest.approx(5.4e9 / 1.3e-3 / 1e9)
    assert bench.gemm_peak("conv_h3d_kernel<5, 3, 1, bf16>")[0] == 2500.0 and bench.gemm_peak("conv_mfma_kernel<5, 4, 4>")[0] == 157.3
    # a kernel the PMC file does not know: traffic null, the reason carried along
    r2 = bench.roofline_of(_rows(), 2, {"kernels": None, "why": "no pass"}, 6.37e9)
    assert r2["traffic"] is None and r2["traffic_whole_step"] is None and r2["traffic_source"] == "no pass"


def test_pmc_file_sections_and_staleness_guard(monkeypatch):
    """profiles/pmc_traffic.json: the headline workload at the top, the extras' own PMC passes under `sections`; bench.py only
    attaches it while its stamp equals the hash of the kernel sources (after a change under csrc/ the file is refused until
    tools/measure_round.sh + tools/collect_round.py have re-measured it) - checked here with the stamp the file carries."""
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert pmc["whole_step_bytes"] > 1.0e10 and any(k.startswith("conv_h3d_kernel<5, 3, 1") for k in pmc["kernels"])
    assert set(pmc["sections"]) >= {"gemm_fp32", "deep16_bf16"}
    monkeypatch.setattr(bench, "source_hash", lambda: pmc["source_hash"])
    assert bench.load_pmc_traffic()["kernels"] is not None
    for section in ("gemm_fp32", "deep16_bf16"):
        sub = bench.load_pmc_traffic(section)
        assert sub["kernels"] is not None and sub["whole_step_bytes"] > 1.0e10, (section, sub.get("why"))
    assert bench.load_pmc_traffic("no-such-section")["kernels"] is None
    monkeypatch.setattr(bench, "source_hash", lambda: "0" * 16)
    stale = bench.load_pmc_traffic()
    assert stale["kernels"] is None and "stale" in stale["why"]


def test_traffic_table_runs_over_the_committed_profiles(tmp_path):
    line = open(os.path.join(ROOT, "profiles", "r3_bench.json")).read().strip().splitlines()[-1]
    assert json.loads(line)["roofline"]["top5"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "traffic_table.py"), os.path.join(ROOT, "profiles", "r3_bench.json")],
                         capture_output=True, text=True, check=True).stdout
    assert "conv_h3d_kernel" in out and "pass_a_kernel" in out and "all kernels" in out


def test_source_hash_ignores_comments_and_white_space():
    """a comment edit after the PMC pass must not void the round's traffic figures; a code edit must."""
    a = "int f(int x) { return x + 1; }   // plus one\n/* block\n   comment */ const char* s = \"// kept /* kept\";\n"
    b = "int f(int x)\n{\n    return x + 1;\n}\nconst char* s = \"// kept /* kept\";   // another remark\n"
    assert bench._code_only(a) == bench._code_only(b).replace("( int", "(int")
    assert bench._code_only(a) != bench._code_only(a.replace("x + 1", "x + 2"))
    assert bench._code_only(a) != bench._code_only(a.replace("// kept", "// dropped"))          # string literals are code
    assert "plus one" not in bench._code_only(a) and "comment" not in bench._code_only(a).replace("a comment", "")


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_multi_rank_branches_run_on_gloo_and_the_emulator(tmp_path, world):
    """`torchrun --nproc-per-node N bench.py --gpus N` green by construction without a multi-GPU node - the world > 1
    branches of bench.py (process group, GradSync attach, 1/world in the Adam step, max-over-ranks timing, the gradient_exchange
    record with what every rank ran, matched profiled passes) driven at world 2, 4 and 8 (the node's size, BASELINE configs[3]) over gloo
    with the kernels on the CPU emulator (WUNET_BENCH_EMU test hook).  Unmeasured on hardware: RCCL has only seen one rank."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, WUNET_BENCH_EMU="1", OMP_NUM_THREADS="1" if world > 2 else "2")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
                        "--layers", "3", "--frame", "256", "--batch", "2"], env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                         # rank 0 prints ONE line
    j = json.loads(lines[0])
    assert j["n_gpus"] == world and j["steps"] == 2 and j["scaling"] == "weak" and j["config"]["global_batch"] == 2 * world
    assert "NOT A MEASUREMENT" in j["data"] and j["roofline"] is None and j["cpu_baseline"] is None
    ex = j["gradient_exchange"]
    assert ex["world"] == world and ex["transport"].startswith("torch.distributed gloo") and ex["scale"] == "1/world in the Adam step"
    assert len(ex["bucket_bytes"]) == 4 and len(ex["allreduce_exposed_ms_per_rank"]) == world
    assert [r["rank"] for r in ex["per_rank"]] == list(range(world))
    assert all(r["world_seen"] == world and r["step_launch"] == "eager launches" and r["transport"] == ex["transport"] for r in ex["per_rank"])
    assert p.stdout.count("\n{") + p.stdout.startswith("{") == 1          # nothing but rank 0's line on stdout
    import importlib
    from conftest import PKG_NAME
    plan = importlib.import_module(PKG_NAME + ".plan")
    numel = sum(co * ci_ * k + 3 * co for ci_, co, k in plan.conv_layer_shapes(3, 24)) + 24 + 1 + 1
    assert sum(ex["bucket_bytes"]) == 4 * numel
    assert j["value"] > 0 and j["final_loss"] == j["final_loss"]
    # the headline shape's buckets (host logic only): four buckets, 40.53 MB together
    parallel = importlib.import_module(PKG_NAME + ".parallel")
    numels = []
    for c_in, c_out, k in plan.conv_layer_shapes(12, 24):
        numels += [c_out * c_in * k, c_out, c_out, c_out]
    numels += [25, 1]
    rs = parallel.bucket_ranges(numels, 25, 4)
    assert len(rs) == 4 and abs(sum(4 * (fe - fb) for _, _, fb, fe in rs) / 1e6 - 40.53) < 0.01
