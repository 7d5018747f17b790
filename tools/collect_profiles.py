"""Copy one tools/round_profiles.sh result (gpurun_out/<src>) into the tracked profiles/<round>_* files.
usage: python tools/collect_profiles.py r3 gpurun_out/r3b"""
import glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha  # the counters are valid for these device sources only (bench.py refuses others)
rnd, src = sys.argv[1], sys.argv[2]
P = os.path.join(ROOT, "profiles")
# the sha the run recorded on the GPU box (tools/round_profiles.sh); a run without one is stamped with the current sources
_sha_file = os.path.join(src, "kernel_sources_sha.txt")
measured_sha = open(_sha_file).read().strip() if os.path.exists(_sha_file) else kernel_sources_sha()
if measured_sha != kernel_sources_sha():
    print(f"WARNING: counters were measured on kernel sources {measured_sha}, the tree is at {kernel_sources_sha()}: bench.py will refuse them")


def cp(rel, name):
    hits = sorted(glob.glob(os.path.join(src, rel), recursive=True))
    if not hits:
        print("missing", rel)
        return
    shutil.copyfile(hits[-1], os.path.join(P, f"{rnd}_{name}"))
    print(f"{rnd}_{name}  <-  {os.path.relpath(hits[-1], src)}")


cp("bench/bench_n1.json", "bench_n1.json")
cp("bench/bench_other_workloads.jsonl", "bench_other_workloads.jsonl")
cp("bench/derivatives_timing.txt", "derivatives_timing.txt")
cp("prof/bench_under_rocprof.json", "bench_under_rocprofv3.json")
cp("prof/derivs_under_rocprof.txt", "derivatives_under_rocprofv3.txt")
cp("prof/stats/**/*kernel_stats.csv", "rocprofv3_kernel_stats.csv")
cp("prof/stats_derivs/**/*kernel_stats.csv", "rocprofv3_kernel_stats_derivatives.csv")
cp("pmc_flops.json", "pmc_flops.json")
fl = os.path.join(P, f"{rnd}_pmc_flops.json")
if os.path.exists(fl):
    doc = json.load(open(fl))
    doc["kernel_sources_sha"] = measured_sha
    json.dump(doc, open(fl, "w"), indent=1)
cp("traffic_calibration.txt", "traffic_calibration.txt")
cp("bench/manifold_derivatives.txt", "manifold_derivatives.txt")
cp("strong_scaling_proxy.txt", "strong_scaling_proxy.txt")
cp("latency_mode_waves.txt", "latency_mode_waves_raw.txt")
cp("gate_f32_oracle.txt", "gate_f32_oracle.txt")
cp("stress_random_models.txt", "stress_random_models.txt")
cp("tello_acc.txt", "tello_acc_f32_vs_float_oracle.txt")
cp("prof/traffic.txt", "pmc_traffic_raw.txt")
for tag, name in (("pmc_mit_aba32", "mit_aba32"), ("pmc_mit_rnea32", "mit_rnea32"), ("pmc_tello_aba32", "tello_aba32"),
                  ("pmc_jvrc1_aba32", "jvrc1_aba32"), ("pmc_minicheetah_aba64", "minicheetah_aba64"), ("pmc_derivs", "derivatives"),
                  ("pmc_four_bar_aba32", "four_bar_aba32"), ("pmc_six_bar_aba32", "six_bar_aba32")):
    cp(f"{tag}/summary.txt", f"rocprofv3_pmc_{name}.txt")

# HBM-side traffic per launch: FETCH_SIZE counts half of the bytes read, WRITE_SIZE the bytes written (traffic_calibration.txt)
entries = []
raw = os.path.join(src, "prof", "traffic.txt")
if os.path.exists(raw):
    for line in open(raw):
        m = re.match(r"(\w+?)_(32|64)_(\w+) batch (\d+) FETCH_SIZE_KiB (\S+) WRITE_SIZE_KiB (\S+) launches (\d+) kernel void (.*)", line.strip())
        if not m:
            continue
        algo, bits, wl, batch, f, w, n, kern = m.groups()
        f, w, batch = float(f), float(w), int(batch)
        total = int((2 * f + w) * 1024)
        entries.append({"workload": wl, "algo": algo, "dtype": "f" + bits, "batch": batch, "fetch_kib_counted": f, "write_kib": w,
                        "bytes_per_launch": total, "bytes_per_state": round(total / batch, 1), "kernel": kern})
    old = os.path.join(P, f"{rnd}_pmc_traffic.json")
    comment = json.load(open(old))["_comment"] if os.path.exists(old) else "bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024"
    json.dump({"_comment": comment, "kernel_sources_sha": measured_sha, "entries": entries}, open(old, "w"), indent=1)
    print(f"{rnd}_pmc_traffic.json  <-  prof/traffic.txt ({len(entries)} entries)")
