"""Turns gpurun_out/r05p/* (tools/r5_profiles.sh) into the committed profiles/r05_* files (tools; not part of the product)."""
import csv, hashlib, json, os, shutil
R = "gpurun_out/r05p"
cp = {
    "bench_c2_kernel_stats.csv": "r05_kernel_stats_bench_c2.csv", "bench_c2_run.json": "r05_bench_c2_rocprof_run.json",
    "bench_c2_serial_kernel_stats.csv": "r05_kernel_stats_bench_c2_serial.csv", "bench_c2_serial_run.json": "r05_bench_c2_serial_rocprof_run.json",
    "bench_c2_noise_kernel_stats.csv": "r05_kernel_stats_bench_c2_noise.csv", "bench_c2_noise_run.json": "r05_bench_c2_noise_rocprof_run.json",
    "bench_c3_kernel_stats.csv": "r05_kernel_stats_bench_c3_64pages.csv", "bench_c3_run.json": "r05_bench_c3_64pages_rocprof_run.json",
    "bench_c2_default_300steps.json": "r05_bench_c2_default_300steps.json", "bench_driver_cmd_20steps.json": "r05_bench_driver_cmd_20steps.json",
    "bench_c2_force_gather.json": "r05_bench_c2_force_gather.json", "bench_c2_stall3.json": "r05_bench_c2_stall3ms.json", "bench_c2_stall6.json": "r05_bench_c2_stall6ms.json",
    "bench_c2_stall3_depth1.json": "r05_bench_c2_stall3ms_depth1.json", "bench_c2_depth1.json": "r05_bench_c2_depth1.json",
    "bench_c4_8192pages_1gpu.json": "r05_bench_c4_8192pages_1gpu.json", "bench_c3_stream_1024pages.json": "r05_bench_c3_stream_1024pages.json",
    "timeline_bench_c2.log": "r05_timeline_bench_c2.log", "bench_thr_sweep.log": "r05_bench_thr_sweep.log",
    "fetch_pmc.csv": "r05_pmc_FETCH_SIZE_kbench_c2.csv", "write_pmc.csv": "r05_pmc_WRITE_SIZE_kbench_c2.csv", "mfma_pmc.csv": "r05_pmc_mfma_busy_kbench_c2.csv",
    "insts_pmc.csv": "r05_pmc_insts_kbench_c2.csv", "cu_busy_c2.log": "r05_cu_busy_c2.log", "cu_busy_c3.log": "r05_cu_busy_c3.log",
    "kprof_c2_alone.log": "r05_kernels_alone_c2.log", "kprof_c3_alone.log": "r05_kernels_alone_c3.log", "cli_e2e_4096_pgm.json": "r05_cli_e2e_4096_pgm.json",
}
for a, b in cp.items():
    if os.path.exists(os.path.join(R, a)):
        shutil.copy(os.path.join(R, a), os.path.join("profiles", b))
    else:
        print("missing", a)

def avg(f, kern, ctr):
    v = [float(x["Counter_Value"]) for x in csv.DictReader(open(f)) if kern in x["Kernel_Name"] and x["Counter_Name"] == ctr]
    return sum(v) / len(v), len(v)

def dur_ms(f, kern, ctr):
    v = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) for x in csv.DictReader(open(f)) if kern in x["Kernel_Name"] and x["Counter_Name"] == ctr]
    return sum(v) / len(v) / 1e6

src = ["font_ocr_amd/csrc/hip/scan_mfma2.hip", "font_ocr_amd/csrc/hip/mfma_common.h", "font_ocr_amd/csrc/hip/scan_mfma.hip"]
h = hashlib.sha256()
for p in src:
    h.update(open(p, "rb").read())
fetch, n1 = avg(R + "/fetch_pmc.csv", "scan_mfma2s", "FETCH_SIZE")
write, n2 = avg(R + "/write_pmc.csv", "scan_mfma2s", "WRITE_SIZE")
t = {"kernel": "scan_mfma2s_kernel<2,2,4,16>", "workload_key": {"pages": 128, "r_w": 608, "r_h": 720, "templates": 380},
     "kernel_source_sha16": h.hexdigest()[:16], "kernel_sources": src,
     "FETCH_SIZE_KB_per_launch": round(fetch, 1), "WRITE_SIZE_KB_per_launch": round(write, 1), "launches_averaged": [n1, n2],
     "traffic_bytes": int(round((2 * fetch + write) * 1024)),
     "note": "separate rocprofv3 --pmc passes over tools/kbench.py (tools/r5_profiles.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md (HBM section); unit KB",
     "sources": ["profiles/r05_pmc_FETCH_SIZE_kbench_c2.csv", "profiles/r05_pmc_WRITE_SIZE_kbench_c2.csv"]}
json.dump(t, open("profiles/r05_traffic.json", "w"), indent=1)
print("traffic", t["traffic_bytes"], t["FETCH_SIZE_KB_per_launch"], t["WRITE_SIZE_KB_per_launch"], "hash", t["kernel_source_sha16"])
out = {c: round(avg(R + "/mfma_pmc.csv", "scan_mfma2s", c)[0]) for c in ("GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES")}
d = dur_ms(R + "/mfma_pmc.csv", "scan_mfma2s", "GRBM_GUI_ACTIVE")
print("kbench scan pmc", out, "mfma pipe busy", round(out["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (out["GRBM_GUI_ACTIVE"] / 8), 4), "kernel ms", round(d, 3), "clock GHz", round(out["GRBM_GUI_ACTIVE"] / 8 / d / 1e6, 3))
ins = {c: avg(R + "/insts_pmc.csv", "scan_mfma2s", c)[0] for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY")}
print("insts", {k: round(v) for k, v in ins.items()}, "non-MFMA VALU per item", round((ins["SQ_INSTS_VALU"] - ins["SQ_INSTS_MFMA"]) / (ins["SQ_INSTS_MFMA"] / 192), 1),
      "wait frac", round(ins["SQ_WAIT_ANY"] / ins["SQ_WAVE_CYCLES"], 3))
for f in ("bench_c2_run", "bench_c2_serial_run", "bench_c2_noise_run", "bench_c3_run", "bench_c3_stream_1024pages", "bench_c4_8192pages_1gpu", "bench_c2_default_300steps", "bench_driver_cmd_20steps",
          "bench_c2_force_gather", "bench_c2_stall3", "bench_c2_stall6", "bench_c2_stall3_depth1", "bench_c2_depth1"):
    d = json.load(open(f"{R}/{f}.json")); r = d["roofline"]; st = d.get("step_stats", {})
    print(f, d["value"], d["ms_per_step"], "kernel", r["avg_kernel_ms"], "frac", r["frac"], r.get("frac_issued"), r.get("frac_whole_step"), r.get("isolated_avg_kernel_ms"), r.get("frac_isolated"),
          "e2e", d.get("e2e_value_incl_h2d_pipelined"), "c3", d.get("c3_value"), "c4s", d.get("c4_stream_value"), "redone", d["size_estimates"]["batches_redone_exact"],
          "dev p50/max", st.get("device_interval_ms_p50"), st.get("device_interval_ms_max"), "host gap", st.get("longest_host_gap_ms"))
for f in ("bench_c2", "bench_c2_serial", "bench_c3", "bench_c2_noise"):
    rows = list(csv.DictReader(open(f"{R}/{f}_kernel_stats.csv")))
    for r_ in rows:
        if "scan_mfma2s" in r_["Name"] or "verify" in r_["Name"] or "stats_kernel" in r_["Name"] or "row_sort" in r_["Name"]:
            print(f, r_["Name"].split("(")[0][-44:], "rocprofv3 avg ms", round(float(r_["AverageNs"]) / 1e6, 4), "calls", r_["Calls"])
