"""Turns gpurun_out/r03/* (tools/r3_profiles.sh) into the committed profiles/r03_* files (tools; not part of the product)."""
import collections, csv, hashlib, json, os, shutil
R = "gpurun_out/r03"
cp = {
    "bench_c2_kernel_stats.csv": "r03_kernel_stats_bench_c2.csv", "bench_c2_run.json": "r03_bench_c2_rocprof_run.json",
    "bench_c2_serial_kernel_stats.csv": "r03_kernel_stats_bench_c2_serial.csv", "bench_c2_serial_run.json": "r03_bench_c2_serial_rocprof_run.json",
    "bench_c2_noise_kernel_stats.csv": "r03_kernel_stats_bench_c2_noise.csv", "bench_c2_noise_run.json": "r03_bench_c2_noise_rocprof_run.json",
    "bench_c3_kernel_stats.csv": "r03_kernel_stats_bench_c3_64pages.csv", "bench_c3_run.json": "r03_bench_c3_64pages_rocprof_run.json",
    "bench_c4_2048pages_1gpu.json": "r03_bench_c4_2048pages_1gpu.json", "bench_c2_upload.json": "r03_bench_c2_upload.json",
    "bench_c2_force_gather.json": "r03_bench_c2_force_gather.json", "bench_c2_inflight4.json": "r03_bench_c2_inflight4.json",
    "timeline_bench_c2.log": "r03_timeline_bench_c2.log", "scan_variants.log": "r03_scan_variants.log", "mfma_shape.log": "r03_mfma_shape.log",
    "fetch_pmc.csv": "r03_pmc_FETCH_SIZE_kbench_c2.csv", "write_pmc.csv": "r03_pmc_WRITE_SIZE_kbench_c2.csv", "mfma_pmc.csv": "r03_pmc_mfma_busy_kbench_c2.csv",
    "insts_pmc.csv": "r03_pmc_insts_kbench_c2.csv", "c5_product_pmc.csv": "r03_pmc_c5_product.csv", "c5_forms_pmc.csv": "r03_pmc_c5_forms.csv",
}
import sys
TRAFFIC_ONLY = "--traffic-only" in sys.argv  # after tools/r3_traffic.sh: only the two PMC passes and profiles/r03_traffic.json
for a, b in cp.items():
    if TRAFFIC_ONLY and a not in ("fetch_pmc.csv", "write_pmc.csv"):
        continue
    shutil.copy(os.path.join(R, a), os.path.join("profiles", b))

def avg(f, kern, ctr):
    v = [float(x["Counter_Value"]) for x in csv.DictReader(open(f)) if kern in x["Kernel_Name"] and x["Counter_Name"] == ctr]
    return sum(v) / len(v), len(v)

def dur_ms(f, kern, ctr):
    v = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) for x in csv.DictReader(open(f)) if kern in x["Kernel_Name"] and x["Counter_Name"] == ctr]
    return sum(v) / len(v) / 1e6

def pm(f, kern):
    out = {c: round(avg(f, kern, c)[0]) for c in ("GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES")}
    d = dur_ms(f, kern, "GRBM_GUI_ACTIVE")
    out["mfma_pipe_busy_frac"] = round(out["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (out["GRBM_GUI_ACTIVE"] / 8), 4)
    out["kernel_ms_under_pmc"] = round(d, 3)
    out["clock_GHz"] = round(out["GRBM_GUI_ACTIVE"] / 8 / d / 1e6, 3)
    return out

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
     "note": "separate rocprofv3 --pmc passes over tools/kbench.py (tools/r3_profiles.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md (HBM section); unit KB",
     "sources": ["profiles/r03_pmc_FETCH_SIZE_kbench_c2.csv", "profiles/r03_pmc_WRITE_SIZE_kbench_c2.csv"]}
json.dump(t, open("profiles/r03_traffic.json", "w"), indent=1)
print("traffic", t["traffic_bytes"], t["FETCH_SIZE_KB_per_launch"], t["WRITE_SIZE_KB_per_launch"])
if TRAFFIC_ONLY:
    sys.exit(0)
prod = json.load(open(R + "/c5_product.json"))
forms = json.load(open(R + "/c5_forms.json"))
json.dump({"what": "BASELINE configs[4], product kernel: int8 MFMA prefilter (scan_mfma2s_kernel) on the 256-template bank, 64 synthetic pages 608x720",
           "run": prod["mfma_gemm"], "workload": prod["workload"], "pmc": pm(R + "/c5_product_pmc.csv", "scan_mfma2s"),
           "source": "tools/c5_gemm_variant.py, tools/r3_profiles.sh"}, open("profiles/r03_c5_i8.json", "w"), indent=1)
json.dump({"what": "BASELINE configs[4], product kernel: exact v_dot4 scan (scan_direct_kernel, the 'LDS-NCC' form) on the same bank and pages",
           "run": prod["dot4_direct"], "workload": prod["workload"], "pmc": pm(R + "/c5_product_pmc.csv", "scan_direct"),
           "source": "tools/c5_gemm_variant.py, tools/r3_profiles.sh"}, open("profiles/r03_c5_dot4.json", "w"), indent=1)
json.dump({"what": "BASELINE configs[4]: the scan's item loop instantiated with v_mfma_i32_16x16x64_i8 and with v_mfma_f32_16x16x32_bf16 over the same noise pages and "
                   "256 zero-sum templates (tools/c5_forms.hip); equal candidate counts",
           "run": forms, "pmc_i8": pm(R + "/c5_forms_pmc.csv", "scan_form<false>"), "pmc_bf16": pm(R + "/c5_forms_pmc.csv", "scan_form<true>"),
           "source": "tools/c5_forms.hip, tools/r3_profiles.sh"}, open("profiles/r03_c5_bf16.json", "w"), indent=1)
print("kbench scan pmc", pm(R + "/mfma_pmc.csv", "scan_mfma2s"))
ins = {c: avg(R + "/insts_pmc.csv", "scan_mfma2s", c)[0] for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY")}
print("insts", {k: round(v) for k, v in ins.items()}, "non-MFMA VALU per item", round((ins["SQ_INSTS_VALU"] - ins["SQ_INSTS_MFMA"]) / (ins["SQ_INSTS_MFMA"] / 192), 1),
      "wait frac", round(ins["SQ_WAIT_ANY"] / ins["SQ_WAVE_CYCLES"], 3))
for k in ("i8", "dot4", "bf16"):
    print(k, json.load(open(f"profiles/r03_c5_{k}.json")).get("pmc") or [json.load(open(f"profiles/r03_c5_{k}.json"))[q] for q in ("pmc_i8", "pmc_bf16")])
for f in ("bench_c2_run", "bench_c2_serial_run", "bench_c2_noise_run", "bench_c3_run", "bench_c4_2048pages_1gpu", "bench_c2_upload"):
    d = json.load(open(f"{R}/{f}.json")); r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["avg_kernel_ms"], r["frac"], r.get("frac_whole_step"), r.get("isolated_avg_kernel_ms"), d.get("e2e_value_incl_h2d"),
          d.get("e2e_value_incl_h2d_pipelined"), r.get("issued_macs_per_launch"))
for f in ("bench_c2", "bench_c2_serial", "bench_c3"):
    rows = list(csv.DictReader(open(f"{R}/{f}_kernel_stats.csv")))
    for r_ in rows:
        if "scan_mfma2s" in r_["Name"]:
            print(f, "scan kernel rocprofv3 avg ms", round(float(r_["AverageNs"]) / 1e6, 4), "calls", r_["Calls"])
