"""Turns the raw rocprofv3 output of tools/collect_profiles.sh (gpurun_out/final/) into the committed profile files.
usage: python profiles/summarize.py [round_tag]   (default r01)"""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = sys.argv[2] if len(sys.argv) > 2 else HERE      # (on the GPU box: a directory under gpurun_out/, copied into profiles/ afterwards)
os.makedirs(OUT, exist_ok=True)


def one(pattern):
    f = glob.glob(os.path.join(SRC, pattern), recursive=True)
    assert f, pattern
    return max(f, key=os.path.getmtime)       # merged scratch directories may still hold an older run's files


def load_line(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])


def pmc_summary(path):
    d = {}
    for l in open(path).read().splitlines():
        p = l.split()
        if len(p) >= 2 and p[0][:1].isupper() and p[0].replace("_", "").isalnum():
            try:
                d[p[0]] = float(p[1])
            except ValueError:
                pass
    return d


import traceback
traffic = {"traffic_bytes_per_launch": 0.0}
fe, wr, sw, kmax = [0.0], [0.0], [], 0


def section(fn):
    try:
        fn()
    except Exception:
        print("section", fn.__name__, "skipped:", traceback.format_exc().splitlines()[-1])


def sec1():
    global traffic, fe, wr, sw, kmax, bench
    # ---- kernel stats (engine kernels only; torch kernels of the synthetic data generator are summed into one line)
    rows = list(csv.DictReader(open(one("kt/**/*_kernel_stats.csv"))))
    ours, other_calls, other_ns = [], 0, 0.0
    for r in rows:
        n = r["Name"]
        if any(k in n for k in ("k_sweep", "k_align_async", "k_async_begin", "k_async_prepare", "k_stream_status", "k_stream_inputs", "k_build_check", "k_sorted_points", "k_calc_score", "k_update", "k_leafsum", "k_rs_", "k_keys", "k_mark", "k_segstart", "k_minmax", "k_voxels", "k_rank",
                                "k_init_state", "k_griddesc", "k_seq_", "k_set_word_off", "k_word_offsets", "k_pose_records", "k_deinterleave", "k_hessian", "k_fitness", "k_cellrange", "k_transform", "rocprim",
                                "rocclr")):
            ours.append(r)
        else:
            other_calls += int(r["Calls"]); other_ns += float(r["TotalDurationNs"])
    with open(os.path.join(OUT, f"{TAG}_final_kernel_stats.csv"), "w") as f:
        f.write("kernel,calls,total_us,avg_us,min_us,max_us\n")
        for r in ours:
            f.write(f"\"{r['Name'][:110]}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e3:.1f},{float(r['AverageNs']) / 1e3:.3f},"
                    f"{float(r['MinNs']) / 1e3:.2f},{float(r['MaxNs']) / 1e3:.2f}\n")
        f.write(f"\"(torch kernels: synthetic data generation, outside the path)\",{other_calls},{other_ns / 1e3:.1f},,,\n")


section(sec1)

def sec2():
    global traffic, fe, wr, sw, kmax, bench
    # ---- every sweep dispatch
    tr = list(csv.DictReader(open(one("kt/**/*_kernel_trace.csv"))))
    sw = [r for r in tr if "k_sweep" in r["Kernel_Name"] or "k_align_async" in r["Kernel_Name"]]
    sw.sort(key=lambda r: int(r["Start_Timestamp"]))
    with open(os.path.join(OUT, f"{TAG}_final_sweep_dispatches.csv"), "w") as f:
        f.write("dispatch,duration_us,grid_size,vgpr,sgpr,lds_bytes,scratch\n")
        for k, r in enumerate(sw):
            f.write(f"{k},{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.2f},{r.get('Grid_Size', r.get('Grid_Size_X', ''))},"
                    f"{r.get('VGPR_Count', '')},{r.get('SGPR_Count', '')},{r.get('LDS_Block_Size', '')},{r.get('Scratch_Size', '')}\n")
    # the same dispatches by launch size: the streamed job's launches leave workgroup slots to the next batch's build (a smaller grid) and run beside
    # it; the synchronous job's (and a stream's flushes) have the GPU to themselves -- bench.py's `roofline` is the first kind, `roofline_synchronous` the second
    kinds = {}
    for r in sw:
        kinds.setdefault(str(r.get('Grid_Size', r.get('Grid_Size_X', ''))), []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    with open(os.path.join(OUT, f"{TAG}_final_sweep_launch_kinds.csv"), "w") as f:
        f.write("grid_size_threads,dispatches,median_us,full_launches,avg_us_of_full_launches\n")
        for g, d in sorted(kinds.items(), key=lambda kv: -len(kv[1])):
            d.sort()
            med = d[len(d) // 2]
            full = [x for x in d if x >= 0.5 * med]
            f.write(f"{g},{len(d)},{med:.2f},{len(full)},{sum(full) / len(full):.2f}\n")


section(sec2)

TRAFFIC_CFGS = (("cfg3", "271x65536:omp:direct7:1.0", "bench.py defaults (271 pairs x 65536 pts, ndt_omp, 1.0 m, DIRECT7)", "{TAG}_traffic.json"),
                ("pca_direct1", "271x65536:pca:direct1:1.0", "the nodelet's registration (271 pairs x 65536 pts, ndt_pca, 1.0 m, DIRECT1)", "{TAG}_traffic_pca_direct1.json"),
                ("cfg5_d7", "128x131072:pca:direct7:0.5", "BASELINE config 5's per-GPU share (128 pairs x 131072 pts, ndt_pca, 0.5 m, DIRECT7)", "{TAG}_traffic_cfg5_d7.json"),
                ("cfg5_d1", "128x131072:pca:direct1:0.5", "BASELINE config 5's per-GPU share with DIRECT1 (128 pairs x 131072 pts, ndt_pca, 0.5 m)", "{TAG}_traffic_cfg5_d1.json"))
TRAFFIC = {}


def sec3():
    global traffic, fe, wr, sw, kmax, bench
    # ---- PMC passes: FETCH_SIZE / WRITE_SIZE per sweep dispatch (rocprofv3 reports them in KB), every timed configuration, both arithmetics
    def counter(dirname, name):
        rows = list(csv.DictReader(open(one(f"{dirname}/**/*_counter_collection.csv"))))
        rows = [r for r in rows if r["Counter_Name"] == name and ("k_sweep" in r["Kernel_Name"] or "k_align_async" in r["Kernel_Name"])]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        return [float(r["Counter_Value"]) for r in rows]

    for sfx, arith in (("", 0), ("_tol", 1)):
        for tag, key, wl, outname in TRAFFIC_CFGS:
            try:
                f_, w_ = counter(f"fetch_{tag}{sfx}", "FETCH_SIZE"), counter(f"write_{tag}{sfx}", "WRITE_SIZE")
            except AssertionError:
                continue
            n = min(len(f_), len(w_))
            if n == 0:
                continue
            b = load_line(os.path.join(SRC, f"traffic_bench_{tag}{sfx}.json"))
            rs = b.get("roofline_synchronous") or b["roofline"]
            fe_avg, wr_avg = sum(f_[:n]) / n, sum(w_[:n]) / n
            k = max(range(n), key=lambda i: f_[i])
            t = {"kernel": "k_align_async (one launch per batch align: derivative sweeps + Newton updates)", "workload": wl, "workload_key": key, "arith": arith,
                 "launches_counted": n, "fetch_size_kb_avg_per_launch": fe_avg, "write_size_kb_avg_per_launch": wr_avg, "fetch_correction": 2.0,
                 "traffic_bytes_per_launch": (2.0 * fe_avg + wr_avg) * 1024, "traffic_bytes_per_launch_uncorrected": (fe_avg + wr_avg) * 1024,
                 "full_launch": {"fetch_kb": f_[k], "write_kb": w_[k]},
                 "algorithmic_bytes_per_launch": rs["alg_bytes_per_launch"], "avg_launch_us_of_that_run": rs["avg_launch_us"],
                 "physical_hbm_gbs": round((2.0 * fe_avg + wr_avg) * 1024 / (rs["avg_launch_us"] * 1e-6) / 1e9, 1),
                 "physical_hbm_frac_of_peak": round((2.0 * fe_avg + wr_avg) * 1024 / (rs["avg_launch_us"] * 1e-6) / 8.0e12, 4),
                 "physical_over_algorithmic": round((2.0 * fe_avg + wr_avg) * 1024 / rs["alg_bytes_per_launch"], 4),
                 "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --cpu-seconds 0 --no-stream --steps 4 --warmup 1 <config>`, "
                           "--kernel-include-regex 'k_sweep|k_align_async', no trace domains; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE x2 is the gfx950 "
                           "half-counting correction of MI355X_MICROARCH.md (HBM section).  One launch = one whole batch align (every sweep of every pair); the launch time is "
                           "the same run's (HIP events inside the engine, under the counter pass)."}
            TRAFFIC[(tag, arith)] = t
            name = outname.replace("{TAG}", TAG)
            if arith:
                name = name.replace("_traffic", "_traffic_tol")
            json.dump(t, open(os.path.join(OUT, name), "w"), indent=1)
            if tag == "cfg3" and arith == 0:
                traffic, fe, wr, kmax = t, f_, w_, k
                with open(os.path.join(OUT, f"{TAG}_final_pmc_sweep.csv"), "w") as f:
                    f.write("dispatch,FETCH_SIZE_KB,WRITE_SIZE_KB\n")
                    for i in range(n):
                        f.write(f"{i},{f_[i]},{w_[i]}\n")
    bench = load_line(os.path.join(SRC, "kt_bench.json"))


section(sec3)

def valu_calibration():
    """tools/pmc_calib.sh: the SQ counters over tools/valu_rate's kernels (nothing but independent VALU instructions).  What they show: on gfx950
    SQ_ACTIVE_INST_VALU is 4 x the instruction count whatever the instruction (plain f32 at 2+ waves per SIMD: one instruction per ~2.5 cycles,
    counter ratio 1.6; f64 add / packed f32 / cvt: one per ~4.4-5.0 cycles, ratio 0.8-0.9) -- it is no busy-time counter.  The calibrated figure
    is therefore built from elapsed cycles: instructions x the MEASURED minimum issue interval of a plain f32 instruction at the sweep's
    occupancy, over the elapsed SIMD cycles -- a floor of the VALU's busy fraction that cannot exceed 1."""
    pth = os.path.join(SRC, "pmc_valu_calib.json")
    if not os.path.exists(pth):
        return None
    raw = json.load(open(pth))
    timed = [o for o in raw["all"] if o["valu_wave_insts"] > 1e9 and o["waves_per_simd"] == 2]      # (the 10-iteration warm-up launches are dropped)
    import re
    short = lambda k: re.sub(r"^(void )?k_", "", k.strip())
    cpi = {short(o["kernel"]): o["cycles"] * 1024.0 / o["valu_wave_insts"] for o in timed}
    cal = {"waves_per_simd": 2, "cycles_per_wave_instruction_per_simd": {k: round(v, 3) for k, v in cpi.items()},
           "counter_ratio_4xACTIVE_over_simd_cycles": {short(o["kernel"]): round(o["ratio"], 3) for o in timed},
           "cpi_plain_f32": min(cpi.get("mul_f32", 9e9), cpi.get("fma_f32", 9e9)), "cpi_f64_add": cpi.get("add_f64"), "cpi_cvt_f64_f32": cpi.get("cvt_f64_f32"),
           "what": "tools/valu_rate.hip under rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE (tools/pmc_calib.sh); cycles = GRBM_GUI_ACTIVE / 8; 1024 SIMDs",
           "all": raw["all"]}
    json.dump(cal, open(os.path.join(OUT, f"{TAG}_valu_calibration.json"), "w"), indent=1)
    return cal


def sec4():
    global traffic, fe, wr, sw, kmax, bench
    cal = valu_calibration()
    # ---- SQ counters of the sweep (tools/pmc_kernel.sh summaries) -> what bounds it: VALU issue
    for tag, key, benchfile, outname in (("sq_direct7", "271x65536:omp:direct7:1.0", "bench.json", f"{TAG}_valu.json"),
                                         ("sq_pca_direct1", "271x65536:pca:direct1:1.0", "bench_pca_d1.json", f"{TAG}_valu_pca_direct1.json"),
                                         ("sq_cfg5_d1", "128x131072:pca:direct1:0.5", "bench_cfg5_d1.json", f"{TAG}_valu_cfg5_d1.json"),
                                         ("sq_cfg5_d7", "128x131072:pca:direct7:0.5", "bench_cfg5.json", f"{TAG}_valu_cfg5_d7.json")):
        pth = os.path.join(SRC, f"pmc_{tag}.txt")
        if not os.path.exists(pth) or not os.path.exists(os.path.join(SRC, benchfile)):
            continue
        open(os.path.join(OUT, f"{TAG}_pmc_{tag}.txt"), "w").write(open(pth).read())
        c = pmc_summary(pth)
        if "SQ_ACTIVE_INST_VALU" not in c:
            continue
        b = load_line(os.path.join(SRC, benchfile))
        rs = b.get("roofline_synchronous") or b["roofline"]          # (the counter passes run the synchronous job: its launches)
        hits_per_launch = b["roofline"]["hits_per_point"] * (rs["alg_bytes_per_launch"] / (12 + 4 * {"direct7": 7, "direct1": 1}[b["config"]["neighbor_mode"]] + 64 * b["roofline"]["hits_per_point"]))
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0                       # the counter sums the eight XCDs
        valu = {
            "workload_key": key, "kernel": "k_align_async", "source_counters": f"profiles/{TAG}_pmc_{tag}.txt (rocprofv3 --pmc passes of tools/pmc_kernel.sh, mean per dispatch)",
            "kernel_cycles_mean_per_dispatch": cycles,
            "valu_active_frac": round(4.0 * c["SQ_ACTIVE_INST_VALU"] / (1024.0 * cycles), 3),      # SQ_ACTIVE_INST_VALU counts quad-cycles; 1024 SIMDs
            # floor of the VALU busy fraction, <= 1 by construction: every instruction charged the measured issue interval of the CHEAPEST class
            "valu_active_frac_calibrated": round(c["SQ_INSTS_VALU"] * cal["cpi_plain_f32"] / (1024.0 * cycles), 3) if cal else None,
            # the same with the evaluation's f64 adds and f32->f64 conversions (46 + 49 of ~460 wave-instructions per 64-hit batch, from the ISA) at their own intervals
            "valu_active_frac_mix_estimate": round(c["SQ_INSTS_VALU"] * ((1 - 95.0 / 460.0) * cal["cpi_plain_f32"] + (46.0 / 460.0) * cal["cpi_f64_add"] + (49.0 / 460.0) * cal["cpi_cvt_f64_f32"]) / (1024.0 * cycles), 3) if cal else None,
            "calibration": (f"SQ_INSTS_VALU x {cal['cpi_plain_f32']:.2f} cycles (measured issue interval of a plain f32 VALU instruction at 2 waves per SIMD, tools/valu_rate.hip under the same "
                            f"counters: profiles/{TAG}_valu_calibration.json) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); the raw `valu_active_frac` = 4 x SQ_ACTIVE_INST_VALU / SIMD cycles "
                            "is 4 x the instruction count on this chip and exceeds 1 for f32-heavy code") if cal else None,
            "valu_wave_insts_per_64_hits": round(64.0 * c["SQ_INSTS_VALU"] / hits_per_launch, 1),  # one wave-instruction serves 64 (point, voxel) evaluations
            "valu_lane_insts_per_hit": round(c["SQ_INSTS_VALU"] / hits_per_launch, 2),
            "wave_wait_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
            "tcp_busy_frac": round(c.get("TCP_GATE_EN1_sum", 0.0) / (256.0 * cycles), 3),
            "tcp_line_accesses_per_dispatch": c.get("TCP_TOTAL_CACHE_ACCESSES_sum"),
            "l2_hit_frac": round(c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3) if "TCC_HIT_sum" in c else None,
            "physical_hbm_frac_of_peak": (TRAFFIC.get(({"sq_direct7": "cfg3", "sq_pca_direct1": "pca_direct1", "sq_cfg5_d1": "cfg5_d1", "sq_cfg5_d7": "cfg5_d7"}[tag], 0)) or {}).get("physical_hbm_frac_of_peak"),
            "physical_hbm_source": "profiles/" + {"sq_direct7": f"{TAG}_traffic.json", "sq_pca_direct1": f"{TAG}_traffic_pca_direct1.json", "sq_cfg5_d1": f"{TAG}_traffic_cfg5_d1.json", "sq_cfg5_d7": f"{TAG}_traffic_cfg5_d7.json"}[tag],
            "note": "mean over the k_align_async dispatches of a short bench run (one dispatch = one batch align: every sweep and every Newton update of every pair); GRBM_GUI_ACTIVE / 8 as elapsed cycles",
        }
        json.dump(valu, open(os.path.join(OUT, outname), "w"), indent=1)


section(sec4)

def sec5():
    global traffic, fe, wr, sw, kmax, bench
    # ---- build / update kernels: counter summaries + reduced ratios (profiles/reduce_pmc.py)
    import subprocess
    for name in ("build", "update", "build_sorted"):
        pth = os.path.join(SRC, f"pmc_{name}.txt")
        if os.path.exists(pth) and os.path.getsize(pth) > 100:
            open(os.path.join(OUT, f"{TAG}_pmc_{name}.txt"), "w").write(open(pth).read())
            red = subprocess.run([sys.executable, os.path.join(HERE, "reduce_pmc.py"), pth], capture_output=True, text=True).stdout
            open(os.path.join(OUT, f"{TAG}_{name}_counters.json"), "w").write(red)

    for name in ("kstats_pca_d1.txt", "kstats_cfg5_d1.txt"):
        pth = os.path.join(SRC, name)
        if os.path.exists(pth):
            txt = [l for l in open(pth).read().splitlines() if l.startswith("kernel ") or l.startswith("k_") or l.startswith("void k_")]
            open(os.path.join(OUT, f"{TAG}_{name}"), "w").write("\n".join(txt) + "\n")

    for src, dst in (("bench_cfg5_d1.json", f"{TAG}_bench_cfg5_d1.json"), ("bench_cfg4_1gpu.json", f"{TAG}_bench_cfg4_1gpu.json"),
                     ("bench_2ranks_1gpu_gloo.json", f"{TAG}_bench_2ranks_1gpu_gloo.json"), ("bench.json", f"{TAG}_bench.json"), ("bench_pca_d1.json", f"{TAG}_bench_pca_d1.json"),
                     ("bench_cfg5.json", f"{TAG}_bench_cfg5.json"), ("bench_1536.json", f"{TAG}_bench_1536pairs.json"), ("bench_prefiltered.json", f"{TAG}_bench_prefiltered.json"),
                     ("kt_bench.json", f"{TAG}_bench_under_rocprof.json")):
        p = os.path.join(SRC, src)
        if os.path.exists(p) and os.path.getsize(p) > 10:
            json.dump(load_line(p), open(os.path.join(OUT, dst), "w"), indent=1)
    for src, dst in (("upload_rate.txt", f"{TAG}_upload_rate.txt"), ("latency.txt", f"{TAG}_latency.txt"), ("sequence.txt", f"{TAG}_sequence.txt")):
        if os.path.exists(os.path.join(SRC, src)):
            open(os.path.join(OUT, dst), "w").write(open(os.path.join(SRC, src)).read())
    print("sweep launches", len(sw), "avg us", sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in sw) / 1e3 / max(1, len(sw)))
    print("traffic per launch MB", traffic["traffic_bytes_per_launch"] / 1e6, "full launch fetch/write KB", fe[kmax], wr[kmax])


section(sec5)


def sec6():
    # ---- the tolerance arithmetic (MI355NDT_ARITH=1): counters of its launches for the same four configurations, its build, its kernel times
    import subprocess
    for tag, key, tcfg in (("sq_tol_direct7", "271x65536:omp:direct7:1.0", "cfg3"), ("sq_tol_pca_direct1", "271x65536:pca:direct1:1.0", "pca_direct1"),
                           ("sq_tol_cfg5_d1", "128x131072:pca:direct1:0.5", "cfg5_d1"), ("sq_tol_cfg5_d7", "128x131072:pca:direct7:0.5", "cfg5_d7")):
        pth = os.path.join(SRC, f"pmc_{tag}.txt")
        if not os.path.exists(pth) or os.path.getsize(pth) < 100:
            continue
        open(os.path.join(OUT, f"{TAG}_pmc_{tag}.txt"), "w").write(open(pth).read())
        red = json.loads(subprocess.run([sys.executable, os.path.join(HERE, "reduce_pmc.py"), pth], capture_output=True, text=True).stdout or "{}")
        r = next(iter(red.values()), None)
        if not r:
            continue
        t = TRAFFIC.get((tcfg, 1))
        r.update({"workload_key": key, "arith": 1, "kernel": "k_align_async<., ., 2> (tolerance arithmetic)",
                  "valu_busy_frac_what": "4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x cycles): VALU-busy share of the launch (1.0 = the vector ALUs never idle)",
                  "valu_wave_insts_per_64_hits": round(64.0 * r["valu_wave_insts"] / (t["algorithmic_bytes_per_launch"] / 1.0), 6) if False else None,
                  "physical_hbm_frac_of_peak": t["physical_hbm_frac_of_peak"] if t else None, "physical_over_algorithmic": t["physical_over_algorithmic"] if t else None})
        if t:
            K = 7 if "direct7" in key else 1
            b = load_line(os.path.join(SRC, f"traffic_bench_{tcfg}_tol.json"))
            hpp = b["roofline"]["hits_per_point"]
            hits = hpp * t["algorithmic_bytes_per_launch"] / (12 + 4 * K + 64 * hpp)
            r["valu_wave_insts_per_64_hits"] = round(64.0 * r["valu_wave_insts"] / hits, 1)
        json.dump(r, open(os.path.join(OUT, f"{TAG}_valu_tol_{tcfg}.json"), "w"), indent=1)
    pth = os.path.join(SRC, "pmc_tol_build.txt")
    if os.path.exists(pth) and os.path.getsize(pth) > 100:
        open(os.path.join(OUT, f"{TAG}_pmc_tol_build.txt"), "w").write(open(pth).read())
        open(os.path.join(OUT, f"{TAG}_tol_build_counters.json"), "w").write(subprocess.run([sys.executable, os.path.join(HERE, "reduce_pmc.py"), pth], capture_output=True, text=True).stdout)
    pth = os.path.join(SRC, "kernel_stats_tol.csv")
    if os.path.exists(pth):
        rows = list(csv.DictReader(open(pth)))
        with open(os.path.join(OUT, f"{TAG}_final_kernel_stats_tol.csv"), "w") as f:
            f.write("kernel,calls,total_us,avg_us,min_us,max_us\n")
            for r in rows:
                if any(k in r["Name"] for k in ("k_align_async", "k_sweep", "k_async_prepare", "k_stream_", "k_leafsum", "k_rs_", "k_mark", "k_minmax", "k_voxels", "k_rank", "k_griddesc", "k_word_offsets", "k_build_check", "k_update")):
                    f.write(f"\"{r['Name'][:110]}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e3:.1f},{float(r['AverageNs']) / 1e3:.3f},{float(r['MinNs']) / 1e3:.2f},{float(r['MaxNs']) / 1e3:.2f}\n")
    p = os.path.join(SRC, "kt_tol_bench.json")
    if os.path.exists(p) and os.path.getsize(p) > 10:
        json.dump(load_line(p), open(os.path.join(OUT, f"{TAG}_bench_tol_under_rocprof.json"), "w"), indent=1)


section(sec6)

