#!/bin/bash
# Calibration of the "VALU busy" figure: the SQ counters of tools/pmc_kernel.sh over tools/valu_rate's kernels, which issue nothing but
# independent VALU instructions.  4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8) of such a kernel at 2 waves per SIMD (the
# sweep's occupancy) is what "the vector ALU is busy all the time" reads as on this chip; the sweep's own ratio is divided by it.
# usage: tools/pmc_calib.sh   -> gpurun_out/pmc_calib/calib.json
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_calib
rm -rf $O; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $R/tools/valu_rate.hip -o $O/valu_rate || exit 1
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-include-regex 'k_mul_f32|k_fma_f32|k_add_f64|k_cvt_f64_f32' \
    --output-format csv -d $O/p -- $O/valu_rate > $O/run.log 2>&1
python - <<PY
import csv, glob, json, collections
rows = collections.defaultdict(dict)
for f in glob.glob("$O/p/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[(r["Kernel_Name"].split("(")[0], int(r["Dispatch_Id"]), int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))][r["Counter_Name"]] = float(r["Counter_Value"])
out = []
for (k, d, grid), c in sorted(rows.items(), key=lambda kv: kv[0][1]):
    if "GRBM_GUI_ACTIVE" not in c or c["SQ_INSTS_VALU"] < 1e6:      # (the short warm-up launches)
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    out.append({"kernel": k, "grid_threads": grid, "waves_per_simd": grid // (256 * 256) if grid else None, "ratio": 4.0 * c["SQ_ACTIVE_INST_VALU"] / (1024.0 * cyc),
                "valu_wave_insts": c["SQ_INSTS_VALU"], "cycles": cyc, "sq_busy_cycles": c.get("SQ_BUSY_CYCLES")})
sel = [o for o in out if o["kernel"].endswith("k_mul_f32") and o["waves_per_simd"] == 2] or [o for o in out if o["waves_per_simd"] == 2] or out
res = {"kernel": "tools/valu_rate.hip " + sel[0]["kernel"], "waves_per_simd": sel[0]["waves_per_simd"], "saturated_ratio": sel[0]["ratio"], "all": out,
       "what": "4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8) of kernels that issue only independent VALU instructions"}
json.dump(res, open("$O/calib.json", "w"), indent=1)
print(json.dumps({k: res[k] for k in ("kernel", "waves_per_simd", "saturated_ratio")}))
for o in out: print(o["kernel"], o["waves_per_simd"], round(o["ratio"], 3))
PY
rm -rf $O/p
