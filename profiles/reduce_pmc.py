"""Reduce a tools/pmc_kernel.sh summary (mean counter values per dispatch, one block per kernel) to the handful of ratios the
DESIGN tables quote.  usage: reduce_pmc.py <summary.txt>  -> JSON on stdout
Units (MI355X_MICROARCH.md, profiles/README.md): GRBM_GUI_ACTIVE sums the 8 XCDs; SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES / SQ_WAIT_* count
quad-cycles per wave; 1024 SIMDs, 256 vector L1s (TCP), 2.4 GHz."""
import json, re, sys
blocks, cur = {}, None
for l in open(sys.argv[1]).read().splitlines():
    m = re.match(r"--- (\S+)", l)
    if m:
        cur = m.group(1); blocks[cur] = {}; continue
    p = l.split()
    if len(p) >= 2 and p[0][:1].isupper() and p[0].replace("_", "").isalnum():
        try:
            blocks.setdefault(cur or "kernel", {})[p[0]] = float(p[1])
        except ValueError:
            pass
out = {}
for k, c in blocks.items():
    if "GRBM_GUI_ACTIVE" not in c:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    g = lambda n: c.get(n, 0.0)
    out[k] = {
        "duration_us": round(cyc / 2400.0, 2),
        "waves": int(g("SQ_WAVES")),
        "mean_waves_per_simd": round(4.0 * g("SQ_WAVE_CYCLES") / (1024.0 * cyc), 2),
        "valu_busy_frac": round(4.0 * g("SQ_ACTIVE_INST_VALU") / (1024.0 * cyc), 3),
        "valu_wave_insts": int(g("SQ_INSTS_VALU")), "salu_wave_insts": int(g("SQ_INSTS_SALU")), "lds_wave_insts": int(g("SQ_INSTS_LDS")),
        "vmem_rd_wave_insts": int(g("SQ_INSTS_VMEM_RD")), "vmem_wr_wave_insts": int(g("SQ_INSTS_VMEM_WR")),
        "wave_wait_frac": round(g("SQ_WAIT_ANY") / max(1.0, g("SQ_WAVE_CYCLES")), 3),
        "wave_wait_on_lds_frac": round(g("SQ_WAIT_INST_LDS") / max(1.0, g("SQ_WAVE_CYCLES")), 3),
        "lds_bank_conflict_cycles": int(g("SQ_LDS_BANK_CONFLICT")),
        "tcp_busy_frac": round(g("TCP_GATE_EN1_sum") / (256.0 * cyc), 3),
        "tcp_stalled_on_l2_frac_of_busy": round(g("TCP_PENDING_STALL_CYCLES_sum") / max(1.0, g("TCP_GATE_EN1_sum")), 3),
        "tcp_line_accesses": int(g("TCP_TOTAL_CACHE_ACCESSES_sum")), "tcp_to_l2_read_reqs": int(g("TCP_TCC_READ_REQ_sum")),
        "l2_hit_frac": round(g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum")), 3),
        "l2_to_hbm_read_reqs": int(g("TCC_EA0_RDREQ_sum")),
    }
json.dump(out, sys.stdout, indent=1)
