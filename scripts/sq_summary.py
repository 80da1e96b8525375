#!/usr/bin/env python
"""Per-kernel SQ / TCC counter summary (the rows of profiles/r0N_wino_sq_counters.md) from the four rocpd_pmc.py CSVs of
scripts/gpu_pmc_sq.sh.  usage: sq_summary.py <dir with pmc_sq1.csv pmc_sq2.csv pmc_sq3.csv pmc_tcc.csv> [steps_profiled]
MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)  (as in r01 / r02: the counter sums over SIMDs,
GRBM_GUI_ACTIVE over the 8 XCDs)."""
import csv, json, os, sys


def load(path):
    d = {}
    for r in csv.DictReader(open(path)):
        d.setdefault(r["kernel"], {})[r["counter"]] = (float(r["sum"]), int(r["dispatches"]))
    return d


def main():
    base = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    tabs = {}
    for f in ("pmc_sq1.csv", "pmc_sq2.csv", "pmc_sq3.csv", "pmc_tcc.csv"):
        for k, v in load(os.path.join(base, f)).items():
            tabs.setdefault(k, {}).update(v)
    out = {}
    for k, c in tabs.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c or not any(s in k for s in ("wino", "up2x", "conv3x3_mfma")):
            continue
        g = lambda n: c.get(n, (0.0, 0))[0]          # noqa: E731
        mfma = g("SQ_INSTS_MFMA") or g("SQ_INSTS_VALU_MFMA_MOPS_F32") / 512.0
        wave = g("SQ_WAVE_CYCLES")
        row = {
            "launches_per_step": c["GRBM_GUI_ACTIVE"][1] // steps,
            "mfma_busy_frac": round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") / 8.0 * 1024.0), 4),
            "wave_cycles_issue_wait_parked_active": [round(g("SQ_WAIT_INST_ANY") / wave, 3), round(g("SQ_WAIT_ANY") / wave, 3),
                                                     round(g("SQ_ACTIVE_INST_ANY") / wave, 3)] if wave else None,
            "per_mfma_valu_lds_salu_vmem": [round((g("SQ_INSTS_VALU") - g("SQ_INSTS_MFMA")) / g("SQ_INSTS_MFMA"), 2),
                                            round(g("SQ_INSTS_LDS") / g("SQ_INSTS_MFMA"), 2), round(g("SQ_INSTS_SALU") / g("SQ_INSTS_MFMA"), 2),
                                            round(g("SQ_INSTS_VMEM_RD") / g("SQ_INSTS_MFMA"), 2)] if g("SQ_INSTS_MFMA") else None,
            "lds_bank_conflict_cycles": g("SQ_LDS_BANK_CONFLICT"), "lds_active_cycles": g("SQ_LDS_IDX_ACTIVE"),
            "lds_bank_conflict_share": round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 4) if g("SQ_LDS_IDX_ACTIVE") else None,
            "l2_hit": round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 3) if g("TCC_HIT_sum") else None,
            "mfma_instructions": g("SQ_INSTS_MFMA"), "gui_active_cycles": g("GRBM_GUI_ACTIVE"),
        }
        out[k] = row
    try:                                              # which build the counters describe (scripts/gpu_session.sh writes it next to them)
        out["lib_sha256"] = open(os.path.join(base, "lib_sha256.txt")).read().strip()
    except OSError:
        out["lib_sha256"] = None
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
