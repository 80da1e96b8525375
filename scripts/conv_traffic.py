"""profiles/conv_traffic.json from the two PMC passes (rocpd_pmc.py CSVs of `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`).
usage: conv_traffic.py <fetch_pmc.csv> <write_pmc.csv> <steps_in_the_profiled_run> <out.json> [commit] [sha256 of the library that ran] [sha256 of its sources: _build.source_sha256()]
bench.py reports `roofline.traffic` from this file only when its launch list (`conv_launches_per_step`, `kernel_set`) matches
the run, together with the commit / time recorded here."""
import csv, json, sys, time


def conv_sum(path, counter):
    tot, disp = 0.0, 0
    for r in csv.DictReader(open(path)):
        if any(k in r["kernel"] for k in ("conv3x3_mfma_kernel", "conv_up2x_mfma_kernel", "conv3x3_wino_mfma_kernel", "conv3x3_wino_split_mfma_kernel", "conv3x3_wino_v3_mfma_kernel", "conv3x3_wino_stream_mfma_kernel", "conv3x3_wino_a128_stream_kernel", "conv3x3_wino43_kernel", "conv3x3_wino43s_kernel", "conv_up2x_wino_stream_kernel")) and r["counter"] == counter:
            tot += float(r["sum"]); disp += int(r["dispatches"])
    return tot, disp


# bytes the 20 conv launches of a batch-10 288x512 step must write: (4 x 64 ch @288x512 + 4 x 128 @144x256 + 6 x 256 @72x128 + 3 x 512 @36x64
# layer outputs) + the three decoder-entry partial sums (256 @72x128, 128 @144x256, 64 @288x512), fp32
# (+ round 4: the three pooled tensors the down blocks' last layers write from their write-out: 64 @144x256, 128 @72x128, 256 @36x64)
KNOWN_WRITE = 10 * 4.0 * (4 * 64 * 288 * 512 + 4 * 128 * 144 * 256 + 6 * 256 * 72 * 128 + 3 * 512 * 36 * 64 + 256 * 72 * 128 + 128 * 144 * 256 + 64 * 288 * 512
                          + 64 * 144 * 256 + 128 * 72 * 128 + 256 * 36 * 64)


def main():
    fetch_csv, write_csv, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    f, nf = conv_sum(fetch_csv, "FETCH_SIZE")
    w, nw = conv_sum(write_csv, "WRITE_SIZE")
    assert nf == nw and nf % steps == 0, (nf, nw, steps)
    launches = nf // steps                        # 17 conv layers = 20 launches (decoder-entry layers are two launches each)
    fetch_raw = f * 1024 / steps                  # counters are in KiB
    write = w * 1024 / steps
    fetch = 2.0 * fetch_raw                       # MI355X_MICROARCH.md: gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes
    json.dump({
        "commit": sys.argv[5] if len(sys.argv) > 5 else None, "taken_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
        "kernel_set": "wino43s+up2x_wino43s", "lib_sha256": sys.argv[6] if len(sys.argv) > 6 else None,
        "src_sha256": sys.argv[7] if len(sys.argv) > 7 else None,      # path-independent key of the build: what bench.py matches
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), "
                  f"bench.py --steps {steps - 1} --warmup 1 --blocks 1 --overlap-streams 0 --infer-split 0, MI355X (raw per-kernel sums: "
                  "the two CSVs given on the command line; made by scripts/conv_traffic.py)",
        "counters_unit": "KiB (x1024 bytes)", "conv_launches_per_step": launches,
        "fetch_size_raw_bytes_per_step": round(fetch_raw, -6), "fetch_size_corrected_bytes_per_step": round(fetch, -6),
        "write_size_bytes_per_step": round(write, -6),
        "correction": "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes -> doubled; WRITE_SIZE is exact: "
                      f"the 17 layer outputs + 3 decoder partial sums of a batch-10 step are {KNOWN_WRITE:.4e} bytes, measured {write / KNOWN_WRITE - 1:+.1%}",
        "traffic_bytes_per_launch": round((fetch + write) / launches, -5),
        "algorithmic_bytes_per_step": 6935500000.0,          # 693.55 MB per sample x 10
        "note": "includes the partial-sum tensors the decoder-entry layers write and re-read; the kernels are MFMA-bound "
                "(under 0.6 TB/s of the 8 TB/s HBM roof), so traffic is not the limiter",
    }, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
