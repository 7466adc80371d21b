#!/usr/bin/env python3
"""Copy what the judge reads out of a tools/profile_bench.sh output directory into profiles/ (tracked):
  profiles/<round>_<tag>_rocprof_summary.txt      header (dominant kernel: kernel-trace average next to the bench's HIP events of the same
                                                  run, algorithmic GB/s) + summary.txt of tools/profile_summary.py
  profiles/<round>_<tag>_kernel_stats.csv         rocprofv3 --stats per-kernel table
  profiles/<round>_<tag>_bench_under_rocprof.json the bench line printed under the profiler
and refresh the dominant kernel's entry of profiles/traffic.json from the separate FETCH_SIZE / WRITE_SIZE passes (units and the
gfx950 correction as /opt/skills/guides/MI355X_MICROARCH.md prescribes: KiB; read bytes = 2 * FETCH_SIZE * 1024).
usage: tools/profile_commit.py ROUND TAG [TAG...]      e.g.  tools/profile_commit.py r03 3d7_1024_default 3d7_512"""
import csv
import glob
import json
import re
import shutil
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "tools"))


def main():
    rnd, tags = sys.argv[1], sys.argv[2:]
    tfile = REPO / "profiles" / "traffic.json"
    traffic = json.loads(tfile.read_text())
    for tag in tags:
        d = REPO / "gpurun_out" / f"prof_{tag}"
        bench = json.loads((d / "bench_stats.json").read_text())
        summary = (d / "summary.txt").read_text()
        stats = sorted(glob.glob(str(d / "stats" / "**" / "*_kernel_stats.csv"), recursive=True))[0]
        cfg = bench["config"]
        variant, chunk = cfg["variant"], cfg["chunk"]
        # the dominant kernel's line of the summary: the tile the bench reports as launched
        line = next(l for l in summary.split("\n") if f"[{variant}]" in l and "calls=" in l)
        calls = int(re.search(r"calls=(\d+)", line).group(1))
        avg = float(re.search(r"avg_ns=([0-9.]+)", line).group(1)) / 1e6
        mn = float(re.search(r"min_ns=([0-9.]+)", line).group(1)) / 1e6
        kname = line.split()[0]
        alg = bench["roofline"]["algorithmic_bytes_per_launch"]
        ev = bench["roofline"]["kernel_ms"]
        cmdline = (d / "README.txt").read_text().split("\n")[0] if (d / "README.txt").exists() else ""
        hdr = (f"# Round {rnd[1:].lstrip('0')}, {tag}: bench.py under rocprofv3 --kernel-trace --stats, then one --pmc pass per counter group "
               f"(tools/profile_bench.sh via tools/profile_all.sh; module cache filled before the profiler starts)\n"
               f"# dominant kernel: {kname}, chunk {chunk} planes: kernel-trace average {avg:.4f} ms over {calls} dispatches (min {mn:.4f}; the dispatches "
               f"include the plan-time tuner's); bench HIP events in the same run {ev:.4f} ms per step\n"
               f"# algorithmic {alg / 1e9:.3f} GB per launch -> {alg / avg / 1e6:.1f} GB/s = {alg / avg / 1e6 / 80:.1f} % of 8 TB/s (kernel trace); "
               f"{alg / ev / 1e6:.1f} GB/s = {alg / ev / 1e6 / 80:.1f} % (bench events)\n")
        (REPO / "profiles" / f"{rnd}_{tag}_rocprof_summary.txt").write_text(hdr + summary)
        shutil.copy(stats, REPO / "profiles" / f"{rnd}_{tag}_kernel_stats.csv")
        (REPO / "profiles" / f"{rnd}_{tag}_bench_under_rocprof.json").write_text(json.dumps(bench) + "\n")
        # traffic of the dominant kernel
        fetch = write = None
        for l in summary.split("\n"):
            if kname in l and " FETCH_SIZE " in l:
                fetch = float(re.search(r"per_dispatch=([0-9.e+]+)", l).group(1))
            if kname in l and " WRITE_SIZE " in l:
                write = float(re.search(r"per_dispatch=([0-9.e+]+)", l).group(1))
        if fetch is not None and write is not None:
            wl = tag.replace("_default", "")
            tb = 2 * fetch * 1024 + write * 1024
            traffic["entries"][f"{wl}|neptune_apply_march|{variant}"] = {
                "fetch_size_kib_per_launch": fetch, "write_size_kib_per_launch": write, "traffic_bytes_per_launch": tb,
                "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": tb / alg,
                "source": f"profiles/{rnd}_{tag}_rocprof_summary.txt"}
            print(f"{tag}: {kname} avg {avg:.4f} ms ({alg / avg / 1e6 / 80:.1f} %), events {ev:.4f} ms, traffic {tb / alg:.3f}x algorithmic")
        else:
            print(f"{tag}: {kname} avg {avg:.4f} ms, events {ev:.4f} ms, no traffic counters found")
    tfile.write_text(json.dumps(traffic, indent=1) + "\n")


if __name__ == "__main__":
    main()
