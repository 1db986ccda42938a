#!/usr/bin/env python3
"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_round.sh (pmc_fetch.txt, pmc_write.txt: per-kernel
averages per launch, KB).  bench.py reads the file for `roofline.traffic` (HBM bytes per launch of the dominant kernels).
    python scripts/make_traffic.py profiles/r03_pmc_fetch.txt profiles/r03_pmc_write.txt profiles/r03_pmc_knn_fetch.txt profiles/r03_pmc_sq_insts.txt > profiles/traffic.json
Correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE is doubled (gfx950 tallies 128-byte requests of 16 B / lane streams at 64 B);
WRITE_SIZE as reported."""
import json, re, sys


def table(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+%s\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$" % counter, line)
        if m:
            out[m.group(1).strip()] = (int(m.group(2)), float(m.group(4)))
    return out


def main(fetch_path, write_path, knn_path=None, insts_path=None):
    f, w = table(fetch_path, "FETCH_SIZE"), table(write_path, "WRITE_SIZE")

    def total_kb(name):
        return 2.0 * f.get(name, (0, 0.0))[1] + w.get(name, (0, 0.0))[1]

    def mean_bytes(names):      # average over the launches of the named kernels, weighted by their call counts
        calls = sum(f.get(n, (0, 0))[0] for n in names)
        return 1024.0 * sum(total_kb(n) * f.get(n, (0, 0))[0] for n in names) / max(calls, 1)
    chain = [n for n in f if n.startswith("mlp_chain4_kernel")]
    tn = [n for n in f if n.startswith("gemm_tn_h3_kernel") or n.startswith("gemm_tn_tr")]       # (round 6: + the LDS-DMA / transposing-read kernels)
    out = {
        "source": "%s + %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps 2 --warmup 1; scripts/make_traffic.py)" % (fetch_path, write_path),
        "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B for 16 B/lane streams, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported (uncalibrated)",
        "mlp_chain_bytes_per_launch": mean_bytes(chain),
        "gemm_tn_h3_bytes_per_launch": mean_bytes(tn),
    }
    for n in sorted(chain + tn):
        out[n + " fetch_kb_raw / write_kb"] = [f[n][1], w.get(n, (0, 0.0))[1]]
    if knn_path:
        vals = [float(m.group(1)) for m in re.finditer(r"ray_knn(?:_blocks)?_kernel\S*\s+FETCH_SIZE\s+\d+\s+[\d.]+\s+([\d.]+)", open(knn_path).read())]
        for P, v in zip((10000, 30000), vals):
            out["ray_knn_fetch_kb_raw_P%d" % P] = v
        out["source_knn"] = "%s (scripts/bench_knn.py: R = 25,600 rays, k = 20): the cloud is L2 / Infinity-Cache resident" % knn_path
    if insts_path:      # wave instructions of one ray_knn launch of the bench (P = 10,000): what binds the kernel
        tot = 0.0
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            tot += table(insts_path, c).get("ray_knn_blocks_kernel", (0, 0.0))[1]
        out["ray_knn_wave_insts_per_launch_P10000"] = tot
        out["source_knn_insts"] = "%s (SQ_INSTS_VALU + SALU + LDS + VMEM_RD + VMEM_WR of ray_knn_blocks_kernel, per launch of 25,600 rays)" % insts_path
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
