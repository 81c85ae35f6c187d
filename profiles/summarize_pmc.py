"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (collected separately, as
MI355X_MICROARCH.md prescribes: the two counters do not fit one pass) into per-kernel HBM traffic.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half their
size, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.

    python profiles/summarize_pmc.py gpurun_out/pmc_traffic_raw.json profiles/r01_pmc_traffic.json
"""
import json
import re
import sys


def readable(mangled):
    m = re.search(r"gemm_big_kernelIDF16(b|_)Li(\d+)ELi(\d+)ELi(\d+)ELi\d+ELi\d+ELb(\d)", mangled)
    if m:
        return f"gemm_big_kernel<{'bf16' if m.group(1) == 'b' else 'f16'},{m.group(2)},{m.group(3)},{m.group(4)},{'conv' if m.group(5) == '1' else 'lin'}>"
    m = re.search(r"gemm_kernelIDF16(b|_)Li(\d+)ELi(\d+)ELb(\d)", mangled)
    if m:
        return f"gemm_kernel<{'bf16' if m.group(1) == 'b' else 'f16'},{m.group(2)},{m.group(3)},{'conv' if m.group(4) == '1' else 'lin'}>"
    m = re.search(r"dfw\d+([a-z_0-9]+?)I", mangled)
    return m.group(1) if m else mangled[:60]


def main(src, dst):
    raw = json.load(open(src))
    out = {}
    for k, v in raw.items():
        f, w = v.get("FETCH_SIZE"), v.get("WRITE_SIZE")
        if not f or not w:
            continue
        out[readable(k)] = dict(dispatches=f["dispatches"], fetch_kib_per_launch=f["per_dispatch"],
                                write_kib_per_launch=w["per_dispatch"],
                                hbm_bytes_per_launch=(2 * f["per_dispatch"] + w["per_dispatch"]) * 1024)
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["dispatches"])[:10]:
        print(f"{k:45s} x{v['dispatches']:4d}  {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main(*sys.argv[1:3])
