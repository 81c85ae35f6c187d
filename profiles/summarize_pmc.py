"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (collected separately, as
MI355X_MICROARCH.md prescribes: the two counters do not fit one pass) into per-kernel HBM traffic.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half their
size, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.

MFMA utilisation (third pass: SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE): the busy counter sums, over all
SIMDs, the cycles their matrix pipe is occupied (32 per v_mfma_f32_32x32x16, 16 per 16x16x32: MI355X_MICROARCH
cycle constants); rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs, so the kernel ran
GRBM_GUI_ACTIVE / 8 shader cycles and
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)
-- the fraction of MFMA issue slots used AT THE CLOCK THE CHIP HELD (unlike TFLOP/s / 2.5 PF, which also
carries the clock).  effective_clock_ghz = GRBM_GUI_ACTIVE / 8 / kernel time is printed by bench-side tools.

    python profiles/summarize_pmc.py gpurun_out/pmc_traffic_raw.json profiles/r02_pmc_traffic.json
    (whole recipe: profiles/run_pmc.sh)
"""
import json
import re
import sys


def readable(mangled):
    m = re.search(r"gemm_big_kernelIDF16(b|_)Li(\d+)ELi(\d+)ELi(\d+)ELi\d+ELi\d+ELb(\d)", mangled)
    if m:
        return f"gemm_big_kernel<{'bf16' if m.group(1) == 'b' else 'f16'},{m.group(2)},{m.group(3)},{m.group(4)},{'conv' if m.group(5) == '1' else 'lin'}>"
    m = re.search(r"gemm8_kernelIDF16(b|_)Lb(\d)ELi(\d+)E", mangled)
    if m:      # the name dfw_gemm_kernel_name() gives bench.py
        return f"gemm8_kernel<{'bf16' if m.group(1) == 'b' else 'f16'},256,{m.group(3)},64,{'conv' if m.group(2) == '1' else 'lin'}>"
    m = re.search(r"gemm8_kernel<bool _Accum, bool, E, (\d+)>", mangled)
    if m:      # rocprofv3's demangler garbles the CONV = true instantiation (the only one whose name it prints demangled)
        return f"gemm8_kernel<bf16,256,{m.group(1)},64,conv>"
    m = re.search(r"conv_patch8_kernelIDF16(b|_)Li(\d+)E", mangled)
    if m:      # the name dfw_gemm_kernel_name() gives bench.py
        return f"conv_patch8_kernel<{'bf16' if m.group(1) == 'b' else 'f16'},256,{m.group(2)}>"
    m = re.search(r"conv_patch_kernelIDF16(b|_)Li(\d+)ELi(\d+)E", mangled)
    if m:      # the name dfw_gemm_kernel_name() gives bench.py
        return f"conv_patch_kernel<{'bf16' if m.group(1) == 'b' else 'f16'},{m.group(2)},{m.group(3)}>"
    m = re.search(r"gemm_kernelIDF16(b|_)Li(\d+)ELi(\d+)ELb(\d)", mangled)
    if m:
        return f"gemm_kernel<{'bf16' if m.group(1) == 'b' else 'f16'},{m.group(2)},{m.group(3)},{'conv' if m.group(4) == '1' else 'lin'}>"
    m = re.search(r"fsa_ring_kernelIDF16(b|_)Li(\d+)ELi(\d+)ELb(\d)", mangled)
    if m:
        return f"fsa_ring_kernel<{'bf16' if m.group(1) == 'b' else 'f16'},{m.group(2)},{m.group(3)},{'pre' if m.group(4) == '1' else 'scale'}>"
    m = re.search(r"dfw\d+([a-z_0-9]+?)I", mangled)
    return m.group(1) if m else mangled[:60]


def main(src, dst):
    raw = json.load(open(src))
    out = {}
    for k, v in raw.items():
        f, w = v.get("FETCH_SIZE"), v.get("WRITE_SIZE")
        if not f or not w:
            continue
        e = dict(dispatches=f["dispatches"], fetch_kib_per_launch=f["per_dispatch"],
                 write_kib_per_launch=w["per_dispatch"],
                 hbm_bytes_per_launch=(2 * f["per_dispatch"] + w["per_dispatch"]) * 1024)
        mb, ga = v.get("SQ_VALU_MFMA_BUSY_CYCLES"), v.get("GRBM_GUI_ACTIVE")
        if mb and ga and ga["per_dispatch"] > 0:
            e["mfma_busy_cycles_per_launch"] = mb["per_dispatch"]
            e["gui_active_per_launch"] = ga["per_dispatch"]
            e["mfma_util"] = round(mb["per_dispatch"] / (ga["per_dispatch"] / 8.0 * 1024.0), 4)
        for extra in ("SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16"):
            if v.get(extra):
                e[extra.lower() + "_per_launch"] = v[extra]["per_dispatch"]
        name = readable(k)
        if name in out:      # template variants folded onto one readable name: keep the one with more launches
            if out[name]["dispatches"] >= e["dispatches"]:
                continue
        out[name] = e
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["dispatches"])[:10]:
        print(f"{k:45s} x{v['dispatches']:4d}  {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main(*sys.argv[1:3])
