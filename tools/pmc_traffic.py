#!/usr/bin/env python3
"""HBM traffic per launch of each GEMM kernel configuration from two rocprofv3 PMC passes (tools/pmc_traffic.sh):
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half their bytes
(MI355X_MICROARCH.md, HBM section) so it is doubled; WRITE_SIZE is exact for 16-byte streaming stores and f32 atomics.
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/r01/d_pmc_traffic.json"""
import csv, glob, json, os, re, sys, collections


def label(name):
    m = re.search(r"igemm_kernelI(DF16b|f)Li(\d+)ELi(\d+)ELi\d+ELi\d+ELi(\d+)ELi\d+ELi(\d+)ELb[01]", name)
    if m:
        dt, bm, bn, kch, kg = m.groups()
        return f"igemm_kernel<{'bf16' if dt == 'DF16b' else 'f32'},{bm}x{bn},k{int(kch) * 16}B" + (f",kg{kg}" if kg != "1" else "") + ">"
    m = re.search(r"pconv_kernelI(DF16b|f)Li(\d+)ELi(\d+)ELi(\d+)E", name)
    if m:
        dt, slb, tm, tn = m.groups()
        return f"pconv_kernel<{'bf16' if dt == 'DF16b' else 'f32'},slab{slb}B,{int(tm) * 4}frag,{int(tn) * 16}ch>"
    if "wgrad_rows_kernel" in name:              # the line-staged kernel of the 3x3 stride-1 layers: same family as the gathered one
        return "wgrad_kernel<bf16,Cd64>"
    m = re.search(r"wgrad(_halo)?_kernelI(DF16b|f)Li(\d+)ELi\d+E", name)
    if m:
        return f"wgrad{m.group(1) or ''}_kernel<{'bf16' if m.group(2) == 'DF16b' else 'f32'},Cd{m.group(3)}>"
    for k in ( "chan_reduce_kernel", "norm_bwd_apply_kernel", "affine_act_kernel", "adam_kernel",
              "recon_loss_kernel", "pack_tiles_kernel", "flush_unpack_tiles_kernel", "flush_inner_tiles_kernel"):
        if k in name:
            return k
    return None


def collect(d, counter):
    acc = collections.defaultdict(list)
    files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    for f in files[-1:]:                       # gpurun merges every run's files into the same directory: newest only
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                lb = label(r["Kernel_Name"])
                if lb:
                    acc[lb].append(float(r["Counter_Value"]))
    return acc


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"_meta": {"git_head": sys.argv[3] if len(sys.argv) > 3 else "unknown", "collected_by": "tools/pmc_traffic.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"},
       "_note": "bytes per launch, mean over all launches of 3 eager bench steps (B=8,S=2,bf16); fetch = 2 x FETCH_SIZE KiB "
                "(gfx950 half-count correction), write = WRITE_SIZE KiB; split-K configurations are counted with their finish pass excluded"}
for k in sorted(set(fetch) & set(write)):
    f = 2.0 * 1024.0 * sum(fetch[k]) / len(fetch[k])
    w = 1024.0 * sum(write[k]) / len(write[k])
    out[k] = {"launches": len(fetch[k]), "fetch_bytes": f, "write_bytes": w, "traffic_bytes": f + w}
print(json.dumps(out, indent=1))
