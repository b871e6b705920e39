#!/usr/bin/env python3
"""bench_traffic.py DIR TAG: aggregate the FETCH_SIZE / WRITE_SIZE passes of tools/bench_traffic.sh over the 3x3x3 conv
launches of bench.py (conv3d_wf_kernel, conv3d_wino_kernel, conv3d_ll_kernel, conv3d_mfma_kernel<3,..>, conv3d_dma_kernel<3,..>)
and the passes in front of them (prologue_apply_kernel, wino_input_kernel).
FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md "HBM"); both counters are KiB."""
import collections, csv, glob, json, re, sys

out, tag = sys.argv[1], sys.argv[2]
K3 = re.compile(r"conv3d_(mfma|dma)_kernel<3,|conv3d_ll_kernel<|conv3d_wino_kernel<|conv3d_wf_kernel<")
PRO = re.compile(r"prologue_apply_kernel|wino_input_kernel")


def load(sub, counter):
    files = glob.glob(f"{out}/{sub}/*/*_counter_collection.csv") + glob.glob(f"{out}/{sub}/*_counter_collection.csv")
    per = collections.OrderedDict()
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != counter:
            continue
        e = per.setdefault(r["Dispatch_Id"], [r["Kernel_Name"], 0.0])
        e[1] += float(r["Counter_Value"])
    return list(per.values())


def split(rows):
    k3 = [v for n, v in rows if K3.search(n)]
    pro = [v for n, v in rows if PRO.search(n)]
    return k3, pro


f3, fp = split(load("fetch", "FETCH_SIZE"))
w3, wp = split(load("write", "WRITE_SIZE"))
assert len(f3) == len(w3) and len(f3) > 0, (len(f3), len(w3))
n = len(f3)
fetch = 2 * 1024 * (sum(f3) + sum(fp)) / n          # bytes per 3x3x3 launch, x2 gfx950 correction, prologue passes folded in
write = 1024 * (sum(w3) + sum(wp)) / n
# algorithmic bytes of the same launches: every 3x3x3 conv of one B=32 forward reads its input + residual and writes its
# output once (fp32): 9.6 GB over 51 launches
res = {"bytes_per_k3_launch": round(fetch + write), "fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write),
       "launches_sampled": n, "prologue_pass_launches": len(fp),
       "algorithmic_bytes_per_launch": round(9.6e9 / 51),
       "collection": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 "
                     "--warmup 1 --no-extras; all conv3d_wf_kernel / conv3d_wino_kernel / conv3d_ll_kernel / "
                     "conv3d_{mfma,dma}_kernel<3,..> dispatches + the prologue_apply_kernel / wino_input_kernel passes in front of "
                     "them, FETCH_SIZE x2 (gfx950), KiB -> bytes"}
sys.path.insert(0, ".")
import bench                                   # (its module level imports nothing GPU-related)
res["source_sha16"] = bench.source_sha16()     # bench.py reports the figure as stale when the kernel sources have changed
res["ratio_to_algorithmic"] = round(res["bytes_per_k3_launch"] / res["algorithmic_bytes_per_launch"], 3)
json.dump(res, open(f"profiles/{tag}_bench_traffic.json", "w"), indent=1)
json.dump(res, open(f"{out}.json", "w"), indent=1)        # gpurun_out/ copy: the only directory gpurun merges back
print(json.dumps(res))
